"""The N > 1 path on CPU: 2 processes, gloo.  The per-rank render is substituted by the oracle
restricted to the rank's tiles (the HIP kernels need a GPU); partition, reduce and develop are the
product's own code (liverrenderer_amd/distributed.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_tile_partition_is_disjoint_and_complete():
    from liverrenderer_amd.distributed import tile_pixels
    for (w, h) in [(256, 256), (854, 480), (1920, 1080), (33, 31), (1, 1)]:
        for world in (1, 2, 3, 8):
            parts = [tile_pixels(r, world, w, h) for r in range(world)]
            allp = np.concatenate(parts)
            assert allp.size == w * h and np.unique(allp).size == w * h
            if world > 1 and w * h > 4096:
                sizes = [p.size for p in parts]
                assert max(sizes) - min(sizes) <= 2 * 32 * 32 + 32 * max(w, h) // world + 1024


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import liverrenderer_amd as mi
    from liverrenderer_amd.distributed import render_distributed, tile_pixels, reduce_gradients
    import orc
    sc = mi.load_dict(mi.cornell_box())
    h, w, _ = sc.film_shape()
    spp = 2
    o = orc.OrcScene(sc)

    def render_rank(r, wd, film):
        # oracle render of the full image, keeping only samples whose pixel this rank owns; with the
        # Gaussian filter a sample splats into neighbouring (foreign) pixels too, exactly as on the GPU
        own = np.zeros(h * w, bool); own[tile_pixels(r, wd, w, h)] = True
        lanes = o.render_samples(0, h * w * spp, threads=2, spp=spp)
        full = np.zeros((h, w, sc.raw_channels()), np.float32)
        # box-filter equivalent accumulation is enough to exercise reduce + develop
        pix = np.arange(h * w * spp) // spp
        sel = own[pix]
        np.add.at(full.reshape(-1, 4), pix[sel], np.concatenate([lanes[sel, :3], np.ones((sel.sum(), 1), np.float32)], 1))
        film.copy_(torch.from_numpy(full))

    img, raw = render_distributed(sc, render_rank_fn=render_rank)
    g = reduce_gradients({"sigma_t": np.full(3, rank + 1.0), "albedo": np.full(3, 0.5), "g": 1.0})
    if rank == 0:
        np.savez(out_path, img=img.numpy(), raw=raw.numpy(), g=np.concatenate([g["sigma_t"], g["albedo"], [g["g"]]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_film_reduce_gloo(tmp_path, mi, orc, cornell):
    port = 29500 + os.getpid() % 2000
    out = str(tmp_path / "dist.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    spp = 2
    h, w, _ = cornell.film_shape()
    lanes = orc.OrcScene(cornell).render_samples(0, h * w * spp, threads=4, spp=spp)
    expect = lanes[:, :3].reshape(h * w, spp, 3).sum(1).reshape(h, w, 3)
    assert (r["raw"][..., 3] == spp).all()                        # every pixel received all its samples exactly once
    assert np.allclose(r["raw"][..., :3], expect, rtol=1e-6, atol=1e-6)
    assert np.allclose(r["img"], expect / spp, rtol=1e-6, atol=1e-6)
    assert np.allclose(r["g"], [3, 3, 3, 1, 1, 1, 2])


def test_develop_zero_weight():
    from liverrenderer_amd.distributed import develop
    raw = np.array([[[2.0, 4.0, 6.0, 2.0], [1.0, 1.0, 1.0, 0.0]]], np.float32)
    assert np.allclose(develop(raw), [[[1, 2, 3], [1, 1, 1]]])


def _gpu_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import liverrenderer_amd as mi
    from liverrenderer_amd.distributed import render_distributed, reduce_gradients
    from conftest import LIVER_XML
    sc = mi.load_file(LIVER_XML, integrator="volpath", spp=8, res_width=320, res_height=180)
    img, raw = render_distributed(sc, spp=8, seed=3)           # HIP back-end on this rank's tiles, film summed by all_reduce
    h, w, c = sc.film_shape()
    gi = np.full((h, w, c), 1.0 / (h * w * c), np.float32)
    g = reduce_gradients(sc.render_backward(gi, spp=8, seed=3, tile_rank=rank, tile_count=world))
    if rank == 0:
        np.savez(out_path, img=img.cpu().numpy(), raw=raw.cpu().numpy(), g=np.concatenate([g["sigma_t"], g["albedo"], [g["g"]]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_render_on_gpu_gloo(tmp_path, mi):
    """The product's own N > 1 path with the HIP back-end: two processes share the one GPU of the test box (RCCL needs a
    GPU per rank, so the reduction runs over gloo here); the summed film and the reduced PRB gradients equal the
    single-process results."""
    from conftest import LIVER_XML
    port = 31500 + os.getpid() % 2000
    out = str(tmp_path / "dist_gpu.npz")
    mp.spawn(_gpu_worker, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    sc = mi.load_file(LIVER_XML, integrator="volpath", spp=8, res_width=320, res_height=180)
    img, raw = sc.render(spp=8, seed=3, return_raw=True)
    assert (r["raw"][..., -1] == 8).all()
    scale = np.maximum(np.abs(raw).max(axis=-1, keepdims=True), 1.0)
    assert (np.abs(r["raw"] - raw) <= 8e-5 * scale).all()
    assert np.allclose(r["img"], img, rtol=1e-4, atol=1e-5)
    h, w, c = sc.film_shape()
    g = sc.render_backward(np.full((h, w, c), 1.0 / (h * w * c), np.float32), spp=8, seed=3)
    ref = np.concatenate([g["sigma_t"], g["albedo"], [g["g"]]])
    assert np.allclose(r["g"], ref, rtol=2e-4, atol=1e-7 + 2e-4 * np.abs(ref).max())
