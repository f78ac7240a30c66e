"""lrt_render_multi / lrt_render_backward_multi (include/liverrt.h [v104]; VERDICT r2 missing 3): one process, several devices, tiles over
the devices, ONE RCCL all-reduce of the film (of the 7 gradient sums), develop after it.  The test box has one GPU: the RCCL calls are
exercised with a one-device list (communicator of one rank, LRT_MULTI_ALWAYS_REDUCE=1), the sharding / threading / summing with a list
that names device 0 several times (the peers' films are added on the device; no collective runs between ranks of one device)."""
import os

import numpy as np
import pytest

from conftest import LIVER_XML, PARENCHYMA_XML
from test_parity_gpu import film_close

pytestmark = pytest.mark.gpu


def test_render_multi_one_device_through_rccl(mi, monkeypatch):
    sc = mi.load_file(LIVER_XML, integrator="volpath", spp=8, res_width=160, res_height=90)
    img, raw = sc.render(return_raw=True, seed=3)
    monkeypatch.setenv("LRT_MULTI_ALWAYS_REDUCE", "1")             # a communicator of one rank: ncclCommInitAll + ncclAllReduce run
    img1, raw1 = sc.render_multi([0], seed=3, return_raw=True)
    assert film_close(raw1, raw).all() and np.allclose(img1, img, rtol=2e-4, atol=1e-6)
    st = sc.stats()
    assert st["n_samples"] == 160 * 90 * 8 and st["n_iter"] > st["n_samples"]


@pytest.mark.parametrize("n", [2, 3])
def test_render_multi_shards_add_up(mi, cornell, n):
    """tiles t % n == i on peer i, films summed, developed after the sum: the image of the unsharded render (Gaussian filter: footprints
    cross tile borders)"""
    img, raw = cornell.render(return_raw=True, seed=5, spp=8)
    imgn, rawn = cornell.render_multi([0] * n, seed=5, spp=8, return_raw=True)
    assert film_close(rawn, raw).all() and np.allclose(imgn, img, rtol=2e-4, atol=1e-6)
    assert cornell.stats()["n_samples"] == 8 * cornell.film_shape()[0] * cornell.film_shape()[1]


def test_render_backward_multi_gradients_add_up(mi):
    sc = mi.load_file(PARENCHYMA_XML, integrator="prbvolpath", spp=16, res_width=96, res_height=54)
    h, w, c = sc.film_shape()
    grad = np.random.default_rng(2).random((h, w, c)).astype(np.float32) / (h * w * c)
    full = sc.render_backward(grad, seed=4)
    two = sc.render_backward_multi(grad, [0, 0], seed=4)
    one = sc.render_backward_multi(grad, [0], seed=4)
    for k in ("sigma_t", "albedo"):
        assert np.allclose(two[k], full[k], rtol=2e-3, atol=1e-7) and np.allclose(one[k], full[k], rtol=2e-3, atol=1e-7)
    assert np.abs(full["sigma_t"]).max() > 0


def test_render_multi_argument_checks(mi, cornell):
    from liverrenderer_amd import _lib
    import ctypes as C
    o = _lib.make_opts(spp=2, tile_rank=1, tile_count=2)
    img = np.empty(cornell.film_shape(), np.float32)
    ids = (C.c_int * 1)(0)
    assert cornell._lib.lrt_render_multi(cornell._h, C.byref(o), 1, ids, None, img.ctypes.data) != 0
    assert b"tile_rank" in cornell._lib.lrt_last_error()
    with pytest.raises(RuntimeError, match="n_devices"):
        cornell.render_multi([], spp=2)
    with pytest.raises(RuntimeError, match="(?i)device"):
        cornell.render_multi([0, 63], spp=2)                        # no such device (and not a repeated one)
    with pytest.raises(RuntimeError, match="(?i)all distinct or one device repeated|device"):
        cornell.render_multi([0, 0, 63], spp=2)
    assert np.isfinite(cornell.render_multi([0], spp=2)).all()      # the scene is still usable
