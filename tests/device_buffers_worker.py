"""Body of tests/test_device_buffers_gpu.py, run in a fresh interpreter: torch has to initialise its HIP runtime before the library
does (as bench.py and the distributed ranks do), which a long-lived pytest process cannot guarantee."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
assert torch.cuda.is_available()
torch.zeros(1, device="cuda")
import liverrenderer_amd as mi
from conftest import LIVER_XML, PARENCHYMA_XML
from test_parity_gpu import film_close
from liverrenderer_amd.distributed import develop

case = sys.argv[1]
if case == "cornell_gaussian": sc, kw = mi.load_dict(mi.cornell_box()), dict(spp=8, seed=5)
elif case == "liver_box_rgba": sc, kw = mi.load_file(LIVER_XML, integrator="volpath", spp=8, res_width=160, res_height=90), dict(seed=2)
else: sc, kw = mi.load_file(PARENCHYMA_XML, spp=16, res_width=160, res_height=90), dict(seed=1)
h, w, c = sc.film_shape(); C = sc.raw_channels()
img_h, raw_h = sc.render(return_raw=True, **kw)
dev = torch.device("cuda", 0)
film = torch.zeros((h, w, C), dtype=torch.float32, device=dev); image = torch.empty((h, w, c), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
sc.render_to_device(film.data_ptr(), image.data_ptr(), **kw)
raw_d, img_d = film.cpu().numpy(), image.cpu().numpy()
assert film_close(raw_d, raw_h).all(), "device-buffer film differs from the host-buffer film"
assert np.allclose(img_d, img_h, rtol=2e-4, atol=1e-6), "device-buffer image differs"
# the library's own develop on device buffers, and the torch develop the multi-GPU path uses after the all-reduce
image2 = torch.empty_like(image); sc.develop(film_ptr=film.data_ptr(), image_ptr=image2.data_ptr()); torch.cuda.synchronize()
assert np.array_equal(image2.cpu().numpy(), img_d), "lrt_film_develop on device buffers differs from lrt_render's image"
assert np.allclose(develop(film).cpu().numpy(), img_d, rtol=1e-6, atol=0), "distributed.develop differs from the library's develop"
# every rank renders its tile shard into its own film; the films add up to the full film (the sum is the RCCL all-reduce)
acc = torch.zeros_like(film)
for r in range(3):
    part = torch.full_like(film, 7.0); torch.cuda.synchronize()           # lrt_render clears the film it is handed: stale contents do not leak
    sc.render_to_device(part.data_ptr(), None, tile_rank=r, tile_count=3, **kw)
    acc += part
assert film_close(acc.cpu().numpy(), raw_h).all(), "three shard films do not add up to the full film"
# render_distributed with one rank is the plain render
from liverrenderer_amd.distributed import render_distributed
img_1, raw_1 = render_distributed(sc, **kw)
assert film_close(raw_1.cpu().numpy(), raw_h).all() and np.allclose(img_1.cpu().numpy(), img_h, rtol=2e-4, atol=1e-6)
print("ok", case)
