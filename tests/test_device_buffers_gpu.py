"""The device-buffer form of lrt_render (`output_on_device`): what bench.py times and liverrenderer_amd/distributed.py reduces over RCCL.
It must produce the film and image of the host-buffer form (same seed: equal up to the order of the float atomics), clear the film it is
handed, develop on device buffers like `distributed.develop`, and honour tile shards (shard films add up to the full film).  The body runs in a fresh interpreter
(tests/device_buffers_worker.py): torch must initialise its HIP runtime before the library does."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["cornell_gaussian", "liver_box_rgba", "parenchyma_tent_ld"])
def test_device_buffers_match_host_buffers(case):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "device_buffers_worker.py"), case], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and f"ok {case}" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("order", ["lrt_first", "torch_first"])
def test_import_order_gpu(order):
    """either import order of liverrenderer_amd and torch leaves both with the GPU (the binding creates PyTorch's
    context before it loads libliverrt.so)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "import_order_worker.py"), order], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and f"ok {order}" in r.stdout, (r.stdout + r.stderr)[-3000:]
