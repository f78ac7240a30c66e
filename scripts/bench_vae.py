"""Throughput of the learned-subsurface scatter network (liverrenderer_amd/vae.py, kernels_vae.h) on one GPU: evaluations per
second through the host-buffer C ABI (copies included) and MACs per second.  python scripts/bench_vae.py [n]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from liverrenderer_amd import vae
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
base = os.path.join(ROOT, "scenes", "assets", "vae3d")
model = vae.load_scatter_model(os.path.join(base, "0487_FinalSharedLs7Mixed3_AbsSharedSimComplexMixed3"), os.path.join(base, "data_stats.json"))
r = np.random.default_rng(0)
pos = r.uniform(-2, 2, (n, 3)).astype(np.float32); d = r.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
poly = (r.normal(size=(n, 20)) * 0.3).astype(np.float32)
args = (pos, d, poly, (0.8, 0.5, 0.3), 0.4, 1.45, (1.0, 2.0, 4.0), 1.0)
model.scatter(*[a[:1000] if isinstance(a, np.ndarray) else a for a in args], 0)          # warm-up
t = time.perf_counter(); out, ab = model.scatter(*args, 1); dt = time.perf_counter() - t
macs = 64 * 23 + 2 * 64 * 64 + 32 * 64 + 32 + (1 - ab.mean()) * (64 * 68 + 2 * 64 * 64 + 3 * 64)
print(json.dumps({"evaluations": n, "seconds_host_buffers": round(dt, 4), "Mevals_per_s_host_buffers": round(n / dt / 1e6, 2),
                  "absorbed_fraction": round(float(ab.mean()), 4), "GMAC_per_s": round(float(n * macs / dt / 1e9), 1)}))
