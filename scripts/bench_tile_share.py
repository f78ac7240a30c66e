"""What one rank of an N-GPU run costs on one GPU: C3 with tile_rank = 0, tile_count = N (32x32 pixel tiles t % N == 0), against the
full frame.  Prints kernel time per rank share, its ratio to full / N, and the per-sample rate.  python scripts/bench_tile_share.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import liverrenderer_amd as mi
sc = mi.load_file(os.path.join(ROOT, "scenes", "Liver-SingleMesh", "mitsuba3", "scene.xml"), integrator="volpath", spp=512, res_width=1920, res_height=1080)
def run(n):
    best = None
    for rep in range(3):
        sc.render(seed=rep, tile_rank=0, tile_count=n, return_raw=True)
        st = sc.stats()
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    return best
full = run(1)
print(json.dumps({"tile_count": 1, "kernel_ms": round(full["kernel_ms"], 3), "Msamples_per_s": round(full["n_samples"] / full["kernel_ms"] / 1e3, 1)}))
for n in (2, 4, 8):
    st = run(n)
    print(json.dumps({"tile_count": n, "kernel_ms": round(st["kernel_ms"], 3), "samples": st["n_samples"], "Msamples_per_s": round(st["n_samples"] / st["kernel_ms"] / 1e3, 1),
                      "efficiency_vs_full_frame_rate": round((st["n_samples"] / st["kernel_ms"]) / (full["n_samples"] / full["kernel_ms"]), 3)}), flush=True)
