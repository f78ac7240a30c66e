"""Cost attribution of the render kernel by tile kind (developer experiment).
Build:  make -C liverrenderer_amd/csrc exp      Run (GPU): python scripts/exp_cost.py
Each run repeats the loop trip of one tile kind (0 proven-free medium, 1 medium + ray query, 2 surface, 3 fresh camera)
on a copy of the path; the slowdown over the plain run is that kind's share of the kernel."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "scripts", "dbg", "libliverrt_exp.so")
base = None
for name, exp in [("plain", 0), ("2x proven-free", 1 << 8), ("2x query", 2 << 8), ("2x surface", 4 << 8), ("2x fresh", 8 << 8), ("machinery only, 4 trips/path (3 records)", 0x1000), ("machinery only, no film", 0x1002), ("machinery only, 3 trips/path (2 records)", 0x3000)]:
    env = dict(os.environ, LRT_LIBRARY=lib, LRT_EXP=str(exp))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--spp", "128"],
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    ms = json.loads(out)["ms_per_step"]
    base = base or ms
    print(f"{name:44s} {ms:8.2f} ms/step   +{(ms / base - 1) * 100:5.1f} %", flush=True)
