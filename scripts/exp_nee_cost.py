"""Experiment: cost of the in-medium NEE (envmap sampling) on C3-like workload."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liverrenderer_amd as mi
xml = open(os.path.join(ROOT, "scenes/Liver-SingleMesh/mitsuba3/scene.xml")).read()
base_dir = os.path.join(ROOT, "scenes/Liver-SingleMesh/mitsuba3")
for name, x in [("sample_emitters=true", xml), ("sample_emitters=false", xml.replace('<phase type="isotropic"/>', '<boolean name="sample_emitters" value="false"/><phase type="isotropic"/>'))]:
    sc = mi.load_string(x, base_dir, integrator="volpath", spp=128, res_width=1920, res_height=1080)
    sc.render(spp=16)
    t = time.time(); sc.render(); dt = time.time() - t; st = sc.stats()
    print(f"{name}: {1920*1080*128/dt/1e6:.1f} Msamples/s, kernel {st['kernel_ms']:.1f} ms, total {st['total_ms']:.1f} ms, n_iter {st['n_iter']}, launches {st['n_launches']}", flush=True)
