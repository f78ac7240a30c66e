"""Times the BASELINE configs other than C3 on one GPU (C2 Cornell path, C4 MultiMesh volpath on 1 GPU, C5 PRB adjoint
on Parenchyma) and prints one JSON line each.  Not the driver's bench (that is bench.py = C3)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import liverrenderer_amd as mi

def timed(name, fn, n_samples, reps=2):
    fn(100)
    t0 = time.perf_counter()
    for i in range(reps): fn(i)
    dt = (time.perf_counter() - t0) / reps
    return {"config": name, "ms": round(dt * 1e3, 2), "Msamples/s": round(n_samples / dt / 1e6, 1)}

which = sys.argv[1:] or ["C2", "C4", "C5"]
if "C2" in which:
    d = mi.cornell_box(); d["sensor"]["film"].update({"width": 1080, "height": 1080})
    sc = mi.load_dict(d)
    r = timed("C2 cornell_box 1080x1080 path 256 spp", lambda s: sc.render(spp=256, seed=s), 1080 * 1080 * 256); r.update(sc.stats()); print(json.dumps(r), flush=True)
if "C4" in which:
    sc = mi.load_file(os.path.join(ROOT, "scenes/Liver-MultiMesh/mitsuba3/scene.xml"), integrator="volpath", spp=256, res_width=1920, res_height=1080)
    r = timed("C4 Liver-MultiMesh 1920x1080 volpath 256 spp (1 GPU)", lambda s: sc.render(seed=s), 1920 * 1080 * 256); r.update(sc.stats()); print(json.dumps(r), flush=True)
if "C5" in which:
    sc = mi.load_file(os.path.join(ROOT, "scenes/Parenchyma/mitsuba3/scene_temp.xml"), integrator="prbvolpath", spp=256, res_width=1920, res_height=1080)
    h, w, c = sc.film_shape()
    g = np.full((h, w, c), 1.0 / (h * w * c), np.float32)
    r = timed("C5 Parenchyma 1920x1080 prbvolpath backward 256 spp (1 GPU, primal + adjoint)", lambda s: sc.render_backward(g, seed=s), 1920 * 1080 * 256, reps=1); r.update(sc.stats()); print(json.dumps(r), flush=True)
    sc2 = mi.load_file(os.path.join(ROOT, "scenes/Parenchyma/mitsuba3/scene_temp.xml"), integrator="volpath", spp=64, res_width=1920, res_height=1080)
    r = timed("Parenchyma 1920x1080 volpath 64 spp", lambda s: sc2.render(seed=s), 1920 * 1080 * 64); r.update(sc2.stats()); print(json.dumps(r), flush=True)
