"""Which pixels carry a PRB gradient mismatch between device and oracle: python scripts/dbg/prb_bisect.py SEED"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import liverrenderer_amd as mi
import orc
from test_fuzz_gpu import random_scene_xml
seed = int(sys.argv[1])
xml, integ = random_scene_xml(seed)
print(xml)
sc = mi.load_string(xml); o = orc.OrcScene(sc)
h, w, c = sc.film_shape()
spp = sc.spp
print("film", w, h, "spp", spp, "max_depth", sc.desc.integrator.max_depth, "rr_depth", sc.desc.integrator.rr_depth, "hide", sc.desc.integrator.hide_emitters)
base = np.random.default_rng(seed).random((h, w, c)).astype(np.float32) / (h * w * c)
def vec(g): return np.concatenate([g["sigma_t"], g["albedo"], [g["g"]]]).astype(np.float64)
for y in range(h):
    for x in range(w):
        grad = np.zeros_like(base); grad[y, x] = base[y, x]
        a, b = vec(sc.render_backward(grad, seed=seed)), vec(o.render_backward(grad, seed=seed))
        if np.abs(a - b).max() > 3e-4 * max(np.abs(b).max(), 1e-9):
            print("pixel", y, x, "gpu", a, "\n            oracle", b, flush=True)
            n0 = (y * w + x) * spp
            g = sc.render_samples(n0, spp, seed=seed); cc = o.render_samples(n0, spp, seed=seed)
            print("   lanes", n0, "radiance", g[:, :3].tolist())
