"""mi.traverse on the random scenes of tests/test_fuzz_gpu.py: after every medium parameter has been changed through the parameter interface
(the oracle is then built from the updated description) the lanes must still agree bit for bit - derived per-medium quantities (scattering weights, the NEE
rejection switch, the bio media's prepared terms) must follow: python scripts/fuzz_params.py FIRST LAST [r2]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import liverrenderer_amd as mi
import orc
from test_fuzz_gpu import random_scene_xml, random_scene_xml_r2
R2 = len(sys.argv) > 3 and sys.argv[3] == "r2"
TMP = tempfile.mkdtemp()
bad = skipped = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    xml, integ = random_scene_xml_r2(seed, TMP) if R2 else random_scene_xml(seed)
    try:
        sc = mi.load_string(xml)
        sc.render_samples(0, 64, seed=seed)                                # the device scene exists before the parameters change
        p = mi.traverse(sc)
        if not len(p): skipped += 1; continue
        r = np.random.default_rng(seed)
        changed = {}
        for k in sorted(p):
            v = np.asarray(p[k], np.float32)
            if k.endswith("phase_function.g"):
                if float(v[0]) == 0.0 and r.random() < 0.5: continue         # isotropic media have no such key
                nv = np.array([r.uniform(-0.8, 0.8)], np.float32)
            elif k.endswith("albedo.value"): nv = np.clip(v * r.uniform(0.6, 1.3, v.shape), 0.01, 0.99).astype(np.float32)
            else: nv = (v * r.uniform(0.5, 1.6, v.shape)).astype(np.float32)
            try:
                sc.param_set(k, nv)
            except RuntimeError as e:
                if "isotropic" in str(e): continue
                raise
            changed[k] = nv
        o = orc.OrcScene(sc)                                                # the oracle is built from the updated description
        h, w, _ = sc.film_shape()
        n = w * h * min(sc.spp, sc.desc.samples_per_pass or sc.spp)
        g = sc.render_samples(0, n, seed=seed); c = o.render_samples(0, n, seed=seed)
        same = (g.view(np.uint32) == c.view(np.uint32)).all(axis=1)
        st = sc.stats()
        ok = same.all() and st["n_iter"] == o.last_stats["n_iter"] and st["n_shadow"] == o.last_stats["n_shadow_needed"]
        if ok and integ == "prbvolpath":
            h, w, cc = sc.film_shape()
            grad = r.random((h, w, cc)).astype(np.float32) / (h * w * cc)
            gg, gc = sc.render_backward(grad, seed=seed), o.render_backward(grad, seed=seed)
            for k in ("sigma_t", "albedo"): ok &= bool(np.abs(gg[k] - gc[k]).max() <= 3e-4 * max(np.abs(gc[k]).max(), 1e-7))
            ok &= abs(gg["g"] - gc["g"]) <= 3e-4 * max(abs(gc["g"]), 1e-6) + 1e-9
        if not ok:
            bad += 1; print(f"seed {seed} ({integ}): {int((~same).sum())} lanes differ after {sorted(changed)}; n_iter {st['n_iter']} vs {o.last_stats['n_iter']}", flush=True)
    except Exception as e:
        bad += 1; print(f"seed {seed}: {type(e).__name__}: {e}", flush=True)
    if (seed + 1) % 100 == 0: print(f"... {seed + 1} done, {bad} failures", flush=True)
print(f"parameter updates {sys.argv[1]}..{sys.argv[2]}{' (r2)' if R2 else ''}: {bad} failures ({skipped} scenes without media parameters)")
