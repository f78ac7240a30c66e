"""One-off wide sweep of tests/test_fuzz_gpu.py's random scenes: python scripts/fuzz_sweep.py FIRST LAST [r2 | prbhet]
(r2: the round-2 families: bio integrators / media, heterogeneous media, volpathmis; prbhet: the PRB adjoint on the round-2 volume scenes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import liverrenderer_amd as mi
import orc
from test_fuzz_gpu import random_scene_xml, random_scene_xml_r2, random_scene_prb_het
import tempfile
R2 = len(sys.argv) > 3 and sys.argv[3] == "r2"
PRBHET = len(sys.argv) > 3 and sys.argv[3] == "prbhet"
TMP = tempfile.mkdtemp()
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    xml, integ = random_scene_xml_r2(seed, TMP) if R2 else ((random_scene_prb_het(seed, TMP), "prbvolpath") if PRBHET else random_scene_xml(seed))
    try:
        sc = mi.load_string(xml); o = orc.OrcScene(sc)
        h, w, _ = sc.film_shape()
        n = w * h * (sc.spp if PRBHET else min(sc.spp, sc.desc.samples_per_pass or sc.spp))
        g = sc.render_samples(0, n, seed=seed); c = o.render_samples(0, n, seed=seed)
        same = (g.view(np.uint32) == c.view(np.uint32)).all(axis=1)
        st = sc.stats()
        ok = same.all() and st["n_iter"] == o.last_stats["n_iter"] and st["n_shadow"] == o.last_stats["n_shadow_needed"]
        if ok and sc.desc.samples_per_pass and not PRBHET:     # multi-pass: compare the films (all passes)
            raw = sc.render(return_raw=True, seed=seed)[1]; ora = o.render(return_raw=True, seed=seed)[1]
            from test_parity_gpu import film_close                # non-finite film values: same pattern on both sides
            if not film_close(raw, ora).all() or sc.stats()["n_iter"] != o.last_stats["n_iter"]:
                ok = False; print(f"seed {seed}: multi-pass film mismatch", flush=True)
        if ok and integ == "prbvolpath":                       # the adjoint too: gradients equal up to summation order
            h, w, c = sc.film_shape()
            grad = np.random.default_rng(seed).random((h, w, c)).astype(np.float32) / (h * w * c)
            gg, gc = sc.render_backward(grad, seed=seed), o.render_backward(grad, seed=seed)
            for k in ("sigma_t", "albedo"):
                if not (np.abs(gg[k] - gc[k]).max() <= 3e-4 * max(np.abs(gc[k]).max(), 1e-7)): ok = False
            if not (abs(gg["g"] - gc["g"]) <= 3e-4 * max(abs(gc["g"]), 1e-6) + 1e-9): ok = False
            if not ok: print(f"seed {seed}: gradient mismatch {gg} vs {gc}", flush=True)
        if not ok:
            bad += 1
            print(f"seed {seed} ({integ}): {int((~same).sum())} lanes differ, n_iter {st['n_iter']} vs {o.last_stats['n_iter']}, n_shadow {st['n_shadow']} vs {o.last_stats['n_shadow_needed']}", flush=True)
    except Exception as e:
        bad += 1; print(f"seed {seed}: {type(e).__name__}: {e}", flush=True)
    if seed % 100 == 99:
        print(f"... {seed + 1} done, {bad} failures", flush=True)
        for f in os.listdir(TMP): os.remove(os.path.join(TMP, f))
print(f"swept {sys.argv[1]}..{sys.argv[2]}{' (r2)' if R2 else (' (prbhet)' if PRBHET else '')}: {bad} failures", flush=True)
