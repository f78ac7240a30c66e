"""Details of fuzz-sweep failures: python scripts/dbg/fuzz_debug.py SEED [r2] ..."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import liverrenderer_amd as mi
import orc
from test_fuzz_gpu import random_scene_xml, random_scene_xml_r2
TMP = tempfile.mkdtemp()
args = sys.argv[1:]
i = 0
while i < len(args):
    seed = int(args[i]); r2 = i + 1 < len(args) and args[i + 1] == "r2"; i += 2 if r2 else 1
    xml, integ = random_scene_xml_r2(seed, TMP) if r2 else random_scene_xml(seed)
    sc = mi.load_string(xml); o = orc.OrcScene(sc)
    h, w, _ = sc.film_shape()
    print(f"== seed {seed} {integ} film {w}x{h} spp {sc.spp} samples_per_pass {sc.desc.samples_per_pass} rfilter {sc.desc.film.rfilter} sampler {sc.desc.sampler_type}", flush=True)
    if sc.desc.samples_per_pass:
        raw = sc.render(return_raw=True, seed=seed)[1]; st = dict(sc.stats())
        ora = o.render(return_raw=True, seed=seed)[1]
        scale = np.maximum(np.abs(ora).max(axis=-1, keepdims=True), 1.0)
        bad = ~(np.abs(raw - ora) <= 8e-5 * scale)
        print("  n_iter gpu", st["n_iter"], "oracle", o.last_stats["n_iter"], " n_launches", st.get("n_launches"), " bad film values", int(bad.sum()), "of", bad.size)
        bad &= np.isfinite(raw) & np.isfinite(ora)
        print('  finite bad', int(bad.sum()), ' non-finite pattern equal', bool(np.array_equal(np.isfinite(raw), np.isfinite(ora))))
        if bad.any():
            idx = np.argwhere(bad)[:6]
            for y, x, c in idx: print("   pixel", y, x, "channel", c, "gpu", raw[y, x], "oracle", ora[y, x])
            print("  nan/inf gpu", int((~np.isfinite(raw)).sum()), "oracle", int((~np.isfinite(ora)).sum()))
    if integ == "prbvolpath":
        h, w, c = sc.film_shape()
        grad = np.random.default_rng(seed).random((h, w, c)).astype(np.float32) / (h * w * c)
        gg, gc = sc.render_backward(grad, seed=seed), o.render_backward(grad, seed=seed)
        print("  gpu   ", gg); print("  oracle", gc)
        n = w * h * sc.spp
        g = sc.render_samples(0, n, seed=seed); cc = o.render_samples(0, n, seed=seed)
        print("  lanes equal", bool((g.view(np.uint32) == cc.view(np.uint32)).all()), " max |L|", float(np.abs(cc[:, :3]).max()), " nonfinite", int((~np.isfinite(cc)).sum()))
        print("  media", sc.desc.n_media, [ (list(sc.desc.media[k].sigma_t), list(sc.desc.media[k].albedo), sc.desc.media[k].g, sc.desc.media[k].phase) for k in range(sc.desc.n_media)])
        for m in range(-1, sc.desc.n_media):
            try:
                a, b = sc.render_backward(grad, seed=seed, medium=m), o.render_backward(grad, seed=seed, medium=m)
                print("  medium", m, "gpu", a, "oracle", b)
            except Exception as e: print("  medium", m, "error", e)
