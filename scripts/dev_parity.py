"""Per-lane parity of the library named by LRT_LIBRARY against the oracle on small volpath renders (developer A/B aid; DEV_SCENE=multimesh: Liver-MultiMesh with its own defaults)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import liverrenderer_amd as mi
import orc
xml = os.path.join(ROOT, "scenes", "Liver-SingleMesh", "mitsuba3", "scene.xml")
kw = dict(integrator="volpath")
if os.environ.get("DEV_SCENE") == "multimesh":                 # a developer build of the biovolpath / ld kernel: the scene's own defaults
    xml = os.path.join(ROOT, "scenes", "Liver-MultiMesh", "mitsuba3", "scene_temp.xml"); kw = {}
ok = True
for (w, h, spp, seed) in ((128, 72, 16, 0), (96, 54, 64, 3)):
    sc = mi.load_file(xml, spp=spp, res_width=w, res_height=h, **kw)
    n = w * h * spp
    g = sc.render_samples(0, n, seed=seed); st = sc.stats()
    o = orc.OrcScene(sc); c = o.render_samples(0, n, seed=seed)
    same = (g.view(np.uint32) == c.view(np.uint32)).all(axis=1).mean()
    print(f"{os.environ.get('LRT_LIBRARY', 'in-tree')}: {w}x{h}x{spp}: lanes identical {same:.6f}, trips {st['n_iter']} (oracle {o.last_stats['n_iter']})")
    ok &= (same == 1.0) and st["n_iter"] == o.last_stats["n_iter"]
    from test_parity_gpu import film_close
    raw = sc.render(return_raw=True, seed=seed)[1]; ora = o.render(return_raw=True, seed=seed)[1]
    fc = film_close(raw, ora).all(); wexact = np.array_equal(raw[..., -1], ora[..., -1])
    print(f"   film close to the oracle's: {fc}; weight channel identical: {wexact}")
    ok &= bool(fc) and bool(wexact)
sys.exit(0 if ok else 1)
