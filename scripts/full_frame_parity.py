"""One-off check: every lane of a full frame of the reference scenes (4 spp) is bit-identical to the oracle."""
import sys, time; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, liverrenderer_amd as mi, orc
for name, xml, kw in [("C3 liver", "scenes/Liver-SingleMesh/mitsuba3/scene.xml", dict(integrator="volpath", spp=4, res_width=1920, res_height=1080)),
                      ("parenchyma ld", "scenes/Parenchyma/mitsuba3/scene_temp.xml", dict(integrator="volpath", spp=4, res_width=1920, res_height=1080)),
                      ("glisson ld", "scenes/GlissonCapsule/mitsuba3/scene_temp.xml", dict(integrator="volpath", spp=4, res_width=1280, res_height=720))]:
    sc = mi.load_file(xml, **kw)
    n = sc.film_shape()[0] * sc.film_shape()[1] * sc.spp
    t = time.time(); g = sc.render_samples(0, n); tg = time.time() - t
    t = time.time(); c = orc.OrcScene(sc).render_samples(0, n, threads=16); tc = time.time() - t
    same = (g.view(np.uint32) == c.view(np.uint32)).all(axis=1)
    print(f"{name}: {n} lanes, {int((~same).sum())} differ (gpu {tg:.2f} s, oracle {tc:.1f} s)", flush=True)
