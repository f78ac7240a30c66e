import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import liverrenderer_amd as mi
from conftest import LIVER_XML
sc = mi.load_file(LIVER_XML, spp=4, res_width=64, res_height=36)
print("integrator", sc.desc.integrator.type, "wide" if os.environ.get("LRT_WIDE_RECORDS") else "compact", flush=True)
g = sc.render_samples(0, 64 * 36 * 4)
print("ok", float(g[:, :3].mean()), sc.stats()["n_iter"], flush=True)
