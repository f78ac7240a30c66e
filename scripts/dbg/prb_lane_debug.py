"""Per-trip view of one lane's PRB passes on the device (experiment build: printf) and in the oracle: python scripts/dbg/prb_lane_debug.py SEED Y X"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("LRT_LIBRARY", os.path.join(ROOT, "scripts", "dbg", "libliverrt_exp.so"))
import numpy as np
import liverrenderer_amd as mi
import orc
from test_fuzz_gpu import random_scene_xml
seed, y, x = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
xml, _ = random_scene_xml(seed); sc = mi.load_string(xml); o = orc.OrcScene(sc)
h, w, c = sc.film_shape(); spp = sc.spp
base = np.random.default_rng(seed).random((h, w, c)).astype(np.float32) / (h * w * c)
grad = np.zeros_like(base); grad[y, x] = base[y, x]
lane = (y * w + x) * spp + (int(sys.argv[4]) if len(sys.argv) > 4 else 0)
os.environ["LRT_PRB_DEBUG_LANE"] = str(lane)
print("lane", lane, "of pixel", y, x, flush=True)
g = sc.render_backward(grad, seed=seed)
print("device gradients", g, flush=True)
os.environ["ORC_PRB_DEBUG"] = "1"
sys.stderr.flush()
go = o.render_backward(grad, seed=seed, threads=1)
print("oracle gradients", go, flush=True)
