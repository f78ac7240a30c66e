"""Lane refill inside the LDS traversal, measured on in-medium segments of the liver scene (developer experiment; needs `make -C
liverrenderer_amd/csrc exp`).  Builds a set of free-flight segments by random walks inside the mesh (isotropic directions, Exp(1) lengths,
as C3's medium draws them), keeps the ones a distance-field proof would not clear, and times k_trace_lds against k_trace_lds_refill<M>
(LRT_TRACE_REFILL) on them; the hits must be identical.  Usage: LRT_BVH_LEAF=4 python scripts/exp_trace_refill.py 0 64 128"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LRT_LIBRARY", os.path.join(ROOT, "scripts", "dbg", "libliverrt_exp.so"))
import numpy as np
import liverrenderer_amd as mi

rng = np.random.default_rng(7)
sc = mi.load_file(os.path.join(ROOT, "scenes", "Liver-SingleMesh", "mitsuba3", "scene.xml"), integrator="volpath", spp=4, res_width=64, res_height=36)
lo, hi = np.array([-54.2, -38.0, -56.8], np.float32), np.array([-22.5, -7.7, -18.6], np.float32)
BIG = np.float32(1e30)

def unit(n):
    z = rng.uniform(-1, 1, n); ph = rng.uniform(0, 2 * np.pi, n); r = np.sqrt(1 - z * z)
    return np.stack([r * np.cos(ph), r * np.sin(ph), z], 1).astype(np.float32)

# enter the mesh: rays from a sphere around it towards points of its bounding box
n0 = 1 << 20
c = (lo + hi) / 2; R = np.linalg.norm(hi - lo)
o = (c + R * unit(n0)).astype(np.float32); tgt = rng.uniform(lo, hi, (n0, 3)).astype(np.float32)
d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
t, u, v, prim = sc.trace(o, d)
hit = prim != 0xffffffff
p = (o[hit] + d[hit] * t[hit, None] + d[hit] * 1e-3).astype(np.float32); dcur = d[hit]
segs = []
for step in range(10):
    n = len(p)
    if n == 0: break
    dcur = unit(n); tt = rng.exponential(1.0, n).astype(np.float32)
    th, _, _, pr = sc.trace(p, dcur, tt)
    segs.append((p.copy(), dcur.copy(), tt.copy(), pr != 0xffffffff))
    keep = pr == 0xffffffff
    p = (p[keep] + dcur[keep] * tt[keep, None]).astype(np.float32)
O = np.concatenate([s[0] for s in segs]); D = np.concatenate([s[1] for s in segs]); T = np.concatenate([s[2] for s in segs]); H = np.concatenate([s[3] for s in segs])
# clearance estimate: nearest hit over 14 probe directions (an upper bound of the distance to the surface)
probe = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]] + [[a, b, cc] for a in (-1, 1) for b in (-1, 1) for cc in (-1, 1)], np.float32)
probe /= np.linalg.norm(probe, axis=1, keepdims=True)
clear = np.full(len(O), np.inf, np.float32)
for q in probe:
    tq, _, _, pq = sc.trace(O, np.broadcast_to(q, O.shape).copy())
    clear = np.minimum(clear, np.where(pq != 0xffffffff, tq, np.inf))
needs_query = T > 0.6 * clear
print(f"segments {len(O)}, with a hit {H.mean():.3f}; not cleared by the (approximate) distance proof: {needs_query.mean():.3f}, of which hit {H[needs_query].mean():.3f}", flush=True)
sel = np.flatnonzero(needs_query); rng.shuffle(sel)
reps = max(1, (1 << 23) // len(sel)); sel = np.tile(sel, reps)[: 1 << 23]; rng.shuffle(sel)
o, d, tm = O[sel], D[sel], T[sel]
ref = None
for m in [int(a) for a in sys.argv[1:]] or [0, 64, 128]:
    os.environ["LRT_TRACE_REFILL"] = str(m)
    try:
        res = sc.trace(o, d, tm)
    except Exception as e:
        print("M", m, "failed:", e); continue
    if ref is None: ref = res
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(ref, res))
    print(f"M={m}: hits identical to the first variant: {same}; hit fraction {np.mean(res[3] != 0xffffffff):.3f}", flush=True)
