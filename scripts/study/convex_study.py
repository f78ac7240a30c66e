"""Offline study: per-triangle 'inward rays are free up to R' proof (all planes of triangles within R of T0 have T0 on
their inner side; normal cone).  Uses /tmp/segs.npz from proof_study.py."""
import numpy as np, sys
sys.argv = [sys.argv[0]] + sys.argv[1:]
exec(open('scripts/study/proof_study.py').read().split("# ---- random walks")[0])
z = np.load('/tmp/segs.npz'); P0, D0, T0, H0, depth0 = z['P0'], z['D0'], z['T0'], z['H0'], z['depth0']
nT = len(F)
# orient normals outward: the walk directions at depth ~0 point inward
cen = (A + B + C) / 3; rad = np.maximum(np.maximum(np.linalg.norm(A - cen, axis=1), np.linalg.norm(B - cen, axis=1)), np.linalg.norm(C - cen, axis=1))
# outward orientation from signed volume
vol = np.einsum('ij,ij->i', A, np.cross(B, C)).sum()
N = NRM if vol > 0 else -NRM
# conservative distance between T0 and Tj: dist(centroid_j, T0) - rad_j
DIST = np.empty((nT, nT))
for s in range(0, nT, 128):
    d, _ = closest_on_tris(cen[s:s+128]); DIST[s:s+128] = d          # [j, T0]
DIST = DIST - rad[:, None]
# side[j, T0] = max over vertices v of T0 of n_j.(v - p_j)
side = np.maximum(np.maximum((N[:, None, :] * (A[None] - A[:, None])).sum(-1), (N[:, None, :] * (B[None] - A[:, None])).sum(-1)), (N[:, None, :] * (C[None] - A[:, None])).sum(-1))
eps = 1e-6 * (1 + np.abs(V).max())
Rs = [0.5, 0.75, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]
best_R = np.zeros(nT); best_thr = np.ones(nT)
cosang = N @ N.T                                                       # [j, T0] cos of angle between normals
for R in Rs:
    S = DIST <= R                                                      # [j, T0]
    safe = ~(S & (side > eps)).any(0)
    # cone around n0: sin(beta) with beta = max angle(n_j, n0) over S
    mincos = np.where(S, cosang, 1).min(0)
    beta = np.arccos(np.clip(mincos, -1, 1))
    thr = np.sin(np.minimum(beta + 0.03, np.pi / 2))
    ok = safe & (beta + 0.03 < np.pi / 2)
    upd = ok
    best_R[upd] = R; best_thr[upd] = thr[upd]
    print(f"R {R}: safe {safe.mean() * 100:.1f} % of triangles, median beta {np.degrees(np.median(beta[safe])) if safe.any() else 0:.1f} deg, ok {ok.mean() * 100:.1f} %")
# note: best_R is the largest passing R but thr belongs to that R (larger R -> larger beta); a second candidate could be kept
sel = np.nonzero(depth0 < 0.02)[0]
d, _ = closest_on_tris(P0[sel]); t0 = d.argmin(1)
dn = (D0[sel] * N[t0]).sum(1)
print("surface-start segments:", len(sel), " inward elevation n.d quantiles:", np.quantile(dn, [.1, .5, .9]).round(3), " hits among them: %.1f %%" % (100 * H0[sel].mean()))
R = best_R[t0]; thr = best_thr[t0]
cone_ok = dn < -thr
full = cone_ok & (T0[sel] * 1.01 <= R)
part = cone_ok & ~full & (R > 0)
print(f"cone ok {cone_ok.mean() * 100:.1f} %, fully proven {full.mean() * 100:.1f} %, head proven only {part.mean() * 100:.1f} %;  wrong: {(full & H0[sel]).sum()}")
# head proven: continue with sphere tracing from p + R d for the rest
def sphere_from(P, D, T, res=192, steps=3):
    cell = ext / res
    rem = T * 1.01; pp = P.copy(); done = np.zeros(len(P), bool); ok = np.zeros(len(P), bool)
    for k in range(steps):
        c, _ = cell_centre(pp, res)
        lb = dist_to_surface(c) * .999 - 1e-4 * np.linalg.norm(hi - lo) - np.linalg.norm(pp - c, axis=1) * 1.001
        newly = ~done & (rem < lb); ok |= newly; done |= newly
        fail = ~done & ~(lb > .5 * cell); done |= fail
        adv = np.where(done, 0, lb * .99); pp = pp + D * adv[:, None]; rem = rem - adv * .995
    return ok
ip = np.nonzero(part)[0]
rest_ok = sphere_from(P0[sel][ip] + D0[sel][ip] * (R[ip] * .99)[:, None], D0[sel][ip], T0[sel][ip] - R[ip] * .99)
tot = full.sum() + rest_ok.sum()
print(f"head + sphere tracing for the rest: {rest_ok.mean() * 100:.1f} % of the partial ones; total proven {tot / len(sel) * 100:.1f} % of surface-start segments (free among them: {(~H0[sel]).mean() * 100:.1f} %); wrong {(H0[sel][ip] & rest_ok).sum()}")
