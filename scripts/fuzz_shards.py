"""Tile sharding on the random scenes of tests/test_fuzz_gpu.py: the raw films of N tile shards must add up to the unsharded film (all
filters, crop windows, multi-pass renders), and the adjoint's shard gradients to the unsharded gradients: python scripts/fuzz_shards.py FIRST LAST [r2]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import liverrenderer_amd as mi
from test_fuzz_gpu import random_scene_xml, random_scene_xml_r2
from test_parity_gpu import film_close
R2 = len(sys.argv) > 3 and sys.argv[3] == "r2"
TMP = tempfile.mkdtemp()
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    xml, integ = random_scene_xml_r2(seed, TMP) if R2 else random_scene_xml(seed)
    try:
        sc = mi.load_string(xml)
        n = 2 + seed % 3
        if integ == "prbvolpath":
            h, w, c = sc.film_shape()
            grad = np.random.default_rng(seed).random((h, w, c)).astype(np.float32) / (h * w * c)
            full = sc.render_backward(grad, seed=seed)
            parts = [sc.render_backward(grad, seed=seed, tile_rank=r, tile_count=n) for r in range(n)]
            ok = True
            for k in ("sigma_t", "albedo"):
                tot = sum(np.asarray(p[k], np.float64) for p in parts)
                ok &= bool(np.abs(tot - full[k]).max() <= 3e-4 * max(np.abs(full[k]).max(), 1e-7))
            ok &= abs(sum(p["g"] for p in parts) - full["g"]) <= 3e-4 * max(abs(full["g"]), 1e-6) + 1e-9
        else:
            full = sc.render(return_raw=True, seed=seed)[1].astype(np.float64)
            tot = sum(sc.render(return_raw=True, seed=seed, tile_rank=r, tile_count=n)[1].astype(np.float64) for r in range(n))
            ok = bool(film_close(tot, full).all())
        if not ok:
            bad += 1; print(f"seed {seed} ({integ}, {n} shards): shard sum differs", flush=True)
    except Exception as e:
        bad += 1; print(f"seed {seed}: {e}", flush=True)
    if (seed + 1) % 100 == 0: print(f"... {seed + 1} done, {bad} failures", flush=True)
print(f"shard sums {sys.argv[1]}..{sys.argv[2]}{' (r2)' if R2 else ''}: {bad} failures")
