"""Ad-hoc GPU probe: per-lane parity of the HIP path against the oracle + a timing."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import liverrenderer_amd as mi
import orc

def compare(name, sc, n=1 << 16, lane0=0, **kw):
    o = orc.OrcScene(sc)
    t = time.time(); g = sc.render_samples(lane0, n, **kw); tg = time.time() - t
    st = sc.stats()
    t = time.time(); c = o.render_samples(lane0, n, **kw); tc = time.time() - t
    same = (g.view(np.uint32) == c.view(np.uint32)).all(axis=1)
    close = np.isclose(g, c, rtol=1e-4, atol=1e-6).all(axis=1)
    print(f"[{name}] lanes={n} bit-exact={same.mean():.6f} close={close.mean():.6f} gpu={tg:.3f}s cpu={tc:.3f}s "
          f"n_iter gpu={st['n_iter']} cpu={o.last_stats['n_iter']} n_shadow gpu={st['n_shadow']} cpu_needed={o.last_stats['n_shadow_needed']} cpu_all={o.last_stats['n_shadow']}", flush=True)
    bad = np.nonzero(~same)[0][:5]
    for b in bad: print("   lane", lane0 + b, g[b], c[b])
    return same.mean()

if __name__ == "__main__":
    sc = mi.load_dict(mi.cornell_box())
    # ray queries
    rng = np.random.default_rng(1)
    o = rng.uniform(-0.9, 0.9, (100000, 3)).astype(np.float32); d = rng.normal(size=(100000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    tg = sc.trace(o, d); tc = orc.OrcScene(sc).trace(o, d, brute_force=True)
    print("trace cornell: t equal", (tg[0].view(np.uint32) == tc[0].view(np.uint32)).mean(), "prim equal", (tg[3] == tc[3]).mean(), flush=True)
    compare("cornell path", sc, spp=64)
    compare("cornell volpath", sc, spp=64, integrator="volpath")
    compare("cornell path hide", sc, spp=16, hide_emitters=True)
    liver = os.path.join(ROOT, "scenes/Liver-SingleMesh/mitsuba3/scene.xml")
    sl = mi.load_file(liver, integrator="volpath", spp=16, res_width=256, res_height=144)
    ol = orc.OrcScene(sl)
    o = (rng.uniform(-60, 60, (200000, 3)) + np.array([-5, 5, -5])).astype(np.float32)
    tg = sl.trace(o, d.repeat(2, 0)); tc = ol.trace(o, d.repeat(2, 0), brute_force=True)
    print("trace liver: t equal", (tg[0].view(np.uint32) == tc[0].view(np.uint32)).mean(), "prim equal", (tg[3] == tc[3]).mean(), "hit frac", np.isfinite(tc[0]).mean(), flush=True)
    compare("liver volpath", sl, n=256 * 144 * 16)
    sl.param_set("LiverMedium.phase_function.g", 0.7)
    compare("liver volpath hg", sl, n=1 << 16, lane0=256 * 72 * 16)
    # full render timing
    for spp in (16, 64):
        t = time.time(); img = sc.render(spp=spp); dt = time.time() - t; st = sc.stats()
        print(f"cornell 256^2 spp={spp}: {dt:.3f}s wall, device {st['total_ms']:.1f} ms, kernels {st['kernel_ms']:.1f} ms, launches {st['n_launches']}, "
              f"{256*256*spp/st['total_ms']/1e3:.1f} Msamples/s, mean {img.mean((0,1))}", flush=True)
    sb = mi.load_file(liver, integrator="volpath", spp=32, res_width=854, res_height=480)
    t = time.time(); img = sb.render(); dt = time.time() - t; st = sb.stats()
    print(f"liver 854x480 spp=32: {dt:.3f}s wall, device {st['total_ms']:.1f} ms, kernels {st['kernel_ms']:.1f} ms, launches {st['n_launches']}, n_iter {st['n_iter']}, "
          f"{854*480*32/st['total_ms']/1e3:.1f} Msamples/s, mean {img.mean((0,1))}", flush=True)
    np.save(os.path.join(ROOT, "gpurun_out", "liver_probe.npy"), img)
