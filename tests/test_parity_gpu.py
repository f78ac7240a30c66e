"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  The path arithmetic is IEEE-exact on both sides (explicit fma,
polynomial transcendentals), so per-lane radiance and ray queries are compared
BIT-EXACTLY; only the film, accumulated with float atomics in arbitrary order,
gets a tolerance (1e-5 relative to the pixel weight, stated where used)."""
import os

import numpy as np
import pytest

from conftest import LIVER_XML, PARENCHYMA_XML, MULTIMESH_XML, GLISSON_XML, REALTIME_XML, ROOT
from test_oracle_pins import _stairs, stairs_rays, _cube

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_lanes_equal(sc, o, lane0, n, **kw):
    g = sc.render_samples(lane0, n, **kw)
    c = o.render_samples(lane0, n, **kw)
    same = (bits(g) == bits(c)).all(axis=1)
    assert same.all(), f"{(~same).sum()} of {n} lanes differ; first: lane {lane0 + int(np.argmin(same))} gpu={g[np.argmin(same)]} cpu={c[np.argmin(same)]}"
    st = sc.stats()
    assert st["n_iter"] == o.last_stats["n_iter"]
    assert st["n_shadow"] == o.last_stats["n_shadow_needed"]
    return g


def center_lane(sc, spp, row_frac=0.5):
    h, w, _ = sc.film_shape()
    return int(h * row_frac) * w * spp


# ------------------------------------------------------------------ ray queries
@pytest.mark.parametrize("lds", [True, False])
def test_trace_staircase_closed_form(mi, orc, monkeypatch, lds):
    """Axis-parallel rays (zero direction components) through both tracers: BVH image in LDS / BVH in global memory."""
    if not lds: monkeypatch.setenv("LRT_NO_LDS_BVH", "1")
    v, f = _stairs(20)
    sc = mi.scene_from_buffers(v, f)
    o, d, yy = stairs_rays()
    tmax = np.full(o.shape[0], 100.0, np.float32)
    t, u, vv, prim = sc.trace(o, d, tmax)
    expected = (2.0 - np.floor(np.float32(yy) * 20) / 20).astype(np.float32)
    assert np.allclose(t, expected, atol=1e-6)
    tb = orc.OrcScene(sc).trace(o, d, tmax, brute_force=True)
    for a, b in zip((t, u, vv, prim), tb):
        assert (bits(a) == bits(b)).all() if a.dtype == np.float32 else (a == b).all()
    assert (sc.trace(o, d, tmax, any_hit=True)[0] == 0).all()


@pytest.mark.parametrize("which", ["cornell", "liver", "liver-global-bvh"])
def test_trace_random_rays_bit_exact(mi, orc, cornell, liver_small, which, monkeypatch):
    rng = np.random.default_rng(11)
    n = 200000
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    if which == "cornell":
        sc, o = cornell, rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
    else:
        if which == "liver-global-bvh":
            monkeypatch.setenv("LRT_NO_LDS_BVH", "1")
            sc = mi.load_file(LIVER_XML, integrator="volpath", spp=4, res_width=64, res_height=36)
        else:
            sc = liver_small
        o = (rng.uniform(-1, 1, (n, 3)) * [25, 20, 25] + [-43, -18, -43]).astype(np.float32)
    tmax = np.where(rng.random(n) < 0.5, np.finfo(np.float32).max, rng.uniform(0.1, 30, n)).astype(np.float32)
    g = sc.trace(o, d, tmax)
    c = orc.OrcScene(sc).trace(o, d, tmax, brute_force=True)
    assert (bits(g[0]) == bits(c[0])).all() and (g[3] == c[3]).all()
    hit = g[3] != 0xffffffff
    assert hit.mean() > 0.1
    assert (bits(g[1][hit]) == bits(c[1][hit])).all() and (bits(g[2][hit]) == bits(c[2][hit])).all()
    ga = sc.trace(o, d, tmax, any_hit=True)[0]
    assert ((ga == 0) == hit).all()


def test_trace_degenerate_inputs(mi, cornell):
    # axis-aligned directions (zero components), zero-length maxt, rays starting on a surface
    o = np.array([[0, 0, 3.9], [0, 0, 3.9], [0, -1, 0], [5, 5, 5]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, -1], [0, 1, 0], [1, 0, 0]], np.float32)
    tmax = np.array([np.finfo(np.float32).max, 0.0, np.finfo(np.float32).max, 10.0], np.float32)
    t, u, v, prim = cornell.trace(o, d, tmax)
    assert np.isfinite(t[0]) and prim[1] == 0xffffffff and prim[3] == 0xffffffff and np.isinf(t[3])
    assert cornell.trace(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))[0].size == 0


# ---------------------------------------------------------------- per-lane radiance
@pytest.mark.parametrize("kw", [
    dict(spp=64), dict(spp=64, integrator="volpath"), dict(spp=16, hide_emitters=True),
    dict(spp=16, integrator="volpath", hide_emitters=True), dict(spp=7, seed=3), dict(spp=32, max_depth=2),
    dict(spp=32, max_depth=0), dict(spp=32, max_depth=1), dict(spp=32, rr_depth=1), dict(spp=32, max_depth=-1, rr_depth=2, seed=9),
])
def test_cornell_lanes_bit_exact(mi, orc, cornell, kw):
    o = orc.OrcScene(cornell)
    for frac in (0.1, 0.5, 0.9):
        assert_lanes_equal(cornell, o, center_lane(cornell, kw["spp"], frac), 1 << 15, **kw)


def test_liver_volpath_lanes_bit_exact(mi, orc):
    sc = mi.load_file(LIVER_XML, integrator="volpath", spp=16, res_width=256, res_height=144)
    o = orc.OrcScene(sc)
    g = assert_lanes_equal(sc, o, 0, 256 * 144 * 16)
    assert g[:, 3].min() == 1.0                       # envmap visible: every ray is valid
    assert_lanes_equal(sc, o, center_lane(sc, 16), 1 << 15, seed=5, max_depth=40)


def test_liver_volpath_hg_and_params_bit_exact(mi, orc):
    sc = mi.load_file(LIVER_XML, integrator="volpath", spp=16, res_width=256, res_height=144)
    o = orc.OrcScene(sc)
    p = mi.traverse(sc)
    assert "LiverMedium.sigma_t.value" in p and "LiverMedium.albedo.value" in p
    p["LiverMedium.phase_function.g"] = 0.7
    p["LiverMedium.sigma_t.value"] = [0.05, 0.08, 0.11]
    p["LiverMedium.albedo.value"] = [0.9, 0.8, 0.7]
    p.update()
    for k in ("LiverMedium.phase_function.g", "LiverMedium.sigma_t.value", "LiverMedium.albedo.value"):
        o.param_set(k, p[k])
    assert_lanes_equal(sc, o, center_lane(sc, 16, 0.4), 1 << 16)
    o.param_set("LiverMedium.phase_function.g", -0.4); sc.param_set("LiverMedium.phase_function.g", -0.4)
    assert_lanes_equal(sc, o, center_lane(sc, 16, 0.5), 1 << 15, seed=2)


def test_compact_and_wide_records_agree(mi, monkeypatch):
    """A scene without area emitters queues 80-byte records (no last scatter position, the sampler's TEA word kept instead of recomputed:
    kernels.h, store_state); LRT_WIDE_RECORDS forces the 88-byte layout.  Same lanes, trips and shadow rays either way, for the independent
    and the ld sampler, volpath / path / the bio integrators; a scene WITH an area emitter (Cornell) is not affected by the switch."""
    import os
    cases = [(LIVER_XML, dict(integrator="volpath", spp=16, res_width=128, res_height=72)), (LIVER_XML, dict(integrator="path", spp=16, res_width=128, res_height=72)),
             (LIVER_XML, dict(spp=16, res_width=128, res_height=72)), (PARENCHYMA_XML, dict(spp=16, res_width=96, res_height=54)), (MULTIMESH_XML, dict(spp=16, res_width=96, res_height=54))]
    for path, kw in cases:
        sc = mi.load_file(path, **kw)
        n = kw["res_width"] * kw["res_height"] * sc.desc.sample_count
        monkeypatch.delenv("LRT_WIDE_RECORDS", raising=False)
        a = sc.render_samples(0, n, seed=3); sa = sc.stats()
        img_a = sc.render(seed=3)
        monkeypatch.setenv("LRT_WIDE_RECORDS", "1")
        b = sc.render_samples(0, n, seed=3); sb = sc.stats()
        img_b = sc.render(seed=3)
        monkeypatch.delenv("LRT_WIDE_RECORDS", raising=False)
        assert (bits(a) == bits(b)).all(), (path, kw)
        assert sa["n_iter"] == sb["n_iter"] and sa["n_shadow"] == sb["n_shadow"] and sa["n_records"] == sb["n_records"]
        assert np.isfinite(img_a).all() and np.allclose(img_a, img_b, rtol=2e-4, atol=1e-5)          # film: float atomics in any order


def test_realtime_scene_bit_exact(mi, orc):
    sc = mi.load_file(REALTIME_XML, integrator="volpath", spp=4, res_width=192, res_height=108)      # rr_depth = max_depth = 12
    assert_lanes_equal(sc, orc.OrcScene(sc), 0, 192 * 108 * 4)


def test_parenchyma_and_multimesh_scenes_bit_exact(mi, orc):
    # Parenchyma: non-spectral medium without emitter sampling, constant emitter, tent filter, hide_emitters
    sc = mi.load_file(PARENCHYMA_XML, integrator="volpath", spp=8, res_width=160, res_height=90, max_depth=65)
    assert_lanes_equal(sc, orc.OrcScene(sc), 0, 160 * 90 * sc.spp)
    sc = mi.load_file(MULTIMESH_XML, integrator="volpath", spp=8, res_width=160, res_height=90)
    assert_lanes_equal(sc, orc.OrcScene(sc), 0, 160 * 90 * sc.spp)
    sc = mi.load_file(MULTIMESH_XML, integrator="path", spp=8, res_width=160, res_height=90)
    assert_lanes_equal(sc, orc.OrcScene(sc), 0, 160 * 90 * sc.spp)


from scene_gen import fog_xml  # noqa: E402  (shared with bench.py)


@pytest.mark.parametrize("variant", ["null_boundary", "camera_in_medium", "constant_env"])
def test_fog_scenes_bit_exact(mi, orc, variant):
    """Null (index-matched) boundaries make the NEE march multi-step; a sensor inside a medium
    exercises finite camera maxt; the checkerboard exercises uv interpolation."""
    kw = {}
    if variant == "camera_in_medium":
        kw = dict(sensor_medium='<ref id="haze"/>', exterior='<ref name="exterior" id="haze"/>', rf="tent", md="8")
    if variant == "constant_env":
        kw = dict(env='<emitter type="constant"><rgb name="radiance" value="0.3, 0.4, 0.6"/></emitter>', rf="box")
    sc = mi.load_string(fog_xml(**kw))
    o = orc.OrcScene(sc)
    assert_lanes_equal(sc, o, 0, 64 * 48 * 32)
    assert sc.stats()["n_shadow"] > 0


def test_furnace_on_gpu(mi):
    v, f = _cube()
    T = mi.ScalarTransform4f
    sc = mi.scene_from_buffers(v, f, reflectance=(1, 1, 1), film=(16, 16), fov=40.0, spp=512, integrator="volpath", max_depth=-1,
                               sensor_to_world=T().look_at([3, 2.5, 4], [0, 0, 0], [0, 1, 0]), constant_radiance=(1, 1, 1))
    img = sc.render()
    assert img.mean() == pytest.approx(1.0, abs=0.01)


# ------------------------------------------------------------------------- film
def film_close(g, c, weight_channel=-1, rtol=1e-5):
    """Float atomics reorder the per-pixel sums: compare relative to the pixel's accumulated weight."""
    fin = np.isfinite(c)
    if fin.all() and np.isfinite(g).all():
        scale = np.maximum(np.abs(c).max(axis=-1, keepdims=True), 1.0)
        return np.abs(g - c) <= rtol * scale * 8
    # a lane with a non-finite radiance (overflowing throughput in extreme random scenes) makes film values inf / NaN: the same values must be
    # non-finite on both sides (value * weight as imageblock.cpp computes it, zero weights included); the finite ones compare as usual
    with np.errstate(invalid="ignore"):
        scale = np.maximum(np.where(fin, np.abs(c), 0.0).max(axis=-1, keepdims=True), 1.0)
        return np.where(fin, np.abs(g - c) <= rtol * scale * 8, ~np.isfinite(g))


@pytest.mark.parametrize("kw", [dict(spp=16), dict(spp=5, seed=4), dict(spp=16, integrator="volpath")])
def test_cornell_film_matches_oracle(mi, orc, cornell, kw):
    """Gaussian reconstruction filter (5x5 splats).  Film tolerance: 8e-5 relative to the pixel's largest raw channel."""
    img, raw = cornell.render(return_raw=True, **kw)
    oimg, oraw = orc.OrcScene(cornell).render(return_raw=True, **kw)
    assert film_close(raw, oraw).all()
    assert np.allclose(img, oimg, rtol=2e-4, atol=1e-5)
    assert np.allclose(cornell.develop(raw), img, rtol=1e-6, atol=0)


def test_liver_film_matches_oracle_box_rgba(mi, orc, liver_small):
    img, raw = liver_small.render(return_raw=True)
    oimg, oraw = orc.OrcScene(liver_small).render(return_raw=True)
    assert raw.shape[-1] == 5 and img.shape[-1] == 4
    assert (raw[..., 4] == 16).all()                       # box filter: W = spp exactly
    assert (raw[..., 3] == 16).all()                       # alpha: envmap -> all rays valid
    assert film_close(raw, oraw).all()
    assert np.allclose(img, oimg, rtol=2e-4, atol=1e-6)


def test_cropped_film_known_answer(mi):
    """src/integrators/tests/test_integrators.py:28-53 on the GPU path."""
    d = mi.cornell_box()
    d['sensor']['film'].update({'crop_offset_x': 124, 'crop_offset_y': 36, 'crop_width': 1, 'crop_height': 1})
    sc = mi.load_dict(d)
    img = sc.render(integrator="path", max_depth=1, hide_emitters=False)
    assert img.shape == (1, 1, 3) and np.allclose(img[0, 0], [18.387, 13.9873, 6.75357], rtol=1e-5)
    assert np.allclose(sc.render(integrator="path", max_depth=1, hide_emitters=True), 0)


def test_principal_point_offset(mi, orc):
    """perspective.cpp:147-150,214-221: the principal point offset shifts the image; lanes bit-exact, and the shifted image is
    the unshifted one moved by offset x film size pixels."""
    d = mi.cornell_box(); d['sensor']['film'].update({'width': 64, 'height': 64}); d['sensor']['sampler']['sample_count'] = 16
    a = mi.load_dict(d)
    d['sensor']['principal_point_offset_x'] = 0.125; d['sensor']['principal_point_offset_y'] = -0.0625
    b = mi.load_dict(d)
    assert b.desc.sensor.principal_point_offset_x == 0.125
    assert_lanes_equal(b, orc.OrcScene(b), 0, 64 * 64 * 16)
    ia, ib = a.render(spp=64), b.render(spp=64)
    assert np.abs(ib[4:60, 0:56] - ia[0:56, 8:64]).mean() < 0.15 * ia.mean()      # same content, moved 8 px left / 4 px down


def test_crop_window_matches_full_render_lanes(mi, orc):
    d = mi.cornell_box()
    d['sensor']['film'].update({'crop_offset_x': 40, 'crop_offset_y': 100, 'crop_width': 50, 'crop_height': 30, 'rfilter': {'type': 'box'}})
    sc = mi.load_dict(d)
    assert_lanes_equal(sc, orc.OrcScene(sc), 0, 50 * 30 * 8, spp=8)
    img, raw = sc.render(spp=8, return_raw=True)
    assert img.shape == (30, 50, 3) and (raw[..., 3] == 8).all()


def test_tile_sharded_films_sum_to_full_film(mi, cornell, liver_small):
    """Multi-GPU partition (32x32 tiles, tile k -> rank k % G): the per-rank raw films add up to the
    1-GPU film (size-independent property of the sharding + RCCL sum)."""
    for sc, kw in ((liver_small, {}), (cornell, dict(spp=8))):
        _, full = sc.render(return_raw=True, **kw)
        for G in (2, 3, 8):
            acc = np.zeros_like(full)
            weights = []
            for r in range(G):
                _, part = sc.render(return_raw=True, tile_rank=r, tile_count=G, **kw)
                acc += part; weights.append(part[..., -1].sum())
            assert film_close(acc, full).all()
            assert np.isclose(sum(weights), full[..., -1].sum(), rtol=1e-5)
            assert min(weights) > 0


def test_determinism_and_seed_sensitivity(mi, cornell):
    a = cornell.render_samples(center_lane(cornell, 16), 1 << 14, spp=16, seed=1)
    b = cornell.render_samples(center_lane(cornell, 16), 1 << 14, spp=16, seed=1)
    c = cornell.render_samples(center_lane(cornell, 16), 1 << 14, spp=16, seed=2)
    assert (bits(a) == bits(b)).all() and not (bits(a) == bits(c)).all()


def test_full_size_c2_properties(mi):
    """BASELINE config C2 geometry (Cornell 1080x1080, Gaussian filter) at reduced spp: invariants that
    do not need the oracle at full size."""
    d = mi.cornell_box(); d['sensor']['film'].update({'width': 1080, 'height': 1080})
    sc = mi.load_dict(d)
    img, raw = sc.render(spp=4, return_raw=True)
    st = sc.stats()
    assert st["n_samples"] == 1080 * 1080 * 4
    assert np.isfinite(img).all() and (img >= 0).all()
    # interior pixels receive the full normalised filter mass: sum W = spp * sum_k f(k)^2 over the lattice
    assert np.isclose(raw[..., 3].sum() / (1080 * 1080 * 4), raw[500:580, 500:580, 3].mean() / 4, rtol=5e-3)
    half = sc.render(spp=4, integrator="path", max_depth=1)
    assert half.max() <= 18.387 * 1.0001 and half.max() > 18.0


def test_full_size_c3_properties(mi):
    """BASELINE config C3 geometry (Liver-SingleMesh 1920x1080 volpath) at reduced spp."""
    sc = mi.load_file(LIVER_XML, integrator="volpath", spp=2, res_width=1920, res_height=1080)
    img, raw = sc.render(return_raw=True)
    st = sc.stats()
    assert st["n_samples"] == 1920 * 1080 * 2 and st["n_iter"] >= st["n_samples"]
    assert (raw[..., 4] == 2).all() and (raw[..., 3] == 2).all()
    assert np.isfinite(img).all() and (img[..., :3] >= 0).all()
    # background pixels see the environment map directly: corner pixel equals 2.5 x envmap texel radiance order
    assert img[0, 0, :3].max() < 2.5 * 1.01
    # two-rank sharding reproduces the image
    a = sc.render(return_raw=True, tile_rank=0, tile_count=2)[1] + sc.render(return_raw=True, tile_rank=1, tile_count=2)[1]
    assert film_close(a, raw).all()


# ------------------------------------------------------------ low-discrepancy sampler
def test_ldsampler_lanes_bit_exact(mi, orc):
    """src/samplers/ldsampler.cpp (what Parenchyma, GlissonCapsule and Liver-MultiMesh request): the sampler itself is pinned
    by the reference's golden values (test_oracle_pins.py); here the integrators' dimension bookkeeping on the device
    must match the oracle's lane for lane, for path, volpath and the PRB primal, and spp is rounded to 4^k."""
    d = mi.cornell_box(); d['sensor']['sampler'] = {'type': 'ldsampler', 'sample_count': 16}
    sc = mi.load_dict(d)
    assert sc.desc.sampler_type == 1 and sc.spp == 16
    o = orc.OrcScene(sc)
    for kw in (dict(), dict(integrator="volpath"), dict(spp=5, seed=3), dict(spp=64, max_depth=3, integrator="volpath"), dict(integrator="prbvolpath", seed=1)):
        spp = {5: 16}.get(kw.get("spp", 16), kw.get("spp", 16))
        assert_lanes_equal(sc, o, center_lane(sc, spp, 0.3), 1 << 14, **kw)
    xml = open(LIVER_XML).read().replace('<sampler type="independent">', '<sampler type="ldsampler">')
    base = os.path.dirname(LIVER_XML)
    sc = mi.load_string(xml, base_dir=base, integrator="volpath", spp=16, res_width=192, res_height=108)
    assert sc.desc.sampler_type == 1
    o = orc.OrcScene(sc)
    assert_lanes_equal(sc, o, 0, 192 * 108 * 16)
    p = mi.traverse(sc); p["LiverMedium.phase_function.g"] = 0.6; p.update(); o.param_set("LiverMedium.phase_function.g", 0.6)
    assert_lanes_equal(sc, o, center_lane(sc, 16), 1 << 15, seed=7, max_depth=30)
    for path in (PARENCHYMA_XML, MULTIMESH_XML, GLISSON_XML):  # these files name the ld sampler themselves
        sc = mi.load_file(path, integrator="volpath", spp=10, res_width=160, res_height=90)
        assert sc.desc.sampler_type == 1 and sc.spp == 16
        assert_lanes_equal(sc, orc.OrcScene(sc), 0, 160 * 90 * 16)
    img, raw = sc.render(return_raw=True)
    assert raw[..., -1].min() > 0 and np.isfinite(img).all()


# ------------------------------------------------------------------ multi-pass renders
@pytest.mark.parametrize("case", ["cornell-independent-gaussian", "cornell-ld-gaussian", "liver-independent-box", "fog-tent-volpath"])
def test_multi_pass_render_matches_oracle(mi, orc, case):
    """`samples_per_pass` (integrator.cpp:176-184; the same loop splits renders of more than 2^32 - 1 samples, :275-293):
    every pass renders spp_per_pass samples per pixel, the independent sampler's streams run on from pass to pass
    (including the Russian-roulette draw of a path's last trip), the ld sampler advances its sample index."""
    if case.startswith("cornell"):
        d = mi.cornell_box(); d['sensor']['film'].update({'width': 64, 'height': 64})
        d['sensor']['sampler'] = {'type': 'ldsampler' if 'ld' in case else 'independent', 'sample_count': 16}
        d['integrator']['samples_per_pass'] = 4
        sc = mi.load_dict(d)
    elif case.startswith("liver"):
        xml = open(LIVER_XML).read().replace('<integer name="max_depth" value="$max_depth"/>', '<integer name="max_depth" value="$max_depth"/><integer name="samples_per_pass" value="2"/>')
        sc = mi.load_string(xml, base_dir=os.path.dirname(LIVER_XML), integrator="volpath", spp=8, res_width=160, res_height=90)
    else:
        sc = mi.load_string(fog_xml(rf="tent").replace('<integer name="max_depth" value="12"/>', '<integer name="max_depth" value="12"/><integer name="samples_per_pass" value="8"/>'))
    assert sc.desc.samples_per_pass in (2, 4, 8)
    o = orc.OrcScene(sc)
    img, raw = sc.render(return_raw=True, seed=2)
    st = sc.stats()
    oimg, oraw = o.render(return_raw=True, seed=2)
    assert st["n_samples"] == o.last_stats["n_samples"] and st["n_iter"] == o.last_stats["n_iter"] and st["n_shadow"] == o.last_stats["n_shadow_needed"]
    assert st["n_launches"] >= sc.spp // sc.desc.samples_per_pass
    assert film_close(raw, oraw).all()
    assert np.allclose(img, oimg, rtol=2e-4, atol=2e-5)
    # tile-sharded passes add up to the same film
    parts = sum(sc.render(return_raw=True, seed=2, tile_rank=r, tile_count=2)[1] for r in range(2))
    assert film_close(parts, raw).all()


# ------------------------------------------------- kernel variants and accelerators
@pytest.mark.parametrize("env", [dict(LRT_NO_LDS_BVH="1"), dict(LRT_NO_DIST_GRID="1"),
                                 dict(LRT_NO_NEE_REJECT="1"), dict(LRT_POOL="64"), dict(LRT_DIST_GRID_RES="24"),
                                 dict(LRT_BVH_LEAF="12"), dict(LRT_BVH_LEAF="1", LRT_NO_LDS_BVH="1")])
def test_kernel_variants_bit_exact(mi, orc, monkeypatch, env):
    """Every build-time decision of the device scene (BVH in LDS or in global memory, distance-field look-ahead, exact NEE
    rejection, pool size, leaf size of the BVH) changes speed only: lanes stay bit-identical to the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc = mi.load_file(LIVER_XML, integrator="volpath", spp=16, res_width=256, res_height=144)
    o = orc.OrcScene(sc)
    assert_lanes_equal(sc, o, 0, 256 * 144 * 16)
    st = sc.stats()
    assert st["n_records"] <= st["n_iter"] and st["n_launches"] == 1
    sc = mi.load_dict(mi.cornell_box())
    assert_lanes_equal(sc, orc.OrcScene(sc), center_lane(sc, 16), 1 << 14, spp=16, integrator="volpath")
    assert_lanes_equal(sc, orc.OrcScene(sc), center_lane(sc, 16), 1 << 14, spp=16)


def _icosphere(levels):
    p = (1 + 5 ** 0.5) / 2
    v = [(-1, p, 0), (1, p, 0), (-1, -p, 0), (1, -p, 0), (0, -1, p), (0, 1, p), (0, -1, -p), (0, 1, -p), (p, 0, -1), (p, 0, 1), (-p, 0, -1), (-p, 0, 1)]
    v = [np.array(x, np.float64) / np.linalg.norm(x) for x in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(levels):
        cache, nf = {}, []
        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]; v.append(m / np.linalg.norm(m)); cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, np.float32), np.array(f, np.uint32)


def test_large_mesh_in_global_memory(mi, orc, tmp_path):
    """A mesh far too large for the LDS image (81 920 triangles, 40 962 vertices): BVH in global memory, 64^3 distance
    field, fog inside a dielectric blob lit by a constant environment; lanes bit-exact, film within tolerance."""
    v, f = _icosphere(6)
    v = v * (1 + 0.15 * np.sin(5 * v[:, :1]) * np.cos(4 * v[:, 1:2])).astype(np.float32) * np.float32(2.0)
    with open(tmp_path / "blob.obj", "w") as fh:
        fh.write("".join(f"v {a:.7g} {b:.7g} {c:.7g}\n" for a, b, c in v) + "".join(f"f {a + 1} {b + 1} {c + 1}\n" for a, b, c in f))
    xml = f"""<scene version="3.0.0">
  <integrator type="volpath"><integer name="max_depth" value="10"/></integrator>
  <medium type="homogeneous" id="fog"><rgb name="sigma_t" value="2.0, 1.5, 1.0"/><rgb name="albedo" value="0.9, 0.8, 0.7"/><phase type="hg"><float name="g" value="0.3"/></phase></medium>
  <sensor type="perspective"><float name="fov" value="40"/>
    <transform name="to_world"><lookat origin="0, 1, 7" target="0, 0, 0" up="0, 1, 0"/></transform>
    <sampler type="independent"><integer name="sample_count" value="16"/></sampler>
    <film type="hdrfilm"><integer name="width" value="96"/><integer name="height" value="96"/><rfilter type="box"/></film>
  </sensor>
  <shape type="obj"><string name="filename" value="blob.obj"/><bsdf type="dielectric"/><ref name="interior" id="fog"/></shape>
  <emitter type="constant"><rgb name="radiance" value="1.0, 0.9, 0.8"/></emitter>
</scene>"""
    sc = mi.load_string(xml, base_dir=tmp_path)
    assert sc.desc.n_faces == 81920
    o = orc.OrcScene(sc)
    assert_lanes_equal(sc, o, center_lane(sc, 16, 0.45), 1 << 15)
    st = sc.stats()
    assert st["n_records"] > 0 and st["n_records"] < st["n_iter"]
    img, raw = sc.render(return_raw=True)
    assert film_close(raw, o.render(return_raw=True)[1]).all()


def test_lookahead_is_invisible_in_statistics(mi, monkeypatch):
    """The look-ahead retires trips early and skips ray queries, but n_iter / n_shadow count what the reference's loop does."""
    a = mi.load_file(LIVER_XML, integrator="volpath", spp=8, res_width=320, res_height=180)
    ia, ra = a.render(return_raw=True); sa = a.stats()
    monkeypatch.setenv("LRT_NO_DIST_GRID", "1"); monkeypatch.setenv("LRT_NO_LDS_BVH", "1")
    b = mi.load_file(LIVER_XML, integrator="volpath", spp=8, res_width=320, res_height=180)
    ib, rb = b.render(return_raw=True); sb = b.stats()
    assert sa["n_iter"] == sb["n_iter"] and sa["n_shadow"] == sb["n_shadow"] and sa["n_samples"] == sb["n_samples"]
    assert film_close(ra, rb).all()


# ------------------------------------------------------- scenes from POD descriptions
def test_scene_from_desc_round_trip(mi, orc):
    """`lrt_scene_from_desc` (the "from buffers" entry of the ABI) with everything a description can hold: the POD
    description of a scene loaded from XML creates a second, independent scene whose lanes are bit-identical."""
    import ctypes as C
    from liverrenderer_amd import _lib
    for src in (mi.load_file(LIVER_XML, integrator="volpath", spp=8, res_width=96, res_height=54),
                mi.load_file(PARENCHYMA_XML, integrator="volpath", spp=16, res_width=96, res_height=54),
                mi.load_string(fog_xml(rf="tent")), mi.load_dict(mi.cornell_box())):
        h = C.c_void_p()
        _lib.check(_lib.lib().lrt_scene_from_desc(C.byref(src.desc), C.byref(h)))
        copy = mi.Scene(h.value)
        n = min(1 << 14, src.film_shape()[0] * src.film_shape()[1] * src.spp)
        a, b = src.render_samples(0, n, seed=3), copy.render_samples(0, n, seed=3)
        assert (bits(a) == bits(b)).all()
        assert copy.desc.n_faces == src.desc.n_faces and copy.desc.sampler_type == src.desc.sampler_type
        ia, ib = src.render(spp=4, seed=1), copy.render(spp=4, seed=1)
        assert np.allclose(ia, ib, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------ edge cases
def test_edge_case_scenes(mi, orc):
    """Degenerate inputs the reference accepts: no geometry at all (environment only), no emitter at all (black image), a
    1x1 film, 1 spp, a non-power-of-two spp with a crop window that leaves a single column."""
    env_only = """<scene version="3.0.0"><integrator type="volpath"/>
      <sensor type="perspective"><sampler type="independent"><integer name="sample_count" value="3"/></sampler>
        <film type="hdrfilm"><integer name="width" value="9"/><integer name="height" value="5"/><rfilter type="box"/></film></sensor>
      <emitter type="constant"><rgb name="radiance" value="0.25, 0.5, 1"/></emitter></scene>"""
    sc = mi.load_string(env_only)
    g = assert_lanes_equal(sc, orc.OrcScene(sc), 0, 9 * 5 * 3)
    assert (g[:, :3] == np.float32([0.25, 0.5, 1.0])).all() and (g[:, 3] == 1).all()
    img = sc.render()
    assert np.allclose(img, [0.25, 0.5, 1.0])
    dark = env_only.replace('<emitter type="constant"><rgb name="radiance" value="0.25, 0.5, 1"/></emitter>',
                            '<shape type="cube"><bsdf type="diffuse"/></shape>').replace('type="volpath"', 'type="path"')
    sc = mi.load_string(dark)
    g = assert_lanes_equal(sc, orc.OrcScene(sc), 0, 9 * 5 * 3)
    assert (g[:, :3] == 0).all() and (sc.render() == 0).all()
    d = mi.cornell_box(); d['sensor']['film'].update({'width': 1, 'height': 1}); d['sensor']['sampler']['sample_count'] = 1
    sc = mi.load_dict(d)
    assert_lanes_equal(sc, orc.OrcScene(sc), 0, 1)
    assert sc.render().shape == (1, 1, 3)
    d = mi.cornell_box(); d['sensor']['film'].update({'width': 33, 'height': 17, 'crop_offset_x': 20, 'crop_offset_y': 3, 'crop_width': 1, 'crop_height': 11})
    d['sensor']['sampler']['sample_count'] = 7
    sc = mi.load_dict(d); o = orc.OrcScene(sc)
    assert sc.film_shape()[:2] == (11, 1)
    assert_lanes_equal(sc, o, 0, 11 * 7)
    assert film_close(sc.render(return_raw=True)[1], o.render(return_raw=True)[1]).all()


# -------------------------------------------------------------------- error paths
def test_error_reporting(mi, cornell):
    with pytest.raises(RuntimeError, match="tile_rank"):
        cornell.render(spp=1, tile_rank=3, tile_count=2)
    with pytest.raises(RuntimeError):
        cornell.param_set("nope.sigma_t.value", [1, 1, 1])
    d = mi.cornell_box(); d['integrator']['samples_per_pass'] = 24
    with pytest.raises(RuntimeError, match="multiple of spp_per_pass"):           # integrator.cpp:180-182
        mi.load_dict(d).render(spp=64)


# --------------------------------------------------------------------------- PRB
from test_oracle_pins import prb_scene_xml, PRB_ENV, PRB_AREA


@pytest.mark.parametrize("case", ["null+area", "dielectric+env", "tent", "ldsampler"])
def test_prb_primal_lanes_and_gradients(mi, orc, case):
    """prbvolpath primal lanes bit-exact; adjoint gradients equal to the oracle's up to summation order
    (float partial sums per workgroup vs. double per lane: 2e-4 relative to the gradient's scale)."""
    if case == "null+area": xml = prb_scene_xml("null", PRB_AREA, res=16)
    elif case == "dielectric+env": xml = prb_scene_xml("dielectric", PRB_ENV, res=16)
    elif case == "ldsampler": xml = prb_scene_xml("null", PRB_AREA, res=16).replace('<sampler type="independent">', '<sampler type="ldsampler">')
    else: xml = prb_scene_xml("null", PRB_AREA, rf="tent", res=16)
    sc = mi.load_string(xml); o = orc.OrcScene(sc)
    spp = 64
    assert_lanes_equal(sc, o, 0, 16 * 16 * spp, spp=spp, integrator="prbvolpath")
    rng = np.random.default_rng(3)
    H, W, T = sc.film_shape()
    grad = rng.random((H, W, T)).astype(np.float32) / (H * W * T)
    gg = sc.render_backward(grad, spp=spp, seed=2)
    gc = o.render_backward(grad, spp=spp, seed=2)
    for k in ("sigma_t", "albedo"):
        assert np.abs(gg[k] - gc[k]).max() <= 2e-4 * np.abs(gc[k]).max(), (k, gg[k], gc[k])
    assert abs(gg["g"] - gc["g"]) <= 2e-4 * max(abs(gc["g"]), 1e-6)
    # sharded adjoint: per-rank gradients add up (the 7-float all-reduce of SURVEY 8e)
    parts = [sc.render_backward(grad, spp=spp, seed=2, tile_rank=r, tile_count=2) for r in range(2)]
    for k in ("sigma_t", "albedo"):
        assert np.abs(parts[0][k] + parts[1][k] - gc[k]).max() <= 2e-4 * np.abs(gc[k]).max()


def test_prb_parenchyma_config5_geometry(mi, orc):
    """BASELINE config C5 geometry (Parenchyma camera + liver mesh, homogeneous reading) at small size."""
    sc = mi.load_file(PARENCHYMA_XML, integrator="prbvolpath", spp=16, res_width=96, res_height=54, max_depth=65)
    o = orc.OrcScene(sc)
    for s_, o_ in ((sc, o),):
        s_.param_set("parenchymaMedium.sigma_t.value", [0.05, 0.08, 0.11]); o_.param_set("parenchymaMedium.sigma_t.value", [0.05, 0.08, 0.11])
    H, W, T = sc.film_shape()
    grad = np.full((H, W, T), 1.0 / (H * W * T), np.float32)
    gg = sc.render_backward(grad, seed=7); gc = o.render_backward(grad, seed=7)
    for k in ("sigma_t", "albedo"):
        assert np.abs(gg[k] - gc[k]).max() <= 5e-4 * np.abs(gc[k]).max(), (k, gg[k], gc[k])
    assert np.abs(gc["albedo"]).max() > 0


def test_device_math_kernels_accuracy_and_twins(mi, orc):
    """csrc/dmath.h and oracle/orc_math.h are the same polynomial kernels written twice, so the bit-identity
    of render lanes says nothing about their closeness to the functions Dr.Jit evaluates.  Here the DEVICE kernels are bounded against
    float64 (the bounds of test_oracle_pins.py::test_math_kernels_accuracy) and compared bit for bit with the oracle's; division,
    square root and reciprocal are checked to be correctly rounded (the reference's `/`, dr::sqrt, dr::rcp up to its own rcp
    approximation)."""
    from test_oracle_pins import _ulp_err
    rng = np.random.default_rng(0)
    def both(fn, x, y=None):
        d = mi.math_eval(fn, x, y); c = orc.math_eval(fn, x, y)
        assert np.array_equal(d[0].view(np.uint32), c[0].view(np.uint32)) and (fn != 2 or np.array_equal(d[1].view(np.uint32), c[1].view(np.uint32))), fn
        return d
    x = np.concatenate([1 - rng.random(200000), rng.random(200000) * 1e-3 + 1e-7, np.exp(rng.uniform(-80, 80, 100000))]).astype(np.float32)
    assert _ulp_err(both(0, x)[0], np.log(x.astype(np.float64))).max() <= 2.0
    assert _ulp_err(both(5, x)[0], np.log2(x.astype(np.float64))).max() <= 2.5
    x = (-rng.random(400000) * 80).astype(np.float32)
    assert _ulp_err(both(1, x)[0], np.exp(x.astype(np.float64))).max() <= 2.0
    x = (rng.random(400000) * 2 * np.pi).astype(np.float32)
    s, c = both(2, x)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 2e-7 and np.abs(c - np.cos(x.astype(np.float64))).max() < 2e-7
    xx = rng.normal(size=400000).astype(np.float32); yy = rng.normal(size=400000).astype(np.float32)
    assert np.abs(both(3, xx, yy)[0] - np.arctan2(yy.astype(np.float64), xx.astype(np.float64))).max() < 5e-7
    x = (rng.random(400000) * 2 - 1).astype(np.float32)
    assert np.abs(both(4, x)[0] - np.arccos(x.astype(np.float64))).max() < 5e-7
    assert mi.math_eval(1, np.array([-200.0], np.float32))[0][0] == 0.0 and mi.math_eval(0, np.array([1.0], np.float32))[0][0] == 0.0
    # IEEE: the device's division, square root and reciprocal are the correctly rounded ones (gfx950: -fhip-fp32-correctly-rounded-divide-sqrt default)
    a = np.exp(rng.uniform(-40, 40, 400000)).astype(np.float32) * rng.choice([-1, 1], 400000).astype(np.float32); b = np.exp(rng.uniform(-40, 40, 400000)).astype(np.float32)
    assert np.array_equal(mi.math_eval(6, a, b)[0].view(np.uint32), (a / b).view(np.uint32))
    assert np.array_equal(mi.math_eval(7, b)[0].view(np.uint32), np.sqrt(b).view(np.uint32))
    assert np.array_equal(mi.math_eval(8, b)[0].view(np.uint32), (np.float32(1) / b).view(np.uint32))
