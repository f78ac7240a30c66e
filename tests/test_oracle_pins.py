"""Pins the CPU oracle against every in-tree known answer the reference holds for
the hot path, plus published vectors and analytic furnace tests.
All tests run on the CPU (no GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT


# ---- src/core/tests/test_random.py:9-26 -------------------------------------------------
TEA32 = [((1, 1), 0.5424730777740479), ((1, 2), 0.5079904794692993), ((1, 3), 0.4171961545944214),
         ((1, 4), 0.008385419845581055), ((1, 5), 0.8085528612136841), ((2, 1), 0.6939879655838013),
         ((3, 1), 0.6978365182876587), ((4, 1), 0.4897364377975464)]
TEA64 = [((1, 1), 0.5424730799533735), ((1, 2), 0.5079905082233922), ((1, 3), 0.4171962610608142),
         ((1, 4), 0.008385529523330604), ((1, 5), 0.80855288317879), ((2, 1), 0.6939880404156831),
         ((3, 1), 0.6978365636630994), ((4, 1), 0.48973647949223253)]


def test_tea_float32_reference_vectors(orc):
    L = orc.lib()
    for (v0, v1), expected in TEA32:
        assert L.orc_tea_float32(v0, v1, 4) == np.float32(expected)


def test_tea_float64_reference_vectors(orc):
    L = orc.lib()
    for (v0, v1), expected in TEA64:
        assert L.orc_tea_float64(v0, v1, 4) == expected


# src/samplers/tests/test_ldsampler.py:43-77 (test03_ldsampler_deterministic_values): scalar sampler, sample_count 1024,
# seed(0); ten next_1d, ten next_2d, advance(), ten next_1d, ten next_2d
LD_1D_S0 = [0.06483602523803711, 0.2767500877380371, 0.006242275238037109, 0.6273360252380371, 0.6273360252380371,
            0.1634688377380371, 0.3734297752380371, 0.7845625877380371, 0.7415938377380371, 0.4085860252380371]
LD_2D_S0 = [[0.014822006225585938, 0.06154896318912506], [0.34001731872558594, 0.7011973857879639], [0.7570095062255859, 0.8974864482879639],
            [0.7374782562255859, 0.8076426982879639], [0.7794704437255859, 0.26174429059028625], [0.5060329437255859, 0.8945567607879639],
            [0.16911888122558594, 0.6533458232879639], [0.34294700622558594, 0.9521739482879639], [0.16618919372558594, 0.09377552568912506],
            [0.15935325622558594, 0.44826772809028625]]
LD_1D_S1 = [0.7005782127380371, 0.9085860252380371, 0.7601485252380371, 0.3939375877380371, 0.4876875877380371,
            0.6576094627380371, 0.6908125877380371, 0.2113204002380371, 0.4847579002380371, 0.7865157127380371]
LD_2D_S1 = [[0.8956813812255859, 0.03615833818912506], [0.8400173187255859, 0.20119740068912506], [0.27556419372558594, 0.23440052568912506],
            [0.026540756225585938, 0.23733021318912506], [0.0021266937255859375, 0.6015880107879639], [0.38884544372558594, 0.6836192607879639],
            [0.8956813812255859, 0.03615833818912506], [0.9542751312255859, 0.02443958818912506], [0.8409938812255859, 0.9502208232879639],
            [0.5636501312255859, 0.6650645732879639]]


def test_ldsampler_reference_vectors(orc):
    """The reference's own golden values for LowDiscrepancySampler pin the oracle's sampler (TEA shuffling-network permutation,
    radical inverse, Sobol' dimension 2, TEA scrambles, dimension counter) bit for bit."""
    L = orc.lib()
    v0, v1 = C.c_uint32(), C.c_uint32()
    L.orc_tea32(0, 0, 4, C.byref(v0), C.byref(v1))            # compute_per_sequence_seed(0): TEA(base_seed 0, sequence 0 + seed 0)
    out = (C.c_float * 2)()
    def draw(idx, dim, two):
        L.orc_ld_sample(1024, v0.value, idx, dim, two, out)
        return [out[0], out[1]] if two else out[0]
    for idx, v1d, v2d in ((0, LD_1D_S0, LD_2D_S0), (1, LD_1D_S1, LD_2D_S1)):
        assert [draw(idx, k, 0) for k in range(10)] == [float(np.float32(x)) for x in v1d]
        assert [draw(idx, 10 + k, 1) for k in range(10)] == [[float(np.float32(a)), float(np.float32(b))] for a, b in v2d]
    assert [L.orc_ld_round_sample_count(n) for n in (1, 4, 5, 16, 17, 64, 128, 256, 512, 1000)] == [4, 4, 16, 16, 64, 64, 256, 256, 1024, 1024]
    for n in (4, 16, 64, 256):                                 # the shuffling network is a bijection for every seed
        for seed in (0, 1, 0xdeadbeef):
            assert sorted(L.orc_permute(i, n, seed) for i in range(n)) == list(range(n))


def test_pcg32_published_vectors(orc):
    # O'Neill's pcg32-demo: seed(42, 54) -> first six outputs (Dr.Jit's PCG32 is this generator;
    # the reference's own test src/samplers/tests/test_independent.py:21-33 only pins sampler == PCG32)
    out = np.zeros(6, np.uint32)
    orc.lib().orc_pcg32_u32(42, 54, 6, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert [hex(x) for x in out] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def _py_pcg32(initstate, initseq, n):
    M = (1 << 64) - 1
    state, inc = 0, ((initseq << 1) | 1) & M
    def nxt():
        nonlocal state
        old = state
        state = (old * 0x5851f42d4c957f2d + inc) & M
        xs = (((old >> 18) ^ old) >> 27) & 0xffffffff
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xffffffff
    nxt(); state = (state + initstate) & M; nxt()
    return [nxt() for _ in range(n)]


def test_lane_stream_is_tea_seeded_pcg32(orc):
    # src/render/sampler.cpp:129-148: (v0, v1) = TEA4(base_seed + seed, lane); rng.seed(v0, v1);
    # float = ((u32 >> 9) | 0x3f800000) - 1 (include/mitsuba/core/random.h:134-139)
    L = orc.lib()
    for base, seed, lane in [(0, 0, 0), (0, 0, 1), (0, 7, 12345), (3, 1, 2 ** 31 + 5)]:
        a, b = C.c_uint32(), C.c_uint32()
        L.orc_tea32(base + seed, lane, 4, C.byref(a), C.byref(b))
        u = np.array(_py_pcg32(a.value, b.value, 64), dtype=np.uint32)
        expected = ((u >> 9) | 0x3f800000).view(np.float32) - np.float32(1)
        got = np.zeros(64, np.float32)
        L.orc_lane_stream(base, seed, lane, 64, got.ctypes.data_as(C.POINTER(C.c_float)))
        assert (got == expected).all()


# ---- transcendental kernels ---------------------------------------------------------------
def _ulp_err(got, ref):
    ref32 = ref.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref) / np.maximum(ulp, 1e-45)


def test_math_kernels_accuracy(orc):
    rng = np.random.default_rng(0)
    x = np.concatenate([1 - rng.random(20000), rng.random(20000) * 1e-3 + 1e-7]).astype(np.float32)
    assert _ulp_err(orc.math_eval(0, x)[0], np.log(x.astype(np.float64))).max() <= 2.0
    x = (-rng.random(40000) * 80).astype(np.float32)
    assert _ulp_err(orc.math_eval(1, x)[0], np.exp(x.astype(np.float64))).max() <= 2.0
    x = (rng.random(40000) * 2 * np.pi).astype(np.float32)
    s, c = orc.math_eval(2, x)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 2e-7 and np.abs(c - np.cos(x.astype(np.float64))).max() < 2e-7
    xx = rng.normal(size=40000).astype(np.float32); yy = rng.normal(size=40000).astype(np.float32)
    assert np.abs(orc.math_eval(3, xx, yy)[0] - np.arctan2(yy.astype(np.float64), xx.astype(np.float64))).max() < 5e-7
    x = (rng.random(40000) * 2 - 1).astype(np.float32)
    assert np.abs(orc.math_eval(4, x)[0] - np.arccos(x.astype(np.float64))).max() < 5e-7
    assert orc.math_eval(1, np.array([-200.0], np.float32))[0][0] == 0.0
    assert orc.math_eval(0, np.array([1.0], np.float32))[0][0] == 0.0


# ---- phase functions: src/phase/tests/test_isotropic.py:11-21, test_hg.py:10-21 -----------
def test_isotropic_phase_value(orc):
    assert orc.lib().orc_hg_eval(0.0, 0.3) == pytest.approx(1.0 / (4 * np.pi), rel=1e-6)
    out = np.zeros(3, np.float32); pdf = C.c_float()
    wi = np.array([0, 0, 1], np.float32)
    FP = C.POINTER(C.c_float)
    orc.lib().orc_hg_sample(0.0, wi.ctypes.data_as(FP), 0.3, 0.6, out.ctypes.data_as(FP), C.byref(pdf))
    assert pdf.value == pytest.approx(1.0 / (4 * np.pi), rel=1e-6)


@pytest.mark.parametrize("g", [0.6, -0.6, 0.7])
def test_hg_sampling_matches_pdf(orc, g):
    """chi^2-style check (src/python/python/chi2.py): histogram of cos(theta) vs integrated pdf."""
    L = orc.lib(); FP = C.POINTER(C.c_float)
    rng = np.random.default_rng(5)
    n = 200000
    wi = np.array([0.3, -0.5, 0.81], np.float64); wi /= np.linalg.norm(wi); wi = wi.astype(np.float32)
    cos = np.zeros(n); wo = np.zeros(3, np.float32); pdf = C.c_float()
    u = rng.random((n, 2)).astype(np.float32)
    for i in range(n):
        L.orc_hg_sample(g, wi.ctypes.data_as(FP), float(u[i, 0]), float(u[i, 1]), wo.ctypes.data_as(FP), C.byref(pdf))
        cos[i] = np.dot(wo.astype(np.float64), wi)
        if i < 100:                       # sample() pdf == eval_pdf(wo) (hg.cpp:89,97)
            assert pdf.value == pytest.approx(L.orc_hg_eval(g, float(np.dot(wo, wi))), rel=2e-4)
            assert abs(np.linalg.norm(wo) - 1) < 1e-5
    bins = np.linspace(-1, 1, 21)
    hist, _ = np.histogram(cos, bins)
    def cdf(c):  # integral of 2*pi*hg(c') dc' from -1 to c
        return (1 - g * g) / (2 * g) * (1 / np.sqrt(1 + g * g + 2 * g * (-1)) - 1 / np.sqrt(1 + g * g + 2 * g * c))
    expected = n * np.diff(cdf(bins))
    chi2 = ((hist - expected) ** 2 / expected).sum()
    assert chi2 < 60.0, chi2             # 19 dof: p(chi2 > 60) ~ 4e-6


def test_cosine_hemisphere_and_uniform_sphere(orc):
    L = orc.lib(); FP = C.POINTER(C.c_float)
    rng = np.random.default_rng(2); v = np.zeros(3, np.float32)
    zs = []
    for _ in range(20000):
        L.orc_square_to_cosine_hemisphere(rng.random(), rng.random(), v.ctypes.data_as(FP))
        assert v[2] >= 0 and abs(np.linalg.norm(v) - 1) < 1e-5
        zs.append(v[2])
    assert np.mean(zs) == pytest.approx(2 / 3, abs=0.01)      # E[cos] under the cosine density
    zs = []
    for _ in range(20000):
        L.orc_square_to_uniform_sphere(rng.random(), rng.random(), v.ctypes.data_as(FP))
        assert abs(np.linalg.norm(v) - 1) < 1e-5
        zs.append(v[2])
    assert abs(np.mean(zs)) < 0.02 and np.mean(np.square(zs)) == pytest.approx(1 / 3, abs=0.01)


def test_fresnel_known_values(orc):
    out = np.zeros(4, np.float32); FP = C.POINTER(C.c_float)
    orc.lib().orc_fresnel(1.0, 1.5, out.ctypes.data_as(FP))
    assert out[0] == pytest.approx(0.04, rel=1e-5) and out[1] == pytest.approx(-1.0) and out[2] == pytest.approx(1.5)
    orc.lib().orc_fresnel(-0.2, 1.5, out.ctypes.data_as(FP))       # inside, beyond the critical angle: TIR
    assert out[0] == 1.0
    orc.lib().orc_fresnel(0.5, 1.0, out.ctypes.data_as(FP))        # index matched
    assert out[0] == 0.0


# ---- src/integrators/tests/test_integrators.py:28-53 --------------------------------------
def test_cornell_box_directly_visible_emitter(mi, orc):
    d = mi.cornell_box()
    d['sensor']['film'].update({'crop_offset_x': 124, 'crop_offset_y': 36, 'crop_width': 1, 'crop_height': 1})
    sc = mi.load_dict(d)
    o = orc.OrcScene(sc)
    img = o.render(integrator="path", max_depth=1, hide_emitters=False)
    assert img.shape == (1, 1, 3)
    assert np.allclose(img[0, 0], [18.387, 13.9873, 6.75357], rtol=1e-5)
    img = o.render(integrator="path", max_depth=1, hide_emitters=True)
    assert np.allclose(img, 0)


# ---- src/render/tests/test_kdtrees.py:8-83 -------------------------------------------------
def _stairs(num_steps):
    size = 1.0 / num_steps
    v = np.zeros((4 * num_steps, 3), np.float32); f = np.zeros((4 * num_steps - 2, 3), np.uint32)
    for i in range(num_steps):
        h, s1, s2, k = i * size, i * size, (i + 1) * size, 4 * i
        v[k], v[k + 1], v[k + 2], v[k + 3] = [0, s1, h], [1, s1, h], [0, s2, h], [1, s2, h]
        f[k], f[k + 1] = [k, k + 1, k + 2], [k + 1, k + 3, k + 2]
        if i < num_steps - 1:
            f[k + 2], f[k + 3] = [k + 2, k + 3, k + 5], [k + 5, k + 4, k + 2]
    return v, f


def stairs_rays(n=128):
    inv_n = 1.0 / (n - 1)
    xs, ys = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    o = np.stack([xs.ravel() * inv_n, ys.ravel() * inv_n, np.full(xs.size, 2.0)], 1).astype(np.float32)
    d = np.tile(np.array([0, 0, -1], np.float32), (o.shape[0], 1))
    return o, d, ys.ravel() * inv_n


def test_staircase_ray_depths(mi, orc):
    n_steps = 20
    v, f = _stairs(n_steps)
    sc = mi.scene_from_buffers(v, f)
    o_ = orc.OrcScene(sc)
    o, d, yy = stairs_rays()
    tmax = np.full(o.shape[0], 100.0, np.float32)
    expected = (2.0 - np.floor(np.float32(yy) * n_steps) / n_steps).astype(np.float32)
    for brute in (True, False):
        t, u, vv, prim = o_.trace(o, d, tmax, brute_force=brute)
        assert np.isfinite(t).all()
        assert np.allclose(t, expected, atol=1e-6)
    tb = o_.trace(o, d, tmax, brute_force=True); ta = o_.trace(o, d, tmax, brute_force=False)
    assert (tb[0] == ta[0]).all() and (tb[3] == ta[3]).all()
    shadow = o_.trace(o, d, tmax, any_hit=True)[0]
    assert (shadow == 0).all()


# ---- weak image golden: /cornell_box.exr of the reference (committed under tests/golden) ----
def test_oracle_cornell_vs_reference_exr(mi, orc):
    """The reference ships cornell_box.exr (256^2 RGB f32, PIZ; spp unknown).  The oracle's C1 render
    must agree with it up to Monte-Carlo noise: channel means within 1 %, 8x8-block means within noise."""
    ref = mi.read_image(os.path.join(ROOT, "tests", "golden", "reference_cornell_box.exr"))
    sc = mi.load_dict(mi.cornell_box())
    img = orc.OrcScene(sc).render(spp=16)
    assert ref.shape == img.shape == (256, 256, 3)
    assert np.allclose(img.mean((0, 1)), ref.mean((0, 1)), rtol=0.01)
    blk = lambda x: x.reshape(32, 8, 32, 8, 3).mean((1, 3))
    a, b = blk(img), blk(ref)
    rel = np.abs(a - b) / (b + 0.05)
    assert np.percentile(rel, 95) < 0.08


# ---- analytic furnace tests ------------------------------------------------------------------
def _cube():
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.uint32)
    return v, f


def test_cornell_1080_weak_golden(mi, orc):
    """Weak image golden for config C2: the reference tree's own 1080x1080 Cornell render (8-bit sRGB PNG; spp, depth and
    variant are not recorded), decoded to linear and box-averaged 8x8 (tests/golden/make_cornell_1080_small.py), against
    the oracle at 135x135.  It pins geometry, materials and light transport at the few-percent level only."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_cornell_box_1080_down8.npy")).astype(np.float64)
    d = mi.cornell_box(); d['sensor']['film'].update({'width': 135, 'height': 135})
    img = orc.OrcScene(mi.load_dict(d)).render(spp=128, seed=0).astype(np.float64)[..., :3]
    c = np.clip(img, 0, 1)
    ok = (g < 0.9).all(-1) & (c < 0.9).all(-1)                  # away from the clipped emitter
    assert np.allclose(c[ok].mean(0), g[ok].mean(0), rtol=0.10)
    assert np.corrcoef(c[ok].ravel(), g[ok].ravel())[0, 1] > 0.995
    assert np.abs(c - g)[ok].mean() < 0.06 * g[ok].mean()


def cornell_fog_scene(mi, width, spp):
    """MitsubaRunner.py:8-40: mi.cornell_box() + homogeneous fog (sigma_t 0.2, albedo 0.75, scale 2.5, isotropic) attached to the
    SENSOR only (no shape references it: the fog fills all space), volpath with max_depth -1."""
    d = mi.cornell_box()
    d['fog_medium_id'] = {'type': 'homogeneous', 'sigma_t': {'type': 'rgb', 'value': [0.2, 0.2, 0.2]},
                          'albedo': {'type': 'rgb', 'value': [0.75, 0.75, 0.75]}, 'scale': 2.5, 'phase': {'type': 'isotropic'}}
    d['integrator'] = {'type': 'volpath', 'max_depth': -1}
    d['sensor']['film'].update({'width': width, 'height': width})
    d['sensor']['sampler']['sample_count'] = spp
    d['sensor']['medium'] = {'type': 'ref', 'id': 'fog_medium_id'}
    return mi.load_dict(d)


def test_cornell_fog_weak_golden(mi, orc):
    """The only reference-held output of `volpath` proper (free flight, medium emitter sampling, phase sampling, sensor inside a
    medium): the tree's own 1080x1080 fog render (8-bit sRGB PNG, 4096 spp per MitsubaRunner.py), decoded to linear and
    box-averaged 8x8 (tests/golden/make_cornell_fog_small.py), against the oracle at 135x135.  Observed at 1024 spp: mean
    radiance +2.0 .. +2.6 % per channel, the fog-only border +0.7 .. +1.5 % (the plain Cornell golden is off by -2.3 % in red)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_cornell_box_fog_1080_down8.npy")).astype(np.float64)
    sc = cornell_fog_scene(mi, 135, 128)
    assert sc.desc.sensor.medium == 0 and sc.desc.n_media == 1 and sc.desc.integrator.max_depth == -1
    c = np.clip(orc.OrcScene(sc).render(seed=0).astype(np.float64)[..., :3], 0, 1)
    ok = (g < 0.9).all(-1) & (c < 0.9).all(-1)                  # away from the clipped emitter
    assert np.allclose(c[ok].mean(0), g[ok].mean(0), rtol=0.06)
    border = np.zeros((135, 135), bool); border[:, :3] = True; border[:, -3:] = True       # outside the box: in-scattered light only
    assert np.allclose(c[border].mean(0), g[border].mean(0), rtol=0.05)
    plain = np.load(os.path.join(ROOT, "tests", "golden", "reference_cornell_box_1080_down8.npy")).astype(np.float64)
    assert g[ok].mean() < 0.5 * plain[ok].mean() and g[border][:, 2].mean() > 0.002         # the fog is what is being compared


def test_liver_singlemesh_weak_golden(mi, orc):
    """The reference's own render of the C3 scene (committed PNG, 1920x1080, fork's biovolpath integrator) pins everything
    that does not depend on the in-tissue transport: where the environment map is seen directly the images agree to
    ~1e-3 (camera, envmap lookup, orientation, scale, film), and the liver silhouettes coincide (mesh transform).
    Fixture: tests/golden/make_liver_singlemesh_small.py (linear, 8x8 box average)."""
    import re
    from conftest import LIVER_XML
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_liver_singlemesh_gpu_down8.npy")).astype(np.float64)
    xml = open(LIVER_XML).read(); base = os.path.dirname(LIVER_XML)
    kw = dict(base_dir=base, integrator="volpath", res_width=240, res_height=135)
    img = np.clip(orc.OrcScene(mi.load_string(xml, spp=16, **kw)).render().astype(np.float64)[..., :3], 0, 1)
    env_only = re.sub(r'<shape type="obj".*?</shape>', '', xml, flags=re.S)
    E = np.clip(orc.OrcScene(mi.load_string(env_only, spp=4, **kw)).render().astype(np.float64)[..., :3], 0, 1)
    mg, mo = np.abs(g - E).max(-1) > 0.08, np.abs(img - E).max(-1) > 0.08
    assert 0.3 < mg.mean() < 0.7 and (mg & mo).sum() / (mg | mo).sum() > 0.98
    bg = ~(mg | mo)
    assert np.abs(g - img)[bg].mean() < 2e-3 and np.allclose(g[bg].mean(0), img[bg].mean(0), rtol=5e-3)


@pytest.mark.parametrize("integrator", ["path", "volpath"])
def test_white_furnace_surface(mi, orc, integrator):
    """Reflectance-1 diffuse cube under a radiance-1 constant environment: every pixel converges to 1."""
    v, f = _cube()
    T = mi.ScalarTransform4f
    sc = mi.scene_from_buffers(v, f, reflectance=(1, 1, 1), film=(16, 16), fov=40.0, spp=256, integrator=integrator, max_depth=-1,
                               sensor_to_world=T().look_at([3, 2.5, 4], [0, 0, 0], [0, 1, 0]), constant_radiance=(1, 1, 1))
    img = orc.OrcScene(sc).render()
    assert img.mean() == pytest.approx(1.0, abs=0.01)
    assert np.abs(img.mean(2) - 1).max() < 0.12


def test_white_furnace_medium(mi, orc):
    """Albedo-1 homogeneous medium (HG g=0.5) behind an index-matched null boundary, radiance-1 environment."""
    xml = """<scene version="3.0.0">
      <integrator type="volpath"><integer name="max_depth" value="-1"/></integrator>
      <sensor type="perspective"><float name="fov" value="40"/>
        <transform name="to_world"><lookat origin="3, 2.5, 4" target="0, 0, 0" up="0, 1, 0"/></transform>
        <sampler type="independent"><integer name="sample_count" value="128"/></sampler>
        <film type="hdrfilm"><integer name="width" value="16"/><integer name="height" value="16"/><rfilter type="box"/></film>
      </sensor>
      <medium type="homogeneous" id="fog"><rgb name="sigma_t" value="1.5, 0.7, 2.0"/><rgb name="albedo" value="1, 1, 1"/>
        <phase type="hg"><float name="g" value="0.5"/></phase></medium>
      <shape type="cube"><bsdf type="null"/><ref name="interior" id="fog"/></shape>
      <emitter type="constant"><rgb name="radiance" value="1, 1, 1"/></emitter>
    </scene>"""
    sc = mi.load_string(xml)
    img = orc.OrcScene(sc).render()
    assert img.mean() == pytest.approx(1.0, abs=0.015)


def _slab_xml(sigma_t, albedo, integrator, spp, max_depth=-1, radiance="1, 1, 1", phase='<phase type="isotropic"/>'):
    # camera on the -z side looking along +z through a cube [-1,1]^3 filled with the medium; null boundary
    return f"""<scene version="3.0.0">
      <integrator type="{integrator}"><integer name="max_depth" value="{max_depth}"/></integrator>
      <sensor type="perspective"><float name="fov" value="2"/>
        <transform name="to_world"><lookat origin="0, 0, -20" target="0, 0, 0" up="0, 1, 0"/></transform>
        <sampler type="independent"><integer name="sample_count" value="{spp}"/></sampler>
        <film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/><rfilter type="box"/></film>
      </sensor>
      <medium type="homogeneous" id="fog"><rgb name="sigma_t" value="{sigma_t}"/><rgb name="albedo" value="{albedo}"/>{phase}</medium>
      <shape type="cube"><bsdf type="null"/><ref name="interior" id="fog"/></shape>
      <emitter type="constant"><rgb name="radiance" value="{radiance}"/></emitter>
    </scene>"""


@pytest.mark.parametrize("integrator", ["volpath", "prbvolpath"])
def test_beer_lambert_transmittance(mi, orc, integrator):
    """Purely absorbing medium (albedo 0) of thickness 2 in front of a radiance-1 environment: the pixel value is
    exp(-2 sigma_t) per channel, the analytic known answer for free-flight sampling + spectral weights
    (volpath.cpp:219-232, medium.cpp:92-104)."""
    sig = np.array([0.3, 0.8, 1.4])
    sc = mi.load_string(_slab_xml("0.3, 0.8, 1.4", "0, 0, 0", integrator, 4096))
    img = orc.OrcScene(sc).render().astype(np.float64)[..., :3]
    expect = np.exp(-2 * sig)
    assert np.allclose(img.mean((0, 1)), expect, rtol=0.02), (img.mean((0, 1)), expect)


def test_single_scattering_closed_form(mi, orc):
    """max_depth 2 in a thin-ish isotropic medium lit by a constant environment L: unscattered light exp(-tau) L plus one
    scattering event.  For a ray along the slab axis the single-scattered radiance is
    a * int_0^d sigma e^(-sigma s) * (1/4pi) int_{S^2} e^(-sigma * l(s, w)) dw ds * L, evaluated here by quadrature over the cube."""
    sigma, a, L = 0.6, 0.9, 1.0
    sc = mi.load_string(_slab_xml(f"{sigma}, {sigma}, {sigma}", f"{a}, {a}, {a}", "volpath", 8192, max_depth=2))
    img = orc.OrcScene(sc).render().astype(np.float64)[..., :3].mean()
    rng = np.random.default_rng(0)
    n = 400000
    s = rng.random(n) * 2.0                                   # depth of the scattering point on the axis (entry z = -1)
    z = rng.random(n) * 2 - 1; ph = rng.random(n) * 2 * np.pi
    w = np.stack([np.sqrt(1 - z * z) * np.cos(ph), np.sqrt(1 - z * z) * np.sin(ph), z], 1)
    p = np.stack([np.zeros(n), np.zeros(n), -1 + s], 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(w > 0, (1 - p) / w, (-1 - p) / w)        # exit distance from the cube along w
    l = np.nanmin(np.where(np.isfinite(t), t, np.inf), axis=1)
    single = a * np.mean(2.0 * sigma * np.exp(-sigma * s) * np.exp(-sigma * l)) * L
    expect = np.exp(-2 * sigma) * L + single
    assert img == pytest.approx(expect, rel=0.015), (img, expect)


# ---- PRB adjoint: gradients vs finite differences (src/integrators/tests/test_ad_integrators.py:1459-1500) ----
def prb_scene_xml(boundary, env, sample_emitters="true", g=0.4, rf="box", res=8):
    return f"""<scene version="3.0.0">
  <integrator type="prbvolpath"><integer name="max_depth" value="8"/></integrator>
  <medium type="homogeneous" id="fog"><rgb name="sigma_t" value="1.2, 0.7, 1.6"/><rgb name="albedo" value="0.8, 0.9, 0.6"/>
    <boolean name="sample_emitters" value="{sample_emitters}"/><phase type="hg"><float name="g" value="{g}"/></phase></medium>
  <sensor type="perspective"><float name="fov" value="35"/>
    <transform name="to_world"><lookat origin="3, 2.5, 4" target="0, 0, 0" up="0, 1, 0"/></transform>
    <sampler type="independent"><integer name="sample_count" value="4"/></sampler>
    <film type="hdrfilm"><integer name="width" value="{res}"/><integer name="height" value="{res}"/><rfilter type="{rf}"/></film>
  </sensor>
  <shape type="cube"><bsdf type="{boundary}"/><ref name="interior" id="fog"/></shape>
  <shape type="rectangle"><transform name="to_world"><scale value="6"/><rotate x="1" angle="-90"/><translate y="-1.001"/></transform><bsdf type="diffuse"/></shape>
  {env}
</scene>"""


PRB_ENV = '<emitter type="constant"><rgb name="radiance" value="1.0, 0.8, 0.6"/></emitter>'
PRB_AREA = ('<shape type="rectangle"><transform name="to_world"><scale value="0.9"/><rotate x="1" angle="90"/><translate y="3.0"/></transform>'
            '<emitter type="area"><rgb name="radiance" value="12, 11, 10"/></emitter></shape>')


@pytest.mark.parametrize("case", ["null+area", "dielectric+env"])
def test_prb_gradients_match_finite_differences(mi, orc, case):
    """No PRB reference values exist in the reference tree (parity unpinned by the reference): the oracle's
    adjoint is pinned by central finite differences of its own primal `prbvolpath` estimator with common
    random numbers.  Thresholds follow test_ad_integrators.py:102-153 in spirit (relative error of the
    gradient, noise-limited): albedo 3 %, sigma_t / g 25 % of the gradient's scale."""
    xml = prb_scene_xml("null", PRB_AREA) if case == "null+area" else prb_scene_xml("dielectric", PRB_ENV)
    sc = mi.load_string(xml); o = orc.OrcScene(sc)
    spp = 8192
    H, W, T = o.film_shape
    grad = np.full((H, W, T), 1.0 / (H * W * T), np.float32)
    g = o.render_backward(grad, spp=spp, seed=1)
    loss = lambda: float(o.render(spp=spp, seed=1, integrator="prbvolpath").astype(np.float64).mean())
    base = {"fog.sigma_t.value": np.array([1.2, 0.7, 1.6], np.float32), "fog.albedo.value": np.array([0.8, 0.9, 0.6], np.float32)}
    fd = {}
    for key, short in (("fog.sigma_t.value", "sigma_t"), ("fog.albedo.value", "albedo")):
        vals = []
        for c in range(3):
            eps = 0.02
            v = base[key].copy(); v[c] += eps; o.param_set(key, v); lp = loss()
            v = base[key].copy(); v[c] -= eps; o.param_set(key, v); lm = loss()
            o.param_set(key, base[key]); vals.append((lp - lm) / (2 * eps))
        fd[short] = np.array(vals)
    o.param_set("fog.phase_function.g", 0.42); lp = loss(); o.param_set("fog.phase_function.g", 0.38); lm = loss()
    o.param_set("fog.phase_function.g", 0.4)
    fd_g = (lp - lm) / 0.04
    assert np.abs(g["albedo"] - fd["albedo"]).max() <= 0.03 * np.abs(fd["albedo"]).max()
    assert np.abs(g["sigma_t"] - fd["sigma_t"]).max() <= 0.25 * np.abs(fd["sigma_t"]).max()
    # d/dg is small next to the FD noise in the refractive case: checked tightly only where it is well resolved
    if case == "null+area":
        assert abs(g["g"] - fd_g) <= 0.15 * abs(fd_g)
    else:
        assert g["g"] * fd_g > 0 and abs(g["g"] - fd_g) <= 0.01
    assert (g["albedo"] > 0).all() and (g["sigma_t"] < 0).all()


def test_ldsampler_render_agrees_with_independent(mi, orc):
    """Integrator-level sanity of the low-discrepancy sampler (no render fixture exists in the reference tree for it):
    the ld estimate agrees with the independent one in expectation, for path, volpath and the PRB primal."""
    xi = prb_scene_xml("null", PRB_AREA, res=8)
    xl = xi.replace('<sampler type="independent">', '<sampler type="ldsampler">')
    si, sl = mi.load_string(xi), mi.load_string(xl)
    oi, ol = orc.OrcScene(si), orc.OrcScene(sl)
    assert sl.desc.sampler_type == 1 and sl.spp == 4
    for integ in ("path", "volpath", "prbvolpath"):
        a = oi.render(spp=4096, seed=5, integrator=integ).astype(np.float64)[..., :3]
        b = ol.render(spp=4096, seed=5, integrator=integ).astype(np.float64)[..., :3]
        assert abs(a.mean() - b.mean()) <= 0.02 * a.mean(), (integ, a.mean(), b.mean())
        assert np.abs(a - b).mean() <= 0.08 * a.mean()


def test_prb_primal_matches_volpath_in_expectation(mi, orc):
    """prbvolpath's primal estimator (RR clamp .99, analytic NEE transmittance) and volpath estimate the same image."""
    sc = mi.load_string(prb_scene_xml("null", PRB_AREA)); o = orc.OrcScene(sc)
    a = o.render(spp=4096, seed=3, integrator="prbvolpath").astype(np.float64)
    b = o.render(spp=4096, seed=4, integrator="volpath").astype(np.float64)
    assert abs(a.mean() - b.mean()) <= 0.02 * b.mean()


# ---- heterogeneous media + null collisions: delta tracking against the homogeneous closed forms ----
def _het_slab_xml(tmp_path, mi, grid, scale, albedo, spp, max_depth=-1, integrator="volpath", spectral="true"):
    vol = os.path.join(str(tmp_path), "density.vol")
    mi.write_volume_grid(vol, grid)
    # the grid's unit cube is mapped onto the cube [-1, 1]^3 that bounds the medium
    return f"""<scene version="3.0.0">
      <integrator type="{integrator}"><integer name="max_depth" value="{max_depth}"/></integrator>
      <sensor type="perspective"><float name="fov" value="2"/>
        <transform name="to_world"><lookat origin="0, 0, -20" target="0, 0, 0" up="0, 1, 0"/></transform>
        <sampler type="independent"><integer name="sample_count" value="{spp}"/></sampler>
        <film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/><rfilter type="box"/></film>
      </sensor>
      <medium type="heterogeneous" id="smoke">
        <volume name="sigma_t" type="gridvolume"><string name="filename" value="{vol}"/>
          <transform name="to_world"><scale value="2"/><translate x="-1" y="-1" z="-1"/></transform></volume>
        <rgb name="albedo" value="{albedo}"/><float name="scale" value="{scale}"/><boolean name="has_spectral_extinction" value="{spectral}"/>
      </medium>
      <shape type="cube"><bsdf type="null"/><ref name="interior" id="smoke"/></shape>
      <emitter type="constant"><rgb name="radiance" value="1, 1, 1"/></emitter>
    </scene>"""


@pytest.mark.parametrize("spectral", ["true", "false"])
def test_heterogeneous_beer_lambert(mi, orc, tmp_path, spectral):
    """Absorbing heterogeneous slab: density varies along the viewing axis (z), so the pixel value is exp(-scale * integral of the
    trilinear grid along the ray); delta tracking with majorant scale * max must reproduce it (volpath.cpp:238-259)."""
    rz = 16
    prof = 0.2 + 0.8 * np.sin(np.linspace(0.2, 2.9, rz)) ** 2
    grid = np.broadcast_to(prof[:, None, None], (rz, 4, 4)).astype(np.float32)
    scale = 1.3
    sc = mi.load_string(_het_slab_xml(tmp_path, mi, grid, scale, "0, 0, 0", 8192, spectral=spectral))
    m = sc.desc.media[0]
    assert m.type == 4 and list(m.grid_res) == [4, 4, rz] and m.grid_max == pytest.approx(prof.max())
    assert list(m.grid_bbox_min) == pytest.approx([-1, -1, -1]) and list(m.grid_bbox_max) == pytest.approx([1, 1, 1])
    img = orc.OrcScene(sc).render().astype(np.float64)[..., :3]
    # trilinear profile along z through the cube: texel centres at (k + .5) / rz of the unit cube, clamped outside
    z = (np.linspace(-1, 1, 20001)[:-1] + 1e-4 + 1) / 2
    f = z * rz - 0.5; k = np.clip(np.floor(f).astype(int), -1, rz - 1); w = f - np.floor(f)
    dens = prof[np.clip(k, 0, rz - 1)] * (1 - w) + prof[np.clip(k + 1, 0, rz - 1)] * w
    tau = scale * dens.mean() * 2.0
    assert img.mean() == pytest.approx(np.exp(-tau), rel=0.03), (img.mean(), np.exp(-tau))


def test_heterogeneous_constant_grid_matches_homogeneous(mi, orc, tmp_path):
    """A constant grid is a homogeneous medium: delta tracking with null collisions (majorant 2.5 x the density, so most collisions
    are null) must give the homogeneous medium's image in expectation, single scattering included."""
    sigma, a = 0.6, 0.9
    grid2 = np.full((6, 6, 6), 0.4, np.float32)
    sc_het = mi.load_string(_het_slab_xml(tmp_path, mi, grid2, sigma / 0.4, f"{a}, {a}, {a}", 8192, max_depth=2))
    sc_hom = mi.load_string(_slab_xml(f"{sigma}, {sigma}, {sigma}", f"{a}, {a}, {a}", "volpath", 8192, max_depth=2))
    het = orc.OrcScene(sc_het).render().astype(np.float64)[..., :3].mean()
    hom = orc.OrcScene(sc_hom).render().astype(np.float64)[..., :3].mean()
    assert het == pytest.approx(hom, rel=0.015), (het, hom)
    # with a loose majorant (a single larger texel OUTSIDE the ray's path) the answer must not move
    g3 = grid2.copy(); g3[0, 0, 0] = 1.0
    sc3 = mi.load_string(_het_slab_xml(tmp_path, mi, g3, sigma / 0.4, f"{a}, {a}, {a}", 8192, max_depth=2))
    o3 = orc.OrcScene(sc3); het3 = o3.render().astype(np.float64)[..., :3].mean()
    assert het3 == pytest.approx(hom, rel=0.02), (het3, hom)
    o1 = orc.OrcScene(sc_het); o1.render()
    assert o3.last_stats["n_iter"] > 1.3 * o1.last_stats["n_iter"]                   # the extra trips are null collisions


def test_heterogeneous_furnace(mi, orc, tmp_path):
    """Albedo 1, radiance-1 environment, varying density: every pixel converges to 1 (energy conservation of the null-collision
    weights in the spectral branch)."""
    rng = np.random.default_rng(1)
    grid = (0.1 + rng.random((8, 8, 8))).astype(np.float32)
    xml = _het_slab_xml(tmp_path, mi, grid, 2.0, "1, 1, 1", 512).replace('<float name="fov" value="2"/>', '<float name="fov" value="8"/>')
    img = orc.OrcScene(mi.load_string(xml)).render().astype(np.float64)[..., :3]
    assert img.mean() == pytest.approx(1.0, abs=0.02)


# ---- volpathmis: same expectations as volpath, on the scenes whose answers are known ----
@pytest.mark.parametrize("smis", ["true", "false"])
def test_volpathmis_known_answers(mi, orc, smis):
    """volpathmis (volpathmis.cpp:127-699) estimates the same integral as volpath: Beer-Lambert through a spectrally varying
    absorber, the single-scattering closed form of test_single_scattering_closed_form, the albedo-1 furnace."""
    def xml(sigma_t, albedo, spp, max_depth=-1):
        return _slab_xml(sigma_t, albedo, "volpathmis", spp, max_depth).replace('<integer name="max_depth"', f'<boolean name="use_spectral_mis" value="{smis}"/><integer name="max_depth"')
    sig = np.array([0.3, 0.8, 1.4])
    sc = mi.load_string(xml("0.3, 0.8, 1.4", "0, 0, 0", 4096))
    assert sc.desc.integrator.type == 5 and sc.desc.use_spectral_mis == (1 if smis == "true" else 0)
    img = orc.OrcScene(sc).render().astype(np.float64)[..., :3]
    assert np.allclose(img.mean((0, 1)), np.exp(-2 * sig), rtol=0.02), (img.mean((0, 1)), np.exp(-2 * sig))
    sigma, a = 0.6, 0.9
    mis = orc.OrcScene(mi.load_string(xml(f"{sigma}, {sigma}, {sigma}", f"{a}, {a}, {a}", 8192, max_depth=2))).render().astype(np.float64)[..., :3].mean()
    ref = orc.OrcScene(mi.load_string(_slab_xml(f"{sigma}, {sigma}, {sigma}", f"{a}, {a}, {a}", "volpath", 8192, max_depth=2))).render().astype(np.float64)[..., :3].mean()
    assert mis == pytest.approx(ref, rel=0.015), (mis, ref)
    furn = orc.OrcScene(mi.load_string(xml("1.5, 0.7, 2.0", "1, 1, 1", 512).replace('<float name="fov" value="2"/>', '<float name="fov" value="8"/>'))).render().astype(np.float64)[..., :3]
    assert furn.mean() == pytest.approx(1.0, abs=0.02)


def test_volpathmis_matches_volpath_on_a_lit_scene(mi, orc, tmp_path):
    """Surfaces, area light, environment, a heterogeneous medium with null collisions and a coloured homogeneous one: image means
    of volpathmis (both settings) and volpath agree within Monte-Carlo noise."""
    from test_round2_gpu import het_xml
    rng = np.random.default_rng(3)
    grid = (0.05 + rng.random((12, 10, 8)) ** 3).astype(np.float32)
    vol = os.path.join(str(tmp_path), "smoke.vol"); mi.write_volume_grid(vol, grid)
    hom = '<medium type="homogeneous" id="fog"><rgb name="sigma_t" value="0.9, 0.3, 1.6"/><rgb name="albedo" value="0.8, 0.8, 0.9"/></medium>'
    base = het_xml(vol, extra_medium=hom, md=8).replace('<shape type="rectangle"><transform name="to_world"><scale value="6"/>',
        '<shape type="cube"><transform name="to_world"><scale value="0.5"/><translate x="2" y="-0.4"/></transform><bsdf type="null"/><ref name="interior" id="fog"/></shape>'
        '<shape type="rectangle"><transform name="to_world"><scale value="6"/>').replace('<integer name="sample_count" value="16"/>', '<integer name="sample_count" value="256"/>')
    means = {}
    for name, integ in (("volpath", '<integrator type="volpath">'), ("mis", '<integrator type="volpathmis">'),
                        ("mis-nospectral", '<integrator type="volpathmis"><boolean name="use_spectral_mis" value="false"/>')):
        img = orc.OrcScene(mi.load_string(base.replace('<integrator type="volpath">', integ))).render().astype(np.float64)[..., :3]
        means[name] = img.mean((0, 1))
    assert np.allclose(means["mis"], means["volpath"], rtol=0.03), means
    assert np.allclose(means["mis-nospectral"], means["volpath"], rtol=0.03), means


def test_oracle_native_flags_build_gives_the_same_lanes(tmp_path, mi, orc, liver_small):
    """bench.py times the oracle's scalar renderer built with -O3 -march=native on the host it runs on: with -ffp-contract=off and no
    fast-math that build computes the same binary32 values as the shipped -O2 -mfma one (lanes and scalar-renderer film bit for bit)."""
    import subprocess
    from conftest import ROOT
    native = os.path.join(str(tmp_path), "liborc_native.so")
    src = [os.path.join(ROOT, "oracle", f) for f in ("orc_scene.cpp", "orc_render.cpp", "orc_api.cpp", "orc_vae.cpp")]
    subprocess.run(["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-shared", "-o", native] + src, check=True)
    n = 128 * 72 * 4
    a = orc.OrcScene(liver_small).render_samples(0, n, seed=2)
    ra = orc.OrcScene(liver_small).render(scalar=True, spp=2, seed=1, threads=4)
    shipped = orc.ORC_LIB
    try:
        orc.use_library(native)
        b = orc.OrcScene(liver_small).render_samples(0, n, seed=2)
        rb = orc.OrcScene(liver_small).render(scalar=True, spp=2, seed=1, threads=4)
    finally:
        orc.use_library(shipped)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.allclose(ra, rb, rtol=1e-5, atol=1e-7)          # (the block renderer's film sums in thread-completion order)


def test_prb_null_collision_gradients_match_finite_differences(mi, orc, tmp_path):
    """prbvolpath.py:178-196,404-415 (VERDICT r2 missing 1): the adjoint through a heterogeneous medium - delta tracking with null
    collisions on the path, ratio tracking on the emitter-sampling march.  No PRB fixture exists in the reference tree (parity unpinned
    by the reference): pinned by central finite differences of the oracle's own primal estimator with common random numbers.  d/d(scale)
    is the sum of the three d_sigma_t (include/liverrt.h); it is taken at scale = 8, where it is well above the FD noise."""
    import scene_gen
    vol = os.path.join(str(tmp_path), "smoke.vol"); mi.write_volume_grid(vol, scene_gen.smoke_grid())
    xml = scene_gen.resized(scene_gen.het_xml(vol, md=8), 12, 9, 16).replace('type="volpath"', 'type="prbvolpath"')
    sc = mi.load_string(xml); o = orc.OrcScene(sc)
    spp = 8192
    H, W, T = o.film_shape
    grad = np.full((H, W, T), 1.0 / (H * W * T), np.float32)
    loss = lambda: float(o.render(spp=spp, seed=1, integrator="prbvolpath").astype(np.float64).mean())
    g = o.render_backward(grad, spp=spp, seed=1)
    base = np.array([0.9, 0.8, 0.6], np.float32); fd = []
    for c in range(3):
        v = base.copy(); v[c] += 0.02; o.param_set("smoke.albedo.value", v); lp = loss()
        v = base.copy(); v[c] -= 0.02; o.param_set("smoke.albedo.value", v); lm = loss()
        o.param_set("smoke.albedo.value", base); fd.append((lp - lm) / 0.04)
    assert np.abs(g["albedo"] - np.array(fd)).max() <= 0.03 * np.abs(fd).max(), (g["albedo"], fd)
    o.param_set("smoke.phase_function.g", 0.32); lp = loss(); o.param_set("smoke.phase_function.g", 0.28); lm = loss(); o.param_set("smoke.phase_function.g", 0.3)
    assert abs(g["g"] - (lp - lm) / 0.04) <= 0.05 * abs((lp - lm) / 0.04)
    o.param_set("smoke.scale", 8.0)
    g8 = o.render_backward(grad, spp=spp, seed=1)
    o.param_set("smoke.scale", 8.4); lp = loss(); o.param_set("smoke.scale", 7.6); lm = loss()
    fd_scale = (lp - lm) / 0.8
    assert fd_scale < 0 and abs(g8["sigma_t"].sum() - fd_scale) <= 0.15 * abs(fd_scale), (g8["sigma_t"], fd_scale)
