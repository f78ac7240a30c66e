"""Network stage of the learned subsurface model (docs/SUBSURFACE_NOTES.md): the scatter network of
include/mitsuba/render/scattereigen.h:249-480 with the reference's own weights (scenes/assets/vae3d, data files).

PARITY UNPINNED BY THE REFERENCE: its tree holds no input / output vector of this network and nothing that can run it.  What is
checked: (CPU) the oracle's float32 restatement against an independent float64 numpy evaluation of the same layers, the sampler
protocol (one draw for the absorption test, four for the latents), the file format; (GPU) the HIP kernel against the oracle, bit
for bit."""
import os

import numpy as np
import pytest

from conftest import ROOT

MODEL = os.path.join(ROOT, "scenes", "assets", "vae3d", "0487_FinalSharedLs7Mixed3_AbsSharedSimComplexMixed3")
STATS = os.path.join(ROOT, "scenes", "assets", "vae3d", "data_stats.json")
MEDIUM = dict(albedo=(0.99975, 0.999554, 0.9966), g=0.0, ior=1.3, sigma_t=(0.20, 0.30, 0.42))     # scenes/SphereLiverPoint/sss/scene.xml:28-31


def inputs(n, seed=1):
    r = np.random.default_rng(seed)
    pos = r.uniform(-2, 2, (n, 3)).astype(np.float32)
    d = r.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    poly = (r.normal(size=(n, 20)) * 0.3).astype(np.float32)
    return pos, d.astype(np.float32), poly


def pcg32_floats(seed, lane, k):
    """the lane's PCG32 stream as the kernels seed it (TEA(seed, lane) -> PCG32.seed), first k floats"""
    v0, v1, s = np.uint32(seed), np.uint32(lane), np.uint32(0)
    with np.errstate(over="ignore"):
        for _ in range(4):
            s = np.uint32(s + np.uint32(0x9e3779b9))
            v0 = np.uint32(v0 + (np.uint32((v1 << np.uint32(4)) + np.uint32(0xa341316c)) ^ np.uint32(v1 + s) ^ np.uint32((v1 >> np.uint32(5)) + np.uint32(0xc8013ea4))))
            v1 = np.uint32(v1 + (np.uint32((v0 << np.uint32(4)) + np.uint32(0xad90777d)) ^ np.uint32(v0 + s) ^ np.uint32((v0 >> np.uint32(5)) + np.uint32(0x7e95761e))))
    M, mask = 0x5851f42d4c957f2d, (1 << 64) - 1
    inc = ((int(v1) << 1) | 1) & mask
    state = 0
    def step():
        nonlocal state
        old = state; state = (old * M + inc) & mask
        xs = (((old >> 18) ^ old) >> 27) & 0xffffffff; rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xffffffff
    step(); state = (state + int(v0)) & mask; step()
    out = []
    for _ in range(k):
        u = np.array([(step() >> 9) | 0x3f800000], np.uint32).view(np.float32)[0]
        out.append(float(u) - 1.0)
    return out


def reference_f64(blob, pos, d, poly, albedo, g, ior, sigma_t, fit_scale, seed):
    """float64 numpy evaluation of the same network, layer by layer (independent of oracle/orc_vae.cpp)"""
    from liverrenderer_amd import vae
    b = blob.astype(np.float64); off = 44; W = {}
    for stem, rows, cols in vae._LAYOUT:
        n = rows * max(cols, 1); W[stem] = b[off:off + n].reshape((rows, cols) if cols else (rows,)); off += n
    albedo, sigma_t = np.asarray(albedo, np.float64), np.asarray(sigma_t, np.float64)
    ss = albedo * sigma_t; sa = sigma_t - ss; ap = (1 - g) * ss / ((1 - g) * ss + sa)
    ea = -np.log(1 - ap * (1 - np.exp(-8.0))) / 8.0
    M = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
    eff = (M @ ea).mean()
    relu = lambda v: np.maximum(v, 0)
    out, absorbed = np.zeros((len(pos), 3)), np.zeros(len(pos))
    for i in range(len(pos)):
        x = np.concatenate([(poly[i].astype(np.float64) - b[4:24]) * b[24:44], [(eff - b[0]) * b[1], (g - b[2]) * b[3], 2 * (ior - 1.25)]])
        f = x
        for k in range(3):
            f = relu(W[f"shared_preproc_mlp_2_shapemlp_fcn_{k}_weights"] @ f + W[f"shared_preproc_mlp_2_shapemlp_fcn_{k}_biases"])
        at = relu(W["absorption_mlp_fcn_0_weights"] @ f + W["absorption_mlp_fcn_0_biases"])
        a = 1 / (1 + np.exp(-(W["absorption_dense_kernel"][0] @ at + W["absorption_dense_bias"][0])))
        u = pcg32_floats(seed, i, 5)
        if not (u[0] > a):
            out[i] = pos[i]; absorbed[i] = 1; continue
        lat = []
        for h in range(2):
            r, phi = np.sqrt(-2 * np.log(1 - u[1 + 2 * h])), 2 * np.pi * u[2 + 2 * h]
            lat += [np.cos(phi) * r, np.sin(phi) * r]
        y = np.concatenate([lat, f])
        for k in range(3):
            y = relu(W[f"scatter_decoder_fcn_fcn_{k}_weights"] @ y + W[f"scatter_decoder_fcn_fcn_{k}_biases"])
        o = W["scatter_dense_2_kernel"] @ y + W["scatter_dense_2_bias"]
        n = -d[i].astype(np.float64); sign = np.copysign(1.0, n[2]); aa = -1 / (sign + n[2]); bb = n[0] * n[1] * aa
        t1 = np.array([1 + sign * n[0] * n[0] * aa, sign * bb, -sign * n[0]]); t2 = np.array([bb, sign + n[1] * n[1] * aa, -n[1]])
        w = pos[i] + o[0] * t1 + o[1] * t2 + o[2] * n
        out[i] = pos[i] + (w - pos[i]) / fit_scale
    return out, absorbed, a


def test_weight_files_and_blob(mi):
    from liverrenderer_amd import vae
    blob = vae.pack_blob(MODEL, STATS)
    assert blob.size == vae.N_FLOATS == 24944 and np.isfinite(blob).all()
    a = vae.read_bin(os.path.join(MODEL, "variables", "scatter_decoder_fcn_fcn_0_weights.bin"))
    assert a.shape == (64, 68) and (blob[44 + 64 * 23 + 64:][:1] == vae.read_bin(os.path.join(MODEL, "variables", "shared_preproc_mlp_2_shapemlp_fcn_1_weights.bin"))[0, :1]).all()
    from liverrenderer_amd import _lib
    for name in ("lrt_vae_model_create", "lrt_vae_model_free", "lrt_vae_scatter"):
        assert hasattr(_lib.lib(), name)


def test_oracle_against_float64_evaluation(mi, orc):
    from liverrenderer_amd import vae
    blob = vae.pack_blob(MODEL, STATS)
    pos, d, poly = inputs(400)
    for fit_scale, seed in ((1.0, 0), (3.7, 11)):
        out, ab = orc.vae_scatter(blob, pos, d, poly, MEDIUM["albedo"], MEDIUM["g"], MEDIUM["ior"], MEDIUM["sigma_t"], fit_scale, seed)
        ref, rab, _ = reference_f64(blob, pos, d, poly, MEDIUM["albedo"], MEDIUM["g"], MEDIUM["ior"], MEDIUM["sigma_t"], fit_scale, seed)
        agree = ab == rab                                     # a draw within float rounding of the absorption probability may flip
        assert agree.mean() > 0.99 and 0 < ab.sum() < len(ab)            # (this medium's albedo is ~1: few samples are absorbed)
        assert np.abs(out[agree] - ref[agree]).max() < 2e-3 * max(1.0, np.abs(ref).max())
        assert (out[ab == 1] == pos[ab == 1]).all()
    # the exit points move with the latents, not with the absorption draw alone: two seeds, different points
    o1, a1 = orc.vae_scatter(blob, pos, d, poly, MEDIUM["albedo"], MEDIUM["g"], MEDIUM["ior"], MEDIUM["sigma_t"], 1.0, 1)
    o2, a2 = orc.vae_scatter(blob, pos, d, poly, MEDIUM["albedo"], MEDIUM["g"], MEDIUM["ior"], MEDIUM["sigma_t"], 1.0, 2)
    both = (a1 == 0) & (a2 == 0)
    assert both.sum() > 50 and (np.abs(o1[both] - o2[both]).max(axis=1) > 1e-4).mean() > 0.9


@pytest.mark.gpu
def test_device_against_oracle_bit_exact(mi, orc):
    from liverrenderer_amd import vae
    model = vae.load_scatter_model(MODEL, STATS)
    for n, fit_scale, seed, med in ((1, 1.0, 0, MEDIUM), (1000, 2.5, 3, MEDIUM), (4099, 0.8, 7, dict(albedo=(0.8, 0.5, 0.3), g=0.4, ior=1.45, sigma_t=(1.0, 2.0, 4.0)))):
        pos, d, poly = inputs(n, seed + 5)
        out, ab = model.scatter(pos, d, poly, med["albedo"], med["g"], med["ior"], med["sigma_t"], fit_scale, seed)
        o, a = orc.vae_scatter(model.blob, pos, d, poly, med["albedo"], med["g"], med["ior"], med["sigma_t"], fit_scale, seed)
        assert (ab == a).all() and (out.view(np.uint32) == o.view(np.uint32)).all()
    with pytest.raises(RuntimeError, match="fit_scale"):
        model.scatter(pos, d, poly, MEDIUM["albedo"], 0.0, 1.3, MEDIUM["sigma_t"], 0.0)
