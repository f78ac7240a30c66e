// TEST INFRASTRUCTURE (CPU oracle).  The network stage of the learned subsurface model, restated from
// /root/reference/include/mitsuba/render/scattereigen.h:316-470 (ScatterModelSimShared<3, 4, 64, 64>::run) with
// :142-175 (preprocessFeatures<3, true>), :140-148 (localToWorld), :21-27 (onb),
// include/mitsuba/render/sss_particle_tracer.h:365-380 (effectiveAlbedo, with its xyz_to_srgb), src/render/vaehelper.cpp:14-27
// (sampleGaussianVector) and include/mitsuba/core/warp.h square_to_std_normal.  Weight blob layout: include/liverrt.h.
//
// PARITY UNPINNED: the reference holds no input / output vector for this network, its own build cannot run, and the host
// compiler's floating-point contraction in `sum += w * x` is not recorded.  Decisions taken here (the device kernel takes the
// same): loops written as loops in the source accumulate with an unfused multiply and add in index order; the two layers the
// source writes as `Matrix * Array` use Dr.Jit's column-wise fmadd chain (first column a plain product); dr::dot is an fmadd
// chain in index order; exp / log / sincos are the oracle's own kernels (orc_math.h).  tests/test_vae.py checks this file
// against an independent float64 numpy evaluation of the same network.
#include <cstdint>
#include <cmath>
#include <cstring>
#include "../include/liverrt.h"
#include "orc_math.h"

namespace {
using orc::m_exp; using orc::m_log; using orc::m_sincos;
struct PCG32 {
    uint64_t state, inc;
    void seed(uint64_t initstate, uint64_t initseq) { state = 0; inc = (initseq << 1) | 1u; next_u32(); state += initstate; next_u32(); }
    uint32_t next_u32() {
        uint64_t old = state; state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t) (((old >> 18) ^ old) >> 27), rot = (uint32_t) (old >> 59);
        return (xs >> rot) | (xs << ((0u - rot) & 31u));
    }
    float next() { uint32_t u = (next_u32() >> 9) | 0x3f800000u; float f; memcpy(&f, &u, 4); return f - 1.f; }
};
void tea32(uint32_t v0, uint32_t v1, uint32_t *o0, uint32_t *o1) {
    uint32_t sum = 0;
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    *o0 = v0; *o1 = v1;
}
// layer written as a loop in the source: sum = 0; sum += w[i][j] * x[j]; y = max(sum + b, 0)
void loop_layer(const float *W, const float *b, int rows, int cols, const float *x, float *y, bool relu) {
    for (int i = 0; i < rows; ++i) {
        float sum = 0.f;
        for (int j = 0; j < cols; ++j) sum = sum + W[i * cols + j] * x[j];
        float v = sum + b[i];
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}
// layer written as Matrix * Array: columns accumulated with fmadd, first column a plain product; then + b, max(., 0)
void matrix_layer(const float *W, const float *b, int n, const float *x, float *y) {
    for (int i = 0; i < n; ++i) {
        float sum = W[i * n] * x[0];
        for (int j = 1; j < n; ++j) sum = fmaf(W[i * n + j], x[j], sum);
        y[i] = fmaxf(sum + b[i], 0.f);
    }
}
float effective_albedo(float a) { return -m_log(1.0f - a * (1.0f - m_exp(-8.0f))) / 8.0f; }        // sss_particle_tracer.h:365
void std_normal(float ux, float uy, float *a, float *b) {                                           // warp.h square_to_std_normal
    float r = sqrtf(-2.f * m_log(1.f - ux)), phi = 2.f * 3.14159265358979323846f * uy, s, c;
    m_sincos(phi, &s, &c); *a = c * r; *b = s * r;
}
}

extern "C" void orc_vae_scatter(const float *blob, uint32_t n, const float *in_pos, const float *in_dir, const float *poly,
                                const float *albedo, float g, float ior, const float *sigma_t, float fit_scale, uint32_t seed,
                                float *out_pos, float *out_absorption) {
    const float *S = blob + LRT_VAE_STATS;
    // preprocessFeatures<3, true>: similarity theory, effective albedo (with the source's xyz_to_srgb), mean of the three
    float ea[3];
    for (int k = 0; k < 3; ++k) {
        float sigma_s = albedo[k] * sigma_t[k], sigma_a = sigma_t[k] - sigma_s;
        float albedo_p = (1 - g) * sigma_s / ((1 - g) * sigma_s + sigma_a);
        ea[k] = effective_albedo(albedo_p);
    }
    const float M[9] = { 3.240479f, -1.537150f, -0.498535f, -0.969256f, 1.875991f, 0.041556f, 0.055648f, -0.204043f, 1.057311f };
    float srgb[3];
    for (int r = 0; r < 3; ++r) srgb[r] = fmaf(M[3 * r + 2], ea[2], fmaf(M[3 * r + 1], ea[1], M[3 * r] * ea[0]));   // Matrix * Color: column-wise fmadd
    const float eff = (srgb[0] + srgb[1] + srgb[2]) * (1.f / 3.f);                                                   // dr::mean
    const float albedo_norm = (eff - S[0]) * S[1], g_norm = (g - S[2]) * S[3], ior_norm = 2.0f * (ior - 1.25f);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t v0, v1; tea32(seed, i, &v0, &v1);
        PCG32 rng; rng.seed(v0, v1);
        float x[23];
        for (int k = 0; k < 20; ++k) x[k] = (poly[20 * (size_t) i + k] - S[4 + k]) * S[24 + k];
        x[20] = albedo_norm; x[21] = g_norm; x[22] = ior_norm;
        float f0[64], f1[64], f2[64];
        loop_layer(blob + LRT_VAE_PRE0_W, blob + LRT_VAE_PRE0_W + 64 * 23, 64, 23, x, f0, true);
        matrix_layer(blob + LRT_VAE_PRE1_W, blob + LRT_VAE_PRE1_W + 64 * 64, 64, f0, f1);
        matrix_layer(blob + LRT_VAE_PRE2_W, blob + LRT_VAE_PRE2_W + 64 * 64, 64, f1, f2);
        float at[32];
        loop_layer(blob + LRT_VAE_ABS0_W, blob + LRT_VAE_ABS0_W + 32 * 64, 32, 64, f2, at, true);
        const float *K = blob + LRT_VAE_ABSD_K;
        float a = K[0] * at[0];
        for (int k = 1; k < 32; ++k) a = fmaf(K[k], at[k], a);
        a = a + K[32];
        const float absorption = 1.0f / (1.0f + m_exp(-a));
        const float *ip = in_pos + 3 * (size_t) i, *id = in_dir + 3 * (size_t) i;
        if (!(rng.next() > absorption)) {                       // all is absorbed
            out_pos[3 * (size_t) i] = ip[0]; out_pos[3 * (size_t) i + 1] = ip[1]; out_pos[3 * (size_t) i + 2] = ip[2];
            out_absorption[i] = 1.0f; continue;
        }
        float fl[68];
        { float ux = rng.next(), uy = rng.next(); std_normal(ux, uy, &fl[0], &fl[1]); }
        { float ux = rng.next(), uy = rng.next(); std_normal(ux, uy, &fl[2], &fl[3]); }
        for (int k = 0; k < 64; ++k) fl[4 + k] = f2[k];
        float y0[64], y1[64], y2[64], o[3];
        loop_layer(blob + LRT_VAE_DEC0_W, blob + LRT_VAE_DEC0_W + 64 * 68, 64, 68, fl, y0, true);
        loop_layer(blob + LRT_VAE_DEC1_W, blob + LRT_VAE_DEC1_W + 64 * 64, 64, 64, y0, y1, true);
        loop_layer(blob + LRT_VAE_DEC2_W, blob + LRT_VAE_DEC2_W + 64 * 64, 64, 64, y1, y2, true);
        loop_layer(blob + LRT_VAE_OUT_K, blob + LRT_VAE_OUT_K + 3 * 64, 3, 64, y2, o, false);
        // localToWorld(inPos, -inDir, outPos, true): onb(n), inPos + o.x t1 + o.y t2 + o.z n
        const float nx = -id[0], ny = -id[1], nz = -id[2];
        const float sign = copysignf(1.0f, nz), aa = -1.0f / (sign + nz), bb = nx * ny * aa;
        const float t1[3] = { 1.0f + sign * nx * nx * aa, sign * bb, -sign * nx }, t2[3] = { bb, sign + ny * ny * aa, -ny }, nn[3] = { nx, ny, nz };
        for (int k = 0; k < 3; ++k) {
            float w = ((ip[k] + o[0] * t1[k]) + o[1] * t2[k]) + o[2] * nn[k];
            out_pos[3 * (size_t) i + k] = ip[k] + (w - ip[k]) / fit_scale;
        }
        out_absorption[i] = 0.0f;
    }
}
