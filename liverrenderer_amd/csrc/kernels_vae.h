// Network stage of the learned subsurface model (include/liverrt.h, docs/SUBSURFACE_NOTES.md):
// ScatterModelSimShared<3, 4, 64, 64>::run of include/mitsuba/render/scattereigen.h:316-470, one lane per sample.
// The weights are wave-uniform: they are read through the constant address space (scalar loads broadcast into the
// multiplies), the per-lane activations live in LDS columns ([feature][thread]).  This is VALU work by design: every
// accumulation keeps the order the source prescribes (loops: unfused multiply-add in index order; `Matrix * Array`
// layers: Dr.Jit's column-wise fmadd chain), so results are bit-identical to oracle/orc_vae.cpp; an MFMA formulation
// would change the order of the sums (and fp32 MFMA has no higher peak than packed fp32 VALU on gfx950).
#pragma once

namespace lrt {

#define LRT_VAE_BLOCK 128

typedef const LRT_CONST float *CW;

// y[i] = [max](sum_j W[i][j] x[j] + b[i], 0): the source's explicit loops (unfused, index order)
template <int ROWS, int COLS, bool RELU>
DEV void vae_loop_layer(CW W, CW b, const float *x /* LDS column */, float *y /* LDS column */) {
    for (int i = 0; i < ROWS; ++i) {
        float sum = 0.f;
        for (int j = 0; j < COLS; ++j) sum = sum + W[i * COLS + j] * x[j * LRT_VAE_BLOCK];
        const float v = sum + b[i];
        y[i * LRT_VAE_BLOCK] = RELU ? fmax_(v, 0.f) : v;
    }
}
// `Matrix * Array`: first column a product, the others fmadd; + b, max(., 0)
DEV void vae_matrix_layer(CW W, CW b, const float *x, float *y) {
    for (int i = 0; i < 64; ++i) {
        float sum = W[i * 64] * x[0];
        for (int j = 1; j < 64; ++j) sum = fma_(W[i * 64 + j], x[j * LRT_VAE_BLOCK], sum);
        y[i * LRT_VAE_BLOCK] = fmax_(sum + b[i], 0.f);
    }
}

struct DVaeArgs {
    const float *blob;                       // LRT_VAE_N_FLOATS floats (device)
    const float *in_pos, *in_dir, *poly;     // 3n, 3n, 20n
    float *out_pos, *out_absorption;         // 3n, n
    float albedo_norm, g_norm, ior_norm, fit_scale;
    uint32_t n, seed;
};

// preprocessFeatures<3, true> (scattereigen.h:142-175): similarity theory, effective albedo per channel
// (sss_particle_tracer.h:365-380, with the source's xyz_to_srgb), mean; out = (albedo, g, ior) features
__global__ void k_vae_medium_features(const float *blob, float a0, float a1, float a2, float g, float ior, float s0, float s1, float s2, float *out) {
    const float albedo[3] = { a0, a1, a2 }, sigma_t[3] = { s0, s1, s2 };
    float ea[3];
    for (int k = 0; k < 3; ++k) {
        const float sigma_s = albedo[k] * sigma_t[k], sigma_a = sigma_t[k] - sigma_s;
        const float albedo_p = (1 - g) * sigma_s / ((1 - g) * sigma_s + sigma_a);
        ea[k] = -m_log(1.0f - albedo_p * (1.0f - m_exp(-8.0f))) / 8.0f;
    }
    const float M[9] = { 3.240479f, -1.537150f, -0.498535f, -0.969256f, 1.875991f, 0.041556f, 0.055648f, -0.204043f, 1.057311f };
    float srgb[3];
    for (int r = 0; r < 3; ++r) srgb[r] = fma_(M[3 * r + 2], ea[2], fma_(M[3 * r + 1], ea[1], M[3 * r] * ea[0]));
    const float eff = (srgb[0] + srgb[1] + srgb[2]) * (1.f / 3.f);
    const float *S = blob + LRT_VAE_STATS;
    out[0] = (eff - S[0]) * S[1]; out[1] = (g - S[2]) * S[3]; out[2] = 2.0f * (ior - 1.25f); out[3] = eff;
}

__global__ void __launch_bounds__(LRT_VAE_BLOCK) k_vae_scatter(DVaeArgs A) {
    extern __shared__ float sm[];                                       // two banks of 68 columns: [68][BLOCK] each
    float *xa = sm + threadIdx.x, *xb = sm + 68 * LRT_VAE_BLOCK + threadIdx.x;
    const uint32_t i = blockIdx.x * LRT_VAE_BLOCK + threadIdx.x;
    if (i >= A.n) return;
    CW B = reinterpret_cast<CW>((uintptr_t) A.blob);
    CW S = B + LRT_VAE_STATS;
    uint32_t v0, v1; tea32(A.seed, i, &v0, &v1);
    SamplerT<false> rng; rng.ld_count = 0; rng.seed(v0, v1);
    for (int k = 0; k < 20; ++k) xa[k * LRT_VAE_BLOCK] = (A.poly[20 * (size_t) i + k] - S[4 + k]) * S[24 + k];
    xa[20 * LRT_VAE_BLOCK] = A.albedo_norm; xa[21 * LRT_VAE_BLOCK] = A.g_norm; xa[22 * LRT_VAE_BLOCK] = A.ior_norm;
    vae_loop_layer<64, 23, true>(B + LRT_VAE_PRE0_W, B + LRT_VAE_PRE0_W + 64 * 23, xa, xb);                 // features: xb
    vae_matrix_layer(B + LRT_VAE_PRE1_W, B + LRT_VAE_PRE1_W + 64 * 64, xb, xa);                           // xa
    vae_matrix_layer(B + LRT_VAE_PRE2_W, B + LRT_VAE_PRE2_W + 64 * 64, xa, xb + 4 * LRT_VAE_BLOCK);       // features at xb[4..68): the decoder's input layout
    float *feat = xb + 4 * LRT_VAE_BLOCK;
    vae_loop_layer<32, 64, true>(B + LRT_VAE_ABS0_W, B + LRT_VAE_ABS0_W + 32 * 64, feat, xa);               // absorption head: xa[0..32)
    CW K = B + LRT_VAE_ABSD_K;
    float a = K[0] * xa[0];
    for (int k = 1; k < 32; ++k) a = fma_(K[k], xa[k * LRT_VAE_BLOCK], a);
    a = a + K[32];
    const float absorption = 1.0f / (1.0f + m_exp(-a));
    const V3 ip(A.in_pos[3 * (size_t) i], A.in_pos[3 * (size_t) i + 1], A.in_pos[3 * (size_t) i + 2]);
    if (!(rng.next() > absorption)) {                                   // all is absorbed
        A.out_pos[3 * (size_t) i] = ip.x; A.out_pos[3 * (size_t) i + 1] = ip.y; A.out_pos[3 * (size_t) i + 2] = ip.z;
        A.out_absorption[i] = 1.0f; return;
    }
    for (int h = 0; h < 2; ++h) {                                       // four Gaussian latents (vaehelper.cpp:14-27, warp.h square_to_std_normal)
        const float ux = rng.next(), uy = rng.next();
        const float r = __builtin_sqrtf(-2.f * m_log(1.f - ux)), phi = 2.f * kPi * uy;
        float s, c; m_sincos(phi, &s, &c);
        xb[(2 * h) * LRT_VAE_BLOCK] = c * r; xb[(2 * h + 1) * LRT_VAE_BLOCK] = s * r;
    }
    vae_loop_layer<64, 68, true>(B + LRT_VAE_DEC0_W, B + LRT_VAE_DEC0_W + 64 * 68, xb, xa);
    vae_loop_layer<64, 64, true>(B + LRT_VAE_DEC1_W, B + LRT_VAE_DEC1_W + 64 * 64, xa, xb);
    vae_loop_layer<64, 64, true>(B + LRT_VAE_DEC2_W, B + LRT_VAE_DEC2_W + 64 * 64, xb, xa);
    vae_loop_layer<3, 64, false>(B + LRT_VAE_OUT_K, B + LRT_VAE_OUT_K + 3 * 64, xa, xb);
    const float o0 = xb[0], o1 = xb[LRT_VAE_BLOCK], o2 = xb[2 * LRT_VAE_BLOCK];
    // localToWorld(inPos, -inDir, outPos, true) with onb(-inDir) (scattereigen.h:21-27,140-148), then the epsilon-space scale
    const float nx = -A.in_dir[3 * (size_t) i], ny = -A.in_dir[3 * (size_t) i + 1], nz = -A.in_dir[3 * (size_t) i + 2];
    const float sign = __builtin_copysignf(1.0f, nz), aa = -1.0f / (sign + nz), bb = nx * ny * aa;
    const float t1[3] = { 1.0f + sign * nx * nx * aa, sign * bb, -sign * nx }, t2[3] = { bb, sign + ny * ny * aa, -ny }, nn[3] = { nx, ny, nz };
    const float ipk[3] = { ip.x, ip.y, ip.z };
    for (int k = 0; k < 3; ++k) {
        const float w = ((ipk[k] + o0 * t1[k]) + o1 * t2[k]) + o2 * nn[k];
        A.out_pos[3 * (size_t) i + k] = ipk[k] + (w - ipk[k]) / A.fit_scale;
    }
    A.out_absorption[i] = 0.0f;
}

} // namespace lrt
