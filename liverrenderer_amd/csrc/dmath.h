// Device float math for the gfx950 kernels.
//
// Every operation is an exactly rounded IEEE binary32 operation (compile with
// -ffp-contract=off; fusion only through explicit __builtin_fmaf, mirroring the
// reference's dr::fmadd / dr::dot / dr::cross usage), so that a path evaluated
// on the GPU reproduces the CPU restatement of the reference bit for bit.
// Transcendentals are Cephes-style polynomial kernels (the family Dr.Jit's own
// implementations derive from) instead of the approximate v_log/v_exp/v_sin
// hardware instructions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lrt {

#define DEV __device__ __forceinline__

static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kTwoPi = 6.28318530717958647692f;
static constexpr float kInvPi = 0.31830988618379067154f;
static constexpr float kInvTwoPi = 0.15915494309189533577f;
static constexpr float kInvFourPi = 0.07957747154594766788f;
static constexpr float kEpsilon = 5.9604644775390625e-8f;
static constexpr float kRayEpsilon = kEpsilon * 1500.f;
static constexpr float kShadowEpsilon = kRayEpsilon * 10.f;
static constexpr float kLargest = 3.402823466e+38f;
#define kInf __builtin_inff()

DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEV f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32 on gfx950
DEV uint32_t f2u(float f) { return __float_as_uint(f); }
DEV float u2f(uint32_t u) { return __uint_as_float(u); }
DEV float sqr(float x) { return x * x; }
DEV float rcp(float x) { return 1.f / x; }
DEV float rsqrt_(float x) { return 1.f / __builtin_sqrtf(x); }
DEV float safe_sqrt(float x) { return __builtin_sqrtf(__builtin_fmaxf(x, 0.f)); }
DEV float safe_rsqrt(float x) { return 1.f / __builtin_sqrtf(__builtin_fmaxf(x, 0.f)); }
// Sum of `v` over the 64 lanes of the wave (every lane must be active), returned to all of them: an inclusive scan with
// DPP row shifts (1, 2, 4, 8) and the two row broadcasts, then a read of lane 63: six v_add_f32_dpp instead of six
// ds_bpermute round trips (lanes the EXEC mask disables and out-of-row sources read as 0: bound_ctrl, old = 0).
template <int CTRL, int ROW_MASK> DEV float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
DEV float wave_sum(float v) {
    v = dpp_add<0x111, 0xf>(v); v = dpp_add<0x112, 0xf>(v); v = dpp_add<0x114, 0xf>(v); v = dpp_add<0x118, 0xf>(v);   // row_shr:1,2,4,8
    v = dpp_add<0x142, 0xa>(v);                                                                                           // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);                                                                                           // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// Segmented inclusive sums over the wave (every lane active): lanes are grouped into runs of consecutive lanes (`head` marks the first lane
// of a run); after the call lane i holds the sum of v over the lanes of its run up to and including i, for each of the N values at once.
// Hillis-Steele steps with DPP row shifts (1, 2, 4, 8) and the two row broadcasts, each guarded by "a run began within the lanes skipped":
// 2 VALU instructions per value and step, no LDS traffic, no loop over the runs.
template <int N>
DEV void wave_segmented_sums(float (&v)[N], bool head) {
    uint32_t f = head ? 1u : 0u;
#define LRT_SEG_STEP(CTRL, ROW_MASK) { \
        const uint32_t pf = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) f, CTRL, ROW_MASK, 0xf, true); \
        _Pragma("unroll") for (int k = 0; k < N; ++k) { \
            const float pv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), CTRL, ROW_MASK, 0xf, true)); \
            v[k] = f ? v[k] : v[k] + pv; } \
        f |= pf; }
    LRT_SEG_STEP(0x111, 0xf) LRT_SEG_STEP(0x112, 0xf) LRT_SEG_STEP(0x114, 0xf) LRT_SEG_STEP(0x118, 0xf)      // row_shr:1, 2, 4, 8
    LRT_SEG_STEP(0x142, 0xa) LRT_SEG_STEP(0x143, 0xc)                                                          // row_bcast:15 -> rows 1, 3; row_bcast:31 -> rows 2, 3
#undef LRT_SEG_STEP
}
// the value of the previous / next lane of the wave (lane 0 / lane 63: `edge`): DPP wave shifts, every lane active
DEV uint32_t wave_prev(uint32_t x, uint32_t edge) { return (uint32_t) __builtin_amdgcn_update_dpp((int) edge, (int) x, 0x138, 0xf, 0xf, false); }   // wave_shr:1
DEV uint32_t wave_next(uint32_t x, uint32_t edge) { return (uint32_t) __builtin_amdgcn_update_dpp((int) edge, (int) x, 0x130, 0xf, 0xf, false); }   // wave_shl:1

DEV float mulsign(float a, float s) { return u2f(f2u(a) ^ (f2u(s) & 0x80000000u)); }
DEV float mulsign_neg(float a, float s) { return u2f(f2u(a) ^ (~f2u(s) & 0x80000000u)); }
DEV float signf_(float x) { return u2f(0x3f800000u | (f2u(x) & 0x80000000u)); }
DEV float lerpf(float a, float b, float t) { return fma_(b, t, fma_(-a, t, a)); }
DEV float fmin_(float a, float b) { return __builtin_fminf(a, b); }
DEV float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
DEV float clampf(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
DEV bool finite_(float x) { return (f2u(x) & 0x7f800000u) != 0x7f800000u; }

DEV float m_log(float x) {
    if (x <= 0.f) return x == 0.f ? -kInf : __builtin_nanf("");
    if (x == kInf) return kInf;
    uint32_t ix = f2u(x);
    int e = (int) (ix >> 23) - 126;
    float m = u2f((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.f; } else { m = m - 1.f; }
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = fma_(y, m, -1.1514610310E-1f);
    y = fma_(y, m, 1.1676998740E-1f);
    y = fma_(y, m, -1.2420140846E-1f);
    y = fma_(y, m, 1.4249322787E-1f);
    y = fma_(y, m, -1.6668057665E-1f);
    y = fma_(y, m, 2.0000714765E-1f);
    y = fma_(y, m, -2.4999993993E-1f);
    y = fma_(y, m, 3.3333331174E-1f);
    y = y * m * z;
    float fe = (float) e;
    y = fma_(-2.12194440e-4f, fe, y);
    y = fma_(-0.5f, z, y);
    z = m + y;
    z = fma_(0.693359375f, fe, z);
    return z;
}

// cephes/single/log2f.c (bio media: liver.cpp:332,376)
DEV float m_log2(float x) {
    if (x <= 0.f) return x == 0.f ? -kInf : __builtin_nanf("");
    if (x == kInf) return kInf;
    uint32_t ix = f2u(x);
    int e = (int) (ix >> 23) - 126;
    float m = u2f((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.f; } else { m = m - 1.f; }
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = fma_(y, m, -1.1514610310E-1f);
    y = fma_(y, m, 1.1676998740E-1f);
    y = fma_(y, m, -1.2420140846E-1f);
    y = fma_(y, m, 1.4249322787E-1f);
    y = fma_(y, m, -1.6668057665E-1f);
    y = fma_(y, m, 2.0000714765E-1f);
    y = fma_(y, m, -2.4999993993E-1f);
    y = fma_(y, m, 3.3333331174E-1f);
    y = y * m * z;
    y = fma_(-0.5f, z, y);
    const float LOG2EA = 0.44269504088896340735992f;
    z = y * LOG2EA;
    z = fma_(m, LOG2EA, z);
    z += y;
    z += m;
    z += (float) e;
    return z;
}

DEV float m_exp(float x) {
    if (x > 88.f) return kInf;
    if (!(x >= -86.f)) return (x != x) ? x : 0.f;
    float z = __builtin_floorf(fma_(1.44269504088896341f, x, 0.5f));
    x = fma_(z, -0.693359375f, x);
    x = fma_(z, 2.12194440e-4f, x);
    int n = (int) z;
    z = x * x;
    float p = 1.9875691500E-4f;
    p = fma_(p, x, 1.3981999507E-3f);
    p = fma_(p, x, 8.3334519073E-3f);
    p = fma_(p, x, 4.1665795894E-2f);
    p = fma_(p, x, 1.6666665459E-1f);
    p = fma_(p, x, 5.0000001201E-1f);
    p = fma_(p, z, x) + 1.f;
    return p * u2f((uint32_t) (n + 127) << 23);
}

DEV void m_sincos(float xx, float *s_out, float *c_out) {
    float x = __builtin_fabsf(xx);
    int j = (int) (1.27323954473516f * x);
    float y = (float) j;
    if (j & 1) { j += 1; y += 1.f; }
    j &= 7;
    x = fma_(y, -0.78515625f, x);
    x = fma_(y, -2.4187564849853515625e-4f, x);
    x = fma_(y, -3.77489497744594108e-8f, x);
    float z = x * x;
    float ps = -1.9515295891E-4f;
    ps = fma_(ps, z, 8.3321608736E-3f);
    ps = fma_(ps, z, -1.6666654611E-1f);
    ps = fma_(ps * z, x, x);
    float pc = 2.443315711809948E-005f;
    pc = fma_(pc, z, -1.388731625493765E-003f);
    pc = fma_(pc, z, 4.166664568298827E-002f);
    pc = fma_(pc * z, z, fma_(-0.5f, z, 1.f));
    int js = j, jc = j;
    float ssign = (xx < 0.f) ? -1.f : 1.f, csign = 1.f;
    if (js > 3) { ssign = -ssign; js -= 4; }
    if (jc > 3) { csign = -csign; jc -= 4; }
    if (jc > 1) csign = -csign;
    bool swap = (js == 1 || js == 2);
    *s_out = ssign * (swap ? pc : ps);
    *c_out = csign * (swap ? ps : pc);
}

DEV float m_atan(float xx) {
    float x = __builtin_fabsf(xx), y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.f) / (x + 1.f); }
    else y = 0.f;
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = fma_(p, z, -1.38776856032E-1f);
    p = fma_(p, z, 1.99777106478E-1f);
    p = fma_(p, z, -3.33329491539E-1f);
    y += fma_(p * z, x, x);
    return (xx < 0.f) ? -y : y;
}

DEV float m_atan2(float y, float x) {
    if (x == 0.f) {
        if (y > 0.f) return 1.5707963267948966192f;
        if (y < 0.f) return -1.5707963267948966192f;
        return 0.f;
    }
    if (y == 0.f) return (x < 0.f) ? kPi : 0.f;
    float w = 0.f;
    if (x < 0.f) w = (y < 0.f) ? -kPi : kPi;
    return w + m_atan(y / x);
}

DEV float m_asin(float xx) {
    float a = __builtin_fabsf(xx), x, z;
    if (a > 1.f) return __builtin_nanf("");
    if (a < 1.0e-4f) return xx;
    bool flag = a > 0.5f;
    if (flag) { z = 0.5f * (1.f - a); x = __builtin_sqrtf(z); }
    else { x = a; z = x * x; }
    float p = 4.2163199048E-2f;
    p = fma_(p, z, 2.4181311049E-2f);
    p = fma_(p, z, 4.5470025998E-2f);
    p = fma_(p, z, 7.4953002686E-2f);
    p = fma_(p, z, 1.6666752422E-1f);
    z = fma_(p * z, x, x);
    if (flag) { z = z + z; z = 1.5707963267948966192f - z; }
    return (xx < 0.f) ? -z : z;
}

DEV float m_acos(float x) {
    if (x < -0.5f) return kPi - 2.f * m_asin(__builtin_sqrtf(0.5f * (1.f + x)));
    if (x > 0.5f) return 2.f * m_asin(__builtin_sqrtf(0.5f * (1.f - x)));
    return 1.5707963267948966192f - m_asin(x);
}
DEV float safe_acos(float x) { return m_acos(clampf(x, -1.f, 1.f)); }

// ------------------------------------------------------------------ vectors
struct V2 { float x, y; };
struct V3 {
    float x, y, z;
    DEV V3() {}
    DEV explicit V3(float a) : x(a), y(a), z(a) {}
    DEV V3(float a, float b, float c) : x(a), y(b), z(c) {}
};
DEV V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV V3 operator/(V3 a, V3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
DEV V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
DEV V3 operator*(float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
DEV V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
DEV V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
DEV float dot(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
DEV V3 cross(V3 a, V3 b) {
    return V3(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}
DEV float squared_norm(V3 a) { return dot(a, a); }
DEV float norm(V3 a) { return __builtin_sqrtf(dot(a, a)); }
DEV V3 normalize(V3 a) { return a * rsqrt_(dot(a, a)); }
DEV V3 fma3(V3 a, float s, V3 b) { return V3(fma_(a.x, s, b.x), fma_(a.y, s, b.y), fma_(a.z, s, b.z)); }
DEV float max3(V3 a) { return fmax_(fmax_(a.x, a.y), a.z); }
DEV V3 abs3(V3 a) { return V3(__builtin_fabsf(a.x), __builtin_fabsf(a.y), __builtin_fabsf(a.z)); }
DEV bool any_nonzero(V3 a) { return a.x != 0.f || a.y != 0.f || a.z != 0.f; }
DEV bool finite3(V3 a) { return (f2u(a.x) & 0x7f800000u) != 0x7f800000u && (f2u(a.y) & 0x7f800000u) != 0x7f800000u && (f2u(a.z) & 0x7f800000u) != 0x7f800000u; }
DEV float mean3(V3 a) { return (a.x + a.y + a.z) * (1.f / 3.f); }
DEV float luminance(V3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }
DEV float idx3(V3 v, uint32_t c) { return c == 0 ? v.x : (c == 1 ? v.y : v.z); }

struct Basis { V3 s, t; };
DEV Basis coordinate_system(V3 n) {
    float sign = signf_(n.z), a = -rcp(sign + n.z), b = n.x * n.y * a;
    Basis r;
    r.s = V3(mulsign(sqr(n.x) * a, n.z) + 1.f, mulsign(b, n.z), mulsign_neg(n.x, n.z));
    r.t = V3(b, fma_(n.y, n.y * a, sign), -n.y);
    return r;
}

struct Frame {
    V3 s, t, n;
    DEV Frame() {}
    DEV explicit Frame(V3 v) : n(v) { Basis b = coordinate_system(v); s = b.s; t = b.t; }
    DEV V3 to_local(V3 v) const { return V3(dot(v, s), dot(v, t), dot(v, n)); }
    DEV V3 to_world(V3 v) const { return fma3(n, v.z, fma3(t, v.y, s * v.x)); }
};

// rows 0..2 of a row-major affine matrix stored as 12 floats
template <typename FP> DEV V3 xform_point12(FP m, V3 p) {      // FP: pointer to 12 floats in any address space
    return V3(fma_(m[2], p.z, fma_(m[1], p.y, fma_(m[0], p.x, m[3]))), fma_(m[6], p.z, fma_(m[5], p.y, fma_(m[4], p.x, m[7]))),
              fma_(m[10], p.z, fma_(m[9], p.y, fma_(m[8], p.x, m[11]))));
}
template <typename FP> DEV V3 xform_vec12(FP m, V3 v) {
    return V3(fma_(m[2], v.z, fma_(m[1], v.y, m[0] * v.x)), fma_(m[6], v.z, fma_(m[5], v.y, m[4] * v.x)),
              fma_(m[10], v.z, fma_(m[9], v.y, m[8] * v.x)));
}
template <typename FP> DEV V3 xform_vec9(FP m, V3 v) {
    return V3(fma_(m[2], v.z, fma_(m[1], v.y, m[0] * v.x)), fma_(m[5], v.z, fma_(m[4], v.y, m[3] * v.x)),
              fma_(m[8], v.z, fma_(m[7], v.y, m[6] * v.x)));
}

// ----------------------------------------------------------------------- RNG
DEV void tea32(uint32_t v0, uint32_t v1, uint32_t *o0, uint32_t *o1) {
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    *o0 = v0; *o1 = v1;
}

DEV void tea32_rounds2(uint32_t v0, uint32_t v1, uint32_t *o0, uint32_t *o1) {
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    *o0 = v0; *o1 = v1;
}

// The lane's sampler as the integrators see it.
//  * independent (src/samplers/independent.cpp): a PCG32 stream (Dr.Jit's PCG32 = O'Neill's XSH-RR); a masked-out call
//    draws nothing, so calls simply sit inside the branches.
//  * low-discrepancy (src/samplers/ldsampler.cpp:112-146, ld_count != 0): a sample is a pure function of (sample index in
//    the pixel, dimension counter, per-pixel scramble seed): (0,2)-sequence point `permute(index, count, scramble + dim)`
//    (TEA shuffling network, include/mitsuba/core/random.h:196-214), radical inverse / Sobol' dimension 2
//    (include/mitsuba/core/qmc.h:189-252) with TEA scrambles.  EVERY call the traced loop body contains bumps the
//    counter of every lane that is in the loop, whatever the call's mask; skip() stands for the calls a lane's own
//    control flow does not reach.  Layout: state = dimension counter, inc = scramble_seed | sample_index << 32.
// LD is a compile-time switch (separate kernel instantiations) so that the independent sampler's hot path carries no
// trace of the other mode.
template <bool LD>
struct SamplerT {
    uint64_t state, inc;
    uint32_t ld_count;         // LD: the sampler's (rounded) sample count, wave-uniform
    uint32_t ld_s1, ld_s2x, ld_s2y;   // LD: the 1-D and 2-D scrambles, functions of the per-pixel seed alone (ldsampler.cpp:118-121,133-137): evaluated once per trip (ld_prepare)
    DEV void ld_prepare() {
        if (LD) { uint32_t t; tea32((uint32_t) inc, 0x48bc48ebu, &ld_s1, &t); tea32((uint32_t) inc, 0x98bc51abu, &ld_s2x, &ld_s2y); }
    }
    DEV uint32_t next_u32() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t) (((old >> 18) ^ old) >> 27), rot = (uint32_t) (old >> 59);
        return (xs >> rot) | (xs << ((0u - rot) & 31u));
    }
    DEV uint32_t ld_point() {                        // permuted sample index for the current dimension; advances it
        const uint32_t scramble_seed = (uint32_t) inc;
        uint32_t index = (uint32_t) (inc >> 32);
        const uint32_t perm_seed = scramble_seed + (uint32_t) state;
        state += 1;
        for (uint32_t bit = 1; bit < ld_count; bit <<= 1) {
            uint32_t r0, r1; tea32_rounds2(index | bit, perm_seed, &r0, &r1);
            if (r0 & bit) index ^= bit;
        }
        return index;
    }
    DEV static float radical_inverse_2(uint32_t index, uint32_t scramble) {
        return u2f(((__builtin_bitreverse32(index) ^ scramble) >> 9) | 0x3f800000u) - 1.f;
    }
    DEV static float sobol_2(uint32_t index, uint32_t scramble) {
        for (uint32_t v = 1u << 31; index != 0; index >>= 1, v ^= v >> 1)
            if (index & 1u) scramble ^= v;
        return (float) scramble / 4294967296.f;
    }
    DEV float next() {
        if (LD) {
            const uint32_t i = ld_point();
            return radical_inverse_2(i, ld_s1);
        }
        return u2f((next_u32() >> 9) | 0x3f800000u) - 1.f;
    }
    DEV void next2(float &x, float &y) {
        if (LD) {
            const uint32_t i = ld_point();
            x = radical_inverse_2(i, ld_s2x); y = sobol_2(i, ld_s2y);
            return;
        }
        x = u2f((next_u32() >> 9) | 0x3f800000u) - 1.f; y = u2f((next_u32() >> 9) | 0x3f800000u) - 1.f;
    }
    DEV void skip(uint32_t n) { if (LD) state += n; }
    DEV void seed(uint64_t initstate, uint64_t initseq) {
        state = 0; inc = (initseq << 1) | 1u; next_u32(); state += initstate; next_u32();
    }
};
typedef SamplerT<false> PCG32;

} // namespace lrt
