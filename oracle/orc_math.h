/*
 * orc_math.h -- float math of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * The oracle restates the reference's `llvm_ad_rgb` arithmetic in IEEE binary32.
 * The reference delegates transcendental functions to Dr.Jit 1.3.1
 * (`dr::log/exp/sincos/atan2/acos`, `pyproject.toml:5,19`), which is an
 * un-vendored submodule (ext/drjit is empty).  Dr.Jit's CPU implementations are
 * Cephes-derived polynomial kernels; they are restated here from the published
 * Cephes single-precision algorithms (cephes/single: logf.c, expf.c, sinf.c,
 * atanf.c, asinf.c) using explicit fmaf, so that every operation is an exactly
 * rounded IEEE operation and the result is reproducible on any IEEE machine.
 *
 * Build with -ffp-contract=off: the only fused operations are the explicit
 * fmaf() calls, which mirror the reference's dr::fmadd / dr::dot / dr::cross
 * usage (cited at each call site).
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

static const float kPi        = 3.14159265358979323846f;
static const float kTwoPi     = 6.28318530717958647692f;
static const float kInvPi     = 0.31830988618379067154f;
static const float kInvTwoPi  = 0.15915494309189533577f;
static const float kInvFourPi = 0.07957747154594766788f;
static const float kInf       = INFINITY;
/* include/mitsuba/core/math.h:18-23: RayEpsilon = eps*1500, ShadowEpsilon = 10*RayEpsilon,
   with eps = std::numeric_limits<float>::epsilon()/2 (Dr.Jit dr::Epsilon<float> = 2^-24) */
static const float kEpsilon       = 5.9604644775390625e-8f;     /* 2^-24 */
static const float kRayEpsilon    = kEpsilon * 1500.f;
static const float kShadowEpsilon = kRayEpsilon * 10.f;
static const float kLargest       = 3.402823466e+38f;           /* dr::Largest<float> */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float sqr(float x) { return x * x; }
static inline float rcp(float x) { return 1.f / x; }
static inline float rsqrt(float x) { return 1.f / sqrtf(x); }
static inline float safe_sqrt(float x) { return sqrtf(fmaxf(x, 0.f)); }
static inline float safe_rsqrt(float x) { return 1.f / sqrtf(fmaxf(x, 0.f)); }
static inline float mulsign(float a, float s) { return u2f(f2u(a) ^ (f2u(s) & 0x80000000u)); }
static inline float mulsign_neg(float a, float s) { return u2f(f2u(a) ^ (~f2u(s) & 0x80000000u)); }
static inline float signf(float x) { return u2f(0x3f800000u | (f2u(x) & 0x80000000u)); }
/* dr::lerp(a, b, t) = fmadd(b, t, fnmadd(a, t, a)) */
static inline float lerpf(float a, float b, float t) { return fmaf(b, t, fmaf(-a, t, a)); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* ---------------------------------------------------------------- logf (Cephes) */
static inline float m_log(float x) {
    if (x <= 0.f) return x == 0.f ? -kInf : NAN;
    if (x == kInf) return kInf;
    uint32_t ix = f2u(x);
    int e = (int) (ix >> 23) - 126;                  /* frexp exponent, x normal */
    float m = u2f((ix & 0x007fffffu) | 0x3f000000u); /* mantissa in [0.5, 1)     */
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.f; } else { m = m - 1.f; }
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = fmaf(y, m, -1.1514610310E-1f);
    y = fmaf(y, m, 1.1676998740E-1f);
    y = fmaf(y, m, -1.2420140846E-1f);
    y = fmaf(y, m, 1.4249322787E-1f);
    y = fmaf(y, m, -1.6668057665E-1f);
    y = fmaf(y, m, 2.0000714765E-1f);
    y = fmaf(y, m, -2.4999993993E-1f);
    y = fmaf(y, m, 3.3333331174E-1f);
    y = y * m * z;
    float fe = (float) e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    z = m + y;
    z = fmaf(0.693359375f, fe, z);
    return z;
}

/* --------------------------------------------------------------- log2f (Cephes)
   cephes/single/log2f.c: same reduction and polynomial as logf, then the fraction's logarithm is multiplied by
   log2(e) in two parts (LOG2EA = log2(e) - 1) and the exponent added.  Used by the bio media only
   (src/media/liver.cpp:332,376 `dr::log2(attIndex + 1.0f) / dr::log2(10.0f)`). */
static inline float m_log2(float x) {
    if (x <= 0.f) return x == 0.f ? -kInf : NAN;
    if (x == kInf) return kInf;
    uint32_t ix = f2u(x);
    int e = (int) (ix >> 23) - 126;
    float m = u2f((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.f; } else { m = m - 1.f; }
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = fmaf(y, m, -1.1514610310E-1f);
    y = fmaf(y, m, 1.1676998740E-1f);
    y = fmaf(y, m, -1.2420140846E-1f);
    y = fmaf(y, m, 1.4249322787E-1f);
    y = fmaf(y, m, -1.6668057665E-1f);
    y = fmaf(y, m, 2.0000714765E-1f);
    y = fmaf(y, m, -2.4999993993E-1f);
    y = fmaf(y, m, 3.3333331174E-1f);
    y = y * m * z;
    y = fmaf(-0.5f, z, y);
    const float LOG2EA = 0.44269504088896340735992f;
    z = y * LOG2EA;
    z = fmaf(m, LOG2EA, z);
    z += y;
    z += m;
    z += (float) e;
    return z;
}

/* ---------------------------------------------------------------- expf (Cephes) */
static inline float m_exp(float x) {
    if (x > 88.f) return kInf;
    if (!(x >= -86.f)) return (x != x) ? x : 0.f;
    float z = floorf(fmaf(1.44269504088896341f, x, 0.5f));
    x = fmaf(z, -0.693359375f, x);
    x = fmaf(z, 2.12194440e-4f, x);
    int n = (int) z;
    z = x * x;
    float p = 1.9875691500E-4f;
    p = fmaf(p, x, 1.3981999507E-3f);
    p = fmaf(p, x, 8.3334519073E-3f);
    p = fmaf(p, x, 4.1665795894E-2f);
    p = fmaf(p, x, 1.6666665459E-1f);
    p = fmaf(p, x, 5.0000001201E-1f);
    p = fmaf(p, z, x) + 1.f;
    return p * u2f((uint32_t) (n + 127) << 23);
}

/* -------------------------------------------------------- sinf / cosf (Cephes) */
static inline void m_sincos(float xx, float *s_out, float *c_out) {
    float x = fabsf(xx);
    int j = (int) (1.27323954473516f * x);           /* 4/pi */
    float y = (float) j;
    if (j & 1) { j += 1; y += 1.f; }
    j &= 7;
    x = fmaf(y, -0.78515625f, x);
    x = fmaf(y, -2.4187564849853515625e-4f, x);
    x = fmaf(y, -3.77489497744594108e-8f, x);
    float z = x * x;
    float ps = -1.9515295891E-4f;
    ps = fmaf(ps, z, 8.3321608736E-3f);
    ps = fmaf(ps, z, -1.6666654611E-1f);
    ps = fmaf(ps * z, x, x);
    float pc = 2.443315711809948E-005f;
    pc = fmaf(pc, z, -1.388731625493765E-003f);
    pc = fmaf(pc, z, 4.166664568298827E-002f);
    pc = fmaf(pc * z, z, fmaf(-0.5f, z, 1.f));
    int js = j, jc = j;
    float ssign = (xx < 0.f) ? -1.f : 1.f, csign = 1.f;
    if (js > 3) { ssign = -ssign; js -= 4; }
    if (jc > 3) { csign = -csign; jc -= 4; }
    if (jc > 1) csign = -csign;
    bool swap = (js == 1 || js == 2);
    *s_out = ssign * (swap ? pc : ps);
    *c_out = csign * (swap ? ps : pc);
}

/* -------------------------------------------------------------- atanf / atan2f */
static inline float m_atan(float xx) {
    float x = fabsf(xx), y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.f) / (x + 1.f); }
    else y = 0.f;
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = fmaf(p, z, -1.38776856032E-1f);
    p = fmaf(p, z, 1.99777106478E-1f);
    p = fmaf(p, z, -3.33329491539E-1f);
    y += fmaf(p * z, x, x);
    return (xx < 0.f) ? -y : y;
}

static inline float m_atan2(float y, float x) {
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

/* -------------------------------------------------------------- asinf / acosf */
static inline float m_asin(float xx) {
    float a = fabsf(xx), x, z;
    if (a > 1.f) return NAN;
    if (a < 1.0e-4f) return xx;
    bool flag = a > 0.5f;
    if (flag) { z = 0.5f * (1.f - a); x = sqrtf(z); }
    else { x = a; z = x * x; }
    float p = 4.2163199048E-2f;
    p = fmaf(p, z, 2.4181311049E-2f);
    p = fmaf(p, z, 4.5470025998E-2f);
    p = fmaf(p, z, 7.4953002686E-2f);
    p = fmaf(p, z, 1.6666752422E-1f);
    z = fmaf(p * z, x, x);
    if (flag) { z = z + z; z = 1.5707963267948966192f - z; }
    return (xx < 0.f) ? -z : z;
}

static inline float m_acos(float x) {
    if (x < -0.5f) return kPi - 2.f * m_asin(sqrtf(0.5f * (1.f + x)));
    if (x > 0.5f) return 2.f * m_asin(sqrtf(0.5f * (1.f - x)));
    return 1.5707963267948966192f - m_asin(x);
}
static inline float safe_acos(float x) { return m_acos(clampf(x, -1.f, 1.f)); }

/* ------------------------------------------------------------------- vectors */
struct V2 { float x, y; };
struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a) : x(a), y(a), z(a) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator/(V3 a, V3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
static inline V3 &operator+=(V3 &a, V3 b) { a = a + b; return a; }
static inline V3 &operator*=(V3 &a, V3 b) { a = a * b; return a; }
static inline V3 &operator*=(V3 &a, float s) { a = a * s; return a; }
/* dr::dot: fmadd chain, x first */
static inline float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
/* dr::cross: fmsub(a.yzx, b.zxy, a.zxy * b.yzx) */
static inline V3 cross(V3 a, V3 b) {
    return V3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float squared_norm(V3 a) { return dot(a, a); }
static inline float norm(V3 a) { return sqrtf(dot(a, a)); }
static inline V3 normalize(V3 a) { return a * rsqrt(dot(a, a)); }
/* dr::fmadd(a, s, b) per component */
static inline V3 fma3(V3 a, float s, V3 b) { return V3(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); }
static inline float max3(V3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
static inline V3 abs3(V3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline bool any_nonzero(V3 a) { return a.x != 0.f || a.y != 0.f || a.z != 0.f; }
static inline float mean3(V3 a) { return (a.x + a.y + a.z) * (1.f / 3.f); }
/* include/mitsuba/core/spectrum.h luminance() for linear sRGB */
static inline float luminance(V3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }

/* include/mitsuba/core/vector.h:118-138 coordinate_system() */
static inline void coordinate_system(V3 n, V3 *s, V3 *t) {
    float sign = signf(n.z), a = -rcp(sign + n.z), b = n.x * n.y * a;
    *s = V3(mulsign(sqr(n.x) * a, n.z) + 1.f, mulsign(b, n.z), mulsign_neg(n.x, n.z));
    *t = V3(b, fmaf(n.y, n.y * a, sign), -n.y);
}

/* include/mitsuba/core/frame.h:20-40 */
struct Frame {
    V3 s, t, n;
    Frame() {}
    explicit Frame(V3 v) : n(v) { coordinate_system(v, &s, &t); }
    V3 to_local(V3 v) const { return V3(dot(v, s), dot(v, t), dot(v, n)); }
    V3 to_world(V3 v) const { return fma3(n, v.z, fma3(t, v.y, s * v.x)); }
};

/* row-major 4x4 helpers */
struct M4 { float m[16]; };
/* include/mitsuba/core/transform.h:296-309 (affine point), :261-272 (vector) */
static inline V3 xform_point(const M4 &M, V3 p) {
    const float *m = M.m;
    return V3(fmaf(m[2], p.z, fmaf(m[1], p.y, fmaf(m[0], p.x, m[3]))),
              fmaf(m[6], p.z, fmaf(m[5], p.y, fmaf(m[4], p.x, m[7]))),
              fmaf(m[10], p.z, fmaf(m[9], p.y, fmaf(m[8], p.x, m[11]))));
}
static inline V3 xform_vec(const M4 &M, V3 v) {
    const float *m = M.m;
    return V3(fmaf(m[2], v.z, fmaf(m[1], v.y, m[0] * v.x)),
              fmaf(m[6], v.z, fmaf(m[5], v.y, m[4] * v.x)),
              fmaf(m[10], v.z, fmaf(m[9], v.y, m[8] * v.x)));
}
/* include/mitsuba/core/transform.h:309-319: projective point with perspective division */
static inline V3 xform_point_proj(const M4 &M, V3 p) {
    const float *m = M.m;
    float r[4];
    for (int i = 0; i < 4; ++i)
        r[i] = fmaf(m[4 * i + 2], p.z, fmaf(m[4 * i + 1], p.y, fmaf(m[4 * i + 0], p.x, m[4 * i + 3])));
    return V3(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
}

} // namespace orc
