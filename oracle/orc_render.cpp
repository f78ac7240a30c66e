/*
 * orc_render.cpp -- the sample loop of the CPU oracle: RNG, sensor, BSDFs,
 * phase functions, media, emitters, the `path` and `volpath` integrators and
 * the film.  TEST INFRASTRUCTURE ONLY (see orc.h).  Every function cites the
 * reference lines it restates (paths relative to the reference tree).
 */
#include "orc_scene.h"
#include <thread>
#include <mutex>
#include <cstdio>
#include <algorithm>

namespace orc {

/* ------------------------------------------------------------------- RNG */
/* include/mitsuba/core/random.h:76-91 */
static inline void tea32(uint32_t v0, uint32_t v1, int rounds, uint32_t *o0, uint32_t *o1) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    *o0 = v0; *o1 = v1;
}

/* Dr.Jit 1.3.1 drjit/random.h PCG32 (un-vendored): O'Neill's PCG32-XSH-RR */
struct PCG32 {
    uint64_t state, inc;
    void seed(uint64_t initstate, uint64_t initseq) {
        state = 0; inc = (initseq << 1) | 1u; next_u32(); state += initstate; next_u32();
    }
    uint32_t next_u32() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t) (((old >> 18) ^ old) >> 27), rot = (uint32_t) (old >> 59);
        return (xs >> rot) | (xs << ((0u - rot) & 31u));
    }
    float next() { return u2f((next_u32() >> 9) | 0x3f800000u) - 1.f; }
};

/* src/render/sampler.cpp:129-148 (JIT branch): per-lane stream */
static inline PCG32 lane_rng(uint32_t base_seed, uint32_t seed, uint32_t lane) {
    uint32_t v0, v1; tea32(base_seed + seed, lane, 4, &v0, &v1);
    PCG32 r; r.seed(v0, v1); return r;
}

/* src/samplers/ldsampler.cpp:75-160 (+ include/mitsuba/core/random.h:196-214 permute,
   include/mitsuba/core/qmc.h:189-252 radical_inverse_2 / sobol_2, src/render/sampler.cpp:97-117).
   A sample is a pure function of (sample index within the pixel, dimension counter, per-pixel scramble seed). */
static inline uint32_t permute_tea(uint32_t index, uint32_t size, uint32_t seed) {     /* random.h:196-214 permute(), rounds = 2 */
    for (uint32_t bit = 1; bit < size; bit <<= 1) {                                     /* size is a power of two */
        uint32_t r0, r1; tea32(index | bit, seed, 2, &r0, &r1);
        if (r0 & bit) index ^= bit;
    }
    return index;
}
static inline float radical_inverse_2(uint32_t index, uint32_t scramble) {
    index = (index << 16) | (index >> 16);
    index = ((index & 0x00ff00ffu) << 8) | ((index & 0xff00ff00u) >> 8);
    index = ((index & 0x0f0f0f0fu) << 4) | ((index & 0xf0f0f0f0u) >> 4);
    index = ((index & 0x33333333u) << 2) | ((index & 0xccccccccu) >> 2);
    index = ((index & 0x55555555u) << 1) | ((index & 0xaaaaaaaau) >> 1);
    return u2f(((index ^ scramble) >> 9) | 0x3f800000u) - 1.f;
}
static inline float sobol_2(uint32_t index, uint32_t scramble) {
    for (uint32_t v = 1u << 31; index != 0; index >>= 1, v ^= v >> 1)
        if (index & 1u) scramble ^= v;
    return (float) scramble / 4294967296.f;             /* Float(scramble) / Float(1ULL << 32) */
}
/* LowDiscrepancySampler::set_sample_count (ldsampler.cpp:83-93): round up to a square power of two */
static inline uint32_t ld_round_sample_count(uint32_t spp) {
    uint32_t res = 2;
    while (res * res < spp) { ++res; uint32_t p = 1; while (p < res) p <<= 1; res = p; }
    return res * res;
}

/* The integrators' view of a sampler.  Independent: the lane's PCG32 stream, a masked-out call draws nothing.
   Low-discrepancy: EVERY call the loop body executes bumps the dimension counter of every lane that is in the loop,
   whatever the call's mask (m_dimension_index++ is unmasked and the body of a symbolic loop is traced once, with all
   its `if (dr::any_or<true>(...))` blocks); skip() stands for the calls a lane's own control flow does not reach. */
struct Sampler {
    int type = LRT_SAMPLER_INDEPENDENT;
    PCG32 rng;
    uint32_t dim = 0, scramble_seed = 0, sample_index = 0, sample_count = 1;
    float next1() {
        if (type != LRT_SAMPLER_LD) return rng.next();
        uint32_t perm_seed = scramble_seed + dim++;
        uint32_t i = permute_tea(sample_index, sample_count, perm_seed);
        uint32_t s0, s1; tea32(scramble_seed, 0x48bc48ebu, 4, &s0, &s1);
        return radical_inverse_2(i, s0);
    }
    void next2(float *x, float *y) {
        if (type != LRT_SAMPLER_LD) { *x = rng.next(); *y = rng.next(); return; }
        uint32_t perm_seed = scramble_seed + dim++;
        uint32_t i = permute_tea(sample_index, sample_count, perm_seed);
        uint32_t sx, sy; tea32(scramble_seed, 0x98bc51abu, 4, &sx, &sy);
        *x = radical_inverse_2(i, sx); *y = sobol_2(i, sy);
    }
    void skip(uint32_t n) { if (type == LRT_SAMPLER_LD) dim += n; }
};
/* Sampler::seed in the JIT branch of SamplingIntegrator::render (integrator.cpp:308-311, sampler.cpp:97-117):
   sequence = the lane's pixel (samples_per_wavefront = spp), sample index = lane % spp */
static inline Sampler lane_sampler(int type, uint32_t base_seed, uint32_t seed, uint32_t lane, uint32_t spp) {
    Sampler s; s.type = type;
    if (type == LRT_SAMPLER_LD) {
        uint32_t v0, v1; tea32(base_seed, spp * (lane / spp) + seed, 4, &v0, &v1);
        s.scramble_seed = v0; s.sample_index = lane % spp; s.sample_count = spp; s.dim = 0;
    } else s.rng = lane_rng(base_seed, seed, lane);
    return s;
}

/* ---------------------------------------------------------------- sensor */
/* src/sensors/perspective.cpp:239-279 (differentials unused: volpath.cpp:107, no ray-differential consumers) */
static Ray sample_ray(const Scene &S, float px, float py) {
    const lrt_film_desc &F = S.d.film;                           /* perspective.cpp:214-221 */
    float ppx = (float) F.width * S.d.sensor.principal_point_offset_x / (float) F.crop_width, ppy = (float) F.height * S.d.sensor.principal_point_offset_y / (float) F.crop_height;
    V3 near_p = xform_point_proj(S.sample_to_camera, V3(px + ppx, py + ppy, 0.f));
    V3 d = normalize(near_p);
    Ray r;
    r.o = V3(S.cam_to_world.m[3], S.cam_to_world.m[7], S.cam_to_world.m[11]);
    r.d = xform_vec(S.cam_to_world, d);
    float inv_z = rcp(d.z), near_t = S.d.sensor.near_clip * inv_z, far_t = S.d.sensor.far_clip * inv_z;
    r.o = r.o + r.d * near_t;
    r.maxt = far_t - near_t;
    return r;
}

/* -------------------------------------------------------------- warping */
/* include/mitsuba/core/warp.h:54-92 */
static V2 square_to_uniform_disk_concentric(float sx, float sy) {
    float x = fmaf(2.f, sx, -1.f), y = fmaf(2.f, sy, -1.f);
    bool is_zero = (x == 0.f) && (y == 0.f), q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * kPi * rp / r;
    if (q13) phi = 0.5f * kPi - phi;
    if (is_zero) phi = 0.f;
    float s, c; m_sincos(phi, &s, &c);
    return { r * c, r * s };
}
/* include/mitsuba/core/warp.h:412-420 */
static V3 square_to_cosine_hemisphere(float sx, float sy) {
    V2 p = square_to_uniform_disk_concentric(sx, sy);
    float z = safe_sqrt(1.f - fmaf(p.x, p.x, p.y * p.y));
    return V3(p.x, p.y, z);
}
/* include/mitsuba/core/warp.h:250-256 */
static V3 square_to_uniform_sphere(float sx, float sy) {
    float z = fmaf(-2.f, sy, 1.f), r = safe_sqrt(fmaf(-z, z, 1.f));
    float s, c; m_sincos(2.f * kPi * sx, &s, &c);
    return V3(r * c, r * s, z);
}

/* ---------------------------------------------------------- interactions */
/* include/mitsuba/render/interaction.h:140-168 */
static inline V3 offset_p(V3 p, V3 n, V3 d) {
    float mag = (1.f + max3(abs3(p))) * kRayEpsilon;
    mag = mulsign(mag, dot(n, d));
    return fma3(n, mag, p);
}
static inline Ray spawn_ray(V3 p, V3 n, V3 d) { Ray r; r.o = offset_p(p, n, d); r.d = d; r.maxt = kLargest; return r; }
static inline Ray spawn_ray_to(V3 p, V3 n, V3 t) {
    Ray r; r.o = offset_p(p, n, t - p);
    V3 d = t - r.o; float dist = norm(d);
    r.d = d / dist; r.maxt = dist * (1.f - kShadowEpsilon);
    return r;
}

/* -------------------------------------------------------------- textures */
static V3 tex_eval(const Scene &S, int tex, const SI &si) {
    const lrt_texture_desc &T = S.textures[tex];
    if (T.type == LRT_TEX_RGB) return V3(T.color0[0], T.color0[1], T.color0[2]);
    if (T.type == LRT_TEX_CHECKERBOARD) {       /* src/textures/checkerboard.cpp:70-88 */
        float u = fmaf(T.to_uv[1], si.uv.y, fmaf(T.to_uv[0], si.uv.x, T.to_uv[2]));
        float v = fmaf(T.to_uv[4], si.uv.y, fmaf(T.to_uv[3], si.uv.x, T.to_uv[5]));
        bool mx = u - floorf(u) > .5f, my = v - floorf(v) > .5f;
        return (mx == my) ? V3(T.color0[0], T.color0[1], T.color0[2]) : V3(T.color1[0], T.color1[1], T.color1[2]);
    }
    return V3(0.f);
}

/* src/textures/bitmap.cpp:509-578 eval_1_grad (bilinear, repeat wrap; texel fetch
   order of Dr.Jit Texture::eval_fetch: (x0,y0),(x1,y0),(x0,y1),(x1,y1)) */
static V2 tex_eval_1_grad(const Scene &S, int tex, const SI &si) {
    const lrt_texture_desc &T = S.textures[tex];
    if (T.type != LRT_TEX_BITMAP) return { 0.f, 0.f };
    float u = fmaf(T.to_uv[1], si.uv.y, fmaf(T.to_uv[0], si.uv.x, T.to_uv[2]));
    float v = fmaf(T.to_uv[4], si.uv.y, fmaf(T.to_uv[3], si.uv.x, T.to_uv[5]));
    int w = T.width, h = T.height;
    float fx = fmaf(u, (float) w, -0.5f), fy = fmaf(v, (float) h, -0.5f);
    int ix = (int) floorf(fx), iy = (int) floorf(fy);
    float w1x = fx - (float) ix, w1y = fy - (float) iy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    auto wrap = [](int i, int n) { int r = i % n; return r < 0 ? r + n : r; };
    int x0 = wrap(ix, w), x1 = wrap(ix + 1, w), y0 = wrap(iy, h), y1 = wrap(iy + 1, h);
    auto fetch = [&](int x, int y) {
        const float *p = &T.data[((size_t) y * w + x) * T.channels];
        return T.channels == 1 ? p[0] : luminance(V3(p[0], p[1], p[2]));
    };
    float f00 = fetch(x0, y0), f10 = fetch(x1, y0), f01 = fetch(x0, y1), f11 = fetch(x1, y1);
    float dx = fmaf(w0y, f10 - f00, w1y * (f11 - f01)), dy = fmaf(w0x, f01 - f00, w1x * (f11 - f10));
    float du = T.to_uv[0] * dx + T.to_uv[3] * dy, dv = T.to_uv[1] * dx + T.to_uv[4] * dy;
    return { (float) w * du, (float) h * dv };
}

/* ----------------------------------------------------------------- BSDFs */
enum { F_DELTA = 1, F_SMOOTH = 2, F_NULL = 4 };   /* subset of BSDFFlags used by the integrators */
struct BSDFSample { V3 wo; float pdf, eta; int type; };

static int bsdf_flags(const Scene &S, int b) {
    const lrt_bsdf_desc &B = S.bsdfs[b];
    switch (B.type) {
        case LRT_BSDF_DIFFUSE: return F_SMOOTH;
        case LRT_BSDF_DIELECTRIC: return F_DELTA;
        case LRT_BSDF_BUMPMAP: return bsdf_flags(S, B.nested);
        default: return F_NULL;
    }
}

/* include/mitsuba/render/fresnel.h:35-73 */
static void fresnel(float cos_theta_i, float eta, float *r, float *cos_theta_t, float *eta_it, float *eta_ti) {
    bool outside = cos_theta_i >= 0.f;
    float rcp_eta = rcp(eta);
    *eta_it = outside ? eta : rcp_eta; *eta_ti = outside ? rcp_eta : eta;
    float cos_theta_t_sqr = fmaf(-fmaf(-cos_theta_i, cos_theta_i, 1.f), *eta_ti * *eta_ti, 1.f);
    float cti = fabsf(cos_theta_i), ctt = safe_sqrt(cos_theta_t_sqr);
    bool index_matched = eta == 1.f, special = index_matched || cti == 0.f;
    float r_sc = index_matched ? 0.f : 1.f;
    float a_s = fmaf(-*eta_it, ctt, cti) / fmaf(*eta_it, ctt, cti);
    float a_p = fmaf(-*eta_it, cti, ctt) / fmaf(*eta_it, cti, ctt);
    float rr = 0.5f * (sqr(a_s) + sqr(a_p));
    if (special) rr = r_sc;
    *r = rr; *cos_theta_t = mulsign_neg(ctt, cos_theta_i);
}

static void bsdf_sample(const Scene &S, int b, const SI &si, float s1, float s2x, float s2y, BSDFSample *bs, V3 *weight);
static V3 bsdf_eval(const Scene &S, int b, const SI &si, V3 wo);
static float bsdf_pdf(const Scene &S, int b, const SI &si, V3 wo);

/* src/bsdfs/bumpmap.cpp:226-251 frame() */
static Frame bump_frame(const Scene &S, const lrt_bsdf_desc &B, const SI &si) {
    V2 g = tex_eval_1_grad(S, B.texture, si);
    float gx = B.scale * g.x, gy = B.scale * g.y;
    V3 dp_du = fma3(si.sh.n, gx - dot(si.sh.n, si.dp_du), si.dp_du);
    V3 dp_dv = fma3(si.sh.n, gy - dot(si.sh.n, si.dp_dv), si.dp_dv);
    Frame r;
    r.n = normalize(cross(dp_du, dp_dv));
    if (dot(si.n, r.n) < 0.f) r.n = r.n * -1.f;
    r.n = si.sh.to_local(r.n);
    if (si.wi.z * dot(si.wi, r.n) <= 0.f) r.n = V3(-r.n.x, -r.n.y, r.n.z);   /* flip_invalid_normals */
    r.s = normalize(fma3(r.n, -dot(r.n, si.dp_du), si.dp_du));
    r.t = cross(r.n, r.s);
    return r;
}
/* include/mitsuba/core/frame.h:80-83, src/bsdfs/normalmap_helpers.h:20-25 */
static float tan_theta_2(V3 v) { float t = fmaf(-v.z, v.z, 1.f); return fmaxf(t, 0.f) / sqr(v.z); }
static float shadow_terminator(V3 pn, V3 wo) {
    float alpha2 = fminf(0.125f * tan_theta_2(pn), 1.f);
    return 2.f / (1.f + sqrtf(1.f + alpha2 * tan_theta_2(wo)));
}

static void bsdf_sample(const Scene &S, int b, const SI &si, float s1, float s2x, float s2y, BSDFSample *bs, V3 *weight) {
    const lrt_bsdf_desc &B = S.bsdfs[b];
    bs->wo = V3(0.f); bs->pdf = 0.f; bs->eta = 0.f; bs->type = 0; *weight = V3(0.f);   /* dr::zeros<BSDFSample3f> */
    switch (B.type) {
        case LRT_BSDF_DIFFUSE: {                 /* src/bsdfs/diffuse.cpp sample() */
            float cti = si.wi.z;
            if (!(cti > 0.f)) return;
            bs->wo = square_to_cosine_hemisphere(s2x, s2y);
            bs->pdf = kInvPi * bs->wo.z; bs->eta = 1.f; bs->type = F_SMOOTH;
            if (bs->pdf > 0.f) *weight = tex_eval(S, B.reflectance, si);
            return;
        }
        case LRT_BSDF_DIELECTRIC: {              /* src/bsdfs/dielectric.cpp:230-367 */
            float r_i, ctt, eta_it, eta_ti;
            fresnel(si.wi.z, B.eta, &r_i, &ctt, &eta_it, &eta_ti);
            float t_i = 1.f - r_i;
            bool sel_r = s1 <= r_i;
            bs->pdf = sel_r ? r_i : t_i;
            bs->type = F_DELTA;
            bs->wo = sel_r ? V3(-si.wi.x, -si.wi.y, si.wi.z) : V3(-eta_ti * si.wi.x, -eta_ti * si.wi.y, ctt);
            bs->eta = sel_r ? 1.f : eta_it;
            *weight = V3(1.f);
            if (!sel_r) *weight = *weight * sqr(eta_ti);
            return;
        }
        case LRT_BSDF_BUMPMAP: {                 /* src/bsdfs/bumpmap.cpp:138-162 */
            SI psi = si;
            psi.sh = bump_frame(S, B, si);
            psi.wi = psi.sh.to_local(si.wi);
            V3 w; bsdf_sample(S, B.nested, psi, s1, s2x, s2y, bs, &w);
            bool active = any_nonzero(w);
            V3 pwo = psi.sh.to_world(bs->wo);
            active = active && (bs->wo.z * pwo.z > 0.f);
            bs->wo = pwo;
            w = w * shadow_terminator(psi.sh.n, bs->wo);
            *weight = active ? w : V3(0.f);
            return;
        }
        default: {                               /* src/bsdfs/null.cpp sample() */
            bs->wo = -si.wi; bs->type = F_NULL; bs->eta = 1.f; bs->pdf = 1.f; *weight = V3(1.f);
            return;
        }
    }
}

static V3 bsdf_eval(const Scene &S, int b, const SI &si, V3 wo) {
    const lrt_bsdf_desc &B = S.bsdfs[b];
    if (B.type == LRT_BSDF_DIFFUSE) {             /* src/bsdfs/diffuse.cpp eval() */
        if (!(si.wi.z > 0.f && wo.z > 0.f)) return V3(0.f);
        return tex_eval(S, B.reflectance, si) * kInvPi * wo.z;
    }
    if (B.type == LRT_BSDF_BUMPMAP) {             /* src/bsdfs/bumpmap.cpp:164-183 */
        SI psi = si; psi.sh = bump_frame(S, B, si); psi.wi = psi.sh.to_local(si.wi);
        V3 pwo = psi.sh.to_local(wo);
        if (!(wo.z * pwo.z > 0.f)) return V3(0.f);
        return bsdf_eval(S, B.nested, psi, pwo) * shadow_terminator(psi.sh.n, wo);
    }
    return V3(0.f);
}
static float bsdf_pdf(const Scene &S, int b, const SI &si, V3 wo) {
    const lrt_bsdf_desc &B = S.bsdfs[b];
    if (B.type == LRT_BSDF_DIFFUSE)
        return (si.wi.z > 0.f && wo.z > 0.f) ? kInvPi * wo.z : 0.f;
    if (B.type == LRT_BSDF_BUMPMAP) {
        SI psi = si; psi.sh = bump_frame(S, B, si); psi.wi = psi.sh.to_local(si.wi);
        V3 pwo = psi.sh.to_local(wo);
        if (!(wo.z * pwo.z > 0.f)) return 0.f;
        return bsdf_pdf(S, B.nested, psi, pwo);
    }
    return 0.f;
}
/* src/render/bsdf.cpp:33-36, src/bsdfs/null.cpp eval_null_transmission */
static float bsdf_null_transmission(const Scene &S, int b) {
    const lrt_bsdf_desc &B = S.bsdfs[b];
    if (B.type == LRT_BSDF_BUMPMAP) return 0.f;   /* BumpMap does not forward it: base-class 0 */
    return B.type == LRT_BSDF_NULL ? 1.f : 0.f;
}

/* -------------------------------------------------------------- emitters */
struct DirSample { V3 p, n, d; float pdf, dist; bool delta; int emitter; };

/* src/emitters/envmap.cpp:528-560 eval_spectrum (RGB) */
static V3 env_eval_uv(const Scene &S, float u, float v) {
    const lrt_emitter_desc &E = S.emitters[S.env];
    uint32_t rx = S.env_w, ry = S.env_h;
    u -= .5f / (float) (rx - 1u);
    u -= floorf(u); v -= floorf(v);
    u *= (float) (rx - 1u); v *= (float) (ry - 1u);
    uint32_t px = std::min((uint32_t) u, rx - 2u), py = std::min((uint32_t) v, ry - 2u);
    float w1x = u - (float) px, w1y = v - (float) py, w0x = 1.f - w1x, w0y = 1.f - w1y;
    uint32_t idx = py * rx + px;
    auto px3 = [&](uint32_t i) { return V3(S.env_data[3 * i], S.env_data[3 * i + 1], S.env_data[3 * i + 2]); };
    V3 v00 = px3(idx), v10 = px3(idx + 1), v01 = px3(idx + rx), v11 = px3(idx + rx + 1);
    auto f3 = [](float a, V3 x, V3 y) { return V3(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z)); };
    V3 v0 = f3(w0x, v00, v10 * w1x), v1 = f3(w0x, v01, v11 * w1x), vv = f3(w0y, v0, v1 * w1y);
    return vv * E.scale;
}
static V3 emitter_eval_env(const Scene &S, V3 dir_world) {   /* dir_world = -si.wi of an invalid si */
    const lrt_emitter_desc &E = S.emitters[S.env];
    if (E.type == LRT_EMITTER_CONSTANT) return V3(E.radiance[0], E.radiance[1], E.radiance[2]);
    /* src/emitters/envmap.cpp:353-362 eval */
    V3 v = xform_vec(S.env_to_local, dir_world);
    float uu = m_atan2(v.x, -v.z) * kInvTwoPi, vv = safe_acos(v.y) * kInvPi;
    return env_eval_uv(S, uu, vv);
}
/* src/emitters/area.cpp eval(): radiance & (cos_theta(wi) > 0) */
static V3 emitter_eval_area(const Scene &S, int e, const SI &si) {
    const lrt_emitter_desc &E = S.emitters[e];
    return (si.wi.z > 0.f) ? V3(E.radiance[0], E.radiance[1], E.radiance[2]) : V3(0.f);
}

/* Scene::sample_emitter_direction (src/render/scene.cpp:333-383) without the
   visibility test; returns the emitter weight (radiance / pdf). */
static V3 sample_emitter_direction(const Scene &S, V3 ref_p, float sx, float sy, DirSample *ds) {
    uint32_t ne = S.d.n_emitters;
    memset((void *) ds, 0, sizeof(*ds)); ds->emitter = -1;
    if (ne == 0) return V3(0.f);
    uint32_t index = 0; float emitter_weight = 1.f, pmf = 1.f;
    if (ne > 1) {                                 /* scene.cpp:265-288 */
        float scaled = sx * (float) ne;
        index = std::min((uint32_t) scaled, ne - 1u);
        emitter_weight = (float) ne; sx = scaled - (float) index; pmf = 1.f / (float) ne;
    }
    const lrt_emitter_desc &E = S.emitters[index];
    ds->emitter = (int) index; ds->delta = false;
    V3 spec(0.f);
    if (E.type == LRT_EMITTER_AREA) {
        /* src/shapes/rectangle.cpp:181-199 + src/render/shape.cpp:343-361 + src/emitters/area.cpp sample_direction */
        const lrt_shape_desc &sd = S.shapes[E.shape];
        M4 tw; memcpy(tw.m, sd.to_world, sizeof(tw.m));
        ds->p = xform_point(tw, V3(fmaf(sx, 2.f, -1.f), fmaf(sy, 2.f, -1.f), 0.f));
        ds->n = S.area[index].n; if (sd.flip_normals) ds->n = -ds->n;
        ds->pdf = S.area[index].inv_area;
        ds->d = ds->p - ref_p;
        float dist2 = squared_norm(ds->d);
        ds->dist = sqrtf(dist2);
        ds->d = ds->d / ds->dist;
        float dp = fabsf(dot(ds->d, ds->n)), x = dist2 / dp;
        ds->pdf *= std::isfinite(x) ? x : 0.f;
        bool active = dot(ds->d, ds->n) < 0.f && ds->pdf != 0.f;
        V3 rad(E.radiance[0], E.radiance[1], E.radiance[2]);
        spec = active ? rad / ds->pdf : V3(0.f);
    } else if (E.type == LRT_EMITTER_ENVMAP) {
        /* src/emitters/envmap.cpp:415-459 */
        float u, v, pdf; S.env_warp.sample(sx, sy, &u, &v, &pdf);
        u += .5f / (float) (S.env_w - 1u);
        bool active = pdf > 0.f;
        float theta = v * kPi, phi = u * kTwoPi;
        float st, ct, sp, cp; m_sincos(theta, &st, &ct); m_sincos(phi, &sp, &cp);
        V3 d(cp * st, sp * st, ct);                /* dr::sphdir */
        d = V3(d.y, d.z, -d.x);
        float radius = fmaxf(S.bsphere_r, norm(ref_p - S.bsphere_c)), dist = 2.f * radius;
        float inv_sin_theta = safe_rsqrt(fmaxf(sqr(d.x) + sqr(d.z), sqr(kEpsilon)));
        d = xform_vec(S.env_to_world, d);
        ds->p = ref_p + d * dist; ds->n = -d;
        ds->pdf = active ? pdf * inv_sin_theta * (1.f / (2.f * sqr(kPi))) : 0.f;
        ds->d = d; ds->dist = dist;
        V3 val = env_eval_uv(S, u, v);
        spec = active ? val / ds->pdf : V3(0.f);
    } else {
        /* src/emitters/constant.cpp sample_direction */
        V3 d = square_to_uniform_sphere(sx, sy);
        float radius = fmaxf(S.bsphere_r, norm(ref_p - S.bsphere_c)), dist = 2.f * radius;
        ds->p = fma3(d, dist, ref_p); ds->n = -d; ds->pdf = kInvFourPi; ds->d = d; ds->dist = dist;
        spec = V3(E.radiance[0], E.radiance[1], E.radiance[2]) / ds->pdf;
    }
    ds->pdf *= pmf;
    spec = spec * emitter_weight;
    return spec;
}

/* DirectionSample(scene, si, ref) (include/mitsuba/render/records.h:173-180) +
   Scene::pdf_emitter_direction (src/render/scene.cpp:395-406) */
static float pdf_emitter_direction(const Scene &S, V3 ref_p, const SI &si, int emitter) {
    V3 rel = si.p - ref_p;
    float dist = norm(rel);
    V3 d = si.valid ? rel / dist : -si.wi;
    float pmf = 1.f / (float) S.d.n_emitters;
    const lrt_emitter_desc &E = S.emitters[emitter];
    float value;
    if (E.type == LRT_EMITTER_AREA) {             /* src/emitters/area.cpp pdf_direction + shape.cpp:363-374 */
        float dp = dot(d, si.n);
        if (!(dp < 0.f)) return 0.f;
        float adp = fabsf(dp);
        value = S.area[emitter].inv_area * (adp != 0.f ? (dist * dist) / adp : 0.f);
    } else if (E.type == LRT_EMITTER_ENVMAP) {    /* src/emitters/envmap.cpp:461-475 */
        V3 dl = xform_vec(S.env_to_local, d);
        float u = m_atan2(dl.x, -dl.z) * kInvTwoPi, v = safe_acos(dl.y) * kInvPi;
        u -= .5f / (float) (S.env_w - 1u);
        u -= floorf(u); v -= floorf(v);
        float inv_sin_theta = safe_rsqrt(fmaxf(sqr(dl.x) + sqr(dl.z), sqr(kEpsilon)));
        value = S.env_warp.eval(u, v) * inv_sin_theta * (1.f / (2.f * sqr(kPi)));
    } else value = kInvFourPi;
    return value * pmf;
}

static inline int si_emitter(const Scene &S, const SI &si) { return si.valid ? S.shapes[si.shape].emitter : S.env; }
static inline V3 emitter_eval(const Scene &S, int e, const SI &si) {
    return si.valid ? emitter_eval_area(S, e, si) : emitter_eval_env(S, -si.wi);
}

static inline float mis_weight(float a, float b) { a *= a; b *= b; float w = a / (a + b); return std::isfinite(w) ? w : 0.f; }
static inline float idx3(V3 v, uint32_t c) { return c == 0 ? v.x : (c == 1 ? v.y : v.z); }

/* ---------------------------------------------------------------- media */
struct MI { float t; V3 p, wi; V3 sigma_s, sigma_n, sigma_t, combined; float mint; int medium;
            bool valid() const { return t != kInf; } };

/* src/volumes/grid.cpp interpolate_1 through Dr.Jit's Texture3f::eval (un-vendored; trilinear, clamp): texel centres at
   (i + .5) / res; lerp along x, then y, then z, each as fmadd(w0, a, w1 * b).  p_local in the grid's unit cube. */
static float grid_eval(const lrt_medium_desc &M, V3 pl) {
    const int rx = M.grid_res[0], ry = M.grid_res[1], rz = M.grid_res[2];
    float fx = fmaf(pl.x, (float) rx, -.5f), fy = fmaf(pl.y, (float) ry, -.5f), fz = fmaf(pl.z, (float) rz, -.5f);
    float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
    float w1x = fx - flx, w1y = fy - fly, w1z = fz - flz, w0x = 1.f - w1x, w0y = 1.f - w1y, w0z = 1.f - w1z;
    auto cl = [](float v, int n) { int i = (v < -2e9f) ? -2000000000 : (v > 2e9f ? 2000000000 : (int) v); return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); };
    int x0 = cl(flx, rx), x1 = cl(flx + 1.f, rx), y0 = cl(fly, ry), y1 = cl(fly + 1.f, ry), z0 = cl(flz, rz), z1 = cl(flz + 1.f, rz);
    auto at = [&](int x, int y, int z) { return M.grid_data[((size_t) z * ry + y) * rx + x]; };
    float c00 = fmaf(w0x, at(x0, y0, z0), w1x * at(x1, y0, z0)), c10 = fmaf(w0x, at(x0, y1, z0), w1x * at(x1, y1, z0));
    float c01 = fmaf(w0x, at(x0, y0, z1), w1x * at(x1, y0, z1)), c11 = fmaf(w0x, at(x0, y1, z1), w1x * at(x1, y1, z1));
    float c0 = fmaf(w0y, c00, w1y * c10), c1 = fmaf(w0y, c01, w1y * c11);
    return fmaf(w0z, c0, w1z * c1);
}

/* include/mitsuba/core/bbox.h:303-340 ray_intersect (Williams et al.) */
static bool bbox_ray_intersect(const float lo[3], const float hi[3], const Ray &ray, float *mint, float *maxt) {
    bool active = ray.d.x != 0.f || ray.d.y != 0.f || ray.d.z != 0.f;
    float dr[3] = { rcp(ray.d.x), rcp(ray.d.y), rcp(ray.d.z) }, o[3] = { ray.o.x, ray.o.y, ray.o.z }, tmin[3], tmax[3];
    for (int a = 0; a < 3; ++a) { bool pos = dr[a] >= 0.f; tmin[a] = ((pos ? lo[a] : hi[a]) - o[a]) * dr[a]; tmax[a] = ((pos ? hi[a] : lo[a]) - o[a]) * dr[a]; }
    auto max_safe = [](float a, float b) { return (a > b || !std::isfinite(b)) ? a : b; };
    auto min_safe = [](float a, float b) { return (a < b || !std::isfinite(b)) ? a : b; };
    active = active && !((tmin[0] > tmax[1]) || (tmin[1] > tmax[0]));
    tmin[0] = max_safe(tmin[0], tmin[1]); tmax[0] = min_safe(tmax[0], tmax[1]);
    active = active && !((tmin[0] > tmax[2]) || (tmin[2] > tmax[0]));
    tmin[0] = max_safe(tmin[0], tmin[2]); tmax[0] = min_safe(tmax[0], tmax[2]);
    *mint = tmin[0]; *maxt = tmax[0];
    return active;
}

static inline bool medium_is_homogeneous(const lrt_medium_desc &M) { return M.type != LRT_MEDIUM_HETEROGENEOUS; }

/* src/render/medium.cpp:40-82 with src/media/homogeneous.cpp:153-181 or src/media/heterogeneous.cpp:178-200 */
static MI medium_sample_interaction(const Scene &S, int m, const Ray &ray, float sample, uint32_t channel) {
    const lrt_medium_desc &M = S.media[m];
    MI mei; mei.wi = -ray.d; mei.medium = m;
    V3 albedo(M.albedo[0], M.albedo[1], M.albedo[2]);
    if (M.type == LRT_MEDIUM_HETEROGENEOUS) {
        float mint, maxt;
        bool active = bbox_ray_intersect(M.grid_bbox_min, M.grid_bbox_max, ray, &mint, &maxt);
        active = active && (std::isfinite(mint) || std::isfinite(maxt));
        if (!active) { mint = 0.f; maxt = kInf; }
        mint = fmaxf(0.f, mint); maxt = fminf(ray.maxt, maxt);
        const float max_density = M.scale * M.grid_max;                           /* get_majorant: not masked */
        float sampled_t = mint + (-m_log(1.f - sample) / max_density);
        bool valid = active && sampled_t <= maxt;
        mei.t = valid ? sampled_t : kInf;
        mei.p = fma3(ray.d, sampled_t, ray.o);
        mei.mint = mint;
        float st = 0.f;
        if (valid) {
            const float *t = M.grid_to_local;
            V3 pl(fmaf(t[2], mei.p.z, fmaf(t[1], mei.p.y, fmaf(t[0], mei.p.x, t[3]))), fmaf(t[6], mei.p.z, fmaf(t[5], mei.p.y, fmaf(t[4], mei.p.x, t[7]))),
                  fmaf(t[10], mei.p.z, fmaf(t[9], mei.p.y, fmaf(t[8], mei.p.x, t[11]))));
            st = M.scale * grid_eval(M, pl);
        }
        mei.sigma_t = V3(st); mei.sigma_s = mei.sigma_t * (valid ? albedo : V3(0.f));
        mei.sigma_n = V3(max_density) - mei.sigma_t;
        mei.combined = V3(max_density);
        return mei;
    }
    float mint = 0.f, maxt = fminf(ray.maxt, kInf);
    V3 sigmat = V3(M.sigma_t[0], M.sigma_t[1], M.sigma_t[2]) * M.scale;
    float mm = idx3(sigmat, channel);
    float sampled_t = mint + (-m_log(1.f - sample) / mm);
    bool valid = sampled_t <= maxt;
    mei.t = valid ? sampled_t : kInf;
    mei.p = fma3(ray.d, sampled_t, ray.o);
    mei.mint = mint;
    mei.sigma_t = valid ? sigmat : V3(0.f);
    mei.sigma_s = valid ? sigmat * albedo : V3(0.f);
    mei.sigma_n = V3(0.f);
    mei.combined = sigmat;
    return mei;
}

/* src/phase/hg.cpp:64-99, src/phase/isotropic.cpp:39-58 */
static float hg_eval(float g, float cos_theta) {
    float temp = 1.f + sqr(g) + 2.f * g * cos_theta;
    return kInvFourPi * (1.f - sqr(g)) / (temp * sqrtf(temp));
}
static void phase_sample(const lrt_medium_desc &M, V3 wi, float s2x, float s2y, V3 *wo, float *pdf) {
    if (M.phase == LRT_PHASE_HG) {
        float g = M.g;
        float sqr_term = (1.f - sqr(g)) / (1.f - g + 2.f * g * s2x);
        float cos_theta = (1.f + sqr(g) - sqr(sqr_term)) / (2.f * g);
        if (fabsf(g) < kEpsilon) cos_theta = 1.f - 2.f * s2x;
        float sin_theta = safe_sqrt(1.f - sqr(cos_theta));
        float sp, cp; m_sincos(2.f * kPi * s2y, &sp, &cp);
        Frame f(wi);
        *wo = f.to_world(V3(sin_theta * cp, sin_theta * sp, -cos_theta));
        *pdf = hg_eval(g, -cos_theta);
    } else {
        *wo = square_to_uniform_sphere(s2x, s2y);
        *pdf = kInvFourPi;
    }
}
static float phase_eval(const lrt_medium_desc &M, V3 wi, V3 wo) {
    return M.phase == LRT_PHASE_HG ? hg_eval(M.g, dot(wo, wi)) : kInvFourPi;
}

/* ------------------------------------------------------------ integrators */
struct Ctx {
    const Scene &S; Sampler smp; int max_depth, rr_depth; bool hide_emitters;
    bool stream_continues = false;  /* a later pass resumes this lane's PCG32 stream (multi-pass renders, independent sampler) */
    int grad_medium = 0;            /* PRB adjoint: medium whose parameters are differentiated, -1: all media into one set */
    bool bio_jit = true;            /* bio transport: JIT-variant reading (orc_bio.h); false: scalar_rgb reading */
    uint64_t n_iter = 0, n_shadow = 0, n_shadow_needed = 0;
    Ctx(const Scene &s) : S(s), bio_jit(!s.bio_scalar) {}
    float next() { return smp.next1(); }
    void next2(float *x, float *y) { smp.next2(x, y); }
    void skip(uint32_t n) { if (bio_jit) smp.skip(n); }   /* scalar code only executes the calls it reaches */
};

static inline int target_medium(const lrt_shape_desc &sd, V3 d, V3 n) {   /* interaction.h:330-344 */
    return dot(d, n) > 0.f ? sd.exterior_medium : sd.interior_medium;
}
static inline bool is_medium_transition(const lrt_shape_desc &sd) { return sd.interior_medium >= 0 || sd.exterior_medium >= 0; }

/* src/integrators/volpath.cpp:400-554.  ref_n is zero for medium interactions. */
static V3 volpath_sample_emitter(Ctx &C, V3 ref_p, V3 ref_n, const SI *ref_si, int medium, uint32_t channel, DirSample *ds_out) {
    const Scene &S = C.S;
    V3 transmittance(1.f);
    float sx, sy; C.next2(&sx, &sy);
    DirSample ds; V3 emitter_val = sample_emitter_direction(S, ref_p, sx, sy, &ds);
    *ds_out = ds;
    if (ds.pdf == 0.f) return V3(0.f);
    Ray ray = spawn_ray_to(ref_p, ref_n, ds.p);
    float max_dist = ray.maxt;
    if (ref_si && is_medium_transition(S.shapes[ref_si->shape]))
        medium = target_medium(S.shapes[ref_si->shape], ray.d, ref_si->n);
    float total_dist = 0.f;
    SI si; memset((void *) &si, 0, sizeof(si));
    bool needs_intersection = true, active = true;
    while (active) {
        float remaining_dist = max_dist - total_dist;
        ray.maxt = remaining_dist;
        if (!(remaining_dist > 0.f)) { C.skip(1); break; }          /* the body still runs (masked) in this last trip */
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (!active_medium) C.skip(1);                                /* volpath.cpp:479 */
        if (active_medium) {
            const lrt_medium_desc &M = S.media[medium];
            MI mei = medium_sample_interaction(S, medium, ray, C.next(), channel);
            if (mei.valid() && medium_is_homogeneous(M)) ray.maxt = fminf(mei.t, remaining_dist);
            bool elide = false;
            if (needs_intersection) {
                /* the query cannot change the result when the sampled collision is real
                   (sigma_n = 0) and every surface blocks (no null BSDF): both outcomes give 0 */
                elide = mei.valid() && !S.has_null_bsdf && medium_is_homogeneous(M);
                C.n_shadow++; if (!elide) C.n_shadow_needed++;
                Hit h = S.intersect(ray, false, false);
                si = S.compute_si(ray, h);
            }
            if (si.t < mei.t) mei.t = kInf;
            needs_intersection = false;
            bool spectral = M.has_spectral_extinction;
            if (spectral) {
                float t = fminf(remaining_dist, fminf(mei.t, si.t)) - mei.mint;
                V3 tr(m_exp(-t * mei.combined.x), m_exp(-t * mei.combined.y), m_exp(-t * mei.combined.z));
                V3 ffp = (si.t < mei.t || mei.t > remaining_dist) ? tr : tr * mei.combined;
                float tr_pdf = idx3(ffp, channel);
                transmittance *= (tr_pdf > 0.f) ? tr / tr_pdf : V3(0.f);
            }
            if ((mei.t > remaining_dist) && mei.valid()) total_dist = ds.dist;
            if (mei.t > remaining_dist) mei.t = kInf;
            escaped_medium = !mei.valid();
            active_medium = mei.valid();
            if (active_medium) {
                total_dist += mei.t;
                ray.o = mei.p;
                si.t = si.t - mei.t;
                if (spectral) transmittance *= mei.sigma_n;
                else transmittance *= mei.sigma_n / mei.combined;
            }
        }
        bool intersect = active_surface && needs_intersection;
        if (intersect) {
            C.n_shadow++; C.n_shadow_needed++;
            Hit h = S.intersect(ray, false, false);
            si = S.compute_si(ray, h);
            needs_intersection = false;
        }
        active_surface = active_surface || escaped_medium;
        if (active_surface) total_dist += si.t;
        active_surface = active_surface && si.valid && !active_medium;
        if (active_surface) {
            int b = S.shapes[si.shape].bsdf;
            transmittance *= bsdf_null_transmission(S, b);
            Ray nr = spawn_ray(si.p, si.n, ray.d);
            ray = nr;
        }
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        active = (active_medium || active_surface) && any_nonzero(transmittance);
        if (active_surface && is_medium_transition(S.shapes[si.shape]))
            medium = target_medium(S.shapes[si.shape], ray.d, si.n);
    }
    return transmittance * emitter_val;
}

/* src/integrators/volpath.cpp:93-396 */
static void volpath_sample(Ctx &C, Ray ray, int medium, V3 *out, bool *out_valid) {
    const Scene &S = C.S;
    bool valid_ray = !C.hide_emitters && S.env >= 0;
    float eta = 1.f;
    V3 throughput(1.f), result(0.f);
    bool specular_chain = !C.hide_emitters;
    uint32_t depth = 0;
    uint32_t channel = std::min((uint32_t) (C.next() * 3.f), 2u);
    SI si; memset((void *) &si, 0, sizeof(si));
    bool needs_intersection = true, active = true;
    V3 last_scatter_p(0.f);
    float last_scatter_direction_pdf = 1.f;
    const uint32_t max_depth = (uint32_t) C.max_depth;
    static const long trace_after = getenv("ORC_TRACE_AFTER") ? atol(getenv("ORC_TRACE_AFTER")) : -1;      /* developer aid: print a path's trips beyond this count */
    uint64_t trips = 0;
    while (active) {
        C.n_iter++; ++trips;
        if (trace_after >= 0 && (long) trips > trace_after && (long) trips <= trace_after + 12)
            fprintf(stderr, "[orc trace] trip %llu depth %u medium %d tp %g %g %g eta %g o %g %g %g d %g %g %g maxt %g si.t %g needs_isect %d\n", (unsigned long long) trips, depth, medium,
                    throughput.x, throughput.y, throughput.z, eta, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, ray.maxt, si.t, (int) needs_intersection);
        active = any_nonzero(throughput);
        float q = fminf(max3(throughput) * sqr(eta), .95f);
        bool perform_rr = depth > (uint32_t) C.rr_depth;
        if (active) { float u = C.next(); active = (u < q) || !perform_rr; }
        if (perform_rr) throughput *= rcp(q);
        active = active && depth < max_depth;
        if (!active) break;

        bool active_medium = medium >= 0, active_surface = !active_medium;
        bool act_medium_scatter = false, escaped_medium = false, act_null_scatter = false;
        MI mei; mei.t = kInf;
        if (!active_medium) C.skip(2);                                /* volpath.cpp:220,239 */
        if (active_medium) {
            const lrt_medium_desc &M = S.media[medium];
            mei = medium_sample_interaction(S, medium, ray, C.next(), channel);
            if (mei.valid() && medium_is_homogeneous(M)) ray.maxt = mei.t;
            if (needs_intersection) { Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
            needs_intersection = false;
            if (si.t < mei.t) mei.t = kInf;
            if (M.has_spectral_extinction) {      /* medium.cpp:92-104 */
                float t = fminf(mei.t, si.t) - mei.mint;
                V3 tr(m_exp(-t * mei.combined.x), m_exp(-t * mei.combined.y), m_exp(-t * mei.combined.z));
                V3 pdf = (si.t < mei.t) ? tr : tr * mei.combined;
                float tr_pdf = idx3(pdf, channel);
                throughput *= (tr_pdf > 0.f) ? tr / tr_pdf : V3(0.f);
            }
            escaped_medium = !mei.valid();
            active_medium = mei.valid();
            if (!active_medium) C.skip(1);                            /* volpath.cpp:239 */
            if (active_medium) {
                float null_scatter_prob = mean3(mei.sigma_n / mei.combined);
                act_null_scatter = C.next() < null_scatter_prob;                  /* volpath.cpp:238-246; never true for sigma_n = 0 */
                act_medium_scatter = !act_null_scatter;
                if (M.has_spectral_extinction && act_null_scatter) throughput *= mei.sigma_n / null_scatter_prob;
                if (act_medium_scatter) { depth += 1; last_scatter_p = mei.p; }
            }
        }
        active = active && depth < max_depth;
        act_medium_scatter = act_medium_scatter && active;
        if (act_null_scatter) { ray.o = mei.p; si.t = si.t - mei.t; }            /* :254-257; the surface found earlier stays in `si` */
        if (!act_medium_scatter) C.skip(3);                           /* volpath.cpp:407 (NEE), 288, 289 */
        if (act_medium_scatter) {
            const lrt_medium_desc &M = S.media[medium];
            if (M.has_spectral_extinction) throughput *= mei.sigma_s / mean3(mei.sigma_t / mei.combined);
            else throughput *= mei.sigma_s / mei.sigma_t;
            bool sample_emitters = M.sample_emitters;
            valid_ray = true;
            specular_chain = !sample_emitters;
            if (!sample_emitters) C.skip(1);
            if (sample_emitters) {
                DirSample ds;
                V3 emitted = volpath_sample_emitter(C, mei.p, V3(0.f), nullptr, medium, channel, &ds);
                float phase_val = phase_eval(M, mei.wi, ds.d);
                result += throughput * phase_val * emitted * mis_weight(ds.pdf, ds.delta ? 0.f : phase_val);
            }
            float s1 = C.next(); (void) s1;
            float s2x, s2y; C.next2(&s2x, &s2y);
            V3 wo; float phase_pdf; phase_sample(M, mei.wi, s2x, s2y, &wo, &phase_pdf);
            act_medium_scatter = phase_pdf > 0.f;
            if (act_medium_scatter) {
                ray = spawn_ray(mei.p, V3(0.f), wo);
                needs_intersection = true;
                last_scatter_direction_pdf = phase_pdf;
                /* throughput *= phase_weight (= 1) */
            }
        }
        /* --------------------- surface interactions --------------------- */
        active_surface = active_surface || escaped_medium;
        bool intersect = active_surface && needs_intersection;
        if (intersect) { Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
        if (active_surface) {
            if (C.hide_emitters && depth == 0 && intersect) {      /* volpath.cpp:304-320, integrator.cpp:96-123 */
                bool skip = si.valid && S.shapes[si.shape].emitter >= 0;
                if (skip) {
                    Ray r2 = spawn_ray(si.p, si.n, ray.d);
                    bool a = true; Hit h; h.prim = 0xffffffffu; h.t = kInf;
                    while (a) {
                        h = S.intersect(r2, false, false);
                        a = h.valid() && S.shapes[S.face_shape[h.prim]].emitter >= 0;
                        if (a) { SI s2 = S.compute_si(r2, h); r2 = spawn_ray(s2.p, s2.n, r2.d); }
                    }
                    si = S.compute_si(r2, h);
                }
            }
            bool ray_from_camera = depth == 0;
            bool count_direct = ray_from_camera || specular_chain;
            int emitter = si_emitter(S, si);
            bool active_e = emitter >= 0 && !(depth == 0 && C.hide_emitters);
            if (active_e) {
                float emitter_pdf = 1.f;
                if (!count_direct) emitter_pdf = pdf_emitter_direction(S, last_scatter_p, si, emitter);
                V3 emitted = emitter_eval(S, emitter, si);
                V3 contrib = count_direct ? throughput * emitted
                                          : throughput * mis_weight(last_scatter_direction_pdf, emitter_pdf) * emitted;
                result += contrib;
            }
        }
        active_surface = active_surface && si.valid;
        if (!active_surface) C.skip(3);                               /* volpath.cpp:407 (NEE), 366, 367 */
        if (active_surface) {
            const lrt_shape_desc &sd = S.shapes[si.shape];
            int b = sd.bsdf;
            int flags = bsdf_flags(S, b);
            bool active_e = (flags & F_SMOOTH) && (depth + 1 < max_depth);
            if (!active_e) C.skip(1);
            if (active_e) {
                DirSample ds;
                V3 emitted = volpath_sample_emitter(C, si.p, si.n, &si, medium, channel, &ds);
                V3 wo = si.sh.to_local(ds.d);
                V3 bsdf_val = bsdf_eval(S, b, si, wo);
                float bpdf = bsdf_pdf(S, b, si, wo);
                result += throughput * bsdf_val * mis_weight(ds.pdf, ds.delta ? 0.f : bpdf) * emitted;
            }
            float s1 = C.next(), s2x, s2y; C.next2(&s2x, &s2y);
            BSDFSample bs; V3 bsdf_val;
            bsdf_sample(S, b, si, s1, s2x, s2y, &bs, &bsdf_val);
            throughput *= bsdf_val;
            eta *= bs.eta;
            ray = spawn_ray(si.p, si.n, si.sh.to_world(bs.wo));
            needs_intersection = true;
            bool non_null = !(bs.type & F_NULL);
            if (non_null) { depth += 1; last_scatter_p = si.p; last_scatter_direction_pdf = bs.pdf; valid_ray = true; }
            specular_chain = specular_chain || (non_null && (bs.type & F_DELTA));
            specular_chain = specular_chain && !(bs.type & F_SMOOTH);
            if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, si.n);
        }
        active = active && (active_surface || active_medium);
    }
    *out = result; *out_valid = valid_ray;
}

/* src/integrators/path.cpp:95-351 */
static void path_sample(Ctx &C, Ray ray, V3 *out, bool *out_valid) {
    const Scene &S = C.S;
    *out = V3(0.f); *out_valid = false;
    if (C.max_depth == 0) return;
    V3 throughput(1.f), result(0.f);
    float eta = 1.f; uint32_t depth = 0;
    bool valid_ray = !C.hide_emitters && S.env >= 0;
    V3 prev_p(0.f); float prev_bsdf_pdf = 1.f; bool prev_bsdf_delta = true;
    const uint32_t max_depth = (uint32_t) C.max_depth;
    Hit pi = S.intersect(ray, false, false);
    if (C.hide_emitters) {                        /* path.cpp:178-192 */
        bool skip = pi.valid() && S.shapes[S.face_shape[pi.prim]].emitter >= 0;
        if (skip) {
            SI s0 = S.compute_si(ray, pi);
            Ray r2 = spawn_ray(s0.p, s0.n, ray.d);
            bool a = true; Hit h; h.prim = 0xffffffffu; h.t = kInf;
            while (a) {
                h = S.intersect(r2, false, false);
                a = h.valid() && S.shapes[S.face_shape[h.prim]].emitter >= 0;
                if (a) { SI s2 = S.compute_si(r2, h); r2 = spawn_ray(s2.p, s2.n, r2.d); }
            }
            pi = h; ray = r2;   /* NB: reference keeps ls.ray; si is recomputed from pi with ls.ray */
        }
    }
    bool active = true;
    while (active) {
        C.n_iter++;
        SI si = S.compute_si(ray, pi);
        int emitter = si_emitter(S, si);
        if (emitter >= 0) {
            float em_pdf = 0.f;
            if (!prev_bsdf_delta) em_pdf = pdf_emitter_direction(S, prev_p, si, emitter);
            float mis_bsdf = mis_weight(prev_bsdf_pdf, em_pdf);
            V3 em = (prev_bsdf_pdf > 0.f) ? emitter_eval(S, emitter, si) : V3(0.f);
            em = em * mis_bsdf;
            result = V3(fmaf(throughput.x, em.x, result.x), fmaf(throughput.y, em.y, result.y), fmaf(throughput.z, em.z, result.z));
        }
        bool active_next = (depth + 1 < max_depth) && si.valid;
        if (!active_next) {
            /* path.cpp:227-231: `if (dr::none_or<false>(active_next))` is never taken in a JIT variant (its body, with the
               `valid_ray |= emitter && !hide_emitters` update, is scalar-only): the rest of the trip runs for the lane with
               active_em = false.  What survives: the six sampler values of :246, :266-267, :326 are drawn, and
               `valid_ray |= active && si.is_valid() && !Null` (:305-306; the sampled type is Null only for a null BSDF). */
            float a0, a1; C.next2(&a0, &a1); (void) C.next(); C.next2(&a0, &a1); (void) C.next();
            if (si.valid && !(bsdf_flags(S, S.shapes[si.shape].bsdf) & F_NULL)) valid_ray = true;
            break;
        }
        const lrt_shape_desc &sd = S.shapes[si.shape];
        int b = sd.bsdf;
        bool active_em = (bsdf_flags(S, b) & F_SMOOTH) != 0;
        DirSample ds; memset((void *) &ds, 0, sizeof(ds));
        V3 em_weight(0.f), wo(0.f);
        /* path.cpp:246-248: ls.sampler->next_2d() carries no mask and sits in an `if (dr::any_or<true>(active_em))`, which a
           symbolic loop always traces: every lane in the loop consumes the two numbers, smooth BSDF or not */
        float sx, sy; C.next2(&sx, &sy);
        if (active_em) {
            em_weight = sample_emitter_direction(S, si.p, sx, sy, &ds);
            if (ds.pdf != 0.f) {                  /* scene.cpp:361-365: test_visibility */
                Ray sr = spawn_ray_to(si.p, si.n, ds.p);
                C.n_shadow++; C.n_shadow_needed++;
                Hit h = S.intersect(sr, true, false);
                if (h.valid()) { em_weight = V3(0.f); ds.pdf = 0.f; }
            }
            active_em = ds.pdf != 0.f;
            wo = si.sh.to_local(ds.d);
        }
        float s1 = C.next(), s2x, s2y; C.next2(&s2x, &s2y);
        V3 bsdf_val = bsdf_eval(S, b, si, wo);
        float bpdf = bsdf_pdf(S, b, si, wo);
        BSDFSample bs; V3 bsdf_weight;
        bsdf_sample(S, b, si, s1, s2x, s2y, &bs, &bsdf_weight);
        if (active_em) {
            float mis_em = ds.delta ? 1.f : mis_weight(ds.pdf, bpdf);
            V3 c = bsdf_val * em_weight * mis_em;
            result = V3(fmaf(throughput.x, c.x, result.x), fmaf(throughput.y, c.y, result.y), fmaf(throughput.z, c.z, result.z));
        }
        ray = spawn_ray(si.p, si.n, si.sh.to_world(bs.wo));
        throughput *= bsdf_weight;
        eta *= bs.eta;
        valid_ray = valid_ray || !(bs.type & F_NULL);
        prev_p = si.p; prev_bsdf_pdf = bs.pdf; prev_bsdf_delta = (bs.type & F_DELTA) != 0;
        depth += 1;
        float tmax = max3(throughput);
        float rr_prob = fminf(tmax * sqr(eta), .95f);
        bool rr_active = depth >= (uint32_t) C.rr_depth, rr_continue = C.next() < rr_prob;
        if (rr_active) throughput *= rcp(rr_prob);
        active = active_next && (!rr_active || rr_continue) && (tmax != 0.f);
        if (active) pi = S.intersect(ray, false, false);
    }
    *out = valid_ray ? result : V3(0.f); *out_valid = valid_ray;
}

#include "orc_bio.h"
#include "orc_mis.h"

/* --------------------------------------------------- PRB (prbvolpath.py) */
/* ORC_PRB_DEBUG=1: per-trip terms of the adjoint on stderr (scripts/dbg/prb_lane_debug.py prints the device's beside them) */
static const bool g_prb_debug = getenv("ORC_PRB_DEBUG") != nullptr;
/* Gradient accumulators of one lane: d/d sigma_t[3] (w.r.t. the `sigma_t` property, i.e. before `scale`),
   d/d albedo[3], d/d g of medium 0..: the reference differentiates whatever parameters have gradients
   enabled; the oracle differentiates every medium's parameters into ONE set (scenes in scope have one medium). */
struct Grads { double sigma_t[3] = { 0, 0, 0 }, albedo[3] = { 0, 0, 0 }, g = 0;
               void add(const Grads &o) { for (int i = 0; i < 3; ++i) { sigma_t[i] += o.sigma_t[i]; albedo[i] += o.albedo[i]; } g += o.g; } };

/* d ln(hg)/dg: hg = (1-g^2) / (4 pi (1+g^2+2 g c)^(3/2)) (src/phase/hg.cpp:64-68) */
static inline float hg_dlog_dg(float g, float c) {
    float temp = 1.f + sqr(g) + 2.f * g * c;
    return -2.f * g / (1.f - sqr(g)) - 1.5f * (2.f * g + 2.f * c) / temp;
}

/* Gradient bookkeeping of a heterogeneous medium (src/media/heterogeneous.cpp): sigma_t(p) = scale * grid(p) carries the gradient,
   the majorant (`m_max_density`, an opaque scalar made in parameters_changed()) and with it the free-flight sampling and
   exp(-t * combined) do not.  lrt_param_grads::d_sigma_t[k] of such a medium is channel k's share of d/d(scale): their sum is the
   derivative w.r.t. the medium's `scale`.  d ln(sigma_s_k) / d scale = 1 / scale at a real collision,
   d ln(sigma_n_k) / d scale = -sigma_t(p) / (sigma_n_k scale) at a null collision (:178-196, :404-415). */
static inline float het_null_dlog_dscale(const lrt_medium_desc &M, float sigma_t, float sigma_n) {
    return sigma_n > 0.f ? -(sigma_t / sigma_n) / M.scale : 0.f;
}

/* src/python/python/ad/integrators/prbvolpath.py:354-444.  adjoint: backpropagates delta_L * adj_emitted
   through the per-segment transmittance: homogeneous media take their analytic transmittance in one step (:403-407,
   `nee_handle_homogeneous`), heterogeneous ones ratio tracking through null collisions (:409-415). */
static V3 prb_sample_emitter(Ctx &C, V3 ref_p, V3 ref_n, const SI *ref_si, int medium, uint32_t channel, DirSample *ds_out,
                             bool adjoint, V3 delta_L, V3 adj_emitted, Grads *G) {
    const int gm = C.grad_medium;
    const Scene &S = C.S;
    float sx, sy; C.next2(&sx, &sy);
    DirSample ds; V3 emitter_val = sample_emitter_direction(S, ref_p, sx, sy, &ds);
    *ds_out = ds;
    bool active = ds.pdf != 0.f;
    if (!active) { emitter_val = V3(0.f); medium = -1; }
    if (ref_si && is_medium_transition(S.shapes[ref_si->shape])) medium = target_medium(S.shapes[ref_si->shape], ds.d, ref_si->n);
    Ray ray = spawn_ray_to(ref_p, ref_n, ds.p);
    float max_dist = ray.maxt, total_dist = 0.f;
    SI si; memset((void *) &si, 0, sizeof(si));
    bool needs_intersection = true;
    V3 transmittance(1.f);
    while (active) {
        float remaining_dist = max_dist - total_dist;
        ray.maxt = remaining_dist;
        active = active && remaining_dist > 0.f;
        needs_intersection = needs_intersection && active;
        if (needs_intersection) { C.n_shadow++; C.n_shadow_needed++; Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
        needs_intersection = false;
        bool active_medium = active && medium >= 0, active_surface = active && !active_medium;
        V3 tr_multiplier(1.f);
        float seg_t = 0.f, mei_t = kInf; bool escaped_medium = false, homogeneous_segment = false;
        float null_dlog[3] = { 0.f, 0.f, 0.f };
        if (!active_medium) C.skip(1);               /* prbvolpath.py:396: the call runs for every lane in the march */
        if (active_medium) {
            const lrt_medium_desc &M = S.media[medium];
            MI mei = medium_sample_interaction(S, medium, ray, C.next(), channel);
            if (si.t < mei.t) mei.t = kInf;
            if (S.prb_nee_handle_homogeneous && medium_is_homogeneous(M)) {         /* :403-407: straight to the next surface / the end of the segment */
                V3 sigmat = V3(M.sigma_t[0], M.sigma_t[1], M.sigma_t[2]) * M.scale;
                float t = fminf(remaining_dist, si.t);   /* mei.t = min(remaining, si.t); tr = exp(-(min(mei.t, si.t) - mint) sigma) */
                seg_t = fminf(t, si.t) - 0.f;
                tr_multiplier = V3(m_exp(-seg_t * sigmat.x), m_exp(-seg_t * sigmat.y), m_exp(-seg_t * sigmat.z));
                mei.t = kInf; homogeneous_segment = true;
            }
            escaped_medium = !mei.valid();
            active_medium = mei.valid();
            if (active_medium) {                     /* ratio tracking: a (null) collision inside the segment (:409-415) */
                ray.o = mei.p; si.t = si.t - mei.t; mei_t = mei.t;
                tr_multiplier *= mei.sigma_n / mei.combined;
                float sn[3] = { mei.sigma_n.x, mei.sigma_n.y, mei.sigma_n.z };
                for (int k = 0; k < 3; ++k) null_dlog[k] = het_null_dlog_dscale(M, mei.sigma_t.x, sn[k]);
            }
        }
        active_surface = (active_surface || escaped_medium) && si.valid && !active_medium;
        if (active_surface) tr_multiplier *= bsdf_null_transmission(S, S.shapes[si.shape].bsdf);
        if (adjoint && (active_surface || active_medium) && medium >= 0 && (gm < 0 || medium == gm)) {   /* :425-427, active_adj = (surface | medium) & tr > 0 */
            const lrt_medium_desc &M = S.media[medium];
            float c[3] = { tr_multiplier.x, tr_multiplier.y, tr_multiplier.z }, dl[3] = { delta_L.x, delta_L.y, delta_L.z }, ae[3] = { adj_emitted.x, adj_emitted.y, adj_emitted.z };
            for (int k = 0; k < 3; ++k) if (c[k] > 0.f) {
                if (homogeneous_segment && escaped_medium) G->sigma_t[k] += (double) (dl[k] * ae[k] * (-seg_t) * M.scale);
                if (active_medium) G->sigma_t[k] += (double) (dl[k] * ae[k] * null_dlog[k]);
            }
            if (g_prb_debug) fprintf(stderr, "  [orc] nee seg_t %.9g tr %.9g %.9g %.9g ae %.9g %.9g %.9g\n", seg_t, c[0], c[1], c[2], ae[0], ae[1], ae[2]);
        }
        transmittance *= tr_multiplier;
        if (active_surface) ray = spawn_ray(si.p, si.n, ray.d);
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        active = active && (active_medium || active_surface) && any_nonzero(transmittance);
        if (active) total_dist += active_medium ? mei_t : si.t;
        if (active_surface && is_medium_transition(S.shapes[si.shape])) medium = target_medium(S.shapes[si.shape], ray.d, si.n);
    }
    return emitter_val * transmittance;
}

/* src/python/python/ad/integrators/prbvolpath.py:96-351.  adjoint == false: primal pass (returns L);
   adjoint == true: replay with L_in (the primal result) and delta_L, accumulating parameter gradients. */
static void prb_sample(Ctx &C, Ray ray, bool adjoint, V3 delta_L, V3 L_in, V3 *L_out, bool *valid_out, Grads *G) {
    const Scene &S = C.S;
    uint32_t depth = 0;
    V3 L = adjoint ? L_in : V3(0.f), throughput(1.f);
    float eta = 1.f;
    bool active = true, needs_intersection = true, valid_ray = false, specular_chain = true;
    SI si; memset((void *) &si, 0, sizeof(si));
    V3 last_scatter_p(0.f); float last_scatter_direction_pdf = 1.f;
    int medium = -1;                                  /* :127-128 "TODO: support sensors inside media" */
    uint32_t channel = std::min((uint32_t) (3.f * C.next()), 2u);
    const uint32_t max_depth = (uint32_t) C.max_depth;
    while (active) {
        C.n_iter++;
        active = any_nonzero(throughput);
        float q = fminf(max3(throughput) * sqr(eta), 0.99f);
        bool perform_rr = depth > (uint32_t) C.rr_depth;
        if (active) { float u = C.next(); active = (u < q) || !perform_rr; }
        if (perform_rr) throughput *= rcp(q);
        bool active_medium = active && medium >= 0, active_surface = active && !active_medium;
        bool escaped_medium = false, act_medium_scatter = false;
        MI mei; mei.t = kInf; mei.wi = -ray.d; mei.p = V3(0.f); mei.medium = medium;
        V3 weight(1.f);
        float seg_t = 0.f; bool in_medium_segment = false;
        bool act_null_scatter = false; float scatter_prob = 1.f;
        if (!active_medium) C.skip(1);               /* prbvolpath.py:158 */
        if (active_medium) {
            mei = medium_sample_interaction(S, medium, ray, C.next(), channel);
            if (mei.valid() && medium_is_homogeneous(S.media[medium])) ray.maxt = mei.t;   /* :163 */
            if (needs_intersection) { Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
            needs_intersection = false;
            if (si.t < mei.t) mei.t = kInf;
            seg_t = fminf(mei.t, si.t) - mei.mint;                                  /* medium.cpp:92-104 */
            V3 tr(m_exp(-seg_t * mei.combined.x), m_exp(-seg_t * mei.combined.y), m_exp(-seg_t * mei.combined.z));
            V3 pdf = (si.t < mei.t) ? tr : tr * mei.combined;
            float tr_pdf = idx3(pdf, channel);
            weight = (tr_pdf > 0.f) ? tr / tr_pdf : V3(0.f);
            escaped_medium = !mei.valid();
            active_medium = mei.valid();
            in_medium_segment = true;
        }
        if (S.prb_handle_null_scattering) {          /* :178-183: one more draw per trip in a scene that holds a heterogeneous medium */
            if (!active_medium) C.skip(1);
            else {
                scatter_prob = mean3(mei.sigma_t / mei.combined);
                act_null_scatter = C.next() >= scatter_prob;
                if (act_null_scatter) weight *= mei.sigma_n / (1.f - scatter_prob);
            }
        }
        if (active_medium && !act_null_scatter) { act_medium_scatter = true; depth += 1; last_scatter_p = mei.p; }
        active = active && depth < max_depth;
        act_medium_scatter = act_medium_scatter && active;
        const float si_t_before = si.t;
        if (act_null_scatter) { ray.o = mei.p; si.t = si.t - mei.t; }                /* :194-196 */
        if (act_medium_scatter) weight *= mei.sigma_s / scatter_prob;
        throughput *= weight;
        (void) si_t_before;
        const int gm = C.grad_medium;
        if (adjoint && in_medium_segment && (gm < 0 || medium == gm)) {             /* :199-204 */
            const lrt_medium_desc &M = S.media[medium];
            float w[3] = { weight.x, weight.y, weight.z }, l[3] = { L.x, L.y, L.z }, dl[3] = { delta_L.x, delta_L.y, delta_L.z };
            float sn[3] = { mei.sigma_n.x, mei.sigma_n.y, mei.sigma_n.z };
            for (int k = 0; k < 3; ++k) {
                float Lo = l[k] / fmaxf(1e-8f, w[k]);
                if (!medium_is_homogeneous(M)) {
                    /* weight_k = exp(-t majorant) / pdf [* sigma_s_k / scatter_prob | * sigma_n_k / (1 - scatter_prob)]: only the bracket depends on `scale` */
                    float dlog = act_medium_scatter ? 1.f / M.scale : (act_null_scatter ? het_null_dlog_dscale(M, mei.sigma_t.x, sn[k]) : 0.f);
                    G->sigma_t[k] += (double) (dl[k] * Lo * (w[k] * dlog));
                    if (act_medium_scatter) G->albedo[k] += (double) (dl[k] * Lo * (w[k] / M.albedo[k]));
                    continue;
                }
                /* weight_k = exp(-t sigma_k) / pdf [* sigma_k a_k]:  d/dsigma_k = w (-t [+ 1/sigma_k]),  d/da_k = w / a_k */
                float st = M.sigma_t[k] * M.scale;
                float dws = w[k] * (-seg_t) + (act_medium_scatter ? w[k] / st : 0.f);
                if (!(seg_t < kInf)) dws = 0.f;                                     /* exp(-inf) = 0: no dependence */
                G->sigma_t[k] += (double) (dl[k] * Lo * dws * M.scale);
                if (act_medium_scatter) G->albedo[k] += (double) (dl[k] * Lo * (w[k] / M.albedo[k]));
                if (g_prb_debug) fprintf(stderr, "  [orc] trip %u k %d seg_t %.9g w %.9g L %.9g dl %.9g scatter %d term %.9g\n", depth, k, seg_t, w[k], l[k], dl[k], (int) act_medium_scatter, dl[k] * Lo * dws * M.scale);
            }
        }
        /* ---- surface interactions */
        active_surface = active_surface || escaped_medium;
        bool intersect = active_surface && needs_intersection;
        if (intersect) { Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
        if (C.hide_emitters && intersect && depth == 0 && si.valid && S.shapes[si.shape].emitter >= 0) {
            Ray r2 = spawn_ray(si.p, si.n, ray.d);
            bool a = true; Hit h; h.prim = 0xffffffffu; h.t = kInf;
            while (a) {
                h = S.intersect(r2, false, false);
                a = h.valid() && S.shapes[S.face_shape[h.prim]].emitter >= 0;
                if (a) { SI s2 = S.compute_si(r2, h); r2 = spawn_ray(s2.p, s2.n, r2.d); }
            }
            si = S.compute_si(r2, h);
        }
        if (active_surface) {
            bool count_direct = (depth == 0) || specular_chain;
            int emitter = si_emitter(S, si);
            bool active_e = emitter >= 0 && !(depth == 0 && C.hide_emitters);
            if (active_e) {
                float emitter_pdf = pdf_emitter_direction(S, last_scatter_p, si, emitter);
                V3 emitted = emitter_eval(S, emitter, si);
                V3 contrib = count_direct ? throughput * emitted : throughput * mis_weight(last_scatter_direction_pdf, emitter_pdf) * emitted;
                L = adjoint ? L - contrib : L + contrib;
                if (g_prb_debug && (delta_L.x != 0.f || !adjoint)) fprintf(stderr, "  [orc] %s emitter hit: depth %u contrib %.9g %.9g %.9g L after %.9g %.9g %.9g\n", adjoint ? "adjoint" : "primal", depth, contrib.x, contrib.y, contrib.z, L.x, L.y, L.z);
            }
        }
        active_surface = active_surface && si.valid;
        /* ---- emitter sampling (:267-297) */
        int b = active_surface ? S.shapes[si.shape].bsdf : -1;
        bool active_e_surface = active_surface && (bsdf_flags(S, b) & F_SMOOTH) && (depth + 1 < max_depth);
        bool sample_emitters = act_medium_scatter ? (S.media[medium].sample_emitters != 0) : false;
        if (act_medium_scatter) specular_chain = !sample_emitters;
        bool active_e_medium = act_medium_scatter && sample_emitters;
        if (!(active_e_surface || active_e_medium)) C.skip(1);        /* prbvolpath.py:365 */
        if (active_e_surface || active_e_medium) {
            Sampler nee_rng = C.smp;                  /* sampler.clone(): the adjoint call replays the same numbers */
            DirSample ds;
            V3 rp = active_e_medium ? mei.p : si.p, rn = active_e_medium ? V3(0.f) : si.n;
            V3 emitted = prb_sample_emitter(C, rp, rn, active_e_surface ? &si : nullptr, medium, channel, &ds, false, V3(0.f), V3(0.f), nullptr);
            V3 nee_weight; float nee_pdf;
            if (active_e_surface) { V3 wo = si.sh.to_local(ds.d); nee_weight = bsdf_eval(S, b, si, wo); nee_pdf = bsdf_pdf(S, b, si, wo); }
            else { float pv = phase_eval(S.media[medium], mei.wi, ds.d); nee_weight = V3(pv); nee_pdf = pv; }
            V3 contrib = throughput * nee_weight * mis_weight(ds.pdf, ds.delta ? 0.f : nee_pdf) * emitted;
            L = adjoint ? L - contrib : L + contrib;
            if (g_prb_debug && (delta_L.x != 0.f || !adjoint)) fprintf(stderr, "  [orc] %s nee: depth %u surface %d contrib %.9g %.9g %.9g L after %.9g %.9g %.9g\n", adjoint ? "adjoint" : "primal", depth, (int) active_e_surface, contrib.x, contrib.y, contrib.z, L.x, L.y, L.z);
            if (adjoint) {
                Sampler saved = C.smp; uint64_t ns = C.n_shadow, nn = C.n_shadow_needed;
                C.smp = nee_rng;
                DirSample ds2;
                prb_sample_emitter(C, rp, rn, active_e_surface ? &si : nullptr, medium, channel, &ds2, true, delta_L, contrib, G);
                C.smp = saved; C.n_shadow = ns; C.n_shadow_needed = nn;
                if (active_e_medium && S.media[medium].phase == LRT_PHASE_HG && (gm < 0 || medium == gm)) {    /* backward(dL * contrib) through phase_val */
                    float dlg = hg_dlog_dg(S.media[medium].g, dot(ds.d, mei.wi));
                    G->g += (double) ((delta_L.x * contrib.x + delta_L.y * contrib.y + delta_L.z * contrib.z) * dlg);
                }
            }
        }
        /* ---- phase function sampling (:299-317) */
        if (!act_medium_scatter) C.skip(2);          /* prbvolpath.py:294-295 */
        if (act_medium_scatter) {
            valid_ray = true;
            const lrt_medium_desc &M = S.media[medium];
            (void) C.next();
            float s2x, s2y; C.next2(&s2x, &s2y);
            V3 wo; float phase_pdf; phase_sample(M, mei.wi, s2x, s2y, &wo, &phase_pdf);
            act_medium_scatter = phase_pdf > 0.f;
            if (act_medium_scatter) {
                if (adjoint && M.phase == LRT_PHASE_HG && (gm < 0 || medium == gm)) {
                    float pe = phase_eval(M, mei.wi, wo);
                    float dlg = hg_dlog_dg(M.g, dot(wo, mei.wi));
                    float l[3] = { L.x, L.y, L.z }, dl[3] = { delta_L.x, delta_L.y, delta_L.z };
                    for (int k = 0; k < 3; ++k) G->g += (double) (dl[k] * (pe * (l[k] / fmaxf(1e-8f, pe))) * dlg);
                }
                ray = spawn_ray(mei.p, V3(0.f), wo);
                needs_intersection = true;
                last_scatter_direction_pdf = phase_pdf;
            }
        }
        /* ---- BSDF sampling (:321-349) */
        if (!active_surface) C.skip(2);              /* prbvolpath.py:317-318 */
        if (active_surface) {
            const lrt_shape_desc &sd = S.shapes[si.shape];
            float s1 = C.next(), s2x, s2y; C.next2(&s2x, &s2y);
            BSDFSample bs; V3 bsdf_weight;
            bsdf_sample(S, b, si, s1, s2x, s2y, &bs, &bsdf_weight);
            active_surface = bs.pdf > 0.f;
            if (active_surface) {
                throughput *= bsdf_weight;
                eta *= bs.eta;
                ray = spawn_ray(si.p, si.n, si.sh.to_world(bs.wo));
                needs_intersection = true;
                bool non_null = !(bs.type & F_NULL);
                if (non_null) { depth += 1; last_scatter_p = si.p; last_scatter_direction_pdf = bs.pdf; valid_ray = true; }
                specular_chain = specular_chain || (non_null && (bs.type & F_DELTA));
                specular_chain = specular_chain && !(bs.type & F_SMOOTH);
                if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, si.n);
            }
        }
        active = active && (active_surface || active_medium);
    }
    *L_out = L; *valid_out = valid_ray;
}

/* ------------------------------------------------------------------ film */
/* spp: samples of ONE pass (= all samples unless the render is split, integrator.cpp:176-184,275-293); spp_total: all passes */
struct Opts { int integrator, max_depth, rr_depth; bool hide_emitters; uint32_t spp, seed; uint32_t spp_total = 0, n_passes = 1, pass = 0; };

static Opts resolve_opts(const Scene &S, const lrt_render_opts *o) {
    Opts r;
    r.integrator = (o && o->integrator >= 0) ? o->integrator : S.d.integrator.type;
    r.max_depth = (o && o->max_depth != -2) ? o->max_depth : S.d.integrator.max_depth;
    if (r.max_depth < 0 || r.max_depth > 65535) r.max_depth = 65535;        /* the product's reading of "unbounded" (device.hip, resolve): 16-bit depth field, and an end for paths Russian roulette cannot stop */
    r.rr_depth = (o && o->rr_depth >= 0) ? o->rr_depth : S.d.integrator.rr_depth;
    r.hide_emitters = (o && o->hide_emitters >= 0) ? (o->hide_emitters != 0) : (S.d.integrator.hide_emitters != 0);
    r.spp = (o && o->spp) ? o->spp : S.d.sample_count;
    if (S.d.sampler_type == LRT_SAMPLER_LD) r.spp = ld_round_sample_count(r.spp);      /* integrator.cpp:169-171 */
    r.seed = o ? o->seed : 0;
    /* passes: `samples_per_pass` (integrator.cpp:176-184), then the 2^32 - 1 wavefront limit (:275-285) */
    r.spp_total = r.spp; r.n_passes = 1; r.pass = 0;
    if (r.integrator != LRT_INTEGRATOR_PRBVOLPATH) {
        uint32_t per = S.d.samples_per_pass ? std::min(S.d.samples_per_pass, r.spp) : r.spp;
        if (per == 0 || r.spp % per != 0) per = r.spp;               /* the reference throws; callers check with orc_passes_valid() */
        uint64_t wavefront = (uint64_t) S.d.film.crop_width * S.d.film.crop_height * per, limit = 0xffffffffull;
        if (wavefront > limit) per /= (uint32_t) ((wavefront + limit - 1) / limit);
        if (per == 0) per = 1;
        r.n_passes = r.spp_total / per; r.spp = per;
    }
    return r;
}

struct SampleOut { float r, g, b, a; float px, py; };

/* src/render/integrator.cpp:321-338 (lane -> pixel) + :449-521 (render_sample) */
/* carry: per-lane PCG32 state handed from pass to pass (Sampler::advance() does not reseed the independent sampler:
   pass p + 1 continues every lane's stream where its path of pass p stopped); null for single-pass renders. */
static SampleOut render_lane(const Scene &S, const Opts &O, uint64_t lane, orc_stats *st, uint64_t *carry = nullptr) {
    const lrt_film_desc &F = S.d.film;
    Ctx C(S); C.max_depth = O.max_depth; C.rr_depth = O.rr_depth; C.hide_emitters = O.hide_emitters;
    C.smp = lane_sampler(S.d.sampler_type, S.d.sampler_seed, O.seed, (uint32_t) lane, O.spp);
    if (S.d.sampler_type == LRT_SAMPLER_LD) {            /* sampler.cpp:69-72,109-117: sample index = pass * spp_per_pass + lane % spp_per_pass */
        C.smp.sample_index = O.pass * O.spp + (uint32_t) (lane % O.spp); C.smp.sample_count = O.spp_total;
    } else if (carry && O.pass > 0) C.smp.rng.state = carry[lane];
    C.stream_continues = carry != nullptr && S.d.sampler_type != LRT_SAMPLER_LD && O.pass + 1 < O.n_passes;
    uint32_t idx = (uint32_t) (lane / O.spp);
    uint32_t W = (uint32_t) F.crop_width;
    uint32_t py = idx / W, px = idx - py * W;
    float posx = (float) ((int) px + F.crop_offset_x), posy = (float) ((int) py + F.crop_offset_y);
    float sclx = 1.f / (float) F.crop_width, scly = 1.f / (float) F.crop_height;
    float offx = -(float) F.crop_offset_x * sclx, offy = -(float) F.crop_offset_y * scly;
    float jx, jy; C.next2(&jx, &jy);
    float spx = posx + jx, spy = posy + jy;
    float ax = fmaf(spx, sclx, offx), ay = fmaf(spy, scly, offy);
    Ray ray = sample_ray(S, ax, ay);
    V3 L; bool valid;
    if (O.integrator == LRT_INTEGRATOR_PATH) path_sample(C, ray, &L, &valid);
    else if (O.integrator == LRT_INTEGRATOR_PRBVOLPATH) prb_sample(C, ray, false, V3(0.f), V3(0.f), &L, &valid, nullptr);
    else if (O.integrator == LRT_INTEGRATOR_BIOVOLPATH) biovolpath_sample(C, ray, S.d.sensor.medium, &L, &valid);
    else if (O.integrator == LRT_INTEGRATOR_BIOVOLPATH06) biovolpath06_sample(C, ray, S.d.sensor.medium, &L, &valid);
    else if (O.integrator == LRT_INTEGRATOR_VOLPATHMIS) { if (S.d.use_spectral_mis) volpathmis_sample<true>(C, ray, S.d.sensor.medium, &L, &valid); else volpathmis_sample<false>(C, ray, S.d.sensor.medium, &L, &valid); }
    else volpath_sample(C, ray, S.d.sensor.medium, &L, &valid);
    if (st) { st->n_iter += C.n_iter; st->n_shadow += C.n_shadow; st->n_shadow_needed += C.n_shadow_needed; st->n_samples += 1; }
    if (carry) carry[lane] = C.smp.rng.state;
    SampleOut o; o.r = L.x; o.g = L.y; o.b = L.z; o.a = valid ? 1.f : 0.f;
    bool box = F.rfilter == LRT_RFILTER_BOX;
    o.px = box ? posx : spx; o.py = box ? posy : spy;
    return o;
}

/* src/render/imageblock.cpp:174-232 (box) and :431-500 (coalesced footprint) */
static void film_put(const Scene &S, float *film, const SampleOut &s) {
    const lrt_film_desc &F = S.d.film;
    int C = F.has_alpha ? 5 : 4, W = F.crop_width, H = F.crop_height;
    float vals[5]; int k = 0;
    vals[k++] = s.r; vals[k++] = s.g; vals[k++] = s.b; if (F.has_alpha) vals[k++] = s.a; vals[k++] = 1.f;
    if (F.rfilter == LRT_RFILTER_BOX) {
        int x = (int) floorf(s.px) - F.crop_offset_x, y = (int) floorf(s.py) - F.crop_offset_y;
        if (x < 0 || y < 0 || x >= W || y >= H) return;
        float *p = film + ((size_t) y * W + x) * C;
        for (int c = 0; c < C; ++c) p[c] += vals[c];
        return;
    }
    int n = (int) ceilf(S.rf_radius - .5f), count = 2 * n + 1;
    int pix = (int) floorf(s.px) - n, piy = (int) floorf(s.py) - n;
    float relx = (float) pix + .5f - s.px, rely = (float) piy + .5f - s.py;
    float wx[16], wy[16];
    for (int i = 0; i < count; ++i) { wx[i] = S.rfilter_eval(relx); wy[i] = S.rfilter_eval(rely); relx += 1.f; rely += 1.f; }
    for (int ys = 0; ys < count; ++ys) {
        int y = piy - F.crop_offset_y + ys;
        if (y < 0 || y >= H) continue;
        for (int xs = 0; xs < count; ++xs) {
            int x = pix - F.crop_offset_x + xs;
            if (x < 0 || x >= W) continue;
            float w = wy[ys] * wx[xs];
            float *p = film + ((size_t) y * W + x) * C;
            for (int c = 0; c < C; ++c) p[c] += vals[c] * w;
        }
    }
}

/* src/films/hdrfilm.cpp:306-410 develop (RGB[A] / W, W == 0 -> 1) */
static void film_develop(const Scene &S, const float *film, float *image) {
    const lrt_film_desc &F = S.d.film;
    int C = F.has_alpha ? 5 : 4, T = F.has_alpha ? 4 : 3;
    size_t np = (size_t) F.crop_width * F.crop_height;
    for (size_t i = 0; i < np; ++i) {
        float w = film[i * C + C - 1]; if (w == 0.f) w = 1.f;
        for (int c = 0; c < T; ++c) image[i * T + c] = film[i * C + c] / w;
    }
}

static int hw_threads(int n) { if (n > 0) return n; unsigned h = std::thread::hardware_concurrency(); return h ? (int) h : 1; }

template <typename Fn> static void parallel_for(uint64_t n, int n_threads, Fn fn) {
    n_threads = hw_threads(n_threads);
    if (n_threads == 1 || n < 256) { fn(0, n, 0); return; }
    std::vector<std::thread> th;
    std::atomic<uint64_t> next(0);
    const uint64_t grain = std::max<uint64_t>(64, std::min<uint64_t>(16384, n / (uint64_t) (n_threads * 8) + 1));
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&, t]() {
            for (;;) { uint64_t b = next.fetch_add(grain); if (b >= n) break; fn(b, std::min(n, b + grain), t); }
        });
    for (auto &x : th) x.join();
}

} // namespace orc

using namespace orc;

static thread_local std::string g_err;
extern "C" const char *orc_last_error(void) { return g_err.c_str(); }

extern "C" int orc_render_samples(orc_scene *s, const lrt_render_opts *opts, uint64_t lane_begin, uint32_t n,
                                  int n_threads, float *out, orc_stats *stats) {
    const Scene &S = s->s; Opts O = resolve_opts(S, opts);
    int nt = hw_threads(n_threads);
    std::vector<orc_stats> st(nt); for (auto &x : st) memset(&x, 0, sizeof(x));
    parallel_for(n, nt, [&](uint64_t b, uint64_t e, int t) {
        for (uint64_t i = b; i < e; ++i) {
            SampleOut o = render_lane(S, O, lane_begin + i, &st[t]);
            out[4 * i] = o.r; out[4 * i + 1] = o.g; out[4 * i + 2] = o.b; out[4 * i + 3] = o.a;
        }
    });
    if (stats) { memset(stats, 0, sizeof(*stats)); for (auto &x : st) { stats->n_iter += x.n_iter; stats->n_shadow += x.n_shadow; stats->n_shadow_needed += x.n_shadow_needed; stats->n_samples += x.n_samples; } }
    return 0;
}

extern "C" int orc_render(orc_scene *s, const lrt_render_opts *opts, int n_threads, float *film_raw, float *image, orc_stats *stats) {
    const Scene &S = s->s; Opts O = resolve_opts(S, opts);
    const lrt_film_desc &F = S.d.film;
    int C = F.has_alpha ? 5 : 4;
    size_t np = (size_t) F.crop_width * F.crop_height;
    uint64_t N = (uint64_t) np * O.spp;                                      /* lanes of one pass */
    if (N > 0xffffffffull) { g_err = "orc_render: more than 2^32 lanes"; return 1; }
    std::vector<float> film(np * C, 0.f);
    int nt = hw_threads(n_threads);
    std::vector<orc_stats> st(nt); for (auto &x : st) memset(&x, 0, sizeof(x));
    const uint64_t chunk = 1u << 20;
    std::vector<SampleOut> buf((size_t) std::min<uint64_t>(chunk, N));
    std::vector<uint64_t> carry;
    if (O.n_passes > 1 && S.d.sampler_type != LRT_SAMPLER_LD) carry.resize((size_t) N);
    for (uint32_t pass = 0; pass < O.n_passes; ++pass) {                     /* integrator.cpp:343-353 */
        O.pass = pass;
        for (uint64_t base = 0; base < N; base += chunk) {
            uint64_t cnt = std::min<uint64_t>(chunk, N - base);
            parallel_for(cnt, nt, [&](uint64_t b, uint64_t e, int t) {
                for (uint64_t i = b; i < e; ++i) buf[i] = render_lane(S, O, base + i, &st[t], carry.empty() ? nullptr : carry.data());
            });
            for (uint64_t i = 0; i < cnt; ++i) film_put(S, film.data(), buf[i]);   /* lane order: deterministic */
        }
    }
    if (film_raw) memcpy(film_raw, film.data(), film.size() * sizeof(float));
    if (image) film_develop(S, film.data(), image);
    if (stats) { memset(stats, 0, sizeof(*stats)); for (auto &x : st) { stats->n_iter += x.n_iter; stats->n_shadow += x.n_shadow; stats->n_shadow_needed += x.n_shadow_needed; stats->n_samples += x.n_samples; } }
    return 0;
}

/* src/render/integrator.cpp:190-273,399-434: scalar-variant tiling (CPU baseline workload) */
extern "C" int orc_render_scalar(orc_scene *s, const lrt_render_opts *opts, int n_threads, float *film_raw, float *image, orc_stats *stats) {
    const Scene &S = s->s; Opts O = resolve_opts(S, opts);
    const lrt_film_desc &F = S.d.film;
    int C = F.has_alpha ? 5 : 4, W = F.crop_width, H = F.crop_height;
    size_t np = (size_t) W * H;
    std::vector<float> film(np * C, 0.f);
    const int bs = 32;
    int bx = (W + bs - 1) / bs, by = (H + bs - 1) / bs;
    int nt = hw_threads(n_threads);
    std::vector<orc_stats> st(nt); for (auto &x : st) memset(&x, 0, sizeof(x));
    std::mutex mtx;
    parallel_for((uint64_t) bx * by, nt, [&](uint64_t b0, uint64_t b1, int t) {
        for (uint64_t blk = b0; blk < b1; ++blk) {
            int ox = (int) (blk % bx) * bs, oy = (int) (blk / bx) * bs;
            int w = std::min(bs, W - ox), h = std::min(bs, H - oy);
            std::vector<SampleOut> local; local.reserve((size_t) w * h * O.spp);
            for (int i = 0; i < bs * bs; ++i) {
                /* dr::morton_decode<Point2u>(i) */
                auto compact = [](uint32_t x) { x &= 0x55555555u; x = (x ^ (x >> 1)) & 0x33333333u; x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
                                                x = (x ^ (x >> 4)) & 0x00ff00ffu; x = (x ^ (x >> 8)) & 0x0000ffffu; return x; };
                int lx = (int) compact((uint32_t) i), ly = (int) compact((uint32_t) i >> 1);
                if (lx >= w || ly >= h) continue;
                Ctx Cx(S); Cx.max_depth = O.max_depth; Cx.rr_depth = O.rr_depth; Cx.hide_emitters = O.hide_emitters;
                uint32_t sd = O.seed * (uint32_t) (W * H) + (uint32_t) blk * (uint32_t) (bs * bs) + (uint32_t) i;
                Cx.smp.rng.seed((uint64_t) (S.d.sampler_seed + sd), 0xda3e39cb94b95bdbULL);
                float posx = (float) (ox + lx + F.crop_offset_x), posy = (float) (oy + ly + F.crop_offset_y);
                for (uint32_t k = 0; k < O.spp; ++k) {
                    float jx = Cx.next(), jy = Cx.next();
                    float spx = posx + jx, spy = posy + jy;
                    const float sclx = 1.f / (float) W, scly = 1.f / (float) H;       /* integrator.cpp:462-466: scale = 1 / crop_size, offset = -crop_offset * scale, adjusted_pos = fmadd(sample_pos, scale, offset) */
                    Ray ray = sample_ray(S, fmaf(spx, sclx, -(float) F.crop_offset_x * sclx), fmaf(spy, scly, -(float) F.crop_offset_y * scly));
                    V3 L; bool valid;
                    if (O.integrator == LRT_INTEGRATOR_PATH) path_sample(Cx, ray, &L, &valid);
                    else if (O.integrator == LRT_INTEGRATOR_BIOVOLPATH) biovolpath_sample(Cx, ray, S.d.sensor.medium, &L, &valid);
                    else if (O.integrator == LRT_INTEGRATOR_BIOVOLPATH06) biovolpath06_sample(Cx, ray, S.d.sensor.medium, &L, &valid);
                    else volpath_sample(Cx, ray, S.d.sensor.medium, &L, &valid);
                    bool box = F.rfilter == LRT_RFILTER_BOX;
                    local.push_back({ L.x, L.y, L.z, valid ? 1.f : 0.f, box ? posx : spx, box ? posy : spy });
                }
                st[t].n_iter += Cx.n_iter; st[t].n_shadow += Cx.n_shadow; st[t].n_shadow_needed += Cx.n_shadow_needed; st[t].n_samples += O.spp;
            }
            std::lock_guard<std::mutex> lk(mtx);
            for (auto &so : local) film_put(S, film.data(), so);
        }
    });
    if (film_raw) memcpy(film_raw, film.data(), film.size() * sizeof(float));
    if (image) film_develop(S, film.data(), image);
    if (stats) { memset(stats, 0, sizeof(*stats)); for (auto &x : st) { stats->n_iter += x.n_iter; stats->n_shadow += x.n_shadow; stats->n_shadow_needed += x.n_shadow_needed; stats->n_samples += x.n_samples; } }
    return 0;
}

extern "C" int orc_trace(orc_scene *s, const lrt_rays_soa *rays, const lrt_hits_soa *hits, uint32_t n, int any_hit, int brute) {
    const Scene &S = s->s;
    parallel_for(n, 0, [&](uint64_t b, uint64_t e, int) {
        for (uint64_t i = b; i < e; ++i) {
            Ray r; r.o = V3(rays->ox[i], rays->oy[i], rays->oz[i]); r.d = V3(rays->dx[i], rays->dy[i], rays->dz[i]); r.maxt = rays->tmax[i];
            Hit h = S.intersect(r, any_hit != 0, brute != 0);
            if (any_hit) { hits->t[i] = h.valid() ? 0.f : kInf; continue; }
            hits->t[i] = h.t; if (hits->u) hits->u[i] = h.u; if (hits->v) hits->v[i] = h.v; if (hits->prim) hits->prim[i] = h.prim;
        }
    });
    return 0;
}

/* ---------------------------------------------------------- unit hooks */
extern "C" void orc_tea32(uint32_t v0, uint32_t v1, int rounds, uint32_t *o0, uint32_t *o1) { tea32(v0, v1, rounds, o0, o1); }
/* LowDiscrepancySampler::next_1d / next_2d for (sample_index, dimension) of the sequence with the given scramble seed */
extern "C" void orc_ld_sample(uint32_t sample_count, uint32_t scramble_seed, uint32_t sample_index, uint32_t dim, int two_d, float *out) {
    Sampler s; s.type = LRT_SAMPLER_LD; s.sample_count = sample_count; s.scramble_seed = scramble_seed; s.sample_index = sample_index; s.dim = dim;
    if (two_d) s.next2(&out[0], &out[1]); else out[0] = s.next1();
}
extern "C" uint32_t orc_ld_round_sample_count(uint32_t spp) { return ld_round_sample_count(spp); }
extern "C" uint32_t orc_permute(uint32_t i, uint32_t n, uint32_t seed) { return permute_tea(i, n, seed); }
extern "C" float orc_tea_float32(uint32_t v0, uint32_t v1, int rounds) {   /* random.h:134-139 */
    uint32_t a, b; tea32(v0, v1, rounds, &a, &b); return u2f((b >> 9) | 0x3f800000u) - 1.f;
}
extern "C" double orc_tea_float64(uint32_t v0, uint32_t v1, int rounds) {  /* random.h sample_tea_64 + float64 */
    uint32_t a, b; tea32(v0, v1, rounds, &a, &b);
    uint64_t u = (uint64_t) a + ((uint64_t) b << 32);
    uint64_t bits = (u >> 12) | 0x3ff0000000000000ull; double d; memcpy(&d, &bits, 8); return d - 1.0;
}
extern "C" void orc_pcg32_u32(uint64_t initstate, uint64_t initseq, uint32_t n, uint32_t *out) {
    PCG32 r; r.seed(initstate, initseq); for (uint32_t i = 0; i < n; ++i) out[i] = r.next_u32();
}
extern "C" void orc_lane_stream(uint32_t base_seed, uint32_t seed, uint32_t lane, uint32_t n, float *out) {
    PCG32 r = lane_rng(base_seed, seed, lane); for (uint32_t i = 0; i < n; ++i) out[i] = r.next();
}
extern "C" void orc_math_eval(int fn, const float *x, const float *y, uint32_t n, float *out, float *out2) {
    for (uint32_t i = 0; i < n; ++i) switch (fn) {
        case 0: out[i] = m_log(x[i]); break;
        case 1: out[i] = m_exp(x[i]); break;
        case 2: m_sincos(x[i], &out[i], &out2[i]); break;
        case 3: out[i] = m_atan2(y[i], x[i]); break;
        case 4: out[i] = m_acos(x[i]); break;
        case 5: out[i] = m_log2(x[i]); break;
    }
}
extern "C" void orc_hg_sample(float g, const float wi[3], float u1, float u2, float wo[3], float *pdf) {
    lrt_medium_desc M; memset(&M, 0, sizeof(M)); M.phase = LRT_PHASE_HG; M.g = g;
    V3 w; phase_sample(M, V3(wi[0], wi[1], wi[2]), u1, u2, &w, pdf); wo[0] = w.x; wo[1] = w.y; wo[2] = w.z;
}
extern "C" float orc_hg_eval(float g, float c) { return hg_eval(g, c); }
extern "C" void orc_square_to_cosine_hemisphere(float u1, float u2, float o[3]) { V3 v = square_to_cosine_hemisphere(u1, u2); o[0] = v.x; o[1] = v.y; o[2] = v.z; }
extern "C" void orc_square_to_uniform_sphere(float u1, float u2, float o[3]) { V3 v = square_to_uniform_sphere(u1, u2); o[0] = v.x; o[1] = v.y; o[2] = v.z; }
extern "C" void orc_fresnel(float c, float eta, float out[4]) { fresnel(c, eta, &out[0], &out[1], &out[2], &out[3]); }
extern "C" void orc_envmap_sample(orc_scene *s, float u1, float u2, float ref[3], float d[3], float *pdf, float rgb[3]) {
    DirSample ds; V3 w = sample_emitter_direction(s->s, V3(ref[0], ref[1], ref[2]), u1, u2, &ds);
    d[0] = ds.d.x; d[1] = ds.d.y; d[2] = ds.d.z; *pdf = ds.pdf; rgb[0] = w.x; rgb[1] = w.y; rgb[2] = w.z;
}
extern "C" float orc_envmap_pdf(orc_scene *s, const float d[3]) {
    SI si; memset((void *) &si, 0, sizeof(si)); si.valid = false; si.wi = V3(-d[0], -d[1], -d[2]);
    return pdf_emitter_direction(s->s, V3(0.f), si, s->s.env);
}
extern "C" void orc_envmap_eval(orc_scene *s, const float d[3], float rgb[3]) {
    V3 v = emitter_eval_env(s->s, V3(d[0], d[1], d[2])); rgb[0] = v.x; rgb[1] = v.y; rgb[2] = v.z;
}
extern "C" float orc_rfilter_eval(orc_scene *s, float x) { return s->s.rfilter_eval(x); }
/* bio media: out = { t, transmittance rgb, p xyz, bio type, candidate distance } of the 5-argument sample_interaction */
extern "C" void orc_bio_sample_interaction(orc_scene *s, int medium, const float o[3], const float d[3], float maxt, float sample,
                                           uint32_t channel, float depth, int jit, float out[9]) {
    const lrt_medium_desc &M = s->s.media[medium];
    Ray r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]); r.maxt = maxt;
    BioMI m = bio_sample_interaction(M, r, sample, channel, depth, jit != 0);
    int bt; float dist; bio_compute_distance(M, channel, sample, depth, &bt, &dist);
    out[0] = m.t; out[1] = m.transmittance.x; out[2] = m.transmittance.y; out[3] = m.transmittance.z;
    out[4] = m.p.x; out[5] = m.p.y; out[6] = m.p.z; out[7] = (float) bt; out[8] = dist;
}
extern "C" void orc_sample_ray(orc_scene *s, float px, float py, float o[3], float d[3], float *maxt) {
    Ray r = sample_ray(s->s, px, py); o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z; *maxt = r.maxt;
}

/* RBIntegrator.render_backward (src/python/python/ad/integrators/common.py:625-783):
   (0) delta_L = d(sum(image * grad_image))/dL per sample through splat + develop (:730-746),
   (1) primal pass, (2) adjoint replay with the same sampler state. */
extern "C" int orc_render_backward(orc_scene *s, const lrt_render_opts *opts, int n_threads, const float *grad_image, lrt_param_grads *out) {
    const Scene &S = s->s; Opts O = resolve_opts(S, opts);
    const lrt_film_desc &F = S.d.film;
    const int T = F.has_alpha ? 4 : 3, W = F.crop_width, H = F.crop_height;
    const size_t np = (size_t) W * H; const uint64_t N = (uint64_t) np * O.spp;
    if (N > 0xffffffffull) { g_err = "orc_render_backward: more than 2^32 lanes"; return 1; }
    const bool box = F.rfilter == LRT_RFILTER_BOX;
    int nt = hw_threads(n_threads);
    const int grad_medium = opts ? opts->grad_medium : 0;
    if (grad_medium < -1 || grad_medium >= (int) S.d.n_media) { g_err = "orc_render_backward: grad_medium is not a medium of the scene"; return 1; }
    auto lane_setup = [&](uint64_t lane, Ctx &C, float *spx, float *spy, Ray *ray) {
        C.grad_medium = grad_medium;
        C.max_depth = O.max_depth; C.rr_depth = O.rr_depth; C.hide_emitters = O.hide_emitters;
        C.smp = lane_sampler(S.d.sampler_type, S.d.sampler_seed, O.seed, (uint32_t) lane, O.spp);
        uint32_t idx = (uint32_t) (lane / O.spp), py = idx / (uint32_t) W, px = idx - py * (uint32_t) W;
        float posx = (float) ((int) px + F.crop_offset_x), posy = (float) ((int) py + F.crop_offset_y);
        float jx, jy; C.next2(&jx, &jy);
        *spx = posx + jx; *spy = posy + jy;
        /* common.py sample_rays: scale = rcp(crop_size), offset = -crop_offset * scale, pos_adjusted = fma(pos_f, scale, offset): the
           same arithmetic as render_sample (render_lane above); a division for the offset differs in the last bit under a crop window */
        const float sclx = 1.f / (float) W, scly = 1.f / (float) H;
        *ray = sample_ray(S, fmaf(*spx, sclx, -(float) F.crop_offset_x * sclx), fmaf(*spy, scly, -(float) F.crop_offset_y * scly));
    };
    /* weight film (sum of filter weights per pixel); box filter: exactly spp */
    std::vector<float> wfilm(np, box ? (float) O.spp : 0.f);
    const int fn = (int) ceilf(S.rf_radius - .5f), fcount = 2 * fn + 1;
    if (!box) {
        for (uint64_t lane = 0; lane < N; ++lane) {
            Ctx C(S); float spx, spy; Ray ray; lane_setup(lane, C, &spx, &spy, &ray);
            int pix = (int) floorf(spx) - fn, piy = (int) floorf(spy) - fn;
            float relx = (float) pix + .5f - spx, rely = (float) piy + .5f - spy;
            for (int ys = 0; ys < fcount; ++ys) for (int xs = 0; xs < fcount; ++xs) {
                int x = pix - F.crop_offset_x + xs, y = piy - F.crop_offset_y + ys;
                if (x < 0 || y < 0 || x >= W || y >= H) continue;
                wfilm[(size_t) y * W + x] += S.rfilter_eval(rely + (float) ys) * S.rfilter_eval(relx + (float) xs);
            }
        }
    }
    std::vector<Grads> G(nt);
    parallel_for(N, nt, [&](uint64_t b, uint64_t e, int t) {
        for (uint64_t lane = b; lane < e; ++lane) {
            Ctx C(S); float spx, spy; Ray ray; lane_setup(lane, C, &spx, &spy, &ray);
            V3 dL(0.f);
            if (box) {
                size_t p = (size_t) (lane / O.spp);
                float w = wfilm[p]; if (w == 0.f) w = 1.f;
                dL = V3(grad_image[p * T] / w, grad_image[p * T + 1] / w, grad_image[p * T + 2] / w);
            } else {
                int pix = (int) floorf(spx) - fn, piy = (int) floorf(spy) - fn;
                float relx = (float) pix + .5f - spx, rely = (float) piy + .5f - spy;
                for (int ys = 0; ys < fcount; ++ys) for (int xs = 0; xs < fcount; ++xs) {
                    int x = pix - F.crop_offset_x + xs, y = piy - F.crop_offset_y + ys;
                    if (x < 0 || y < 0 || x >= W || y >= H) continue;
                    size_t p = (size_t) y * W + x;
                    float w = S.rfilter_eval(rely + (float) ys) * S.rfilter_eval(relx + (float) xs), wp = wfilm[p]; if (wp == 0.f) wp = 1.f;
                    float f = w / wp;
                    dL = dL + V3(grad_image[p * T] * f, grad_image[p * T + 1] * f, grad_image[p * T + 2] * f);
                }
            }
            Sampler start = C.smp;
            V3 L; bool valid;
            prb_sample(C, ray, false, V3(0.f), V3(0.f), &L, &valid, nullptr);
            C.smp = start; C.n_iter = 0;
            V3 L2; Grads g;
            prb_sample(C, ray, true, dL, L, &L2, &valid, &g);
            G[t].add(g);
        }
    });
    Grads tot; for (auto &g : G) tot.add(g);
    for (int k = 0; k < 3; ++k) { out->d_sigma_t[k] = (float) tot.sigma_t[k]; out->d_albedo[k] = (float) tot.albedo[k]; }
    out->d_g = (float) tot.g;
    return 0;
}
