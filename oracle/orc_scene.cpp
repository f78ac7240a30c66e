/*
 * orc_scene.cpp -- scene set-up, ray queries and surface interactions of the
 * CPU oracle.  TEST INFRASTRUCTURE ONLY (see orc.h).
 */
#include "orc_scene.h"
#include <algorithm>
#include <cstdio>
#include <cstring>

namespace orc {

/* ------------------------------------------------------------ 4x4 helpers */
static M4 m4_identity() { M4 r; memset(r.m, 0, sizeof(r.m)); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }
static M4 m4_transpose(const M4 &a) { M4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[4 * i + j] = a.m[4 * j + i]; return r; }
/* Dr.Jit matrix product: column-wise fmadd accumulation, k ascending */
static M4 m4_mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = a.m[4 * i + 0] * b.m[0 + j];
            for (int k = 1; k < 4; ++k) s = fmaf(a.m[4 * i + k], b.m[4 * k + j], s);
            r.m[4 * i + j] = s;
        }
    return r;
}
static M4 m4_inverse_affine(const M4 &a) {           /* host set-up only: double precision */
    double m[3][3], inv[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = a.m[4 * i + j];
    double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                 m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    double id = 1.0 / det;
    inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) * id; inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
    inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id; inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * id;
    inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id; inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
    inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * id; inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
    inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
    M4 r = m4_identity();
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) r.m[4 * i + j] = (float) inv[i][j];
        r.m[4 * i + 3] = (float) -(inv[i][0] * a.m[3] + inv[i][1] * a.m[7] + inv[i][2] * a.m[11]);
    }
    return r;
}

/* --------------------------------------------------------------- Hier2D */
static uint32_t log2i_ceil(uint32_t v) { uint32_t r = 0; while ((1u << r) < v) ++r; return r; }

/* include/mitsuba/core/distr_2d.h:403-510 (normalize = true, one slice) */
void Hier2D::build(const float *data, uint32_t w, uint32_t h) {
    uint32_t npx = w - 1, npy = h - 1;
    patch_size = { 1.f / (float) npx, 1.f / (float) npy };
    inv_patch_size = { (float) npx, (float) npy };
    max_patch_x = npx - 1; max_patch_y = npy - 1;
    uint32_t max_level = log2i_ceil(std::max(npx, npy));
    levels.clear();
    auto add = [&](uint32_t lw, uint32_t lh) { Level l; l.size = lw * lh; l.width = lw; l.data.assign(l.size, 0.f); levels.push_back(std::move(l)); };
    add(w, h);
    uint32_t lx = npx, ly = npy;
    for (int level = (int) max_level; level >= 0; --level) {
        lx += lx & 1u; ly += ly & 1u;
        add(lx, ly);
        lx >>= 1; ly >>= 1;
    }
    Level &L0 = levels[0], &L1 = levels[1];
    const float *in = data;
    double sum = 0.0;
    for (uint32_t y = 0; y < npy; ++y) {
        for (uint32_t x = 0; x < npx; ++x) {
            float avg = .25f * (in[0] + in[1] + in[w] + in[w + 1]);
            sum += (double) avg;
            L1.data[L1.index(x, y)] = avg;
            ++in;
        }
        ++in;
    }
    float scale = (float) ((double) (npx * npy) / sum);
    for (uint32_t i = 0; i < L0.size; ++i) L0.data[i] = data[i] * scale;
    for (uint32_t i = 0; i < L1.size; ++i) L1.data[i] *= scale;
    lx = npx; ly = npy;
    for (uint32_t level = 2; level <= max_level + 1; ++level) {
        const Level &a = levels[level - 1];
        Level &b = levels[level];
        lx = (lx + 1u) >> 1; ly = (ly + 1u) >> 1;
        for (uint32_t y = 0; y < ly; ++y)
            for (uint32_t x = 0; x < lx; ++x) {
                const float *d0 = &a.data[a.index(x * 2, y * 2)];
                b.data[b.index(x, y)] = d0[0] + d0[1] + d0[2] + d0[3];
            }
    }
}

/* include/mitsuba/core/warp.h:446-453 */
static float interval_to_linear(float v0, float v1, float sample) {
    if (fabsf(v0 - v1) > 1e-4f * (v0 + v1))
        return (v0 - safe_sqrt(lerpf(sqr(v0), sqr(v1), sample))) / (v0 - v1);
    return sample;
}

/* include/mitsuba/core/distr_2d.h:517-602 */
void Hier2D::sample(float sx, float sy, float *ox, float *oy, float *pdf) const {
    sx = clampf(sx, 0.f, 1.f); sy = clampf(sy, 0.f, 1.f);
    uint32_t offx = 0, offy = 0;
    for (int l = (int) levels.size() - 2; l > 0; --l) {
        const Level &lv = levels[l];
        offx <<= 1; offy <<= 1;
        uint32_t oi = lv.index(offx, offy);
        float v00 = lv.data[oi], v10 = lv.data[oi + 1], v01 = lv.data[oi + 2], v11 = lv.data[oi + 3];
        sx = clampf(sx, 0.f, 1.f); sy = clampf(sy, 0.f, 1.f);
        float r0 = v00 + v10, r1 = v01 + v11;
        sy *= r0 + r1;
        bool mask = sy > r0;
        if (mask) { offy += 1; sy -= r0; }
        sy /= mask ? r1 : r0;
        float c0 = mask ? v01 : v00, c1 = mask ? v11 : v10;
        sx *= c0 + c1;
        mask = sx > c0;
        if (mask) sx -= c0;
        sx /= mask ? c1 : c0;
        if (mask) offx += 1;
    }
    const Level &l0 = levels[0];
    uint32_t oi = offx + offy * l0.width;
    float v00 = l0.data[oi], v10 = l0.data[oi + 1], v01 = l0.data[oi + l0.width], v11 = l0.data[oi + l0.width + 1];
    /* warp::square_to_bilinear, include/mitsuba/core/warp.h:478-494 */
    float r0 = v00 + v10, r1 = v01 + v11;
    sy = interval_to_linear(r0, r1, sy);
    float c0 = lerpf(v00, v01, sy), c1 = lerpf(v10, v11, sy);
    sx = interval_to_linear(c0, c1, sx);
    *pdf = lerpf(c0, c1, sx);
    *ox = ((float) (int) offx + sx) * patch_size.x;
    *oy = ((float) (int) offy + sy) * patch_size.y;
}

/* include/mitsuba/core/distr_2d.h:695-726 */
float Hier2D::eval(float px, float py) const {
    px = clampf(px, 0.f, 1.f); py = clampf(py, 0.f, 1.f);
    px *= inv_patch_size.x; py *= inv_patch_size.y;
    uint32_t ox = std::min((uint32_t) (int) px, max_patch_x), oy = std::min((uint32_t) (int) py, max_patch_y);
    px -= (float) (int) ox; py -= (float) (int) oy;
    const Level &l0 = levels[0];
    uint32_t oi = ox + oy * l0.width;
    float v00 = l0.data[oi], v10 = l0.data[oi + 1], v01 = l0.data[oi + l0.width], v11 = l0.data[oi + l0.width + 1];
    return lerpf(lerpf(v00, v10, px), lerpf(v01, v11, px), py);
}

/* ------------------------------------------------------------------- BVH */
struct BuildPrim { float lo[3], hi[3], c[3]; uint32_t id; };

static void bvh_build(Scene &S) {
    uint32_t n = S.d.n_faces;
    std::vector<BuildPrim> prims(n);
    for (uint32_t f = 0; f < n; ++f) {
        BuildPrim &bp = prims[f]; bp.id = f;
        for (int a = 0; a < 3; ++a) { bp.lo[a] = kInf; bp.hi[a] = -kInf; }
        for (int k = 0; k < 3; ++k) {
            const float *p = &S.positions[3 * S.faces[3 * f + k]];
            for (int a = 0; a < 3; ++a) { bp.lo[a] = fminf(bp.lo[a], p[a]); bp.hi[a] = fmaxf(bp.hi[a], p[a]); }
        }
        for (int a = 0; a < 3; ++a) bp.c[a] = 0.5f * (bp.lo[a] + bp.hi[a]);
    }
    S.nodes.clear(); S.prim_ids.clear();
    if (n == 0) return;
    S.nodes.reserve(2 * n);
    S.nodes.push_back(BVHNode());
    struct Task { uint32_t node, begin, end; };
    std::vector<Task> stack; stack.push_back({ 0, 0, n });
    while (!stack.empty()) {
        Task t = stack.back(); stack.pop_back();
        float lo[3] = { kInf, kInf, kInf }, hi[3] = { -kInf, -kInf, -kInf }, clo[3] = { kInf, kInf, kInf }, chi[3] = { -kInf, -kInf, -kInf };
        for (uint32_t i = t.begin; i < t.end; ++i)
            for (int a = 0; a < 3; ++a) {
                lo[a] = fminf(lo[a], prims[i].lo[a]); hi[a] = fmaxf(hi[a], prims[i].hi[a]);
                clo[a] = fminf(clo[a], prims[i].c[a]); chi[a] = fmaxf(chi[a], prims[i].c[a]);
            }
        BVHNode nd;
        for (int a = 0; a < 3; ++a) {   /* conservative padding: slab test must never cull a hit */
            float pad = 1e-5f * (hi[a] - lo[a]) + 1e-6f * fmaxf(fabsf(lo[a]), fabsf(hi[a])) + 1e-30f;
            nd.lo[a] = lo[a] - pad; nd.hi[a] = hi[a] + pad;
        }
        uint32_t cnt = t.end - t.begin;
        int axis = 0; float ext = chi[0] - clo[0];
        for (int a = 1; a < 3; ++a) if (chi[a] - clo[a] > ext) { ext = chi[a] - clo[a]; axis = a; }
        if (cnt <= 4 || ext <= 0.f) {
            nd.left = (uint32_t) S.prim_ids.size(); nd.count = cnt;
            for (uint32_t i = t.begin; i < t.end; ++i) S.prim_ids.push_back(prims[i].id);
            S.nodes[t.node] = nd;
            continue;
        }
        uint32_t mid = (t.begin + t.end) / 2;
        std::nth_element(prims.begin() + t.begin, prims.begin() + mid, prims.begin() + t.end,
                         [axis](const BuildPrim &a, const BuildPrim &b) { return a.c[axis] < b.c[axis]; });
        nd.left = (uint32_t) S.nodes.size(); nd.count = 0;
        S.nodes[t.node] = nd;
        S.nodes.push_back(BVHNode()); S.nodes.push_back(BVHNode());
        stack.push_back({ nd.left, t.begin, mid });
        stack.push_back({ nd.left + 1, mid, t.end });
    }
}

/* include/mitsuba/render/mesh.h:506-527 moeller_trumbore.  A hit replaces the
   current one when it is strictly closer, or equally close with a lower
   primitive index (makes the result independent of traversal order). */
static inline void test_tri(const Scene &S, const Ray &r, uint32_t f, Hit &best) {
    const uint32_t *fi = &S.faces[3 * f];
    const float *a = &S.positions[3 * fi[0]], *b = &S.positions[3 * fi[1]], *c = &S.positions[3 * fi[2]];
    V3 p0(a[0], a[1], a[2]), p1(b[0], b[1], b[2]), p2(c[0], c[1], c[2]);
    V3 e1 = p1 - p0, e2 = p2 - p0;
    V3 pvec = cross(r.d, e2);
    float inv_det = rcp(dot(e1, pvec));
    V3 tvec = r.o - p0;
    float u = dot(tvec, pvec) * inv_det;
    if (!(u >= 0.f && u <= 1.f)) return;
    V3 qvec = cross(tvec, e1);
    float v = dot(r.d, qvec) * inv_det;
    if (!(v >= 0.f && u + v <= 1.f)) return;
    float t = dot(e2, qvec) * inv_det;
    if (!(t >= 0.f && t <= r.maxt)) return;
    if (t < best.t || (t == best.t && f < best.prim)) { best.t = t; best.u = u; best.v = v; best.prim = f; }
}

Hit Scene::intersect(const Ray &r, bool any_hit, bool brute) const {
    Hit best; best.t = kInf; best.u = best.v = 0.f; best.prim = 0xffffffffu;
    if (d.n_faces == 0) return best;
    if (brute) {
        for (uint32_t f = 0; f < d.n_faces; ++f) test_tri(*this, r, f, best);
        return best;
    }
    float inv[3] = { 1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z }, o[3] = { r.o.x, r.o.y, r.o.z };
    uint32_t stack[64]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const BVHNode &nd = nodes[stack[--sp]];
        float limit = fminf(best.t, r.maxt);
        float tmin = 0.f, tmax = limit;
        for (int a = 0; a < 3; ++a) {
            float t0 = (nd.lo[a] - o[a]) * inv[a], t1 = (nd.hi[a] - o[a]) * inv[a];
            tmin = fmaxf(tmin, fminf(t0, t1)); tmax = fminf(tmax, fmaxf(t0, t1));
        }
        if (!(tmin <= tmax * 1.0000005f + 1e-30f)) continue;
        if (nd.count) {
            for (uint32_t i = 0; i < nd.count; ++i) test_tri(*this, r, prim_ids[nd.left + i], best);
            if (any_hit && best.valid()) return best;
        } else { stack[sp++] = nd.left; stack[sp++] = nd.left + 1; }
    }
    return best;
}

/* src/render/mesh.cpp:1489-1659 + include/mitsuba/render/interaction.h:290-300,516-536 */
SI Scene::compute_si(const Ray &r, const Hit &h) const {
    SI si; memset((void *) &si, 0, sizeof(si));
    si.valid = h.valid();
    if (!si.valid) {                    /* interaction.h:516-536: wi = -ray.d for invalid interactions */
        si.t = kInf; si.wi = -r.d; si.shape = 0xffffffffu; si.prim = 0xffffffffu;
        return si;
    }
    uint32_t f = h.prim, shp = face_shape[f];
    const lrt_shape_desc &sd = shapes[shp];
    const uint32_t *fi = &faces[3 * f];
    auto P = [&](uint32_t i) { return V3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]); };
    auto N = [&](uint32_t i) { return V3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]); };
    V3 p0 = P(fi[0]), p1 = P(fi[1]), p2 = P(fi[2]);
    float b1 = h.u, b2 = h.v, b0 = 1.f - b1 - b2;
    si.t = h.t; si.prim = f; si.shape = shp;
    si.p = V3(fmaf(p0.x, b0, fmaf(p1.x, b1, p2.x * b2)), fmaf(p0.y, b0, fmaf(p1.y, b1, p2.y * b2)),
              fmaf(p0.z, b0, fmaf(p1.z, b1, p2.z * b2)));
    V3 dp0 = p1 - p0, dp1 = p2 - p0;
    si.n = normalize(cross(dp0, dp1));
    si.uv = { b1, b2 };
    coordinate_system(si.n, &si.dp_du, &si.dp_dv);
    if (sd.has_texcoords) {
        V2 uv0 = { texcoords[2 * fi[0]], texcoords[2 * fi[0] + 1] }, uv1 = { texcoords[2 * fi[1]], texcoords[2 * fi[1] + 1] },
           uv2 = { texcoords[2 * fi[2]], texcoords[2 * fi[2] + 1] };
        si.uv = { fmaf(uv2.x, b2, fmaf(uv1.x, b1, uv0.x * b0)), fmaf(uv2.y, b2, fmaf(uv1.y, b1, uv0.y * b0)) };
        V2 duv0 = { uv1.x - uv0.x, uv1.y - uv0.y }, duv1 = { uv2.x - uv0.x, uv2.y - uv0.y };
        float det = fmaf(duv0.x, duv1.y, -(duv0.y * duv1.x)), inv_det = rcp(det);
        if (det != 0.f) {
            /* fmsub(duv1.y, dp0, duv0.y * dp1) * inv_det ; fnmadd(duv1.x, dp0, duv0.x * dp1) * inv_det */
            si.dp_du = V3(fmaf(duv1.y, dp0.x, -(duv0.y * dp1.x)), fmaf(duv1.y, dp0.y, -(duv0.y * dp1.y)), fmaf(duv1.y, dp0.z, -(duv0.y * dp1.z))) * inv_det;
            si.dp_dv = V3(fmaf(-duv1.x, dp0.x, duv0.x * dp1.x), fmaf(-duv1.x, dp0.y, duv0.x * dp1.y), fmaf(-duv1.x, dp0.z, duv0.x * dp1.z)) * inv_det;
        }
    }
    if (sd.has_normals) {
        V3 n0 = N(fi[0]), n1 = N(fi[1]), n2 = N(fi[2]);
        V3 n(fmaf(n2.x, b2, fmaf(n1.x, b1, n0.x * b0)), fmaf(n2.y, b2, fmaf(n1.y, b1, n0.y * b0)), fmaf(n2.z, b2, fmaf(n1.z, b1, n0.z * b0)));
        float il = rsqrt(squared_norm(n));
        si.sh.n = n * il;
    } else si.sh.n = si.n;
    if (sd.flip_normals) { si.n = -si.n; si.sh.n = -si.sh.n; }
    /* initialize_sh_frame */
    si.sh.s = normalize(fma3(si.sh.n, -dot(si.sh.n, si.dp_du), si.dp_du));
    if (si.dp_du.x == 0.f && si.dp_du.y == 0.f && si.dp_du.z == 0.f) { V3 tmp; coordinate_system(si.sh.n, &si.sh.s, &tmp); }
    si.sh.t = cross(si.sh.n, si.sh.s);
    si.wi = si.sh.to_local(-r.d);
    return si;
}

/* dr::detail::estrin_impl for 10 coefficients (src/rfilters/gaussian.cpp:93-95) */
static float estrin10(float x, const float *c) {
    float a[5], x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    for (int i = 0; i < 5; ++i) a[i] = fmaf(x, c[2 * i + 1], c[2 * i]);
    float b0 = fmaf(x2, a[1], a[0]), b1 = fmaf(x2, a[3], a[2]), b2 = a[4];
    float c0 = fmaf(x4, b1, b0), c1 = b2;
    return fmaf(x8, c1, c0);
}

float Scene::rfilter_eval(float x) const {
    switch (d.film.rfilter) {
        case LRT_RFILTER_GAUSSIAN: return fmaxf(estrin10(sqr(x), rf_coeff), 0.f);
        case LRT_RFILTER_TENT: return fmaxf(0.f, 1.f - fabsf(x * rf_inv_radius));
        default: return (fabsf(x) <= 0.5f) ? 1.f : 0.f;
    }
}

void Scene::finalize() {
    /* ---- camera: include/mitsuba/render/sensor.h:234-269, include/mitsuba/core/transform.h:393-410 */
    {
        const lrt_film_desc &F = d.film; const lrt_sensor_desc &C = d.sensor;
        float fw = (float) F.width, fh = (float) F.height;
        float rsx = (float) F.crop_width / fw, rsy = (float) F.crop_height / fh;
        float rox = (float) F.crop_offset_x / fw, roy = (float) F.crop_offset_y / fh;
        float aspect = fw / fh;
        float recip = 1.f / (C.far_clip - C.near_clip);
        float tn = (float) tan((double) (C.fov_x * .5f) * (3.14159265358979323846 / 180.0)), cot = 1.f / tn;
        auto scale = [&](float x, float y, float z, M4 &m, M4 &it) { m = m4_identity(); it = m4_identity(); m.m[0] = x; m.m[5] = y; m.m[10] = z; it.m[0] = rcp(x); it.m[5] = rcp(y); it.m[10] = rcp(z); };
        auto translate = [&](float x, float y, float z, M4 &m, M4 &it) { m = m4_identity(); it = m4_identity(); m.m[3] = x; m.m[7] = y; m.m[11] = z; it.m[12] = -x; it.m[13] = -y; it.m[14] = -z; };
        M4 S1, S1i, T1, T1i, S2, S2i, T2, T2i, P, Pinv;
        scale(1.f / rsx, 1.f / rsy, 1.f, S1, S1i);
        translate(-rox, -roy, 0.f, T1, T1i);
        scale(-0.5f, -0.5f * aspect, 1.f, S2, S2i);
        translate(-1.f, -1.f / aspect, 0.f, T2, T2i);
        P = m4_identity(); P.m[0] = cot; P.m[5] = cot; P.m[10] = C.far_clip * recip; P.m[15] = 0.f;
        P.m[11] = -C.near_clip * C.far_clip * recip; P.m[14] = 1.f;
        Pinv = m4_identity(); Pinv.m[0] = tn; Pinv.m[5] = tn; Pinv.m[10] = 0.f; Pinv.m[15] = rcp(C.near_clip);
        Pinv.m[11] = 1.f; Pinv.m[14] = (C.near_clip - C.far_clip) / (C.far_clip * C.near_clip);
        M4 Pit = m4_transpose(Pinv);
        M4 it = m4_mul(m4_mul(m4_mul(m4_mul(S1i, T1i), S2i), T2i), Pit);
        sample_to_camera = m4_transpose(it);
        memcpy(cam_to_world.m, C.to_world, sizeof(float) * 16);
    }
    /* ---- acceleration + bounds (src/render/scene.cpp:49, include/mitsuba/core/bbox.h:343-346) */
    bvh_build(*this);
    {
        V3 lo(kInf), hi(-kInf);
        for (uint32_t i = 0; i < d.n_vertices; ++i) {
            V3 p(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
            lo = V3(fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z));
            hi = V3(fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z));
        }
        if (d.n_vertices) {
            V3 c = (hi + lo) * 0.5f;
            bsphere_c = c; bsphere_r = norm(c - hi);
            /* src/emitters/envmap.cpp:337-351 set_scene */
            bsphere_r = fmaxf(kRayEpsilon, bsphere_r * (1.f + kRayEpsilon));
        } else { bsphere_c = V3(0.f); bsphere_r = kRayEpsilon; }
    }
    /* ---- emitters */
    env = -1; area.assign(d.n_emitters, AreaInfo{ V3(0.f), 0.f });
    for (uint32_t e = 0; e < d.n_emitters; ++e) {
        const lrt_emitter_desc &E = emitters[e];
        if (E.type == LRT_EMITTER_AREA) {
            /* src/shapes/rectangle.cpp:108-119: frame + inverse surface area */
            const lrt_shape_desc &sd = shapes[E.shape];
            M4 tw; memcpy(tw.m, sd.to_world, sizeof(tw.m));
            V3 dp_du = xform_vec(tw, V3(2.f, 0.f, 0.f)), dp_dv = xform_vec(tw, V3(0.f, 2.f, 0.f));
            uint32_t v0 = faces[3 * sd.first_face];
            area[e].n = V3(normals[3 * v0], normals[3 * v0 + 1], normals[3 * v0 + 2]);
            area[e].inv_area = rcp(norm(cross(dp_du, dp_dv)));
        } else {
            env = (int) e;
            if (E.type == LRT_EMITTER_ENVMAP) {
                /* src/emitters/envmap.cpp:139-236: extra column, luminance * sin(theta) */
                uint32_t w = (uint32_t) E.width, h = (uint32_t) E.height;
                env_w = w + 1; env_h = h;
                env_data.assign((size_t) env_w * env_h * 3, 0.f);
                std::vector<float> lum((size_t) env_w * env_h);
                float theta_scale = 1.f / (float) (h - 1) * kPi;
                const float *in = E.data;
                for (uint32_t y = 0; y < h; ++y) {
                    float sin_theta = (float) sin((double) ((float) y * theta_scale));
                    for (uint32_t x = 0; x < w; ++x) {
                        V3 rgb(in[0], in[1], in[2]);
                        float l = fmaxf(luminance(rgb) - 0.f, 0.f);
                        lum[(size_t) y * env_w + x] = l * sin_theta;
                        float *o = &env_data[((size_t) y * env_w + x) * 3];
                        o[0] = rgb.x; o[1] = rgb.y; o[2] = rgb.z;
                        in += 3;
                    }
                    lum[(size_t) y * env_w + w] = lum[(size_t) y * env_w];
                    for (int k = 0; k < 3; ++k) env_data[((size_t) y * env_w + w) * 3 + k] = env_data[((size_t) y * env_w) * 3 + k];
                }
                env_warp.build(lum.data(), env_w, env_h);
                memcpy(env_to_world.m, E.to_world, sizeof(float) * 16);
                env_to_local = m4_inverse_affine(env_to_world);
            }
        }
    }
    /* ---- reconstruction filter (src/rfilters/gaussian.cpp:52-95, tent.cpp) */
    if (d.film.rfilter == LRT_RFILTER_GAUSSIAN) {
        float stddev = d.film.rfilter_param;
        rf_radius = 4.f * stddev;
        static const double coeff[10] = { 9.992604880e-1, -4.977025247e-1, 1.222248550e-1, -1.932406282e-2, 2.136713061e-3,
                                          -1.679873860e-4, 9.202145248e-6, -3.329417433e-7, 7.128382794e-9, -6.821193280e-11 };
        double sc = 1;
        for (int i = 0; i < 10; ++i) { rf_coeff[i] = (float) (coeff[i] * sc); sc /= (double) stddev * (double) stddev; }
        rf_coeff[0] -= estrin10(sqr(rf_radius), rf_coeff);
    } else if (d.film.rfilter == LRT_RFILTER_TENT) {
        rf_radius = d.film.rfilter_param; rf_inv_radius = 1.f / rf_radius;
    } else rf_radius = 0.5f;
    has_null_bsdf = false;
    for (auto &b : bsdfs) if (b.type == LRT_BSDF_NULL) has_null_bsdf = true;
    for (auto &sh : shapes) for (int m : { sh.interior_medium, sh.exterior_medium }) if (m >= 0) {
        if (media[m].type == LRT_MEDIUM_HETEROGENEOUS) prb_handle_null_scattering = true; else prb_nee_handle_homogeneous = true;
    }
}

} // namespace orc
