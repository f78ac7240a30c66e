// Device functions of the hip_ad_rgb hot path: BVH traversal, surface
// interactions, BSDFs, phase functions, media and emitters.  Each function
// cites the reference lines it implements (paths relative to the reference).
#pragma once
#include "device_types.h"
#include "dmath.h"
#include "../../include/liverrt.h"

namespace lrt {

#define LRT_BLOCK 256
#define LRT_STACK 32

// Read-only scene tables (shapes, BSDFs, media, ...): when every active lane of the wave asks for the same entry (one
// medium, one shape: the liver scenes), the entry comes through the scalar cache (constant address space, s_load) instead
// of one vector load per lane and dword.  The kernels never write these tables.
template <class T> DEV T tab(const T *table, uint32_t i, bool worth_a_look = true) {
    if (!worth_a_look) return table[i];                  // (a wave-uniform flag: scenes with many shapes skip the test)
    const uint32_t i0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) i);
    // (scalar loads ignore EXEC: with no active lane i0 is not an index anyone vouches for, so that case takes the masked vector path)
    if (__builtin_amdgcn_ballot_w64(true) != 0ull && __builtin_amdgcn_ballot_w64(i != i0) == 0ull)
    { T out; __builtin_memcpy(&out, reinterpret_cast<const LRT_CONST T *>((uintptr_t) table) + i0, sizeof(T)); return out; }
    return table[i];
}

struct Ray { V3 o, d; float maxt; };
struct Hit { float t, u, v; uint32_t prim; uint32_t slot = 0xffffffffu; };   // slot: the LDS tracer's triangle slot (vertex indices without a global load), else none
struct SI { bool valid; float t; V3 p, n; Frame sh; V2 uv; V3 dp_du, dp_dv, wi; uint32_t prim, shape; };

// ------------------------------------------------------------- traversal
DEV float slab_rcp(float x) { return fmin_(fmax_(__builtin_amdgcn_rcpf(x), -1e20f), 1e20f); }
// Moeller-Trumbore on a pre-gathered triangle slot (include/mitsuba/render/mesh.h:506-527).
// Ties are resolved towards the lower primitive index so that the result does
// not depend on the traversal order.
DEV void test_tri(const float4 *__restrict__ tris, uint32_t slot, V3 o, V3 d, float maxt, Hit &best) {
    float4 a = tris[3 * slot], b = tris[3 * slot + 1], c = tris[3 * slot + 2];
    V3 p0(a.x, a.y, a.z), e1(b.x, b.y, b.z), e2(c.x, c.y, c.z);
    uint32_t f = f2u(a.w);
    V3 pvec = cross(d, e2);
    float inv_det = rcp(dot(e1, pvec));
    V3 tvec = o - p0;
    float u = dot(tvec, pvec) * inv_det;
    if (!(u >= 0.f && u <= 1.f)) return;
    V3 qvec = cross(tvec, e1);
    float v = dot(d, qvec) * inv_det;
    if (!(v >= 0.f && u + v <= 1.f)) return;
    float t = dot(e2, qvec) * inv_det;
    if (!(t >= 0.f && t <= maxt)) return;
    if (t < best.t || (t == best.t && f < best.prim)) { best.t = t; best.u = u; best.v = v; best.prim = f; }
}

// Stack-based BVH2 traversal; the per-lane stack lives in LDS, laid out
// [entry][thread] so that a wave's pushes/pops hit 64 consecutive banks.
template <bool ANY_HIT>
DEV Hit trace(SceneRef sc, const Ray &r, int *__restrict__ stack /* &lds[threadIdx.x] */) {
    Hit best; best.t = kInf; best.u = best.v = 0.f; best.prim = 0xffffffffu;
    if (sc.n_faces == 0) return best;
    const V3 o = r.o, d = r.d;
    if (sc.root_is_leaf) {
        for (uint32_t i = 0; i < sc.root_leaf_count; ++i) test_tri(sc.tris, sc.root_leaf_first + i, o, d, r.maxt, best);
        return best;
    }
    // "while-while" traversal (see trace_lds below for the rationale): descend to the next leaf with box tests only, then
    // test the leaf's triangles.  Work items: inner node index (< 0x7fffffff), or 0x80000000 | first slot; a leaf
    // ends at the slot whose e1.w is non-zero (bvh.cpp).
    // The slab arithmetic only culls (padded boxes, hits decided by the exact triangle tests and the tie rule).
    // reciprocals clamped to +-1e20: a direction component of (almost) zero must give (n - o) * huge = -+huge per plane, not
    // inf - inf = NaN, so that rays parallel to a slab are kept or culled by the side of the origin (the padding decides ties)
    const float ix = slab_rcp(d.x), iy = slab_rcp(d.y), iz = slab_rcp(d.z);
    const float ox = -o.x * ix, oy = -o.y * iy, oz = -o.z * iz;
    const f32x2 vix = { ix, ix }, viy = { iy, iy }, viz = { iz, iz }, vox = { ox, ox }, voy = { oy, oy }, voz = { oz, oz };
    const uint32_t DONE = 0x7fffffffu;
    int sp = 0; uint32_t cur = 0;
    for (;;) {
        while (cur < DONE) {
            const float4 *nd = sc.nodes + 4 * (size_t) cur;
            float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
            float limit = fmin_(best.t, r.maxt);
            // twelve slab planes as six packed fmas (v_pk_fma_f32: both halves in one issue slot; IEEE fma per element)
            const f32x2 vax = pk_fma(f32x2{ n0.x, n0.y }, vix, vox), vay = pk_fma(f32x2{ n0.z, n0.w }, viy, voy), vaz = pk_fma(f32x2{ n2.x, n2.y }, viz, voz);
            const f32x2 vbx = pk_fma(f32x2{ n1.x, n1.y }, vix, vox), vby = pk_fma(f32x2{ n1.z, n1.w }, viy, voy), vbz = pk_fma(f32x2{ n2.z, n2.w }, viz, voz);
            const float ax0 = vax.x, ax1 = vax.y, ay0 = vay.x, ay1 = vay.y, az0 = vaz.x, az1 = vaz.y;
            const float bx0 = vbx.x, bx1 = vbx.y, by0 = vby.x, by1 = vby.y, bz0 = vbz.x, bz1 = vbz.y;
            float tmin0 = fmax_(fmax_(fmin_(ax0, ax1), fmin_(ay0, ay1)), fmax_(fmin_(az0, az1), 0.f));
            float tmax0 = fmin_(fmin_(fmax_(ax0, ax1), fmax_(ay0, ay1)), fmin_(fmax_(az0, az1), limit));
            float tmin1 = fmax_(fmax_(fmin_(bx0, bx1), fmin_(by0, by1)), fmax_(fmin_(bz0, bz1), 0.f));
            float tmax1 = fmin_(fmin_(fmax_(bx0, bx1), fmax_(by0, by1)), fmin_(fmax_(bz0, bz1), limit));
            bool h0 = tmin0 <= tmax0 * 1.000002f, h1 = tmin1 <= tmax1 * 1.000002f;      // (tmin >= 0, so an additive floor on the right-hand side decides nothing)
            int r0 = (int) f2u(n3.x), r1 = (int) f2u(n3.y);
            uint32_t c0 = r0 < 0 ? (0x80000000u | (uint32_t) ~r0) : (uint32_t) r0;
            uint32_t c1 = r1 < 0 ? (0x80000000u | (uint32_t) ~r1) : (uint32_t) r1;
            if (h0 && h1) {
                bool swap = tmin1 < tmin0;
                stack[sp * LRT_BLOCK] = (int) (swap ? c0 : c1); ++sp;
                cur = swap ? c1 : c0;
            } else if (h0 || h1) cur = h0 ? c0 : c1;                    // (two flat predicated regions instead of these nested three: measured, +0.4 % time)
            else if (sp == 0) cur = DONE;
            else { --sp; cur = (uint32_t) stack[sp * LRT_BLOCK]; }
        }
        if (cur == DONE) break;
        uint32_t slot = cur & 0x7fffffffu; bool last;
        do { last = sc.tris[3 * slot + 1].w != 0.f; test_tri(sc.tris, slot, o, d, r.maxt, best); ++slot; } while (!last);
        if (ANY_HIT && best.prim != 0xffffffffu) return best;
        if (sp == 0) break;
        --sp; cur = (uint32_t) stack[sp * LRT_BLOCK];
    }
    return best;
}

// Ray-query back-ends the integrator loops are templated on.
struct GlobalTracer {                      // BVH in global memory (any scene size), 32-bit stack entries in LDS
    SceneRef sc; int *stack;
    DEV Hit closest(const Ray &r) const { return trace<false>(sc, r, stack); }
    DEV Hit any(const Ray &r) const { return trace<true>(sc, r, stack); }
    DEV SI surface(SceneRef s, const Ray &r, const Hit &h) const;
};

// Whole BVH resident in LDS (scenes whose image fits next to the traversal stacks: the liver meshes and the Cornell
// box do): nodes as in bvh.h, vertices padded to float4, triangle slots as 4 x u16 vertex indices, 16-bit stack
// entries.  Edge vectors are formed in the kernel with the same float subtractions the host builder uses, so hits
// are bit-identical to the global-memory path.
struct LdsScene {
    const float4 *nodes; const float4 *verts; const uint2 *tris;
    uint32_t n_faces, root_is_leaf, root_first, root_count;
};
#define LRT_LDS_BLOCK_MAX 1024

// `ix` = the slot's index words: three 16-bit vertex indices, then (face index << 1 | last-slot-of-the-leaf flag)
// While a traversal runs, best.prim = face index | slot << 16 (both < 0x8000; 0xffffffff = no hit yet).
DEV void test_tri_lds(const LdsScene &L, uint2 ix, uint32_t slot, V3 o, V3 d, float maxt, Hit &best) {
    float4 a = L.verts[ix.x & 0xffffu], b = L.verts[ix.x >> 16], c = L.verts[ix.y & 0xffffu];
    V3 p0(a.x, a.y, a.z), e1(b.x - a.x, b.y - a.y, b.z - a.z), e2(c.x - a.x, c.y - a.y, c.z - a.z);
    V3 pvec = cross(d, e2);
    float inv_det = rcp(dot(e1, pvec));
    V3 tvec = o - p0;
    float u = dot(tvec, pvec) * inv_det;
    if (!(u >= 0.f && u <= 1.f)) return;
    V3 qvec = cross(tvec, e1);
    float v = dot(d, qvec) * inv_det;
    if (!(v >= 0.f && u + v <= 1.f)) return;
    float t = dot(e2, qvec) * inv_det;
    if (!(t >= 0.f && t <= maxt)) return;
    if (t > best.t) return;
    uint32_t f = ix.y >> 17;
    if (t < best.t || f < (best.prim & 0x7fffu)) { best.t = t; best.u = u; best.v = v; best.prim = f | (slot << 16); }
}

#ifdef LRT_TRAV_STATS
__device__ unsigned long long g_trav[8];            // developer counters: [ANY ? 4 : 0] + { node-loop wave iterations, active lanes in them, leaf-loop wave iterations, active lanes }
#define LRT_TRAV_COUNT(k) { const unsigned long long m_ = __ballot(true); if ((threadIdx.x & 63u) == (uint32_t) (__ffsll((long long) m_) - 1)) { atomicAdd(&g_trav[(ANY_HIT ? 4 : 0) + k], 1ull); atomicAdd(&g_trav[(ANY_HIT ? 4 : 0) + k + 1], (unsigned long long) __popcll(m_)); } }
#else
#define LRT_TRAV_COUNT(k)
#endif
template <bool ANY_HIT, int STRIDE>
DEV Hit trace_lds(const LdsScene &L, const Ray &r, uint16_t *__restrict__ stack /* &lds_stack[threadIdx.x] */) {
    Hit best; best.t = kInf; best.u = best.v = 0.f; best.prim = 0xffffffffu;
    if (L.n_faces == 0) return best;
    const V3 o = r.o, d = r.d;
    if (L.root_is_leaf) {
        for (uint32_t i = 0; i < L.root_count; ++i) test_tri_lds(L, L.tris[L.root_first + i], L.root_first + i, o, d, r.maxt, best);
        if (best.prim != 0xffffffffu) { best.slot = best.prim >> 16; best.prim &= 0x7fffu; }
        return best;
    }
    // The slab arithmetic only culls (hits are decided by the Moeller-Trumbore tests and the tie rule), so it may differ
    // from the oracle's: approximate reciprocal, one fma per plane, 3-input min/max.  The padded boxes absorb the error.
    // "while-while" traversal: every lane first descends to its next leaf (inner loop: box tests only), then the lanes
    // test their leaf triangles together; this keeps far more lanes busy than testing leaves where they are met.
    // Work items are 16-bit: inner node index (< 0x8000) or 0x8000 | first triangle slot; a leaf ends at the slot whose
    // index word carries the "last" flag (device.hip).
    // reciprocals clamped to +-1e20: a direction component of (almost) zero must give (n - o) * huge = -+huge per plane, not
    // inf - inf = NaN, so that rays parallel to a slab are kept or culled by the side of the origin (the padding decides ties)
    const float ix = slab_rcp(d.x), iy = slab_rcp(d.y), iz = slab_rcp(d.z);
    const float ox = -o.x * ix, oy = -o.y * iy, oz = -o.z * iz;
    const f32x2 vix = { ix, ix }, viy = { iy, iy }, viz = { iz, iz }, vox = { ox, ox }, voy = { oy, oy }, voz = { oz, oz };
    const uint32_t DONE = 0x10000u;
    int sp = 0; uint32_t cur = 0;
    for (;;) {
        while (cur < 0x8000u) {
            LRT_TRAV_COUNT(0)
            const float4 *nd = L.nodes + 4 * cur;
            float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
            float limit = fmin_(best.t, r.maxt);
            // twelve slab planes as six packed fmas (v_pk_fma_f32: both halves in one issue slot; IEEE fma per element)
            const f32x2 vax = pk_fma(f32x2{ n0.x, n0.y }, vix, vox), vay = pk_fma(f32x2{ n0.z, n0.w }, viy, voy), vaz = pk_fma(f32x2{ n2.x, n2.y }, viz, voz);
            const f32x2 vbx = pk_fma(f32x2{ n1.x, n1.y }, vix, vox), vby = pk_fma(f32x2{ n1.z, n1.w }, viy, voy), vbz = pk_fma(f32x2{ n2.z, n2.w }, viz, voz);
            const float ax0 = vax.x, ax1 = vax.y, ay0 = vay.x, ay1 = vay.y, az0 = vaz.x, az1 = vaz.y;
            const float bx0 = vbx.x, bx1 = vbx.y, by0 = vby.x, by1 = vby.y, bz0 = vbz.x, bz1 = vbz.y;
            float tmin0 = fmax_(fmax_(fmin_(ax0, ax1), fmin_(ay0, ay1)), fmax_(fmin_(az0, az1), 0.f));
            float tmax0 = fmin_(fmin_(fmax_(ax0, ax1), fmax_(ay0, ay1)), fmin_(fmax_(az0, az1), limit));
            float tmin1 = fmax_(fmax_(fmin_(bx0, bx1), fmin_(by0, by1)), fmax_(fmin_(bz0, bz1), 0.f));
            float tmax1 = fmin_(fmin_(fmax_(bx0, bx1), fmax_(by0, by1)), fmin_(fmax_(bz0, bz1), limit));
            bool h0 = tmin0 <= tmax0 * 1.000002f, h1 = tmin1 <= tmax1 * 1.000002f;      // (tmin >= 0, so an additive floor on the right-hand side decides nothing)
            const uint32_t c0 = f2u(n3.x), c1 = f2u(n3.y);             // already work items (device.hip builds the LDS image)
            if (h0 && h1) {
                bool swap = tmin1 < tmin0;
                stack[sp * STRIDE] = (uint16_t) (swap ? c0 : c1); ++sp;
                cur = swap ? c1 : c0;
            } else if (h0 || h1) cur = h0 ? c0 : c1;
            else if (sp == 0) cur = DONE;
            else { --sp; cur = stack[sp * STRIDE]; }
        }
        if (cur == DONE) break;
        uint32_t slot = cur & 0x7fffu, last;
        do {
            LRT_TRAV_COUNT(2)
            const uint2 ix = L.tris[slot];
            last = (ix.y >> 16) & 1u;
            test_tri_lds(L, ix, slot, o, d, r.maxt, best);
            ++slot;
        } while (!last);
        if (ANY_HIT && best.prim != 0xffffffffu) return best;
        if (sp == 0) break;
        --sp; cur = stack[sp * STRIDE];
    }
    if (best.prim != 0xffffffffu) { best.slot = best.prim >> 16; best.prim &= 0x7fffu; }
    return best;
}

template <int STRIDE>
struct LdsTracer {
    const LdsScene &L; uint16_t *stack;
    DEV Hit closest(const Ray &r) const { return trace_lds<false, STRIDE>(L, r, stack); }
    DEV Hit any(const Ray &r) const { return trace_lds<true, STRIDE>(L, r, stack); }
    DEV SI surface(SceneRef s, const Ray &r, const Hit &h) const;
};

// --------------------------------------------------- conservative distance field
// Lower bound of the distance from p to the nearest triangle (0: unknown / outside the grid).  |p - centre| is subtracted
// exactly (1-Lipschitz), the stored value already carries the safety margins (device.hip: build_dist_grid).
DEV float dist_grid_lower_bound(GridRef g, V3 p) {
    // D(c) - |p - c| bounds the distance for ANY p, so the cell index is simply clamped into the grid (a point outside the
    // grid gets a small or negative bound; NaN coordinates give NaN, which proves nothing).
    float fx = (p.x - g.lo[0]) * g.inv_cell, fy = (p.y - g.lo[1]) * g.inv_cell, fz = (p.z - g.lo[2]) * g.inv_cell;
    int ix = min(max((int) fx, 0), g.n[0] - 1), iy = min(max((int) fy, 0), g.n[1] - 1), iz = min(max((int) fz, 0), g.n[2] - 1);
    union { uint16_t u; _Float16 h; } cv; cv.u = g.d[(uint32_t) ((iz * g.n[1] + iy) * g.n[0] + ix)];
    float D = (float) cv.h;
    float cx = fx - ((float) ix + .5f), cy = fy - ((float) iy + .5f), cz = fz - ((float) iz + .5f);
    return D - __builtin_amdgcn_sqrtf(cx * cx + cy * cy + cz * cz) * (g.cell * 1.001f);   // 1-ulp hardware sqrt: inside the margins
}

// True when the segment o + t d, t in [0, maxt], provably meets no triangle: a few sphere-tracing steps through the
// distance field.  Margins: 1 % on the segment length, every advance counted 0.5 % short (f32 error of the
// Moller-Trumbore t is ~1e-6 relative away from grazing incidence).  False means "unknown": run the ray query.
#ifndef LRT_GRID_STEPS
#define LRT_GRID_STEPS 3
#endif
template <int STEPS = LRT_GRID_STEPS>
DEV bool segment_proven_empty(GridRef g, V3 o, V3 d, float maxt) {
    if (!g.enabled || !(maxt < 1e30f)) return false;
    float len = __builtin_amdgcn_sqrtf(dot(d, d));                       // hardware sqrt / rcp (1 ulp): inside the margins
    float remaining = maxt * len * 1.01f, inv_len = __builtin_amdgcn_rcpf(len);
    V3 p = o;
    for (int k = 0; k < STEPS; ++k) {
        float lb = dist_grid_lower_bound(g, p);
        if (remaining < lb) return true;
        if (!(lb > .5f * g.cell)) return false;
        float adv = lb * .99f;
        p = p + d * (adv * inv_len);
        remaining -= adv * .995f;
    }
    return false;
}

// The same question answered with K INDEPENDENT lookups instead of a chain of dependent ones: lower bounds at K points spread over the
// segment; the segment is free of surfaces when the balls around those points (radius = the bound) cover it end to end.  One memory round
// trip instead of three (the dependent lookups were the largest stall of a medium trip: section timers of the proven-free tiles), and the
// cover test uses the bounds actually found, so one large ball in the middle can do it alone.  Same margins as above.
template <int K>
DEV bool segment_covered_by_balls(GridRef g, V3 o, V3 d, float maxt) {
    if (!g.enabled || !(maxt < 1e30f)) return false;
    const float len = __builtin_amdgcn_sqrtf(dot(d, d));                 // hardware sqrt (1 ulp): inside the margins
    const float L = maxt * len * 1.01f;                                  // the segment, 1 % long
#define LRT_FRAC(k) (((float) (k) + .5f) / (float) K)          /* K = 2: at 1/4 and 3/4 (0.2 / 0.7, 0.3 / 0.8, 0.15 / 0.6 measured: no better) */
    float lb[K];
#pragma unroll
    for (int k = 0; k < K; ++k) lb[k] = dist_grid_lower_bound(g, fma3(d, maxt * LRT_FRAC(k), o)) * .995f;
    float covered = 0.f; bool ok = true;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float c = L * LRT_FRAC(k);                                // the point's position along the (lengthened) segment
        ok = ok && (lb[k] > 0.f) && (c - lb[k] <= covered);
        covered = fmax_(covered, c + lb[k]);
    }
    return ok && covered >= L;
#undef LRT_FRAC
}

// What the kernels call.  Two balls: measured on C3 against the three dependent steps above: 4.6 % more segments proven (20.35 M
// proven-free tiles per launch against 19.45 M), every medium tile 1 - 2 % shorter, launch -2.3 ... -3.5 %; 3, 4 and 6 balls prove more
// still but the gathers cost more than they save (226 / 231 / 242 ms against 222), one ball at the midpoint proves less (229).
// LRT_SPHERE_TRACE (build switch) brings the dependent steps back.
DEV bool segment_free_of_surfaces(GridRef g, V3 o, V3 d, float maxt) {
#ifdef LRT_SPHERE_TRACE
    return segment_proven_empty(g, o, d, maxt);
#else
    return segment_covered_by_balls<2>(g, o, d, maxt);
#endif
}

// --------------------------------------------------- surface interaction
// src/render/mesh.cpp:1489-1659 + include/mitsuba/render/interaction.h:290-300,516-536
// `L` (LDS tracer only): vertex indices and positions of the hit triangle come from the LDS image (same values as the
// global arrays), which takes the index loads out of the dependent chain of global loads.
DEV SI compute_si(SceneRef sc, const Ray &r, const Hit &h, const LdsScene *L = nullptr) {
    const bool valid = h.prim != 0xffffffffu;
    // every field is written through plain locals (no member addresses escape): keeps the record in VGPRs
    float t = kInf; V3 p(0.f), n(0.f), shn(0.f), shs(0.f), sht(0.f), dp_du(0.f), dp_dv(0.f), wi = -r.d;
    V2 uv = { 0.f, 0.f }; uint32_t f = 0xffffffffu, shp = 0xffffffffu;
    if (valid) {
        f = h.prim; shp = 0u; if (!sc.one_shape) shp = sc.face_shape[f];
        const DShape sd = tab(sc.shapes, shp, sc.one_shape);
        uint32_t i0, i1, i2; V3 p0, p1, p2;
        if (L && h.slot != 0xffffffffu) {
            const uint2 ix = L->tris[h.slot];
            i0 = ix.x & 0xffffu; i1 = ix.x >> 16; i2 = ix.y & 0xffffu;
            const float4 a = L->verts[i0], b = L->verts[i1], c = L->verts[i2];
            p0 = V3(a.x, a.y, a.z); p1 = V3(b.x, b.y, b.z); p2 = V3(c.x, c.y, c.z);
        } else {
            i0 = sc.faces[3 * f]; i1 = sc.faces[3 * f + 1]; i2 = sc.faces[3 * f + 2];
            p0 = V3(sc.positions[3 * i0], sc.positions[3 * i0 + 1], sc.positions[3 * i0 + 2]);
            p1 = V3(sc.positions[3 * i1], sc.positions[3 * i1 + 1], sc.positions[3 * i1 + 2]);
            p2 = V3(sc.positions[3 * i2], sc.positions[3 * i2 + 1], sc.positions[3 * i2 + 2]);
        }
        float b1 = h.u, b2 = h.v, b0 = 1.f - b1 - b2;
        t = h.t;
        p = V3(fma_(p0.x, b0, fma_(p1.x, b1, p2.x * b2)), fma_(p0.y, b0, fma_(p1.y, b1, p2.y * b2)), fma_(p0.z, b0, fma_(p1.z, b1, p2.z * b2)));
        V3 dp0 = p1 - p0, dp1 = p2 - p0;
        n = normalize(cross(dp0, dp1));
        uv = { b1, b2 };
        Basis bs = coordinate_system(n);
        dp_du = bs.s; dp_dv = bs.t;
        if (sd.has_texcoords) {
            const float2 t0 = *reinterpret_cast<const float2 *>(sc.vattr + 2 * (size_t) i0 + 1), t1 = *reinterpret_cast<const float2 *>(sc.vattr + 2 * (size_t) i1 + 1),
                         t2 = *reinterpret_cast<const float2 *>(sc.vattr + 2 * (size_t) i2 + 1);
            V2 uv0 = { t0.x, t0.y }, uv1 = { t1.x, t1.y }, uv2 = { t2.x, t2.y };
            uv = { fma_(uv2.x, b2, fma_(uv1.x, b1, uv0.x * b0)), fma_(uv2.y, b2, fma_(uv1.y, b1, uv0.y * b0)) };
            V2 duv0 = { uv1.x - uv0.x, uv1.y - uv0.y }, duv1 = { uv2.x - uv0.x, uv2.y - uv0.y };
            float det = fma_(duv0.x, duv1.y, -(duv0.y * duv1.x)), inv_det = rcp(det);
            if (det != 0.f) {
                dp_du = V3(fma_(duv1.y, dp0.x, -(duv0.y * dp1.x)), fma_(duv1.y, dp0.y, -(duv0.y * dp1.y)), fma_(duv1.y, dp0.z, -(duv0.y * dp1.z))) * inv_det;
                dp_dv = V3(fma_(-duv1.x, dp0.x, duv0.x * dp1.x), fma_(-duv1.x, dp0.y, duv0.x * dp1.y), fma_(-duv1.x, dp0.z, duv0.x * dp1.z)) * inv_det;
            }
        }
        if (sd.has_normals) {
            const float4 a0 = sc.vattr[2 * (size_t) i0], a1 = sc.vattr[2 * (size_t) i1], a2 = sc.vattr[2 * (size_t) i2];
            V3 n0(a0.x, a0.y, a0.z), n1(a1.x, a1.y, a1.z), n2(a2.x, a2.y, a2.z);
            V3 ni(fma_(n2.x, b2, fma_(n1.x, b1, n0.x * b0)), fma_(n2.y, b2, fma_(n1.y, b1, n0.y * b0)), fma_(n2.z, b2, fma_(n1.z, b1, n0.z * b0)));
            float il = rsqrt_(squared_norm(ni));
            shn = ni * il;
        } else shn = n;
        if (sd.flip_normals) { n = V3(-n.x, -n.y, -n.z); shn = V3(-shn.x, -shn.y, -shn.z); }
        shs = normalize(fma3(shn, -dot(shn, dp_du), dp_du));
        if (dp_du.x == 0.f && dp_du.y == 0.f && dp_du.z == 0.f) shs = coordinate_system(shn).s;
        sht = cross(shn, shs);
        V3 md(-r.d.x, -r.d.y, -r.d.z);
        wi = V3(dot(md, shs), dot(md, sht), dot(md, shn));
    }
    SI si;
    si.valid = valid; si.t = t; si.p = p; si.n = n; si.sh.s = shs; si.sh.t = sht; si.sh.n = shn; si.uv = uv;
    si.dp_du = dp_du; si.dp_dv = dp_dv; si.wi = wi; si.prim = f; si.shape = shp;
    return si;
}

DEV SI GlobalTracer::surface(SceneRef s, const Ray &r, const Hit &h) const { return compute_si(s, r, h); }
template <int STRIDE> DEV SI LdsTracer<STRIDE>::surface(SceneRef s, const Ray &r, const Hit &h) const { return compute_si(s, r, h, &L); }

// include/mitsuba/render/interaction.h:140-168
DEV V3 offset_p(V3 p, V3 n, V3 d) {
    float mag = (1.f + max3(abs3(p))) * kRayEpsilon;
    mag = mulsign(mag, dot(n, d));
    return fma3(n, mag, p);
}
DEV Ray spawn_ray(V3 p, V3 n, V3 d) { Ray r; r.o = offset_p(p, n, d); r.d = d; r.maxt = kLargest; return r; }
DEV Ray spawn_ray_to(V3 p, V3 n, V3 t) {
    Ray r; r.o = offset_p(p, n, t - p);
    V3 d = t - r.o; float dist = norm(d);
    r.d = d / dist; r.maxt = dist * (1.f - kShadowEpsilon);
    return r;
}

// --------------------------------------------------------------- warping
// include/mitsuba/core/warp.h:54-92, :412-420, :250-256
DEV V2 square_to_uniform_disk_concentric(float sx, float sy) {
    float x = fma_(2.f, sx, -1.f), y = fma_(2.f, sy, -1.f);
    bool is_zero = (x == 0.f) && (y == 0.f), q13 = __builtin_fabsf(x) < __builtin_fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * kPi * rp / r;
    if (q13) phi = 0.5f * kPi - phi;
    if (is_zero) phi = 0.f;
    float s, c; m_sincos(phi, &s, &c);
    return { r * c, r * s };
}
DEV V3 square_to_cosine_hemisphere(float sx, float sy) {
    V2 p = square_to_uniform_disk_concentric(sx, sy);
    float z = safe_sqrt(1.f - fma_(p.x, p.x, p.y * p.y));
    return V3(p.x, p.y, z);
}
DEV V3 square_to_uniform_sphere(float sx, float sy) {
    float z = fma_(-2.f, sy, 1.f), r = safe_sqrt(fma_(-z, z, 1.f));
    float s, c; m_sincos(2.f * kPi * sx, &s, &c);
    return V3(r * c, r * s, z);
}

// -------------------------------------------------------------- textures
DEV V3 tex_eval(SceneRef sc, int tex, const SI &si) {
    const DTexture &T = sc.textures[tex];
    if (T.type == LRT_TEX_CHECKERBOARD) {        // src/textures/checkerboard.cpp:70-88
        float u = fma_(T.to_uv[1], si.uv.y, fma_(T.to_uv[0], si.uv.x, T.to_uv[2]));
        float v = fma_(T.to_uv[4], si.uv.y, fma_(T.to_uv[3], si.uv.x, T.to_uv[5]));
        bool mx = u - __builtin_floorf(u) > .5f, my = v - __builtin_floorf(v) > .5f;
        return (mx == my) ? V3(T.color0[0], T.color0[1], T.color0[2]) : V3(T.color1[0], T.color1[1], T.color1[2]);
    }
    if (T.type == LRT_TEX_RGB) return V3(T.color0[0], T.color0[1], T.color0[2]);
    return V3(0.f);
}

// src/textures/bitmap.cpp:509-578 eval_1_grad; texels hold the per-texel
// luminance (precomputed on the host with the same float expression).
DEV V2 tex_eval_1_grad(SceneRef sc, int tex, const SI &si) {
    const DTexture &T = sc.textures[tex];
    if (T.type != LRT_TEX_BITMAP) return { 0.f, 0.f };
    float u = fma_(T.to_uv[1], si.uv.y, fma_(T.to_uv[0], si.uv.x, T.to_uv[2]));
    float v = fma_(T.to_uv[4], si.uv.y, fma_(T.to_uv[3], si.uv.x, T.to_uv[5]));
    int w = T.width, h = T.height;
    float fx = fma_(u, (float) w, -0.5f), fy = fma_(v, (float) h, -0.5f);
    int ix = (int) __builtin_floorf(fx), iy = (int) __builtin_floorf(fy);
    float w1x = fx - (float) ix, w1y = fy - (float) iy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    int x0 = ix % w; if (x0 < 0) x0 += w; int x1 = (ix + 1) % w; if (x1 < 0) x1 += w;
    int y0 = iy % h; if (y0 < 0) y0 += h; int y1 = (iy + 1) % h; if (y1 < 0) y1 += h;
    const float *d = sc.tex_data + T.data_offset;
    float f00 = d[y0 * w + x0], f10 = d[y0 * w + x1], f01 = d[y1 * w + x0], f11 = d[y1 * w + x1];
    float dx = fma_(w0y, f10 - f00, w1y * (f11 - f01)), dy = fma_(w0x, f01 - f00, w1x * (f11 - f10));
    float du = T.to_uv[0] * dx + T.to_uv[3] * dy, dv = T.to_uv[1] * dx + T.to_uv[4] * dy;
    return { (float) w * du, (float) h * dv };
}

// ----------------------------------------------------------------- BSDFs
enum { F_DELTA = 1, F_SMOOTH = 2, F_NULL = 4 };
struct BSDFSample { V3 wo; float pdf, eta; int type; V3 weight; };

// include/mitsuba/render/fresnel.h:35-73
DEV void fresnel(float cos_theta_i, float eta, float *r, float *cos_theta_t, float *eta_it, float *eta_ti) {
    bool outside = cos_theta_i >= 0.f;
    float rcp_eta = rcp(eta);
    *eta_it = outside ? eta : rcp_eta; *eta_ti = outside ? rcp_eta : eta;
    float cos_theta_t_sqr = fma_(-fma_(-cos_theta_i, cos_theta_i, 1.f), *eta_ti * *eta_ti, 1.f);
    float cti = __builtin_fabsf(cos_theta_i), ctt = safe_sqrt(cos_theta_t_sqr);
    bool index_matched = eta == 1.f, special = index_matched || cti == 0.f;
    float r_sc = index_matched ? 0.f : 1.f;
    float a_s = fma_(-*eta_it, ctt, cti) / fma_(*eta_it, ctt, cti);
    float a_p = fma_(-*eta_it, cti, ctt) / fma_(*eta_it, cti, ctt);
    float rr = 0.5f * (sqr(a_s) + sqr(a_p));
    if (special) rr = r_sc;
    *r = rr; *cos_theta_t = mulsign_neg(ctt, cos_theta_i);
}

// src/bsdfs/bumpmap.cpp:226-251
DEV Frame bump_frame(SceneRef sc, const DBsdf &B, const SI &si) {
    V2 g = tex_eval_1_grad(sc, B.texture, si);
    float gx = B.scale * g.x, gy = B.scale * g.y;
    V3 dp_du = fma3(si.sh.n, gx - dot(si.sh.n, si.dp_du), si.dp_du);
    V3 dp_dv = fma3(si.sh.n, gy - dot(si.sh.n, si.dp_dv), si.dp_dv);
    Frame r;
    r.n = normalize(cross(dp_du, dp_dv));
    if (dot(si.n, r.n) < 0.f) r.n = r.n * -1.f;
    r.n = si.sh.to_local(r.n);
    if (si.wi.z * dot(si.wi, r.n) <= 0.f) r.n = V3(-r.n.x, -r.n.y, r.n.z);
    r.s = normalize(fma3(r.n, -dot(r.n, si.dp_du), si.dp_du));
    r.t = cross(r.n, r.s);
    return r;
}
DEV float tan_theta_2(V3 v) { float t = fma_(-v.z, v.z, 1.f); return fmax_(t, 0.f) / sqr(v.z); }
DEV float shadow_terminator(V3 pn, V3 wo) {       // src/bsdfs/normalmap_helpers.h:20-25
    float alpha2 = fmin_(0.125f * tan_theta_2(pn), 1.f);
    return 2.f / (1.f + __builtin_sqrtf(1.f + alpha2 * tan_theta_2(wo)));
}

// Leaf BSDFs (diffuse / dielectric / null), evaluated in the frame `wi` is given in.
DEV BSDFSample leaf_sample(SceneRef sc, const DBsdf &B, const SI &si, V3 wi, float s1, float s2x, float s2y) {
    BSDFSample bs;
    bs.wo = V3(0.f); bs.pdf = 0.f; bs.eta = 0.f; bs.type = 0; bs.weight = V3(0.f);       // dr::zeros<BSDFSample3f>
    if (B.type == LRT_BSDF_DIFFUSE) {               // src/bsdfs/diffuse.cpp sample()
        if (wi.z > 0.f) {
            bs.wo = square_to_cosine_hemisphere(s2x, s2y);
            bs.pdf = kInvPi * bs.wo.z; bs.eta = 1.f; bs.type = F_SMOOTH;
            if (bs.pdf > 0.f) bs.weight = tex_eval(sc, B.reflectance, si);
        }
    } else if (B.type == LRT_BSDF_DIELECTRIC) {     // src/bsdfs/dielectric.cpp:230-367
        float r_i, ctt, eta_it, eta_ti;
        fresnel(wi.z, B.eta, &r_i, &ctt, &eta_it, &eta_ti);
        float t_i = 1.f - r_i;
        bool sel_r = s1 <= r_i;
        bs.pdf = sel_r ? r_i : t_i;
        bs.type = F_DELTA;
        bs.wo = sel_r ? V3(-wi.x, -wi.y, wi.z) : V3(-eta_ti * wi.x, -eta_ti * wi.y, ctt);
        bs.eta = sel_r ? 1.f : eta_it;
        bs.weight = V3(1.f);
        if (!sel_r) bs.weight = bs.weight * sqr(eta_ti);
    } else {                                       // src/bsdfs/null.cpp sample()
        // ROCm 7.2 / gfx950 code generation drops the negated z component when this branch's values meet the
        // other branches' phis (reproducer: scripts/dbg/t_null.hip); the empty asm keeps them in their own VGPRs.
        float nx = -wi.x, ny = -wi.y, nz = -wi.z;
        asm volatile("" : "+v"(nx), "+v"(ny), "+v"(nz));
        bs.wo = V3(nx, ny, nz); bs.type = F_NULL; bs.eta = 1.f; bs.pdf = 1.f; bs.weight = V3(1.f);
    }
    return bs;
}
DEV V3 leaf_eval(SceneRef sc, const DBsdf &B, const SI &si, V3 wi, V3 wo) {
    if (B.type == LRT_BSDF_DIFFUSE) {
        if (!(wi.z > 0.f && wo.z > 0.f)) return V3(0.f);
        return tex_eval(sc, B.reflectance, si) * kInvPi * wo.z;
    }
    return V3(0.f);
}
DEV float leaf_pdf(const DBsdf &B, V3 wi, V3 wo) {
    if (B.type == LRT_BSDF_DIFFUSE) return (wi.z > 0.f && wo.z > 0.f) ? kInvPi * wo.z : 0.f;
    return 0.f;
}

DEV BSDFSample bsdf_sample(SceneRef sc, int b, const SI &si, float s1, float s2x, float s2y) {
    const DBsdf B = tab(sc.bsdfs, b, sc.one_shape);
    if (B.type == LRT_BSDF_BUMPMAP) {               // src/bsdfs/bumpmap.cpp:138-162
        Frame pf = bump_frame(sc, B, si);
        V3 pwi = pf.to_local(si.wi);
        BSDFSample bs = leaf_sample(sc, tab(sc.bsdfs, B.nested, sc.one_shape), si, pwi, s1, s2x, s2y);
        bool active = any_nonzero(bs.weight);
        V3 pwo = pf.to_world(bs.wo);
        active = active && (bs.wo.z * pwo.z > 0.f);
        bs.wo = pwo;
        V3 w = bs.weight * shadow_terminator(pf.n, bs.wo);
        bs.weight = active ? w : V3(0.f);
        return bs;
    }
    return leaf_sample(sc, B, si, si.wi, s1, s2x, s2y);
}
DEV V3 bsdf_eval(SceneRef sc, int b, const SI &si, V3 wo) {
    const DBsdf B = tab(sc.bsdfs, b, sc.one_shape);
    if (B.type == LRT_BSDF_BUMPMAP) {               // src/bsdfs/bumpmap.cpp:164-183
        Frame pf = bump_frame(sc, B, si);
        V3 pwi = pf.to_local(si.wi), pwo = pf.to_local(wo);
        if (!(wo.z * pwo.z > 0.f)) return V3(0.f);
        return leaf_eval(sc, tab(sc.bsdfs, B.nested, sc.one_shape), si, pwi, pwo) * shadow_terminator(pf.n, wo);
    }
    return leaf_eval(sc, B, si, si.wi, wo);
}
DEV float bsdf_pdf(SceneRef sc, int b, const SI &si, V3 wo) {
    const DBsdf B = tab(sc.bsdfs, b, sc.one_shape);
    if (B.type == LRT_BSDF_BUMPMAP) {
        Frame pf = bump_frame(sc, B, si);
        V3 pwi = pf.to_local(si.wi), pwo = pf.to_local(wo);
        if (!(wo.z * pwo.z > 0.f)) return 0.f;
        return leaf_pdf(tab(sc.bsdfs, B.nested, sc.one_shape), pwi, pwo);
    }
    return leaf_pdf(B, si.wi, wo);
}
DEV float bsdf_null_transmission(SceneRef sc, int b) { return tab(sc.bsdfs, b, sc.one_shape).type == LRT_BSDF_NULL ? 1.f : 0.f; }

// -------------------------------------------------------------- emitters
struct DirSample { V3 p, n, d; float pdf, dist; bool delta; int emitter; };

// include/mitsuba/core/distr_2d.h:517-602 + include/mitsuba/core/warp.h:446-494
DEV float interval_to_linear(float v0, float v1, float sample) {
    if (__builtin_fabsf(v0 - v1) > 1e-4f * (v0 + v1))
        return (v0 - safe_sqrt(lerpf(sqr(v0), sqr(v1), sample))) / (v0 - v1);
    return sample;
}
DEV void hier_sample(SceneRef sc, float sx, float sy, float *ox, float *oy, float *pdf) {
    EnvRef E = sc.env;
    sx = clampf(sx, 0.f, 1.f); sy = clampf(sy, 0.f, 1.f);
    uint32_t offx = 0, offy = 0;
    for (int l = E.n_levels - 2; l > 0; --l) {
        offx <<= 1; offy <<= 1;
        uint32_t width = E.level_width[l];
        uint32_t oi = ((offx & 1u) | (((offx & ~1u) | (offy & 1u)) << 1)) + ((offy & ~1u) * width);
        const float4 q = *reinterpret_cast<const float4 *>(sc.env_hier + E.level_offset[l] + oi);   // 2x2 patches are contiguous
        float v00 = q.x, v10 = q.y, v01 = q.z, v11 = q.w;
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
    uint32_t w0 = E.level_width[0];
    const float *l0 = sc.env_hier + E.level_offset[0];
    uint32_t oi = offx + offy * w0;
    float v00 = l0[oi], v10 = l0[oi + 1], v01 = l0[oi + w0], v11 = l0[oi + w0 + 1];
    float r0 = v00 + v10, r1 = v01 + v11;
    sy = interval_to_linear(r0, r1, sy);
    float c0 = lerpf(v00, v01, sy), c1 = lerpf(v10, v11, sy);
    sx = interval_to_linear(c0, c1, sx);
    *pdf = lerpf(c0, c1, sx);
    *ox = ((float) (int) offx + sx) * E.patch_size[0];
    *oy = ((float) (int) offy + sy) * E.patch_size[1];
}
// include/mitsuba/core/distr_2d.h:695-726
DEV float hier_eval(SceneRef sc, float px, float py) {
    EnvRef E = sc.env;
    px = clampf(px, 0.f, 1.f); py = clampf(py, 0.f, 1.f);
    px *= E.inv_patch_size[0]; py *= E.inv_patch_size[1];
    uint32_t ox = min((uint32_t) (int) px, E.max_patch[0]), oy = min((uint32_t) (int) py, E.max_patch[1]);
    px -= (float) (int) ox; py -= (float) (int) oy;
    uint32_t w0 = E.level_width[0];
    const float *l0 = sc.env_hier + E.level_offset[0];
    uint32_t oi = ox + oy * w0;
    float v00 = l0[oi], v10 = l0[oi + 1], v01 = l0[oi + w0], v11 = l0[oi + w0 + 1];
    return lerpf(lerpf(v00, v10, px), lerpf(v01, v11, px), py);
}

// src/emitters/envmap.cpp:528-560
DEV V3 env_eval_uv(SceneRef sc, float u, float v) {
    EnvRef E = sc.env;
    uint32_t rx = E.w, ry = E.h;
    u -= .5f / (float) (rx - 1u);
    u -= __builtin_floorf(u); v -= __builtin_floorf(v);
    u *= (float) (rx - 1u); v *= (float) (ry - 1u);
    uint32_t px = min((uint32_t) u, rx - 2u), py = min((uint32_t) v, ry - 2u);
    float w1x = u - (float) px, w1y = v - (float) py, w0x = 1.f - w1x, w0y = 1.f - w1y;
    uint32_t idx = py * rx + px;
    float4 a = sc.env_data[idx], b = sc.env_data[idx + 1], c = sc.env_data[idx + rx], d = sc.env_data[idx + rx + 1];
    V3 v00(a.x, a.y, a.z), v10(b.x, b.y, b.z), v01(c.x, c.y, c.z), v11(d.x, d.y, d.z);
    V3 t0 = v10 * w1x, t1 = v11 * w1x;
    V3 v0(fma_(w0x, v00.x, t0.x), fma_(w0x, v00.y, t0.y), fma_(w0x, v00.z, t0.z));
    V3 v1(fma_(w0x, v01.x, t1.x), fma_(w0x, v01.y, t1.y), fma_(w0x, v01.z, t1.z));
    V3 t2 = v1 * w1y;
    V3 vv(fma_(w0y, v0.x, t2.x), fma_(w0y, v0.y, t2.y), fma_(w0y, v0.z, t2.z));
    return vv * E.scale;
}
DEV V3 emitter_eval_env(SceneRef sc, V3 dir_world) {
    EnvRef E = sc.env;
    if (E.type == LRT_EMITTER_CONSTANT) return V3(E.radiance[0], E.radiance[1], E.radiance[2]);
    V3 v = xform_vec9(E.to_local, dir_world);           // src/emitters/envmap.cpp:353-362
    float uu = m_atan2(v.x, -v.z) * kInvTwoPi, vv = safe_acos(v.y) * kInvPi;
    return env_eval_uv(sc, uu, vv);
}

// Scene::sample_emitter_direction without visibility test (src/render/scene.cpp:333-383)
DEV V3 sample_emitter_direction(SceneRef sc, V3 ref_p, float sx, float sy, DirSample *ds) {
    uint32_t ne = sc.n_emitters;
    ds->p = V3(0.f); ds->n = V3(0.f); ds->d = V3(0.f); ds->pdf = 0.f; ds->dist = 0.f; ds->delta = false; ds->emitter = -1;
    if (ne == 0) return V3(0.f);
    uint32_t index = 0; float emitter_weight = 1.f, pmf = 1.f;
    if (ne > 1) {
        float scaled = sx * (float) ne;
        index = min((uint32_t) scaled, ne - 1u);
        emitter_weight = (float) ne; sx = scaled - (float) index; pmf = 1.f / (float) ne;
    }
    const DEmitter &E = sc.emitters[index];
    ds->emitter = (int) index;
    V3 spec(0.f);
    if (E.type == LRT_EMITTER_AREA) {
        // src/shapes/rectangle.cpp:181-199, src/render/shape.cpp:343-361, src/emitters/area.cpp sample_direction
        ds->p = xform_point12(E.to_world, V3(fma_(sx, 2.f, -1.f), fma_(sy, 2.f, -1.f), 0.f));
        ds->n = V3(E.n[0], E.n[1], E.n[2]);
        ds->pdf = E.inv_area;
        ds->d = ds->p - ref_p;
        float dist2 = squared_norm(ds->d);
        ds->dist = __builtin_sqrtf(dist2);
        ds->d = ds->d / ds->dist;
        float dp = __builtin_fabsf(dot(ds->d, ds->n)), x = dist2 / dp;
        ds->pdf *= finite_(x) ? x : 0.f;
        bool active = dot(ds->d, ds->n) < 0.f && ds->pdf != 0.f;
        V3 rad(E.radiance[0], E.radiance[1], E.radiance[2]);
        spec = active ? rad / ds->pdf : V3(0.f);
    } else if (E.type == LRT_EMITTER_ENVMAP) {          // src/emitters/envmap.cpp:415-459
        EnvRef EV = sc.env;
        float u, v, pdf; hier_sample(sc, sx, sy, &u, &v, &pdf);
        u += .5f / (float) (EV.w - 1u);
        bool active = pdf > 0.f;
        float theta = v * kPi, phi = u * kTwoPi;
        float st, ct, sp, cp; m_sincos(theta, &st, &ct); m_sincos(phi, &sp, &cp);
        V3 d(cp * st, sp * st, ct);
        d = V3(d.y, d.z, -d.x);
        V3 c(EV.bsphere_c[0], EV.bsphere_c[1], EV.bsphere_c[2]);
        float radius = fmax_(EV.bsphere_r, norm(ref_p - c)), dist = 2.f * radius;
        float inv_sin_theta = safe_rsqrt(fmax_(sqr(d.x) + sqr(d.z), sqr(kEpsilon)));
        d = xform_vec9(EV.to_world, d);
        ds->p = ref_p + d * dist; ds->n = -d;
        ds->pdf = active ? pdf * inv_sin_theta * (1.f / (2.f * sqr(kPi))) : 0.f;
        ds->d = d; ds->dist = dist;
        V3 val = env_eval_uv(sc, u, v);
        spec = active ? val / ds->pdf : V3(0.f);
    } else {                                            // src/emitters/constant.cpp sample_direction
        EnvRef EV = sc.env;
        V3 d = square_to_uniform_sphere(sx, sy);
        V3 c(EV.bsphere_c[0], EV.bsphere_c[1], EV.bsphere_c[2]);
        float radius = fmax_(EV.bsphere_r, norm(ref_p - c)), dist = 2.f * radius;
        ds->p = fma3(d, dist, ref_p); ds->n = -d; ds->pdf = kInvFourPi; ds->d = d; ds->dist = dist;
        spec = V3(E.radiance[0], E.radiance[1], E.radiance[2]) / ds->pdf;
    }
    ds->pdf *= pmf;
    spec = spec * emitter_weight;
    return spec;
}

// DirectionSample(scene, si, ref) + Scene::pdf_emitter_direction
// (include/mitsuba/render/records.h:173-180, src/render/scene.cpp:395-406)
DEV float pdf_emitter_direction(SceneRef sc, V3 ref_p, const SI &si, int emitter) {
    V3 rel = si.p - ref_p;
    float dist = norm(rel);
    V3 d = si.valid ? rel / dist : -si.wi;
    float pmf = 1.f / (float) sc.n_emitters;
    const DEmitter &E = sc.emitters[emitter];
    float value;
    if (E.type == LRT_EMITTER_AREA) {
        float dp = dot(d, si.n);
        if (!(dp < 0.f)) return 0.f;
        float adp = __builtin_fabsf(dp);
        value = E.inv_area * (adp != 0.f ? (dist * dist) / adp : 0.f);
    } else if (E.type == LRT_EMITTER_ENVMAP) {          // src/emitters/envmap.cpp:461-475
        EnvRef EV = sc.env;
        V3 dl = xform_vec9(EV.to_local, d);
        float u = m_atan2(dl.x, -dl.z) * kInvTwoPi, v = safe_acos(dl.y) * kInvPi;
        u -= .5f / (float) (EV.w - 1u);
        u -= __builtin_floorf(u); v -= __builtin_floorf(v);
        float inv_sin_theta = safe_rsqrt(fmax_(sqr(dl.x) + sqr(dl.z), sqr(kEpsilon)));
        value = hier_eval(sc, u, v) * inv_sin_theta * (1.f / (2.f * sqr(kPi)));
    } else value = kInvFourPi;
    return value * pmf;
}

DEV int si_emitter(SceneRef sc, const SI &si) { return si.valid ? tab(sc.shapes, si.shape, sc.one_shape).emitter : sc.env.emitter; }
DEV V3 emitter_eval(SceneRef sc, int e, const SI &si) {
    if (!si.valid) return emitter_eval_env(sc, -si.wi);
    const DEmitter &E = sc.emitters[e];                 // src/emitters/area.cpp eval()
    return (si.wi.z > 0.f) ? V3(E.radiance[0], E.radiance[1], E.radiance[2]) : V3(0.f);
}

// exp(-t * c) per channel; a medium whose channels are equal (the usual case: sigma_t = 1 * scale) pays one exponential.  The
// values are the ones the three separate calls give (the same operation on the same operands).
DEV V3 exp_neg(float t, V3 c) {
    const float ex = m_exp(-t * c.x);
    if (c.y == c.x && c.z == c.x) return V3(ex);
    return V3(ex, m_exp(-t * c.y), m_exp(-t * c.z));
}
DEV V3 div_uniform(V3 a, float b) {              // a / b with one correctly rounded division when the channels of a are equal
    const float q = a.x / b;
    if (a.y == a.x && a.z == a.x) return V3(q);
    return V3(q, a.y / b, a.z / b);
}

DEV float mis_weight(float a, float b) { a *= a; b *= b; float w = a / (a + b); return finite_(w) ? w : 0.f; }

// ----------------------------------------------------------------- media
struct MI { float t; V3 p, wi; V3 sigma_s, sigma_n, sigma_t, combined; float mint;
            DEV bool valid() const { return t != kInf; } };

// src/render/medium.cpp:40-82 + src/media/homogeneous.cpp:153-181
// The free-flight distance alone (the look-ahead of volpath_iteration draws it one trip early and keeps it in the record)
DEV float medium_sampled_t(const DMedium &M, float sample, uint32_t channel) {
    const float mm = channel == 0 ? M.sigma_t[0] : (channel == 1 ? M.sigma_t[1] : M.sigma_t[2]);
    return 0.f + (-m_log(1.f - sample) / mm);
}
DEV MI medium_interaction_at(const DMedium &M, const Ray &ray, float sampled_t) {
    MI mei; mei.wi = -ray.d;
    float mint = 0.f, maxt = fmin_(ray.maxt, kInf);
    V3 sigmat(M.sigma_t[0], M.sigma_t[1], M.sigma_t[2]);
    bool valid = sampled_t <= maxt;
    mei.t = valid ? sampled_t : kInf;
    mei.p = fma3(ray.d, sampled_t, ray.o);
    mei.mint = mint;
    V3 albedo(M.albedo[0], M.albedo[1], M.albedo[2]);
    mei.sigma_t = valid ? sigmat : V3(0.f);
    mei.sigma_s = valid ? sigmat * albedo : V3(0.f);
    mei.sigma_n = V3(0.f);
    mei.combined = sigmat;
    return mei;
}
DEV MI medium_sample_interaction(const DMedium &M, const Ray &ray, float sample, uint32_t channel) {
    return medium_interaction_at(M, ray, medium_sampled_t(M, sample, channel));
}

// src/volumes/grid.cpp (one channel) through Dr.Jit's Texture3f::eval (trilinear, clamp): texel centres at (i + .5) / res,
// lerp along x, then y, then z, each as fmadd(w0, a, w1 * b)
DEV float grid_eval(const DHetMedium &H, V3 pl) {
    const int rx = H.res[0], ry = H.res[1], rz = H.res[2];
    float fx = fma_(pl.x, (float) rx, -.5f), fy = fma_(pl.y, (float) ry, -.5f), fz = fma_(pl.z, (float) rz, -.5f);
    float flx = __builtin_floorf(fx), fly = __builtin_floorf(fy), flz = __builtin_floorf(fz);
    float w1x = fx - flx, w1y = fy - fly, w1z = fz - flz, w0x = 1.f - w1x, w0y = 1.f - w1y, w0z = 1.f - w1z;
    auto cl = [](float v, int n) { int i = (v < -2e9f) ? -2000000000 : (v > 2e9f ? 2000000000 : (int) v); return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); };
    int x0 = cl(flx, rx), x1 = cl(flx + 1.f, rx), y0 = cl(fly, ry), y1 = cl(fly + 1.f, ry), z0 = cl(flz, rz), z1 = cl(flz + 1.f, rz);
    const float *d = H.data;
    auto at = [&](int x, int y, int z) { return d[((size_t) z * ry + y) * rx + x]; };
    float c00 = fma_(w0x, at(x0, y0, z0), w1x * at(x1, y0, z0)), c10 = fma_(w0x, at(x0, y1, z0), w1x * at(x1, y1, z0));
    float c01 = fma_(w0x, at(x0, y0, z1), w1x * at(x1, y0, z1)), c11 = fma_(w0x, at(x0, y1, z1), w1x * at(x1, y1, z1));
    float c0 = fma_(w0y, c00, w1y * c10), c1 = fma_(w0y, c01, w1y * c11);
    return fma_(w0z, c0, w1z * c1);
}

// include/mitsuba/core/bbox.h:303-340 (Williams et al.)
DEV bool bbox_ray_intersect(const float *lo, const float *hi, const Ray &ray, float &mint, float &maxt) {
    bool active = ray.d.x != 0.f || ray.d.y != 0.f || ray.d.z != 0.f;
    const float dx = rcp(ray.d.x), dy = rcp(ray.d.y), dz = rcp(ray.d.z);
    float t0x = ((dx >= 0.f ? lo[0] : hi[0]) - ray.o.x) * dx, t1x = ((dx >= 0.f ? hi[0] : lo[0]) - ray.o.x) * dx;
    float t0y = ((dy >= 0.f ? lo[1] : hi[1]) - ray.o.y) * dy, t1y = ((dy >= 0.f ? hi[1] : lo[1]) - ray.o.y) * dy;
    float t0z = ((dz >= 0.f ? lo[2] : hi[2]) - ray.o.z) * dz, t1z = ((dz >= 0.f ? hi[2] : lo[2]) - ray.o.z) * dz;
    auto max_safe = [](float a, float b) { return (a > b || !finite_(b)) ? a : b; };
    auto min_safe = [](float a, float b) { return (a < b || !finite_(b)) ? a : b; };
    active = active && !((t0x > t1y) || (t0y > t1x));
    t0x = max_safe(t0x, t0y); t1x = min_safe(t1x, t1y);
    active = active && !((t0x > t1z) || (t0z > t1x));
    t0x = max_safe(t0x, t0z); t1x = min_safe(t1x, t1z);
    mint = t0x; maxt = t1x;
    return active;
}

// src/render/medium.cpp:40-82 + src/media/heterogeneous.cpp:178-200: delta-tracking majorant, sigma_n = majorant - sigma_t(p)
DEV MI het_sample_interaction(const DMedium &M, const DHetMedium &H, const Ray &ray, float sample) {
    MI mei; mei.wi = -ray.d;
    float mint, maxt;
    bool active = bbox_ray_intersect(H.bbox_min, H.bbox_max, ray, mint, maxt);
    active = active && (finite_(mint) || finite_(maxt));
    if (!active) { mint = 0.f; maxt = kInf; }
    mint = fmax_(0.f, mint); maxt = fmin_(ray.maxt, maxt);
    const float max_density = H.max_density;
    float sampled_t = mint + (-m_log(1.f - sample) / max_density);
    bool valid = active && sampled_t <= maxt;
    mei.t = valid ? sampled_t : kInf;
    mei.p = fma3(ray.d, sampled_t, ray.o);
    mei.mint = mint;
    float st = 0.f;
    if (valid) st = H.scale * grid_eval(H, xform_point12(H.to_local, mei.p));
    V3 albedo(M.albedo[0], M.albedo[1], M.albedo[2]);
    mei.sigma_t = V3(st); mei.sigma_s = mei.sigma_t * (valid ? albedo : V3(0.f));
    mei.sigma_n = V3(max_density) - mei.sigma_t;
    mei.combined = V3(max_density);
    return mei;
}

// src/phase/hg.cpp:64-99, src/phase/isotropic.cpp:39-58
DEV float hg_eval(float g, float cos_theta) {
    float temp = 1.f + sqr(g) + 2.f * g * cos_theta;
    return kInvFourPi * (1.f - sqr(g)) / (temp * __builtin_sqrtf(temp));
}
DEV void phase_sample(const DMedium &M, V3 wi, float s2x, float s2y, V3 *wo, float *pdf) {
    if (M.phase == LRT_PHASE_HG) {
        float g = M.g;
        float sqr_term = (1.f - sqr(g)) / (1.f - g + 2.f * g * s2x);
        float cos_theta = (1.f + sqr(g) - sqr(sqr_term)) / (2.f * g);
        if (__builtin_fabsf(g) < kEpsilon) cos_theta = 1.f - 2.f * s2x;
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
DEV float phase_eval(const DMedium &M, V3 wi, V3 wo) { return M.phase == LRT_PHASE_HG ? hg_eval(M.g, dot(wo, wi)) : kInvFourPi; }

DEV int target_medium(const DShape &sd, V3 d, V3 n) { return dot(d, n) > 0.f ? sd.exterior_medium : sd.interior_medium; }
DEV bool is_medium_transition(const DShape &sd) { return sd.interior_medium >= 0 || sd.exterior_medium >= 0; }

} // namespace lrt
