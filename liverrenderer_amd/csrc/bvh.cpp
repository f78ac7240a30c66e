#include "bvh.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace lrt {

namespace {
const int kMaxDepth = 22;          // traversal stacks hold 32 (global path) / 24 (LDS path) entries
const int kBins = 16;

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; } }
    void grow(const Box &b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    void grow(const float *p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    float area() const { float d[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] }; return d[0] < 0 ? 0.f : 2.f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]); }
};
struct Prim { Box b; float c[3]; uint32_t id; };

struct Builder {
    const float *pos; const uint32_t *faces; std::vector<Prim> prims; HostBVH &out; int kLeafSize = 4;
    Builder(HostBVH &o) : out(o) {}

    Box padded(const Box &b) const {
        Box r = b;
        for (int a = 0; a < 3; ++a) {
            float pad = 1e-5f * (b.hi[a] - b.lo[a]) + 4e-6f * std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])) + 1e-30f;
            r.lo[a] = b.lo[a] - pad; r.hi[a] = b.hi[a] + pad;
        }
        return r;
    }
    int emit_leaf(uint32_t b, uint32_t e, uint32_t *count) {
        uint32_t first = (uint32_t) (out.tris.size() / 12);
        for (uint32_t i = b; i < e; ++i) {
            uint32_t f = prims[i].id;
            const float *p0 = pos + 3 * faces[3 * f], *p1 = pos + 3 * faces[3 * f + 1], *p2 = pos + 3 * faces[3 * f + 2];
            float t[12] = { p0[0], p0[1], p0[2], 0.f, p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2], 0.f, p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2], 0.f };
            memcpy(&t[3], &f, 4);
            if (i + 1 == e) t[7] = 1.f;                 // e1.w: last triangle of the leaf (while-while traversal)
            out.tris.insert(out.tris.end(), t, t + 12);
        }
        *count = e - b;
        return ~(int) first;
    }
    // Returns the child reference for [b, e) and its bounds.
    int build(uint32_t b, uint32_t e, int depth, Box *bounds, uint32_t *count) {
        Box bb, cb; bb.reset(); cb.reset();
        for (uint32_t i = b; i < e; ++i) { bb.grow(prims[i].b); cb.grow(prims[i].c); }
        *bounds = bb;
        out.max_depth = std::max(out.max_depth, depth);
        uint32_t n = e - b;
        if (n <= (uint32_t) kLeafSize || depth >= kMaxDepth) return emit_leaf(b, e, count);
        // binned SAH over the three axes
        int best_axis = -1, best_bin = -1; float best_cost = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3; ++a) {
            float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.f)) continue;
            Box binb[kBins]; uint32_t binc[kBins];
            for (int k = 0; k < kBins; ++k) { binb[k].reset(); binc[k] = 0; }
            float scale = (float) kBins / ext;
            for (uint32_t i = b; i < e; ++i) { int k = std::min(kBins - 1, (int) ((prims[i].c[a] - cb.lo[a]) * scale)); binb[k].grow(prims[i].b); binc[k]++; }
            float la[kBins], ra[kBins]; uint32_t lc[kBins], rc[kBins];
            Box acc; acc.reset(); uint32_t cnt = 0;
            for (int k = 0; k < kBins; ++k) { acc.grow(binb[k]); cnt += binc[k]; la[k] = acc.area(); lc[k] = cnt; }
            acc.reset(); cnt = 0;
            for (int k = kBins - 1; k >= 0; --k) { acc.grow(binb[k]); cnt += binc[k]; ra[k] = acc.area(); rc[k] = cnt; }
            for (int k = 0; k < kBins - 1; ++k) {
                if (!lc[k] || !rc[k + 1]) continue;
                float cost = la[k] * (float) lc[k] + ra[k + 1] * (float) rc[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
            }
        }
        uint32_t mid;
        if (best_axis < 0) {
            if (n <= (uint32_t) std::max(8, kLeafSize)) return emit_leaf(b, e, count);          // coincident centroids
            mid = (b + e) / 2;
        } else {
            float ext = cb.hi[best_axis] - cb.lo[best_axis], scale = (float) kBins / ext, lo = cb.lo[best_axis]; int a = best_axis, kb = best_bin;
            auto it = std::partition(prims.begin() + b, prims.begin() + e, [&](const Prim &p) { return std::min(kBins - 1, (int) ((p.c[a] - lo) * scale)) <= kb; });
            mid = (uint32_t) (it - prims.begin());
            if (mid == b || mid == e) mid = (b + e) / 2;
        }
        uint32_t node = (uint32_t) (out.nodes.size() / 16);
        out.nodes.resize(out.nodes.size() + 16, 0.f);
        Box b0, b1; uint32_t c0 = 0, c1 = 0;
        int r0 = build(b, mid, depth + 1, &b0, &c0), r1 = build(mid, e, depth + 1, &b1, &c1);
        Box p0 = padded(b0), p1 = padded(b1);
        float *nd = &out.nodes[16 * (size_t) node];
        nd[0] = p0.lo[0]; nd[1] = p0.hi[0]; nd[2] = p0.lo[1]; nd[3] = p0.hi[1];
        nd[4] = p1.lo[0]; nd[5] = p1.hi[0]; nd[6] = p1.lo[1]; nd[7] = p1.hi[1];
        nd[8] = p0.lo[2]; nd[9] = p0.hi[2]; nd[10] = p1.lo[2]; nd[11] = p1.hi[2];
        int32_t refs[4] = { r0, r1, (int32_t) c0, (int32_t) c1 };
        memcpy(&nd[12], refs, 16);
        *count = 0;
        return (int) node;
    }
};
} // namespace

void build_bvh(const float *positions, const uint32_t *faces, uint32_t n_faces, HostBVH &out, int leaf_size) {
    out = HostBVH();
    Builder B(out); B.kLeafSize = std::max(1, leaf_size); B.pos = positions; B.faces = faces; B.prims.resize(n_faces);
    for (uint32_t f = 0; f < n_faces; ++f) {
        Prim &p = B.prims[f]; p.id = f; p.b.reset();
        for (int k = 0; k < 3; ++k) p.b.grow(positions + 3 * faces[3 * f + k]);
        for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.b.lo[a] + p.b.hi[a]);
    }
    if (n_faces == 0) { out.root_is_leaf = true; out.root_first = 0; out.root_count = 0; out.nodes.assign(16, 0.f); out.tris.assign(12, 0.f); return; }
    Box bb; uint32_t cnt = 0;
    int r = B.build(0, n_faces, 0, &bb, &cnt);
    if (r < 0) { out.root_is_leaf = true; out.root_first = (uint32_t) ~r; out.root_count = cnt; out.nodes.assign(16, 0.f); }
}

} // namespace lrt
