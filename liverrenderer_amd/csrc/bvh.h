// Host BVH builder for the device traversal kernels.  Replaces the reference's
// SAH kd-tree (include/mitsuba/render/kdtree.h) / OptiX GAS: binned-SAH BVH2 in a
// 64-byte "two child boxes per node" layout, leaves referencing contiguous
// triangle slots that hold pre-gathered (p0, e1, e2, prim id).
//
// Node n = 4 x float4:
//   [0] = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)
//   [1] = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//   [2] = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)
//   [3] = (ref0, ref1, count0, count1) as int bits; ref >= 0: inner node index,
//         ref < 0: leaf whose first triangle slot is ~ref, count = number of slots.
// Child boxes are padded conservatively so that the slab test can never cull a
// triangle the Moeller-Trumbore test would accept.
#pragma once
#include <vector>
#include <cstdint>

namespace lrt {

struct HostBVH {
    std::vector<float> nodes;          // 16 floats per node
    std::vector<float> tris;           // 12 floats per slot
    bool root_is_leaf = false;         // scenes with <= leaf_size triangles
    uint32_t root_first = 0, root_count = 0;
    int max_depth = 0;
};

// leaf_size: the builder splits until a node holds at most this many triangles (4 by default; the device side asks for fatter
// leaves when that makes the LDS image of a larger mesh fit: device.hip)
void build_bvh(const float *positions, const uint32_t *faces, uint32_t n_faces, HostBVH &out, int leaf_size = 4);

} // namespace lrt
