/*
 * orc_scene.h -- derived scene data of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 * Everything here is rebuilt by the oracle itself from the POD scene
 * description; nothing is shared with the product's device structures.
 */
#pragma once
#include "orc.h"
#include "orc_math.h"
#include <vector>
#include <string>
#include <atomic>

namespace orc {

struct Ray { V3 o, d; float maxt; };
struct Hit { float t, u, v; uint32_t prim; bool valid() const { return prim != 0xffffffffu; } };

struct BVHNode {           /* plain BVH2, children adjacent, leaves hold a prim range */
    float lo[3], hi[3];
    uint32_t left;         /* inner: index of left child (right = left+1); leaf: first prim slot */
    uint32_t count;        /* 0: inner */
};

/* include/mitsuba/core/distr_2d.h:371-760 Hierarchical2D<Float, 0> */
struct Hier2D {
    struct Level { uint32_t size, width; std::vector<float> data;
        uint32_t index(uint32_t x, uint32_t y) const {
            return ((x & 1u) | (((x & ~1u) | (y & 1u)) << 1)) + ((y & ~1u) * width);
        } };
    std::vector<Level> levels;
    V2 patch_size, inv_patch_size;
    uint32_t max_patch_x, max_patch_y;
    void build(const float *data, uint32_t w, uint32_t h);
    void sample(float sx, float sy, float *ox, float *oy, float *pdf) const;
    float eval(float px, float py) const;
};

struct SI {                /* SurfaceInteraction3f subset (include/mitsuba/render/interaction.h:205-242) */
    bool valid; float t; V3 p, n; Frame sh; V2 uv; V3 dp_du, dp_dv, wi; uint32_t prim, shape;
};

struct Scene {
    lrt_scene_desc d;                         /* deep copy (arrays owned below) */
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> faces, face_shape;
    std::vector<lrt_shape_desc> shapes;
    std::vector<lrt_bsdf_desc> bsdfs;
    std::vector<lrt_texture_desc> textures;
    std::vector<std::vector<float>> texdata;
    std::vector<lrt_medium_desc> media;
    std::vector<std::vector<float>> meddata;
    std::vector<lrt_emitter_desc> emitters;
    std::vector<std::vector<float>> emdata;

    /* camera (src/sensors/perspective.cpp:174-198) */
    M4 sample_to_camera, cam_to_world;
    /* acceleration */
    std::vector<BVHNode> nodes; std::vector<uint32_t> prim_ids;
    /* environment */
    int env = -1;                             /* emitter index of the environment, -1: none */
    V3 bsphere_c; float bsphere_r;
    M4 env_to_world, env_to_local;
    std::vector<float> env_data; uint32_t env_w = 0, env_h = 0;   /* (w+1) x h x 3 */
    Hier2D env_warp;
    /* area lights: per emitter frame of the owning rectangle */
    struct AreaInfo { V3 n; float inv_area; };
    std::vector<AreaInfo> area;
    /* reconstruction filter */
    float rf_radius = 0.5f; float rf_coeff[10]; float rf_inv_radius = 1.f;
    bool has_null_bsdf = false;
    /* prbvolpath.py:84-91 prepare_scene(): over the media attached to shapes */
    bool prb_handle_null_scattering = false, prb_nee_handle_homogeneous = false;
    bool bio_scalar = false;                  /* bio transport reading: false = JIT variants (default), true = scalar_rgb */

    void finalize();
    Hit intersect(const Ray &r, bool any_hit, bool brute) const;
    SI  compute_si(const Ray &r, const Hit &h) const;
    float rfilter_eval(float x) const;
};

} // namespace orc

struct orc_scene { orc::Scene s; };
