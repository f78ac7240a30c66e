// POD structures shared by the host driver (render.cpp side of device.hip) and
// the gfx950 kernels.  Uniform scene data is passed to kernels BY VALUE
// (kernarg segment -> scalar loads), bulk data by pointer.
#pragma once
#include <stdint.h>
#include <hip/hip_vector_types.h>

namespace lrt {

struct DShape {
    int32_t bsdf, emitter, interior_medium, exterior_medium;
    int32_t has_normals, has_texcoords, flip_normals, kind;
};

struct DBsdf {
    int32_t type, reflectance, nested, texture;
    float eta, scale; int32_t flags, pad;
};

struct DTexture {
    int32_t type, width, height, channels;
    float color0[3]; float color1[3];
    float to_uv[6];            // first two rows of the 3x3 uv transform
    uint32_t data_offset;      // into tex_data (floats); bitmap: 1 float per texel (luminance or Y)
    uint32_t pad;
};

struct DMedium {
    float sigma_t[3]; float albedo[3];   // sigma_t already multiplied by scale
    int32_t has_spectral_extinction, sample_emitters, phase; float g;
    float scale; int32_t het;            // sigma_t = property * scale (needed by the PRB adjoint); het: heterogeneous (sc.het[index])
    float w_spec[3], w_plain[3];         // real-scattering weights sigma_s / mean(sigma_t / combined), sigma_s / sigma_t (k_medium_prepare)
    float nee_vmin[3];                   // per channel: 1 - u >= nee_vmin proves that the free-flight distance -log(1 - u) / sigma_t stays below the distance
    float pad2[3];                       // of every sample of an infinite emitter (upload_media; volpath_iteration's early NEE rejection)
};

// bio media (liver / parenchyma / glissonCapsule): element coefficients of the 5-argument sample_interaction
// (kernels_bio.h); kept apart from DMedium so that the path / volpath kernels' medium loads stay as they are
struct DBioMedium {
    int32_t type, has_spectral_extinction;    // LRT_MEDIUM_*
    float layer_limit[4];
    float collagen[4][3], elastin[4][3];      // per layer, channel order as the plugins store them
    float blood[3], bile[3], lipid_water[3];
    float hepatocity, log10_hep;              // log10_hep = log2(hepatocity + 1) / log2(10), filled by k_bio_prepare (device arithmetic)
    float sigmat[3];                          // what get_majorant reports: sigma_t * scale, or parenchyma's constants
    float pad;
};

// heterogeneous media (src/media/heterogeneous.cpp): sigma_t = scale * grid, majorant = scale * max(grid)
struct DHetMedium {
    const float *data;                        // x fastest
    int32_t res[3]; float max_density;        // scale * grid_max (get_majorant)
    float to_local[12];                       // world -> unit cube
    float bbox_min[3], bbox_max[3];
    float scale, pad;
};

struct DEmitter {
    int32_t type, shape; float radiance[3]; float scale;
    float n[3]; float inv_area;          // area: rectangle frame normal (flip applied), 1/area
    float to_world[12];                  // area: rectangle to_world rows 0..2
};

#define LRT_MAX_HIER_LEVELS 16

struct DCamera {
    float s2c[16];             // sample_to_camera, row-major, projective
    float to_world[12];        // rows 0..2
    float near_clip, far_clip;
    int32_t medium, pad;
    float ppo_x, ppo_y;        // film size * principal_point_offset / crop size (perspective.cpp:214-215)
};

struct DFilm {
    int32_t width, height;               // crop size
    int32_t crop_offset_x, crop_offset_y;
    float scale_x, scale_y, offset_x, offset_y;   // render_sample(): adjusted = fmadd(pos, scale, offset)
    int32_t channels, has_alpha, rfilter, fcount; // fcount = 2*ceil(radius-.5)+1
    float rf_radius, rf_inv_radius; int32_t fn, pad;
    float rf_coeff[10];
    int32_t pad2[2];
};

struct DEnv {
    int32_t type;              // -1: none, LRT_EMITTER_ENVMAP, LRT_EMITTER_CONSTANT
    int32_t emitter;           // emitter index
    uint32_t w, h;             // (w+1) x h storage
    float scale;
    float bsphere_c[3]; float bsphere_r;
    float to_world[9];         // rotation part rows
    float to_local[9];
    float patch_size[2], inv_patch_size[2];
    uint32_t max_patch[2];
    int32_t n_levels;
    uint32_t level_offset[LRT_MAX_HIER_LEVELS];   // into env_hier (floats)
    uint32_t level_width[LRT_MAX_HIER_LEVELS];
    float radiance[3];         // constant emitter
};

// Conservative distance field over the triangle soup (see dist_grid_lower_bound() in dshade.h): value = distance from
// the cell centre to the nearest triangle, rounded down.  Lets short in-medium segments prove "no surface within reach"
// without touching the BVH.  The answer of a ray query never depends on it.
struct DDistGrid {
    const uint16_t *d;         // binary16, rounded DOWN (half the bytes of f32: the 192^3 field is 14 MB and caches better)
    float lo[3]; float cell, inv_cell;
    int32_t n[3]; int32_t enabled;
};

struct DScene {
    // acceleration structure
    const float4 *nodes;       // 4 x float4 per BVH2 node (see bvh.h)
    const float4 *tris;        // 3 x float4 per triangle slot: p0 | e1 | e2, prim id in .w of the first
    // geometry attributes
    const float *positions;
    const float4 *vattr;       // per vertex: (n.x, n.y, n.z, 0), (u, v, 0, 0): one 16-byte and one 8-byte load instead of five dword loads
    const uint32_t *faces, *face_shape;
    const DShape *shapes; const DBsdf *bsdfs; const DTexture *textures; const DMedium *media; const DBioMedium *bio; const DHetMedium *het; const DEmitter *emitters;
    const float *tex_data;
    const float4 *env_data;    // (w+1) x h RGBx
    const float *env_hier;
    uint32_t n_faces, n_emitters;
    int32_t root_is_leaf, has_null_bsdf;
    int32_t nee_fast_reject;         // see volpath_iteration(): in-medium NEE can be rejected before sampling the emitter
    int32_t one_shape;               // the scene has a single shape: shape / BSDF table reads are wave-uniform (tab() in dshade.h)
    uint32_t root_leaf_first, root_leaf_count;
    DCamera cam; DFilm film; DEnv env;
    DDistGrid grid;
};

// The scene record lives in device memory and is read through the CONSTANT address space: every `sc.field` is a scalar
// load (s_load_dword*) issued where the value is used, instead of ~190 kernarg SGPRs that stay live through the whole
// persistent kernel and are spilled to VGPR lanes (v_writelane / v_readlane: VALU work) - round-1 k_render: 347 SGPR spills.
#define LRT_CONST __attribute__((address_space(4)))
typedef const LRT_CONST DScene &SceneRef;
typedef const LRT_CONST DFilm &FilmRef;
typedef const LRT_CONST DEnv &EnvRef;
typedef const LRT_CONST DDistGrid &GridRef;
typedef const LRT_CONST DScene *ScenePtr;      // kernel argument (a plain device pointer on the host side)

struct DRenderParams {
    int32_t integrator, max_depth, rr_depth, hide_emitters;
    uint32_t spp, log2_spp;    // log2_spp = 0xffffffff when spp is not a power of two
    uint32_t seed_value;       // sampler base seed + render seed (independent sampler)
    uint32_t base_seed, seed;  // the two terms (the ld sampler keys its per-pixel scramble on them separately)
    uint32_t ld_count, pad1;   // 0: independent sampler; else the ld sampler's sample count (= spp)
    uint32_t tile_rank, tile_count;
    int32_t grad_medium;       // PRB adjoint: medium whose parameters are differentiated, -1: all media into one set
    uint32_t tiles_x, tiles_y, profile;   // profile: per-region tile timing into DCounters (developer aid, LRT_DEBUG_LAUNCH)
    uint32_t compact;          // 80-byte records (scenes without area emitters: the last scatter position is never read): see store_state
    uint64_t n_lanes;          // lanes this launch renders
    const uint32_t *pixel_slot; // tile-sharded renders: pixel -> index in the rank's pixel list (rank-local lane index), else null
    // multi-pass renders (integrator.cpp:176-184,275-293,343-353): spp above is the samples of ONE pass
    uint32_t pass_index, spp_total;
    const unsigned long long *pass_in;   // independent sampler: every lane's PCG32 state at the start of this pass (pass > 0)
    unsigned long long *pass_out;        // ... and where finished paths leave it for the next pass (null: last / only pass)
};

// Path-state streams (SoA, one float4 / uint2 per path and stream)
struct DPathStreams {
    float4 *o_maxt;            // ray origin, maxt
    float4 *d_eta;             // ray direction, eta
    float4 *tp_pdf;            // throughput rgb, last_scatter_direction_pdf
    float4 *res_flags;         // result rgb, packed flags (bits)
    float4 *lp_lane;           // last scatter position, lane id (bits)
    uint2  *rng;               // PCG32 state
    float2 *tdepth;            // biovolpath / biovolpath06 only: the loop state `tissueDepth` (sign bit: the cached competition was won by the hepatocytes), and the
                               // free-flight distance the look-ahead already drew for the next trip (NaN: none)
    float4 *hit;               // volpath with heterogeneous media / volpathmis: the surface interaction a null collision keeps (t, u, v, prim)
    float4 *w1, *w2, *w3, *w4; // volpathmis only: with tp_pdf, the 18 floats of p_over_f and p_over_f_nee
};
#define LRT_STATE_BYTES 88     // bytes per path record across all streams (path / volpath)
#define LRT_STATE_BYTES_MIS 168 // volpathmis: o, d, res, lp, rng, hit + five float4 of MIS weights
#define LRT_STATE_BYTES_HET 104 // volpath with heterogeneous media: + the kept surface hit (float4)
#define LRT_STATE_BYTES_BIO 96 // biovolpath*: + tissueDepth and the look-ahead's free-flight distance; the maxt slot carries the previous ray query's distance

#define LRT_INTEGRATOR_VOLPATHMIS_PLAIN 102   // kernel selector: volpathmis with use_spectral_mis = false
#define LRT_INTEGRATOR_VOLPATH_HET 101   // kernel selector (not an API value): volpath on a scene with heterogeneous media

// flag word layout
#define PF_DEPTH_MASK   0x0000ffffu
#define PF_MEDIUM_SHIFT 16             // (medium index + 1), 8 bits
#define PF_MEDIUM_MASK  0x00ff0000u
#define PF_CHANNEL_SHIFT 24            // 2 bits
#define PF_SPECULAR     (1u << 26)     // volpath specular_chain / path prev_bsdf_delta
#define PF_VALID        (1u << 27)
#define PF_NOHIT        (1u << 28)     // look-ahead proved that the next free-flight segment reaches no surface
#define PF_BIO_SCATTERED (1u << 29)    // biovolpath06: scattered_chain
#define PF_LONG_QUERY   (1u << 29)     // biovolpath: the queued record's ray query is unbounded (the competition's distance lies beyond the previous query's hit): queue region L
#define PF_HAVE_SI      (1u << 29)     // volpath, heterogeneous media: needs_intersection == false, the record's hit stream holds `si`
#define PF_LAST_NULL    (1u << 30)     // volpathmis: last_event_was_null
#define PF_BIO_EMIT     (1u << 30)     // biovolpath06: type & 0x0001 (EmittedRadiance)
#define PF_BIO_FULL     (1u << 31)     // biovolpath06: type & 0x0004 and type & 0x0008 (they only appear together)

struct DCounters {             // device-resident queue / statistics words
    uint32_t pad[4];
    unsigned long long n_shadow;   // NEE ray queries actually needed
    unsigned long long n_iter;     // loop trips (render kernel)
    unsigned long long next_lane;  // global camera-lane ticket (render kernel)
    unsigned long long n_records;  // path records loaded from the queues
    unsigned long long prof_cycles[4], prof_tiles[4];   // per tile kind (A, C, B, fresh): wall_clock64 ticks and tiles (LRT_DEBUG_LAUNCH)
    unsigned long long prof_wg[6];   // LRT_DEBUG_LAUNCH: workgroup timeline, 100 MHz ticks: sum of start, sum of (loop start - start), sum of end, max end, min start + 2^62 trick see kernels.h, sum of barrier waits of thread 0
};

// LDS image of the scene for the persistent traversal kernel: [nodes | verts (float4) | tris (4 x u16)] copied verbatim
// from `blob`, followed by the per-thread traversal stacks.
struct DLdsInfo {
    const uint4 *blob;
    uint32_t blob_bytes, nodes_off, verts_off, tris_off, stack_off, total_bytes;
};

// Every argument of one launch of k_render / k_render_prb / k_splat_lanes, read through the constant address space like the
// scene record (the host fills one slot of a small ring per launch, device.hip).
struct DLaunch {
    DRenderParams rp; DLdsInfo li; DPathStreams q0, q1;
    float4 *dl0, *dl1;                    // PRB: delta_L streams
    uint32_t P, pad; DCounters *cnt; const uint32_t *pixel_list; uint64_t lane_begin, n;
    float4 *L_buf; const float *grad_image; const float *wfilm; double *grads;
    float *film; float *sample_out; uint64_t sample_base;
};
typedef const LRT_CONST DRenderParams &RpRef;
typedef const LRT_CONST DLaunch *LaunchPtr;

} // namespace lrt
