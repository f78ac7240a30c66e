/*
 * liverrt.h -- C ABI of the MI355X-native `hip_ad_rgb` rendering back-end.
 *
 * This is the drop-in boundary for the ONE hot path of mmigas/LiverRenderer
 * (a Mitsuba 3.8 fork): the forward path / volpath sample loop and its PRB
 * adjoint. Everything below is plain C: opaque handles, POD structs, caller
 * owned buffers, no exceptions (errors: status code + lrt_last_error()).
 *
 * Reference interfaces each entry point replaces (paths relative to the
 * reference tree):
 *
 *   lrt_scene_load_xml / _xml_string    src/core/parser.cpp (mi.load_file / mi.load_string,
 *                                        `-Dkey=value` defines: src/mitsuba/mitsuba.cpp:161,240-246)
 *   lrt_scene_from_desc                  mi.load_dict (src/python/python/util.py) -- "from buffers"
 *   lrt_render                           Integrator::render(scene, sensor, seed, spp, develop, evaluate)
 *                                        include/mitsuba/render/integrator.h:74,
 *                                        src/render/integrator.cpp:151-395 (JIT branch :274-388)
 *   lrt_render_backward                  Integrator::render_backward include/mitsuba/render/integrator.h:253,
 *                                        src/python/python/ad/integrators/common.py:625-783
 *   lrt_trace                            the ray-tracing callback seam of the LLVM variants
 *                                        src/render/scene_native.inl:135-202 (SoA RayHit layout)
 *   lrt_param_set / lrt_param_get        mi.traverse(scene)[key] (src/media/homogeneous.cpp:146-151,
 *                                        src/phase/hg.cpp:60-62)
 *   lrt_film_develop                     HDRFilm::develop src/films/hdrfilm.cpp:306-410
 *   lrt_last_error                       Throw(...) -> Python RuntimeError
 *
 * Threading: one host thread per lrt_scene; device work runs on a private HIP
 * stream owned by the scene.  All host pointers are caller-owned unless
 * returned by a *_load / *_from_desc call.
 */
#ifndef LIVERRT_H
#define LIVERRT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRT_API __attribute__((visibility("default")))

typedef enum {
    LRT_OK = 0,
    LRT_ERR_INVALID = 1,      /* bad argument / unsupported plugin / parse error */
    LRT_ERR_IO = 2,           /* file not found / unreadable                      */
    LRT_ERR_DEVICE = 3,       /* HIP runtime failure, or no GPU present           */
    LRT_ERR_UNSUPPORTED = 4
} lrt_status;

/* ---------------------------------------------------------------- enums */
enum { LRT_INTEGRATOR_PATH = 0, LRT_INTEGRATOR_VOLPATH = 1, LRT_INTEGRATOR_PRBVOLPATH = 2,
       LRT_INTEGRATOR_VOLPATHMIS = 5,    /* src/integrators/volpathmis.cpp (spectral MIS)                 */
       LRT_INTEGRATOR_BIOVOLPATH = 3,    /* src/integrators/biovolpath.cpp, JIT-variant lane semantics   */
       LRT_INTEGRATOR_BIOVOLPATH06 = 4   /* src/integrators/biovolpath06.cpp, scalar semantics per lane   */ };
/* medium plugins: src/media/homogeneous.cpp, liver.cpp, parenchyma.cpp, glissonCapsule.cpp (docs/BIO_TRANSPORT_SPEC.md) */
enum { LRT_MEDIUM_HOMOGENEOUS = 0, LRT_MEDIUM_LIVER = 1, LRT_MEDIUM_PARENCHYMA = 2, LRT_MEDIUM_GLISSON = 3,
       LRT_MEDIUM_HETEROGENEOUS = 4   /* src/media/heterogeneous.cpp: sigma_t from a grid volume, delta tracking */ };
enum { LRT_BSDF_DIFFUSE = 0, LRT_BSDF_DIELECTRIC = 1, LRT_BSDF_BUMPMAP = 2, LRT_BSDF_NULL = 3 };
enum { LRT_TEX_RGB = 0, LRT_TEX_CHECKERBOARD = 1, LRT_TEX_BITMAP = 2 };
enum { LRT_PHASE_ISOTROPIC = 0, LRT_PHASE_HG = 1 };
enum { LRT_EMITTER_AREA = 0, LRT_EMITTER_ENVMAP = 1, LRT_EMITTER_CONSTANT = 2 };
enum { LRT_SHAPE_MESH = 0, LRT_SHAPE_RECTANGLE = 1 };
enum { LRT_RFILTER_BOX = 0, LRT_RFILTER_GAUSSIAN = 1, LRT_RFILTER_TENT = 2 };

/* ------------------------------------------------- scene description (POD)
 * The flattened, plugin-free form of a loaded scene.  Produced by the XML
 * loader (lrt_scene_desc_get) and accepted by lrt_scene_from_desc; it is also
 * the data format the CPU oracle under oracle/ consumes, so that both sides
 * see byte-identical inputs.  All geometry is in world space.               */

typedef struct {
    int32_t  kind;            /* LRT_SHAPE_*                                       */
    uint32_t first_face;      /* range into faces[]                                */
    uint32_t n_faces;
    int32_t  bsdf;            /* index into bsdfs[], -1: none                      */
    int32_t  emitter;         /* index into emitters[] (area light), -1: none      */
    int32_t  interior_medium; /* index into media[], -1: none                      */
    int32_t  exterior_medium;
    int32_t  has_normals;     /* per-vertex shading normals present                */
    int32_t  has_texcoords;
    int32_t  flip_normals;
    float    to_world[16];    /* row-major; used by LRT_SHAPE_RECTANGLE sampling   */
} lrt_shape_desc;

typedef struct {
    int32_t type;             /* LRT_TEX_*                                         */
    float   color0[3];        /* RGB value (LRT_TEX_RGB) / checkerboard color0     */
    float   color1[3];
    float   to_uv[9];         /* row-major 3x3 affine uv transform                 */
    int32_t width, height, channels;   /* bitmap: 1 or 3 channels                  */
    const float *data;        /* bitmap texels, linear, already rounded to the
                                 storage precision the reference uses (fp16 for
                                 8/16-bit inputs: src/textures/bitmap.cpp:268-283) */
} lrt_texture_desc;

typedef struct {
    int32_t type;             /* LRT_BSDF_*                                        */
    int32_t reflectance;      /* texture index (diffuse)                           */
    float   eta;              /* int_ior / ext_ior (dielectric)                    */
    int32_t nested;           /* bumpmap: nested bsdf index                        */
    int32_t texture;          /* bumpmap: height texture index                     */
    float   scale;            /* bumpmap scale                                     */
} lrt_bsdf_desc;

typedef struct {
    float   sigma_t[3];
    float   albedo[3];
    float   scale;
    int32_t has_spectral_extinction;
    int32_t sample_emitters;
    int32_t phase;            /* LRT_PHASE_*                                       */
    float   g;
    char    id[64];           /* XML id, used to form parameter keys               */
    /* --- bio media (fork): element-competition sampling of the 5-argument Medium::sample_interaction,
       used by the biovolpath integrators only; path / volpath / prbvolpath see the fields above.      */
    int32_t type;             /* LRT_MEDIUM_*                                      */
    float   layer_limit[4];   /* liver / glissonCapsule: layer1Limit..layer4Limit (liver.cpp:143-146)     */
    float   sigma_collagen[4][3];  /* per layer, in the channel order the medium STORES them: the plugins read
                                 G from "..._B" and B from "..._G" (liver.cpp:148-166)                    */
    float   sigma_elastin[4][3];   /* layers 1-2 swapped the same way, layers 3-4 not (liver.cpp:168-186) */
    float   sigma_blood[3], sigma_bile[3], sigma_lipid_water[3];   /* liver / parenchyma                 */
    float   sigma_hepatocity;
    /* --- heterogeneous medium: sigma_t = scale * gridvolume (src/volumes/grid.cpp: one channel, trilinear, clamp), albedo
       constant; majorant = scale * grid_max.  The 4-argument sample_interaction of src/render/medium.cpp:40-82.        */
    int32_t grid_res[3];      /* x, y, z; all 0 when there is no grid                */
    float   grid_to_local[12];/* world -> the grid's unit cube [0,1]^3, rows 0..2 (Volume::m_to_local)            */
    float   grid_bbox_min[3], grid_bbox_max[3];   /* world-space bounds of that cube (Volume::update_bbox)       */
    float   grid_max;         /* maximum of the grid values (Volume::max)            */
    const float *grid_data;   /* x fastest: data[(z * res_y + y) * res_x + x]         */
} lrt_medium_desc;

typedef struct {
    int32_t type;             /* LRT_EMITTER_*                                     */
    float   radiance[3];      /* area / constant                                   */
    int32_t shape;            /* area: owning shape                                */
    float   scale;            /* envmap                                            */
    float   to_world[16];     /* envmap, row-major                                 */
    int32_t width, height;    /* envmap: ORIGINAL bitmap resolution (w, h)         */
    const float *data;        /* envmap: h * w * 3 linear RGB floats               */
} lrt_emitter_desc;

/* samplers: src/samplers/independent.cpp (PCG32 stream per lane), src/samplers/ldsampler.cpp ((0,2)-sequence,
   TEA-shuffled permutation + per-pixel scramble) */
enum { LRT_SAMPLER_INDEPENDENT = 0, LRT_SAMPLER_LD = 1 };

typedef struct {
    float   to_world[16];     /* camera-to-world, row-major                        */
    float   fov_x;            /* horizontal field of view, degrees                 */
    float   near_clip, far_clip;
    int32_t medium;           /* medium the sensor sits in, -1: none               */
    float   principal_point_offset_x, principal_point_offset_y;   /* perspective.cpp:147-150,214-221 */
    int32_t pad;
} lrt_sensor_desc;

typedef struct {
    int32_t width, height;
    int32_t crop_offset_x, crop_offset_y, crop_width, crop_height;
    int32_t has_alpha;        /* pixel_format rgba                                 */
    int32_t rfilter;          /* LRT_RFILTER_*                                     */
    float   rfilter_param;    /* gaussian: stddev; tent: radius                    */
} lrt_film_desc;

typedef struct {
    int32_t type;             /* LRT_INTEGRATOR_*                                  */
    int32_t max_depth;        /* -1: unbounded = 65535 (so is any larger value)    */
    int32_t rr_depth;
    int32_t hide_emitters;
} lrt_integrator_desc;

typedef struct {
    uint32_t n_vertices, n_faces, n_shapes, n_bsdfs, n_textures, n_media, n_emitters;
    const float    *positions;    /* 3 * n_vertices                                */
    const float    *normals;      /* 3 * n_vertices (zero where absent)            */
    const float    *texcoords;    /* 2 * n_vertices (zero where absent)            */
    const uint32_t *faces;        /* 3 * n_faces, global vertex ids                */
    const uint32_t *face_shape;   /* n_faces                                       */
    const lrt_shape_desc   *shapes;
    const lrt_bsdf_desc    *bsdfs;
    const lrt_texture_desc *textures;
    const lrt_medium_desc  *media;
    const lrt_emitter_desc *emitters;
    lrt_sensor_desc     sensor;
    lrt_film_desc       film;
    lrt_integrator_desc integrator;
    uint32_t sample_count;        /* sampler sample_count (default spp)            */
    uint32_t sampler_seed;        /* sampler `seed` property (m_base_seed)         */
    uint32_t sampler_type;        /* LRT_SAMPLER_*; ld rounds spp up to 4, 16, 64, 256, 1024, ... */
    uint32_t samples_per_pass;    /* integrator `samples_per_pass` (integrator.cpp:176-184), 0: unset; renders of more than
                                     2^32 - 1 samples are split into passes as well (integrator.cpp:275-293)  */
    uint32_t use_spectral_mis;    /* volpathmis `use_spectral_mis` (volpathmis.cpp:47, default true)                */
    uint32_t pad;
} lrt_scene_desc;

/* ----------------------------------------------------------- render call */
typedef struct {
    int32_t  integrator;      /* LRT_INTEGRATOR_*, -1: use the scene's             */
    int32_t  max_depth;       /* -2: use the scene's                               */
    int32_t  rr_depth;        /* -1: use the scene's                               */
    int32_t  hide_emitters;   /* -1: use the scene's                               */
    uint32_t spp;             /* 0: use the scene's sample_count; with the ld sampler
                                 rounded up to 4, 16, 64, 256, ... (ldsampler.cpp:83-93)  */
    uint32_t seed;
    /* image-tile sharding (multi-GPU): this call renders only the 32x32 pixel
       tiles t with t % tile_count == tile_rank, into a full-size zeroed film. */
    uint32_t tile_rank, tile_count;
    int32_t  device;          /* HIP device ordinal                                */
    int32_t  output_on_device;/* film_raw / image are device pointers              */
    int32_t  grad_medium;     /* lrt_render_backward: index (into media[]) of the medium whose sigma_t / albedo / g
                                 the gradients refer to; -1: the sum over all media (one shared parameter set) */
    int32_t  pad;
} lrt_render_opts;

typedef struct {
    uint64_t n_samples;       /* camera samples traced                             */
    uint64_t n_iter;          /* path-loop iterations executed (all paths)         */
    uint64_t n_shadow;        /* NEE visibility / march ray queries traced         */
    uint64_t n_launches;      /* iteration-kernel launches                         */
    uint64_t n_records;       /* path records read by those launches (<= n_iter:
                                 trips retired by look-ahead move no record)        */
    double   kernel_ms;       /* sum of iteration-kernel durations (HIP events)    */
    double   total_ms;        /* whole lrt_render device time (HIP events)         */
    uint64_t lds_resident;    /* 1: the kernels ran on the LDS-resident BVH image (1024-thread workgroups);
                                 0: the mesh did not fit, BVH in global memory (256-thread workgroups)  [v103] */
} lrt_render_stats;

typedef struct {
    float d_sigma_t[3];       /* homogeneous medium: w.r.t. the `sigma_t` property (before `scale`), per channel.
                                 heterogeneous medium [v104]: channel k's share of d / d(scale); the three add up to the derivative
                                 w.r.t. the medium's `scale` (sigma_t(p) = scale * grid(p); the majorant is detached,
                                 src/media/heterogeneous.cpp parameters_changed) */
    float d_albedo[3];
    float d_g;
} lrt_param_grads;

typedef struct lrt_scene lrt_scene;

LRT_API const char *lrt_last_error(void);
LRT_API int         lrt_version(void);

LRT_API lrt_status lrt_scene_load_xml(const char *path, const char *const *defines,
                                      int n_defines, lrt_scene **out);
LRT_API lrt_status lrt_scene_load_xml_string(const char *xml, const char *base_dir,
                                             const char *const *defines, int n_defines,
                                             lrt_scene **out);
LRT_API lrt_status lrt_scene_from_desc(const lrt_scene_desc *desc, lrt_scene **out);
LRT_API const lrt_scene_desc *lrt_scene_desc_get(const lrt_scene *scene);
LRT_API void       lrt_scene_free(lrt_scene *scene);

/* film_raw: crop_h * crop_w * C floats (C = 5 if has_alpha else 4: R,G,B,[A],W),
 * image:    crop_h * crop_w * (4 if has_alpha else 3) developed floats.
 * Either may be NULL.  Both are overwritten: the film is cleared before the samples are accumulated (ImageBlock::clear, as
 * Film::prepare / SamplingIntegrator::render do before a render); with opts->output_on_device the two pointers are device
 * memory and the call returns after the library's stream has finished with them.  A tile shard (tile_count > 1) fills only
 * its own tiles' samples into the full-size film: shard films add up to the unsharded film.  */
LRT_API lrt_status lrt_render(lrt_scene *scene, const lrt_render_opts *opts,
                              float *film_raw, float *image);
LRT_API lrt_status lrt_render_stats_get(const lrt_scene *scene, lrt_render_stats *out);
LRT_API lrt_status lrt_film_develop(lrt_scene *scene, const float *film_raw, float *image,
                                    int on_device);

/* Test hook: per-lane radiance of wavefront lanes [lane_begin, lane_begin+n)
 * of the render described by opts, without film accumulation (of its FIRST pass
 * when `samples_per_pass` or the 2^32 - 1 limit splits the render: lanes then
 * count spp_per_pass samples per pixel).  out: n * 4 floats {R, G, B, valid}.  */
LRT_API lrt_status lrt_render_samples(lrt_scene *scene, const lrt_render_opts *opts,
                                      uint64_t lane_begin, uint32_t n, float *out);

/* PRB adjoint: d(sum(image * grad_image)) / d(sigma_t, albedo, g) of medium opts->grad_medium (the reference
 * differentiates whichever parameters have gradients enabled: one call per medium gives the same numbers).
 * Homogeneous and [v104] heterogeneous media (null collisions, src/python/python/ad/integrators/prbvolpath.py:178-196,
 * 404-415); the bio media are rejected (LRT_ERR_UNSUPPORTED): the reference's prbvolpath reads them as homogeneous. */
LRT_API lrt_status lrt_render_backward(lrt_scene *scene, const lrt_render_opts *opts,
                                       const float *grad_image, lrt_param_grads *out);

/* One process, several devices [v104] (SURVEY.md 8e through the C ABI; no reference counterpart: the reference renders on one device):
 * device i of the list (device_ids, or 0 .. n_devices - 1 when NULL) renders the 32x32 tiles t with t % n_devices == i into its own
 * full-size zeroed film with GLOBAL lane ids - one host thread and one stream per device -, ONE ncclAllReduce (RCCL, bound at run time
 * from /opt/rocm/lib/librccl.so; LRT_RCCL_LIBRARY overrides) sums the films, the first device develops.  film_raw / image as in
 * lrt_render (with output_on_device: pointers on the first device of the list).  The result is the image lrt_render gives (up to the
 * order of the float sums).  opts->tile_rank / tile_count / device are ignored (must be unset).  A list that names ONE device several
 * times is a rehearsal for boxes with a single GPU: the peers' films are added on that device, no collective runs.
 * lrt_render_backward_multi: the same sharding for the PRB adjoint; the 7 gradient sums are reduced with one all-reduce.
 * What the reference-side C++ plugin calls from SamplingIntegrator::render / render_backward
 * (include/mitsuba/render/integrator.h:74, :253) to use a whole node without a second process. */
LRT_API lrt_status lrt_render_multi(lrt_scene *scene, const lrt_render_opts *opts, int n_devices, const int *device_ids,
                                    float *film_raw, float *image);
LRT_API lrt_status lrt_render_backward_multi(lrt_scene *scene, const lrt_render_opts *opts, int n_devices, const int *device_ids,
                                             const float *grad_image, lrt_param_grads *out);

/* Test hook [v104]: the device's own log / exp / sincos / atan2 / acos / log2 kernels (csrc/dmath.h: the arithmetic every path value
 * goes through, standing in for Dr.Jit's dr::log / dr::exp / ... of include/mitsuba/core/math.h users) and its division / sqrt / rcp,
 * one value per lane.  fn: 0 log(x), 1 exp(x), 2 sincos(x) -> out, out2, 3 atan2(y, x), 4 acos(x), 5 log2(x), 6 x / y, 7 sqrt(x), 8 1 / x. */
LRT_API lrt_status lrt_math_eval(int fn, const float *x, const float *y, uint32_t n, float *out, float *out2, int device);

/* SoA ray queries (layout mirrors RayHit of src/render/scene_native.inl:135-142).
 * Miss: t = +inf, prim = 0xffffffff.  any_hit: only t (0 on hit, +inf on miss). */
typedef struct { const float *ox, *oy, *oz, *dx, *dy, *dz, *tmax; } lrt_rays_soa;
typedef struct { float *t, *u, *v; uint32_t *prim; } lrt_hits_soa;
LRT_API lrt_status lrt_trace(lrt_scene *scene, const lrt_rays_soa *rays,
                             const lrt_hits_soa *hits, uint32_t n, int any_hit);

/* Keys follow mi.traverse(): "<medium id>.sigma_t.value" (3 floats),
 * "<medium id>.albedo.value" (3), "<medium id>.scale" (1),
 * "<medium id>.phase_function.g" (1; switches the phase to HG when != 0);
 * a `parenchyma` medium also has what src/media/parenchyma.cpp:154-160 traverses:
 * "<id>.sigma_blood.value", "<id>.sigma_bile.value", "<id>.sigma_lipid_water.value" (3 each)
 * and "<id>.sigma_hepatocity" (1).                                           */
LRT_API lrt_status lrt_param_set(lrt_scene *scene, const char *key, const float *v, int n);
LRT_API lrt_status lrt_param_get(const lrt_scene *scene, const char *key, float *v, int n);

/* Image files around the path (mi.Bitmap(path) / Bitmap.write, src/core/bitmap.cpp): 8/16-bit PNG and
 * scanline OpenEXR (NONE/ZIPS/ZIP/PIZ) readers, uncompressed float32 EXR writer, 8-bit sRGB PNG writer.  *data is
 * h * w * channels floats in R,G,B[,A] (or Y[,A]) order, PNG values in [0,1] as stored (no
 * gamma conversion); release it with lrt_image_free.  */
LRT_API lrt_status lrt_image_read(const char *path, int *width, int *height, int *channels, float **data);
LRT_API void       lrt_image_free(float *data);
LRT_API lrt_status lrt_image_write_exr(const char *path, int width, int height, int channels, const float *data);
/* 8-bit PNG of a LINEAR float image, the export step of LiverRenderer.py:383-385
 * (Bitmap.convert(RGBA, UInt8, srgb_gamma=True) + write): colour channels through the sRGB transfer
 * function, alpha linear, clamped to [0,1], rounded to nearest; 1 to 4 channels.  The reference's
 * blue-noise dithering of 8-bit conversions (src/core/struct.cpp:823-845) is not applied: values agree
 * with its PNGs to within one code value.                                                    */
LRT_API lrt_status lrt_image_write_png(const char *path, int width, int height, int channels, const float *data);

/* ---------------------------------------------------------------------------------------------------------------
 * Learned subsurface model, network stage only: the shape-adaptive scatter network of
 * include/mitsuba/render/scattereigen.h:249-480 (ScatterModelSimShared<3, 4, 64, 64>::run): feature MLP 23 -> 64 -> 64 -> 64,
 * absorption head 64 -> 32 -> 1 (sigmoid, one sampler draw), decoder (4 Gaussian latents + 64) -> 64 -> 64 -> 64 -> 3, result
 * mapped from the tangent frame of -in_dir to world space and scaled by 1 / fit_scale.  docs/SUBSURFACE_NOTES.md explains
 * why the plugin around it (src/subsurface/vaescatter.cpp) is not built.
 *
 * Weights: one float32 blob in this order (row-major matrices, rows = outputs), as the .bin files of
 * pysrc/outputs/vae3d/models/<model>/variables/ hold them, preceded by the normalisation statistics of
 * pysrc/outputs/vae3d/datasets/<dataset>/train/data_stats.json ("effAlbedo", "g", "mlsPoly3"):                       */
#define LRT_VAE_STATS        0      /* albedo mean, albedo 1/std, g mean, g 1/std, shape mean[20], shape 1/std[20]        */
#define LRT_VAE_PRE0_W      44      /* shared_preproc_mlp_2_shapemlp_fcn_0_weights [64][23], then _biases [64]            */
#define LRT_VAE_PRE1_W    (LRT_VAE_PRE0_W + 64 * 23 + 64)      /* fcn_1 [64][64] + [64]                                   */
#define LRT_VAE_PRE2_W    (LRT_VAE_PRE1_W + 64 * 64 + 64)      /* fcn_2 [64][64] + [64]                                   */
#define LRT_VAE_ABS0_W    (LRT_VAE_PRE2_W + 64 * 64 + 64)      /* absorption_mlp_fcn_0 [32][64] + [32]                    */
#define LRT_VAE_ABSD_K    (LRT_VAE_ABS0_W + 32 * 64 + 32)      /* absorption_dense_kernel [32] + bias [1]                 */
#define LRT_VAE_DEC0_W    (LRT_VAE_ABSD_K + 32 + 1)            /* scatter_decoder_fcn_fcn_0 [64][68] + [64]               */
#define LRT_VAE_DEC1_W    (LRT_VAE_DEC0_W + 64 * 68 + 64)      /* fcn_1 [64][64] + [64]                                   */
#define LRT_VAE_DEC2_W    (LRT_VAE_DEC1_W + 64 * 64 + 64)      /* fcn_2 [64][64] + [64]                                   */
#define LRT_VAE_OUT_K     (LRT_VAE_DEC2_W + 64 * 64 + 64)      /* scatter_dense_2_kernel [3][64] + bias [3]               */
#define LRT_VAE_N_FLOATS  (LRT_VAE_OUT_K + 3 * 64 + 3)
typedef struct lrt_vae_model lrt_vae_model;
LRT_API lrt_status lrt_vae_model_create(const float *blob, uint64_t n_floats, lrt_vae_model **out);
LRT_API void       lrt_vae_model_free(lrt_vae_model *model);
/* n samples; sample i uses in_pos[3i..], in_dir[3i..], poly_coeffs[20i..] (the order-3 polynomial of the surface around in_pos,
 * in the network's "LS" space) and its own PCG32 stream seeded like a render lane (TEA(seed, i)).  Outputs: out_pos[3i..] and
 * out_absorption[i] (1: the sample was absorbed, out_pos = in_pos; 0: out_pos is the predicted exit point).  Host pointers. */
LRT_API lrt_status lrt_vae_scatter(lrt_vae_model *model, uint32_t n, const float *in_pos, const float *in_dir,
                                   const float *poly_coeffs, const float albedo[3], float g, float ior,
                                   const float sigma_t[3], float fit_scale, uint32_t seed,
                                   float *out_pos, float *out_absorption, int device);

#ifdef __cplusplus
}
#endif
#endif /* LIVERRT_H */
