/*
 * orc.h -- C interface of the CPU oracle (liborc.so).  TEST INFRASTRUCTURE ONLY.
 *
 * The oracle is a scalar CPU restatement of the reference's path / volpath
 * sample loop with `llvm_ad_rgb` lane semantics (one PCG32 stream per wavefront
 * lane, lane -> pixel mapping of src/render/integrator.cpp:321-338).  It is the
 * checker for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product library never links or calls it.
 *
 * PARITY PINNING: the reference cannot be built or imported in this
 * environment (Dr.Jit/nanobind/... are empty submodules), so the
 * oracle is pinned by the reference's own in-tree known answers
 * (tests/test_oracle_pins.py): TEA vectors (src/core/tests/test_random.py:9-26),
 * the Cornell-box radiance known answer (src/integrators/tests/test_integrators.py:28-53),
 * the staircase ray depths (src/render/tests/test_kdtrees.py:47-83), phase
 * function values (src/phase/tests/test_isotropic.py:11-21), published PCG32
 * vectors, plus analytic furnace tests.  volpath radiance and PRB gradients
 * have no in-tree numeric fixture: "parity unpinned by the reference" for those,
 * pinned by analytic tests + finite differences instead.
 */
#ifndef ORC_H
#define ORC_H
#include "../include/liverrt.h"   /* scene description POD + option structs only */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

typedef struct {
    uint64_t n_samples, n_iter;
    uint64_t n_shadow;          /* all NEE ray queries the reference semantics perform */
    uint64_t n_shadow_needed;   /* those whose outcome can influence the result         */
} orc_stats;

orc_scene *orc_scene_create(const lrt_scene_desc *desc);
void       orc_scene_free(orc_scene *s);
const char *orc_last_error(void);

/* Bio transport (biovolpath + liver / parenchyma / glissonCapsule): which reading of the reference's source the oracle
 * follows (oracle/orc_bio.h): 0 = the JIT variants' lane semantics (llvm_ad_rgb / cuda_rgb; default, what hip_ad_rgb
 * implements), 1 = scalar_rgb (the reference's CPU renders).  `biovolpath06` is scalar-only and ignores it. */
void orc_scene_set_bio_reading(orc_scene *s, int scalar);

/* Medium / phase parameter edits (mirror of lrt_param_set on the copied desc). */
int orc_param_set(orc_scene *s, const char *key, const float *v, int n);

/* Full render (wavefront lane order, deterministic film accumulation in lane
 * order).  film_raw: h*w*C, image: h*w*(3|4); either may be NULL.
 * n_threads <= 0: hardware concurrency.  */
int orc_render(orc_scene *s, const lrt_render_opts *opts, int n_threads,
               float *film_raw, float *image, orc_stats *stats);

/* Per-lane radiance {R,G,B,valid} of lanes [lane_begin, lane_begin+n). */
int orc_render_samples(orc_scene *s, const lrt_render_opts *opts, uint64_t lane_begin,
                       uint32_t n, int n_threads, float *out, orc_stats *stats);

/* Scalar-variant tiling (src/render/integrator.cpp:190-273,399-434): 32x32
 * spiral-free row-major blocks, per-pixel seeding `seed + block_id*1024 + i`,
 * PCG32 default stream.  Used only as the CPU baseline workload of bench.py. */
int orc_render_scalar(orc_scene *s, const lrt_render_opts *opts, int n_threads,
                      float *film_raw, float *image, orc_stats *stats);

/* PRB adjoint (src/python/python/ad/integrators/prbvolpath.py:96-444,
 * common.py:625-783): gradient of sum(image * grad_image) w.r.t. medium 0. */
int orc_render_backward(orc_scene *s, const lrt_render_opts *opts, int n_threads,
                        const float *grad_image, lrt_param_grads *out);

/* Ray queries.  brute_force != 0 tests every triangle. */
int orc_trace(orc_scene *s, const lrt_rays_soa *rays, const lrt_hits_soa *hits,
              uint32_t n, int any_hit, int brute_force);

/* Unit-level entry points for pinning tests. */
void  orc_tea32(uint32_t v0, uint32_t v1, int rounds, uint32_t *out0, uint32_t *out1);
void  orc_ld_sample(uint32_t sample_count, uint32_t scramble_seed, uint32_t sample_index, uint32_t dim, int two_d, float *out);
uint32_t orc_ld_round_sample_count(uint32_t spp);
uint32_t orc_permute(uint32_t i, uint32_t n, uint32_t seed);
float orc_tea_float32(uint32_t v0, uint32_t v1, int rounds);
double orc_tea_float64(uint32_t v0, uint32_t v1, int rounds);
void  orc_pcg32_u32(uint64_t initstate, uint64_t initseq, uint32_t n, uint32_t *out);
void  orc_lane_stream(uint32_t base_seed, uint32_t seed, uint32_t lane, uint32_t n, float *out);
void  orc_math_eval(int fn, const float *x, const float *y, uint32_t n, float *out, float *out2);
void  orc_hg_sample(float g, const float wi[3], float u1, float u2, float wo[3], float *pdf);
float orc_hg_eval(float g, float cos_theta);
void  orc_square_to_cosine_hemisphere(float u1, float u2, float out[3]);
void  orc_square_to_uniform_sphere(float u1, float u2, float out[3]);
void  orc_fresnel(float cos_theta_i, float eta, float out[4]);
void  orc_envmap_sample(orc_scene *s, float u1, float u2, float ref[3], float d[3], float *pdf, float rgb[3]);
float orc_envmap_pdf(orc_scene *s, const float d[3]);
void  orc_envmap_eval(orc_scene *s, const float d[3], float rgb[3]);
float orc_rfilter_eval(orc_scene *s, float x);
void  orc_bio_sample_interaction(orc_scene *s, int medium, const float o[3], const float d[3], float maxt, float sample,
                                 uint32_t channel, float depth, int jit, float out[9]);
void  orc_sample_ray(orc_scene *s, float px, float py, float o[3], float d[3], float *maxt);

#ifdef __cplusplus
}
#endif
#endif
