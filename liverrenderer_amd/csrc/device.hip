// Host driver of the wavefront renderer: device scene upload, queue management,
// kernel launches, timing.  Everything device-related lives behind this file so
// that the rest of the library is plain host C++.
#include "host_scene.h"
#include "device_scene.h"
#include "kernels_prb.h"
#include "kernels_vae.h"
#include "bvh.h"
#include <cmath>
#include <array>
#include <cstring>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <chrono>
#include <exception>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and enums only: the functions are bound with dlopen (librccl.so is not a link-time dependency)

namespace lrt {

#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

template <typename T> static T *dev_upload(const T *src, size_t n, hipStream_t st) {
    T *p = nullptr; size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIP_CHECK(hipMalloc((void **) &p, bytes));
    if (n) HIP_CHECK(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, st));
    return p;
}

#define LRT_LAUNCH_SLOTS 64
#define LRT_WIDE_BLOCK 768          // workgroup size of the render kernels with wide path records (volpathmis, volpath with heterogeneous media) on the LDS BVH

struct DeviceScene {
    int device = 0;
    hipStream_t stream = nullptr;
    DScene sc{};                           // host copy of the scene record; the kernels read d_sc (constant address space)
    DScene *d_sc = nullptr;
    // launch-argument slots (DLaunch, read by the kernels through the constant address space): a ring in device memory
    // filled from a pinned host ring, one slot per launch; a slot is reused only after LRT_LAUNCH_SLOTS further launches
    // of the same in-order stream, i.e. long after its kernel has finished
    DLaunch *d_launch = nullptr, *h_launch = nullptr; uint32_t launch_next = 0;
    std::vector<void *> allocs;            // everything freed in the destructor
    // wavefront workspace
    uint32_t capacity = 0;
    DPathStreams q[2]{};
    float4 *dl[2] = { nullptr, nullptr }; float4 *L_buf = nullptr; uint32_t prb_capacity = 0; uint64_t l_buf_lanes = 0;   // PRB: delta_L streams, primal radiance
    double *d_grads = nullptr; float *wfilm = nullptr; size_t wfilm_floats = 0; float *grad_image = nullptr; size_t grad_floats = 0;
    DCounters *counters = nullptr;
    DCounters *h_counters = nullptr;       // pinned
    float *film = nullptr; size_t film_floats = 0;
    float *image = nullptr; size_t image_floats = 0;
    unsigned long long *pass_state[2] = { nullptr, nullptr }; uint64_t pass_state_lanes = 0;   // multi-pass renders: per-lane PCG32 states
    const unsigned long long *cur_pass_in = nullptr; unsigned long long *cur_pass_out = nullptr;
    uint32_t *pixel_slot = nullptr;         // inverse of pixel_list (pixel -> index in the list), tile-sharded renders
    uint32_t *pixel_list = nullptr; uint32_t pixel_list_rank = 0xffffffffu, pixel_list_count = 0, n_owned_pixels = 0;
    uint32_t *halo_list = nullptr; uint32_t halo_rank = 0xffffffffu, halo_count = 0, halo_width = 0, n_halo_pixels = 0;   // own tiles dilated by `halo_width` pixels (PRB weight film)
    std::vector<hipEvent_t> ev_pool;
    std::vector<DMedium> h_media; DMedium *d_media = nullptr;
    std::vector<DBioMedium> h_bio; DBioMedium *d_bio = nullptr;
    std::vector<DHetMedium> h_het; DHetMedium *d_het = nullptr; std::vector<float *> het_data; bool has_het = false, has_non_bio = false, need_mis = false, mis_alloc = false;
    bool has_area_emitter = false;         // decides the record layout: only an area emitter's pdf reads the last scatter position (kernels.h, store_state)
    bool prb_null = false;                 // prbvolpath.py:84-91 `handle_null_scattering`: a heterogeneous medium is attached to a shape
    DLdsInfo lds{}; bool use_lds = false; int n_cus = 256; int bvh_leaf = 4;

    template <typename T> T *track(T *p) { allocs.push_back((void *) p); return p; }
    void release(void *p) {                // free one tracked allocation now (a workspace that is being replaced by a larger one)
        if (!p) return;
        for (auto it = allocs.begin(); it != allocs.end(); ++it) if (*it == p) { allocs.erase(it); break; }
        (void) hipFree(p);
    }
    ~DeviceScene() {
        for (void *p : allocs) (void) hipFree(p);
        for (auto e : ev_pool) (void) hipEventDestroy(e);
        if (h_counters) (void) hipHostFree(h_counters);
        if (h_launch) (void) hipHostFree(h_launch);
        if (stream) (void) hipStreamDestroy(stream);
    }
};

void device_scene_destroy(DeviceScene *d) { delete d; }

static void m4_mul(const float *a, const float *b, float *r) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float s = a[4 * i] * b[j]; for (int k = 1; k < 4; ++k) s = fmaf(a[4 * i + k], b[4 * k + j], s); r[4 * i + j] = s; }
}
static void m4_ident(float *m) { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.f : 0.f; }

// include/mitsuba/render/sensor.h:234-269 + include/mitsuba/core/transform.h:393-410:
// sample_to_camera = inverse of (scale * translate * scale * translate * perspective), built
// from the analytic inverses exactly as Transform::inverse_transpose composes them.
static void build_camera(const lrt_scene_desc &d, DCamera &cam, DFilm &film) {
    const lrt_film_desc &F = d.film; const lrt_sensor_desc &C = d.sensor;
    float fw = (float) F.width, fh = (float) F.height;
    float rsx = (float) F.crop_width / fw, rsy = (float) F.crop_height / fh, rox = (float) F.crop_offset_x / fw, roy = (float) F.crop_offset_y / fh;
    float aspect = fw / fh, recip = 1.f / (C.far_clip - C.near_clip);
    float tn = (float) tan((double) (C.fov_x * .5f) * (3.14159265358979323846 / 180.0));
    (void) recip;
    float S1i[16], T1i[16], S2i[16], T2i[16], Pit[16], t0[16], t1[16], t2[16], it[16];
    m4_ident(S1i); S1i[0] = 1.f / (1.f / rsx); S1i[5] = 1.f / (1.f / rsy); S1i[10] = 1.f / 1.f;
    m4_ident(T1i); T1i[12] = rox; T1i[13] = roy; T1i[14] = -0.f;
    m4_ident(S2i); S2i[0] = 1.f / -0.5f; S2i[5] = 1.f / (-0.5f * aspect); S2i[10] = 1.f;
    m4_ident(T2i); T2i[12] = 1.f; T2i[13] = 1.f / aspect; T2i[14] = -0.f;
    // inverse of the perspective matrix, transposed
    float Pinv[16]; m4_ident(Pinv); Pinv[0] = tn; Pinv[5] = tn; Pinv[10] = 0.f; Pinv[15] = 1.f / C.near_clip; Pinv[11] = 1.f;
    Pinv[14] = (C.near_clip - C.far_clip) / (C.far_clip * C.near_clip);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Pit[4 * i + j] = Pinv[4 * j + i];
    m4_mul(S1i, T1i, t0); m4_mul(t0, S2i, t1); m4_mul(t1, T2i, t2); m4_mul(t2, Pit, it);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) cam.s2c[4 * i + j] = it[4 * j + i];
    memcpy(cam.to_world, C.to_world, sizeof(float) * 12);
    cam.near_clip = C.near_clip; cam.far_clip = C.far_clip; cam.medium = C.medium;
    cam.ppo_x = fw * C.principal_point_offset_x / (float) F.crop_width; cam.ppo_y = fh * C.principal_point_offset_y / (float) F.crop_height;

    film.width = F.crop_width; film.height = F.crop_height; film.crop_offset_x = F.crop_offset_x; film.crop_offset_y = F.crop_offset_y;
    film.scale_x = 1.f / (float) F.crop_width; film.scale_y = 1.f / (float) F.crop_height;
    film.offset_x = -(float) F.crop_offset_x * film.scale_x; film.offset_y = -(float) F.crop_offset_y * film.scale_y;
    film.has_alpha = F.has_alpha; film.channels = F.has_alpha ? 5 : 4; film.rfilter = F.rfilter;
    film.rf_radius = 0.5f; film.rf_inv_radius = 1.f;
    if (F.rfilter == LRT_RFILTER_GAUSSIAN) {             // src/rfilters/gaussian.cpp:52-95
        float stddev = F.rfilter_param; film.rf_radius = 4.f * stddev;
        static const double coeff[10] = { 9.992604880e-1, -4.977025247e-1, 1.222248550e-1, -1.932406282e-2, 2.136713061e-3,
                                          -1.679873860e-4, 9.202145248e-6, -3.329417433e-7, 7.128382794e-9, -6.821193280e-11 };
        double sc = 1; for (int i = 0; i < 10; ++i) { film.rf_coeff[i] = (float) (coeff[i] * sc); sc /= (double) stddev * (double) stddev; }
        float x = film.rf_radius * film.rf_radius, x2 = x * x, x4 = x2 * x2, x8 = x4 * x4; const float *c = film.rf_coeff;
        float a0 = fmaf(x, c[1], c[0]), a1 = fmaf(x, c[3], c[2]), a2 = fmaf(x, c[5], c[4]), a3 = fmaf(x, c[7], c[6]), a4 = fmaf(x, c[9], c[8]);
        float b0 = fmaf(x2, a1, a0), b1 = fmaf(x2, a3, a2), c0 = fmaf(x4, b1, b0);
        film.rf_coeff[0] -= fmaf(x8, a4, c0);
    } else if (F.rfilter == LRT_RFILTER_TENT) { film.rf_radius = F.rfilter_param; film.rf_inv_radius = 1.f / film.rf_radius; }
    film.fn = (int) ceilf(film.rf_radius - .5f); film.fcount = 2 * film.fn + 1;
}

static void inverse3(const float *m16, float *out9) {    // double-precision inverse of the linear part
    double m[3][3]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = m16[4 * i + j];
    double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    double id = 1.0 / det;
    out9[0] = (float) ((m[1][1] * m[2][2] - m[1][2] * m[2][1]) * id); out9[1] = (float) ((m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id); out9[2] = (float) ((m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id);
    out9[3] = (float) ((m[1][2] * m[2][0] - m[1][0] * m[2][2]) * id); out9[4] = (float) ((m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id); out9[5] = (float) ((m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id);
    out9[6] = (float) ((m[1][0] * m[2][1] - m[1][1] * m[2][0]) * id); out9[7] = (float) ((m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id); out9[8] = (float) ((m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id);
}

static uint32_t log2i_ceil(uint32_t v) { uint32_t r = 0; while ((1u << r) < v) ++r; return r; }

// include/mitsuba/core/distr_2d.h:403-510 (normalised, single slice); levels are
// concatenated with 16-byte aligned offsets so that a 2x2 patch is one float4 load.
static void build_hierarchy(const std::vector<float> &lum, uint32_t w, uint32_t h, DEnv &E, std::vector<float> &out) {
    uint32_t npx = w - 1, npy = h - 1;
    E.patch_size[0] = 1.f / (float) npx; E.patch_size[1] = 1.f / (float) npy;
    E.inv_patch_size[0] = (float) npx; E.inv_patch_size[1] = (float) npy;
    E.max_patch[0] = npx - 1; E.max_patch[1] = npy - 1;
    uint32_t max_level = log2i_ceil(std::max(npx, npy));
    struct L { uint32_t w, h, off; };
    std::vector<L> lv; uint32_t total = 0;
    auto add = [&](uint32_t lw, uint32_t lh) { lv.push_back({ lw, lh, total }); total += (lw * lh + 3u) & ~3u; };
    add(w, h);
    uint32_t lx = npx, ly = npy;
    for (int level = (int) max_level; level >= 0; --level) { lx += lx & 1u; ly += ly & 1u; add(lx, ly); lx >>= 1; ly >>= 1; }
    if (lv.size() > LRT_MAX_HIER_LEVELS) throw std::runtime_error("environment map too large for the sampling hierarchy");
    out.assign(total, 0.f);
    auto index = [](uint32_t x, uint32_t y, uint32_t width) { return ((x & 1u) | (((x & ~1u) | (y & 1u)) << 1)) + ((y & ~1u) * width); };
    float *l0 = &out[lv[0].off], *l1 = &out[lv[1].off];
    const float *in = lum.data(); double sum = 0.0;
    for (uint32_t y = 0; y < npy; ++y) { for (uint32_t x = 0; x < npx; ++x) { float avg = .25f * (in[0] + in[1] + in[w] + in[w + 1]); sum += (double) avg; l1[index(x, y, lv[1].w)] = avg; ++in; } ++in; }
    float scale = (float) ((double) (npx * npy) / sum);
    for (uint32_t i = 0; i < w * h; ++i) l0[i] = lum[i] * scale;
    for (uint32_t i = 0; i < lv[1].w * lv[1].h; ++i) l1[i] *= scale;
    lx = npx; ly = npy;
    for (uint32_t level = 2; level <= max_level + 1; ++level) {
        const float *a = &out[lv[level - 1].off]; float *b = &out[lv[level].off];
        lx = (lx + 1u) >> 1; ly = (ly + 1u) >> 1;
        for (uint32_t y = 0; y < ly; ++y) for (uint32_t x = 0; x < lx; ++x) { const float *d0 = a + index(x * 2, y * 2, lv[level - 1].w); b[index(x, y, lv[level].w)] = d0[0] + d0[1] + d0[2] + d0[3]; }
    }
    E.n_levels = (int) lv.size();
    for (size_t i = 0; i < lv.size(); ++i) { E.level_offset[i] = lv[i].off; E.level_width[i] = lv[i].w; }
}

// log10 of the hepatocyte coefficient with the device's own arithmetic (liver.cpp:376 `dr::log2(att + 1) / dr::log2(10)`)
__global__ void k_bio_prepare(DBioMedium *bio, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bio[i].log10_hep = m_log2(bio[i].hepatocity + 1.f) / m_log2(10.f);
}

// Per-medium constants of the real-scattering weight (volpath.cpp:261-265), evaluated once with the device's own arithmetic:
// sigma_s / mean(sigma_t / combined) (spectral branch) and sigma_s / sigma_t, with sigma_s = sigma_t * albedo, combined = sigma_t.
__global__ void k_medium_prepare(DMedium *media, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DMedium &M = media[i];
    const V3 st(M.sigma_t[0], M.sigma_t[1], M.sigma_t[2]), ss = st * V3(M.albedo[0], M.albedo[1], M.albedo[2]);
    const V3 ws = ss / mean3(st / st), wp = ss / st;
    M.w_spec[0] = ws.x; M.w_spec[1] = ws.y; M.w_spec[2] = ws.z; M.w_plain[0] = wp.x; M.w_plain[1] = wp.y; M.w_plain[2] = wp.z;
}

static void upload_media(DeviceScene *D, const lrt_scene_desc &d) {
    D->h_bio.assign(std::max<uint32_t>(d.n_media, 1), DBioMedium{});
    for (uint32_t i = 0; i < d.n_media; ++i) {
        const lrt_medium_desc &M = d.media[i]; DBioMedium &o = D->h_bio[i];
        o.type = M.type; o.has_spectral_extinction = M.has_spectral_extinction;
        for (int l = 0; l < 4; ++l) { o.layer_limit[l] = M.layer_limit[l]; for (int k = 0; k < 3; ++k) { o.collagen[l][k] = M.sigma_collagen[l][k]; o.elastin[l][k] = M.sigma_elastin[l][k]; } }
        for (int k = 0; k < 3; ++k) { o.blood[k] = M.sigma_blood[k]; o.bile[k] = M.sigma_bile[k]; o.lipid_water[k] = M.sigma_lipid_water[k]; }
        o.hepatocity = M.sigma_hepatocity; o.log10_hep = 0.f;
        if (M.type == LRT_MEDIUM_PARENCHYMA) { o.sigmat[0] = 77.2f / 255; o.sigmat[1] = 105.0f / 255; o.sigmat[2] = 149.0f / 255; }   // parenchyma.cpp:165
        else for (int k = 0; k < 3; ++k) o.sigmat[k] = M.sigma_t[k] * M.scale;                                                       // liver.cpp:204-209
    }
    HIP_CHECK(hipMemcpyAsync(D->d_bio, D->h_bio.data(), D->h_bio.size() * sizeof(DBioMedium), hipMemcpyHostToDevice, D->stream));
    k_bio_prepare<<<1, 64, 0, D->stream>>>(D->d_bio, (uint32_t) std::min<size_t>(D->h_bio.size(), 64));
    HIP_CHECK(hipGetLastError());
    D->h_media.resize(std::max<uint32_t>(d.n_media, 1));
    for (uint32_t i = 0; i < d.n_media; ++i) {
        const lrt_medium_desc &M = d.media[i]; DMedium &o = D->h_media[i];
        for (int k = 0; k < 3; ++k) { o.sigma_t[k] = M.sigma_t[k] * M.scale; o.albedo[k] = M.albedo[k]; }   // homogeneous.cpp:121-126 eval_sigmat
        o.has_spectral_extinction = M.has_spectral_extinction; o.sample_emitters = M.sample_emitters; o.phase = M.phase; o.g = M.g;
        o.scale = M.scale; o.het = M.type == LRT_MEDIUM_HETEROGENEOUS ? 1 : 0;
        // Early NEE rejection (volpath_iteration): an infinite emitter's sample lies at distance >= 2 r (r: the scene's bounding sphere); the
        // kernel rejects when -log(1 - u) / sigma_t <= bound(dist), bound(x) = x * 0.998f - 1e-3f (monotone in x).  1 - u >= vmin implies that
        // for dist = 2 r, hence for every sample: vmin = exp(-sigma_t bound(2 r) (1 - 1e-4)) (1 + 1e-4), far outside the 3e-7 relative error
        // of the kernels' logarithm and the rounding of the division; 2 (never true) when sigma_t bound is too small for the margin to mean anything.
        const float bound = (2.f * D->sc.env.bsphere_r) * 0.998f - 1e-3f;
        for (int k = 0; k < 3; ++k) {
            const double x = (double) o.sigma_t[k] * (double) bound;
            double v = (x > 0.05 && std::isfinite(x)) ? std::exp(-x * (1.0 - 1e-4)) * (1.0 + 1e-4) : 2.0;
            float vf = (float) v; if ((double) vf < v) vf = std::nextafter(vf, 3.f);
            o.nee_vmin[k] = vf;
        }
    }
    D->h_het.assign(std::max<uint32_t>(d.n_media, 1), DHetMedium{});
    for (uint32_t i = 0; i < d.n_media; ++i) {
        const lrt_medium_desc &M = d.media[i]; DHetMedium &o = D->h_het[i];
        if (M.type != LRT_MEDIUM_HETEROGENEOUS) continue;
        o.data = D->het_data[i];
        for (int k = 0; k < 3; ++k) { o.res[k] = M.grid_res[k]; o.bbox_min[k] = M.grid_bbox_min[k]; o.bbox_max[k] = M.grid_bbox_max[k]; }
        for (int k = 0; k < 12; ++k) o.to_local[k] = M.grid_to_local[k];
        o.scale = M.scale; o.max_density = M.scale * M.grid_max;                    // heterogeneous.cpp:164,171
    }
    HIP_CHECK(hipMemcpyAsync(D->d_het, D->h_het.data(), D->h_het.size() * sizeof(DHetMedium), hipMemcpyHostToDevice, D->stream));
    HIP_CHECK(hipMemcpyAsync(D->d_media, D->h_media.data(), D->h_media.size() * sizeof(DMedium), hipMemcpyHostToDevice, D->stream));
    k_medium_prepare<<<1, 64, 0, D->stream>>>(D->d_media, (uint32_t) std::min<size_t>(D->h_media.size(), 64));
    HIP_CHECK(hipGetLastError());
}

DeviceScene *device_scene_create(const lrt_scene_desc &d, int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        throw std::runtime_error("no HIP device available: the hip_ad_rgb back-end has no CPU fallback");
    if (device < 0 || device >= count) throw std::runtime_error("invalid HIP device ordinal " + std::to_string(device));
    HIP_CHECK(hipSetDevice(device));
    std::unique_ptr<DeviceScene> D(new DeviceScene());
    D->device = device;
    HIP_CHECK(hipStreamCreateWithFlags(&D->stream, hipStreamNonBlocking));
    hipStream_t st = D->stream;
    DScene &sc = D->sc;
    // ---- acceleration structure
    // LDS a workgroup may ask for: the device's opt-in maximum (160 KiB on gfx950), minus the kernel's static __shared__ words
    hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, device)); D->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int optin = 0; if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, device) != hipSuccess || optin <= 0) optin = (int) prop.sharedMemPerBlock;
    (void) hipGetLastError();
    const size_t lds_limit = std::min<size_t>((size_t) std::max(optin, 0), 160 * 1024) - 512;
    // bytes of the LDS image of a BVH: nodes, float4 vertices, 8-byte triangle slots, 16-bit traversal stacks of 1024 threads
    auto lds_image_bytes = [&](const HostBVH &b) {
        const size_t nodes_b = b.nodes.size() / 16 * 64, verts_b = ((size_t) d.n_vertices * 16 + 15) & ~size_t(15), tris_b = (b.tris.size() / 12 * 8 + 15) & ~size_t(15);
        return nodes_b + verts_b + tris_b + (size_t) 2 * (b.max_depth + 2) * 1024;
    };
    // Leaf size: the smallest of 2, 3, 4, 6, 8, 12, 16 triangles whose BVH image fits the LDS next to the traversal stacks (fatter
    // leaves = fewer nodes: more triangle tests per leaf, but every fetch of the traversal stays in LDS; Liver-MultiMesh's two meshes
    // run 44 % faster with 8-triangle leaves in LDS than with 4-triangle leaves in global memory); 4 when nothing fits.
    // LRT_BVH_LEAF=n forces a size.
    HostBVH bvh;
    {
        const int forced = getenv("LRT_BVH_LEAF") ? std::max(1, atoi(getenv("LRT_BVH_LEAF"))) : 0;
        const bool lds_possible = d.n_faces > 0 && d.n_faces <= 32767 && d.n_vertices <= 65535 && !getenv("LRT_NO_LDS_BVH");
        int chosen = forced ? forced : 4;
        bool done = false;
        if (!forced && lds_possible) {                     // the thinnest leaves whose image still fits (measured: 3 beats 4 by 1 % on the liver; 2 does not fit there)
            for (int leaf : { 2, 3, 4, 6, 8, 12, 16 }) {
                HostBVH b2; build_bvh(d.positions, d.faces, d.n_faces, b2, leaf);
                if (lds_image_bytes(b2) <= lds_limit && b2.nodes.size() / 16 <= 32767 && b2.tris.size() / 12 <= 32767) { bvh = std::move(b2); chosen = leaf; done = true; break; }
            }
        }
        if (!done) build_bvh(d.positions, d.faces, d.n_faces, bvh, chosen);
        D->bvh_leaf = chosen;
    }
    sc.nodes = (const float4 *) D->track(dev_upload(bvh.nodes.data(), bvh.nodes.size(), st));
    sc.tris = (const float4 *) D->track(dev_upload(bvh.tris.data(), bvh.tris.size(), st));
    sc.root_is_leaf = bvh.root_is_leaf; sc.root_leaf_first = bvh.root_first; sc.root_leaf_count = bvh.root_count;
    sc.n_faces = d.n_faces; sc.n_emitters = d.n_emitters; sc.one_shape = d.n_shapes == 1;
    // ---- LDS image of the acceleration structure (persistent kernel): used when it fits next to the traversal stacks
    {
        const size_t n_nodes = bvh.nodes.size() / 16, n_slots = bvh.tris.size() / 12, n_verts = d.n_vertices;
        const size_t nodes_b = n_nodes * 64, verts_b = (n_verts * 16 + 15) & ~size_t(15), tris_b = (n_slots * 8 + 15) & ~size_t(15);
        const size_t total = lds_image_bytes(bvh);                   // the per-lane stacks hold max_depth + 2 entries
        if (d.n_faces > 0 && d.n_faces <= 32767 && n_verts <= 65535 && n_nodes <= 32767 && n_slots <= 32767 && total <= lds_limit && !getenv("LRT_NO_LDS_BVH")) {
            std::vector<unsigned char> blob(nodes_b + verts_b + tris_b, 0);
            memcpy(blob.data(), bvh.nodes.data(), nodes_b);
            float *v = reinterpret_cast<float *>(blob.data() + nodes_b);
            for (size_t i = 0; i < n_verts; ++i) { v[4 * i] = d.positions[3 * i]; v[4 * i + 1] = d.positions[3 * i + 1]; v[4 * i + 2] = d.positions[3 * i + 2]; v[4 * i + 3] = 0.f; }
            uint16_t *t = reinterpret_cast<uint16_t *>(blob.data() + nodes_b + verts_b);
            for (size_t sl = 0; sl < n_slots; ++sl) {               // 4th word: face index << 1 | "last slot of its leaf"
                uint32_t f; memcpy(&f, &bvh.tris[12 * sl + 3], 4);
                for (int k = 0; k < 3; ++k) t[4 * sl + k] = (uint16_t) d.faces[3 * (size_t) f + k];
                t[4 * sl + 3] = (uint16_t) (f << 1);
            }
            for (size_t n = 0; n < n_nodes; ++n) {                 // mark the last slot of every leaf (trace_lds)
                int32_t refs[4]; memcpy(refs, &bvh.nodes[16 * n + 12], 16);
                for (int c = 0; c < 2; ++c) if (refs[c] < 0 && refs[2 + c] > 0) t[4 * ((size_t) (uint32_t) ~refs[c] + (size_t) refs[2 + c] - 1) + 3] |= 1;
                // child references as trace_lds's 16-bit work items (inner node index, or 0x8000 | first slot of the leaf)
                uint32_t enc[2]; for (int c = 0; c < 2; ++c) enc[c] = refs[c] < 0 ? (0x8000u | (uint32_t) ~refs[c]) : (uint32_t) refs[c];
                memcpy(blob.data() + 64 * n + 48, enc, 8);
            }
            D->lds.blob = (const uint4 *) D->track(dev_upload(blob.data(), blob.size(), st));
            D->lds.blob_bytes = (uint32_t) blob.size(); D->lds.nodes_off = 0; D->lds.verts_off = (uint32_t) nodes_b; D->lds.tris_off = (uint32_t) (nodes_b + verts_b);
            D->lds.stack_off = (uint32_t) blob.size(); D->lds.total_bytes = (uint32_t) total;
            #define LRT_SMEM(K) HIP_CHECK(hipFuncSetAttribute((const void *) K, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_limit))
#ifdef LRT_DEV_VOLPATH_ONLY                 // developer build (make dev): only ONE render kernel is compiled (default: volpath, independent sampler, LDS BVH;
#ifndef LRT_DEV_INTEGRATOR                  // make dev DEVFLAGS="-DLRT_DEV_INTEGRATOR=LRT_INTEGRATOR_BIOVOLPATH -DLRT_DEV_LD=true" for another)
#define LRT_DEV_INTEGRATOR LRT_INTEGRATOR_VOLPATH
#endif
#ifndef LRT_DEV_LD
#define LRT_DEV_LD false
#endif
#ifndef LRT_DEV_BLOCK
#define LRT_DEV_BLOCK 1024
#endif
            LRT_SMEM((k_render<LRT_DEV_INTEGRATOR, LRT_DEV_BLOCK, true, LRT_DEV_LD>)); LRT_SMEM((k_render<LRT_DEV_INTEGRATOR, LRT_DEV_BLOCK, true, LRT_DEV_LD, true>)); LRT_SMEM((k_trace_lds<true>)); LRT_SMEM((k_trace_lds<false>));
#else
            LRT_SMEM((k_render<LRT_INTEGRATOR_PATH, 1024, true, false>)); LRT_SMEM((k_render<LRT_INTEGRATOR_VOLPATH, 1024, true, false>));
            LRT_SMEM((k_render<LRT_INTEGRATOR_PATH, 1024, true, true>)); LRT_SMEM((k_render<LRT_INTEGRATOR_VOLPATH, 1024, true, true>));
            LRT_SMEM((k_render<LRT_INTEGRATOR_BIOVOLPATH, 1024, true, false>)); LRT_SMEM((k_render<LRT_INTEGRATOR_BIOVOLPATH, 1024, true, true>));
            LRT_SMEM((k_render<LRT_INTEGRATOR_BIOVOLPATH06, 1024, true, false>)); LRT_SMEM((k_render<LRT_INTEGRATOR_BIOVOLPATH06, 1024, true, true>));
            LRT_SMEM((k_render<LRT_INTEGRATOR_PATH, 1024, true, false, true>)); LRT_SMEM((k_render<LRT_INTEGRATOR_VOLPATH, 1024, true, false, true>));       // (compact records)
            LRT_SMEM((k_render<LRT_INTEGRATOR_PATH, 1024, true, true, true>)); LRT_SMEM((k_render<LRT_INTEGRATOR_VOLPATH, 1024, true, true, true>));
            LRT_SMEM((k_render<LRT_INTEGRATOR_BIOVOLPATH, 1024, true, false, true>)); LRT_SMEM((k_render<LRT_INTEGRATOR_BIOVOLPATH, 1024, true, true, true>));
            LRT_SMEM((k_render<LRT_INTEGRATOR_BIOVOLPATH06, 1024, true, false, true>)); LRT_SMEM((k_render<LRT_INTEGRATOR_BIOVOLPATH06, 1024, true, true, true>));
            // (the wide-record integrators run 768-thread workgroups: 3 waves per SIMD, 168 VGPRs instead of 128 + 200 - 430 B of scratch per lane;
            //  measured on the f4 bench configs: volpathmis +58 %, volpath with heterogeneous media +9 %; 512 threads: +54 % / -18 %)
            LRT_SMEM((k_render<LRT_INTEGRATOR_VOLPATH_HET, LRT_WIDE_BLOCK, true, false>)); LRT_SMEM((k_render<LRT_INTEGRATOR_VOLPATH_HET, LRT_WIDE_BLOCK, true, true>));
            LRT_SMEM((k_render<LRT_INTEGRATOR_VOLPATHMIS, LRT_WIDE_BLOCK, true, false>)); LRT_SMEM((k_render<LRT_INTEGRATOR_VOLPATHMIS, LRT_WIDE_BLOCK, true, true>));
            LRT_SMEM((k_render<LRT_INTEGRATOR_VOLPATHMIS_PLAIN, LRT_WIDE_BLOCK, true, false>)); LRT_SMEM((k_render<LRT_INTEGRATOR_VOLPATHMIS_PLAIN, LRT_WIDE_BLOCK, true, true>));
            LRT_SMEM((k_render_prb<false, 1024, true, false>)); LRT_SMEM((k_render_prb<true, 1024, true, false>));
            LRT_SMEM((k_render_prb<false, 1024, true, true>)); LRT_SMEM((k_render_prb<true, 1024, true, true>));
            LRT_SMEM((k_render_prb<false, 1024, true, false, true>)); LRT_SMEM((k_render_prb<true, 1024, true, false, true>));
            LRT_SMEM((k_render_prb<false, 1024, true, true, true>)); LRT_SMEM((k_render_prb<true, 1024, true, true, true>));
            LRT_SMEM((k_trace_lds<true>)); LRT_SMEM((k_trace_lds<false>));
#endif
            #undef LRT_SMEM
            D->use_lds = true;
        }
    }
    // ---- conservative distance field (scenes with participating media: short free-flight segments deep inside a
    // volume are proven surface-free with one lookup instead of a BVH traversal)
    sc.grid = DDistGrid{};
    if (d.n_media > 0 && d.n_faces > 0 && d.n_faces <= (1u << 17) && !getenv("LRT_NO_DIST_GRID")) {
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (uint32_t f = 0; f < 3 * d.n_faces; ++f) for (int a = 0; a < 3; ++a) { float v = d.positions[3 * (size_t) d.faces[f] + a]; lo[a] = std::min(lo[a], v); hi[a] = std::max(hi[a], v); }
        const float ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
        const int res = getenv("LRT_DIST_GRID_RES") ? std::max(8, std::min(256, atoi(getenv("LRT_DIST_GRID_RES")))) : (d.n_faces <= (1u << 14) ? 192 : 64);
        if (ext > 0.f && std::isfinite(ext)) {
            DDistGrid g{};
            g.cell = ext / (float) res; g.inv_cell = 1.f / g.cell;
            for (int a = 0; a < 3; ++a) { g.lo[a] = lo[a]; g.n[a] = std::max(1, std::min(res, (int) std::ceil((hi[a] - lo[a]) / g.cell))); }
            const size_t n_cells = (size_t) g.n[0] * g.n[1] * g.n[2];
            uint16_t *buf = nullptr; HIP_CHECK(hipMalloc((void **) &buf, n_cells * sizeof(uint16_t))); D->track(buf);
            float diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
            float amax = 0.f; for (int a = 0; a < 3; ++a) amax = std::max(amax, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
            const float abs_margin = 1e-4f * diag + 1e-5f * amax;          // >> f32 rounding of positions and of the field itself
            k_build_dist_grid<<<(uint32_t) ((n_cells + LRT_BLOCK - 1) / LRT_BLOCK), LRT_BLOCK, 0, st>>>(sc.tris, (uint32_t) (bvh.tris.size() / 12), g, buf, abs_margin);
            HIP_CHECK(hipGetLastError());
            g.d = buf; g.enabled = 1; sc.grid = g;
        }
    }
    // ---- geometry attributes
    sc.positions = D->track(dev_upload(d.positions, 3 * (size_t) d.n_vertices, st));
    {
        std::vector<float> va(8 * (size_t) d.n_vertices, 0.f);
        for (size_t v = 0; v < d.n_vertices; ++v) {
            if (d.normals) for (int a = 0; a < 3; ++a) va[8 * v + a] = d.normals[3 * v + a];
            if (d.texcoords) for (int a = 0; a < 2; ++a) va[8 * v + 4 + a] = d.texcoords[2 * v + a];
        }
        sc.vattr = (const float4 *) D->track(dev_upload(va.data(), va.size(), st));
    }
    sc.faces = D->track(dev_upload(d.faces, 3 * (size_t) d.n_faces, st));
    sc.face_shape = D->track(dev_upload(d.face_shape, d.n_faces, st));
    std::vector<DShape> shapes(d.n_shapes);
    for (uint32_t i = 0; i < d.n_shapes; ++i) { const lrt_shape_desc &s = d.shapes[i]; shapes[i] = { s.bsdf, s.emitter, s.interior_medium, s.exterior_medium, s.has_normals, s.has_texcoords, s.flip_normals, s.kind }; }
    sc.shapes = D->track(dev_upload(shapes.data(), shapes.size(), st));
    // ---- textures (bitmaps: one luminance float per texel, src/textures/bitmap.cpp:540-552)
    std::vector<DTexture> tex(d.n_textures); std::vector<float> tex_data;
    for (uint32_t i = 0; i < d.n_textures; ++i) {
        const lrt_texture_desc &T = d.textures[i]; DTexture &o = tex[i]; memset(&o, 0, sizeof(o));
        o.type = T.type; o.width = T.width; o.height = T.height; o.channels = T.channels;
        for (int k = 0; k < 3; ++k) { o.color0[k] = T.color0[k]; o.color1[k] = T.color1[k]; }
        for (int k = 0; k < 6; ++k) o.to_uv[k] = T.to_uv[k];
        if (T.type == LRT_TEX_BITMAP) {
            o.data_offset = (uint32_t) tex_data.size();
            size_t np = (size_t) T.width * T.height;
            for (size_t p = 0; p < np; ++p) {
                const float *px = T.data + p * T.channels;
                tex_data.push_back(T.channels == 1 ? px[0] : px[0] * 0.212671f + px[1] * 0.715160f + px[2] * 0.072169f);
            }
        }
    }
    sc.textures = D->track(dev_upload(tex.data(), tex.size(), st));
    sc.tex_data = D->track(dev_upload(tex_data.data(), tex_data.size(), st));
    // ---- BSDFs
    std::vector<DBsdf> bsdfs(d.n_bsdfs); sc.has_null_bsdf = 0;
    auto leaf_flags = [](int type) { return type == LRT_BSDF_DIFFUSE ? F_SMOOTH : type == LRT_BSDF_DIELECTRIC ? F_DELTA : F_NULL; };
    for (uint32_t i = 0; i < d.n_bsdfs; ++i) {
        const lrt_bsdf_desc &B = d.bsdfs[i];
        bsdfs[i] = { B.type, B.reflectance, B.nested, B.texture, B.eta, B.scale, 0, 0 };
        if (B.type == LRT_BSDF_BUMPMAP) {
            if (B.nested < 0 || (uint32_t) B.nested >= d.n_bsdfs || d.bsdfs[B.nested].type == LRT_BSDF_BUMPMAP) throw std::runtime_error("bumpmap: invalid nested BSDF");
            bsdfs[i].flags = leaf_flags(d.bsdfs[B.nested].type);
        } else bsdfs[i].flags = leaf_flags(B.type);
        if (B.type == LRT_BSDF_DIFFUSE && (B.reflectance < 0 || (uint32_t) B.reflectance >= d.n_textures || d.textures[B.reflectance].type == LRT_TEX_BITMAP))
            throw std::runtime_error("unsupported: a bitmap texture as diffuse reflectance (bitmaps are supported as bump-map heights only)");
        if (B.type == LRT_BSDF_NULL) sc.has_null_bsdf = 1;
    }
    sc.bsdfs = D->track(dev_upload(bsdfs.data(), bsdfs.size(), st));
    // ---- bounds (src/render/scene.cpp:49; include/mitsuba/core/bbox.h:343-346; envmap.cpp:337-351)
    DEnv &E = sc.env; memset(&E, 0, sizeof(E)); E.type = -1; E.emitter = -1;
    {
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (uint32_t i = 0; i < d.n_vertices; ++i) for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], d.positions[3 * i + a]); hi[a] = fmaxf(hi[a], d.positions[3 * i + a]); }
        const float ray_eps = 5.9604644775390625e-8f * 1500.f;
        if (d.n_vertices) {
            float c[3], dd[3]; for (int a = 0; a < 3; ++a) { c[a] = (hi[a] + lo[a]) * 0.5f; dd[a] = c[a] - hi[a]; E.bsphere_c[a] = c[a]; }
            float r = sqrtf(fmaf(dd[2], dd[2], fmaf(dd[1], dd[1], dd[0] * dd[0])));
            E.bsphere_r = fmaxf(ray_eps, r * (1.f + ray_eps));
        } else { E.bsphere_c[0] = E.bsphere_c[1] = E.bsphere_c[2] = 0.f; E.bsphere_r = ray_eps; }
    }
    // ---- media
    HIP_CHECK(hipMalloc((void **) &D->d_media, std::max<uint32_t>(d.n_media, 1) * sizeof(DMedium))); D->track(D->d_media);
    HIP_CHECK(hipMalloc((void **) &D->d_bio, std::max<uint32_t>(d.n_media, 1) * sizeof(DBioMedium))); D->track(D->d_bio);
    if (d.n_media > 64) throw std::runtime_error("at most 64 media are supported");
    HIP_CHECK(hipMalloc((void **) &D->d_het, std::max<uint32_t>(d.n_media, 1) * sizeof(DHetMedium))); D->track(D->d_het);
    D->het_data.assign(std::max<uint32_t>(d.n_media, 1), nullptr);
    for (uint32_t i = 0; i < d.n_media; ++i) {
        const lrt_medium_desc &M = d.media[i];
        if (M.type == LRT_MEDIUM_HOMOGENEOUS || M.type == LRT_MEDIUM_HETEROGENEOUS) D->has_non_bio = true;
        if (M.type != LRT_MEDIUM_HETEROGENEOUS) continue;
        if (!M.grid_data || M.grid_res[0] < 1 || M.grid_res[1] < 1 || M.grid_res[2] < 1 || !(M.grid_max > 0.f)) throw std::runtime_error("heterogeneous medium without a valid grid");
        D->het_data[i] = D->track(dev_upload(M.grid_data, (size_t) M.grid_res[0] * M.grid_res[1] * M.grid_res[2], st));
        D->has_het = true;
    }
    for (uint32_t i = 0; i < d.n_shapes; ++i) for (int m : { d.shapes[i].interior_medium, d.shapes[i].exterior_medium })
        if (m >= 0 && d.media[m].type == LRT_MEDIUM_HETEROGENEOUS) D->prb_null = true;
    upload_media(D.get(), d); sc.media = D->d_media; sc.bio = D->d_bio; sc.het = D->d_het;
    // ---- emitters
    std::vector<DEmitter> em(d.n_emitters); std::vector<float> env_rgbx, hier; bool env_interior_positive = false;
    for (uint32_t i = 0; i < d.n_emitters; ++i) {
        const lrt_emitter_desc &S = d.emitters[i]; DEmitter &o = em[i]; memset(&o, 0, sizeof(o));
        o.type = S.type; o.shape = S.shape; o.scale = S.scale; for (int k = 0; k < 3; ++k) o.radiance[k] = S.radiance[k];
        if (S.type == LRT_EMITTER_AREA) {
            D->has_area_emitter = true;
            const lrt_shape_desc &sd = d.shapes[S.shape];
            if (sd.kind != LRT_SHAPE_RECTANGLE) throw std::runtime_error("area emitters are supported on rectangle shapes only");
            memcpy(o.to_world, sd.to_world, sizeof(float) * 12);
            auto xv = [&](float x, float y, float z, float *r) { const float *m = sd.to_world; for (int a = 0; a < 3; ++a) r[a] = fmaf(m[4 * a + 2], z, fmaf(m[4 * a + 1], y, m[4 * a] * x)); };
            float du[3], dv[3]; xv(2.f, 0.f, 0.f, du); xv(0.f, 2.f, 0.f, dv);
            float cx = fmaf(du[1], dv[2], -(du[2] * dv[1])), cy = fmaf(du[2], dv[0], -(du[0] * dv[2])), cz = fmaf(du[0], dv[1], -(du[1] * dv[0]));
            o.inv_area = 1.f / sqrtf(fmaf(cz, cz, fmaf(cy, cy, cx * cx)));
            uint32_t v0 = d.faces[3 * sd.first_face];
            for (int a = 0; a < 3; ++a) o.n[a] = sd.flip_normals ? -d.normals[3 * v0 + a] : d.normals[3 * v0 + a];
        } else {
            E.type = S.type; E.emitter = (int) i; E.scale = S.scale; for (int k = 0; k < 3; ++k) E.radiance[k] = S.radiance[k];
            if (S.type == LRT_EMITTER_ENVMAP) {             // src/emitters/envmap.cpp:139-236
                uint32_t w = (uint32_t) S.width, h = (uint32_t) S.height; E.w = w + 1; E.h = h;
                env_rgbx.assign((size_t) E.w * h * 4, 0.f); std::vector<float> lum((size_t) E.w * h);
                float theta_scale = 1.f / (float) (h - 1) * 3.14159265358979323846f;
                const float *in = S.data;
                for (uint32_t y = 0; y < h; ++y) {
                    float sin_theta = (float) sin((double) ((float) y * theta_scale));
                    for (uint32_t x = 0; x < w; ++x) {
                        float l = fmaxf((in[0] * 0.212671f + in[1] * 0.715160f + in[2] * 0.072169f) - 0.f, 0.f);
                        lum[(size_t) y * E.w + x] = l * sin_theta;
                        float *o4 = &env_rgbx[((size_t) y * E.w + x) * 4]; o4[0] = in[0]; o4[1] = in[1]; o4[2] = in[2];
                        in += 3;
                    }
                    lum[(size_t) y * E.w + w] = lum[(size_t) y * E.w];
                    for (int k = 0; k < 3; ++k) env_rgbx[((size_t) y * E.w + w) * 4 + k] = env_rgbx[((size_t) y * E.w) * 4 + k];
                }
                build_hierarchy(lum, E.w, h, E, hier);
                env_interior_positive = true;          // rows 1..h-2 of the sampling density strictly positive?
                for (uint32_t y = 1; y + 1 < h && env_interior_positive; ++y)
                    for (uint32_t x = 0; x < E.w; ++x) if (!(lum[(size_t) y * E.w + x] > 0.f)) { env_interior_positive = false; break; }
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) E.to_world[3 * a + b] = S.to_world[4 * a + b];
                inverse3(S.to_world, E.to_local);
            }
        }
    }
    sc.nee_fast_reject = (d.n_emitters == 1 && !sc.has_null_bsdf && !D->has_het && !getenv("LRT_NO_NEE_REJECT") &&
                          (E.type == LRT_EMITTER_CONSTANT || (E.type == LRT_EMITTER_ENVMAP && env_interior_positive))) ? 1 : 0;
    sc.emitters = D->track(dev_upload(em.data(), em.size(), st));
    sc.env_data = (const float4 *) D->track(dev_upload(env_rgbx.data(), env_rgbx.size(), st));
    sc.env_hier = D->track(dev_upload(hier.data(), hier.size(), st));
    build_camera(d, sc.cam, sc.film);
    HIP_CHECK(hipMalloc((void **) &D->d_sc, sizeof(DScene))); D->track(D->d_sc);
    HIP_CHECK(hipMemcpyAsync(D->d_sc, &sc, sizeof(DScene), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMalloc((void **) &D->d_launch, LRT_LAUNCH_SLOTS * sizeof(DLaunch))); D->track(D->d_launch);
    HIP_CHECK(hipHostMalloc((void **) &D->h_launch, LRT_LAUNCH_SLOTS * sizeof(DLaunch)));
    HIP_CHECK(hipMalloc((void **) &D->counters, sizeof(DCounters))); D->track(D->counters);
    HIP_CHECK(hipHostMalloc((void **) &D->h_counters, sizeof(DCounters)));
    HIP_CHECK(hipStreamSynchronize(st));
    return D.release();
}

void device_scene_update_params(DeviceScene *D, const lrt_scene_desc &d) {
    HIP_CHECK(hipSetDevice(D->device));
    upload_media(D, d);
    HIP_CHECK(hipStreamSynchronize(D->stream));
}

static void ensure_workspace(DeviceScene *D, uint32_t capacity) {
    if (D->capacity >= capacity && D->mis_alloc == D->need_mis) return;
    HIP_CHECK(hipStreamSynchronize(D->stream));
    // Exception safety: the old streams are released and every pointer cleared BEFORE anything is allocated, and capacity / mis_alloc are
    // set only after every allocation succeeded: a failed hipMalloc (out of memory) leaves capacity = 0, so the next render allocates again
    // instead of launching on freed memory.
    const bool want_mis = D->need_mis;
    D->capacity = 0;
    for (DPathStreams *q : { &D->q[0], &D->q[1] }) {
        for (void *p : { (void *) q->o_maxt, (void *) q->d_eta, (void *) q->tp_pdf, (void *) q->res_flags, (void *) q->lp_lane, (void *) q->rng, (void *) q->tdepth,
                         (void *) q->hit, (void *) q->w1, (void *) q->w2, (void *) q->w3, (void *) q->w4 }) D->release(p);
        *q = DPathStreams{};
    }
    auto alloc = [&](size_t bytes) { void *p = nullptr; HIP_CHECK(hipMalloc(&p, bytes)); D->track(p); return p; };
    for (DPathStreams *q : { &D->q[0], &D->q[1] }) {
        q->o_maxt = (float4 *) alloc((size_t) capacity * 16); q->d_eta = (float4 *) alloc((size_t) capacity * 16); q->tp_pdf = (float4 *) alloc((size_t) capacity * 16);
        q->res_flags = (float4 *) alloc((size_t) capacity * 16); q->lp_lane = (float4 *) alloc((size_t) capacity * 16);
        q->rng = (uint2 *) alloc((size_t) capacity * 16); q->tdepth = (float2 *) alloc((size_t) capacity * 8);     // rng: 16 B per entry (compact records keep state | lane | sampler word there)
        if (D->has_het || want_mis) q->hit = (float4 *) alloc((size_t) capacity * 16);
        if (want_mis) for (float4 **w : { &q->w1, &q->w2, &q->w3, &q->w4 }) *w = (float4 *) alloc((size_t) capacity * 16);
    }
    D->capacity = capacity; D->mis_alloc = want_mis;
}

static hipEvent_t get_event(DeviceScene *D, size_t i) {
    while (D->ev_pool.size() <= i) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); D->ev_pool.push_back(e); }
    return D->ev_pool[i];
}

// spp: samples of ONE pass; spp_total = n_passes * spp (integrator.cpp:176-184,275-293)
struct ResolvedOpts { int integrator, max_depth, rr_depth, hide_emitters; uint32_t spp, seed, tile_rank, tile_count; uint32_t spp_total = 0, n_passes = 1, pass = 0; };
static ResolvedOpts resolve(const lrt_scene_desc &d, const lrt_render_opts *o) {
    ResolvedOpts r;
    if (o && o->integrator > LRT_INTEGRATOR_VOLPATHMIS) throw std::invalid_argument("lrt_render_opts.integrator " + std::to_string(o->integrator) + " is not an integrator (LRT_INTEGRATOR_PATH .. LRT_INTEGRATOR_VOLPATHMIS, or -1 for the scene's own)");
    r.integrator = (o && o->integrator >= 0) ? o->integrator : d.integrator.type;
    r.max_depth = (o && o->max_depth != -2) ? o->max_depth : d.integrator.max_depth;
    // A path's depth lives in 16 bits of its record, and -1 ("unbounded") has no end at all when Russian roulette cannot stop a path (the ld sampler's 1-D
    // sample takes `sample_count` values per pixel: for a fifth of the pixels none of them reaches 0.95; found by a fuzz scene whose path had left a leaky
    // mesh with its medium flag set).  -1 and anything above mean 65535 here and in the oracle; the reference would go on.
    if (r.max_depth < 0 || r.max_depth > 65535) r.max_depth = 65535;
    r.rr_depth = (o && o->rr_depth >= 0) ? o->rr_depth : d.integrator.rr_depth;
    r.hide_emitters = (o && o->hide_emitters >= 0) ? (o->hide_emitters != 0) : (d.integrator.hide_emitters != 0);
    r.spp = (o && o->spp) ? o->spp : d.sample_count;
    if (d.sampler_type == LRT_SAMPLER_LD) {            // integrator.cpp:169-171 + ldsampler.cpp:83-93: a square power of two
        uint32_t res = 2;
        while (res * res < r.spp) { ++res; uint32_t p2 = 1; while (p2 < res) p2 <<= 1; res = p2; }
        r.spp = res * res;
    }
    r.seed = o ? o->seed : 0;
    r.tile_rank = o ? o->tile_rank : 0; r.tile_count = (o && o->tile_count) ? o->tile_count : 1;
    if (r.tile_rank >= r.tile_count) throw std::runtime_error("tile_rank must be smaller than tile_count");
    if (r.spp == 0) throw std::runtime_error("spp must be positive");
    // passes: the integrator's `samples_per_pass`, then the 2^32 - 1 limit of a wavefront (whole image, not this rank's share)
    r.spp_total = r.spp; r.n_passes = 1; r.pass = 0;
    if (r.integrator != LRT_INTEGRATOR_PRBVOLPATH) {
        uint32_t per = d.samples_per_pass ? std::min(d.samples_per_pass, r.spp) : r.spp;
        if (r.spp % per != 0) throw std::runtime_error("sample_count (" + std::to_string(r.spp) + ") must be a multiple of spp_per_pass (" + std::to_string(per) + ").");
        const uint64_t wavefront = (uint64_t) d.film.crop_width * d.film.crop_height * per, limit = 0xffffffffull;
        if (wavefront > limit) per /= (uint32_t) ((wavefront + limit - 1) / limit);
        if (per == 0) throw std::runtime_error("film too large: a single sample per pixel exceeds 2^32 lanes");
        r.n_passes = r.spp_total / per; r.spp = per;
    }
    return r;
}

// 32x32 pixel tiles in row-major tile order, tile k -> rank k % tile_count
static void ensure_pixel_list(DeviceScene *D, const ResolvedOpts &O) {
    const DFilm &F = D->sc.film;
    if (O.tile_count == 1) { D->n_owned_pixels = (uint32_t) F.width * F.height; return; }
    if (D->pixel_list && D->pixel_list_rank == O.tile_rank && D->pixel_list_count == O.tile_count) return;
    std::vector<uint32_t> px;
    uint32_t tx = (F.width + 31) / 32, ty = (F.height + 31) / 32;
    for (uint32_t t = O.tile_rank; t < tx * ty; t += O.tile_count) {
        uint32_t x0 = (t % tx) * 32, y0 = (t / tx) * 32;
        for (uint32_t y = y0; y < std::min<uint32_t>(y0 + 32, F.height); ++y)
            for (uint32_t x = x0; x < std::min<uint32_t>(x0 + 32, F.width); ++x) px.push_back(y * F.width + x);
    }
    D->release(D->pixel_list); D->release(D->pixel_slot); D->pixel_list = nullptr; D->pixel_slot = nullptr;
    D->pixel_list = D->track(dev_upload(px.data(), px.size(), D->stream));
    std::vector<uint32_t> inv((size_t) F.width * F.height, 0u);
    for (size_t k = 0; k < px.size(); ++k) inv[px[k]] = (uint32_t) k;
    D->pixel_slot = D->track(dev_upload(inv.data(), inv.size(), D->stream));
    D->pixel_list_rank = O.tile_rank; D->pixel_list_count = O.tile_count; D->n_owned_pixels = (uint32_t) px.size();
}

// The rank's tiles dilated by `halo` pixels, as a pixel list (row-major order inside the dilated tiles, every pixel once).  The PRB
// adjoint normalises by the per-pixel sum of reconstruction-filter weights (common.py:730-746): a rank reads that sum at the pixels its
// own lanes' footprints touch (own tiles + fn pixels), and those sums are complete once every lane within fn of them has been added:
// halo = 2 fn.  Without this every rank would walk all W H spp lanes of the image.
static void ensure_halo_list(DeviceScene *D, const ResolvedOpts &O, uint32_t halo) {
    const DFilm &F = D->sc.film;
    if (D->halo_list && D->halo_rank == O.tile_rank && D->halo_count == O.tile_count && D->halo_width == halo) return;
    std::vector<uint8_t> mark((size_t) F.width * F.height, 0);
    const uint32_t tx = (F.width + 31) / 32, ty = (F.height + 31) / 32;
    for (uint32_t t = O.tile_rank; t < tx * ty; t += O.tile_count) {
        const int x0 = (int) (t % tx) * 32 - (int) halo, y0 = (int) (t / tx) * 32 - (int) halo;
        const int x1 = std::min<int>((int) (t % tx) * 32 + 32, F.width) + (int) halo, y1 = std::min<int>((int) (t / tx) * 32 + 32, F.height) + (int) halo;
        for (int y = std::max(y0, 0); y < std::min(y1, (int) F.height); ++y)
            for (int x = std::max(x0, 0); x < std::min(x1, (int) F.width); ++x) mark[(size_t) y * F.width + x] = 1;
    }
    std::vector<uint32_t> px;
    for (size_t k = 0; k < mark.size(); ++k) if (mark[k]) px.push_back((uint32_t) k);
    D->release(D->halo_list); D->halo_list = nullptr;
    D->halo_list = D->track(dev_upload(px.data(), px.size(), D->stream));
    D->halo_rank = O.tile_rank; D->halo_count = O.tile_count; D->halo_width = halo; D->n_halo_pixels = (uint32_t) px.size();
}

static void ensure_prb_workspace(DeviceScene *D, uint32_t records, uint64_t l_buf_lanes) {
    if (D->prb_capacity < records) {
        HIP_CHECK(hipStreamSynchronize(D->stream));
        for (int k = 0; k < 2; ++k) { D->release(D->dl[k]); HIP_CHECK(hipMalloc((void **) &D->dl[k], (size_t) records * 16)); D->track(D->dl[k]); }
        D->prb_capacity = records;
    }
    if (D->l_buf_lanes < l_buf_lanes) { HIP_CHECK(hipStreamSynchronize(D->stream)); D->release(D->L_buf); HIP_CHECK(hipMalloc((void **) &D->L_buf, (size_t) l_buf_lanes * 16)); D->track(D->L_buf); D->l_buf_lanes = l_buf_lanes; }
    if (!D->d_grads) { HIP_CHECK(hipMalloc((void **) &D->d_grads, 7 * sizeof(double))); D->track(D->d_grads); }
}

static DRenderParams make_params(const lrt_scene_desc &d, const ResolvedOpts &O, uint64_t n_lanes) {
    DRenderParams rp{};
    rp.integrator = O.integrator; rp.max_depth = O.max_depth; rp.rr_depth = O.rr_depth; rp.hide_emitters = O.hide_emitters;
    rp.spp = O.spp; rp.log2_spp = ((O.spp & (O.spp - 1)) == 0) ? (uint32_t) __builtin_ctz(O.spp) : 0xffffffffu;
    rp.profile = getenv("LRT_DEBUG_LAUNCH") ? 1u : 0u;                 // bit 0: per-tile-kind timing, results unchanged
#ifdef LRT_EXPERIMENT
    if (getenv("LRT_EXP")) rp.profile |= (uint32_t) atoi(getenv("LRT_EXP"));   // cost-attribution switches (`make exp` build only)
    if (getenv("LRT_PRB_DEBUG_LANE")) rp.pad1 = (uint32_t) atoi(getenv("LRT_PRB_DEBUG_LANE")) + 1u;   // per-trip printf of one lane's PRB passes
#endif
    rp.seed_value = d.sampler_seed + O.seed; rp.base_seed = d.sampler_seed; rp.seed = O.seed;
    rp.ld_count = d.sampler_type == LRT_SAMPLER_LD ? O.spp_total : 0u; rp.pass_index = O.pass; rp.spp_total = O.spp_total; rp.tile_rank = O.tile_rank; rp.tile_count = O.tile_count; rp.n_lanes = n_lanes;
    return rp;
}

struct LaunchLog { std::vector<std::pair<hipEvent_t, hipEvent_t>> launches; std::vector<uint32_t> sizes; unsigned long long n_records = 0; size_t ev = 0; uint64_t n_iter = 0; };

static void finish_stats(DeviceScene *D, LaunchLog &log, hipEvent_t e_begin, hipEvent_t e_end, uint64_t n_lanes, lrt_render_stats &stats) {
    hipStream_t st = D->stream;
    HIP_CHECK(hipEventRecord(e_end, st));
    HIP_CHECK(hipMemcpyAsync(D->h_counters, D->counters, sizeof(DCounters), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipGetLastError());
    stats.n_samples = n_lanes; stats.n_iter = log.n_iter; stats.n_shadow = D->h_counters->n_shadow; stats.n_launches = log.launches.size();
    stats.n_records = log.n_records; if (!log.n_records) for (uint32_t n : log.sizes) stats.n_records += n;
    float ms = 0.f; double ksum = 0.0;
    for (size_t i = 0; i < log.launches.size(); ++i) {
        HIP_CHECK(hipEventElapsedTime(&ms, log.launches[i].first, log.launches[i].second)); ksum += ms;
        if (getenv("LRT_DEBUG_LAUNCH") && i < 40) {
            fprintf(stderr, "[lrt] launch %zu: %.3f ms\n", i, ms);
        }
    }
    stats.kernel_ms = ksum; stats.lds_resident = D->use_lds ? 1 : 0;
    if (getenv("LRT_DEBUG_LAUNCH")) for (int r = 0; r < 4; ++r) {
        fprintf(stderr, "[lrt] tile kind %d (0 proven-free, 1 query, 2 surface, 3 fresh): %llu tiles, %.1f ticks/tile (100 MHz wall clock)\n", r, D->h_counters->prof_tiles[r], D->h_counters->prof_tiles[r] ? (double) D->h_counters->prof_cycles[r] / D->h_counters->prof_tiles[r] : 0.0);
    }
    if (getenv("LRT_DEBUG_LAUNCH") && D->h_counters->prof_wg[1]) {
        const unsigned long long *w = D->h_counters->prof_wg; const double n_wg = D->use_lds ? D->n_cus : 4.0 * D->n_cus, first = (double) ~w[2];
        fprintf(stderr, "[lrt] workgroups (last launch; us): first start 0, last start %.1f, last end %.1f, mean run time %.1f, mean LDS-image copy %.1f, mean barrier wait per wave %.1f\n",
                ((double) w[5] - first) / 100.0, ((double) w[1] - first) / 100.0, (double) w[3] / n_wg / 100.0, (double) w[0] / n_wg / 100.0, (double) w[4] / (n_wg * (D->use_lds ? 16.0 : 4.0)) / 100.0);
    }
    HIP_CHECK(hipEventElapsedTime(&ms, e_begin, e_end)); stats.total_ms = ms;
}

// Geometry of the persistent kernels: one 1024-thread workgroup per CU when the BVH lives in LDS, else four 256-thread
// ones; P = paths in flight per workgroup (multiple of 64), queues of 2P records per workgroup.
struct PoolGeometry { uint32_t n_wg, P, block; size_t smem; };
static PoolGeometry pool_geometry(DeviceScene *D, uint64_t n_lanes) {
    PoolGeometry g;
    g.n_wg = D->use_lds ? (uint32_t) D->n_cus : 4u * (uint32_t) D->n_cus;
    const uint32_t pool_max = getenv("LRT_POOL") ? std::max(64, atoi(getenv("LRT_POOL"))) : (D->use_lds ? 32768u : 8192u);
    g.P = (uint32_t) std::min<uint64_t>(pool_max, std::max<uint64_t>(64, ((n_lanes + g.n_wg - 1) / g.n_wg + 63) / 64 * 64));
    // a launch with few lanes per workgroup (a rank's share of the frame at 8 GPUs: ~0.5 M) spends a visible part of its time
    // filling and draining the pool: pools of 8 K - 16 K paths were measured best for a 1/8 share of C3 (31.5 ms against 32.4 ms with the full pool): the pool follows the launch, about 48 turnovers per workgroup
    if (!getenv("LRT_POOL") && D->use_lds) {
        const uint64_t per_wg = (n_lanes + g.n_wg - 1) / g.n_wg, want = (per_wg / 48 + 63) / 64 * 64;
        if (want < g.P) g.P = (uint32_t) std::max<uint64_t>(want, std::min<uint64_t>(g.P, 8192));
    }
    g.block = D->use_lds ? 1024u : (uint32_t) LRT_BLOCK;
    g.smem = D->use_lds ? D->lds.total_bytes : (size_t) LRT_STACK * LRT_BLOCK * sizeof(int);
    return g;
}

// Copies the launch arguments into the next ring slot (stream-ordered) and returns the device pointer the kernel reads.
static LaunchPtr push_launch(DeviceScene *D, const DLaunch &a) {
    const uint32_t slot = D->launch_next++ % LRT_LAUNCH_SLOTS;
    if (slot == 0 && D->launch_next > 1) HIP_CHECK(hipStreamSynchronize(D->stream));    // the pinned source ring wraps: every earlier copy has been consumed
    D->h_launch[slot] = a;
    HIP_CHECK(hipMemcpyAsync(&D->d_launch[slot], &D->h_launch[slot], sizeof(DLaunch), hipMemcpyHostToDevice, D->stream));
    return (LaunchPtr) &D->d_launch[slot];
}

template <bool ADJOINT>
static void launch_prb(DeviceScene *D, const DRenderParams &rp, const PoolGeometry &g, const uint32_t *pixel_list, uint64_t lane_begin,
                       float4 *L_buf, const float *grad_image, double *grads, float *film, float *sample_out) {
    hipStream_t st = D->stream;
    HIP_CHECK(hipMemsetAsync(&D->counters->next_lane, 0, sizeof(unsigned long long), st));
    DLaunch a{}; a.rp = rp; a.li = D->lds; a.q0 = D->q[0]; a.q1 = D->q[1]; a.dl0 = D->dl[0]; a.dl1 = D->dl[1]; a.P = g.P; a.cnt = D->counters;
    a.pixel_list = pixel_list; a.lane_begin = lane_begin; a.n = rp.n_lanes; a.L_buf = L_buf; a.grad_image = grad_image; a.wfilm = D->wfilm; a.grads = grads;
    a.film = film; a.sample_out = sample_out; a.sample_base = lane_begin;
    const LaunchPtr lp = push_launch(D, a);             // (compact records, as run_wavefront queues them, were measured on C5: 96-byte records, 5 % SLOWER; the PRB kernels keep the wide layout)
#ifdef LRT_DEV_VOLPATH_ONLY
    (void) lp; throw std::runtime_error("developer build: volpath only");
#else
    #define LRT_LAUNCH_PRB(BS, LDSB, LD) do { if (D->prb_null) k_render_prb<ADJOINT, BS, LDSB, LD, true><<<g.n_wg, BS, g.smem, st>>>((ScenePtr) D->d_sc, lp); \
                                              else k_render_prb<ADJOINT, BS, LDSB, LD, false><<<g.n_wg, BS, g.smem, st>>>((ScenePtr) D->d_sc, lp); } while (0)
    if (D->use_lds) { if (rp.ld_count) LRT_LAUNCH_PRB(1024, true, true); else LRT_LAUNCH_PRB(1024, true, false); }
    else { if (rp.ld_count) LRT_LAUNCH_PRB(LRT_BLOCK, false, true); else LRT_LAUNCH_PRB(LRT_BLOCK, false, false); }
    #undef LRT_LAUNCH_PRB
#endif
    HIP_CHECK(hipGetLastError());
}

// Which media an integrator can meet: the bio integrators call the 5-argument Medium::sample_interaction, which the base class
// (homogeneous / heterogeneous media) answers with NotImplementedError (src/render/medium.cpp:83-90).
static void check_integrator_media(DeviceScene *D, int integrator) {
    if ((integrator == LRT_INTEGRATOR_BIOVOLPATH || integrator == LRT_INTEGRATOR_BIOVOLPATH06) && D->has_non_bio)
        throw std::runtime_error("NotImplementedError: sample_interaction (the bio integrators need liver / parenchyma / glissonCapsule media)");
    if (integrator == LRT_INTEGRATOR_VOLPATHMIS) D->need_mis = true;          // its wider path record is allocated on first use
}

// One persistent launch per render (k_render / k_render_prb): per-workgroup path pools, in-kernel regeneration; see
// kernels.h.  sample_out != nullptr: per-lane test hook for lanes [lane_begin, lane_begin + n_lanes).
static void run_wavefront(DeviceScene *D, const lrt_scene_desc &d, const ResolvedOpts &O, uint64_t lane_begin, uint64_t n_lanes,
                          const uint32_t *pixel_list, float *film, float *sample_out, lrt_render_stats &stats) {
    hipStream_t st = D->stream;
    DRenderParams rp = make_params(d, O, n_lanes);
    rp.pixel_slot = pixel_list ? D->pixel_slot : nullptr;
    rp.pass_in = D->cur_pass_in; rp.pass_out = D->cur_pass_out;
    const bool prb = O.integrator == LRT_INTEGRATOR_PRBVOLPATH;
    check_integrator_media(D, O.integrator);
    PoolGeometry g = pool_geometry(D, n_lanes);
    if (prb && D->use_lds) g.block = 1024;
    const uint32_t records = (uint32_t) std::min<uint64_t>((uint64_t) g.n_wg * 2u * g.P, 0xffffffffull);
    ensure_workspace(D, records);
    if (prb) ensure_prb_workspace(D, records, 0);
    HIP_CHECK(hipMemsetAsync(D->counters, 0, sizeof(DCounters), st));
    LaunchLog log;
    hipEvent_t e_begin = get_event(D, log.ev++), e_end = get_event(D, log.ev++);
    HIP_CHECK(hipEventRecord(e_begin, st));
    // path.cpp:103-104 returns before the loop when max_depth == 0: that launch only retires the lanes
    const bool count_iter = !(O.integrator == LRT_INTEGRATOR_PATH && O.max_depth == 0);
    hipEvent_t a = get_event(D, log.ev++), b = get_event(D, log.ev++);
    HIP_CHECK(hipEventRecord(a, st));
    if (prb) launch_prb<false>(D, rp, g, pixel_list, lane_begin, nullptr, nullptr, nullptr, film, sample_out);
    else {
        DLaunch a{}; a.rp = rp; a.li = D->lds; a.q0 = D->q[0]; a.q1 = D->q[1]; a.P = g.P; a.cnt = D->counters; a.pixel_list = pixel_list;
        a.lane_begin = lane_begin; a.n = n_lanes; a.film = film; a.sample_out = sample_out; a.sample_base = lane_begin;
        // Compact records (kernels.h, store_state): only an area emitter's pdf reads the last scatter position, so a scene without one does not queue it.
        // Instances exist for the 1024-thread LDS kernels of path / volpath (homogeneous media) / biovolpath / biovolpath06.  LRT_WIDE_RECORDS: developer switch.
        const bool compact = D->use_lds && !D->has_area_emitter && !getenv("LRT_WIDE_RECORDS") && O.integrator != LRT_INTEGRATOR_VOLPATHMIS && !(O.integrator == LRT_INTEGRATOR_VOLPATH && D->has_het);
        a.rp.compact = compact ? 1u : 0u;
        const LaunchPtr lp = push_launch(D, a);
#ifdef LRT_DEV_VOLPATH_ONLY
        if (!(D->use_lds && (O.integrator == LRT_DEV_INTEGRATOR || (LRT_DEV_INTEGRATOR == LRT_INTEGRATOR_VOLPATH_HET && D->has_het)) && (rp.ld_count != 0) == LRT_DEV_LD)) throw std::runtime_error("developer build: one integrator / sampler / LDS BVH only");
        if (compact) k_render<LRT_DEV_INTEGRATOR, LRT_DEV_BLOCK, true, LRT_DEV_LD, true><<<g.n_wg, LRT_DEV_BLOCK, g.smem, st>>>((ScenePtr) D->d_sc, lp);
        else k_render<LRT_DEV_INTEGRATOR, LRT_DEV_BLOCK, true, LRT_DEV_LD><<<g.n_wg, LRT_DEV_BLOCK, g.smem, st>>>((ScenePtr) D->d_sc, lp);
        #define LRT_LAUNCH_I(BS, LDSB)
        #define LRT_LAUNCH(I, BS, LDSB)
#else
        #define LRT_LAUNCH(I, BS, LDSB) do { if (rp.ld_count) k_render<I, BS, LDSB, true><<<g.n_wg, BS, g.smem, st>>>((ScenePtr) D->d_sc, lp); \
                                             else k_render<I, BS, LDSB, false><<<g.n_wg, BS, g.smem, st>>>((ScenePtr) D->d_sc, lp); } while (0)
        #define LRT_LAUNCH_COMPACT(I) do { if (rp.ld_count) k_render<I, 1024, true, true, true><<<g.n_wg, 1024, g.smem, st>>>((ScenePtr) D->d_sc, lp); \
                                           else k_render<I, 1024, true, false, true><<<g.n_wg, 1024, g.smem, st>>>((ScenePtr) D->d_sc, lp); } while (0)
        #define LRT_LAUNCH_I(BS, LDSB) do { switch (O.integrator) { \
            case LRT_INTEGRATOR_PATH: LRT_LAUNCH(LRT_INTEGRATOR_PATH, BS, LDSB); break; \
            case LRT_INTEGRATOR_BIOVOLPATH: LRT_LAUNCH(LRT_INTEGRATOR_BIOVOLPATH, BS, LDSB); break; \
            case LRT_INTEGRATOR_BIOVOLPATH06: LRT_LAUNCH(LRT_INTEGRATOR_BIOVOLPATH06, BS, LDSB); break; \
            case LRT_INTEGRATOR_VOLPATHMIS: if (d.use_spectral_mis) LRT_LAUNCH(LRT_INTEGRATOR_VOLPATHMIS, (LDSB ? LRT_WIDE_BLOCK : BS), LDSB); else LRT_LAUNCH(LRT_INTEGRATOR_VOLPATHMIS_PLAIN, (LDSB ? LRT_WIDE_BLOCK : BS), LDSB); break; \
            default: if (D->has_het) LRT_LAUNCH(LRT_INTEGRATOR_VOLPATH_HET, (LDSB ? LRT_WIDE_BLOCK : BS), LDSB); else LRT_LAUNCH(LRT_INTEGRATOR_VOLPATH, BS, LDSB); } } while (0)
        if (compact) switch (O.integrator) {
            case LRT_INTEGRATOR_PATH: LRT_LAUNCH_COMPACT(LRT_INTEGRATOR_PATH); break;
            case LRT_INTEGRATOR_BIOVOLPATH: LRT_LAUNCH_COMPACT(LRT_INTEGRATOR_BIOVOLPATH); break;
            case LRT_INTEGRATOR_BIOVOLPATH06: LRT_LAUNCH_COMPACT(LRT_INTEGRATOR_BIOVOLPATH06); break;
            default: LRT_LAUNCH_COMPACT(LRT_INTEGRATOR_VOLPATH); }
        else if (D->use_lds) LRT_LAUNCH_I(1024, true); else LRT_LAUNCH_I(LRT_BLOCK, false);
        #undef LRT_LAUNCH_COMPACT
#endif
        #undef LRT_LAUNCH_I
        #undef LRT_LAUNCH
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipEventRecord(b, st));
    log.launches.emplace_back(a, b); log.sizes.push_back(0);
    HIP_CHECK(hipMemcpyAsync(D->h_counters, D->counters, sizeof(DCounters), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    log.n_iter = count_iter ? D->h_counters->n_iter : 0;
    log.n_records = D->h_counters->n_records;
    finish_stats(D, log, e_begin, e_end, n_lanes, stats);
}

void device_render(DeviceScene *D, const lrt_scene_desc &d, const lrt_render_opts *opts, float *film_raw, float *image, lrt_render_stats &stats) {
    HIP_CHECK(hipSetDevice(D->device));
    ResolvedOpts O = resolve(d, opts);
    const DFilm &F = D->sc.film;
    bool on_device = opts && opts->output_on_device;
    size_t np = (size_t) F.width * F.height, film_floats = np * F.channels, image_floats = np * (F.has_alpha ? 4 : 3);
    ensure_pixel_list(D, O);
    const uint64_t n_lanes = (uint64_t) D->n_owned_pixels * O.spp;            // this rank's lanes of ONE pass
    float *film = nullptr;
    if (on_device && film_raw) film = film_raw;
    else { if (D->film_floats < film_floats) { D->release(D->film); D->film = nullptr; HIP_CHECK(hipMalloc((void **) &D->film, film_floats * 4)); D->track(D->film); D->film_floats = film_floats; } film = D->film; }
    HIP_CHECK(hipMemsetAsync(film, 0, film_floats * 4, D->stream));
    const uint32_t *pixel_list = O.tile_count > 1 ? D->pixel_list : nullptr;
    const bool carry = O.n_passes > 1 && d.sampler_type != LRT_SAMPLER_LD;     // the independent sampler's streams run on from pass to pass
    if (carry && D->pass_state_lanes < n_lanes) {
        HIP_CHECK(hipStreamSynchronize(D->stream));
        for (int k = 0; k < 2; ++k) { D->release(D->pass_state[k]); HIP_CHECK(hipMalloc((void **) &D->pass_state[k], std::max<uint64_t>(n_lanes, 1) * 8)); D->track(D->pass_state[k]); }
        D->pass_state_lanes = n_lanes;
    }
    const bool lane_splat = F.rfilter != LRT_RFILTER_BOX && (O.n_passes > 1 || !getenv("LRT_NO_LANE_SPLAT"));
    lrt_render_stats total{};
    for (uint32_t pass = 0; pass < O.n_passes; ++pass) {                       // integrator.cpp:343-353
        O.pass = pass;
        D->cur_pass_in = carry ? D->pass_state[pass & 1] : nullptr;
        D->cur_pass_out = (carry && pass + 1 < O.n_passes) ? D->pass_state[(pass & 1) ^ 1] : nullptr;
        if (!lane_splat) {
            lrt_render_stats st1{};
            run_wavefront(D, d, O, 0, n_lanes, pixel_list, film, nullptr, st1);
            total.n_samples += st1.n_samples; total.n_iter += st1.n_iter; total.n_shadow += st1.n_shadow; total.n_launches += st1.n_launches;
            total.n_records += st1.n_records; total.kernel_ms += st1.kernel_ms; total.total_ms += st1.total_ms;
            continue;
        }
        // wide reconstruction filters: per-lane radiance first (16 B / lane, chunks of at most 2^28 lanes), then an in-order
        // splat pass that reduces each pixel's samples inside the wave (k_splat_lanes)
        const uint64_t chunk = std::min<uint64_t>(std::max<uint64_t>(n_lanes, 1), 1ull << 28);
        ensure_prb_workspace(D, 0, chunk);
        for (uint64_t base = 0; base < n_lanes; base += chunk) {
            const uint64_t n = std::min<uint64_t>(chunk, n_lanes - base);
            lrt_render_stats st1{};
            run_wavefront(D, d, O, base, n, pixel_list, nullptr, reinterpret_cast<float *>(D->L_buf), st1);
            DRenderParams rp = make_params(d, O, n); rp.pass_in = D->cur_pass_in;
            DLaunch a{}; a.rp = rp; a.L_buf = D->L_buf; a.pixel_list = pixel_list; a.lane_begin = base; a.n = n; a.film = film;
            k_splat_lanes<false><<<(uint32_t) ((n + LRT_BLOCK - 1) / LRT_BLOCK), LRT_BLOCK, 0, D->stream>>>((ScenePtr) D->d_sc, push_launch(D, a));
            HIP_CHECK(hipGetLastError());
            total.n_samples += st1.n_samples; total.n_iter += st1.n_iter; total.n_shadow += st1.n_shadow; total.n_launches += st1.n_launches;
            total.n_records += st1.n_records; total.kernel_ms += st1.kernel_ms; total.total_ms += st1.total_ms;
        }
    }
    D->cur_pass_in = nullptr; D->cur_pass_out = nullptr;
    total.lds_resident = D->use_lds ? 1 : 0;
    stats = total;
    if (image) {
        float *img = image;
        if (!on_device) { if (D->image_floats < image_floats) { D->release(D->image); D->image = nullptr; HIP_CHECK(hipMalloc((void **) &D->image, image_floats * 4)); D->track(D->image); D->image_floats = image_floats; } img = D->image; }
        k_develop<<<(uint32_t) ((np + 255) / 256), 256, 0, D->stream>>>(F, film, img, (uint32_t) np);
        if (!on_device) HIP_CHECK(hipMemcpyAsync(image, img, image_floats * 4, hipMemcpyDeviceToHost, D->stream));
    }
    if (film_raw && !on_device) HIP_CHECK(hipMemcpyAsync(film_raw, film, film_floats * 4, hipMemcpyDeviceToHost, D->stream));
    HIP_CHECK(hipStreamSynchronize(D->stream));
    HIP_CHECK(hipGetLastError());
}

void device_develop(DeviceScene *D, const float *film_raw, float *image, int on_device) {
    HIP_CHECK(hipSetDevice(D->device));
    const DFilm &F = D->sc.film;
    size_t np = (size_t) F.width * F.height, film_floats = np * F.channels, image_floats = np * (F.has_alpha ? 4 : 3);
    const float *film = film_raw; float *img = image;
    if (!on_device) {
        if (D->film_floats < film_floats) { D->release(D->film); D->film = nullptr; HIP_CHECK(hipMalloc((void **) &D->film, film_floats * 4)); D->track(D->film); D->film_floats = film_floats; }
        if (D->image_floats < image_floats) { D->release(D->image); D->image = nullptr; HIP_CHECK(hipMalloc((void **) &D->image, image_floats * 4)); D->track(D->image); D->image_floats = image_floats; }
        HIP_CHECK(hipMemcpyAsync(D->film, film_raw, film_floats * 4, hipMemcpyHostToDevice, D->stream));
        film = D->film; img = D->image;
    }
    k_develop<<<(uint32_t) ((np + 255) / 256), 256, 0, D->stream>>>(F, film, img, (uint32_t) np);
    if (!on_device) HIP_CHECK(hipMemcpyAsync(image, img, image_floats * 4, hipMemcpyDeviceToHost, D->stream));
    HIP_CHECK(hipStreamSynchronize(D->stream));
    HIP_CHECK(hipGetLastError());
}

void device_render_samples(DeviceScene *D, const lrt_scene_desc &d, const lrt_render_opts *opts, uint64_t lane_begin, uint32_t n, float *out, lrt_render_stats &stats) {
    HIP_CHECK(hipSetDevice(D->device));
    ResolvedOpts O = resolve(d, opts);
    if (lane_begin + n > 0x100000000ull) throw std::runtime_error("lane range exceeds 2^32");
    float *d_out = nullptr;
    HIP_CHECK(hipMalloc((void **) &d_out, (size_t) std::max<uint32_t>(n, 1) * 16));
    try {
        HIP_CHECK(hipMemsetAsync(d_out, 0, (size_t) n * 16, D->stream));
        run_wavefront(D, d, O, lane_begin, n, nullptr, nullptr, d_out, stats);
        HIP_CHECK(hipMemcpyAsync(out, d_out, (size_t) n * 16, hipMemcpyDeviceToHost, D->stream));
        HIP_CHECK(hipStreamSynchronize(D->stream));
    } catch (...) { (void) hipFree(d_out); throw; }
    HIP_CHECK(hipFree(d_out));
}

void device_trace(DeviceScene *D, const lrt_rays_soa *rays, const lrt_hits_soa *hits, uint32_t n, int any_hit) {
    HIP_CHECK(hipSetDevice(D->device));
    hipStream_t st = D->stream;
    std::vector<void *> tmp;
    auto up = [&](const float *h) { float *p; HIP_CHECK(hipMalloc((void **) &p, (size_t) std::max<uint32_t>(n, 1) * 4)); tmp.push_back(p); HIP_CHECK(hipMemcpyAsync(p, h, (size_t) n * 4, hipMemcpyHostToDevice, st)); return p; };
    auto mk = [&]() { float *p; HIP_CHECK(hipMalloc((void **) &p, (size_t) std::max<uint32_t>(n, 1) * 4)); tmp.push_back(p); return p; };
    try {
        float *ox = up(rays->ox), *oy = up(rays->oy), *oz = up(rays->oz), *dx = up(rays->dx), *dy = up(rays->dy), *dz = up(rays->dz), *tm = up(rays->tmax);
        float *t = mk(), *u = mk(), *v = mk(); uint32_t *prim = (uint32_t *) mk();
        uint32_t grid = (n + LRT_BLOCK - 1) / LRT_BLOCK;
        if (n) {
            if (D->use_lds) {                     // the render kernels' tracer: BVH image in LDS
                const uint32_t g = std::min<uint32_t>((uint32_t) D->n_cus, (n + 1023) / 1024);
                if (any_hit) k_trace_lds<true><<<g, 1024, D->lds.total_bytes, st>>>((ScenePtr) D->d_sc, D->lds, ox, oy, oz, dx, dy, dz, tm, t, u, v, prim, n);
                else k_trace_lds<false><<<g, 1024, D->lds.total_bytes, st>>>((ScenePtr) D->d_sc, D->lds, ox, oy, oz, dx, dy, dz, tm, t, u, v, prim, n);
            } else if (any_hit) k_trace<true><<<grid, LRT_BLOCK, 0, st>>>((ScenePtr) D->d_sc, ox, oy, oz, dx, dy, dz, tm, t, u, v, prim, n);
            else k_trace<false><<<grid, LRT_BLOCK, 0, st>>>((ScenePtr) D->d_sc, ox, oy, oz, dx, dy, dz, tm, t, u, v, prim, n);
        }
        HIP_CHECK(hipMemcpyAsync(hits->t, t, (size_t) n * 4, hipMemcpyDeviceToHost, st));
        if (!any_hit) {
            if (hits->u) HIP_CHECK(hipMemcpyAsync(hits->u, u, (size_t) n * 4, hipMemcpyDeviceToHost, st));
            if (hits->v) HIP_CHECK(hipMemcpyAsync(hits->v, v, (size_t) n * 4, hipMemcpyDeviceToHost, st));
            if (hits->prim) HIP_CHECK(hipMemcpyAsync(hits->prim, prim, (size_t) n * 4, hipMemcpyDeviceToHost, st));
        }
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipGetLastError());
    } catch (...) { for (void *p : tmp) (void) hipFree(p); throw; }
    for (void *p : tmp) (void) hipFree(p);
}

} // namespace lrt

namespace lrt {

// RBIntegrator.render_backward (common.py:625-783): weight film (non-box filters) -> per chunk: primal pass into L_buf,
// adjoint replay accumulating d(sum(image * grad_image)) / d(sigma_t, albedo, g) into 7 doubles.
void device_render_backward(DeviceScene *D, const lrt_scene_desc &d, const lrt_render_opts *opts, const float *grad_image, lrt_param_grads *out, lrt_render_stats &stats) {
    HIP_CHECK(hipSetDevice(D->device));
    lrt_render_opts oprb = opts ? *opts : lrt_render_opts{ -1, -2, -1, -1, 0, 0, 0, 1, 0, 0, 0, 0 };
    oprb.integrator = LRT_INTEGRATOR_PRBVOLPATH;                               // resolve as the adjoint integrator: RBIntegrator.render_backward has no pass split (common.py prepare())
    ResolvedOpts O = resolve(d, &oprb);
    check_integrator_media(D, LRT_INTEGRATOR_PRBVOLPATH);
    const int grad_medium = opts ? opts->grad_medium : 0;
    if (grad_medium < -1 || grad_medium >= (int) d.n_media) throw std::runtime_error("lrt_render_backward: grad_medium " + std::to_string(grad_medium) + " is not a medium of the scene");
    hipStream_t st = D->stream;
    const DFilm &F = D->sc.film;
    const size_t np = (size_t) F.width * F.height, T = F.has_alpha ? 4 : 3;
    if ((uint64_t) np * O.spp > 0xffffffffull) throw std::runtime_error("more than 2^32 samples per render");
    ensure_pixel_list(D, O);
    const uint64_t n_lanes = (uint64_t) D->n_owned_pixels * O.spp;
    const uint32_t *pixel_list = O.tile_count > 1 ? D->pixel_list : nullptr;
    DRenderParams rp = make_params(d, O, n_lanes);
    rp.grad_medium = grad_medium;
    // primal radiance of every lane of a pass is kept (16 B / lane); passes of at most 2^28 lanes bound that buffer to 4.3 GB
    const uint64_t pass = std::min<uint64_t>(std::max<uint64_t>(n_lanes, 1), 1ull << 28);
    PoolGeometry g = pool_geometry(D, pass);
    if (D->use_lds) g.block = 1024;
    const uint32_t records = (uint32_t) std::min<uint64_t>((uint64_t) g.n_wg * 2u * g.P, 0xffffffffull);
    ensure_workspace(D, records); ensure_prb_workspace(D, records, pass);
    const float *g_img = grad_image;
    if (!(opts && opts->output_on_device)) {
        if (D->grad_floats < np * T) { D->release(D->grad_image); D->grad_image = nullptr; HIP_CHECK(hipMalloc((void **) &D->grad_image, np * T * 4)); D->track(D->grad_image); D->grad_floats = np * T; }
        HIP_CHECK(hipMemcpyAsync(D->grad_image, grad_image, np * T * 4, hipMemcpyHostToDevice, st));
        g_img = D->grad_image;
    }
    if (F.rfilter != LRT_RFILTER_BOX) {
        if (D->wfilm_floats < np) { D->release(D->wfilm); D->wfilm = nullptr; HIP_CHECK(hipMalloc((void **) &D->wfilm, np * 4)); D->track(D->wfilm); D->wfilm_floats = np; }
        HIP_CHECK(hipMemsetAsync(D->wfilm, 0, np * 4, st));
        // sum of reconstruction-filter weights per pixel: over every lane of the image, or (tile-sharded) over the lanes of the rank's
        // tiles dilated by 2 fn pixels, which completes the sums at every pixel the rank's own footprints read (ensure_halo_list)
        const uint32_t *wlist = nullptr; uint64_t all = (uint64_t) np * O.spp;
        if (O.tile_count > 1) { ensure_halo_list(D, O, 2u * (uint32_t) F.fn); wlist = D->halo_list; all = (uint64_t) D->n_halo_pixels * O.spp; }
        DRenderParams rw = make_params(d, O, all);
        for (uint64_t base = 0; base < all; base += (1ull << 30)) {
            const uint64_t n = std::min<uint64_t>(1ull << 30, all - base);
            DLaunch a{}; a.rp = rw; a.lane_begin = base; a.n = n; a.film = D->wfilm; a.pixel_list = wlist;
            k_splat_lanes<true><<<(uint32_t) ((n + LRT_BLOCK - 1) / LRT_BLOCK), LRT_BLOCK, 0, st>>>((ScenePtr) D->d_sc, push_launch(D, a));
        }
    }
    HIP_CHECK(hipMemsetAsync(D->d_grads, 0, 7 * sizeof(double), st));
    HIP_CHECK(hipMemsetAsync(D->counters, 0, sizeof(DCounters), st));
    LaunchLog log;
    hipEvent_t e_begin = get_event(D, log.ev++), e_end = get_event(D, log.ev++);
    HIP_CHECK(hipEventRecord(e_begin, st));
    for (uint64_t base = 0; base < n_lanes; base += pass) {
        rp.n_lanes = std::min<uint64_t>(pass, n_lanes - base);
        for (int adjoint = 0; adjoint < 2; ++adjoint) {
            hipEvent_t a = get_event(D, log.ev++), b = get_event(D, log.ev++);
            HIP_CHECK(hipEventRecord(a, st));
            if (!adjoint) launch_prb<false>(D, rp, g, pixel_list, base, D->L_buf, nullptr, nullptr, nullptr, nullptr);
            else launch_prb<true>(D, rp, g, pixel_list, base, D->L_buf, g_img, D->d_grads, nullptr, nullptr);
            HIP_CHECK(hipEventRecord(b, st));
            log.launches.emplace_back(a, b); log.sizes.push_back(0);
        }
    }
    HIP_CHECK(hipMemcpyAsync(D->h_counters, D->counters, sizeof(DCounters), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    log.n_iter = D->h_counters->n_iter; log.n_records = D->h_counters->n_records;
    finish_stats(D, log, e_begin, e_end, n_lanes, stats);
    double h[7];
    HIP_CHECK(hipMemcpy(h, D->d_grads, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) { out->d_sigma_t[k] = (float) h[k]; out->d_albedo[k] = (float) h[3 + k]; }
    out->d_g = (float) h[6];
}

// ---- learned subsurface model, network stage (include/liverrt.h; one lane per sample, weights through the scalar cache)
void device_vae_scatter(const float *blob, uint32_t n, const float *in_pos, const float *in_dir, const float *poly, const float albedo[3], float g, float ior,
                        const float sigma_t[3], float fit_scale, uint32_t seed, float *out_pos, float *out_absorption, int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        throw std::runtime_error("no HIP device available: the hip_ad_rgb back-end has no CPU fallback");
    if (device < 0 || device >= count) throw std::runtime_error("invalid HIP device ordinal " + std::to_string(device));
    HIP_CHECK(hipSetDevice(device));
    if (n == 0) return;
    // the medium-level features are the same for every sample: evaluated once, on the device (one thread), with the kernels' own log / exp
    struct Tmp { float *p = nullptr; ~Tmp() { if (p) (void) hipFree(p); } };
    Tmp d_blob, d_pos, d_dir, d_poly, d_opos, d_oabs, d_feat;
    auto up = [&](Tmp &t, const float *src, size_t cnt) { HIP_CHECK(hipMalloc((void **) &t.p, cnt * sizeof(float))); if (src) HIP_CHECK(hipMemcpy(t.p, src, cnt * sizeof(float), hipMemcpyHostToDevice)); };
    up(d_blob, blob, LRT_VAE_N_FLOATS); up(d_pos, in_pos, 3 * (size_t) n); up(d_dir, in_dir, 3 * (size_t) n); up(d_poly, poly, 20 * (size_t) n);
    up(d_opos, nullptr, 3 * (size_t) n); up(d_oabs, nullptr, n); up(d_feat, nullptr, 4);
    k_vae_medium_features<<<1, 1>>>(d_blob.p, albedo[0], albedo[1], albedo[2], g, ior, sigma_t[0], sigma_t[1], sigma_t[2], d_feat.p);
    HIP_CHECK(hipGetLastError());
    float feat[4]; HIP_CHECK(hipMemcpy(feat, d_feat.p, sizeof feat, hipMemcpyDeviceToHost));
    DVaeArgs A{}; A.blob = d_blob.p; A.in_pos = d_pos.p; A.in_dir = d_dir.p; A.poly = d_poly.p; A.out_pos = d_opos.p; A.out_absorption = d_oabs.p;
    A.albedo_norm = feat[0]; A.g_norm = feat[1]; A.ior_norm = feat[2]; A.fit_scale = fit_scale; A.n = n; A.seed = seed;
    const size_t smem = (size_t) 2 * 68 * LRT_VAE_BLOCK * sizeof(float);
    HIP_CHECK(hipFuncSetAttribute((const void *) k_vae_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem));
    k_vae_scatter<<<(n + LRT_VAE_BLOCK - 1) / LRT_VAE_BLOCK, LRT_VAE_BLOCK, smem>>>(A);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(out_pos, d_opos.p, 3 * (size_t) n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_absorption, d_oabs.p, (size_t) n * sizeof(float), hipMemcpyDeviceToHost));
}

// ------------------------------------------------------------------ one process, several devices (include/liverrt.h: lrt_render_multi)
// SURVEY.md 8e through the C ABI: device i renders the 32x32 tiles t with t % N == i into its own full-size zeroed film (global lane
// ids), ONE ncclAllReduce (RCCL over xGMI, sum, f32, H*W*C) merges the films, device 0 develops.  One host thread and one stream per
// device.  RCCL is bound at run time from /opt/rocm/lib/librccl.so (LRT_RCCL_LIBRARY overrides): the library that matches the HIP
// runtime libliverrt.so links, whatever copy another framework in the process carries.
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr; ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    void load() {
        if (h) return;
        const char *env = getenv("LRT_RCCL_LIBRARY");
        for (const char *name : { env ? env : "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so" }) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
        if (!h) throw std::runtime_error(std::string("hip: cannot load librccl.so (") + (dlerror() ? dlerror() : "?") + "); lrt_render_multi on distinct devices needs RCCL");
        auto sym = [&](const char *n) { void *p = dlsym(h, n); if (!p) throw std::runtime_error(std::string("hip: librccl.so lacks ") + n); return p; };
        CommInitAll = (decltype(CommInitAll)) sym("ncclCommInitAll"); AllReduce = (decltype(AllReduce)) sym("ncclAllReduce");
        GroupStart = (decltype(GroupStart)) sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd)) sym("ncclGroupEnd");
        CommDestroy = (decltype(CommDestroy)) sym("ncclCommDestroy"); GetErrorString = (decltype(GetErrorString)) sym("ncclGetErrorString");
    }
    void check(ncclResult_t r, const char *what) { if (r != ncclSuccess) throw std::runtime_error(std::string("hip: rccl ") + what + ": " + (GetErrorString ? GetErrorString(r) : "error")); }
};
static Rccl g_rccl;

struct MultiContext {                     // the communicators of one device list (ncclCommInitAll takes about a second: kept with the scene)
    std::vector<int> devices; std::vector<ncclComm_t> comms;
    ~MultiContext() { for (auto c : comms) if (c && g_rccl.CommDestroy) (void) g_rccl.CommDestroy(c); }
};
void multi_context_destroy(MultiContext *m) { delete m; }

__global__ void k_film_add(float *__restrict__ dst, const float *__restrict__ src, size_t n) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

static float *own_film(DeviceScene *D, size_t film_floats) {
    if (D->film_floats < film_floats) { D->release(D->film); D->film = nullptr; HIP_CHECK(hipMalloc((void **) &D->film, film_floats * 4)); D->track(D->film); D->film_floats = film_floats; }
    return D->film;
}

// Sums buf[i] (`count` floats or doubles on device i, stream of device i) over the devices, in place.  Distinct devices: one grouped
// ncclAllReduce.  The same device named several times (a rehearsal on a one-GPU box): the peers share the device, so rank 0's buffer
// takes the others' with a kernel and nothing crosses a link.
template <typename T>
static void reduce_across(std::vector<DeviceScene *> &devs, MultiContext *&ctx, std::vector<T *> &buf, size_t count) {
    const int N = (int) devs.size();
    bool distinct = true;
    for (int i = 0; i < N; ++i) for (int j = 0; j < i; ++j) if (devs[i]->device == devs[j]->device) distinct = false;
    for (auto *D : devs) { HIP_CHECK(hipSetDevice(D->device)); HIP_CHECK(hipStreamSynchronize(D->stream)); }
    if (distinct) {
        g_rccl.load();
        std::vector<int> ids; for (auto *D : devs) ids.push_back(D->device);
        if (!ctx || ctx->devices != ids) {
            delete ctx; ctx = new MultiContext(); ctx->devices = ids; ctx->comms.assign(N, nullptr);
            g_rccl.check(g_rccl.CommInitAll(ctx->comms.data(), N, ids.data()), "ncclCommInitAll");
        }
        g_rccl.check(g_rccl.GroupStart(), "ncclGroupStart");
        for (int i = 0; i < N; ++i) {
            HIP_CHECK(hipSetDevice(devs[i]->device));
            g_rccl.check(g_rccl.AllReduce(buf[i], buf[i], count, sizeof(T) == 8 ? ncclDouble : ncclFloat, ncclSum, ctx->comms[i], devs[i]->stream), "ncclAllReduce");
        }
        g_rccl.check(g_rccl.GroupEnd(), "ncclGroupEnd");
        for (auto *D : devs) { HIP_CHECK(hipSetDevice(D->device)); HIP_CHECK(hipStreamSynchronize(D->stream)); }
    } else {
        for (int i = 1; i < N; ++i) if (devs[i]->device != devs[0]->device) throw std::runtime_error("lrt_render_multi: a device list is either all distinct or one device repeated");
        static_assert(sizeof(T) == 4 || sizeof(T) == 8, "float or double");
        HIP_CHECK(hipSetDevice(devs[0]->device));
        for (int i = 1; i < N; ++i) {
            if (sizeof(T) == 4) k_film_add<<<(uint32_t) ((count + 255) / 256), 256, 0, devs[0]->stream>>>((float *) buf[0], (const float *) buf[i], count);
            else { std::vector<double> a(count), b(count); HIP_CHECK(hipMemcpy(a.data(), buf[0], count * 8, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(b.data(), buf[i], count * 8, hipMemcpyDeviceToHost));
                   for (size_t k = 0; k < count; ++k) a[k] += b[k]; HIP_CHECK(hipMemcpy(buf[0], a.data(), count * 8, hipMemcpyHostToDevice)); }
        }
        HIP_CHECK(hipGetLastError()); HIP_CHECK(hipStreamSynchronize(devs[0]->stream));
    }
}

template <typename F> static void on_every_device(size_t n, F fn) {
    std::vector<std::exception_ptr> err(n);
    std::vector<std::thread> th;
    for (size_t i = 0; i < n; ++i) th.emplace_back([&, i]() { try { fn(i); } catch (...) { err[i] = std::current_exception(); } });
    for (auto &t : th) t.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);
}

void device_render_multi(std::vector<DeviceScene *> &devs, MultiContext *&ctx, const lrt_scene_desc &d, const lrt_render_opts *opts, float *film_raw, float *image, lrt_render_stats &stats) {
    const int N = (int) devs.size();
    const DFilm &F = devs[0]->sc.film;
    const size_t np = (size_t) F.width * F.height, film_floats = np * F.channels, image_floats = np * (F.has_alpha ? 4 : 3);
    const bool on_device = opts && opts->output_on_device;                     // film_raw / image then live on the FIRST device of the list
    std::vector<float *> films(N); std::vector<lrt_render_stats> st(N);
    const auto t0 = std::chrono::steady_clock::now();
    on_every_device(N, [&](size_t i) {
        HIP_CHECK(hipSetDevice(devs[i]->device));
        films[i] = (i == 0 && on_device && film_raw) ? film_raw : own_film(devs[i], film_floats);
        lrt_render_opts o = opts ? *opts : lrt_render_opts{ -1, -2, -1, -1, 0, 0, 0, 1, 0, 0, 0, 0 };
        o.tile_rank = (uint32_t) i; o.tile_count = (uint32_t) N; o.device = devs[i]->device; o.output_on_device = 1;
        device_render(devs[i], d, &o, films[i], nullptr, st[i]);
    });
    if (N > 1 || getenv("LRT_MULTI_ALWAYS_REDUCE")) reduce_across(devs, ctx, films, film_floats);
    DeviceScene *D0 = devs[0];
    HIP_CHECK(hipSetDevice(D0->device));
    if (image) {
        float *img = image;
        if (!on_device) { if (D0->image_floats < image_floats) { D0->release(D0->image); D0->image = nullptr; HIP_CHECK(hipMalloc((void **) &D0->image, image_floats * 4)); D0->track(D0->image); D0->image_floats = image_floats; } img = D0->image; }
        k_develop<<<(uint32_t) ((np + 255) / 256), 256, 0, D0->stream>>>(F, films[0], img, (uint32_t) np);
        if (!on_device) HIP_CHECK(hipMemcpyAsync(image, img, image_floats * 4, hipMemcpyDeviceToHost, D0->stream));
    }
    if (film_raw && !on_device) HIP_CHECK(hipMemcpyAsync(film_raw, films[0], film_floats * 4, hipMemcpyDeviceToHost, D0->stream));
    HIP_CHECK(hipStreamSynchronize(D0->stream)); HIP_CHECK(hipGetLastError());
    lrt_render_stats total{};
    for (auto &x : st) { total.n_samples += x.n_samples; total.n_iter += x.n_iter; total.n_shadow += x.n_shadow; total.n_launches += x.n_launches; total.n_records += x.n_records;
                         total.kernel_ms = std::max(total.kernel_ms, x.kernel_ms); total.lds_resident = x.lds_resident; }
    total.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    stats = total;
}

void device_render_backward_multi(std::vector<DeviceScene *> &devs, MultiContext *&ctx, const lrt_scene_desc &d, const lrt_render_opts *opts, const float *grad_image, lrt_param_grads *out, lrt_render_stats &stats) {
    const int N = (int) devs.size();
    if (opts && opts->output_on_device) throw std::runtime_error("lrt_render_backward_multi: grad_image is a host buffer (every device takes its own copy)");
    std::vector<lrt_param_grads> g(N); std::vector<lrt_render_stats> st(N);
    on_every_device(N, [&](size_t i) {
        lrt_render_opts o = opts ? *opts : lrt_render_opts{ -1, -2, -1, -1, 0, 0, 0, 1, 0, 0, 0, 0 };
        o.tile_rank = (uint32_t) i; o.tile_count = (uint32_t) N; o.device = devs[i]->device; o.output_on_device = 0;
        device_render_backward(devs[i], d, &o, grad_image, &g[i], st[i]);
    });
    // the 7 gradient doubles of every device (still in its d_grads) are reduced like the film: one all-reduce
    std::vector<double *> bufs; for (auto *D : devs) bufs.push_back(D->d_grads);
    if (N > 1) reduce_across(devs, ctx, bufs, 7);
    double h[7]; HIP_CHECK(hipSetDevice(devs[0]->device)); HIP_CHECK(hipMemcpy(h, devs[0]->d_grads, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) { out->d_sigma_t[k] = (float) h[k]; out->d_albedo[k] = (float) h[3 + k]; }
    out->d_g = (float) h[6];
    lrt_render_stats total{};
    for (auto &x : st) { total.n_samples += x.n_samples; total.n_iter += x.n_iter; total.n_shadow += x.n_shadow; total.n_launches += x.n_launches; total.n_records += x.n_records;
                         total.kernel_ms = std::max(total.kernel_ms, x.kernel_ms); total.total_ms = std::max(total.total_ms, x.total_ms); total.lds_resident = x.lds_resident; }
    stats = total;
}

// Test hook (include/liverrt.h lrt_math_eval): the transcendental kernels of csrc/dmath.h evaluated on the device, one lane per value.
// orc_math.h holds the same polynomials written a second time, so bit-equality of the render lanes says nothing about their accuracy:
// tests/test_parity_gpu.py bounds the DEVICE values against float64 and checks them bit for bit against the oracle's twins.
__global__ void k_math_eval(int fn, const float *__restrict__ x, const float *__restrict__ y, uint32_t n, float *__restrict__ out, float *__restrict__ out2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.f, b = 0.f;
    switch (fn) {
        case 0: a = m_log(x[i]); break;
        case 1: a = m_exp(x[i]); break;
        case 2: m_sincos(x[i], &a, &b); break;
        case 3: a = m_atan2(y[i], x[i]); break;
        case 4: a = m_acos(x[i]); break;
        case 5: a = m_log2(x[i]); break;
        case 6: a = x[i] / y[i]; break;
        case 7: a = __builtin_sqrtf(x[i]); break;
        case 8: a = rcp(x[i]); break;
        default: break;
    }
    out[i] = a; out2[i] = b;
}
void device_math_eval(int fn, const float *x, const float *y, uint32_t n, float *out, float *out2, int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) throw std::runtime_error("no HIP device available: the hip_ad_rgb back-end has no CPU fallback");
    if (device < 0 || device >= count) throw std::runtime_error("invalid HIP device ordinal " + std::to_string(device));
    if (fn < 0 || fn > 8) throw std::invalid_argument("lrt_math_eval: function 0 .. 8");
    HIP_CHECK(hipSetDevice(device));
    if (!n) return;
    struct Tmp { float *p = nullptr; ~Tmp() { if (p) (void) hipFree(p); } } dx, dy, d1, d2;
    for (Tmp *t : { &dx, &dy, &d1, &d2 }) HIP_CHECK(hipMalloc((void **) &t->p, (size_t) n * 4));
    HIP_CHECK(hipMemcpy(dx.p, x, (size_t) n * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dy.p, y ? y : x, (size_t) n * 4, hipMemcpyHostToDevice));
    k_math_eval<<<(n + 255) / 256, 256>>>(fn, dx.p, dy.p, n, d1.p, d2.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(out, d1.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    if (out2) HIP_CHECK(hipMemcpy(out2, d2.p, (size_t) n * 4, hipMemcpyDeviceToHost));
}

} // namespace lrt
