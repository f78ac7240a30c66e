// gfx950 kernels of the hip_ad_rgb hot path (included by device.hip).
//
//   k_render          the whole sample loop of one lrt_render in ONE persistent launch: camera-lane generation
//                     (src/render/integrator.cpp:308-338,449-486; src/render/sampler.cpp:97-148;
//                      src/sensors/perspective.cpp:239-279), one trip of the integrator's dr::while_loop per tile visit
//                     (src/integrators/volpath.cpp:170-391 incl. sample_emitter :400-554, src/integrators/path.cpp:194-338),
//                     survivor compaction with __ballot/popcount into per-workgroup queues, film splat
//                     (src/render/imageblock.cpp:174-232,431-500)
//   k_splat_lanes     film accumulation pass for reconstruction filters wider than a pixel
//   k_build_dist_grid conservative distance field (scene upload)
//   k_develop         HDRFilm::develop (src/films/hdrfilm.cpp:306-410)
//   k_trace[_lds]     SoA ray queries (test hook, src/render/scene_native.inl:135-202)
#pragma once
#include "dshade.h"

namespace lrt {

#ifndef LRT_STAMP_KIND
#define LRT_STAMP_KIND 0        // which tile kind the section timers watch: 0 proven-free, 1 query, 2 surface, 3 fresh
#endif
// Developer aid (make dev DEVFLAGS=-DLRT_STAMP [-DLRT_STAMP_KIND=k]): section timers of one tile kind (shader-clock cycles summed over the
// waves' first lanes into DCounters::prof_wg is not used; the sums go to LDS and are printed by thread 0 at the end of the kernel).
struct StampClock {
    unsigned long long *acc; unsigned long long t; bool on;
    DEV void start(unsigned long long *a, bool enable) { acc = a; on = enable; t = on ? __builtin_amdgcn_s_memtime() : 0ull; }
    DEV void at(int k) { if (on) { const unsigned long long now = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63u) == 0) atomicAdd(&acc[k], now - t); t = now; } }
};

struct PathState {
    V3 o, d, tp, res, lp; float maxt, eta, last_pdf; uint32_t flags, lane; uint64_t rng_state;
    uint32_t rng_word;         // compact records: the TEA word behind the lane's sampler stream (independent: v1, ld: v0), kept instead of recomputed every trip
    float tdepth, si_t;        // biovolpath / biovolpath06: `tissueDepth`, distance returned by the previous trip's ray query
    float ff_t;                // volpath (homogeneous media): the free-flight distance the look-ahead already drew for this trip (NaN: none); rides in the maxt slot
    float bio_dist; bool bio_hep;   // biovolpath: the element competition the look-ahead already ran for this trip (distance, NaN = none; hepatocytes won)
    float4 hit;                // volpath, heterogeneous media: the surface hit (t, u, v, prim) a null collision keeps (PF_HAVE_SI)
    float W[2][3][3];          // volpathmis: p_over_f, p_over_f_nee
};

// BIO: a queued path's ray always comes from spawn_ray (maxt = largest float), so its maxt slot carries si_t instead, and
// the seventh stream holds tissueDepth and the element competition the look-ahead already ran (96 B records)
// MODE: 0 path / volpath (88 B), 1 biovolpath* (96 B), 2 volpath with heterogeneous media (104 B), 3 volpathmis (168 B)
// A record as the stream loads return it.  Fetching (the loads) and unpacking (the first use of their results) are apart so that the
// render kernel can ask for a tile's records while the previous tile is still being compacted and stored.
struct RawState { float4 a, b, c, d, e; uint2 r; uint4 r4; float2 td; float4 hit, w1, w2, w3, w4; };
template <int MODE = 0, typename QS>
DEV void fetch_state(const QS &q, size_t i, RawState &w, bool compact = false) {
    w.a = q.o_maxt[i]; w.b = q.d_eta[i]; w.c = q.tp_pdf[i]; w.d = q.res_flags[i];
    if (compact) w.r4 = reinterpret_cast<const uint4 *>(q.rng)[i];                       // state | lane | sampler word: no last-scatter-position stream
    else { w.e = q.lp_lane[i]; w.r = q.rng[i]; }
    if (MODE == 1) w.td = q.tdepth[i];
    if (MODE == 2 || MODE == 3) w.hit = q.hit[i];
    if (MODE == 3) { w.w1 = q.w1[i]; w.w2 = q.w2[i]; w.w3 = q.w3[i]; w.w4 = q.w4[i]; }
}
template <int MODE = 0>
DEV void unpack_state(const RawState &w, PathState &s, bool compact = false) {
    const float4 a = w.a, b = w.b, c = w.c, d = w.d;
    s.o = V3(a.x, a.y, a.z); s.maxt = a.w; s.d = V3(b.x, b.y, b.z); s.eta = b.w;
    s.tp = V3(c.x, c.y, c.z); s.last_pdf = c.w; s.res = V3(d.x, d.y, d.z); s.flags = f2u(d.w);
    if (compact) { const uint4 r4 = w.r4; s.lp = V3(0.f); s.lane = r4.z; s.rng_word = r4.w; s.rng_state = ((uint64_t) r4.y << 32) | r4.x; }
    else { const float4 e = w.e; const uint2 r = w.r; s.lp = V3(e.x, e.y, e.z); s.lane = f2u(e.w); s.rng_state = ((uint64_t) r.y << 32) | r.x; }
    if (MODE == 0) { s.ff_t = a.w; s.maxt = kLargest; }          // a queued ray always comes from spawn_ray: maxt = largest float
    if (MODE == 1) { s.si_t = a.w; s.maxt = kLargest; const float2 td = w.td; s.tdepth = __builtin_fabsf(td.x); s.bio_hep = (f2u(td.x) >> 31) != 0u; s.bio_dist = td.y; }
    if (MODE == 2 || MODE == 3) s.hit = w.hit;
    if (MODE == 3) {
        const float4 w1 = w.w1, w2 = w.w2, w3 = w.w3, w4 = w.w4;
        float *W = &s.W[0][0][0];
        W[0] = c.x; W[1] = c.y; W[2] = c.z; W[3] = c.w; W[4] = w1.x; W[5] = w1.y; W[6] = w1.z; W[7] = w1.w; W[8] = w2.x;
        W[9] = w2.y; W[10] = w2.z; W[11] = w2.w; W[12] = w3.x; W[13] = w3.y; W[14] = w3.z; W[15] = w3.w; W[16] = w4.x; W[17] = w4.y;
    }
}
template <int MODE = 0, typename QS>
DEV void load_state(const QS &q, size_t i, PathState &s, bool compact = false) { RawState w; fetch_state<MODE>(q, i, w, compact); unpack_state<MODE>(w, s, compact); }
template <int MODE = 0, typename QS>
DEV void store_state(const QS &q, size_t i, const PathState &s, bool compact = false) {
    q.o_maxt[i] = make_float4(s.o.x, s.o.y, s.o.z, MODE == 1 ? s.si_t : (MODE == 0 ? s.ff_t : s.maxt));
    q.d_eta[i] = make_float4(s.d.x, s.d.y, s.d.z, s.eta);
    if (MODE == 3) {
        const float *W = &s.W[0][0][0];
        q.tp_pdf[i] = make_float4(W[0], W[1], W[2], W[3]); q.w1[i] = make_float4(W[4], W[5], W[6], W[7]); q.w2[i] = make_float4(W[8], W[9], W[10], W[11]);
        q.w3[i] = make_float4(W[12], W[13], W[14], W[15]); q.w4[i] = make_float4(W[16], W[17], 0.f, 0.f);
    } else q.tp_pdf[i] = make_float4(s.tp.x, s.tp.y, s.tp.z, s.last_pdf);
    q.res_flags[i] = make_float4(s.res.x, s.res.y, s.res.z, u2f(s.flags));
    // Compact records (DRenderParams::compact: the scene has no area emitter, so pdf_emitter_direction never reads the last scatter position): the
    // lane id and the sampler's TEA word ride with the generator state in one 16-byte word and the position stream is not touched: 80 B
    // instead of 88, five memory instructions each way instead of six, and no TEA rounds at the start of a trip.
    if (compact) reinterpret_cast<uint4 *>(q.rng)[i] = make_uint4((uint32_t) s.rng_state, (uint32_t) (s.rng_state >> 32), s.lane, s.rng_word);
    else {
        q.lp_lane[i] = make_float4(s.lp.x, s.lp.y, s.lp.z, u2f(s.lane));
        q.rng[i] = make_uint2((uint32_t) s.rng_state, (uint32_t) (s.rng_state >> 32));
    }
    if (MODE == 1) q.tdepth[i] = make_float2(s.bio_hep ? -s.tdepth : s.tdepth, s.bio_dist);
    if (MODE == 2 || MODE == 3) q.hit[i] = s.hit;
}

// Sampler::seed in the JIT branch of SamplingIntegrator::render (integrator.cpp:308-311): independent: TEA4(base + seed,
// lane) seeds the PCG32 stream (sampler.cpp:129-148); ld: the sequence is the lane's pixel, scramble seed =
// TEA4(base, spp * pixel + seed).first (sampler.cpp:97-107), sample index = lane % spp (:109-117).
template <bool LD>
DEV uint64_t lane_rng_inc(RpRef rp, uint32_t lane) {
    if (LD) {
        const uint32_t pixel = (rp.log2_spp != 0xffffffffu) ? (lane >> rp.log2_spp) : (lane / rp.spp);
        uint32_t v0, v1; tea32(rp.base_seed, rp.spp * pixel + rp.seed, &v0, &v1);
        return (uint64_t) v0 | ((uint64_t) (rp.pass_index * rp.spp + (lane - pixel * rp.spp)) << 32);    // sampler.cpp:69-72,109-117
    }
    uint32_t v0, v1; tea32(rp.seed_value, lane, &v0, &v1);
    return ((uint64_t) v1 << 1) | 1u;
}
template <bool LD>
DEV SamplerT<LD> lane_rng_fresh(RpRef rp, uint32_t lane) {
    SamplerT<LD> r; r.ld_count = rp.ld_count;
    if (LD) { r.state = 0; r.inc = lane_rng_inc<LD>(rp, lane); r.ld_prepare(); return r; }
    uint32_t v0, v1; tea32(rp.seed_value, lane, &v0, &v1);
    r.seed(v0, v1); return r;
}
template <bool LD>
DEV SamplerT<LD> lane_rng_resume(RpRef rp, uint32_t lane, uint64_t state) {
    SamplerT<LD> r; r.ld_count = rp.ld_count; r.state = state; r.inc = lane_rng_inc<LD>(rp, lane); r.ld_prepare(); return r;
}
// the same from the TEA word a compact record carries (lane_rng_word below)
template <bool LD>
DEV SamplerT<LD> lane_rng_resume_word(RpRef rp, uint32_t lane, uint64_t state, uint32_t word) {
    SamplerT<LD> r; r.ld_count = rp.ld_count; r.state = state;
    if (LD) {
        const uint32_t pixel = (rp.log2_spp != 0xffffffffu) ? (lane >> rp.log2_spp) : (lane / rp.spp);
        r.inc = (uint64_t) word | ((uint64_t) (rp.pass_index * rp.spp + (lane - pixel * rp.spp)) << 32);
    } else r.inc = ((uint64_t) word << 1) | 1u;
    r.ld_prepare(); return r;
}
template <bool LD> DEV uint32_t lane_rng_word(const SamplerT<LD> &r) { return LD ? (uint32_t) r.inc : (uint32_t) (r.inc >> 1); }
// Rank-local index of a lane within the current pass (the index space of per-lane buffers)
DEV uint64_t lane_local_index(RpRef rp, uint32_t lane) {
    if (!rp.pixel_slot) return lane;
    const uint32_t pixel = (rp.log2_spp != 0xffffffffu) ? (lane >> rp.log2_spp) : (lane / rp.spp);
    return (uint64_t) rp.pixel_slot[pixel] * rp.spp + (lane - pixel * rp.spp);
}
// The sampler of a lane at the start of the current pass: freshly seeded in pass 0; in later passes the independent sampler
// continues from the state its path of the previous pass left (Sampler::advance() does not reseed), index j = rank-local lane.
template <bool LD>
DEV SamplerT<LD> lane_rng_pass_start(RpRef rp, uint32_t lane, uint64_t j) {
    SamplerT<LD> r = lane_rng_fresh<LD>(rp, lane);
    if (!LD && rp.pass_index > 0) r.state = rp.pass_in[j];
    return r;
}
// the pixel jitter: the sampler's first 2-D sample of the pass (integrator.cpp:465), needed again wherever a film footprint is formed
DEV void lane_jitter(RpRef rp, uint32_t lane, uint64_t j, float &jx, float &jy) {
    if (rp.ld_count) { SamplerT<true> r = lane_rng_pass_start<true>(rp, lane, j); r.next2(jx, jy); }
    else { SamplerT<false> r = lane_rng_pass_start<false>(rp, lane, j); r.next2(jx, jy); }
}

// lane -> pixel (src/render/integrator.cpp:321-338); tile-sharded renders go
// through the rank's pixel list so that seeding uses the GLOBAL lane index.
DEV void lane_to_pixel(SceneRef sc, RpRef rp, uint32_t lane, int *px, int *py) {
    uint32_t idx = (rp.log2_spp != 0xffffffffu) ? (lane >> rp.log2_spp) : (lane / rp.spp);
    uint32_t W = (uint32_t) sc.film.width;
    uint32_t y = idx / W, x = idx - y * W;
    *px = (int) x + sc.film.crop_offset_x; *py = (int) y + sc.film.crop_offset_y;
}

// src/sensors/perspective.cpp:239-279
DEV Ray camera_ray(SceneRef sc, float ax, float ay) {
    const LRT_CONST float *m = sc.cam.s2c;
    V3 p(ax + sc.cam.ppo_x, ay + sc.cam.ppo_y, 0.f);
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fma_(m[4 * i + 2], p.z, fma_(m[4 * i + 1], p.y, fma_(m[4 * i + 0], p.x, m[4 * i + 3])));
    V3 near_p(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
    V3 d = normalize(near_p);
    Ray ray;
    ray.o = V3(sc.cam.to_world[3], sc.cam.to_world[7], sc.cam.to_world[11]);
    ray.d = xform_vec12(sc.cam.to_world, d);
    float inv_z = rcp(d.z), near_t = sc.cam.near_clip * inv_z, far_t = sc.cam.far_clip * inv_z;
    ray.o = ray.o + ray.d * near_t;
    ray.maxt = far_t - near_t;
    return ray;
}

// A fresh camera path for rank-local lane index j (integrator.cpp:321-338,449-470; volpath.cpp:93-140 /
// path.cpp:95-170 up to the loop).
template <bool LD>
DEV PathState generate_camera_path(SceneRef sc, RpRef rp, const uint32_t *__restrict__ pixel_list, uint64_t j) {
    uint32_t lane;
    if (pixel_list) { uint32_t pj = (rp.log2_spp != 0xffffffffu) ? (uint32_t) (j >> rp.log2_spp) : (uint32_t) (j / rp.spp); lane = pixel_list[pj] * rp.spp + (uint32_t) (j - (uint64_t) pj * rp.spp); }
    else lane = (uint32_t) j;
    SamplerT<LD> rng = lane_rng_pass_start<LD>(rp, lane, j);
    int px, py; lane_to_pixel(sc, rp, lane, &px, &py);
    float jx, jy; rng.next2(jx, jy);
    float spx = (float) px + jx, spy = (float) py + jy;
    Ray ray = camera_ray(sc, fma_(spx, sc.film.scale_x, sc.film.offset_x), fma_(spy, sc.film.scale_y, sc.film.offset_y));
    PathState s;
    s.o = ray.o; s.d = ray.d; s.maxt = ray.maxt; s.eta = 1.f; s.tp = V3(1.f); s.res = V3(0.f); s.lp = V3(0.f); s.last_pdf = 1.f; s.lane = lane;
    s.tdepth = 0.f; s.si_t = kInf; s.ff_t = u2f(0x7fc00000u); s.bio_dist = u2f(0x7fc00000u); s.bio_hep = false; s.hit = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < 2; ++a) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) s.W[a][i][j] = 1.f;         // volpathmis.cpp:158-159
                                                     // biovolpath.cpp:125,129: si = zeros (t = inf), tissueDepth = 0
    bool env_visible = !rp.hide_emitters && sc.env.type >= 0;
    uint32_t flags = env_visible ? PF_VALID : 0u;
    if (rp.integrator == LRT_INTEGRATOR_PATH) flags |= PF_SPECULAR;                 // prev_bsdf_delta = true
    else {
        if (!rp.hide_emitters) flags |= PF_SPECULAR;                                 // specular_chain = active && !hide_emitters
        uint32_t channel = min((uint32_t) (rng.next() * 3.f), 2u);                   // volpath.cpp:117-121
        flags |= channel << PF_CHANNEL_SHIFT;
        flags |= (uint32_t) (sc.cam.medium + 1) << PF_MEDIUM_SHIFT;
        if (rp.integrator == LRT_INTEGRATOR_BIOVOLPATH06) flags |= PF_BIO_EMIT | PF_BIO_FULL;     // biovolpath06.cpp:111 type = 127
    }
    s.flags = flags; s.rng_state = rng.state; s.rng_word = lane_rng_word<LD>(rng);
    return s;
}

// ------------------------------------------------------------------ film
template <typename FP> DEV float estrin10(float x, FP c) {
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    float a0 = fma_(x, c[1], c[0]), a1 = fma_(x, c[3], c[2]), a2 = fma_(x, c[5], c[4]), a3 = fma_(x, c[7], c[6]), a4 = fma_(x, c[9], c[8]);
    float b0 = fma_(x2, a1, a0), b1 = fma_(x2, a3, a2);
    float c0 = fma_(x4, b1, b0);
    return fma_(x8, a4, c0);
}
DEV float rfilter_eval(FilmRef F, float x) {
    if (F.rfilter == LRT_RFILTER_GAUSSIAN) return fmax_(estrin10(sqr(x), F.rf_coeff), 0.f);
    if (F.rfilter == LRT_RFILTER_TENT) return fmax_(0.f, 1.f - __builtin_fabsf(x * F.rf_inv_radius));
    return (__builtin_fabsf(x) <= 0.5f) ? 1.f : 0.f;
}

// A finished path: splat {R,G,B,[A],W=1} (integrator.cpp:499-520, imageblock.cpp:174-232,431-500),
// or, for the per-lane test hook, store the radiance.
DEV void finish_path(SceneRef sc, RpRef rp, float *__restrict__ film, float *__restrict__ sample_out,
                     uint64_t sample_base, uint32_t lane, V3 L, bool valid) {
    if (rp.integrator == LRT_INTEGRATOR_PATH && !valid) L = V3(0.f);                 // path.cpp:342-345
    if (sample_out) {                               // per-lane output, indexed by the rank-local lane index
        const uint64_t j = lane_local_index(rp, lane);
        float4 *o = reinterpret_cast<float4 *>(sample_out) + (j - sample_base);
        *o = make_float4(L.x, L.y, L.z, valid ? 1.f : 0.f);
        return;
    }
    FilmRef F = sc.film;
    int px, py; lane_to_pixel(sc, rp, lane, &px, &py);
    const int C = F.channels;
    const float alpha = valid ? 1.f : 0.f;
    auto splat = [&](float *p, float w) {
        atomicAdd(p + 0, L.x * w); atomicAdd(p + 1, L.y * w); atomicAdd(p + 2, L.z * w);
        if (F.has_alpha) { atomicAdd(p + 3, alpha * w); atomicAdd(p + 4, 1.f * w); } else atomicAdd(p + 3, 1.f * w);
    };
    if (F.rfilter == LRT_RFILTER_BOX) {
        int x = px - F.crop_offset_x, y = py - F.crop_offset_y;
        float *p = film + ((size_t) y * F.width + x) * C;
        atomicAdd(p + 0, L.x); atomicAdd(p + 1, L.y); atomicAdd(p + 2, L.z);
        if (F.has_alpha) { atomicAdd(p + 3, alpha); atomicAdd(p + 4, 1.f); } else atomicAdd(p + 3, 1.f);
        return;
    }
    float jx, jy; lane_jitter(rp, lane, lane_local_index(rp, lane), jx, jy);
    float spx = (float) px + jx, spy = (float) py + jy;
    int n = F.fn, count = F.fcount;
    int pix = (int) __builtin_floorf(spx) - n, piy = (int) __builtin_floorf(spy) - n;
    float relx = (float) pix + .5f - spx, rely = (float) piy + .5f - spy;
    for (int ys = 0; ys < count; ++ys) {
        int y = piy - F.crop_offset_y + ys;
        float wy = rfilter_eval(F, rely + (float) ys);
        if (y < 0 || y >= F.height || (wy == 0.f && finite3(L))) continue;
        for (int xs = 0; xs < count; ++xs) {
            int x = pix - F.crop_offset_x + xs;
            if (x < 0 || x >= F.width) continue;
            float w = wy * rfilter_eval(F, relx + (float) xs);
            if (w == 0.f && finite3(L)) continue;                  // a zero weight adds nothing unless the radiance is non-finite (value * 0 = NaN, as imageblock.cpp computes it)
            splat(film + ((size_t) y * F.width + x) * C, w);
        }
    }
}

// Film accumulation called by EVERY lane of a wave (`finishing` selects the lanes that retire a path).  Box filter:
// lanes that splat into the same pixel are summed inside the wave first (the wavefront keeps a pixel's samples in
// neighbouring lanes, so a wave usually holds one or two distinct pixels) and one lane issues the atomics.
DEV void finish_paths_wave(SceneRef sc, RpRef rp, float *__restrict__ film, float *__restrict__ sample_out,
                           uint64_t sample_base, bool finishing, uint32_t lane, V3 L, bool valid) {
    FilmRef F = sc.film;
    if (sample_out || F.rfilter != LRT_RFILTER_BOX) {
        if (finishing) finish_path(sc, rp, film, sample_out, sample_base, lane, L, valid);
        return;
    }
    if (rp.integrator == LRT_INTEGRATOR_PATH && !valid) L = V3(0.f);
    uint32_t pixel = 0xffffffffu;
    if (finishing) { int px, py; lane_to_pixel(sc, rp, lane, &px, &py); pixel = (uint32_t) (py - F.crop_offset_y) * (uint32_t) F.width + (uint32_t) (px - F.crop_offset_x); }
    if (__ballot(finishing) == 0ull) return;
#ifdef LRT_FILM_LEADER_LOOP
    unsigned long long todo = __ballot(finishing);
    const uint32_t me = threadIdx.x & 63u;
    while (todo) {
        int leader = __ffsll((long long) todo) - 1;
        uint32_t key = __shfl(pixel, leader);
        bool mine = finishing && pixel == key;
        unsigned long long grp = __ballot(mine);
        float r = mine ? L.x : 0.f, g = mine ? L.y : 0.f, b = mine ? L.z : 0.f;
        r = wave_sum(r); g = wave_sum(g); b = wave_sum(b);
        const float w = (float) __popcll(grp), al = F.has_alpha ? (float) __popcll(__ballot(mine && valid)) : 0.f;
        if ((int) me == leader) {
            float *p = film + (size_t) key * F.channels;
            atomicAdd(p + 0, r); atomicAdd(p + 1, g); atomicAdd(p + 2, b);
            if (F.has_alpha) { atomicAdd(p + 3, al); atomicAdd(p + 4, w); } else atomicAdd(p + 3, w);
        }
        todo &= ~grp;
    }
#else
    // Queues keep lanes in lane order, so the finishing lanes of one pixel sit in RUNS of consecutive lanes: one segmented sum over the
    // wave adds up every run at once (no loop over the pixels of the tile, no LDS round trip) and the last lane of each run issues the
    // atomics.  A pixel that appears in two runs simply gets two sets of atomics.  Lanes that do not finish are runs of their own (zeros).
    const uint32_t prev = wave_prev(pixel, 0xfffffffeu), next = wave_next(pixel, 0xfffffffeu);
    float v[5] = { finishing ? L.x : 0.f, finishing ? L.y : 0.f, finishing ? L.z : 0.f, finishing ? 1.f : 0.f, (finishing && valid) ? 1.f : 0.f };
    wave_segmented_sums(v, pixel != prev || !finishing);
    if (finishing && pixel != next) {
        float *p = film + (size_t) pixel * F.channels;
        atomicAdd(p + 0, v[0]); atomicAdd(p + 1, v[1]); atomicAdd(p + 2, v[2]);
        if (F.has_alpha) { atomicAdd(p + 3, v[4]); atomicAdd(p + 4, v[3]); } else atomicAdd(p + 3, v[3]);
    }
#endif
}

// ---------------------------------------------------------- volpath NEE
// src/integrators/volpath.cpp:400-554.  ref_n is zero for medium interactions.
template <bool HET, typename SMP, typename TR>
DEV V3 volpath_sample_emitter(SceneRef sc, SMP &rng, V3 ref_p, V3 ref_n, bool ref_is_surface, uint32_t ref_shape, V3 ref_geo_n,
                              int medium, uint32_t channel, DirSample *ds_out, const TR &tr, uint32_t &n_shadow) {
    V3 transmittance(1.f);
    float sx, sy; rng.next2(sx, sy);
    DirSample ds; V3 emitter_val = sample_emitter_direction(sc, ref_p, sx, sy, &ds);
    *ds_out = ds;
    if (ds.pdf == 0.f) return V3(0.f);
    Ray ray = spawn_ray_to(ref_p, ref_n, ds.p);
    float max_dist = ray.maxt;
    if (ref_is_surface) { const DShape sd = tab(sc.shapes, ref_shape, sc.one_shape); if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, ref_geo_n); }
    float total_dist = 0.f;
    SI si; si.valid = false; si.t = 0.f; si.shape = 0; si.p = V3(0.f); si.n = V3(0.f);
    bool needs_intersection = true, active = true;
    while (active) {
        float remaining_dist = max_dist - total_dist;
        ray.maxt = remaining_dist;
        if (!(remaining_dist > 0.f)) { rng.skip(1); break; }           // the body still runs (masked) in this last trip
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (!active_medium) rng.skip(1);                                // volpath.cpp:479
        if (active_medium) {
            const DMedium M = tab(sc.media, medium);
            const bool het = HET && M.het;
            MI mei = het ? het_sample_interaction(M, tab(sc.het, medium), ray, rng.next()) : medium_sample_interaction(M, ray, rng.next(), channel);
            if (mei.valid() && !het) ray.maxt = fmin_(mei.t, remaining_dist);    // medium->is_homogeneous() only (volpath.cpp:481)
            if (needs_intersection) {
                // Exact shortcut: a real collision (sigma_n = 0) kills the sample whether or not a
                // surface lies in front of it when every surface blocks (no null BSDF): skip the query.
                bool elide = mei.valid() && !sc.has_null_bsdf && !het;
                if (!elide) { n_shadow++; Hit h = tr.closest(ray); si = tr.surface(sc, ray, h); }
                else { si.valid = false; si.t = kInf; }
            }
            if (si.t < mei.t) mei.t = kInf;
            needs_intersection = false;
            bool spectral = M.has_spectral_extinction;
            if (spectral) {
                float t = fmin_(remaining_dist, fmin_(mei.t, si.t)) - mei.mint;
                V3 tr = exp_neg(t, mei.combined);
                V3 ffp = (si.t < mei.t || mei.t > remaining_dist) ? tr : tr * mei.combined;
                float tr_pdf = idx3(ffp, channel);
                transmittance = transmittance * ((tr_pdf > 0.f) ? div_uniform(tr, tr_pdf) : V3(0.f));
            }
            if ((mei.t > remaining_dist) && mei.valid()) total_dist = ds.dist;
            if (mei.t > remaining_dist) mei.t = kInf;
            escaped_medium = !mei.valid();
            active_medium = mei.valid();
            if (active_medium) {
                total_dist += mei.t;
                ray.o = mei.p;
                si.t = si.t - mei.t;
                if (spectral) transmittance = transmittance * mei.sigma_n;
                else transmittance = transmittance * (mei.sigma_n / mei.combined);
            }
        }
        bool intersect = active_surface && needs_intersection;
        if (intersect) { n_shadow++; Hit h = tr.closest(ray); si = tr.surface(sc, ray, h); needs_intersection = false; }
        active_surface = active_surface || escaped_medium;
        if (active_surface) total_dist += si.t;
        active_surface = active_surface && si.valid && !active_medium;
        if (active_surface) {
            transmittance = transmittance * bsdf_null_transmission(sc, tab(sc.shapes, si.shape, sc.one_shape).bsdf);
            ray = spawn_ray(si.p, si.n, ray.d);
        }
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        active = (active_medium || active_surface) && any_nonzero(transmittance);
        if (active_surface) { const DShape sd = tab(sc.shapes, si.shape, sc.one_shape); if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, si.n); }
    }
    return transmittance * emitter_val;
}

// One trip of volpath's while_loop (src/integrators/volpath.cpp:170-391).
// Returns true when the path survives.
// HET: the scene holds heterogeneous media (delta tracking, volpath.cpp:238-259): a null collision moves the ray origin
// and keeps the surface interaction found earlier (`needs_intersection` stays false), which the record carries as a hit.
// PRE (every kernel but the heterogeneous-media one): the first stage of a trip - the termination test (volpath.cpp:190-203) and, inside
// a medium, the free-flight draw (:220, medium.cpp:40-82) - runs one trip early, at the end of the previous trip (`fresh`: at the start
// of a camera lane's first trip), on the lane's own generator: same draws in the same order.  A queued record therefore holds a path
// that is known to run its next trip, with the throughput already divided by the survival probability, the free-flight distance
// in the record (ff_t) and, when the distance field proves that distance free of surfaces, PF_NOHIT.
template <bool HET, typename SMP, typename TR>
DEV bool volpath_iteration(SceneRef sc, RpRef rp, PathState &s, SMP &rng, const TR &tr, uint32_t &n_shadow, uint32_t &n_extra, bool fresh = false, StampClock *clk = nullptr) {
    constexpr bool PRE = !HET;
    uint32_t depth = s.flags & PF_DEPTH_MASK;
    bool proven_empty = (s.flags & PF_NOHIT) != 0;
    float ff_t = s.ff_t;
    const bool needs_intersection = !(HET && (s.flags & PF_HAVE_SI));
    bool act_null_scatter = false;
    Hit hkeep; hkeep.t = s.hit.x; hkeep.u = s.hit.y; hkeep.v = s.hit.z; hkeep.prim = f2u(s.hit.w);
    int medium = (int) ((s.flags & PF_MEDIUM_MASK) >> PF_MEDIUM_SHIFT) - 1;
    const uint32_t channel = (s.flags >> PF_CHANNEL_SHIFT) & 3u;
    bool specular_chain = (s.flags & PF_SPECULAR) != 0, valid_ray = (s.flags & PF_VALID) != 0;
    const uint32_t max_depth = (uint32_t) rp.max_depth;
    V3 throughput = s.tp, result = s.res;
    float eta = s.eta;
    Ray ray; ray.o = s.o; ray.d = s.d; ray.maxt = s.maxt;
    auto commit = [&]() {
        s.tp = throughput; s.res = result; s.eta = eta; s.o = ray.o; s.d = ray.d; s.maxt = ray.maxt;
        s.flags = (depth & PF_DEPTH_MASK) | ((uint32_t) (medium + 1) << PF_MEDIUM_SHIFT) | (channel << PF_CHANNEL_SHIFT) |
                  (specular_chain ? PF_SPECULAR : 0u) | (valid_ray ? PF_VALID : 0u);
    };
    // ---- termination (volpath.cpp:190-203) of the trip about to run (PRE: called for the NEXT trip, see above)
    auto termination_stage = [&]() -> bool {
        bool a = any_nonzero(throughput);
        const float q = fmin_(max3(throughput) * sqr(eta), .95f);
        const bool perform_rr = depth > (uint32_t) rp.rr_depth;
        if (a) { const float u = rng.next(); a = (u < q) || !perform_rr; }
        if (perform_rr) throughput = throughput * rcp(q);
        return a && depth < max_depth;
    };
    // PRE, inside a medium: the free-flight draw of the trip about to run, and the attempt to prove that distance free of surfaces
    uint32_t nohit = 0;
    float cache_t = u2f(0x7fc00000u);
    auto free_flight_stage = [&]() {
        if (medium < 0) return;
        const DMedium M = tab(sc.media, medium);
        cache_t = medium_sampled_t(M, rng.next(), channel);
        if (sc.grid.enabled) { const MI m2 = medium_interaction_at(M, ray, cache_t); if (m2.valid() && segment_free_of_surfaces(sc.grid, ray.o, ray.d, m2.t)) nohit = PF_NOHIT; }
    };
    bool active = true;
    if (!PRE) active = termination_stage();
    else if (fresh) { active = termination_stage(); if (active) { free_flight_stage(); ff_t = cache_t; proven_empty = nohit != 0; nohit = 0; cache_t = u2f(0x7fc00000u); } }
    if (!active) { commit(); return false; }

    bool active_medium = medium >= 0, active_surface = !active_medium;
    bool act_medium_scatter = false, escaped_medium = false;
    MI mei; mei.t = kInf;
    SI si; si.valid = false; si.t = kInf;
    if (!active_medium) rng.skip(2);                                    // volpath.cpp:220,239
    if (active_medium) {
        const DMedium M = tab(sc.media, medium);
        const bool het = HET && M.het;
        if (PRE) mei = medium_interaction_at(M, ray, ff_t);          // the free-flight stage drew this distance (same sample, channel, medium)
        else { const float sample = rng.next(); mei = het ? het_sample_interaction(M, tab(sc.het, medium), ray, sample) : medium_sample_interaction(M, ray, sample, channel); }
        if (mei.valid() && !het) ray.maxt = mei.t;                              // medium->is_homogeneous() only (volpath.cpp:221)
        if (!needs_intersection) si = compute_si(sc, ray, hkeep);                // the interaction a null collision kept
        else if (!proven_empty) { hkeep = tr.closest(ray); si = tr.surface(sc, ray, hkeep); }   // else: no surface within mei.t (look-ahead of the previous trip)
        if (si.t < mei.t) mei.t = kInf;
        if (M.has_spectral_extinction) {
            float t = fmin_(mei.t, si.t) - mei.mint;
            V3 tr = exp_neg(t, mei.combined);
            V3 pdf = (si.t < mei.t) ? tr : tr * mei.combined;
            float tr_pdf = idx3(pdf, channel);
            throughput = throughput * ((tr_pdf > 0.f) ? div_uniform(tr, tr_pdf) : V3(0.f));
        }
        escaped_medium = !mei.valid();
        active_medium = mei.valid();
        if (!active_medium) rng.skip(1);              // volpath.cpp:239
        if (active_medium) {
            const float u_null = rng.next();          // null/real collision draw (sigma_n = 0: always real)
            if (HET) {                                // volpath.cpp:238-250
                const float null_scatter_prob = mean3(mei.sigma_n / mei.combined);
                act_null_scatter = u_null < null_scatter_prob;
                if (M.has_spectral_extinction && act_null_scatter) throughput = throughput * (mei.sigma_n / null_scatter_prob);
            }
            act_medium_scatter = !act_null_scatter;
            if (act_medium_scatter) { depth += 1; s.lp = mei.p; }
        }
    }
    active = active && depth < max_depth;
    act_medium_scatter = act_medium_scatter && active;
    if (HET && act_null_scatter) { ray.o = mei.p; hkeep.t = si.t - mei.t; }    // :254-257 (si.t -= mei.t)
    if (clk) clk->at(2);                                                // (developer section timer: record arrival + medium interaction)
    if (!act_medium_scatter) rng.skip(3);                               // volpath.cpp:407 (NEE), 288, 289
    if (act_medium_scatter) {
        const DMedium M = tab(sc.media, medium);
        if (HET && M.het) {
            if (M.has_spectral_extinction) throughput = throughput * (mei.sigma_s / mean3(mei.sigma_t / mei.combined));
            else throughput = throughput * (mei.sigma_s / mei.sigma_t);
        } else if (M.has_spectral_extinction) throughput = throughput * V3(M.w_spec[0], M.w_spec[1], M.w_spec[2]);   // sigma_s / mean(sigma_t / combined), per-medium constant (k_medium_prepare)
        else throughput = throughput * V3(M.w_plain[0], M.w_plain[1], M.w_plain[2]);                                 // sigma_s / sigma_t
        bool sample_emitters = M.sample_emitters != 0;
        valid_ray = true;
        specular_chain = !sample_emitters;
        if (!sample_emitters) rng.skip(1);
        if (sample_emitters) {
            // Exact early rejection (sc.nee_fast_reject: one infinite emitter, no null BSDF in the scene).  The emitter
            // sample lies at distance 2*max(r_bsphere, |p - c|) whatever its direction (envmap.cpp:431-433,
            // constant.cpp sample_direction), the march draws ONE free-flight distance, and a collision inside the
            // segment zeroes the sample (sigma_n = 0; every surface blocks).  So when the free-flight distance is
            // safely below that bound the contribution is exactly 0 and only the three random numbers are consumed.
            // The guards keep (sx, sy) away from the map's border rows/columns, the only place where the sampled
            // density can be 0 (which would skip the third draw); everything else takes the full routine.
            bool rejected = false;
            if (sc.nee_fast_reject) {
                SMP saved = rng;
                float sx, sy; rng.next2(sx, sy);                        // the emitter sample, then the march's single draw
                float u3 = rng.next();
                const float lo = 9.5367431640625e-7f, hi = 1.f - 9.5367431640625e-7f;
                bool interior = sc.env.type == LRT_EMITTER_CONSTANT || (sx > lo && sx < hi && sy > lo && sy < hi);
                // (the emitter sample is at least 2 r away: 1 - u3 >= nee_vmin decides the comparison below without the logarithm, upload_media)
                rejected = interior;
                if (!(1.f - u3 >= idx3(V3(M.nee_vmin[0], M.nee_vmin[1], M.nee_vmin[2]), channel))) {
                    float sampled_t = 0.f + (-m_log(1.f - u3) / idx3(V3(M.sigma_t[0], M.sigma_t[1], M.sigma_t[2]), channel));
                    V3 cc(sc.env.bsphere_c[0], sc.env.bsphere_c[1], sc.env.bsphere_c[2]);
                    float dist = 2.f * fmax_(sc.env.bsphere_r, norm(mei.p - cc));
                    rejected = interior && sampled_t <= dist * 0.998f - 1e-3f;
                }
                if (!rejected) rng = saved;
            }
            if (!rejected) {
                DirSample ds;
                V3 emitted = volpath_sample_emitter<HET>(sc, rng, mei.p, V3(0.f), false, 0, V3(0.f), medium, channel, &ds, tr, n_shadow);
                float phase_val = phase_eval(M, mei.wi, ds.d);
                V3 c = throughput * phase_val * emitted * mis_weight(ds.pdf, ds.delta ? 0.f : phase_val);
                result = result + c;
            }
        }
        (void) rng.next();
        float s2x, s2y; rng.next2(s2x, s2y);
        V3 wo; float phase_pdf; phase_sample(M, mei.wi, s2x, s2y, &wo, &phase_pdf);
        if (phase_pdf > 0.f) {
            ray = spawn_ray(mei.p, V3(0.f), wo);
            s.last_pdf = phase_pdf;
        }
    }
    // ---- surface interactions
    active_surface = active_surface || escaped_medium;
    bool intersect = active_surface && !escaped_medium;   // medium lanes already hold si
    if (intersect) { Hit h = tr.closest(ray); si = tr.surface(sc, ray, h); }
    if (active_surface) {
        if (rp.hide_emitters && depth == 0 && intersect) {         // volpath.cpp:304-320, integrator.cpp:96-123
            bool skip = si.valid && tab(sc.shapes, si.shape, sc.one_shape).emitter >= 0;
            if (skip) {
                Ray r2 = spawn_ray(si.p, si.n, ray.d);
                bool a = true; Hit h; h.prim = 0xffffffffu; h.t = kInf; h.u = h.v = 0.f;
                while (a) {
                    h = tr.closest(r2);
                    a = h.prim != 0xffffffffu && tab(sc.shapes, sc.face_shape[h.prim], sc.one_shape).emitter >= 0;
                    if (a) { SI s2 = compute_si(sc, r2, h); r2 = spawn_ray(s2.p, s2.n, r2.d); }
                }
                si = compute_si(sc, r2, h);
            }
        }
        bool count_direct = (depth == 0) || specular_chain;
        int emitter = si_emitter(sc, si);
        bool active_e = emitter >= 0 && !(depth == 0 && rp.hide_emitters);
        if (active_e) {
            float emitter_pdf = 1.f;
            if (!count_direct) emitter_pdf = pdf_emitter_direction(sc, s.lp, si, emitter);
            V3 emitted = emitter_eval(sc, emitter, si);
            V3 contrib = count_direct ? throughput * emitted : throughput * mis_weight(s.last_pdf, emitter_pdf) * emitted;
            result = result + contrib;
        }
    }
    active_surface = active_surface && si.valid;
    if (!active_surface) rng.skip(3);                                   // volpath.cpp:407 (NEE), 366, 367
    if (active_surface) {
        const DShape sd = tab(sc.shapes, si.shape, sc.one_shape);
        int b = sd.bsdf;
        int flags = tab(sc.bsdfs, b, sc.one_shape).flags;
        bool active_e = (flags & F_SMOOTH) && (depth + 1 < max_depth);
        if (!active_e) rng.skip(1);
        if (active_e) {
            DirSample ds;
            V3 emitted = volpath_sample_emitter<HET>(sc, rng, si.p, si.n, true, si.shape, si.n, medium, channel, &ds, tr, n_shadow);
            V3 wo = si.sh.to_local(ds.d);
            V3 bsdf_val = bsdf_eval(sc, b, si, wo);
            float bpdf = bsdf_pdf(sc, b, si, wo);
            V3 c = throughput * bsdf_val * mis_weight(ds.pdf, ds.delta ? 0.f : bpdf) * emitted;
            result = result + c;
        }
        float s1 = rng.next(), s2x, s2y; rng.next2(s2x, s2y);
        const BSDFSample bs = bsdf_sample(sc, b, si, s1, s2x, s2y);
        throughput = throughput * bs.weight;
        eta *= bs.eta;
        ray = spawn_ray(si.p, si.n, si.sh.to_world(bs.wo));
        bool non_null = !(bs.type & F_NULL);
        if (non_null) { depth += 1; s.lp = si.p; s.last_pdf = bs.pdf; valid_ray = true; }
        specular_chain = specular_chain || (non_null && (bs.type & F_DELTA));
        specular_chain = specular_chain && !(bs.type & F_SMOOTH);
        if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, si.n);
    }
    active = active && (active_surface || active_medium);
    // ---- the first stage of the next trip.  PRE: for real (see above): a path that stops there is retired now and that trip is counted
    // in n_extra; inside a medium the next free-flight distance is then known and the distance field may prove that the segment
    // reaches no surface: such paths are queued apart and skip their ray query.
    // !PRE (heterogeneous media): the same as a look-ahead on a copy of the generator; only the retirement and the proof are kept.
    if (clk) clk->at(3);                                                // (scatter: weights, NEE rejection, phase sample)
    if (PRE) {
        if (active) { if (!termination_stage()) { active = false; n_extra += 1; } else free_flight_stage(); }
        if (clk) clk->at(4);                                            // (termination + free-flight stage: draws, log, distance field)
    } else if (active) {
        SMP pk = rng;
        bool a2 = any_nonzero(throughput);
        float q2 = fmin_(max3(throughput) * sqr(eta), .95f);
        if (a2) { float u = pk.next(); a2 = (u < q2) || !(depth > (uint32_t) rp.rr_depth); }
        a2 = a2 && depth < max_depth;
        if (!a2) { active = false; n_extra += 1; rng = pk; }      // the retired trip's Russian-roulette draw stays consumed (multi-pass renders)
        else if (medium >= 0 && sc.grid.enabled && !(HET && act_null_scatter)) {
            const DMedium M = tab(sc.media, medium);
            if (!(HET && M.het)) {                    // a heterogeneous medium does not shorten the ray: its query is always the full one
                const float t2 = medium_sampled_t(M, pk.next(), channel);
                const MI m2 = medium_interaction_at(M, ray, t2);
                if (m2.valid() && segment_free_of_surfaces(sc.grid, ray.o, ray.d, m2.t)) nohit = PF_NOHIT;
            }
        }
    }
    commit();
    s.flags |= nohit; s.ff_t = cache_t;
    if (HET && act_null_scatter && active) { s.flags |= PF_HAVE_SI; s.hit = make_float4(hkeep.t, hkeep.u, hkeep.v, u2f(hkeep.prim)); }
    return active;
}

// One trip of path's while_loop (src/integrators/path.cpp:194-338); the
// ray_intersect_preliminary of the previous trip (:332-337, or :164-169 for the
// first one) is the trace at the top.
template <typename SMP, typename TR>
DEV bool path_iteration(SceneRef sc, RpRef rp, PathState &s, SMP &rng, const TR &tr, uint32_t &n_shadow) {
    uint32_t depth = s.flags & PF_DEPTH_MASK;
    bool prev_bsdf_delta = (s.flags & PF_SPECULAR) != 0, valid_ray = (s.flags & PF_VALID) != 0;
    const uint32_t max_depth = (uint32_t) rp.max_depth;
    V3 throughput = s.tp, result = s.res;
    float eta = s.eta;
    Ray ray; ray.o = s.o; ray.d = s.d; ray.maxt = s.maxt;
    auto commit = [&]() {
        s.tp = throughput; s.res = result; s.eta = eta; s.o = ray.o; s.d = ray.d; s.maxt = ray.maxt;
        s.flags = (depth & PF_DEPTH_MASK) | (prev_bsdf_delta ? PF_SPECULAR : 0u) | (valid_ray ? PF_VALID : 0u);
    };
    if (max_depth == 0) { commit(); return false; }
    Hit pi = tr.closest(ray);
    if (rp.hide_emitters && depth == 0) {                          // path.cpp:178-192
        bool skip = pi.prim != 0xffffffffu && tab(sc.shapes, sc.face_shape[pi.prim], sc.one_shape).emitter >= 0;
        if (skip) {
            SI s0 = compute_si(sc, ray, pi);
            Ray r2 = spawn_ray(s0.p, s0.n, ray.d);
            bool a = true; Hit h; h.prim = 0xffffffffu; h.t = kInf; h.u = h.v = 0.f;
            while (a) {
                h = tr.closest(r2);
                a = h.prim != 0xffffffffu && tab(sc.shapes, sc.face_shape[h.prim], sc.one_shape).emitter >= 0;
                if (a) { SI s2 = compute_si(sc, r2, h); r2 = spawn_ray(s2.p, s2.n, r2.d); }
            }
            pi = h; ray = r2;
        }
    }
    SI si = compute_si(sc, ray, pi);
    int emitter = si_emitter(sc, si);
    if (emitter >= 0) {
        float em_pdf = 0.f;
        if (!prev_bsdf_delta) em_pdf = pdf_emitter_direction(sc, s.lp, si, emitter);
        float mis_bsdf = mis_weight(s.last_pdf, em_pdf);
        V3 em = (s.last_pdf > 0.f) ? emitter_eval(sc, emitter, si) : V3(0.f);
        em = em * mis_bsdf;
        result = V3(fma_(throughput.x, em.x, result.x), fma_(throughput.y, em.y, result.y), fma_(throughput.z, em.z, result.z));
    }
    bool active_next = (depth + 1 < max_depth) && si.valid;
    if (!active_next) {
        // path.cpp:227-231: a JIT variant never takes the `dr::none_or<false>` exit; the rest of the trip runs with active_em =
        // false: six sampler values are drawn (:246, :266-267, :326) and valid_ray |= si.is_valid() && !Null (:305-306)
        float a0, a1; rng.next2(a0, a1); (void) rng.next(); rng.next2(a0, a1); (void) rng.next();
        if (si.valid && !(tab(sc.bsdfs, tab(sc.shapes, si.shape, sc.one_shape).bsdf, sc.one_shape).flags & F_NULL)) valid_ray = true;
        commit(); return false;
    }
    const DShape sd = tab(sc.shapes, si.shape, sc.one_shape);
    int b = sd.bsdf;
    bool active_em = (tab(sc.bsdfs, b, sc.one_shape).flags & F_SMOOTH) != 0;
    DirSample ds; ds.pdf = 0.f; ds.delta = false; ds.d = V3(0.f);
    V3 em_weight(0.f), wo(0.f);
    // path.cpp:246-248: ls.sampler->next_2d() carries no mask and sits in an `if (dr::any_or<true>(active_em))`, which a
    // symbolic loop always traces: every lane in the loop consumes the two numbers, smooth BSDF or not
    float sx, sy; rng.next2(sx, sy);
    if (active_em) {
        em_weight = sample_emitter_direction(sc, si.p, sx, sy, &ds);
        if (ds.pdf != 0.f) {                                       // scene.cpp:361-365 test_visibility
            Ray sr = spawn_ray_to(si.p, si.n, ds.p);
            n_shadow++;
            Hit h = tr.any(sr);
            if (h.prim != 0xffffffffu) { em_weight = V3(0.f); ds.pdf = 0.f; }
        }
        active_em = ds.pdf != 0.f;
        wo = si.sh.to_local(ds.d);
    }
    float s1 = rng.next(), s2x, s2y; rng.next2(s2x, s2y);
    V3 bsdf_val = bsdf_eval(sc, b, si, wo);
    float bpdf = bsdf_pdf(sc, b, si, wo);
    const BSDFSample bs = bsdf_sample(sc, b, si, s1, s2x, s2y);
    const V3 bsdf_weight = bs.weight;
    if (active_em) {
        float mis_em = ds.delta ? 1.f : mis_weight(ds.pdf, bpdf);
        V3 c = bsdf_val * em_weight * mis_em;
        result = V3(fma_(throughput.x, c.x, result.x), fma_(throughput.y, c.y, result.y), fma_(throughput.z, c.z, result.z));
    }
    ray = spawn_ray(si.p, si.n, si.sh.to_world(bs.wo));
    throughput = throughput * bsdf_weight;
    eta *= bs.eta;
    valid_ray = valid_ray || !(bs.type & F_NULL);
    s.lp = si.p; s.last_pdf = bs.pdf; prev_bsdf_delta = (bs.type & F_DELTA) != 0;
    depth += 1;
    float tmax = max3(throughput);
    float rr_prob = fmin_(tmax * sqr(eta), .95f);
    bool rr_active = depth >= (uint32_t) rp.rr_depth, rr_continue = rng.next() < rr_prob;
    if (rr_active) throughput = throughput * rcp(rr_prob);
    bool active = active_next && (!rr_active || rr_continue) && (tmax != 0.f);
    commit();
    return active;
}

} // namespace lrt
#include "kernels_bio.h"
#include "kernels_mis.h"
namespace lrt {

// ---- the render kernel.  ONE launch per lrt_render: every workgroup is persistent and owns a private pool of path records
// in HBM (two queues of cap = 2P records, P = paths in flight per workgroup).  A round of a workgroup:
//   1. top the pool up to P paths with fresh camera lanes, taken from the global lane ticket (one atomic per round);
//   2. its waves pull 64-path tiles from an LDS ticket, independently of each other (no workgroup barrier inside a
//      round, so one wave's memory latency is hidden by the other waves of its SIMD); a tile is one region of the in-queue
//      or a batch of fresh lanes, which are generated in registers and run their first trip without touching HBM;
//   3. survivors go to the out-queue (slots from LDS counters: wave ballot + one LDS atomic per region), finished paths
//      are reduced per pixel inside the wave and splatted;  barrier, swap queues.
// A queue holds three regions, so that a wave's lanes stay on the same branch of the loop body:
//   A  [0, n_a)            in-medium paths whose next segment is proven free of surfaces (no ray query)
//   C  [P, P + n_c)        in-medium paths that need their ray query
//   B  2P-1-j, j < n_b     paths outside media
//   L  P-1-j, j < n_l      (biovolpath) in-medium paths whose ray query is unbounded: apart from the bounded ones of C, whose traversals are short
//                          (Liver-MultiMesh with both kinds in one region: 17 of 64 lanes busy in an average traversal step)
// n_a + n_b + n_c + n_l <= P, so the regions never collide.  No cross-workgroup dependency exists besides the lane ticket and the
// film atomics; every wave leaves its loops once the ticket is exhausted and its pool is empty.
template <typename QS> DEV DPathStreams offset_streams(const QS &q, size_t off) {
    DPathStreams r; r.o_maxt = q.o_maxt + off; r.d_eta = q.d_eta + off; r.tp_pdf = q.tp_pdf + off; r.res_flags = q.res_flags + off; r.lp_lane = q.lp_lane + off; r.rng = q.rng + off; r.tdepth = q.tdepth + off; r.hit = q.hit + off; r.w1 = q.w1 + off; r.w2 = q.w2 + off; r.w3 = q.w3 + off; r.w4 = q.w4 + off;
    return r;
}

// READLANE: how the three regions' slot bases reach the lanes (see below)
template <int MODE = 0, bool READLANE = true, bool COMPACT = false, bool LONGQ = false>
DEV void retire_and_compact_wave(SceneRef sc, RpRef rp, bool had_path, bool alive, const PathState &s,
                                 float *__restrict__ film, float *__restrict__ sample_out, uint64_t sample_base,
                                 const LRT_CONST DPathStreams &qout, size_t pool, uint32_t P, uint32_t *s_out /* LDS [3] */, StampClock *clk = nullptr) {
    const uint32_t lane_in_wave = threadIdx.x & 63u;
    // survivors first: their stores are the oldest memory operations the next tile's record loads have to wait for (the loads reuse
    // the registers the stores read), so they go out before the film sums, not after them
    // LONGQ (biovolpath): region 3 = L, in-medium paths whose next ray query is unbounded (PF_LONG_QUERY), apart from the bounded ones of region C
    const int region = !(s.flags & PF_MEDIUM_MASK) ? 2 : ((s.flags & PF_NOHIT) ? 0 : ((LONGQ && (s.flags & PF_LONG_QUERY)) ? 3 : 1));
    const unsigned long long m0 = __ballot(alive && region == 0), m1 = __ballot(alive && region == 1), m2 = __ballot(alive && region == 2);
    const unsigned long long m3 = LONGQ ? __ballot(alive && region == 3) : 0ull;
    uint32_t base = 0;
    if (lane_in_wave < (LONGQ ? 4u : 3u)) {
        const uint32_t c = (uint32_t) __popcll(lane_in_wave == 0 ? m0 : (lane_in_wave == 1 ? m1 : ((!LONGQ || lane_in_wave == 2) ? m2 : m3)));
        if (c) base = atomicAdd(&s_out[lane_in_wave], c);
    }
    // Lanes 0 .. 2 hold the three regions' slot bases.  READLANE: three v_readlane and two selects instead of one shuffle through LDS (ds_bpermute):
    // in the volpath kernel the LDS is busy with the other waves' traversals and a round trip through it costs the wave hundreds of cycles
    // (C3 -4.3 % time); the kernels that run at the VALU issue limit (path, biovolpath06: section 6a) lose 1 % to the extra instructions and keep the shuffle.
    uint32_t b;
    if (READLANE) {
        const uint32_t b0 = (uint32_t) __builtin_amdgcn_readlane((int) base, 0), b1 = (uint32_t) __builtin_amdgcn_readlane((int) base, 1), b2 = (uint32_t) __builtin_amdgcn_readlane((int) base, 2);
        b = region == 0 ? b0 : (region == 1 ? b1 : b2);
        if (LONGQ) { const uint32_t b3 = (uint32_t) __builtin_amdgcn_readlane((int) base, 3); if (region == 3) b = b3; }
    } else b = __shfl(base, region);
    if (clk) clk->at(5);                                                // (developer section timer: ballots, slot counters)
    if (alive) {
        const unsigned long long mine = region == 0 ? m0 : (region == 1 ? m1 : ((!LONGQ || region == 2) ? m2 : m3));
        const uint32_t slot = b + (uint32_t) __popcll(mine & ((1ull << lane_in_wave) - 1ull));
        store_state<MODE>(qout, pool + (region == 0 ? slot : (region == 1 ? P + slot : ((!LONGQ || region == 2) ? 2u * P - 1u - slot : P - 1u - slot))), s, COMPACT);
    }
    if (clk) clk->at(6);                                                // (stores issued)
    if (rp.pass_out && had_path && !alive) rp.pass_out[lane_local_index(rp, s.lane)] = s.rng_state;       // next pass continues this stream
    finish_paths_wave(sc, rp, film, sample_out, sample_base, had_path && !alive, s.lane, s.res, (s.flags & PF_VALID) != 0);
}

// 4 waves per SIMD (128 VGPRs): one 1024-thread workgroup per CU, or four 256-thread ones.  BLOCK = 512 (one workgroup per CU, 2 waves per
// SIMD, 256 VGPRs): the wide-record integrators (volpathmis: 18 weights per path), which at 128 registers spill 430 B per lane
// COMPACT: 80-byte records (store_state): the host launches these instances for scenes without area emitters (DRenderParams::compact)
template <int INTEGRATOR, int BLOCK, bool LDS_BVH, bool LD, bool COMPACT = false>
__global__ void __launch_bounds__(BLOCK, BLOCK == 512 ? 2 : (BLOCK == 768 ? 3 : 4))
k_render(ScenePtr scp, LaunchPtr lp) {
    SceneRef sc = *scp;
    const LRT_CONST DLaunch &A = *lp;                          // launch arguments: scalar loads where they are used
    RpRef rp = A.rp;
    const LRT_CONST DLdsInfo &li = A.li;
    const uint32_t P = A.P;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t s_in[4], s_out[4], s_ticket, s_fresh;
    __shared__ unsigned long long s_fresh_base, s_prof[8];
#ifdef LRT_STAMP
    __shared__ unsigned long long s_stamp[16];
    if (threadIdx.x < 16) s_stamp[threadIdx.x] = 0;
#endif
    const uint32_t tid = threadIdx.x, lane_in_wave = tid & 63u;
    constexpr bool READLANE = INTEGRATOR != LRT_INTEGRATOR_PATH && INTEGRATOR != LRT_INTEGRATOR_BIOVOLPATH06;
    constexpr bool LONGQ = INTEGRATOR == LRT_INTEGRATOR_BIOVOLPATH;          // fourth queue region (retire_and_compact_wave)
    constexpr int MODE = (INTEGRATOR == LRT_INTEGRATOR_BIOVOLPATH || INTEGRATOR == LRT_INTEGRATOR_BIOVOLPATH06) ? 1 : (INTEGRATOR == LRT_INTEGRATOR_VOLPATH_HET ? 2 : ((INTEGRATOR == LRT_INTEGRATOR_VOLPATHMIS || INTEGRATOR == LRT_INTEGRATOR_VOLPATHMIS_PLAIN) ? 3 : 0));
    if (tid < 8) s_prof[tid] = 0;
    const unsigned long long t_wg_start = (rp.profile & 1u) ? wall_clock64() : 0ull;
    unsigned long long t_loop_start = 0ull, t_barrier = 0ull;
    LdsScene L{};
    if (LDS_BVH) {
        const uint4 *src = li.blob; uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t k = tid; k < li.blob_bytes / 16u; k += BLOCK) dst[k] = src[k];
        L.nodes = reinterpret_cast<const float4 *>(smem + li.nodes_off); L.verts = reinterpret_cast<const float4 *>(smem + li.verts_off);
        L.tris = reinterpret_cast<const uint2 *>(smem + li.tris_off);
        L.n_faces = sc.n_faces; L.root_is_leaf = (uint32_t) sc.root_is_leaf; L.root_first = sc.root_leaf_first; L.root_count = sc.root_leaf_count;
    }
    const LdsTracer<BLOCK> tr_lds{ L, reinterpret_cast<uint16_t *>(smem + li.stack_off) + tid };
    const GlobalTracer tr_glb{ sc, reinterpret_cast<int *>(smem) + tid };
    const size_t pool = (size_t) blockIdx.x * 2u * P;
    uint32_t parity = 0;                                      // queue the round reads: parity ? A.q1 : A.q0 (the stream pointers are scalar loads at the point of use)
    if (tid == 0) { s_in[0] = s_in[1] = s_in[2] = 0; if (LONGQ) s_in[3] = 0; }
    bool lanes_left = true;                                   // thread 0
    uint32_t n_shadow = 0, n_extra = 0, n_trips = 0, n_loaded = 0;
    if (rp.profile & 1u) t_loop_start = wall_clock64();
    for (;;) {
        if (tid == 0) {
            const uint32_t want = P - (s_in[0] + s_in[1] + s_in[2] + (LONGQ ? s_in[3] : 0u));
            uint32_t got = 0; unsigned long long base = 0;
            if (want && lanes_left) {
                base = atomicAdd(&A.cnt->next_lane, (unsigned long long) want);
                if (base < rp.n_lanes) got = (uint32_t) (rp.n_lanes - base < (unsigned long long) want ? rp.n_lanes - base : (unsigned long long) want);
                lanes_left = base + want < rp.n_lanes;
            }
            s_fresh = got; s_fresh_base = base; s_ticket = 0; s_out[0] = s_out[1] = s_out[2] = 0; if (LONGQ) s_out[3] = 0;
        }
        __syncthreads();
        const uint32_t n_a = s_in[0], n_c = s_in[1], n_b = s_in[2], n_l = LONGQ ? s_in[3] : 0u, fresh = s_fresh;
        const unsigned long long fresh_base = s_fresh_base;
        if (n_a + n_c + n_b + n_l + fresh == 0) break;
        const uint32_t ta = (n_a + 63u) >> 6, tc = (n_c + 63u) >> 6, tb = (n_b + 63u) >> 6, tf = (fresh + 63u) >> 6;
        const uint32_t tcl = tc + (LONGQ ? (n_l + 63u) >> 6 : 0u);          // tiles of C, then of L
        const uint32_t n_tiles = ta + tcl + tb + tf;
        for (;;) {
            uint32_t t = 0;
            if (lane_in_wave == 0) t = atomicAdd(&s_ticket, 1u);            // (tickets of 2 or 4 consecutive tiles, one LDS round trip each, and the next tile's ticket drawn before the compaction: measured, no better)
            t = (uint32_t) __builtin_amdgcn_readfirstlane((int) t);
            if (t >= n_tiles) break;
            const unsigned long long t_begin = (rp.profile & 1u) ? wall_clock64() : 0ull;
#ifdef LRT_STAMP
            StampClock clk; clk.start(s_stamp, LRT_STAMP_KIND == 0 ? t < ta : (LRT_STAMP_KIND == 1 ? (t >= ta && t < ta + tcl) : (LRT_STAMP_KIND == 2 ? (t >= ta + tcl && t < ta + tcl + tb) : t >= ta + tcl + tb)));
#endif
            bool had_path = false, alive = false;
            PathState s; s.flags = 0; s.lane = 0; s.res = V3(0.f);
            if (t < ta + tcl + tb) {
                uint32_t i;
                if (t < ta) { i = (t << 6) + lane_in_wave; had_path = i < n_a; }
                else if (t < ta + tc) { i = ((t - ta) << 6) + lane_in_wave; had_path = i < n_c; i += P; }
                else if (LONGQ && t < ta + tcl) { i = ((t - ta - tc) << 6) + lane_in_wave; had_path = i < n_l; i = P - 1u - i; }
                else { i = ((t - ta - tcl) << 6) + lane_in_wave; had_path = i < n_b; i = 2u * P - 1u - i; }
                if (had_path) { load_state<MODE>(parity ? A.q1 : A.q0, pool + i, s, COMPACT); n_loaded += 1; }
            } else {
                const uint32_t i = ((t - ta - tcl - tb) << 6) + lane_in_wave;
                had_path = i < fresh;
                if (had_path) s = generate_camera_path<LD>(sc, rp, A.pixel_list, A.lane_begin + fresh_base + i);
            }
#ifdef LRT_STAMP
            clk.at(0);                                                  // ticket, index arithmetic, load issue
#endif
            if (had_path) {
                SamplerT<LD> rng = COMPACT ? lane_rng_resume_word<LD>(rp, s.lane, s.rng_state, s.rng_word) : lane_rng_resume<LD>(rp, s.lane, s.rng_state);
#ifdef LRT_STAMP
                clk.at(1);                                              // sampler resume (TEA) - needs the lane id: first wait for the record
#endif
                if (INTEGRATOR == LRT_INTEGRATOR_PATH) alive = LDS_BVH ? path_iteration(sc, rp, s, rng, tr_lds, n_shadow) : path_iteration(sc, rp, s, rng, tr_glb, n_shadow);
                else if (INTEGRATOR == LRT_INTEGRATOR_BIOVOLPATH) { const bool fresh_tile = t >= ta + tcl + tb; alive = LDS_BVH ? biovolpath_iteration(sc, rp, s, rng, tr_lds, n_shadow, n_extra, fresh_tile) : biovolpath_iteration(sc, rp, s, rng, tr_glb, n_shadow, n_extra, fresh_tile); }
                else if (INTEGRATOR == LRT_INTEGRATOR_BIOVOLPATH06) alive = LDS_BVH ? biovolpath06_iteration(sc, rp, s, rng, tr_lds) : biovolpath06_iteration(sc, rp, s, rng, tr_glb);
                else if (INTEGRATOR == LRT_INTEGRATOR_VOLPATHMIS) alive = LDS_BVH ? volpathmis_iteration<true>(sc, rp, s, rng, tr_lds, n_shadow) : volpathmis_iteration<true>(sc, rp, s, rng, tr_glb, n_shadow);
                else if (INTEGRATOR == LRT_INTEGRATOR_VOLPATHMIS_PLAIN) alive = LDS_BVH ? volpathmis_iteration<false>(sc, rp, s, rng, tr_lds, n_shadow) : volpathmis_iteration<false>(sc, rp, s, rng, tr_glb, n_shadow);
                else if (INTEGRATOR == LRT_INTEGRATOR_VOLPATH_HET) alive = LDS_BVH ? volpath_iteration<true>(sc, rp, s, rng, tr_lds, n_shadow, n_extra) : volpath_iteration<true>(sc, rp, s, rng, tr_glb, n_shadow, n_extra);
                else {
                    const bool fresh_tile = t >= ta + tcl + tb;
#ifdef LRT_STAMP
                    alive = LDS_BVH ? volpath_iteration<false>(sc, rp, s, rng, tr_lds, n_shadow, n_extra, fresh_tile, &clk) : volpath_iteration<false>(sc, rp, s, rng, tr_glb, n_shadow, n_extra, fresh_tile, &clk);
#else
                    alive = LDS_BVH ? volpath_iteration<false>(sc, rp, s, rng, tr_lds, n_shadow, n_extra, fresh_tile) : volpath_iteration<false>(sc, rp, s, rng, tr_glb, n_shadow, n_extra, fresh_tile);
#endif
                }
                s.rng_state = rng.state;
                n_trips += 1;
            }
#ifdef LRT_STAMP
            retire_and_compact_wave<MODE, READLANE, COMPACT, LONGQ>(sc, rp, had_path, alive, s, A.film, A.sample_out, A.sample_base, parity ? A.q0 : A.q1, pool, P, s_out, &clk);
            clk.at(7);                                                  // film sums + atomics
            if (clk.on && lane_in_wave == 0) atomicAdd(&s_stamp[15], 1ull);
#else
            retire_and_compact_wave<MODE, READLANE, COMPACT, LONGQ>(sc, rp, had_path, alive, s, A.film, A.sample_out, A.sample_base, parity ? A.q0 : A.q1, pool, P, s_out);
#endif
            if ((rp.profile & 1u) && lane_in_wave == 0) {
                const int region = t < ta ? 0 : (t < ta + tcl ? 1 : (t < ta + tcl + tb ? 2 : 3));
                atomicAdd(&s_prof[region], wall_clock64() - t_begin); atomicAdd(&s_prof[4 + region], 1ull);
            }
        }
        { const unsigned long long tb0 = (rp.profile & 1u) ? wall_clock64() : 0ull;
        __syncthreads();
        if (rp.profile & 1u) t_barrier += wall_clock64() - tb0; }
        if (tid == 0) { s_in[0] = s_out[0]; s_in[1] = s_out[1]; s_in[2] = s_out[2]; if (LONGQ) s_in[3] = s_out[3]; }
        parity ^= 1u;
    }
    if ((rp.profile & 1u) && (tid & 63u) == 0) {              // workgroup timeline; barrier waits summed over the 16 waves' first lanes
        const unsigned long long t_end = wall_clock64();
        if (tid == 0) { atomicAdd(&A.cnt->prof_wg[0], t_loop_start - t_wg_start); atomicMax(&A.cnt->prof_wg[1], t_end); atomicMax(&A.cnt->prof_wg[2], ~t_wg_start); atomicAdd(&A.cnt->prof_wg[3], t_end - t_wg_start); atomicMax(&A.cnt->prof_wg[5], t_wg_start); }
        atomicAdd(&A.cnt->prof_wg[4], t_barrier);
    }
    if ((rp.profile & 1u) && tid < 8) { if (tid < 4) atomicAdd(&A.cnt->prof_cycles[tid], s_prof[tid]); else atomicAdd(&A.cnt->prof_tiles[tid - 4], s_prof[tid]); }
#ifdef LRT_TRAV_STATS
    if (tid == 0 && blockIdx.x == 0) printf("[trav, so far] closest: node steps %llu (lanes %.1f), leaf steps %llu (lanes %.1f); any-hit: node steps %llu (lanes %.1f), leaf steps %llu (lanes %.1f)\n", g_trav[0], (double) g_trav[1] / g_trav[0], g_trav[2], (double) g_trav[3] / g_trav[2], g_trav[4], (double) g_trav[5] / g_trav[4], g_trav[6], (double) g_trav[7] / g_trav[6]);
#endif
#ifdef LRT_STAMP
    __syncthreads();
    if (tid == 0 && blockIdx.x == 7) printf("[stamp] tiles of kind %d in workgroup 7: %llu; cycles per tile: ticket+load issue %.0f, record wait+TEA %.0f, to the end of the medium interaction / ray query %.0f, scatter + surface %.0f, termination+free flight %.0f, ballots+slots %.0f, stores %.0f, film %.0f\n", LRT_STAMP_KIND, s_stamp[15],
        (double) s_stamp[0] / s_stamp[15], (double) s_stamp[1] / s_stamp[15], (double) s_stamp[2] / s_stamp[15], (double) s_stamp[3] / s_stamp[15], (double) s_stamp[4] / s_stamp[15], (double) s_stamp[5] / s_stamp[15], (double) s_stamp[6] / s_stamp[15], (double) s_stamp[7] / s_stamp[15]);
#endif
    n_trips += n_extra;
    for (int off = 32; off > 0; off >>= 1) {
        n_shadow += __shfl_down(n_shadow, off); n_trips += __shfl_down(n_trips, off); n_loaded += __shfl_down(n_loaded, off);
    }
    if (lane_in_wave == 0) {
        if (n_shadow) atomicAdd(&A.cnt->n_shadow, (unsigned long long) n_shadow);
        if (n_trips) atomicAdd(&A.cnt->n_iter, (unsigned long long) n_trips);
        if (n_loaded) atomicAdd(&A.cnt->n_records, (unsigned long long) n_loaded);
    }
}

// Distance field build (scene upload): one thread per cell, exact point-triangle distance (closest-feature regions of
// the triangle: vertices, edges, face) against every triangle slot of the BVH image (p0 | e1 | e2).
DEV float point_triangle_dist2(V3 p, V3 a, V3 ab, V3 ac) {
    V3 ap = p - a;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) return dot(ap, ap);
    V3 bp = ap - ab;
    float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) return dot(bp, bp);
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { V3 r = ap - ab * (d1 / (d1 - d3)); return dot(r, r); }
    V3 cp = ap - ac;
    float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.f && d5 <= d6) return dot(cp, cp);
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { V3 r = ap - ac * (d2 / (d2 - d6)); return dot(r, r); }
    float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) { V3 r = bp - (ac - ab) * ((d4 - d3) / ((d4 - d3) + (d5 - d6))); return dot(r, r); }
    float denom = 1.f / (va + vb + vc);
    V3 r = ap - ab * (vb * denom) - ac * (vc * denom);
    return dot(r, r);
}

__global__ void __launch_bounds__(LRT_BLOCK)
k_build_dist_grid(const float4 *__restrict__ tris, uint32_t n_slots, DDistGrid g, uint16_t *__restrict__ out, float abs_margin) {
    const size_t n_cells = (size_t) g.n[0] * g.n[1] * g.n[2];
    const size_t c = (size_t) blockIdx.x * LRT_BLOCK + threadIdx.x;
    if (c >= n_cells) return;
    const int ix = (int) (c % g.n[0]), iy = (int) ((c / g.n[0]) % g.n[1]), iz = (int) (c / ((size_t) g.n[0] * g.n[1]));
    const V3 p(g.lo[0] + ((float) ix + .5f) * g.cell, g.lo[1] + ((float) iy + .5f) * g.cell, g.lo[2] + ((float) iz + .5f) * g.cell);
    float best = kInf;
    for (uint32_t sl = 0; sl < n_slots; ++sl) {
        const float4 a = tris[3 * sl], b = tris[3 * sl + 1], cc = tris[3 * sl + 2];
        float d2 = point_triangle_dist2(p, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(cc.x, cc.y, cc.z));
        if (!(d2 >= 0.f)) d2 = 0.f;                 // degenerate triangle (NaN): be conservative
        best = fmin_(best, d2);
    }
    float d = __builtin_sqrtf(best) * .999f - abs_margin;
    d = d > 0.f ? fmin_(d, 60000.f) : 0.f;
    union { uint16_t u; _Float16 h; } cv; cv.h = (_Float16) d;                      // round to nearest, then step down if that went up
    if ((float) cv.h > d) cv.u -= 1;
    out[c] = cv.u;
}

// Film accumulation for reconstruction filters wider than a pixel (Gaussian, tent; imageblock.cpp:174-232,431-500).  The
// render kernel stores each lane's radiance (16 B / lane); this pass then walks the lanes in order, so the 64 lanes of a
// wave belong to one or two pixels: the filter footprint of a group of lanes with the same footprint origin is reduced
// inside the wave (one butterfly per cell and channel) and lane c issues the atomics of cell c.  Compared with splatting
// where paths happen to retire this divides the float atomics by the group size (up to 64).
// WEIGHTS_ONLY: accumulate just the filter weights into a one-channel film (the PRB adjoint's normalisation image,
// common.py:730-746); lane_L is not read.
template <bool WEIGHTS_ONLY>
__global__ void __launch_bounds__(LRT_BLOCK)
k_splat_lanes(ScenePtr scp, LaunchPtr lp) {
    RpRef rp = lp->rp;
    const float4 *__restrict__ lane_L = lp->L_buf; const uint32_t *__restrict__ pixel_list = lp->pixel_list;
    const uint64_t slot_base = lp->lane_begin, n = lp->n;
    float *__restrict__ film = lp->film;
    SceneRef sc = *scp;
    FilmRef F = sc.film;
    const uint64_t i = (uint64_t) blockIdx.x * LRT_BLOCK + threadIdx.x;
    const uint32_t me = threadIdx.x & 63u;
    const bool have = i < n;
    V3 L(0.f); float alpha = 0.f, relx = 0.f, rely = 0.f; int pix = 0, piy = 0; uint32_t key = 0xffffffffu;
    if (have) {
        const uint64_t j = slot_base + i;
        uint32_t lane;
        if (pixel_list) { uint32_t pj = (rp.log2_spp != 0xffffffffu) ? (uint32_t) (j >> rp.log2_spp) : (uint32_t) (j / rp.spp); lane = pixel_list[pj] * rp.spp + (uint32_t) (j - (uint64_t) pj * rp.spp); }
        else lane = (uint32_t) j;
        if (!WEIGHTS_ONLY) {
            const float4 v = lane_L[i];
            L = V3(v.x, v.y, v.z); alpha = v.w;
            if (rp.integrator == LRT_INTEGRATOR_PATH && alpha == 0.f) L = V3(0.f);     // path.cpp:342-345
        }
        int px, py; lane_to_pixel(sc, rp, lane, &px, &py);
        float jx, jy; lane_jitter(rp, lane, j, jx, jy);
        float spx = (float) px + jx, spy = (float) py + jy;
        pix = (int) __builtin_floorf(spx) - F.fn; piy = (int) __builtin_floorf(spy) - F.fn;
        relx = (float) pix + .5f - spx; rely = (float) piy + .5f - spy;
        key = (uint32_t) (piy + 0x4000) << 16 | (uint32_t) (pix + 0x4000);
    }
    const int count = F.fcount, C = F.channels;
    unsigned long long todo = __ballot(have);
    while (todo) {
        const int leader = __ffsll((long long) todo) - 1;
        const uint32_t k0 = __shfl(key, leader);
        const bool mine = have && key == k0;
        const int gx = __shfl(pix, leader), gy = __shfl(piy, leader);
        float tr = 0.f, tg = 0.f, tb = 0.f, ta = 0.f, tw = 0.f;         // lane c keeps the totals of footprint cell (chunk base + c)
        const int n_cells = count * count;
        for (int ys = 0, ci = 0; ys < count; ++ys) {
            const float wy = mine ? rfilter_eval(F, rely + (float) ys) : 0.f;
            for (int xs = 0; xs < count; ++xs, ++ci) {
                const float w = mine ? wy * rfilter_eval(F, relx + (float) xs) : 0.f;
                // lanes outside the group add exact zeros (a product with their weight 0 would turn a non-finite radiance into NaN for this group's pixels)
                float r = mine ? L.x * w : 0.f, g = mine ? L.y * w : 0.f, b = mine ? L.z * w : 0.f, a = alpha * w, ww = w;
                if (!WEIGHTS_ONLY) { r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); if (F.has_alpha) a = wave_sum(a); }
                ww = wave_sum(ww);
                if ((int) me == (ci & 63)) { tr = r; tg = g; tb = b; ta = a; tw = ww; }
                if ((ci & 63) == 63 || ci == n_cells - 1) {            // a chunk of (up to) 64 cells is complete: lane c flushes cell base + c
                    const int cell = (ci & ~63) + (int) me;            // (footprints wider than 8 x 8 pixels take several chunks: gaussian stddev > 0.875, tent radius > 3.5)
                    if (cell <= ci && (tw != 0.f || tr != tr || tg != tg || tb != tb)) {   // zero-weight cells only matter when a non-finite radiance made them NaN (imageblock.cpp adds value * 0 there)
                        const int cy = cell / count, cx = cell - cy * count;
                        const int x = gx - F.crop_offset_x + cx, y = gy - F.crop_offset_y + cy;
                        if (x >= 0 && x < F.width && y >= 0 && y < F.height) {
                            if (WEIGHTS_ONLY) atomicAdd(film + (size_t) y * F.width + x, tw);
                            else {
                                float *p = film + ((size_t) y * F.width + x) * C;
                                atomicAdd(p + 0, tr); atomicAdd(p + 1, tg); atomicAdd(p + 2, tb);
                                if (F.has_alpha) { atomicAdd(p + 3, ta); atomicAdd(p + 4, tw); } else atomicAdd(p + 3, tw);
                            }
                        }
                    }
                    tr = tg = tb = ta = tw = 0.f;
                }
            }
        }
        todo &= ~__ballot(mine);
    }
}

// src/films/hdrfilm.cpp:306-410
__global__ void k_develop(DFilm F, const float *__restrict__ film, float *__restrict__ image, uint32_t n_pixels) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    int C = F.channels, T = F.has_alpha ? 4 : 3;
    float w = film[(size_t) i * C + C - 1]; if (w == 0.f) w = 1.f;
    for (int c = 0; c < T; ++c) image[(size_t) i * T + c] = film[(size_t) i * C + c] / w;
}

template <bool ANY_HIT>
__global__ void __launch_bounds__(LRT_BLOCK)
k_trace(ScenePtr scp, const float *ox, const float *oy, const float *oz, const float *dx, const float *dy, const float *dz, const float *tmax,
        float *t, float *u, float *v, uint32_t *prim, uint32_t n) {
    SceneRef sc = *scp;
    __shared__ int s_stack[LRT_STACK * LRT_BLOCK];
    uint32_t i = blockIdx.x * LRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    Ray r; r.o = V3(ox[i], oy[i], oz[i]); r.d = V3(dx[i], dy[i], dz[i]); r.maxt = tmax[i];
    Hit h = trace<ANY_HIT>(sc, r, s_stack + threadIdx.x);
    if (ANY_HIT) { t[i] = h.prim != 0xffffffffu ? 0.f : kInf; return; }
    t[i] = h.t; if (u) u[i] = h.u; if (v) v[i] = h.v; if (prim) prim[i] = h.prim;
}

// The same queries through the LDS-resident BVH image (1024 threads per workgroup, as in k_render)
template <bool ANY_HIT>
__global__ void __launch_bounds__(1024)
k_trace_lds(ScenePtr scp, DLdsInfo li, const float *ox, const float *oy, const float *oz, const float *dx, const float *dy, const float *dz, const float *tmax,
            float *t, float *u, float *v, uint32_t *prim, uint32_t n) {
    SceneRef sc = *scp;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    {
        const uint4 *src = li.blob; uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t k = tid; k < li.blob_bytes / 16u; k += 1024) dst[k] = src[k];
    }
    LdsScene L;
    L.nodes = reinterpret_cast<const float4 *>(smem + li.nodes_off); L.verts = reinterpret_cast<const float4 *>(smem + li.verts_off);
    L.tris = reinterpret_cast<const uint2 *>(smem + li.tris_off);
    L.n_faces = sc.n_faces; L.root_is_leaf = (uint32_t) sc.root_is_leaf; L.root_first = sc.root_leaf_first; L.root_count = sc.root_leaf_count;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 1024u + tid; i < n; i += gridDim.x * 1024u) {
        Ray r; r.o = V3(ox[i], oy[i], oz[i]); r.d = V3(dx[i], dy[i], dz[i]); r.maxt = tmax[i];
        Hit h = trace_lds<ANY_HIT, 1024>(L, r, reinterpret_cast<uint16_t *>(smem + li.stack_off) + tid);
        if (ANY_HIT) { t[i] = h.prim != 0xffffffffu ? 0.f : kInf; continue; }
        t[i] = h.t; if (u) u[i] = h.u; if (v) v[i] = h.v; if (prim) prim[i] = h.prim;
    }
}


} // namespace lrt
