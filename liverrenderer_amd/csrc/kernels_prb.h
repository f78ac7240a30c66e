// Path Replay Backpropagation for homogeneous media: the primal estimator and the
// hand-derived adjoint of src/python/python/ad/integrators/prbvolpath.py:96-444
// (driver: src/python/python/ad/integrators/common.py:625-783), as wavefront kernels.
//
// Differentiated parameters: sigma_t[3], albedo[3] (src/media/homogeneous.cpp:146-151) and the HG
// asymmetry g (src/phase/hg.cpp:60-62).  With detached sampling the local derivatives are closed forms:
//   free-flight weight  w_c = exp(-t sigma_c) / pdf_k [* sigma_c a_c at a real scatter]
//       d w_c / d sigma_c = w_c (-t [+ 1/sigma_c]),   d w_c / d a_c = w_c / a_c
//   NEE transmittance   tr_c = exp(-t_seg sigma_c):  d ln tr_c / d sigma_c = -t_seg
//   phase value         d ln hg / d g = -2g/(1-g^2) - 3 (g + c) / (1 + g^2 + 2 g c)
// and PRB multiplies them by delta_L * (radiance still to be collected).
#pragma once
#include "kernels.h"

namespace lrt {

struct PrbGrads { float sigma_t[3], albedo[3], g; };   // only ever indexed with constants (stays in VGPRs)

DEV float hg_dlog_dg(float g, float c) {
    float temp = 1.f + sqr(g) + 2.f * g * c;
    return -2.f * g / (1.f - sqr(g)) - 1.5f * (2.f * g + 2.f * c) / temp;
}

// prbvolpath.py:354-444.  Returns emitter_val * transmittance; seg_sum[c] accumulates -t_seg * scale over the
// medium segments the reference backpropagates through (segments that end on a surface with tr_c > 0, :425-427).
template <typename TR>
DEV V3 prb_sample_emitter(const DScene &sc, PCG32 &rng, V3 ref_p, V3 ref_n, bool ref_is_surface, uint32_t ref_shape, V3 ref_geo_n,
                          int medium, uint32_t channel, DirSample *ds_out, const TR &tr, uint32_t &n_shadow, V3 *seg_sum) {
    float sx = rng.next(), sy = rng.next();
    DirSample ds; V3 emitter_val = sample_emitter_direction(sc, ref_p, sx, sy, &ds);
    *ds_out = ds;
    bool active = ds.pdf != 0.f;
    if (!active) { emitter_val = V3(0.f); medium = -1; }
    if (ref_is_surface) { const DShape sd = sc.shapes[ref_shape]; if (is_medium_transition(sd)) medium = target_medium(sd, ds.d, ref_geo_n); }
    Ray ray = spawn_ray_to(ref_p, ref_n, ds.p);
    float max_dist = ray.maxt, total_dist = 0.f;
    SI si; si.valid = false; si.t = 0.f; si.shape = 0; si.p = V3(0.f); si.n = V3(0.f);
    bool needs_intersection = true;
    V3 transmittance(1.f), sum(0.f);
    while (active) {
        float remaining_dist = max_dist - total_dist;
        ray.maxt = remaining_dist;
        active = active && remaining_dist > 0.f;
        needs_intersection = needs_intersection && active;
        if (needs_intersection) { n_shadow++; Hit h = tr.closest(ray); si = compute_si(sc, ray, h); }
        needs_intersection = false;
        bool active_medium = active && medium >= 0, active_surface = active && !active_medium;
        V3 tr_multiplier(1.f);
        float seg_t = 0.f, scale_t = 0.f; bool escaped_medium = false;
        if (active_medium) {
            (void) rng.next();
            const DMedium M = sc.media[medium];
            float t = fmin_(remaining_dist, si.t);
            seg_t = fmin_(t, si.t) - 0.f;
            tr_multiplier = V3(m_exp(-seg_t * M.sigma_t[0]), m_exp(-seg_t * M.sigma_t[1]), m_exp(-seg_t * M.sigma_t[2]));
            scale_t = -seg_t * M.scale;
            escaped_medium = true; active_medium = false;
        }
        active_surface = (active_surface || escaped_medium) && si.valid && !active_medium;
        if (active_surface) tr_multiplier = tr_multiplier * bsdf_null_transmission(sc, sc.shapes[si.shape].bsdf);
        if (escaped_medium && active_surface) {
            if (tr_multiplier.x > 0.f) sum.x += scale_t;
            if (tr_multiplier.y > 0.f) sum.y += scale_t;
            if (tr_multiplier.z > 0.f) sum.z += scale_t;
        }
        transmittance = transmittance * tr_multiplier;
        if (active_surface) ray = spawn_ray(si.p, si.n, ray.d);
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        active = active && (active_medium || active_surface) && any_nonzero(transmittance);
        if (active) total_dist += si.t;
        if (active_surface) { const DShape sd = sc.shapes[si.shape]; if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, si.n); }
    }
    *seg_sum = sum;
    return emitter_val * transmittance;
}

// One trip of prbvolpath's loop (prbvolpath.py:139-349).  s.res holds L: accumulated radiance (primal) or the
// radiance still to be collected (adjoint).  Returns true when the path survives.
template <bool ADJOINT, typename TR>
DEV bool prb_iteration(const DScene &sc, const DRenderParams &rp, PathState &s, PCG32 &rng, const TR &tr, uint32_t &n_shadow,
                       V3 delta_L, PrbGrads &G) {
    uint32_t depth = s.flags & PF_DEPTH_MASK;
    int medium = (int) ((s.flags & PF_MEDIUM_MASK) >> PF_MEDIUM_SHIFT) - 1;
    const uint32_t channel = (s.flags >> PF_CHANNEL_SHIFT) & 3u;
    bool specular_chain = (s.flags & PF_SPECULAR) != 0, valid_ray = (s.flags & PF_VALID) != 0;
    const uint32_t max_depth = (uint32_t) rp.max_depth;
    V3 throughput = s.tp, L = s.res;
    float eta = s.eta;
    Ray ray; ray.o = s.o; ray.d = s.d; ray.maxt = s.maxt;
    auto commit = [&]() {
        s.tp = throughput; s.res = L; s.eta = eta; s.o = ray.o; s.d = ray.d; s.maxt = ray.maxt;
        s.flags = (depth & PF_DEPTH_MASK) | ((uint32_t) (medium + 1) << PF_MEDIUM_SHIFT) | (channel << PF_CHANNEL_SHIFT) |
                  (specular_chain ? PF_SPECULAR : 0u) | (valid_ray ? PF_VALID : 0u);
    };
    bool active = any_nonzero(throughput);
    float q = fmin_(max3(throughput) * sqr(eta), 0.99f);
    bool perform_rr = depth > (uint32_t) rp.rr_depth;
    if (active) { float u = rng.next(); active = (u < q) || !perform_rr; }
    if (perform_rr) throughput = throughput * rcp(q);
    bool active_medium = active && medium >= 0, active_surface = active && !active_medium;
    bool escaped_medium = false, act_medium_scatter = false, in_medium_segment = false;
    MI mei; mei.t = kInf; mei.wi = -ray.d; mei.p = V3(0.f);
    SI si; si.valid = false; si.t = kInf;
    V3 weight(1.f);
    float seg_t = 0.f;
    if (active_medium) {
        const DMedium M = sc.media[medium];
        mei = medium_sample_interaction(M, ray, rng.next(), channel);
        if (mei.valid()) ray.maxt = mei.t;
        { Hit h = tr.closest(ray); si = compute_si(sc, ray, h); }
        if (si.t < mei.t) mei.t = kInf;
        seg_t = fmin_(mei.t, si.t) - mei.mint;
        V3 tr(m_exp(-seg_t * mei.combined.x), m_exp(-seg_t * mei.combined.y), m_exp(-seg_t * mei.combined.z));
        V3 pdf = (si.t < mei.t) ? tr : tr * mei.combined;
        float tr_pdf = idx3(pdf, channel);
        weight = (tr_pdf > 0.f) ? tr / tr_pdf : V3(0.f);
        escaped_medium = !mei.valid();
        active_medium = mei.valid();
        in_medium_segment = true;
        if (active_medium) { act_medium_scatter = true; depth += 1; s.lp = mei.p; }
    }
    active = active && depth < max_depth;
    act_medium_scatter = act_medium_scatter && active;
    if (act_medium_scatter) weight = weight * mei.sigma_s;
    throughput = throughput * weight;
    if (ADJOINT && in_medium_segment) {                                 // prbvolpath.py:199-204
        const DMedium M = sc.media[medium];
        auto term = [&](float w, float l, float dl, float st, float al, float &gs, float &ga) {
            float Lo = l / fmax_(1e-8f, w);
            float dws = w * (-seg_t) + (act_medium_scatter ? w / st : 0.f);
            if (!(seg_t < kInf)) dws = 0.f;
            gs += dl * Lo * dws * M.scale;
            if (act_medium_scatter) ga += dl * Lo * (w / al);
        };
        term(weight.x, L.x, delta_L.x, M.sigma_t[0], M.albedo[0], G.sigma_t[0], G.albedo[0]);
        term(weight.y, L.y, delta_L.y, M.sigma_t[1], M.albedo[1], G.sigma_t[1], G.albedo[1]);
        term(weight.z, L.z, delta_L.z, M.sigma_t[2], M.albedo[2], G.sigma_t[2], G.albedo[2]);
    }
    // ---- surface interactions
    active_surface = active_surface || escaped_medium;
    bool intersect = active_surface && !escaped_medium;
    if (intersect) { Hit h = tr.closest(ray); si = compute_si(sc, ray, h); }
    if (rp.hide_emitters && intersect && depth == 0 && si.valid && sc.shapes[si.shape].emitter >= 0) {
        Ray r2 = spawn_ray(si.p, si.n, ray.d);
        bool a = true; Hit h; h.prim = 0xffffffffu; h.t = kInf; h.u = h.v = 0.f;
        while (a) {
            h = tr.closest(r2);
            a = h.prim != 0xffffffffu && sc.shapes[sc.face_shape[h.prim]].emitter >= 0;
            if (a) { SI s2 = compute_si(sc, r2, h); r2 = spawn_ray(s2.p, s2.n, r2.d); }
        }
        si = compute_si(sc, r2, h);
    }
    if (active_surface) {
        bool count_direct = (depth == 0) || specular_chain;
        int emitter = si_emitter(sc, si);
        bool active_e = emitter >= 0 && !(depth == 0 && rp.hide_emitters);
        if (active_e) {
            float emitter_pdf = pdf_emitter_direction(sc, s.lp, si, emitter);
            V3 emitted = emitter_eval(sc, emitter, si);
            V3 contrib = count_direct ? throughput * emitted : throughput * mis_weight(s.last_pdf, emitter_pdf) * emitted;
            L = ADJOINT ? L - contrib : L + contrib;
        }
    }
    active_surface = active_surface && si.valid;
    // ---- emitter sampling (prbvolpath.py:267-297)
    int b = active_surface ? sc.shapes[si.shape].bsdf : 0;
    bool active_e_surface = active_surface && (sc.bsdfs[b].flags & F_SMOOTH) && (depth + 1 < max_depth);
    bool sample_emitters = act_medium_scatter ? (sc.media[medium].sample_emitters != 0) : false;
    if (act_medium_scatter) specular_chain = !sample_emitters;
    bool active_e_medium = act_medium_scatter && sample_emitters;
    if (active_e_surface || active_e_medium) {
        DirSample ds; V3 seg_sum;
        V3 rp_ = active_e_medium ? mei.p : si.p, rn = active_e_medium ? V3(0.f) : si.n;
        V3 emitted = prb_sample_emitter(sc, rng, rp_, rn, active_e_surface, active_e_surface ? si.shape : 0u, si.n, medium, channel, &ds, tr, n_shadow, &seg_sum);
        V3 nee_weight; float nee_pdf;
        if (active_e_surface) { V3 wo = si.sh.to_local(ds.d); nee_weight = bsdf_eval(sc, b, si, wo); nee_pdf = bsdf_pdf(sc, b, si, wo); }
        else { float pv = phase_eval(sc.media[medium], mei.wi, ds.d); nee_weight = V3(pv); nee_pdf = pv; }
        V3 contrib = throughput * nee_weight * mis_weight(ds.pdf, ds.delta ? 0.f : nee_pdf) * emitted;
        L = ADJOINT ? L - contrib : L + contrib;
        if (ADJOINT) {
            G.sigma_t[0] += delta_L.x * contrib.x * seg_sum.x; G.sigma_t[1] += delta_L.y * contrib.y * seg_sum.y; G.sigma_t[2] += delta_L.z * contrib.z * seg_sum.z;
            if (active_e_medium && sc.media[medium].phase == LRT_PHASE_HG)
                G.g += (delta_L.x * contrib.x + delta_L.y * contrib.y + delta_L.z * contrib.z) * hg_dlog_dg(sc.media[medium].g, dot(ds.d, mei.wi));
        }
    }
    // ---- phase function sampling (prbvolpath.py:299-317)
    if (act_medium_scatter) {
        valid_ray = true;
        const DMedium M = sc.media[medium];
        (void) rng.next();
        float s2x = rng.next(), s2y = rng.next();
        V3 wo; float phase_pdf; phase_sample(M, mei.wi, s2x, s2y, &wo, &phase_pdf);
        act_medium_scatter = phase_pdf > 0.f;
        if (act_medium_scatter) {
            if (ADJOINT && M.phase == LRT_PHASE_HG) {
                float pe = phase_eval(M, mei.wi, wo), dlg = hg_dlog_dg(M.g, dot(wo, mei.wi));
                G.g += delta_L.x * (pe * (L.x / fmax_(1e-8f, pe))) * dlg;
                G.g += delta_L.y * (pe * (L.y / fmax_(1e-8f, pe))) * dlg;
                G.g += delta_L.z * (pe * (L.z / fmax_(1e-8f, pe))) * dlg;
            }
            ray = spawn_ray(mei.p, V3(0.f), wo);
            s.last_pdf = phase_pdf;
        }
    }
    // ---- BSDF sampling (prbvolpath.py:321-349)
    if (active_surface) {
        const DShape sd = sc.shapes[si.shape];
        float s1 = rng.next(), s2x = rng.next(), s2y = rng.next();
        const BSDFSample bs = bsdf_sample(sc, b, si, s1, s2x, s2y);
        active_surface = bs.pdf > 0.f;
        if (active_surface) {
            throughput = throughput * bs.weight;
            eta *= bs.eta;
            ray = spawn_ray(si.p, si.n, si.sh.to_world(bs.wo));
            bool non_null = !(bs.type & F_NULL);
            if (non_null) { depth += 1; s.lp = si.p; s.last_pdf = bs.pdf; valid_ray = true; }
            specular_chain = specular_chain || (non_null && (bs.type & F_DELTA));
            specular_chain = specular_chain && !(bs.type & F_SMOOTH);
            if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, si.n);
        }
    }
    active = active && (active_surface || active_medium);
    commit();
    return active;
}

// Filter footprint helpers shared by the weight-film and delta_L kernels (imageblock.cpp:431-500)
DEV void lane_sample_pos(const DScene &sc, const DRenderParams &rp, uint32_t lane, float *spx, float *spy, int *px, int *py) {
    PCG32 rng = lane_rng_fresh(rp.seed_value, lane);
    lane_to_pixel(sc, rp, lane, px, py);
    float jx = rng.next(), jy = rng.next();
    *spx = (float) *px + jx; *spy = (float) *py + jy;
}

// Sum of reconstruction-filter weights per pixel over ALL lanes of the render (non-box filters).
__global__ void __launch_bounds__(LRT_BLOCK)
k_weight_film(DScene sc, DRenderParams rp, float *__restrict__ wfilm, uint64_t n_lanes) {
    uint64_t i = (uint64_t) blockIdx.x * LRT_BLOCK + threadIdx.x;
    if (i >= n_lanes) return;
    const DFilm &F = sc.film;
    float spx, spy; int px, py; lane_sample_pos(sc, rp, (uint32_t) i, &spx, &spy, &px, &py);
    int n = F.fn, count = F.fcount;
    int pix = (int) __builtin_floorf(spx) - n, piy = (int) __builtin_floorf(spy) - n;
    float relx = (float) pix + .5f - spx, rely = (float) piy + .5f - spy;
    for (int ys = 0; ys < count; ++ys) {
        int y = piy - F.crop_offset_y + ys;
        if (y < 0 || y >= F.height) continue;
        float wy = rfilter_eval(F, rely + (float) ys);
        for (int xs = 0; xs < count; ++xs) {
            int x = pix - F.crop_offset_x + xs;
            if (x < 0 || x >= F.width) continue;
            float w = wy * rfilter_eval(F, relx + (float) xs);
            if (w != 0.f) atomicAdd(wfilm + (size_t) y * F.width + x, w);
        }
    }
}

// delta_L of a lane: gradient of sum(image * grad_image) w.r.t. the lane's radiance through splat + develop
// (common.py:730-746).  Box filter: grad[pixel] / W[pixel].
DEV V3 lane_delta_L(const DScene &sc, const DRenderParams &rp, uint32_t lane, const float *__restrict__ grad_image, const float *__restrict__ wfilm) {
    const DFilm &F = sc.film;
    const int T = F.has_alpha ? 4 : 3;
    if (F.rfilter == LRT_RFILTER_BOX) {
        int px, py; lane_to_pixel(sc, rp, lane, &px, &py);
        size_t p = (size_t) (py - F.crop_offset_y) * F.width + (px - F.crop_offset_x);
        float w = (float) rp.spp;
        return V3(grad_image[p * T] / w, grad_image[p * T + 1] / w, grad_image[p * T + 2] / w);
    }
    float spx, spy; int px, py; lane_sample_pos(sc, rp, lane, &spx, &spy, &px, &py);
    int n = F.fn, count = F.fcount;
    int pix = (int) __builtin_floorf(spx) - n, piy = (int) __builtin_floorf(spy) - n;
    float relx = (float) pix + .5f - spx, rely = (float) piy + .5f - spy;
    V3 dL(0.f);
    for (int ys = 0; ys < count; ++ys) {
        int y = piy - F.crop_offset_y + ys;
        if (y < 0 || y >= F.height) continue;
        float wy = rfilter_eval(F, rely + (float) ys);
        for (int xs = 0; xs < count; ++xs) {
            int x = pix - F.crop_offset_x + xs;
            if (x < 0 || x >= F.width) continue;
            size_t p = (size_t) y * F.width + x;
            float w = wy * rfilter_eval(F, relx + (float) xs), wp = wfilm[p]; if (wp == 0.f) wp = 1.f;
            float f = w / wp;
            dL = dL + V3(grad_image[p * T] * f, grad_image[p * T + 1] * f, grad_image[p * T + 2] * f);
        }
    }
    return dL;
}

// ADJOINT == false: primal PRB pass; finished lanes store L into L_buf[slot] (slot = chunk-local lane index).
// ADJOINT == true : replay; finished lanes only retire, gradients are block-reduced and added to grads[7] (f64).
template <bool ADJOINT>
__global__ void __launch_bounds__(LRT_BLOCK)
k_iterate_prb(DScene sc, DRenderParams rp, DPathStreams qin, DPathStreams qout, const float4 *__restrict__ dl_in, float4 *__restrict__ dl_out,
              DCounters *__restrict__ cnt, uint32_t n_in, float4 *__restrict__ L_buf, double *__restrict__ grads,
              float *__restrict__ film, float *__restrict__ sample_out, uint64_t sample_base) {
    __shared__ int s_stack[LRT_STACK * LRT_BLOCK];
    __shared__ uint32_t s_wave_count[LRT_BLOCK / 64];
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_shadow;
    __shared__ float s_grad[7];
    const uint32_t tid = threadIdx.x, i = blockIdx.x * LRT_BLOCK + tid;
    const uint32_t wave = tid >> 6, lane_in_wave = tid & 63u;
    if (tid == 0) s_shadow = 0;
    if (tid < 7) s_grad[tid] = 0.f;
    bool alive = false;
    PathState s; float4 dl = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t n_shadow = 0;
    PrbGrads G; G.sigma_t[0] = G.sigma_t[1] = G.sigma_t[2] = G.albedo[0] = G.albedo[1] = G.albedo[2] = G.g = 0.f;
    if (i < n_in) {
        load_state(qin, i, s); dl = dl_in[i];
        PCG32 rng; rng.state = s.rng_state; rng.inc = lane_rng_inc(rp.seed_value, s.lane);
        const GlobalTracer tr{ sc, s_stack + tid };
        alive = prb_iteration<ADJOINT>(sc, rp, s, rng, tr, n_shadow, V3(dl.x, dl.y, dl.z), G);
        s.rng_state = rng.state;
        if (!alive && !ADJOINT) {
            if (L_buf) L_buf[f2u(dl.w)] = make_float4(s.res.x, s.res.y, s.res.z, (s.flags & PF_VALID) ? 1.f : 0.f);
            else finish_path(sc, rp, film, sample_out, sample_base, s.lane, s.res, (s.flags & PF_VALID) != 0);
        }
    }
    const unsigned long long m = __ballot(alive);
    const uint32_t wcount = (uint32_t) __popcll(m);
    const uint32_t wprefix = (uint32_t) __popcll(m & ((1ull << lane_in_wave) - 1ull));
    if (lane_in_wave == 0) s_wave_count[wave] = wcount;
    for (int off = 32; off > 0; off >>= 1) n_shadow += __shfl_down(n_shadow, off);
    auto wave_sum = [&](float v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off); return v; };
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f, g5 = 0.f, g6 = 0.f;
    if (ADJOINT) {
        g0 = wave_sum(G.sigma_t[0]); g1 = wave_sum(G.sigma_t[1]); g2 = wave_sum(G.sigma_t[2]);
        g3 = wave_sum(G.albedo[0]); g4 = wave_sum(G.albedo[1]); g5 = wave_sum(G.albedo[2]); g6 = wave_sum(G.g);
    }
    __syncthreads();
    if (lane_in_wave == 0 && n_shadow) atomicAdd(&s_shadow, n_shadow);
    if (ADJOINT && lane_in_wave == 0) {
        atomicAdd(&s_grad[0], g0); atomicAdd(&s_grad[1], g1); atomicAdd(&s_grad[2], g2); atomicAdd(&s_grad[3], g3);
        atomicAdd(&s_grad[4], g4); atomicAdd(&s_grad[5], g5); atomicAdd(&s_grad[6], g6);
    }
    if (tid == 0) {
        uint32_t total = 0;
        for (int w = 0; w < LRT_BLOCK / 64; ++w) total += s_wave_count[w];
        s_base = total ? atomicAdd(&cnt->n_out, total) : 0u;
    }
    __syncthreads();
    if (alive) {
        uint32_t slot = s_base + wprefix;
        for (uint32_t w = 0; w < wave; ++w) slot += s_wave_count[w];
        store_state(qout, slot, s); dl_out[slot] = dl;
    }
    if (tid == 0 && s_shadow) atomicAdd(&cnt->n_shadow, (unsigned long long) s_shadow);
    if (ADJOINT && tid < 7 && s_grad[tid] != 0.f) atomicAdd(&grads[tid], (double) s_grad[tid]);
}

// Ray generation for the PRB passes (common.py:231-309 + prbvolpath.py:113-137).
template <bool ADJOINT>
__global__ void __launch_bounds__(LRT_BLOCK)
k_raygen_prb(DScene sc, DRenderParams rp, DPathStreams q, float4 *__restrict__ dl_out, const uint32_t *__restrict__ pixel_list,
             uint64_t lane_base, uint32_t n, const float4 *__restrict__ L_buf, const float *__restrict__ grad_image, const float *__restrict__ wfilm) {
    uint32_t i = blockIdx.x * LRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint64_t j = lane_base + i;
    uint32_t lane;
    if (pixel_list) { uint32_t pj = (uint32_t) (j / rp.spp); lane = pixel_list[pj] * rp.spp + (uint32_t) (j - (uint64_t) pj * rp.spp); }
    else lane = (uint32_t) j;
    PCG32 rng = lane_rng_fresh(rp.seed_value, lane);
    int px, py; lane_to_pixel(sc, rp, lane, &px, &py);
    float jx = rng.next(), jy = rng.next();
    float spx = (float) px + jx, spy = (float) py + jy;
    Ray ray = camera_ray(sc, fma_(spx, sc.film.scale_x, sc.film.offset_x), fma_(spy, sc.film.scale_y, sc.film.offset_y));
    PathState s;
    s.o = ray.o; s.d = ray.d; s.maxt = ray.maxt; s.eta = 1.f; s.tp = V3(1.f); s.lp = V3(0.f); s.last_pdf = 1.f; s.lane = lane;
    uint32_t channel = min((uint32_t) (3.f * rng.next()), 2u);
    s.flags = PF_SPECULAR | (channel << PF_CHANNEL_SHIFT);      // valid_ray = false, specular_chain = true, medium = none
    s.rng_state = rng.state;
    V3 dL(0.f); s.res = V3(0.f);
    if (ADJOINT) { float4 l = L_buf[i]; s.res = V3(l.x, l.y, l.z); dL = lane_delta_L(sc, rp, lane, grad_image, wfilm); }
    store_state(q, i, s);
    dl_out[i] = make_float4(dL.x, dL.y, dL.z, u2f(i));
}

} // namespace lrt
