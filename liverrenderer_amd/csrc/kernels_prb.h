// Path Replay Backpropagation for homogeneous media: the primal estimator and the
// hand-derived adjoint of src/python/python/ad/integrators/prbvolpath.py:96-444
// (driver: src/python/python/ad/integrators/common.py:625-783), on the persistent render kernel of kernels.h.
//
// Differentiated parameters: sigma_t[3], albedo[3] (src/media/homogeneous.cpp:146-151) and the HG
// asymmetry g (src/phase/hg.cpp:60-62).  With detached sampling the local derivatives are closed forms:
//   free-flight weight  w_c = exp(-t sigma_c) / pdf_k [* sigma_c a_c at a real scatter]
//       d w_c / d sigma_c = w_c (-t [+ 1/sigma_c]),   d w_c / d a_c = w_c / a_c
//   NEE transmittance   tr_c = exp(-t_seg sigma_c):  d ln tr_c / d sigma_c = -t_seg
//   phase value         d ln hg / d g = -2g/(1-g^2) - 3 (g + c) / (1 + g^2 + 2 g c)
// and PRB multiplies them by delta_L * (radiance still to be collected).
//
// HET (a heterogeneous medium is attached to a shape: prbvolpath.py:84-91 `handle_null_scattering`): delta tracking with null
// collisions on the path (:178-196), ratio tracking on the emitter-sampling march (:404-415).  Only sigma_t(p) = scale * grid(p)
// carries a gradient there (the majorant is an opaque scalar made in parameters_changed(), src/media/heterogeneous.cpp):
//   real collision  d ln(sigma_s_c) / d scale = 1 / scale,   d ln(sigma_s_c) / d a_c = 1 / a_c
//   null collision  d ln(sigma_n_c) / d scale = -sigma_t(p) / (sigma_n_c scale)
// and d_sigma_t[c] of such a medium is channel c's share of d / d scale (include/liverrt.h).
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
DEV float het_null_dlog_dscale(const DMedium &M, float sigma_t, float sigma_n) { return sigma_n > 0.f ? -(sigma_t / sigma_n) / M.scale : 0.f; }

template <bool HET, typename SMP, typename TR>
DEV V3 prb_sample_emitter(SceneRef sc, SMP &rng, V3 ref_p, V3 ref_n, bool ref_is_surface, uint32_t ref_shape, V3 ref_geo_n,
                          int medium, uint32_t channel, DirSample *ds_out, const TR &tr, uint32_t &n_shadow, V3 *seg_sum, int grad_medium) {
    float sx, sy; rng.next2(sx, sy);
    DirSample ds; V3 emitter_val = sample_emitter_direction(sc, ref_p, sx, sy, &ds);
    *ds_out = ds;
    bool active = ds.pdf != 0.f;
    if (!active) { emitter_val = V3(0.f); medium = -1; }
    if (ref_is_surface) { const DShape sd = tab(sc.shapes, ref_shape, sc.one_shape); if (is_medium_transition(sd)) medium = target_medium(sd, ds.d, ref_geo_n); }
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
        if (needs_intersection) { n_shadow++; Hit h = tr.closest(ray); si = tr.surface(sc, ray, h); }
        needs_intersection = false;
        bool active_medium = active && medium >= 0, active_surface = active && !active_medium;
        V3 tr_multiplier(1.f);
        float seg_t = 0.f, scale_t = 0.f, mei_t = kInf; bool escaped_medium = false;
        V3 null_dlog(0.f);
        if (!active_medium) rng.skip(1);             // prbvolpath.py:396: the call runs for every lane in the march
        if (active_medium) {
            const DMedium M = tab(sc.media, medium);
            const bool graded = grad_medium < 0 || medium == grad_medium;   // only the differentiated medium's segments
            if (HET && M.het) {                      // ratio tracking (:409-415): a collision inside the segment is a null collision of the march
                MI mei = het_sample_interaction(M, tab(sc.het, medium), ray, rng.next());
                if (si.t < mei.t) mei.t = kInf;
                escaped_medium = !mei.valid(); active_medium = mei.valid();
                if (active_medium) {
                    ray.o = mei.p; si.t = si.t - mei.t; mei_t = mei.t;
                    tr_multiplier = mei.sigma_n / mei.combined;
                    if (graded) null_dlog = V3(het_null_dlog_dscale(M, mei.sigma_t.x, mei.sigma_n.x), het_null_dlog_dscale(M, mei.sigma_t.x, mei.sigma_n.y), het_null_dlog_dscale(M, mei.sigma_t.x, mei.sigma_n.z));
                }
            } else {                                 // homogeneous: straight to the next surface / the end of the segment (:403-407)
                (void) rng.next();
                float t = fmin_(remaining_dist, si.t);
                seg_t = fmin_(t, si.t) - 0.f;
                tr_multiplier = V3(m_exp(-seg_t * M.sigma_t[0]), m_exp(-seg_t * M.sigma_t[1]), m_exp(-seg_t * M.sigma_t[2]));
                scale_t = graded ? -seg_t * M.scale : 0.f;
                escaped_medium = true; active_medium = false;
            }
        }
        active_surface = (active_surface || escaped_medium) && si.valid && !active_medium;
        if (active_surface) tr_multiplier = tr_multiplier * bsdf_null_transmission(sc, tab(sc.shapes, si.shape, sc.one_shape).bsdf);
        if (escaped_medium && active_surface) {      // :425-427: active_adj = (surface | medium) & tr > 0
            if (tr_multiplier.x > 0.f) sum.x += scale_t;
            if (tr_multiplier.y > 0.f) sum.y += scale_t;
            if (tr_multiplier.z > 0.f) sum.z += scale_t;
        }
        if (HET && active_medium) {
            if (tr_multiplier.x > 0.f) sum.x += null_dlog.x;
            if (tr_multiplier.y > 0.f) sum.y += null_dlog.y;
            if (tr_multiplier.z > 0.f) sum.z += null_dlog.z;
        }
        transmittance = transmittance * tr_multiplier;
        if (active_surface) ray = spawn_ray(si.p, si.n, ray.d);
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        active = active && (active_medium || active_surface) && any_nonzero(transmittance);
        if (active) total_dist += (HET && active_medium) ? mei_t : si.t;
        if (active_surface) { const DShape sd = tab(sc.shapes, si.shape, sc.one_shape); if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, si.n); }
    }
    *seg_sum = sum;
    return emitter_val * transmittance;
}

// One trip of prbvolpath's loop (prbvolpath.py:139-349).  s.res holds L: accumulated radiance (primal) or the
// radiance still to be collected (adjoint).  Returns true when the path survives.
template <bool ADJOINT, bool HET, typename SMP, typename TR>
DEV bool prb_iteration(SceneRef sc, RpRef rp, PathState &s, SMP &rng, const TR &tr, uint32_t &n_shadow,
                       V3 delta_L, PrbGrads &G) {
    uint32_t depth = s.flags & PF_DEPTH_MASK;
    const bool proven_empty = (s.flags & PF_NOHIT) != 0;               // look-ahead of the previous trip, see below
    const bool needs_intersection = !(HET && (s.flags & PF_HAVE_SI));  // HET: a null collision keeps the surface interaction found earlier (record's hit stream)
    Hit hkeep; hkeep.t = s.hit.x; hkeep.u = s.hit.y; hkeep.v = s.hit.z; hkeep.prim = f2u(s.hit.w);
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
    bool escaped_medium = false, act_medium_scatter = false, act_null_scatter = false, in_medium_segment = false, het = false;
    MI mei; mei.t = kInf; mei.wi = -ray.d; mei.p = V3(0.f); mei.sigma_t = V3(0.f); mei.sigma_n = V3(0.f); mei.combined = V3(1.f);
    SI si; si.valid = false; si.t = kInf;
    V3 weight(1.f);
    float seg_t = 0.f, scatter_prob = 1.f;
    if (!active_medium) rng.skip(1);                  // prbvolpath.py:158
    if (active_medium) {
        const DMedium M = tab(sc.media, medium);
        het = HET && M.het;
        mei = het ? het_sample_interaction(M, tab(sc.het, medium), ray, rng.next()) : medium_sample_interaction(M, ray, rng.next(), channel);
        if (mei.valid() && !het) ray.maxt = mei.t;                                  // medium.is_homogeneous() only (:163)
        if (!needs_intersection) si = compute_si(sc, ray, hkeep);
        else if (!proven_empty) { hkeep = tr.closest(ray); si = tr.surface(sc, ray, hkeep); }
        if (si.t < mei.t) mei.t = kInf;
        seg_t = fmin_(mei.t, si.t) - mei.mint;
        V3 tr(m_exp(-seg_t * mei.combined.x), m_exp(-seg_t * mei.combined.y), m_exp(-seg_t * mei.combined.z));
        V3 pdf = (si.t < mei.t) ? tr : tr * mei.combined;
        float tr_pdf = idx3(pdf, channel);
        weight = (tr_pdf > 0.f) ? tr / tr_pdf : V3(0.f);
        escaped_medium = !mei.valid();
        active_medium = mei.valid();
        in_medium_segment = true;
    }
    if (HET) {                                        // :178-183: one more draw per trip in a scene that holds a heterogeneous medium
        if (!active_medium) rng.skip(1);
        else {
            scatter_prob = mean3(mei.sigma_t / mei.combined);
            act_null_scatter = rng.next() >= scatter_prob;
            if (act_null_scatter) weight = weight * (mei.sigma_n / (1.f - scatter_prob));
        }
    }
    if (active_medium && !act_null_scatter) { act_medium_scatter = true; depth += 1; s.lp = mei.p; }
    active = active && depth < max_depth;
    act_medium_scatter = act_medium_scatter && active;
    if (HET && act_null_scatter) { ray.o = mei.p; hkeep.t = si.t - mei.t; }         // :194-196 (si.t -= mei.t)
    if (act_medium_scatter) weight = weight * (HET ? mei.sigma_s / scatter_prob : mei.sigma_s);
    throughput = throughput * weight;
    const int gm = rp.grad_medium;
    if (ADJOINT && in_medium_segment && (gm < 0 || medium == gm)) {     // prbvolpath.py:199-204
        const DMedium M = tab(sc.media, medium);
        auto term = [&](float w, float l, float dl, float st, float al, float sn, float &gs, float &ga) {
            float Lo = l / fmax_(1e-8f, w);
            if (het) {                                // only the collision coefficient depends on `scale` (header comment)
                const float dlog = act_medium_scatter ? 1.f / M.scale : (act_null_scatter ? het_null_dlog_dscale(M, mei.sigma_t.x, sn) : 0.f);
                gs += dl * Lo * (w * dlog);
                if (act_medium_scatter) ga += dl * Lo * (w / al);
                return;
            }
            float dws = w * (-seg_t) + (act_medium_scatter ? w / st : 0.f);
            if (!(seg_t < kInf)) dws = 0.f;
            gs += dl * Lo * dws * M.scale;
            if (act_medium_scatter) ga += dl * Lo * (w / al);
        };
#ifdef LRT_EXPERIMENT
        if (rp.pad1 && s.lane == rp.pad1 - 1u) printf("  [dev] medium term: depth %u seg_t %.9g w %.9g %.9g %.9g L %.9g %.9g %.9g dl %.9g scatter %d\n", depth, seg_t, weight.x, weight.y, weight.z, L.x, L.y, L.z, delta_L.x, (int) act_medium_scatter);
#endif
        term(weight.x, L.x, delta_L.x, M.sigma_t[0], M.albedo[0], mei.sigma_n.x, G.sigma_t[0], G.albedo[0]);
        term(weight.y, L.y, delta_L.y, M.sigma_t[1], M.albedo[1], mei.sigma_n.y, G.sigma_t[1], G.albedo[1]);
        term(weight.z, L.z, delta_L.z, M.sigma_t[2], M.albedo[2], mei.sigma_n.z, G.sigma_t[2], G.albedo[2]);
    }
    // ---- surface interactions
    active_surface = active_surface || escaped_medium;
    bool intersect = active_surface && !escaped_medium;
    if (intersect) { Hit h = tr.closest(ray); si = tr.surface(sc, ray, h); }
    if (rp.hide_emitters && intersect && depth == 0 && si.valid && tab(sc.shapes, si.shape, sc.one_shape).emitter >= 0) {
        Ray r2 = spawn_ray(si.p, si.n, ray.d);
        bool a = true; Hit h; h.prim = 0xffffffffu; h.t = kInf; h.u = h.v = 0.f;
        while (a) {
            h = tr.closest(r2);
            a = h.prim != 0xffffffffu && tab(sc.shapes, sc.face_shape[h.prim], sc.one_shape).emitter >= 0;
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
#ifdef LRT_EXPERIMENT
            if (rp.pad1 && s.lane == rp.pad1 - 1u) printf("  [dev] %s emitter hit: depth %u contrib %.9g %.9g %.9g L after %.9g %.9g %.9g\n", ADJOINT ? "adjoint" : "primal", depth, contrib.x, contrib.y, contrib.z, L.x, L.y, L.z);
#endif
        }
    }
    active_surface = active_surface && si.valid;
    // ---- emitter sampling (prbvolpath.py:267-297)
    int b = active_surface ? tab(sc.shapes, si.shape, sc.one_shape).bsdf : 0;
    bool active_e_surface = active_surface && (tab(sc.bsdfs, b, sc.one_shape).flags & F_SMOOTH) && (depth + 1 < max_depth);
    bool sample_emitters = act_medium_scatter ? (tab(sc.media, medium).sample_emitters != 0) : false;
    if (act_medium_scatter) specular_chain = !sample_emitters;
    bool active_e_medium = act_medium_scatter && sample_emitters;
    if (!(active_e_surface || active_e_medium)) rng.skip(1);           // prbvolpath.py:365
    if (active_e_surface || active_e_medium) {
        DirSample ds; V3 seg_sum;
        V3 rp_ = active_e_medium ? mei.p : si.p, rn = active_e_medium ? V3(0.f) : si.n;
        V3 emitted = prb_sample_emitter<HET>(sc, rng, rp_, rn, active_e_surface, active_e_surface ? si.shape : 0u, si.n, medium, channel, &ds, tr, n_shadow, &seg_sum, gm);
        V3 nee_weight; float nee_pdf;
        if (active_e_surface) { V3 wo = si.sh.to_local(ds.d); nee_weight = bsdf_eval(sc, b, si, wo); nee_pdf = bsdf_pdf(sc, b, si, wo); }
        else { float pv = phase_eval(tab(sc.media, medium), mei.wi, ds.d); nee_weight = V3(pv); nee_pdf = pv; }
        V3 contrib = throughput * nee_weight * mis_weight(ds.pdf, ds.delta ? 0.f : nee_pdf) * emitted;
        L = ADJOINT ? L - contrib : L + contrib;
#ifdef LRT_EXPERIMENT
        if (rp.pad1 && s.lane == rp.pad1 - 1u) printf("  [dev] %s nee: depth %u surface %d contrib %.9g %.9g %.9g L after %.9g %.9g %.9g seg_sum %.9g %.9g %.9g\n", ADJOINT ? "adjoint" : "primal", depth, (int) active_e_surface, contrib.x, contrib.y, contrib.z, L.x, L.y, L.z, seg_sum.x, seg_sum.y, seg_sum.z);
#endif
        if (ADJOINT) {
            G.sigma_t[0] += delta_L.x * contrib.x * seg_sum.x; G.sigma_t[1] += delta_L.y * contrib.y * seg_sum.y; G.sigma_t[2] += delta_L.z * contrib.z * seg_sum.z;
            if (active_e_medium && tab(sc.media, medium).phase == LRT_PHASE_HG && (gm < 0 || medium == gm))
                G.g += (delta_L.x * contrib.x + delta_L.y * contrib.y + delta_L.z * contrib.z) * hg_dlog_dg(tab(sc.media, medium).g, dot(ds.d, mei.wi));
        }
    }
    // ---- phase function sampling (prbvolpath.py:299-317)
    if (!act_medium_scatter) rng.skip(2);             // prbvolpath.py:294-295
    if (act_medium_scatter) {
        valid_ray = true;
        const DMedium M = tab(sc.media, medium);
        (void) rng.next();
        float s2x, s2y; rng.next2(s2x, s2y);
        V3 wo; float phase_pdf; phase_sample(M, mei.wi, s2x, s2y, &wo, &phase_pdf);
        act_medium_scatter = phase_pdf > 0.f;
        if (act_medium_scatter) {
            if (ADJOINT && M.phase == LRT_PHASE_HG && (gm < 0 || medium == gm)) {
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
    if (!active_surface) rng.skip(2);                 // prbvolpath.py:317-318
    if (active_surface) {
        const DShape sd = tab(sc.shapes, si.shape, sc.one_shape);
        float s1 = rng.next(), s2x, s2y; rng.next2(s2x, s2y);
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
    // Look-ahead (as in volpath_iteration): replay the next trip's Russian-roulette and free-flight draws on a copy of the
    // sampler; when the distance field proves that the segment reaches no surface, the path is queued apart and its next
    // trip runs no ray query.  Exact: same draws, same functions, conservative proof.
    uint32_t nohit = 0;
    if (active && medium >= 0 && sc.grid.enabled && !(HET && (act_null_scatter || tab(sc.media, medium).het))) {   // (a heterogeneous medium does not shorten the ray: its query is always the full one)
        SMP pk = rng;
        bool a2 = any_nonzero(throughput);
        float q2 = fmin_(max3(throughput) * sqr(eta), 0.99f);
        if (a2) { float u = pk.next(); a2 = (u < q2) || !(depth > (uint32_t) rp.rr_depth); }
        if (a2) {
            const DMedium M = tab(sc.media, medium);
            MI m2 = medium_sample_interaction(M, ray, pk.next(), channel);
            if (m2.valid() && segment_free_of_surfaces(sc.grid, ray.o, ray.d, m2.t)) nohit = PF_NOHIT;
        }
    }
    commit();
    s.flags |= nohit;
    if (HET && act_null_scatter && active) { s.flags |= PF_HAVE_SI; s.hit = make_float4(hkeep.t, hkeep.u, hkeep.v, u2f(hkeep.prim)); }
    return active;
}

// Filter footprint helpers shared by the weight-film and delta_L kernels (imageblock.cpp:431-500)
DEV void lane_sample_pos(SceneRef sc, RpRef rp, uint32_t lane, float *spx, float *spy, int *px, int *py) {
    lane_to_pixel(sc, rp, lane, px, py);
    float jx, jy; lane_jitter(rp, lane, lane_local_index(rp, lane), jx, jy);
    *spx = (float) *px + jx; *spy = (float) *py + jy;
}

// delta_L of a lane: gradient of sum(image * grad_image) w.r.t. the lane's radiance through splat + develop
// (common.py:730-746).  Box filter: grad[pixel] / W[pixel].
DEV V3 lane_delta_L(SceneRef sc, RpRef rp, uint32_t lane, const float *__restrict__ grad_image, const float *__restrict__ wfilm) {
    FilmRef F = sc.film;
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

// PRB passes on the persistent render kernel of kernels.h (same rounds, pools, tiles, queue regions and look-ahead;
// one extra float4 stream carries delta_L and the lane's slot).
// ADJOINT == false: primal pass; finished lanes store L into L_buf[slot] (slot = index of the lane in this launch), or
//                   splat into the film / sample_out when L_buf is null (lrt_render with integrator prbvolpath).
// ADJOINT == true : replay; finished lanes only retire; the parameter gradients of a tile are summed inside the wave,
//                   accumulated per workgroup in f64 (LDS) and added to grads[7] once at the end.
// 4 waves per SIMD for every variant (128 VGPRs): one 1024-thread workgroup per CU, or four 256-thread ones
// HET: the scene holds a heterogeneous medium: null collisions (prb_iteration<.., true>), 120-B records (+ the kept surface hit)
template <bool ADJOINT, int BLOCK, bool LDS_BVH, bool LD, bool HET = false>
__global__ void __launch_bounds__(BLOCK, 4)
k_render_prb(ScenePtr scp, LaunchPtr lp) {
    constexpr int MODE = HET ? 2 : 0;
    SceneRef sc = *scp;
    const LRT_CONST DLaunch &A = *lp;
    RpRef rp = A.rp;
    const LRT_CONST DLdsInfo &li = A.li;
    const uint32_t P = A.P;
    float4 *__restrict__ L_buf = A.L_buf; const float *__restrict__ grad_image = A.grad_image; const float *__restrict__ wfilm = A.wfilm;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t s_in[3], s_out[3], s_ticket, s_fresh;
    __shared__ unsigned long long s_fresh_base;
    __shared__ double s_grad[7];
    const uint32_t tid = threadIdx.x, lane_in_wave = tid & 63u;
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
    uint32_t parity = 0;                                      // queue the round reads: parity ? q1 : q0 (scalar loads at the point of use)
    if (tid == 0) { s_in[0] = s_in[1] = s_in[2] = 0; }
    if (tid < 7) s_grad[tid] = 0.0;
    bool lanes_left = true;                                   // thread 0
    uint32_t n_shadow = 0, n_trips = 0, n_loaded = 0;
    for (;;) {
        if (tid == 0) {
            const uint32_t want = P - (s_in[0] + s_in[1] + s_in[2]);
            uint32_t got = 0; unsigned long long base = 0;
            if (want && lanes_left) {
                base = atomicAdd(&A.cnt->next_lane, (unsigned long long) want);
                if (base < rp.n_lanes) got = (uint32_t) (rp.n_lanes - base < (unsigned long long) want ? rp.n_lanes - base : (unsigned long long) want);
                lanes_left = base + want < rp.n_lanes;
            }
            s_fresh = got; s_fresh_base = base; s_ticket = 0; s_out[0] = s_out[1] = s_out[2] = 0;
        }
        __syncthreads();
        // queue regions as in k_render: A [0, n_a) proven-free in-medium paths, C [P, P + n_c) in-medium paths that need their
        // ray query, B 2P-1-j paths outside media
        const uint32_t n_a = s_in[0], n_c = s_in[1], n_s = s_in[2], fresh = s_fresh;
        const unsigned long long fresh_base = s_fresh_base;
        if (n_a + n_c + n_s + fresh == 0) break;
        const uint32_t ta = (n_a + 63u) >> 6, tc = (n_c + 63u) >> 6, ts = (n_s + 63u) >> 6, tf = (fresh + 63u) >> 6, tm = ta + tc;
        for (;;) {
            uint32_t t = 0;
            if (lane_in_wave == 0) t = atomicAdd(&s_ticket, 1u);
            t = (uint32_t) __builtin_amdgcn_readfirstlane((int) t);
            if (t >= tm + ts + tf) break;
            bool had_path = false, alive = false;
            PathState s; s.flags = 0; s.lane = 0; s.res = V3(0.f);
            float4 dl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < tm + ts) {
                uint32_t i;
                if (t < ta) { i = (t << 6) + lane_in_wave; had_path = i < n_a; }
                else if (t < tm) { i = ((t - ta) << 6) + lane_in_wave; had_path = i < n_c; i += P; }
                else { i = ((t - tm) << 6) + lane_in_wave; had_path = i < n_s; i = 2u * P - 1u - i; }
                if (had_path) { load_state<MODE>(parity ? A.q1 : A.q0, pool + i, s); dl = (parity ? A.dl1 : A.dl0)[pool + i]; n_loaded += 1; }
            } else {
                const uint32_t i = ((t - tm - ts) << 6) + lane_in_wave;
                had_path = i < fresh;
                if (had_path) {                                // common.py:231-309 + prbvolpath.py:113-137
                    const unsigned long long slot = fresh_base + i;
                    s = generate_camera_path<LD>(sc, rp, A.pixel_list, A.lane_begin + slot);
                    s.flags = PF_SPECULAR | (s.flags & (3u << PF_CHANNEL_SHIFT));      // valid_ray = false, specular_chain = true, medium = none
                    V3 dL(0.f);
                    if (ADJOINT) { float4 l = L_buf[slot]; s.res = V3(l.x, l.y, l.z); dL = lane_delta_L(sc, rp, s.lane, grad_image, wfilm); }
                    dl = make_float4(dL.x, dL.y, dL.z, u2f((uint32_t) slot));
                }
            }
            PrbGrads G; G.sigma_t[0] = G.sigma_t[1] = G.sigma_t[2] = G.albedo[0] = G.albedo[1] = G.albedo[2] = G.g = 0.f;
            if (had_path) {
                SamplerT<LD> rng = lane_rng_resume<LD>(rp, s.lane, s.rng_state);
                alive = LDS_BVH ? prb_iteration<ADJOINT, HET>(sc, rp, s, rng, tr_lds, n_shadow, V3(dl.x, dl.y, dl.z), G)
                                : prb_iteration<ADJOINT, HET>(sc, rp, s, rng, tr_glb, n_shadow, V3(dl.x, dl.y, dl.z), G);
                s.rng_state = rng.state;
                n_trips += 1;
            }
            if (!ADJOINT) {
                if (L_buf) { if (had_path && !alive) L_buf[f2u(dl.w)] = make_float4(s.res.x, s.res.y, s.res.z, (s.flags & PF_VALID) ? 1.f : 0.f); }
                else finish_paths_wave(sc, rp, A.film, A.sample_out, A.sample_base, had_path && !alive, s.lane, s.res, (s.flags & PF_VALID) != 0);
            } else {
                float g[7] = { G.sigma_t[0], G.sigma_t[1], G.sigma_t[2], G.albedo[0], G.albedo[1], G.albedo[2], G.g };
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const float v = wave_sum(g[k]);
                    if (lane_in_wave == 0 && v != 0.f) atomicAdd(&s_grad[k], (double) v);
                }
            }
            // compaction into the three regions
            const int region = !(s.flags & PF_MEDIUM_MASK) ? 2 : ((s.flags & PF_NOHIT) ? 0 : 1);
            const unsigned long long m0 = __ballot(alive && region == 0), m1 = __ballot(alive && region == 1), m2 = __ballot(alive && region == 2);
            uint32_t base = 0;
            if (lane_in_wave < 3) { const uint32_t c = (uint32_t) __popcll(lane_in_wave == 0 ? m0 : (lane_in_wave == 1 ? m1 : m2)); if (c) base = atomicAdd(&s_out[lane_in_wave], c); }
            const uint32_t b0 = (uint32_t) __builtin_amdgcn_readlane((int) base, 0), b1 = (uint32_t) __builtin_amdgcn_readlane((int) base, 1), b2 = (uint32_t) __builtin_amdgcn_readlane((int) base, 2);
            const uint32_t b = region == 0 ? b0 : (region == 1 ? b1 : b2);          // (v_readlane, not a shuffle through LDS: see retire_and_compact_wave)
            if (alive) {
                const uint32_t slot = b + (uint32_t) __popcll((region == 0 ? m0 : (region == 1 ? m1 : m2)) & ((1ull << lane_in_wave) - 1ull));
                const uint32_t rec = region == 0 ? slot : (region == 1 ? P + slot : 2u * P - 1u - slot);
                store_state<MODE>(parity ? A.q0 : A.q1, pool + rec, s); (parity ? A.dl0 : A.dl1)[pool + rec] = dl;
            }
        }
        __syncthreads();
        if (tid == 0) { s_in[0] = s_out[0]; s_in[1] = s_out[1]; s_in[2] = s_out[2]; }
        parity ^= 1u;
    }
    if (ADJOINT && tid < 7 && s_grad[tid] != 0.0) atomicAdd(&A.grads[tid], s_grad[tid]);
    for (int off = 32; off > 0; off >>= 1) {
        n_shadow += __shfl_down(n_shadow, off); n_trips += __shfl_down(n_trips, off); n_loaded += __shfl_down(n_loaded, off);
    }
    if (lane_in_wave == 0) {
        if (n_shadow) atomicAdd(&A.cnt->n_shadow, (unsigned long long) n_shadow);
        if (n_trips) atomicAdd(&A.cnt->n_iter, (unsigned long long) n_trips);
        if (n_loaded) atomicAdd(&A.cnt->n_records, (unsigned long long) n_loaded);
    }
}

} // namespace lrt
