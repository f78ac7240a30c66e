// `volpathmis` (src/integrators/volpathmis.cpp:127-699: Miller et al. 2019, MIS over free-flight / null-collision / spectral
// channels) on the persistent render kernel.  One trip of its while_loop per tile visit; the path record carries the two
// weight sets p_over_f and p_over_f_nee (3 x 3 each with use_spectral_mis, the default) instead of a throughput, plus the
// surface hit a null collision keeps (168 B records).  Included by kernels.h.
#pragma once

namespace lrt {

template <bool SMIS> struct MisW { float w[3][3]; };

// volpathmis.cpp:631-652 update_weights
template <bool SMIS> DEV void mis_update(MisW<SMIS> &W, V3 p, V3 f, uint32_t channel, bool active) {
    if (!active) return;
    const float pp[3] = { p.x, p.y, p.z }, ff[3] = { f.x, f.y, f.z };
    if (SMIS) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float ratio = pp[j] / ff[i];
                if (!finite_(ratio)) ratio = 0.f;
                ratio *= W.w[i][j];
                W.w[i][j] = (ratio != ratio) ? 0.f : ratio;
            }
        }
    } else {
        const float pdf = idx3(p, channel);
#pragma unroll
        for (int j = 0; j < 3; ++j) { float ratio = W.w[0][j] * (pdf / ff[j]); W.w[0][j] = finite_(ratio) ? ratio : 0.f; }
    }
}
// :654-668
template <bool SMIS> DEV V3 mis_weight1(const MisW<SMIS> &W) {
    if (SMIS) {
        float r[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { float sum = (W.w[i][0] + W.w[i][1]) + W.w[i][2]; r[i] = sum == 0.f ? 0.f : 3.f / sum; }
        return V3(r[0], r[1], r[2]);
    }
    bool invalid = fmin_(fmin_(__builtin_fabsf(W.w[0][0]), __builtin_fabsf(W.w[0][1])), __builtin_fabsf(W.w[0][2])) == 0.f;
    return invalid ? V3(0.f) : V3(1.f / W.w[0][0], 1.f / W.w[0][1], 1.f / W.w[0][2]);
}
// :671-685
template <bool SMIS> DEV V3 mis_weight2(const MisW<SMIS> &A, const MisW<SMIS> &B) {
    if (SMIS) {
        float r[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { float sum = ((A.w[i][0] + B.w[i][0]) + (A.w[i][1] + B.w[i][1])) + (A.w[i][2] + B.w[i][2]); r[i] = sum == 0.f ? 0.f : 3.f / sum; }
        return V3(r[0], r[1], r[2]);
    }
    float s0 = A.w[0][0] + B.w[0][0], s1 = A.w[0][1] + B.w[0][1], s2 = A.w[0][2] + B.w[0][2];
    bool invalid = fmin_(fmin_(__builtin_fabsf(s0), __builtin_fabsf(s1)), __builtin_fabsf(s2)) == 0.f;
    return invalid ? V3(0.f) : V3(1.f / s0, 1.f / s1, 1.f / s2);
}

DEV MI any_medium_sample(SceneRef sc, int medium, const DMedium &M, const Ray &ray, float sample, uint32_t channel) {
    return M.het ? het_sample_interaction(M, tab(sc.het, medium), ray, sample) : medium_sample_interaction(M, ray, sample, channel);
}

// volpathmis.cpp:449-629 sample_emitter
template <bool SMIS, typename SMP, typename TR>
DEV V3 mis_sample_emitter(SceneRef sc, SMP &rng, V3 ref_p, V3 ref_n, bool ref_is_surface, uint32_t ref_shape, int medium, const MisW<SMIS> &p_over_f,
                          uint32_t channel, DirSample *ds_out, MisW<SMIS> &nee, MisW<SMIS> &uni, const TR &tr, uint32_t &n_shadow) {
    nee = p_over_f; uni = p_over_f;
    float sx, sy; rng.next2(sx, sy);
    DirSample ds; V3 w = sample_emitter_direction(sc, ref_p, sx, sy, &ds);
    V3 emitter_val = w * ds.pdf;
    if (ds.pdf == 0.f) emitter_val = V3(0.f);
    bool active = ds.pdf != 0.f;
    mis_update(nee, V3(ds.pdf), V3(1.f), channel, active);
    *ds_out = ds;
    if (!active) return emitter_val;
    Ray ray = spawn_ray_to(ref_p, ref_n, ds.p);
    float max_dist = ray.maxt;
    if (ref_is_surface) { const DShape sd = tab(sc.shapes, ref_shape, sc.one_shape); if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, ref_n); }
    float total_dist = 0.f;
    SI si; si.valid = false; si.t = kInf; si.shape = 0; si.p = V3(0.f); si.n = V3(0.f);
    bool needs_intersection = true;
    while (active) {
        float remaining_dist = max_dist - total_dist;
        ray.maxt = remaining_dist;
        if (!(remaining_dist > 0.f)) { rng.skip(1); break; }
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (!active_medium) rng.skip(1);
        if (active_medium) {
            const DMedium M = tab(sc.media, medium);
            MI mei = any_medium_sample(sc, medium, M, ray, rng.next(), channel);
            if (mei.valid() && !M.het) ray.maxt = fmin_(mei.t, remaining_dist);
            if (needs_intersection) { n_shadow++; Hit h = tr.closest(ray); si = tr.surface(sc, ray, h); }
            if (si.t < mei.t) mei.t = kInf;
            needs_intersection = false;
            bool is_spectral = M.has_spectral_extinction != 0, not_spectral = !is_spectral;
            if (is_spectral) {
                float t = fmin_(remaining_dist, fmin_(mei.t, si.t)) - mei.mint;
                V3 trm(m_exp(-t * mei.combined.x), m_exp(-t * mei.combined.y), m_exp(-t * mei.combined.z));
                V3 ffp = (si.t < mei.t || mei.t > remaining_dist) ? trm : trm * mei.combined;
                mis_update(nee, ffp, trm, channel, true); mis_update(uni, ffp, trm, channel, true);
            }
            if ((mei.t > remaining_dist) && mei.valid()) total_dist = ds.dist;
            if (mei.t > remaining_dist) mei.t = kInf;
            escaped_medium = !mei.valid();
            active_medium = mei.valid();
            is_spectral = is_spectral && active_medium; not_spectral = not_spectral && active_medium;
            if (active_medium) {
                total_dist += mei.t;
                ray.o = mei.p;
                si.t = si.t - mei.t;
                if (is_spectral) { mis_update(nee, V3(1.f), mei.sigma_n, channel, true); mis_update(uni, V3(mean3(mei.sigma_n / mei.combined)), mei.sigma_n, channel, true); }
                if (not_spectral) { mis_update(nee, V3(1.f), mei.sigma_n / mei.combined, channel, true); mis_update(uni, mei.sigma_n, mei.sigma_n, channel, true); }
            }
        }
        bool intersect = active_surface && needs_intersection;
        if (intersect) { n_shadow++; Hit h = tr.closest(ray); si = tr.surface(sc, ray, h); }
        active_surface = active_surface || escaped_medium;
        if (active_surface) total_dist += si.t;
        active_surface = active_surface && si.valid && !active_medium;
        if (active_surface) {
            V3 bv(bsdf_null_transmission(sc, tab(sc.shapes, si.shape, sc.one_shape).bsdf));
            mis_update(nee, V3(1.f), bv, channel, true); mis_update(uni, V3(1.f), bv, channel, true);
            ray = spawn_ray(si.p, si.n, ray.d);
        }
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        if (SMIS) active = (active_medium || active_surface) && any_nonzero(mis_weight1(uni));
        else active = (active_medium || active_surface) && (uni.w[0][0] != 0.f || uni.w[0][1] != 0.f || uni.w[0][2] != 0.f || nee.w[0][0] != 0.f || nee.w[0][1] != 0.f || nee.w[0][2] != 0.f);
        if (active_surface) { const DShape sd = tab(sc.shapes, si.shape, sc.one_shape); if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, si.n); }
    }
    return emitter_val;
}

// One trip of volpathmis's while_loop (volpathmis.cpp:212-443).  Flag bits: PF_HAVE_SI = !needs_intersection (the record's hit
// stream holds `si`), PF_LAST_NULL = last_event_was_null.
template <bool SMIS, typename SMP, typename TR>
DEV bool volpathmis_iteration(SceneRef sc, RpRef rp, PathState &s, SMP &rng, const TR &tr, uint32_t &n_shadow) {
    uint32_t depth = s.flags & PF_DEPTH_MASK;
    int medium = (int) ((s.flags & PF_MEDIUM_MASK) >> PF_MEDIUM_SHIFT) - 1;
    const uint32_t channel = (s.flags >> PF_CHANNEL_SHIFT) & 3u;
    bool specular_chain = (s.flags & PF_SPECULAR) != 0, valid_ray = (s.flags & PF_VALID) != 0;
    bool needs_intersection = !(s.flags & PF_HAVE_SI), last_event_was_null = (s.flags & PF_LAST_NULL) != 0;
    const uint32_t max_depth = (uint32_t) rp.max_depth;
    V3 result = s.res;
    float eta = s.eta;
    Ray ray; ray.o = s.o; ray.d = s.d; ray.maxt = s.maxt;
    MisW<SMIS> p_over_f, p_over_f_nee;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { p_over_f.w[i][j] = s.W[0][i][j]; p_over_f_nee.w[i][j] = s.W[1][i][j]; }
    }
    Hit hkeep; hkeep.t = s.hit.x; hkeep.u = s.hit.y; hkeep.v = s.hit.z; hkeep.prim = f2u(s.hit.w);
    auto commit = [&]() {
        s.res = result; s.eta = eta; s.o = ray.o; s.d = ray.d; s.maxt = ray.maxt;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { s.W[0][i][j] = p_over_f.w[i][j]; s.W[1][i][j] = p_over_f_nee.w[i][j]; }
        }
        s.hit = make_float4(hkeep.t, hkeep.u, hkeep.v, u2f(hkeep.prim));
        s.flags = (depth & PF_DEPTH_MASK) | ((uint32_t) (medium + 1) << PF_MEDIUM_SHIFT) | (channel << PF_CHANNEL_SHIFT) |
                  (specular_chain ? PF_SPECULAR : 0u) | (valid_ray ? PF_VALID : 0u) | (needs_intersection ? 0u : PF_HAVE_SI) | (last_event_was_null ? PF_LAST_NULL : 0u);
    };
    // ---- termination (:233-247)
    const V3 mis_throughput = mis_weight1(p_over_f);
    const float q = fmin_(max3(mis_throughput) * sqr(eta), .95f);
    const bool perform_rr = !last_event_was_null && depth > (uint32_t) rp.rr_depth;
    const float u_rr = rng.next();
    bool active = !(u_rr >= q && perform_rr);
    mis_update(p_over_f, V3(q), V3(1.f), channel, perform_rr);
    last_event_was_null = false;
    active = active && depth < max_depth;
    active = active && any_nonzero(mis_weight1(p_over_f));
    if (!active) { commit(); return false; }

    bool active_medium = medium >= 0, active_surface = !active_medium;
    bool act_null_scatter = false, act_medium_scatter = false, escaped_medium = false, is_spectral = false, not_spectral = false;
    MI mei; mei.t = kInf;
    SI si; si.valid = false; si.t = kInf;
    if (!needs_intersection) si = compute_si(sc, ray, hkeep);              // a null collision kept this interaction
    if (!active_medium) rng.skip(2);
    if (active_medium) {
        const DMedium M = tab(sc.media, medium);
        is_spectral = M.has_spectral_extinction != 0; not_spectral = !is_spectral;
        mei = any_medium_sample(sc, medium, M, ray, rng.next(), channel);
        if (mei.valid() && !M.het) ray.maxt = mei.t;
        if (needs_intersection) { hkeep = tr.closest(ray); si = tr.surface(sc, ray, hkeep); }
        needs_intersection = false;
        if (si.t < mei.t) mei.t = kInf;
        if (is_spectral) {
            float t = fmin_(mei.t, si.t) - mei.mint;
            V3 trm(m_exp(-t * mei.combined.x), m_exp(-t * mei.combined.y), m_exp(-t * mei.combined.z));
            V3 pdf = (si.t < mei.t) ? trm : trm * mei.combined;
            mis_update(p_over_f, pdf, trm, channel, true); mis_update(p_over_f_nee, pdf, trm, channel, true);
        }
        escaped_medium = !mei.valid();
        active_medium = mei.valid();
        is_spectral = is_spectral && active_medium; not_spectral = not_spectral && active_medium;
        if (!active_medium) rng.skip(1);
    }
    if (!active_medium) rng.skip(3);
    if (active_medium) {
        const DMedium M = tab(sc.media, medium);
        const float null_scatter_prob = mean3(mei.sigma_n / mei.combined);
        act_null_scatter = rng.next() < null_scatter_prob;
        act_medium_scatter = !act_null_scatter;
        last_event_was_null = act_null_scatter;
        if (act_medium_scatter) { depth += 1; s.lp = mei.p; }
        const bool sample_emitters = M.sample_emitters != 0;
        active = active && depth < max_depth;
        act_medium_scatter = act_medium_scatter && active;
        if (act_medium_scatter) specular_chain = !sample_emitters;
        if (act_null_scatter) {
            if (is_spectral) { mis_update(p_over_f, V3(null_scatter_prob), mei.sigma_n, channel, true); mis_update(p_over_f_nee, V3(1.f), mei.sigma_n, channel, true); }
            if (not_spectral) { mis_update(p_over_f, mei.sigma_n, mei.sigma_n, channel, true); mis_update(p_over_f_nee, V3(1.f), V3(null_scatter_prob), channel, true); }
            ray.o = mei.p; hkeep.t = si.t - mei.t; si.t = hkeep.t;
        }
        if (!act_medium_scatter) rng.skip(3);
        if (act_medium_scatter) {
            if (is_spectral) mis_update(p_over_f, V3(1.f - null_scatter_prob), mei.sigma_s, channel, true);
            if (not_spectral) mis_update(p_over_f, mei.sigma_t, mei.sigma_s, channel, true);
            valid_ray = true;
            if (!sample_emitters) rng.skip(1);
            if (sample_emitters) {
                DirSample ds; MisW<SMIS> nee_end, uni_end;
                V3 emitted = mis_sample_emitter<SMIS>(sc, rng, mei.p, V3(0.f), false, 0u, medium, p_over_f, channel, &ds, nee_end, uni_end, tr, n_shadow);
                const float pv = phase_eval(M, mei.wi, ds.d);
                mis_update(nee_end, V3(1.f), V3(pv), channel, true);
                mis_update(uni_end, V3(ds.delta ? 0.f : pv), V3(pv), channel, true);
                result = result + mis_weight2(nee_end, uni_end) * emitted;
            }
            p_over_f_nee = p_over_f;
            (void) rng.next();
            float s2x, s2y; rng.next2(s2x, s2y);
            V3 wo; float phase_pdf; phase_sample(M, mei.wi, s2x, s2y, &wo, &phase_pdf);
            ray = spawn_ray(mei.p, V3(0.f), wo);
            needs_intersection = true;
            mis_update(p_over_f, V3(phase_pdf), V3(1.f * phase_pdf), channel, true);
            mis_update(p_over_f_nee, V3(1.f), V3(1.f * phase_pdf), channel, true);
        }
    }
    // ---- surface interactions
    active_surface = active_surface || escaped_medium;
    const bool intersect = active_surface && needs_intersection;
    if (intersect) { hkeep = tr.closest(ray); si = tr.surface(sc, ray, hkeep); }
    if (active_surface) {
        if (rp.hide_emitters && depth == 0 && intersect) {
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
        if (active_e) {                                                 // :384-389: the weight update is masked by active_e alone
            float emitter_pdf = pdf_emitter_direction(sc, s.lp, si, emitter);
            mis_update(p_over_f_nee, V3(emitter_pdf), V3(1.f), channel, true);
            V3 emitted = emitter_eval(sc, emitter, si);
            V3 contrib = count_direct ? mis_weight1(p_over_f) * emitted : mis_weight2(p_over_f, p_over_f_nee) * emitted;
            result = result + contrib;
        }
    }
    active_surface = active_surface && si.valid;
    if (!active_surface) rng.skip(3);
    if (active_surface) {
        const DShape sd = tab(sc.shapes, si.shape, sc.one_shape);
        int b = sd.bsdf;
        int flags = tab(sc.bsdfs, b, sc.one_shape).flags;
        bool active_e = (flags & F_SMOOTH) && (depth + 1 < max_depth);
        if (!active_e) rng.skip(1);
        if (active_e) {
            DirSample ds; MisW<SMIS> nee_end, uni_end;
            V3 emitted = mis_sample_emitter<SMIS>(sc, rng, si.p, si.n, true, si.shape, medium, p_over_f, channel, &ds, nee_end, uni_end, tr, n_shadow);
            V3 wo = si.sh.to_local(ds.d);
            V3 bsdf_val = bsdf_eval(sc, b, si, wo);
            float bpdf = bsdf_pdf(sc, b, si, wo);
            mis_update(nee_end, V3(1.f), bsdf_val, channel, true);
            mis_update(uni_end, V3(ds.delta ? 0.f : bpdf), bsdf_val, channel, true);
            result = result + mis_weight2(nee_end, uni_end) * emitted;
        }
        float s1 = rng.next(), s2x, s2y; rng.next2(s2x, s2y);
        const BSDFSample bs = bsdf_sample(sc, b, si, s1, s2x, s2y);
        const bool invalid_bsdf_sample = bs.pdf == 0.f;
        active_surface = bs.pdf > 0.f;
        if (active_surface) {
            eta *= bs.eta;
            ray = spawn_ray(si.p, si.n, si.sh.to_world(bs.wo));
            needs_intersection = true;
        }
        const bool non_null = active_surface && !(bs.type & F_NULL);
        valid_ray = valid_ray || non_null || invalid_bsdf_sample;
        specular_chain = specular_chain || (non_null && (bs.type & F_DELTA));
        specular_chain = specular_chain && !(active_surface && (bs.type & F_SMOOTH));
        if (non_null) { depth += 1; s.lp = si.p; p_over_f_nee = p_over_f; }
        const V3 f = bs.weight * bs.pdf;
        mis_update(p_over_f, V3(bs.pdf), f, channel, active_surface);
        mis_update(p_over_f_nee, V3(1.f), f, channel, non_null);
        if (active_surface && is_medium_transition(sd)) medium = target_medium(sd, ray.d, si.n);
    }
    active = active && (active_surface || active_medium);
    commit();
    return active;
}

} // namespace lrt
