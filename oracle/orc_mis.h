/*
 * orc_mis.h -- `volpathmis` (src/integrators/volpathmis.cpp:127-699, Miller et al. 2019 spectral / null-collision MIS) in the
 * CPU oracle.  TEST INFRASTRUCTURE ONLY (see orc.h).  Included by orc_render.cpp inside namespace orc.  JIT-variant lane
 * semantics as everywhere in the oracle.  SMIS: the plugin's `use_spectral_mis` (default true: a 3 x 3 weight matrix per path).
 */
template <bool SMIS> struct MisW {
    float w[3][3];                                  /* SMIS: w[i] = p_over_f[i], a 3-vector; else only w[0] is used */
    static MisW ones() { MisW r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.w[i][j] = 1.f; return r; }
};

/* volpathmis.cpp:631-652 update_weights */
template <bool SMIS> static inline void mis_update(MisW<SMIS> &W, V3 p, V3 f, uint32_t channel, bool active) {
    if (!active) return;
    const float pp[3] = { p.x, p.y, p.z }, ff[3] = { f.x, f.y, f.z };
    if (SMIS) {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            float ratio = pp[j] / ff[i];
            if (!std::isfinite(ratio)) ratio = 0.f;
            ratio *= W.w[i][j];
            W.w[i][j] = std::isnan(ratio) ? 0.f : ratio;
        }
    } else {
        const float pdf = pp[channel];
        for (int j = 0; j < 3; ++j) { float ratio = W.w[0][j] * (pdf / ff[j]); W.w[0][j] = std::isfinite(ratio) ? ratio : 0.f; }
    }
}
/* :654-668 */
template <bool SMIS> static inline V3 mis_weight1(const MisW<SMIS> &W) {
    if (SMIS) {
        float r[3];
        for (int i = 0; i < 3; ++i) { float sum = (W.w[i][0] + W.w[i][1]) + W.w[i][2]; r[i] = sum == 0.f ? 0.f : 3.f / sum; }
        return V3(r[0], r[1], r[2]);
    }
    bool invalid = fminf(fminf(fabsf(W.w[0][0]), fabsf(W.w[0][1])), fabsf(W.w[0][2])) == 0.f;
    return invalid ? V3(0.f) : V3(1.f / W.w[0][0], 1.f / W.w[0][1], 1.f / W.w[0][2]);
}
/* :671-685 */
template <bool SMIS> static inline V3 mis_weight2(const MisW<SMIS> &A, const MisW<SMIS> &B) {
    if (SMIS) {
        float r[3];
        for (int i = 0; i < 3; ++i) { float sum = ((A.w[i][0] + B.w[i][0]) + (A.w[i][1] + B.w[i][1])) + (A.w[i][2] + B.w[i][2]); r[i] = sum == 0.f ? 0.f : 3.f / sum; }
        return V3(r[0], r[1], r[2]);
    }
    float s[3] = { A.w[0][0] + B.w[0][0], A.w[0][1] + B.w[0][1], A.w[0][2] + B.w[0][2] };
    bool invalid = fminf(fminf(fabsf(s[0]), fabsf(s[1])), fabsf(s[2])) == 0.f;
    return invalid ? V3(0.f) : V3(1.f / s[0], 1.f / s[1], 1.f / s[2]);
}

/* volpathmis.cpp:449-629 sample_emitter */
template <bool SMIS>
static V3 mis_sample_emitter(Ctx &C, V3 ref_p, V3 ref_n, const SI *ref_si, int medium, const MisW<SMIS> &p_over_f, uint32_t channel,
                             DirSample *ds_out, MisW<SMIS> *nee_out, MisW<SMIS> *uni_out) {
    const Scene &S = C.S;
    MisW<SMIS> nee = p_over_f, uni = p_over_f;
    float sx, sy; C.next2(&sx, &sy);
    DirSample ds; V3 w = sample_emitter_direction(S, ref_p, sx, sy, &ds);
    V3 emitter_val = w * ds.pdf;                                   /* emitter_sample_weight * ds.pdf */
    if (ds.pdf == 0.f) emitter_val = V3(0.f);
    bool active = ds.pdf != 0.f;
    mis_update(nee, V3(ds.pdf), V3(1.f), channel, active);
    *ds_out = ds;
    if (!active) { *nee_out = nee; *uni_out = uni; return emitter_val; }
    Ray ray = spawn_ray_to(ref_p, ref_n, ds.p);
    float max_dist = ray.maxt;
    if (ref_si && is_medium_transition(S.shapes[ref_si->shape])) medium = target_medium(S.shapes[ref_si->shape], ray.d, ref_si->n);
    float total_dist = 0.f;
    SI si; memset((void *) &si, 0, sizeof(si)); si.t = kInf;
    bool needs_intersection = true;
    while (active) {
        float remaining_dist = max_dist - total_dist;
        ray.maxt = remaining_dist;
        if (!(remaining_dist > 0.f)) { C.skip(1); break; }
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (!active_medium) C.skip(1);
        if (active_medium) {
            const lrt_medium_desc &M = S.media[medium];
            MI mei = medium_sample_interaction(S, medium, ray, C.next(), channel);
            if (mei.valid() && medium_is_homogeneous(M)) ray.maxt = fminf(mei.t, remaining_dist);
            if (needs_intersection) { C.n_shadow++; C.n_shadow_needed++; Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
            if (si.t < mei.t) mei.t = kInf;
            needs_intersection = false;
            bool is_spectral = M.has_spectral_extinction, not_spectral = !is_spectral;
            if (is_spectral) {
                float t = fminf(remaining_dist, fminf(mei.t, si.t)) - mei.mint;
                V3 tr(m_exp(-t * mei.combined.x), m_exp(-t * mei.combined.y), m_exp(-t * mei.combined.z));
                V3 ffp = (si.t < mei.t || mei.t > remaining_dist) ? tr : tr * mei.combined;
                mis_update(nee, ffp, tr, channel, true); mis_update(uni, ffp, tr, channel, true);
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
        if (intersect) { C.n_shadow++; C.n_shadow_needed++; Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
        active_surface = active_surface || escaped_medium;
        if (active_surface) total_dist += si.t;
        active_surface = active_surface && si.valid && !active_medium;
        if (active_surface) {
            V3 bv(bsdf_null_transmission(S, S.shapes[si.shape].bsdf));
            mis_update(nee, V3(1.f), bv, channel, true); mis_update(uni, V3(1.f), bv, channel, true);
            ray = spawn_ray(si.p, si.n, ray.d);
        }
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        if (SMIS) active = (active_medium || active_surface) && any_nonzero(mis_weight1(uni));
        else active = (active_medium || active_surface) && (uni.w[0][0] != 0.f || uni.w[0][1] != 0.f || uni.w[0][2] != 0.f || nee.w[0][0] != 0.f || nee.w[0][1] != 0.f || nee.w[0][2] != 0.f);
        if (active_surface && is_medium_transition(S.shapes[si.shape])) medium = target_medium(S.shapes[si.shape], ray.d, si.n);
    }
    *nee_out = nee; *uni_out = uni;
    return emitter_val;
}

/* volpathmis.cpp:127-446 */
template <bool SMIS>
static void volpathmis_sample(Ctx &C, Ray ray, int medium, V3 *out, bool *out_valid) {
    const Scene &S = C.S;
    bool valid_ray = !C.hide_emitters && S.env >= 0;
    float eta = 1.f;
    V3 result(0.f);
    bool specular_chain = !C.hide_emitters;
    uint32_t depth = 0;
    MisW<SMIS> p_over_f = MisW<SMIS>::ones(), p_over_f_nee = MisW<SMIS>::ones();
    uint32_t channel = std::min((uint32_t) (C.next() * 3.f), 2u);
    SI si; memset((void *) &si, 0, sizeof(si)); si.t = kInf;
    bool needs_intersection = true, last_event_was_null = false, active = true;
    V3 last_scatter_p(0.f);
    const uint32_t max_depth = (uint32_t) C.max_depth;
    while (active) {
        C.n_iter++;
        V3 mis_throughput = mis_weight1(p_over_f);
        float q = fminf(max3(mis_throughput) * sqr(eta), .95f);
        bool perform_rr = !last_event_was_null && depth > (uint32_t) C.rr_depth;
        float u = C.next();
        active = !(u >= q && perform_rr);
        mis_update(p_over_f, V3(q), V3(1.f), channel, perform_rr);
        last_event_was_null = false;
        active = active && depth < max_depth;
        active = active && any_nonzero(mis_weight1(p_over_f));
        if (!active) break;

        bool active_medium = medium >= 0, active_surface = !active_medium;
        bool act_null_scatter = false, act_medium_scatter = false, escaped_medium = false;
        bool is_spectral = false, not_spectral = false;
        MI mei; mei.t = kInf;
        if (!active_medium) C.skip(2);
        if (active_medium) {
            const lrt_medium_desc &M = S.media[medium];
            is_spectral = M.has_spectral_extinction; not_spectral = !is_spectral;
            mei = medium_sample_interaction(S, medium, ray, C.next(), channel);
            if (mei.valid() && medium_is_homogeneous(M)) ray.maxt = mei.t;
            if (needs_intersection) { Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
            needs_intersection = false;
            if (si.t < mei.t) mei.t = kInf;
            if (is_spectral) {
                float t = fminf(mei.t, si.t) - mei.mint;
                V3 tr(m_exp(-t * mei.combined.x), m_exp(-t * mei.combined.y), m_exp(-t * mei.combined.z));
                V3 pdf = (si.t < mei.t) ? tr : tr * mei.combined;
                mis_update(p_over_f, pdf, tr, channel, true); mis_update(p_over_f_nee, pdf, tr, channel, true);
            }
            escaped_medium = !mei.valid();
            active_medium = mei.valid();
            is_spectral = is_spectral && active_medium; not_spectral = not_spectral && active_medium;
            if (!active_medium) C.skip(1);
        }
        if (!active_medium) C.skip(3);                                  /* NEE next_2d, phase next_1d / next_2d */
        if (active_medium) {
            const lrt_medium_desc &M = S.media[medium];
            float null_scatter_prob = mean3(mei.sigma_n / mei.combined);
            act_null_scatter = C.next() < null_scatter_prob;
            act_medium_scatter = !act_null_scatter;
            last_event_was_null = act_null_scatter;
            if (act_medium_scatter) { depth += 1; last_scatter_p = mei.p; }
            bool sample_emitters = M.sample_emitters != 0;
            active = active && depth < max_depth;
            act_medium_scatter = act_medium_scatter && active;
            if (act_medium_scatter) specular_chain = !sample_emitters;
            if (act_null_scatter) {
                if (is_spectral) { mis_update(p_over_f, V3(null_scatter_prob), mei.sigma_n, channel, true); mis_update(p_over_f_nee, V3(1.f), mei.sigma_n, channel, true); }
                if (not_spectral) { mis_update(p_over_f, mei.sigma_n, mei.sigma_n, channel, true); mis_update(p_over_f_nee, V3(1.f), V3(null_scatter_prob), channel, true); }
                ray.o = mei.p; si.t = si.t - mei.t;
            }
            if (!act_medium_scatter) C.skip(3);
            if (act_medium_scatter) {
                if (is_spectral) mis_update(p_over_f, V3(1.f - null_scatter_prob), mei.sigma_s, channel, true);
                if (not_spectral) mis_update(p_over_f, mei.sigma_t, mei.sigma_s, channel, true);
                valid_ray = true;
                if (!sample_emitters) C.skip(1);
                if (sample_emitters) {
                    DirSample ds; MisW<SMIS> nee_end, uni_end;
                    V3 emitted = mis_sample_emitter<SMIS>(C, mei.p, V3(0.f), nullptr, medium, p_over_f, channel, &ds, &nee_end, &uni_end);
                    float pv = phase_eval(M, mei.wi, ds.d);
                    bool a = ds.pdf != 0.f;                                 /* the callee's `active` does not flow back: active_e masks these */
                    (void) a;
                    mis_update(nee_end, V3(1.f), V3(pv), channel, true);
                    mis_update(uni_end, V3(ds.delta ? 0.f : pv), V3(pv), channel, true);
                    result += mis_weight2(nee_end, uni_end) * emitted;
                }
                p_over_f_nee = p_over_f;
                float s1 = C.next(); (void) s1;
                float s2x, s2y; C.next2(&s2x, &s2y);
                V3 wo; float phase_pdf; phase_sample(M, mei.wi, s2x, s2y, &wo, &phase_pdf);
                ray = spawn_ray(mei.p, V3(0.f), wo);
                needs_intersection = true;
                /* phase_weight = 1: f = phase_weight * phase_pdf = phase_pdf */
                mis_update(p_over_f, V3(phase_pdf), V3(1.f * phase_pdf), channel, true);
                mis_update(p_over_f_nee, V3(1.f), V3(1.f * phase_pdf), channel, true);
            }
        }
        /* --------------------- surface interactions --------------------- */
        active_surface = active_surface || escaped_medium;
        bool intersect = active_surface && needs_intersection;
        if (intersect) { Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
        if (active_surface) {
            if (C.hide_emitters && depth == 0 && intersect) {
                bool skip = si.valid && S.shapes[si.shape].emitter >= 0;
                if (skip) {
                    Ray r2 = spawn_ray(si.p, si.n, ray.d);
                    bool a = true; Hit h; h.prim = 0xffffffffu; h.t = kInf;
                    while (a) {
                        h = S.intersect(r2, false, false);
                        a = h.valid() && S.shapes[S.face_shape[h.prim]].emitter >= 0;
                        if (a) { SI s2 = S.compute_si(r2, h); r2 = spawn_ray(s2.p, s2.n, r2.d); }
                    }
                    si = S.compute_si(r2, h);
                }
            }
            bool count_direct = depth == 0 || specular_chain;
            int emitter = si_emitter(S, si);
            bool active_e = emitter >= 0 && !(depth == 0 && C.hide_emitters);
            if (active_e) {
                /* :384-389: the update carries the mask active_e, not active_e && !count_direct (JIT: the block is always traced) */
                float emitter_pdf = pdf_emitter_direction(S, last_scatter_p, si, emitter);
                mis_update(p_over_f_nee, V3(emitter_pdf), V3(1.f), channel, true);
                V3 emitted = emitter_eval(S, emitter, si);
                V3 contrib = count_direct ? mis_weight1(p_over_f) * emitted : mis_weight2(p_over_f, p_over_f_nee) * emitted;
                result += contrib;
            }
        }
        active_surface = active_surface && si.valid;
        if (!active_surface) C.skip(3);
        if (active_surface) {
            const lrt_shape_desc &sd = S.shapes[si.shape];
            int b = sd.bsdf;
            int flags = bsdf_flags(S, b);
            bool active_e = (flags & F_SMOOTH) && (depth + 1 < max_depth);
            if (!active_e) C.skip(1);
            if (active_e) {
                DirSample ds; MisW<SMIS> nee_end, uni_end;
                V3 emitted = mis_sample_emitter<SMIS>(C, si.p, si.n, &si, medium, p_over_f, channel, &ds, &nee_end, &uni_end);
                V3 wo = si.sh.to_local(ds.d);
                V3 bsdf_val = bsdf_eval(S, b, si, wo);
                float bpdf = bsdf_pdf(S, b, si, wo);
                mis_update(nee_end, V3(1.f), bsdf_val, channel, true);
                mis_update(uni_end, V3(ds.delta ? 0.f : bpdf), bsdf_val, channel, true);
                result += mis_weight2(nee_end, uni_end) * emitted;
            }
            float s1 = C.next(), s2x, s2y; C.next2(&s2x, &s2y);
            BSDFSample bs; V3 bsdf_weight;
            bsdf_sample(S, b, si, s1, s2x, s2y, &bs, &bsdf_weight);
            bool invalid_bsdf_sample = bs.pdf == 0.f;
            active_surface = bs.pdf > 0.f;
            if (active_surface) {
                eta *= bs.eta;
                ray = spawn_ray(si.p, si.n, si.sh.to_world(bs.wo));
                needs_intersection = true;
            }
            bool non_null = active_surface && !(bs.type & F_NULL);
            valid_ray = valid_ray || non_null || invalid_bsdf_sample;
            specular_chain = specular_chain || (non_null && (bs.type & F_DELTA));
            specular_chain = specular_chain && !(active_surface && (bs.type & F_SMOOTH));
            if (non_null) { depth += 1; last_scatter_p = si.p; p_over_f_nee = p_over_f; }
            V3 f = bsdf_weight * bs.pdf;
            mis_update(p_over_f, V3(bs.pdf), f, channel, active_surface);
            mis_update(p_over_f_nee, V3(1.f), f, channel, non_null);
            if (active_surface && is_medium_transition(sd)) medium = target_medium(sd, ray.d, si.n);
        }
        active = active && (active_surface || active_medium);
    }
    *out = result; *out_valid = valid_ray;
}
