// The fork's bio transport on the persistent render kernel: the `liver`, `parenchyma` and `glissonCapsule` media
// (element-competition free-flight sampling of the 5-argument Medium::sample_interaction: src/media/liver.cpp:227-539,
// src/media/parenchyma.cpp:193-368, src/media/glissonCapsule.cpp:229-353) and one trip of the `biovolpath`
// (src/integrators/biovolpath.cpp:95-541) and `biovolpath06` (src/integrators/biovolpath06.cpp:88-473) loops.
// Specification with every decision: docs/BIO_TRANSPORT_SPEC.md.  Included by kernels.h (needs PathState).
//
// `biovolpath` follows the JIT variants' lane semantics (`dr::any_or<true>` is the constant true, `dr::none_or<false>`
// never returns early); `biovolpath06` is scalar-only code in the reference and runs with scalar semantics per lane.
#pragma once

namespace lrt {

enum { BIO_ABSORBER = 0, BIO_ATTENUATOR = 1, BIO_ABSORBER_AND_ATTENUATOR = 2 };   // src/media/organic_material.h:29-34

struct BioMI { float t; V3 p; V3 transmittance; V3 combined; DEV bool valid() const { return t != kInf; } };

// one element of the competition (liver.cpp:320-335 / :358-381): candidate distance from the inner generator
DEV void bio_element(PCG32 &rng, float att, bool hepatocyte, float log10_hep, bool guard_positive, bool first, int i, int &element, float &distance) {
    float r = rng.next();
    if (r == 0.f) r = 0.5f;
    if (guard_positive && !(att > 0.f)) return;
    const float lr = m_log(r);
    float aux = -(1.f / att) * lr;
    if (hepatocyte) aux = -(log10_hep * lr);                       // EAbsorberAndAttenuator: -(log2(att + 1) / log2(10) * log(r))
    if (first || aux < distance) { element = i; distance = aux; }
}

// `computeDistance`: a fresh PCG32 seeded with the bit pattern of the free-flight sample, default stream (liver.cpp:233-235)
DEV void bio_compute_distance(const DBioMedium &B, uint32_t channel, float sample, float depth, int &bio_type, float &distance) {
    distance = kInf; int element = 0;
    PCG32 rng; rng.ld_count = 0; rng.seed((uint64_t) f2u(sample), 0xda3e39cb94b95bdbULL);
    const bool layered = B.type == LRT_MEDIUM_LIVER || B.type == LRT_MEDIUM_GLISSON;
    int layer = 0;
    if (layered) {                                                 // liver.cpp:246-251: later tests overwrite earlier ones
        if (depth <= B.layer_limit[1]) layer = 1;
        if (depth <= B.layer_limit[2]) layer = 2;
        if (depth <= B.layer_limit[3]) layer = 3;
        if (depth > B.layer_limit[3]) layer = 4;
    }
    if (layered && layer < 4) {
        const float c = B.collagen[layer][channel], e = B.elastin[layer][channel];
        bio_element(rng, c, false, 0.f, false, true, 0, element, distance);
        bio_element(rng, e, false, 0.f, false, false, 1, element, distance);
        bio_type = BIO_ATTENUATOR; return;
    }
    if (B.type == LRT_MEDIUM_GLISSON) { bio_type = BIO_ATTENUATOR; return; }
    bio_element(rng, B.blood[channel], false, 0.f, true, true, 0, element, distance);
    bio_element(rng, B.bile[channel], false, 0.f, true, false, 1, element, distance);
    bio_element(rng, B.lipid_water[channel], false, 0.f, true, false, 2, element, distance);
    bio_element(rng, B.hepatocity, true, B.log10_hep, true, false, 3, element, distance);
    bio_type = element == 3 ? BIO_ABSORBER_AND_ATTENUATOR : BIO_ABSORBER;
}

// 5-argument sample_interaction (liver.cpp:479-539, parenchyma.cpp:303-368, glissonCapsule.cpp:309-353); maxt is the
// surface distance the integrator hands in (`Ray3f(ray, si.t)`).  JIT: parenchyma's `if / else if` chains keep their
// first branch only (any_or<true>); liver's plain ifs and glissonCapsule's forced `active = true` read the same both ways.
template <bool JIT>
DEV BioMI bio_finish_interaction(const DBioMedium &B, V3 o, V3 d, float maxt, uint32_t channel, int bio_type, float distance) {
    BioMI mei; mei.t = kInf; mei.p = V3(0.f); mei.transmittance = V3(1.f);
    const bool inside = distance > 0.f && distance < maxt;
    if (inside) { mei.t = distance; mei.p = fma3(d, distance, o); }
    bool active = true;
    if (B.type != LRT_MEDIUM_GLISSON) {
        const bool chain = JIT && B.type == LRT_MEDIUM_PARENCHYMA;
        if (bio_type == BIO_ABSORBER) active = false;
        if (!chain && bio_type == BIO_ABSORBER_AND_ATTENUATOR && distance <= 0.0025f) active = false;   // (double) distance < 0.0025
    }
    const V3 onehot(channel == 0 ? 1.f : 0.f, channel == 1 ? 1.f : 0.f, channel == 2 ? 1.f : 0.f);
    if (B.type == LRT_MEDIUM_LIVER || !JIT) {
        if (inside) mei.transmittance = active ? onehot : V3(0.f);
    } else if (inside && active) mei.transmittance = onehot;
    mei.combined = V3(B.sigmat[0], B.sigmat[1], B.sigmat[2]);
    return mei;
}
template <bool JIT>
DEV BioMI bio_sample_interaction(const DBioMedium &B, V3 o, V3 d, float maxt, float sample, uint32_t channel, float depth) {
    int bio_type; float distance;
    bio_compute_distance(B, channel, sample, depth, bio_type, distance);
    return bio_finish_interaction<JIT>(B, o, d, maxt, channel, bio_type, distance);
}
// The winner's class from what a path record keeps of a competition already run (its distance and "the hepatocytes won"):
// the same case analysis as bio_compute_distance.
DEV int bio_cached_type(const DBioMedium &B, float depth, bool hep) {
    const bool layered = B.type == LRT_MEDIUM_LIVER || B.type == LRT_MEDIUM_GLISSON;
    if ((layered && !(depth > B.layer_limit[3])) || B.type == LRT_MEDIUM_GLISSON) return BIO_ATTENUATOR;
    return hep ? BIO_ABSORBER_AND_ATTENUATOR : BIO_ABSORBER;
}

// biovolpath.cpp:383-541 sample_emitter (surface reference points only)
template <typename SMP, typename TR>
DEV V3 bio_sample_emitter(SceneRef sc, SMP &rng, V3 ref_p, V3 ref_n, uint32_t ref_shape, int medium, uint32_t channel, float tissue_depth,
                          DirSample *ds_out, const TR &tr, uint32_t &n_shadow) {
    V3 transmittance(1.f);
    float sx, sy; rng.next2(sx, sy);
    DirSample ds; V3 emitter_val = sample_emitter_direction(sc, ref_p, sx, sy, &ds);
    *ds_out = ds;
    if (ds.pdf == 0.f) return V3(0.f);
    Ray ray = spawn_ray_to(ref_p, ref_n, ds.p);
    float max_dist = ray.maxt;
    { const DShape sd = tab(sc.shapes, ref_shape, sc.one_shape); if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, ref_n); }
    float total_dist = 0.f;
    SI si; si.valid = false; si.t = kInf; si.shape = 0; si.p = V3(0.f); si.n = V3(0.f);
    bool needs_intersection = true, active = true;
    while (active) {
        float remaining_dist = max_dist - total_dist;
        ray.maxt = remaining_dist;
        if (!(remaining_dist > 0.f)) { rng.skip(1); break; }
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (!active_medium) rng.skip(1);                                // biovolpath.cpp:464
        if (active_medium) {
            const DBioMedium &B = sc.bio[medium];
            BioMI mei = bio_sample_interaction<true>(B, ray.o, ray.d, si.t, rng.next(), channel, tissue_depth);
            if (mei.valid()) ray.maxt = fmin_(mei.t, remaining_dist);
            if (needs_intersection) { n_shadow++; Hit h = tr.closest(ray); si = tr.surface(sc, ray, h); }
            if (si.t < mei.t) mei.t = kInf;
            needs_intersection = false;
            if (B.has_spectral_extinction) {
                float t = fmin_(remaining_dist, fmin_(mei.t, si.t)) - 0.f;
                V3 trm = exp_neg(t, mei.combined);
                V3 ffp = (si.t < mei.t || mei.t > remaining_dist) ? trm : trm * mei.combined;
                float tr_pdf = idx3(ffp, channel);
                transmittance = transmittance * ((tr_pdf > 0.f) ? div_uniform(trm, tr_pdf) : V3(0.f));
            }
            if ((mei.t > remaining_dist) && mei.valid()) total_dist = ds.dist;
            if (mei.t > remaining_dist) mei.t = kInf;
            escaped_medium = !mei.valid();
            active_medium = mei.valid();
            if (active_medium) {
                total_dist += mei.t;
                ray.o = mei.p;
                si.t = si.t - mei.t;
                transmittance = transmittance * mei.transmittance;      // :500-503
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

// One trip of biovolpath's while_loop (biovolpath.cpp:177-374), JIT-variant lane semantics.  s.si_t carries the distance
// the previous trip's ray query returned (`Ray3f(ray, si.t)` at :226), s.tdepth the loop state `tissueDepth`.
// As in volpath_iteration (kernels.h), the first stage of a trip - the termination test (:200-208) and, inside a medium, the free-flight
// draw with its element competition (:226) - runs one trip early on the lane's own generator (`fresh`: at the start of a camera lane's
// first trip): a queued record holds a path that is known to run its next trip, with the competition's outcome in the record.
template <typename SMP, typename TR>
DEV bool biovolpath_iteration(SceneRef sc, RpRef rp, PathState &s, SMP &rng, const TR &tr, uint32_t &n_shadow, uint32_t &n_extra, bool fresh = false) {
    uint32_t depth = s.flags & PF_DEPTH_MASK;
    bool proven_empty = (s.flags & PF_NOHIT) != 0;                     // the free-flight stage of the previous trip (below)
    int medium = (int) ((s.flags & PF_MEDIUM_MASK) >> PF_MEDIUM_SHIFT) - 1;
    const uint32_t channel = (s.flags >> PF_CHANNEL_SHIFT) & 3u;
    bool specular_chain = (s.flags & PF_SPECULAR) != 0, valid_ray = (s.flags & PF_VALID) != 0;
    const uint32_t max_depth = (uint32_t) rp.max_depth;
    V3 throughput = s.tp, result = s.res;
    float eta = s.eta, tissue_depth = s.tdepth, si_t = s.si_t;
    float bio_dist = s.bio_dist; bool bio_hep = s.bio_hep;
    Ray ray; ray.o = s.o; ray.d = s.d; ray.maxt = s.maxt;
    auto commit = [&]() {
        s.tp = throughput; s.res = result; s.eta = eta; s.o = ray.o; s.d = ray.d; s.maxt = ray.maxt; s.tdepth = tissue_depth; s.si_t = si_t;
        s.flags = (depth & PF_DEPTH_MASK) | ((uint32_t) (medium + 1) << PF_MEDIUM_SHIFT) | (channel << PF_CHANNEL_SHIFT) |
                  (specular_chain ? PF_SPECULAR : 0u) | (valid_ray ? PF_VALID : 0u);
    };
    // ---- termination (:200-208) of the trip about to run
    auto termination_stage = [&]() -> bool {
        bool a = any_nonzero(throughput);
        const float q = fmin_(max3(throughput) * sqr(eta), .95f);
        const bool perform_rr = depth > (uint32_t) rp.rr_depth;
        if (a) { const float u = rng.next(); a = (u < q) || !perform_rr; }
        if (perform_rr) throughput = throughput * rcp(q);
        return a && depth < max_depth;
    };
    // inside a medium: the free-flight draw of the trip about to run, its element competition, and the attempt to prove the segment free
    uint32_t nohit = 0, longq = 0;
    float cache_dist = u2f(0x7fc00000u); bool cache_hep = false;
    auto free_flight_stage = [&]() {
        if (medium < 0) return;
        int type2; float dist2;
        bio_compute_distance(sc.bio[medium], channel, rng.next(), tissue_depth, type2, dist2);
        const BioMI m2 = bio_finish_interaction<true>(sc.bio[medium], ray.o, ray.d, si_t, channel, type2, dist2);
        if (sc.grid.enabled && m2.valid() && segment_free_of_surfaces(sc.grid, ray.o, ray.d, m2.t)) nohit = PF_NOHIT;
        if (!m2.valid()) longq = PF_LONG_QUERY;                         // the next trip's query runs to the largest float: its own queue region (k_render)
        cache_dist = dist2 == dist2 ? dist2 : kInf; cache_hep = type2 == BIO_ABSORBER_AND_ATTENUATOR;   // kept in the record: the next trip starts from it
    };
    if (fresh) {
        if (!termination_stage()) {
            // the body still runs for this lane: its (masked) virtual call returns a zero `mei`, so :297-300 clears the result
            result = V3(0.f);
            commit(); return false;
        }
        free_flight_stage(); bio_dist = cache_dist; bio_hep = cache_hep; proven_empty = nohit != 0;
        nohit = 0; longq = 0; cache_dist = u2f(0x7fc00000u); cache_hep = false;
    }
    bool active = true;
    bool active_medium = medium >= 0, active_surface = !active_medium;
    const bool in_medium_lane = active_medium;
    bool act_medium_scatter = false, escaped_medium = false;
    BioMI mei; mei.t = kInf; mei.p = V3(0.f); mei.transmittance = V3(0.f); mei.combined = V3(0.f);
    SI si; si.valid = false; si.t = kInf;
    if (!active_medium) rng.skip(2);                                    // :226, :244
    if (active_medium) {
        const DBioMedium &B = sc.bio[medium];
        // the free-flight stage ran this trip's competition (same sample, channel, depth)
        mei = bio_finish_interaction<true>(B, ray.o, ray.d, si_t, channel, bio_cached_type(B, tissue_depth, bio_hep), bio_dist);
        if (mei.valid()) ray.maxt = mei.t;
        if (!proven_empty) { Hit h = tr.closest(ray); si = tr.surface(sc, ray, h); }   // else: no surface within mei.t, the query returns "none"
        si_t = si.t;
        if (si.t < mei.t) mei.t = kInf;
        if (B.has_spectral_extinction) {                                // Medium::transmittance_eval_pdf (medium.cpp:92-104)
            float t = fmin_(mei.t, si.t) - 0.f;
            V3 trm = exp_neg(t, mei.combined);
            V3 pdf = (si.t < mei.t) ? trm : trm * mei.combined;
            float tr_pdf = idx3(pdf, channel);
            throughput = throughput * ((tr_pdf > 0.f) ? div_uniform(trm, tr_pdf) : V3(0.f));
        }
        escaped_medium = !mei.valid();
        active_medium = mei.valid();
        if (!active_medium) rng.skip(1);
        if (active_medium) {
            (void) rng.next();                                          // :244 null / real draw: sigma_t / combined = 1, always real
            act_medium_scatter = true;
            depth += 1;
            s.lp = mei.p;
        }
    }
    active = active && depth < max_depth;
    act_medium_scatter = act_medium_scatter && active;
    if (!act_medium_scatter) rng.skip(2);                               // :283, :284
    if (act_medium_scatter) {
        const DMedium M = tab(sc.media, medium);
        throughput = throughput * mei.transmittance;                    // :268 / :272
        tissue_depth += __builtin_fabsf(-ray.d.z * mei.t);              // |cos_theta(-ray.d) * mei.t|
        (void) rng.next();
        float s2x, s2y; rng.next2(s2x, s2y);
        V3 wo; float phase_pdf; phase_sample(M, -ray.d, s2x, s2y, &wo, &phase_pdf);
        if (phase_pdf > 0.f) {
            ray = spawn_ray(mei.p, V3(0.f), wo);
            s.last_pdf = phase_pdf;
        }
    }
    // ---- surface interactions
    active_surface = active_surface || escaped_medium;
    const bool intersect = active_surface && !escaped_medium;
    {   // :297-300, unconditional in the JIT variants; lanes outside a medium see the zero `mei` of the masked call
        const V3 T = in_medium_lane ? mei.transmittance : V3(0.f);
        if (T.x == 0.f) result.x = 0.f;
        if (T.y == 0.f) result.y = 0.f;
        if (T.z == 0.f) result.z = 0.f;
        if (medium >= 0) throughput = throughput * T;
    }
    if (intersect) { Hit h = tr.closest(ray); si = tr.surface(sc, ray, h); si_t = si.t; }
    if (active_surface) {
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
    if (!active_surface) rng.skip(3);                                   // :390 (NEE), :348, :349
    if (active_surface) {
        const DShape sd = tab(sc.shapes, si.shape, sc.one_shape);
        int b = sd.bsdf;
        int flags = tab(sc.bsdfs, b, sc.one_shape).flags;
        bool active_e = (flags & F_SMOOTH) && (depth + 1 < max_depth);
        if (!active_e) rng.skip(1);
        if (active_e) {
            DirSample ds;
            V3 emitted = bio_sample_emitter(sc, rng, si.p, si.n, si.shape, medium, channel, tissue_depth, &ds, tr, n_shadow);
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
    // ---- the first stage of the next trip, for real (see the top).  A path that stops there is retired now, with the cleared result that
    // trip leaves (JIT reading of :200-208 + :297-300); the trip is counted in n_extra.  In a medium the next competition's outcome is then
    // known: when it places an interaction inside the surface distance and the distance field proves the segment free of surfaces,
    // the path is queued in region A and skips its ray query.
    if (active) {
        if (!termination_stage()) { active = false; n_extra += 1; result = V3(0.f); }
        else free_flight_stage();
    }
    commit();
    s.flags |= nohit | longq; s.bio_dist = cache_dist; s.bio_hep = cache_hep;
    return active;
}

// One trip of biovolpath06's `while` (biovolpath06.cpp:176-305), scalar semantics.  The ray query the source issues at
// the END of a trip (:195, :281; :120 for the camera ray) runs at the top of the next one here: same ray, same answer.
// Flag bits: PF_SPECULAR = null_chain, PF_BIO_SCATTERED = scattered_chain; the recursion-type word only ever holds 127,
// 0x27e, 0x27f or 0x1, i.e. two facts: PF_BIO_EMIT = (type & 0x0001), PF_BIO_FULL = (type & 0x0004) = (type & 0x0008).
// No sampler call is skipped: scalar code executes only the calls it reaches.
template <typename SMP, typename TR>
DEV bool biovolpath06_iteration(SceneRef sc, RpRef rp, PathState &s, SMP &rng, const TR &tr) {
    uint32_t depth = s.flags & PF_DEPTH_MASK;
    int medium = (int) ((s.flags & PF_MEDIUM_MASK) >> PF_MEDIUM_SHIFT) - 1;
    const uint32_t channel = (s.flags >> PF_CHANNEL_SHIFT) & 3u;
    bool null_chain = (s.flags & PF_SPECULAR) != 0, scattered_chain = (s.flags & PF_BIO_SCATTERED) != 0;
    bool type_emit = (s.flags & PF_BIO_EMIT) != 0, type_full = (s.flags & PF_BIO_FULL) != 0;
    const uint32_t valid_bit = s.flags & PF_VALID;
    const uint32_t max_depth = min((uint32_t) rp.max_depth, 65534u);   // depth counts trips and lives in 16 flag bits: paths stop after 65535 trips
    V3 throughput = s.tp, result = s.res;
    float eta = s.eta, tissue_depth = s.tdepth;
    Ray ray; ray.o = s.o; ray.d = s.d; ray.maxt = s.maxt;
    auto commit = [&]() {
        s.tp = throughput; s.res = result; s.eta = eta; s.o = ray.o; s.d = ray.d; s.maxt = ray.maxt; s.tdepth = tissue_depth;
        s.flags = (depth & PF_DEPTH_MASK) | ((uint32_t) (medium + 1) << PF_MEDIUM_SHIFT) | (channel << PF_CHANNEL_SHIFT) |
                  (null_chain ? PF_SPECULAR : 0u) | (scattered_chain ? PF_BIO_SCATTERED : 0u) | (type_emit ? PF_BIO_EMIT : 0u) |
                  (type_full ? PF_BIO_FULL : 0u) | valid_bit;
    };
    // (look-ahead of the previous trip, below: no surface within the competition's distance -> the query's answer cannot matter)
    const bool proven_empty = (s.flags & PF_NOHIT) != 0;
    SI si; si.valid = false; si.t = kInf;
    if (!proven_empty) { Hit h = tr.closest(ray); si = tr.surface(sc, ray, h); }
    const bool in_medium = medium >= 0;
    BioMI mei; mei.t = kInf; mei.p = V3(0.f); mei.transmittance = V3(0.f); mei.combined = V3(0.f);
    if (in_medium) {
        const float sample = rng.next();
        if (s.bio_dist == s.bio_dist)                                   // the competition the look-ahead already ran (same sample, channel, depth)
            mei = bio_finish_interaction<false>(sc.bio[medium], ray.o, ray.d, si.t, channel, bio_cached_type(sc.bio[medium], tissue_depth, s.bio_hep), s.bio_dist);
        else mei = bio_sample_interaction<false>(sc.bio[medium], ray.o, ray.d, si.t, sample, channel, tissue_depth);
    }
    bool alive = true;
    if (in_medium && mei.valid()) {                                     // :185-198
        const DMedium M = tab(sc.media, medium);
        throughput = throughput * mei.transmittance;
        (void) rng.next();
        float s2x, s2y; rng.next2(s2x, s2y);
        V3 wo; float phase_pdf; phase_sample(M, -ray.d, s2x, s2y, &wo, &phase_pdf);
        tissue_depth += __builtin_fabsf(-ray.d.z * mei.t);
        ray.o = mei.p; ray.d = wo; ray.maxt = kLargest;                 // Ray3f(mei.p, wo, time, wavelengths)
        null_chain = false; scattered_chain = true;
    } else {
        if (in_medium) throughput = throughput * mei.transmittance;
        if (!si.valid) {                                                // :210-224 (and the null-BSDF dereference behind it: the path ends)
            const bool active_e = (scattered_chain || !rp.hide_emitters) && type_emit && sc.env.type >= 0;
            if (active_e) {
                V3 contrib = throughput * emitter_eval(sc, sc.env.emitter, si);
                if (in_medium) {                                        // transmittance_eval_pdf(mei, si, true)
                    float t = fmin_(mei.t, si.t) - 0.f;
                    contrib = contrib * V3(m_exp(-t * mei.combined.x), m_exp(-t * mei.combined.y), m_exp(-t * mei.combined.z));
                }
                result = result + contrib;
            }
            commit(); return false;
        }
        const DShape sd = tab(sc.shapes, si.shape, sc.one_shape);
        const int b = sd.bsdf;
        float s1 = rng.next(), s2x, s2y; rng.next2(s2x, s2y);
        const BSDFSample bs = bsdf_sample(sc, b, si, s1, s2x, s2y);
        if (!any_nonzero(bs.weight)) { commit(); return false; }
        const bool new_full = depth + 1 < max_depth && type_full;       // :243-247 IndirectSurfaceRadiance
        bool new_emit = false;
        if (depth < max_depth && type_full && (bs.type & F_DELTA) && (!(bs.type & F_NULL) || null_chain)) { new_emit = true; null_chain = true; }   // :249-254
        else null_chain = null_chain && (bs.type & F_NULL);
        if (!new_full && !new_emit) { commit(); return false; }         // recursiveType == 0
        type_full = new_full; type_emit = new_emit;
        V3 wo = si.sh.to_world(bs.wo);
        throughput = throughput * bs.weight;
        eta *= bs.eta;
        ray = spawn_ray(si.p, si.n, wo);
        if (is_medium_transition(sd)) medium = target_medium(sd, wo, si.n);
        scattered_chain = scattered_chain || !(bs.type & F_NULL);
    }
    if (depth++ > (uint32_t) rp.rr_depth) {                             // :298-304
        float q = fmin_(max3(throughput) * sqr(eta), .95f);
        if (rng.next() >= q) alive = false;
        else throughput = throughput / q;
    }
    if (!(depth <= max_depth)) alive = false;                           // `while (depth <= m_max_depth)`
    // exact shortcut: an absorbed path (throughput exactly 0) can add nothing any more; the source keeps scattering it until the
    // roulette catches it.  Retired now unless a later pass continues the lane's random-number stream (rp.pass_out).
    if (!rp.pass_out && !any_nonzero(throughput)) alive = false;
    commit();
    // ---- look-ahead (exact): the next trip's first draw is its free-flight sample.  The competition is run now on a copy of
    // the generator and kept in the record; when the distance field proves that no surface lies within its distance, the
    // next trip's interaction is inside the medium whatever the ray query would return (`distance < si.t`), and the path
    // is queued in region A: no query at all.
    s.bio_dist = u2f(0x7fc00000u); s.bio_hep = false;
    if (alive && medium >= 0 && sc.grid.enabled) {
        SMP pk = rng;
        int type2; float dist2;
        bio_compute_distance(sc.bio[medium], channel, pk.next(), tissue_depth, type2, dist2);
        s.bio_dist = dist2 == dist2 ? dist2 : kInf; s.bio_hep = type2 == BIO_ABSORBER_AND_ATTENUATOR;
        if (dist2 > 0.f && dist2 < kInf && segment_free_of_surfaces(sc.grid, ray.o, ray.d, dist2)) s.flags |= PF_NOHIT;
    }
    return alive;
}

} // namespace lrt
