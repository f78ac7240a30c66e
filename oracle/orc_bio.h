/*
 * orc_bio.h -- the fork's bio transport in the CPU oracle: the `liver`, `parenchyma` and `glissonCapsule` media
 * (element-competition free-flight sampling of the 5-argument Medium::sample_interaction) and the `biovolpath` /
 * `biovolpath06` integrators.  TEST INFRASTRUCTURE ONLY (see orc.h).  Included by orc_render.cpp inside namespace orc.
 * Specification with every decision: docs/BIO_TRANSPORT_SPEC.md.  Paths are relative to the reference tree.
 *
 * Two readings of the same source exist in the reference, because `dr::any_or<true>(mask)` is the constant `true` for
 * JIT arrays and the value of the mask for scalars (so `if / else if` chains keyed on it lose their `else` branches in
 * the JIT variants), and `dr::none_or<false>(mask)` never returns early there:
 *   jit == true    llvm_ad_rgb / cuda_rgb lane semantics (what `hip_ad_rgb` implements; default)
 *   jit == false   scalar_rgb semantics (the reference's CPU renders; `biovolpath06` only makes sense this way)
 *
 * PARITY PINNING: no numeric fixture for any of this exists in the reference tree ("parity unpinned by the reference").
 * Pins available: the committed CPU / GPU renders of Liver-SingleMesh as weak goldens (tests/test_bio_oracle.py), the
 * inner generator against published PCG32 vectors, closed-form checks of the element competition.
 */

/* src/media/organic_material.h:29-34 */
enum { BIO_ABSORBER = 0, BIO_ATTENUATOR = 1, BIO_ABSORBER_AND_ATTENUATOR = 2 };

struct BioMI { float t; V3 p; V3 transmittance; V3 sigma_t, combined; bool valid() const { return t != kInf; } };

/* `computeDistance` (liver.cpp:227-477, parenchyma.cpp:193-301, glissonCapsule.cpp:229-307).
   The inner generator is a fresh PCG32 seeded with the BIT PATTERN of the free-flight sample, default stream
   (liver.cpp:233-235: `rng.seed(reinterpret_array<UInt32>(sample))`, Dr.Jit PCG32_DEFAULT_STREAM). */
static inline void bio_element(PCG32 &rng, float att, bool hepatocyte, bool guard_positive, int i, int *element, float *distance) {
    float r = rng.next();
    if (r == 0.f) r = 0.5f;
    if (guard_positive && !(att > 0.f)) return;                     /* `attIndexPositive` masks the update */
    float aux = -(1.f / att) * m_log(r);                            /* -(1.0 / attIndex) * log(r) */
    if (hepatocyte) {                                               /* EAbsorberAndAttenuator: -(log10(att + 1) * log(r)) */
        float log10 = m_log2(att + 1.f) / m_log2(10.f);
        aux = -(log10 * m_log(r));
    }
    if (i == 0 || aux < *distance) { *element = i; *distance = aux; }
}

static void bio_compute_distance(const lrt_medium_desc &M, uint32_t channel, float sample, float depth, int *bio_type, float *dist_out) {
    float distance = kInf; int element = 0;
    PCG32 rng; rng.seed((uint64_t) f2u(sample), 0xda3e39cb94b95bdbULL);
    bool layered = M.type == LRT_MEDIUM_LIVER || M.type == LRT_MEDIUM_GLISSON;
    int layer = 0;
    if (layered) {                                                  /* liver.cpp:246-251: every later test overwrites the earlier ones */
        if (depth <= M.layer_limit[0]) layer = 0;
        if (depth <= M.layer_limit[1]) layer = 1;
        if (depth <= M.layer_limit[2]) layer = 2;
        if (depth <= M.layer_limit[3]) layer = 3;
        if (depth > M.layer_limit[3]) layer = 4;
    }
    if (layered && layer < 4) {                                     /* collagen vs elastin of the layer, both EAttenuator */
        for (int i = 0; i < 2; ++i) {
            const float *sg = i == 0 ? M.sigma_collagen[layer] : M.sigma_elastin[layer];
            /* glissonCapsule.cpp:293-300 wraps the update in `if (any_or<true>(attIndexPositive))` but masks it with
               activeLayer only: in the JIT reading nothing guards it, in the scalar one the block is skipped; the two only
               differ for non-positive coefficients, where both end with "no interaction" */
            bio_element(rng, sg[channel], false, false, i, &element, &distance);
        }
        *bio_type = BIO_ATTENUATOR; *dist_out = distance; return;
    }
    if (M.type == LRT_MEDIUM_GLISSON) { *bio_type = BIO_ATTENUATOR; *dist_out = kInf; return; }   /* below the capsule: nothing competes */
    /* parenchyma elements: blood, bile, lipid/water (EAbsorber), hepatocytes (EAbsorberAndAttenuator, scalar coefficient) */
    for (int i = 0; i < 4; ++i) {
        float att = i == 0 ? M.sigma_blood[channel] : i == 1 ? M.sigma_bile[channel] : i == 2 ? M.sigma_lipid_water[channel] : M.sigma_hepatocity;
        bio_element(rng, att, i == 3, true, i, &element, &distance);
    }
    *bio_type = element == 3 ? BIO_ABSORBER_AND_ATTENUATOR : BIO_ABSORBER;
    *dist_out = distance;
}

/* The homogeneous coefficients the bio media report (get_scattering_coefficients / get_majorant):
   liver / glissonCapsule: sigma_t * scale and albedo (liver.cpp:204-223); parenchyma: hard-coded (parenchyma.cpp:163-189) */
static inline void bio_coefficients(const lrt_medium_desc &M, V3 *sigmat, V3 *sigmas) {
    if (M.type == LRT_MEDIUM_PARENCHYMA) {
        *sigmat = V3(77.2f / 255, 105.0f / 255, 149.0f / 255); *sigmas = V3(74.0f / 255, 88.0f / 255, 101.0f / 255);
    } else {
        *sigmat = V3(M.sigma_t[0], M.sigma_t[1], M.sigma_t[2]) * M.scale;
        *sigmas = *sigmat * V3(M.albedo[0], M.albedo[1], M.albedo[2]);
    }
}

/* 5-argument sample_interaction (liver.cpp:479-539, parenchyma.cpp:303-368, glissonCapsule.cpp:309-353).  ray.maxt is the
   surface distance the integrator hands in (`Ray3f(ray, si.t)`). */
static BioMI bio_sample_interaction(const lrt_medium_desc &M, const Ray &ray, float sample, uint32_t channel, float depth, bool jit) {
    BioMI mei; mei.t = kInf; mei.p = V3(0.f); mei.transmittance = V3(1.f);
    int bio_type; float distance;
    bio_compute_distance(M, channel, sample, depth, &bio_type, &distance);
    const float dist_surf = ray.maxt;
    const bool valid_mi = distance <= ray.maxt;
    const bool inside = distance > 0.f && distance < dist_surf;       /* "Hit layer boundary: No" */
    if (inside) { mei.t = distance; mei.p = fma3(ray.d, mei.t, ray.o); }
    bool active = true;
    const bool onehot_only = M.type == LRT_MEDIUM_GLISSON;            /* glissonCapsule.cpp:331 `active = true;` */
    if (!onehot_only) {
        const bool chain = M.type == LRT_MEDIUM_PARENCHYMA && jit;    /* parenchyma.cpp:334-344: `if ... else if ... else if` on any_or<true> */
        if (bio_type == BIO_ABSORBER) active = false;
        if (!chain) {
            if (bio_type == BIO_ATTENUATOR) active = true;
            if (bio_type == BIO_ABSORBER_AND_ATTENUATOR && (double) distance < 0.0025) active = false;   /* Float64 r = 0.0025 */
        }
    }
    const V3 onehot = channel == 0 ? V3(1.f, 0.f, 0.f) : channel == 1 ? V3(0.f, 1.f, 0.f) : V3(0.f, 0.f, 1.f);
    if (M.type == LRT_MEDIUM_LIVER || !jit) {                         /* three plain ifs (liver) = the full chain in scalar mode */
        if (inside && active) mei.transmittance = onehot;
        if (inside && !active) mei.transmittance = V3(0.f);
        if (!inside) { mei.transmittance = V3(1.f); mei.t = kInf; }
    } else {                                                          /* parenchyma / glissonCapsule, JIT: only the first branch exists */
        if (inside && active) mei.transmittance = onehot;
    }
    V3 sigmat, sigmas; bio_coefficients(M, &sigmat, &sigmas);
    mei.sigma_t = valid_mi ? sigmat : V3(0.f);
    mei.combined = sigmat;
    return mei;
}

/* biovolpath.cpp:383-541 sample_emitter.  ref_n is zero for medium interactions (unused: biovolpath samples emitters at
   surfaces only). */
static V3 bio_sample_emitter(Ctx &C, V3 ref_p, V3 ref_n, const SI *ref_si, int medium, uint32_t channel, float tissue_depth, DirSample *ds_out) {
    const Scene &S = C.S;
    V3 transmittance(1.f);
    float sx, sy; C.next2(&sx, &sy);
    DirSample ds; V3 emitter_val = sample_emitter_direction(S, ref_p, sx, sy, &ds);
    *ds_out = ds;
    if (ds.pdf == 0.f) return V3(0.f);
    Ray ray = spawn_ray_to(ref_p, ref_n, ds.p);
    float max_dist = ray.maxt;
    if (ref_si && is_medium_transition(S.shapes[ref_si->shape])) medium = target_medium(S.shapes[ref_si->shape], ray.d, ref_si->n);
    float total_dist = 0.f;
    SI si; memset((void *) &si, 0, sizeof(si)); si.t = kInf;              /* dr::zeros<SurfaceInteraction3f>(): t = inf */
    bool needs_intersection = true, active = true;
    while (active) {
        float remaining_dist = max_dist - total_dist;
        ray.maxt = remaining_dist;
        if (!(remaining_dist > 0.f)) { C.skip(1); break; }
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (!active_medium) C.skip(1);                                    /* biovolpath.cpp:464 */
        if (active_medium) {
            const lrt_medium_desc &M = S.media[medium];
            Ray mr = ray; mr.maxt = si.t;                                 /* Ray3f(ray, si.t) */
            BioMI mei = bio_sample_interaction(M, mr, C.next(), channel, tissue_depth, C.bio_jit);
            if (mei.valid()) ray.maxt = fminf(mei.t, remaining_dist);
            if (needs_intersection) { C.n_shadow++; C.n_shadow_needed++; Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
            if (si.t < mei.t) mei.t = kInf;
            needs_intersection = false;
            if (M.has_spectral_extinction) {
                float t = fminf(remaining_dist, fminf(mei.t, si.t)) - 0.f;
                V3 tr(m_exp(-t * mei.combined.x), m_exp(-t * mei.combined.y), m_exp(-t * mei.combined.z));
                V3 ffp = (si.t < mei.t || mei.t > remaining_dist) ? tr : tr * mei.combined;
                float tr_pdf = idx3(ffp, channel);
                transmittance *= (tr_pdf > 0.f) ? tr / tr_pdf : V3(0.f);
            }
            if ((mei.t > remaining_dist) && mei.valid()) total_dist = ds.dist;
            if (mei.t > remaining_dist) mei.t = kInf;
            escaped_medium = !mei.valid();
            active_medium = mei.valid();
            if (active_medium) {
                total_dist += mei.t;
                ray.o = mei.p;
                si.t = si.t - mei.t;
                transmittance *= mei.transmittance;                       /* :500-503, both branches */
            }
        }
        bool intersect = active_surface && needs_intersection;
        if (intersect) { C.n_shadow++; C.n_shadow_needed++; Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); needs_intersection = false; }
        active_surface = active_surface || escaped_medium;
        if (active_surface) total_dist += si.t;
        active_surface = active_surface && si.valid && !active_medium;
        if (active_surface) {
            transmittance *= bsdf_null_transmission(S, S.shapes[si.shape].bsdf);
            ray = spawn_ray(si.p, si.n, ray.d);
        }
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        active = (active_medium || active_surface) && any_nonzero(transmittance);
        if (active_surface && is_medium_transition(S.shapes[si.shape])) medium = target_medium(S.shapes[si.shape], ray.d, si.n);
    }
    return transmittance * emitter_val;
}

/* biovolpath.cpp:95-379.  JIT reading (C.bio_jit): the loop body runs to its end for a lane that fails the termination
   tests at its top, `mei` is the (zero for masked lanes) return value of the virtual call in every trip, and the
   `result[mei.transmittance == 0] = 0` block is unconditional.  Scalar reading: early return, blocks keyed on the lane's
   own masks, `mei` persists across trips. */
static void biovolpath_sample(Ctx &C, Ray ray, int medium, V3 *out, bool *out_valid) {
    const Scene &S = C.S;
    const bool jit = C.bio_jit;
    bool valid_ray = !C.hide_emitters && S.env >= 0;
    float eta = 1.f;
    V3 throughput(1.f), result(0.f);
    bool specular_chain = !C.hide_emitters;
    uint32_t depth = 0;
    uint32_t channel = std::min((uint32_t) (C.next() * 3.f), 2u);
    SI si; memset((void *) &si, 0, sizeof(si)); si.t = kInf;
    bool active = true;
    V3 last_scatter_p(0.f);
    float last_scatter_direction_pdf = 1.f, tissue_depth = 0.f;
    V3 mei_transmittance(0.f);                                            /* scalar reading: mei is loop state, zero-initialised */
    bool mei_was_valid = false;
    const uint32_t max_depth = (uint32_t) C.max_depth;
    while (active) {
        C.n_iter++;
        active = any_nonzero(throughput);
        float q = fminf(max3(throughput) * sqr(eta), .95f);
        bool perform_rr = depth > (uint32_t) C.rr_depth;
        if (active) { float u = C.next(); active = (u < q) || !perform_rr; }
        if (perform_rr) throughput *= rcp(q);
        active = active && depth < max_depth;
        if (!active) {
            /* JIT: the masked virtual call returns zeros, so :297-300 clears every channel of the result */
            if (jit) { result = V3(0.f); if (medium >= 0) throughput = V3(0.f); }
            break;
        }
        bool active_medium = medium >= 0, active_surface = !active_medium;
        const bool in_medium_lane = active_medium;
        bool act_medium_scatter = false, escaped_medium = false;
        BioMI mei; mei.t = kInf; mei.p = V3(0.f); mei.transmittance = V3(0.f); mei.sigma_t = mei.combined = V3(0.f);
        if (!active_medium) C.skip(2);                                    /* :226, :244 */
        if (active_medium) {
            const lrt_medium_desc &M = S.media[medium];
            Ray mr = ray; mr.maxt = si.t;                                 /* Ray3f(ray, si.t): the PREVIOUS query's distance */
            mei = bio_sample_interaction(M, mr, C.next(), channel, tissue_depth, jit);
            if (mei.valid()) ray.maxt = mei.t;
            { Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
            if (si.t < mei.t) mei.t = kInf;
            if (M.has_spectral_extinction) {                              /* Medium::transmittance_eval_pdf, medium.cpp:92-104 */
                float t = fminf(mei.t, si.t) - 0.f;
                V3 tr(m_exp(-t * mei.combined.x), m_exp(-t * mei.combined.y), m_exp(-t * mei.combined.z));
                V3 pdf = (si.t < mei.t) ? tr : tr * mei.combined;
                float tr_pdf = idx3(pdf, channel);
                throughput *= (tr_pdf > 0.f) ? tr / tr_pdf : V3(0.f);
            }
            escaped_medium = !mei.valid();
            active_medium = mei.valid();
            if (!active_medium) C.skip(1);
            if (active_medium) {
                /* :244 null/real draw: sigma_t / combined is 1 (or NaN for a zero coefficient) for every valid interaction, so the
                   collision is always real; the oracle checks that instead of assuming it */
                bool null_scatter = C.next() >= idx3(mei.sigma_t, channel) / idx3(mei.combined, channel);
                if (null_scatter) { fprintf(stderr, "orc: biovolpath null collision (unreachable per specification)\n"); abort(); }
                act_medium_scatter = true;
                depth += 1;
                last_scatter_p = mei.p;
            }
            mei_transmittance = mei.transmittance; mei_was_valid = true;
        }
        active = active && depth < max_depth;
        act_medium_scatter = act_medium_scatter && active;
        if (!act_medium_scatter) C.skip(2);                               /* :283, :284 */
        if (act_medium_scatter) {
            const lrt_medium_desc &M = S.media[medium];
            throughput *= mei.transmittance;                              /* :268 / :272 */
            tissue_depth += fabsf(-ray.d.z * mei.t);                      /* |Frame3f::cos_theta(-ray.d) * mei.t| */
            float s1 = C.next(); (void) s1;
            float s2x, s2y; C.next2(&s2x, &s2y);
            V3 wo; float phase_pdf; phase_sample(M, -ray.d, s2x, s2y, &wo, &phase_pdf);
            if (phase_pdf > 0.f) {
                ray = spawn_ray(mei.p, V3(0.f), wo);
                last_scatter_direction_pdf = phase_pdf;
            }
        }
        /* --------------------- surface interactions --------------------- */
        active_surface = active_surface || escaped_medium;
        bool intersect = active_surface && !escaped_medium;
        /* :297-300 */
        {
            V3 T = jit ? (in_medium_lane ? mei.transmittance : V3(0.f)) : mei_transmittance;
            bool block = jit ? true : (medium >= 0);
            (void) mei_was_valid;
            if (block) {
                if (T.x == 0.f) result.x = 0.f;
                if (T.y == 0.f) result.y = 0.f;
                if (T.z == 0.f) result.z = 0.f;
                if (medium >= 0) throughput *= T;
            }
        }
        if (intersect) { Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h); }
        if (active_surface) {
            bool count_direct = depth == 0 || specular_chain;
            int emitter = si_emitter(S, si);
            bool active_e = emitter >= 0 && !(depth == 0 && C.hide_emitters);
            if (active_e) {
                float emitter_pdf = 1.f;
                if (!count_direct) emitter_pdf = pdf_emitter_direction(S, last_scatter_p, si, emitter);
                V3 emitted = emitter_eval(S, emitter, si);
                V3 contrib = count_direct ? throughput * emitted : throughput * mis_weight(last_scatter_direction_pdf, emitter_pdf) * emitted;
                result += contrib;
            }
        }
        active_surface = active_surface && si.valid;
        if (!active_surface) C.skip(3);                                   /* :390 (NEE), :348, :349 */
        if (active_surface) {
            const lrt_shape_desc &sd = S.shapes[si.shape];
            int b = sd.bsdf;
            int flags = bsdf_flags(S, b);
            bool active_e = (flags & F_SMOOTH) && (depth + 1 < max_depth);
            if (!active_e) C.skip(1);
            if (active_e) {
                DirSample ds;
                V3 emitted = bio_sample_emitter(C, si.p, si.n, &si, medium, channel, tissue_depth, &ds);
                V3 wo = si.sh.to_local(ds.d);
                V3 bsdf_val = bsdf_eval(S, b, si, wo);
                float bpdf = bsdf_pdf(S, b, si, wo);
                result += throughput * bsdf_val * mis_weight(ds.pdf, ds.delta ? 0.f : bpdf) * emitted;
            }
            float s1 = C.next(), s2x, s2y; C.next2(&s2x, &s2y);
            BSDFSample bs; V3 bsdf_val;
            bsdf_sample(S, b, si, s1, s2x, s2y, &bs, &bsdf_val);
            throughput *= bsdf_val;
            eta *= bs.eta;
            ray = spawn_ray(si.p, si.n, si.sh.to_world(bs.wo));
            bool non_null = !(bs.type & F_NULL);
            if (non_null) { depth += 1; last_scatter_p = si.p; last_scatter_direction_pdf = bs.pdf; valid_ray = true; }
            specular_chain = specular_chain || (non_null && (bs.type & F_DELTA));
            specular_chain = specular_chain && !(bs.type & F_SMOOTH);
            if (is_medium_transition(sd)) medium = target_medium(sd, ray.d, si.n);
        }
        active = active && (active_surface || active_medium);
    }
    *out = result; *out_valid = valid_ray;
}

/* biovolpath06.cpp:88-473: a scalar-variant integrator (C++ `while` / `break` on `dr::all_nested(...)` of per-path
   conditions).  Its recursion-type bits follow Mitsuba 0.6's volpath_simple.  Where the source dereferences a null
   BSDF (a ray that leaves the scene without qualifying for the emitter branch) the path ends here.  Scalar code
   executes only the sampler calls it reaches, so nothing is skipped for the low-discrepancy sampler's dimension counter. */
static void biovolpath06_sample(Ctx &C, Ray ray, int medium, V3 *out, bool *out_valid) {
    const Scene &S = C.S;
    const bool valid_ray = !C.hide_emitters && S.env >= 0;
    float eta = 1.f;
    V3 throughput(1.f), result(0.f);
    bool null_chain = !C.hide_emitters, scattered_chain = false;
    uint32_t depth = 0, type = 127;
    uint32_t channel = std::min((uint32_t) (C.next() * 3.f), 2u);
    Hit h0 = S.intersect(ray, false, false);
    SI si = S.compute_si(ray, h0);
    float tissue_depth = 0.f;
    /* uint32_t m_max_depth: -1 -> 2^32 - 1.  `depth` counts loop trips here and lives in 16 bits of the device's path
       record: both sides stop a path after 65535 trips (docs/BIO_TRANSPORT_SPEC.md) */
    const uint32_t max_depth = std::min((uint32_t) C.max_depth, 65534u);
    BioMI mei; mei.t = kInf; mei.p = V3(0.f); mei.transmittance = V3(0.f); mei.sigma_t = mei.combined = V3(0.f);
    while (depth <= max_depth) {
        C.n_iter++;
        const bool in_medium = medium >= 0;
        if (in_medium) {
            Ray mr = ray; mr.maxt = si.t;
            mei = bio_sample_interaction(S.media[medium], mr, C.next(), channel, tissue_depth, false);
        }
        if (in_medium && mei.valid()) {
            const lrt_medium_desc &M = S.media[medium];
            throughput *= mei.transmittance;
            float s1 = C.next(); (void) s1;
            float s2x, s2y; C.next2(&s2x, &s2y);
            V3 wo; float phase_pdf; phase_sample(M, -ray.d, s2x, s2y, &wo, &phase_pdf);
            /* throughput *= phase_weight (= 1) */
            tissue_depth += fabsf(-ray.d.z * mei.t);
            ray.o = mei.p; ray.d = wo; ray.maxt = kLargest;                /* Ray3f(mei.p, wo, time, wavelengths) */
            Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h);
            null_chain = false; scattered_chain = true;
        } else {
            if (in_medium) throughput *= mei.transmittance;
            if (!si.valid) {
                bool active_e = (scattered_chain || !C.hide_emitters) && (type & 0x0001u) && S.env >= 0;
                if (active_e) {
                    V3 contrib = throughput * emitter_eval(S, S.env, si);
                    if (in_medium) {                                      /* medium->transmittance_eval_pdf(mei, si, true) */
                        float t = fminf(mei.t, si.t) - 0.f;
                        contrib *= V3(m_exp(-t * mei.combined.x), m_exp(-t * mei.combined.y), m_exp(-t * mei.combined.z));
                    }
                    result += contrib;
                }
                break;                                                    /* emitter branch, or the null-BSDF dereference (see above) */
            }
            const lrt_shape_desc &sd = S.shapes[si.shape];
            int b = sd.bsdf;
            float s1 = C.next(), s2x, s2y; C.next2(&s2x, &s2y);
            BSDFSample bs; V3 bsdf_val;
            bsdf_sample(S, b, si, s1, s2x, s2y, &bs, &bsdf_val);
            if (bsdf_val.x == 0.f && bsdf_val.y == 0.f && bsdf_val.z == 0.f) break;
            uint32_t recursive_type = 0;
            if (depth + 1 < max_depth && (type & 0x0008u)) recursive_type |= (0x0002u | 0x0004u | 0x0008u | 0x0010u | 0x0020u | 0x0040u | 0x0200u);
            if (depth < max_depth && (type & 0x0004u) && (bs.type & F_DELTA) && (!(bs.type & F_NULL) || null_chain)) { recursive_type |= 0x0001u; null_chain = true; }
            else null_chain = null_chain && (bs.type & F_NULL);
            if (recursive_type == 0) break;
            type = recursive_type;
            V3 wo = si.sh.to_world(bs.wo);
            throughput *= bsdf_val;
            eta *= bs.eta;
            ray = spawn_ray(si.p, si.n, wo);
            if (is_medium_transition(sd)) medium = target_medium(sd, wo, si.n);
            Hit h = S.intersect(ray, false, false); si = S.compute_si(ray, h);
            scattered_chain = scattered_chain || !(bs.type & F_NULL);
        }
        if (depth++ > (uint32_t) C.rr_depth) {
            float q = fminf(max3(throughput) * sqr(eta), .95f);
            if (C.next() >= q) break;
            throughput = throughput / q;
        }
        /* Exact shortcut (both sides, docs/BIO_TRANSPORT_SPEC.md section 4): an absorbed path (throughput exactly 0) can add nothing
           any more; the source keeps scattering it until the roulette catches it (q = 0 once depth > rr_depth).  It is retired here
           unless a later pass continues this lane's random-number stream. */
        if (!C.stream_continues && !any_nonzero(throughput)) break;
    }
    *out = result; *out_valid = valid_ray;
}
