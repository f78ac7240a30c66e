// extern "C" entry points of libliverrt.so (see include/liverrt.h).
// No exception crosses the ABI: every call returns a status and records a
// thread-local message for lrt_last_error().
#include "host_scene.h"
#include "device_scene.h"
#include "image_io.h"
#include <cstring>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>

using namespace lrt;

static thread_local std::string g_error;
static lrt_status fail(lrt_status st, const std::string &msg) { g_error = msg; return st; }

#define LRT_TRY try {
#define LRT_CATCH } catch (const std::exception &e) { \
        std::string m = e.what(); \
        lrt_status st = LRT_ERR_INVALID; \
        if (m.find("hip") != std::string::npos || m.find("HIP") != std::string::npos) st = LRT_ERR_DEVICE; \
        if (m.find("cannot open") != std::string::npos || m.find("file not found") != std::string::npos) st = LRT_ERR_IO; \
        if (m.find("unsupported") != std::string::npos || m.find("not supported") != std::string::npos) st = LRT_ERR_UNSUPPORTED; \
        return fail(st, m); \
    } catch (...) { return fail(LRT_ERR_INVALID, "unknown error"); }

extern "C" {

const char *lrt_last_error(void) { return g_error.c_str(); }
int lrt_version(void) { return 104; }    // 1.2: bio media fields in lrt_medium_desc, biovolpath integrators, grad_medium in lrt_render_opts; 1.3: lrt_render_stats.lds_resident; 1.4: lrt_render_multi / lrt_render_backward_multi, PRB through heterogeneous media

static std::vector<std::pair<std::string, std::string>> parse_defines(const char *const *defines, int n) {
    std::vector<std::pair<std::string, std::string>> r;
    for (int i = 0; i < n; ++i) {
        std::string s = defines[i]; size_t eq = s.find('=');
        if (eq == std::string::npos) throw std::runtime_error("define \"" + s + "\" must have the form key=value");
        r.emplace_back(s.substr(0, eq), s.substr(eq + 1));
    }
    return r;
}

lrt_status lrt_scene_load_xml_string(const char *xml, const char *base_dir, const char *const *defines, int n_defines, lrt_scene **out) {
    if (!xml || !out) return fail(LRT_ERR_INVALID, "lrt_scene_load_xml_string: null argument");
    *out = nullptr;
    LRT_TRY
        std::unique_ptr<lrt_scene> s(new lrt_scene());
        load_scene_xml(xml, base_dir ? base_dir : "", parse_defines(defines, n_defines), s->st);
        *out = s.release();
        return LRT_OK;
    LRT_CATCH
}

lrt_status lrt_scene_load_xml(const char *path, const char *const *defines, int n_defines, lrt_scene **out) {
    if (!path || !out) return fail(LRT_ERR_INVALID, "lrt_scene_load_xml: null argument");
    *out = nullptr;
    std::ifstream f(path, std::ios::binary);
    if (!f) return fail(LRT_ERR_IO, std::string("cannot open \"") + path + "\"");
    std::stringstream ss; ss << f.rdbuf();
    std::string p = path, dir; size_t sl = p.rfind('/'); dir = sl == std::string::npos ? "." : p.substr(0, sl);
    return lrt_scene_load_xml_string(ss.str().c_str(), dir.c_str(), defines, n_defines, out);
}

// Every index, range and pointer of a caller-built description (the XML loader's own output passes by construction).
static void validate_desc(const lrt_scene_desc &d) {
    auto bad = [](const std::string &m) { throw std::runtime_error("lrt_scene_from_desc: " + m); };
    if ((d.n_vertices && (!d.positions || !d.normals || !d.texcoords)) || (d.n_faces && (!d.faces || !d.face_shape))) bad("null geometry array");
    if ((d.n_shapes && !d.shapes) || (d.n_bsdfs && !d.bsdfs) || (d.n_textures && !d.textures) || (d.n_media && !d.media) || (d.n_emitters && !d.emitters)) bad("null object array");
    for (uint32_t f = 0; f < 3 * d.n_faces; ++f) if (d.faces[f] >= d.n_vertices) bad("face references an invalid vertex");
    for (uint32_t f = 0; f < d.n_faces; ++f) if (d.face_shape[f] >= d.n_shapes) bad("face references an invalid shape");
    for (uint32_t i = 0; i < d.n_textures; ++i) {
        const lrt_texture_desc &T = d.textures[i];
        if (T.type < LRT_TEX_RGB || T.type > LRT_TEX_BITMAP) bad("invalid texture type");
        if (T.type == LRT_TEX_BITMAP && (!T.data || T.width < 1 || T.height < 1 || (T.channels != 1 && T.channels != 3))) bad("bitmap texture without texels or with invalid dimensions");
    }
    for (uint32_t i = 0; i < d.n_bsdfs; ++i) {
        const lrt_bsdf_desc &B = d.bsdfs[i];
        if (B.type < LRT_BSDF_DIFFUSE || B.type > LRT_BSDF_NULL) bad("invalid bsdf type");
        if (B.type == LRT_BSDF_DIFFUSE) {
            if (B.reflectance < 0 || (uint32_t) B.reflectance >= d.n_textures) bad("diffuse bsdf references an invalid texture");
            if (d.textures[B.reflectance].type == LRT_TEX_BITMAP) throw std::runtime_error("unsupported: a bitmap texture as diffuse reflectance (bitmaps are supported as bump-map heights only)");
        }
        if (B.type == LRT_BSDF_BUMPMAP) {
            if (B.nested < 0 || (uint32_t) B.nested >= d.n_bsdfs || d.bsdfs[B.nested].type == LRT_BSDF_BUMPMAP) bad("bumpmap references an invalid nested bsdf");
            if (B.texture < 0 || (uint32_t) B.texture >= d.n_textures) bad("bumpmap references an invalid texture");
        }
        if (B.type == LRT_BSDF_DIELECTRIC && !(B.eta > 0.f)) bad("dielectric with a non-positive relative index of refraction");
    }
    for (uint32_t i = 0; i < d.n_media; ++i) {
        const lrt_medium_desc &M = d.media[i];
        if (M.type < LRT_MEDIUM_HOMOGENEOUS || M.type > LRT_MEDIUM_HETEROGENEOUS) bad("invalid medium type");
        if (M.type == LRT_MEDIUM_HETEROGENEOUS && (!M.grid_data || M.grid_res[0] < 1 || M.grid_res[1] < 1 || M.grid_res[2] < 1)) bad("heterogeneous medium without grid data");
        if (M.phase != LRT_PHASE_ISOTROPIC && M.phase != LRT_PHASE_HG) bad("invalid phase function");
        if (M.phase == LRT_PHASE_HG && !(M.g > -1.f && M.g < 1.f)) bad("The asymmetry parameter must lie in the interval (-1, 1)!");
    }
    for (uint32_t i = 0; i < d.n_shapes; ++i) {
        const lrt_shape_desc &s = d.shapes[i];
        if (s.bsdf < 0 || (uint32_t) s.bsdf >= d.n_bsdfs) bad("shape references an invalid bsdf");
        if (s.emitter < -1 || s.emitter >= (int) d.n_emitters || s.interior_medium < -1 || s.interior_medium >= (int) d.n_media || s.exterior_medium < -1 || s.exterior_medium >= (int) d.n_media)
            bad("shape references an invalid emitter/medium");
        if ((uint64_t) s.first_face + s.n_faces > d.n_faces) bad("shape face range exceeds the face array");
        if (s.emitter >= 0 && (d.emitters[s.emitter].type != LRT_EMITTER_AREA || d.emitters[s.emitter].shape != (int) i)) bad("shape and area emitter do not reference each other");
    }
    uint32_t n_env = 0;
    for (uint32_t i = 0; i < d.n_emitters; ++i) {
        const lrt_emitter_desc &E = d.emitters[i];
        if (E.type < LRT_EMITTER_AREA || E.type > LRT_EMITTER_CONSTANT) bad("invalid emitter type");
        if (E.type == LRT_EMITTER_AREA) {
            if (E.shape < 0 || (uint32_t) E.shape >= d.n_shapes) bad("area emitter references an invalid shape");
            const lrt_shape_desc &s = d.shapes[E.shape];
            if (s.kind != LRT_SHAPE_RECTANGLE || s.n_faces < 1) bad("area emitters are supported on rectangle shapes only");
        } else ++n_env;
        if (E.type == LRT_EMITTER_ENVMAP && (!E.data || E.width < 2 || E.height < 3)) bad("environment map without data or smaller than 2x3 pixels");
    }
    if (n_env > 1) bad("Only one environment emitter can be specified per scene.");
    if (d.sensor.medium < -1 || d.sensor.medium >= (int) d.n_media) bad("sensor references an invalid medium");
    if (!(d.sensor.near_clip > 0.f) || !(d.sensor.near_clip < d.sensor.far_clip)) bad("invalid clipping planes");
    if (!(d.sensor.fov_x > 0.f && d.sensor.fov_x < 180.f)) bad("The horizontal field of view must be in the range [0, 180]!");
    const lrt_film_desc &F = d.film;
    if (F.width <= 0 || F.height <= 0 || F.crop_width <= 0 || F.crop_height <= 0 || F.crop_offset_x < 0 || F.crop_offset_y < 0 ||
        F.crop_offset_x + F.crop_width > F.width || F.crop_offset_y + F.crop_height > F.height) bad("Invalid crop window specification!");
    if (F.rfilter < LRT_RFILTER_BOX || F.rfilter > LRT_RFILTER_TENT || (F.rfilter != LRT_RFILTER_BOX && !(F.rfilter_param > 0.f))) bad("invalid reconstruction filter");
    if (d.integrator.type < LRT_INTEGRATOR_PATH || d.integrator.type > LRT_INTEGRATOR_VOLPATHMIS) bad("invalid integrator type");
    if (d.integrator.max_depth < -1 || d.integrator.rr_depth <= 0) bad("invalid max_depth / rr_depth");
    if (d.sampler_type > LRT_SAMPLER_LD || d.sample_count == 0) bad("invalid sampler");
}

lrt_status lrt_scene_from_desc(const lrt_scene_desc *desc, lrt_scene **out) {
    if (!desc || !out) return fail(LRT_ERR_INVALID, "lrt_scene_from_desc: null argument");
    *out = nullptr;
    LRT_TRY
        validate_desc(*desc);
        std::unique_ptr<lrt_scene> s(new lrt_scene());
        s->st.copy_from(*desc);
        *out = s.release();
        return LRT_OK;
    LRT_CATCH
}

const lrt_scene_desc *lrt_scene_desc_get(const lrt_scene *scene) { return scene ? &scene->st.desc : nullptr; }

static void multi_release(lrt_scene *s) {
    if (s->multi_ctx) { multi_context_destroy(s->multi_ctx); s->multi_ctx = nullptr; }
    for (auto *D : s->multi) device_scene_destroy(D);
    s->multi.clear(); s->multi_ids.clear();
}

void lrt_scene_free(lrt_scene *scene) {
    if (!scene) return;
    multi_release(scene);
    if (scene->dev) device_scene_destroy(scene->dev);
    delete scene;
}

// One device image per entry of the device list (device_ids == NULL: devices 0 .. n - 1), kept until another list is asked for.
static void ensure_devices(lrt_scene *s, int n, const int *ids) {
    if (n < 1 || n > 64) throw std::invalid_argument("lrt_render_multi: n_devices must be 1 .. 64");
    std::vector<int> want(n); for (int i = 0; i < n; ++i) want[i] = ids ? ids[i] : i;
    if (want != s->multi_ids) {
        multi_release(s);
        for (int dev : want) s->multi.push_back(device_scene_create(s->st.desc, dev));
        s->multi_ids = want; s->multi_params_dirty = false;
    } else if (s->multi_params_dirty) { for (auto *D : s->multi) device_scene_update_params(D, s->st.desc); s->multi_params_dirty = false; }
}

// device < 0: whatever device the image already lives on (0 when there is none yet).  A render that names another device
// than the current one moves the scene there (one process per GPU is the intended use; this keeps a mistake from reading
// another GPU's pointers).
static void ensure_device(lrt_scene *s, int device) {
    if (s->dev && device >= 0 && device != s->dev_ordinal) { device_scene_destroy(s->dev); s->dev = nullptr; }
    if (!s->dev) { s->dev_ordinal = device < 0 ? 0 : device; s->dev = device_scene_create(s->st.desc, s->dev_ordinal); s->params_dirty = false; }
    else if (s->params_dirty) { device_scene_update_params(s->dev, s->st.desc); s->params_dirty = false; }
}

lrt_status lrt_render(lrt_scene *scene, const lrt_render_opts *opts, float *film_raw, float *image) {
    if (!scene) return fail(LRT_ERR_INVALID, "lrt_render: null scene");
    LRT_TRY
        ensure_device(scene, opts ? opts->device : 0);
        device_render(scene->dev, scene->st.desc, opts, film_raw, image, scene->stats);
        return LRT_OK;
    LRT_CATCH
}

lrt_status lrt_render_multi(lrt_scene *scene, const lrt_render_opts *opts, int n_devices, const int *device_ids, float *film_raw, float *image) {
    if (!scene) return fail(LRT_ERR_INVALID, "lrt_render_multi: null scene");
    LRT_TRY
        if (opts && (opts->tile_count > 1 || opts->tile_rank != 0)) throw std::invalid_argument("lrt_render_multi shards the image itself: tile_rank / tile_count must be 0 / 1 (or 0 / 0)");
        ensure_devices(scene, n_devices, device_ids);
        device_render_multi(scene->multi, scene->multi_ctx, scene->st.desc, opts, film_raw, image, scene->stats);
        return LRT_OK;
    LRT_CATCH
}

lrt_status lrt_render_backward_multi(lrt_scene *scene, const lrt_render_opts *opts, int n_devices, const int *device_ids, const float *grad_image, lrt_param_grads *out) {
    if (!scene || !grad_image || !out) return fail(LRT_ERR_INVALID, "lrt_render_backward_multi: null argument");
    LRT_TRY
        if (opts && (opts->tile_count > 1 || opts->tile_rank != 0)) throw std::invalid_argument("lrt_render_backward_multi shards the image itself: tile_rank / tile_count must be 0 / 1 (or 0 / 0)");
        ensure_devices(scene, n_devices, device_ids);
        device_render_backward_multi(scene->multi, scene->multi_ctx, scene->st.desc, opts, grad_image, out, scene->stats);
        return LRT_OK;
    LRT_CATCH
}

lrt_status lrt_math_eval(int fn, const float *x, const float *y, uint32_t n, float *out, float *out2, int device) {
    if (!x || !out) return fail(LRT_ERR_INVALID, "lrt_math_eval: null argument");
    LRT_TRY
        device_math_eval(fn, x, y, n, out, out2, device);
        return LRT_OK;
    LRT_CATCH
}

lrt_status lrt_render_stats_get(const lrt_scene *scene, lrt_render_stats *out) {
    if (!scene || !out) return fail(LRT_ERR_INVALID, "lrt_render_stats_get: null argument");
    *out = scene->stats; return LRT_OK;
}

lrt_status lrt_film_develop(lrt_scene *scene, const float *film_raw, float *image, int on_device) {
    if (!scene || !film_raw || !image) return fail(LRT_ERR_INVALID, "lrt_film_develop: null argument");
    LRT_TRY
        ensure_device(scene, -1);
        device_develop(scene->dev, film_raw, image, on_device);
        return LRT_OK;
    LRT_CATCH
}

lrt_status lrt_render_samples(lrt_scene *scene, const lrt_render_opts *opts, uint64_t lane_begin, uint32_t n, float *out) {
    if (!scene || !out) return fail(LRT_ERR_INVALID, "lrt_render_samples: null argument");
    LRT_TRY
        ensure_device(scene, opts ? opts->device : 0);
        device_render_samples(scene->dev, scene->st.desc, opts, lane_begin, n, out, scene->stats);
        return LRT_OK;
    LRT_CATCH
}

lrt_status lrt_render_backward(lrt_scene *scene, const lrt_render_opts *opts, const float *grad_image, lrt_param_grads *out) {
    if (!scene || !grad_image || !out) return fail(LRT_ERR_INVALID, "lrt_render_backward: null argument");
    LRT_TRY
        ensure_device(scene, opts ? opts->device : 0);
        device_render_backward(scene->dev, scene->st.desc, opts, grad_image, out, scene->stats);
        return LRT_OK;
    LRT_CATCH
}

lrt_status lrt_trace(lrt_scene *scene, const lrt_rays_soa *rays, const lrt_hits_soa *hits, uint32_t n, int any_hit) {
    if (!scene || !rays || !hits) return fail(LRT_ERR_INVALID, "lrt_trace: null argument");
    LRT_TRY
        ensure_device(scene, -1);
        device_trace(scene->dev, rays, hits, n, any_hit);
        return LRT_OK;
    LRT_CATCH
}

static lrt_medium_desc *find_medium(lrt_scene *s, const char *key, const char **rest) {
    for (auto &M : s->st.media) {
        size_t L = strlen(M.id);
        if (!strncmp(key, M.id, L) && key[L] == '.') { *rest = key + L + 1; return &M; }
    }
    return nullptr;
}

lrt_status lrt_param_set(lrt_scene *scene, const char *key, const float *v, int n) {
    if (!scene || !key || !v) return fail(LRT_ERR_INVALID, "lrt_param_set: null argument");
    const char *rest = nullptr; lrt_medium_desc *M = find_medium(scene, key, &rest);
    if (!M) return fail(LRT_ERR_INVALID, std::string("unknown parameter \"") + key + "\"");
    if (!strcmp(rest, "sigma_t.value") || !strcmp(rest, "albedo.value")) {
        if (n != 1 && n != 3) return fail(LRT_ERR_INVALID, "expected 1 or 3 values");
        float *dst = rest[0] == 's' ? M->sigma_t : M->albedo;
        for (int i = 0; i < 3; ++i) dst[i] = v[n == 3 ? i : 0];
    } else if (!strcmp(rest, "scale")) M->scale = v[0];
    else if (M->type == LRT_MEDIUM_PARENCHYMA && (!strcmp(rest, "sigma_blood.value") || !strcmp(rest, "sigma_bile.value") || !strcmp(rest, "sigma_lipid_water.value"))) {
        // what `parenchyma` puts into mi.traverse (src/media/parenchyma.cpp:154-160): the absorbers' coefficients and sigma_hepatocity
        if (n != 1 && n != 3) return fail(LRT_ERR_INVALID, "expected 1 or 3 values");
        float *dst;
        if (!strncmp(rest, "sigma_bile", 10)) dst = M->sigma_bile;
        else if (!strncmp(rest, "sigma_blood", 11)) dst = M->sigma_blood;
        else dst = M->sigma_lipid_water;
        for (int i = 0; i < 3; ++i) dst[i] = v[n == 3 ? i : 0];
    } else if (M->type == LRT_MEDIUM_PARENCHYMA && !strcmp(rest, "sigma_hepatocity")) M->sigma_hepatocity = v[0];
    else if (!strcmp(rest, "phase_function.g")) {
        // mi.traverse exposes `g` for an hg phase function only (src/phase/hg.cpp:60-62; isotropic.cpp has no parameter).  The
        // one extension kept from round 1: a NON-ZERO g on an isotropic medium turns it into hg ("HG variant ...
        // supplied through lrt_param_set"); g = 0 on an isotropic medium is rejected like any unknown key.
        if (!(v[0] > -1.f && v[0] < 1.f)) return fail(LRT_ERR_INVALID, "The asymmetry parameter must lie in the interval (-1, 1)!");
        if (M->phase != LRT_PHASE_HG && v[0] == 0.f) return fail(LRT_ERR_INVALID, std::string("unknown parameter \"") + key + "\" (the medium's phase function is isotropic)");
        M->g = v[0]; M->phase = LRT_PHASE_HG;
    } else return fail(LRT_ERR_INVALID, std::string("unknown parameter \"") + key + "\"");
    scene->params_dirty = true; scene->multi_params_dirty = true;
    return LRT_OK;
}

lrt_status lrt_param_get(const lrt_scene *scene, const char *key, float *v, int n) {
    if (!scene || !key || !v) return fail(LRT_ERR_INVALID, "lrt_param_get: null argument");
    const char *rest = nullptr; lrt_medium_desc *M = find_medium(const_cast<lrt_scene *>(scene), key, &rest);
    if (!M) return fail(LRT_ERR_INVALID, std::string("unknown parameter \"") + key + "\"");
    if (!strcmp(rest, "sigma_t.value")) { for (int i = 0; i < n && i < 3; ++i) v[i] = M->sigma_t[i]; }
    else if (!strcmp(rest, "albedo.value")) { for (int i = 0; i < n && i < 3; ++i) v[i] = M->albedo[i]; }
    else if (!strcmp(rest, "scale")) v[0] = M->scale;
    else if (!strcmp(rest, "phase_function.g")) v[0] = M->g;
    else if (M->type == LRT_MEDIUM_PARENCHYMA && !strcmp(rest, "sigma_blood.value")) { for (int i = 0; i < n && i < 3; ++i) v[i] = M->sigma_blood[i]; }
    else if (M->type == LRT_MEDIUM_PARENCHYMA && !strcmp(rest, "sigma_bile.value")) { for (int i = 0; i < n && i < 3; ++i) v[i] = M->sigma_bile[i]; }
    else if (M->type == LRT_MEDIUM_PARENCHYMA && !strcmp(rest, "sigma_lipid_water.value")) { for (int i = 0; i < n && i < 3; ++i) v[i] = M->sigma_lipid_water[i]; }
    else if (M->type == LRT_MEDIUM_PARENCHYMA && !strcmp(rest, "sigma_hepatocity")) v[0] = M->sigma_hepatocity;
    else return fail(LRT_ERR_INVALID, std::string("unknown parameter \"") + key + "\"");
    return LRT_OK;
}

lrt_status lrt_image_read(const char *path, int *width, int *height, int *channels, float **data) {
    if (!path || !width || !height || !channels || !data) return fail(LRT_ERR_INVALID, "lrt_image_read: null argument");
    *data = nullptr;
    LRT_TRY
        Image im = read_image_rgb(path);
        float *p = (float *) malloc(im.data.size() * sizeof(float));
        if (!p) throw std::runtime_error("out of memory");
        memcpy(p, im.data.data(), im.data.size() * sizeof(float));
        *width = im.width; *height = im.height; *channels = im.channels; *data = p;
        return LRT_OK;
    LRT_CATCH
}

void lrt_image_free(float *data) { free(data); }

// ---- learned subsurface model, network stage
struct lrt_vae_model { std::vector<float> blob; };

lrt_status lrt_vae_model_create(const float *blob, uint64_t n_floats, lrt_vae_model **out) {
    if (!blob || !out) return fail(LRT_ERR_INVALID, "lrt_vae_model_create: null argument");
    if (n_floats != (uint64_t) LRT_VAE_N_FLOATS) return fail(LRT_ERR_INVALID, "lrt_vae_model_create: expected " + std::to_string(LRT_VAE_N_FLOATS) + " floats (include/liverrt.h)");
    for (uint64_t i = 0; i < n_floats; ++i) if (!std::isfinite(blob[i])) return fail(LRT_ERR_INVALID, "lrt_vae_model_create: non-finite weight");
    lrt_vae_model *m = new lrt_vae_model(); m->blob.assign(blob, blob + n_floats); *out = m;
    return LRT_OK;
}
void lrt_vae_model_free(lrt_vae_model *model) { delete model; }

lrt_status lrt_vae_scatter(lrt_vae_model *model, uint32_t n, const float *in_pos, const float *in_dir, const float *poly_coeffs, const float albedo[3], float g, float ior,
                           const float sigma_t[3], float fit_scale, uint32_t seed, float *out_pos, float *out_absorption, int device) {
    if (!model || !albedo || !sigma_t || (n && (!in_pos || !in_dir || !poly_coeffs || !out_pos || !out_absorption))) return fail(LRT_ERR_INVALID, "lrt_vae_scatter: null argument");
    if (!(fit_scale > 0.f)) return fail(LRT_ERR_INVALID, "lrt_vae_scatter: fit_scale must be positive");
    LRT_TRY
        device_vae_scatter(model->blob.data(), n, in_pos, in_dir, poly_coeffs, albedo, g, ior, sigma_t, fit_scale, seed, out_pos, out_absorption, device);
        return LRT_OK;
    LRT_CATCH
}

lrt_status lrt_image_write_exr(const char *path, int width, int height, int channels, const float *data) {
    if (!path || !data) return fail(LRT_ERR_INVALID, "lrt_image_write_exr: null argument");
    LRT_TRY
        write_exr(path, width, height, channels, data);
        return LRT_OK;
    LRT_CATCH
}

lrt_status lrt_image_write_png(const char *path, int width, int height, int channels, const float *data) {
    if (!path || !data) return fail(LRT_ERR_INVALID, "lrt_image_write_png: null argument");
    LRT_TRY
        write_png(path, width, height, channels, data);
        return LRT_OK;
    LRT_CATCH
}

} // extern "C"
