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
int lrt_version(void) { return 101; }    // 1.1: sampler_type / samples_per_pass in lrt_scene_desc, n_records in lrt_render_stats, lrt_image_write_png

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

lrt_status lrt_scene_from_desc(const lrt_scene_desc *desc, lrt_scene **out) {
    if (!desc || !out) return fail(LRT_ERR_INVALID, "lrt_scene_from_desc: null argument");
    *out = nullptr;
    LRT_TRY
        for (uint32_t f = 0; f < 3 * desc->n_faces; ++f) if (desc->faces[f] >= desc->n_vertices) throw std::runtime_error("face references an invalid vertex");
        for (uint32_t f = 0; f < desc->n_faces; ++f) if (desc->face_shape[f] >= desc->n_shapes) throw std::runtime_error("face references an invalid shape");
        for (uint32_t i = 0; i < desc->n_shapes; ++i) {
            const lrt_shape_desc &s = desc->shapes[i];
            if (s.bsdf < 0 || (uint32_t) s.bsdf >= desc->n_bsdfs) throw std::runtime_error("shape references an invalid bsdf");
            if (s.emitter >= (int) desc->n_emitters || s.interior_medium >= (int) desc->n_media || s.exterior_medium >= (int) desc->n_media) throw std::runtime_error("shape references an invalid emitter/medium");
        }
        std::unique_ptr<lrt_scene> s(new lrt_scene());
        s->st.copy_from(*desc);
        *out = s.release();
        return LRT_OK;
    LRT_CATCH
}

const lrt_scene_desc *lrt_scene_desc_get(const lrt_scene *scene) { return scene ? &scene->st.desc : nullptr; }

void lrt_scene_free(lrt_scene *scene) {
    if (!scene) return;
    if (scene->dev) device_scene_destroy(scene->dev);
    delete scene;
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
    else if (!strcmp(rest, "phase_function.g")) {
        if (!(v[0] > -1.f && v[0] < 1.f)) return fail(LRT_ERR_INVALID, "The asymmetry parameter must lie in the interval (-1, 1)!");
        M->g = v[0]; M->phase = LRT_PHASE_HG;
    } else return fail(LRT_ERR_INVALID, std::string("unknown parameter \"") + key + "\"");
    scene->params_dirty = true;
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
