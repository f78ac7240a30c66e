/*
 * orc_api.cpp -- scene creation (deep copy of the POD description) and
 * parameter edits of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see orc.h).
 */
#include "orc_scene.h"
#include <cstring>
#include <cstdio>

using namespace orc;

extern "C" orc_scene *orc_scene_create(const lrt_scene_desc *desc) {
    orc_scene *o = new orc_scene();
    Scene &S = o->s;
    S.d = *desc;
    S.positions.assign(desc->positions, desc->positions + 3 * (size_t) desc->n_vertices);
    S.normals.assign(desc->normals, desc->normals + 3 * (size_t) desc->n_vertices);
    S.texcoords.assign(desc->texcoords, desc->texcoords + 2 * (size_t) desc->n_vertices);
    S.faces.assign(desc->faces, desc->faces + 3 * (size_t) desc->n_faces);
    S.face_shape.assign(desc->face_shape, desc->face_shape + desc->n_faces);
    S.shapes.assign(desc->shapes, desc->shapes + desc->n_shapes);
    S.bsdfs.assign(desc->bsdfs, desc->bsdfs + desc->n_bsdfs);
    S.textures.assign(desc->textures, desc->textures + desc->n_textures);
    S.texdata.resize(desc->n_textures);
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        lrt_texture_desc &T = S.textures[i];
        if (T.type == LRT_TEX_BITMAP && T.data) {
            S.texdata[i].assign(T.data, T.data + (size_t) T.width * T.height * T.channels);
            T.data = S.texdata[i].data();
        }
    }
    S.media.assign(desc->media, desc->media + desc->n_media);
    S.meddata.resize(desc->n_media);
    for (uint32_t i = 0; i < desc->n_media; ++i) {
        lrt_medium_desc &M = S.media[i];
        if (M.type == LRT_MEDIUM_HETEROGENEOUS && M.grid_data) {
            S.meddata[i].assign(M.grid_data, M.grid_data + (size_t) M.grid_res[0] * M.grid_res[1] * M.grid_res[2]);
            M.grid_data = S.meddata[i].data();
        }
    }
    S.emitters.assign(desc->emitters, desc->emitters + desc->n_emitters);
    S.emdata.resize(desc->n_emitters);
    for (uint32_t i = 0; i < desc->n_emitters; ++i) {
        lrt_emitter_desc &E = S.emitters[i];
        if (E.type == LRT_EMITTER_ENVMAP && E.data) {
            S.emdata[i].assign(E.data, E.data + (size_t) E.width * E.height * 3);
            E.data = S.emdata[i].data();
        }
    }
    S.d.positions = S.positions.data(); S.d.normals = S.normals.data(); S.d.texcoords = S.texcoords.data();
    S.d.faces = S.faces.data(); S.d.face_shape = S.face_shape.data(); S.d.shapes = S.shapes.data();
    S.d.bsdfs = S.bsdfs.data(); S.d.textures = S.textures.data(); S.d.media = S.media.data(); S.d.emitters = S.emitters.data();
    S.finalize();
    return o;
}

extern "C" void orc_scene_free(orc_scene *s) { delete s; }
extern "C" void orc_scene_set_bio_reading(orc_scene *s, int scalar) { s->s.bio_scalar = scalar != 0; }

/* keys: "<medium id>.sigma_t.value", ".albedo.value", ".scale", ".phase_function.g"
   (src/media/homogeneous.cpp:146-151, src/phase/hg.cpp:60-62) */
extern "C" int orc_param_set(orc_scene *s, const char *key, const float *v, int n) {
    Scene &S = s->s;
    for (auto &M : S.media) {
        size_t L = strlen(M.id);
        if (strncmp(key, M.id, L) != 0 || key[L] != '.') continue;
        const char *rest = key + L + 1;
        if (!strcmp(rest, "sigma_t.value")) { for (int i = 0; i < 3; ++i) M.sigma_t[i] = v[n == 3 ? i : 0]; return 0; }
        if (!strcmp(rest, "albedo.value")) { for (int i = 0; i < 3; ++i) M.albedo[i] = v[n == 3 ? i : 0]; return 0; }
        if (!strcmp(rest, "scale")) { M.scale = v[0]; return 0; }
        if (!strcmp(rest, "phase_function.g")) { M.g = v[0]; M.phase = LRT_PHASE_HG; return 0; }
    }
    return 1;
}
