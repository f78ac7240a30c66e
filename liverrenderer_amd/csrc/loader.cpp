// Scene loading: Mitsuba XML subset -> flattened POD scene description.
//
// Replaces, for the plugins the liver scenes and mi.cornell_box() use, the
// reference's src/core/parser.cpp + Properties + PluginManager instantiation.
// Supported plugins: integrator {path, volpath, prbvolpath, biovolpath, biovolpath06}; sensor perspective;
// sampler {independent, ldsampler}; film hdrfilm; rfilter {box,
// gaussian, tent}; bsdf {diffuse, dielectric, bumpmap, null}; texture {bitmap,
// checkerboard}; medium {homogeneous, liver, parenchyma, glissonCapsule} (the
// bio media carry both the homogeneous parameters that path / volpath /
// prbvolpath see through the 4-argument sample_interaction() and the element
// coefficients of the 5-argument one, docs/BIO_TRANSPORT_SPEC.md); phase
// {isotropic, hg}; shape {obj, rectangle, cube}; emitter {area, envmap, constant}.
#include "host_scene.h"
#include "xml.h"
#include "image_io.h"
#include <map>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <stdexcept>
#include <algorithm>

namespace lrt {

void SceneStorage::fix_pointers() {
    desc.n_vertices = (uint32_t) (positions.size() / 3); desc.n_faces = (uint32_t) (faces.size() / 3);
    desc.n_shapes = (uint32_t) shapes.size(); desc.n_bsdfs = (uint32_t) bsdfs.size(); desc.n_textures = (uint32_t) textures.size();
    desc.n_media = (uint32_t) media.size(); desc.n_emitters = (uint32_t) emitters.size();
    texdata.resize(textures.size()); emdata.resize(emitters.size()); meddata.resize(media.size());
    for (size_t i = 0; i < media.size(); ++i) media[i].grid_data = meddata[i].empty() ? nullptr : meddata[i].data();
    for (size_t i = 0; i < textures.size(); ++i) textures[i].data = texdata[i].empty() ? nullptr : texdata[i].data();
    for (size_t i = 0; i < emitters.size(); ++i) emitters[i].data = emdata[i].empty() ? nullptr : emdata[i].data();
    desc.positions = positions.data(); desc.normals = normals.data(); desc.texcoords = texcoords.data();
    desc.faces = faces.data(); desc.face_shape = face_shape.data(); desc.shapes = shapes.data(); desc.bsdfs = bsdfs.data();
    desc.textures = textures.data(); desc.media = media.data(); desc.emitters = emitters.data();
}

void SceneStorage::copy_from(const lrt_scene_desc &d) {
    desc = d;
    positions.assign(d.positions, d.positions + 3 * (size_t) d.n_vertices);
    normals.assign(d.normals, d.normals + 3 * (size_t) d.n_vertices);
    texcoords.assign(d.texcoords, d.texcoords + 2 * (size_t) d.n_vertices);
    faces.assign(d.faces, d.faces + 3 * (size_t) d.n_faces);
    face_shape.assign(d.face_shape, d.face_shape + d.n_faces);
    shapes.assign(d.shapes, d.shapes + d.n_shapes);
    bsdfs.assign(d.bsdfs, d.bsdfs + d.n_bsdfs);
    textures.assign(d.textures, d.textures + d.n_textures);
    media.assign(d.media, d.media + d.n_media);
    emitters.assign(d.emitters, d.emitters + d.n_emitters);
    texdata.assign(textures.size(), {}); emdata.assign(emitters.size(), {}); meddata.assign(media.size(), {});
    for (size_t i = 0; i < media.size(); ++i)
        if (media[i].type == LRT_MEDIUM_HETEROGENEOUS && media[i].grid_data)
            meddata[i].assign(media[i].grid_data, media[i].grid_data + (size_t) media[i].grid_res[0] * media[i].grid_res[1] * media[i].grid_res[2]);
    for (size_t i = 0; i < textures.size(); ++i)
        if (textures[i].type == LRT_TEX_BITMAP && textures[i].data)
            texdata[i].assign(textures[i].data, textures[i].data + (size_t) textures[i].width * textures[i].height * textures[i].channels);
    for (size_t i = 0; i < emitters.size(); ++i)
        if (emitters[i].type == LRT_EMITTER_ENVMAP && emitters[i].data)
            emdata[i].assign(emitters[i].data, emitters[i].data + (size_t) emitters[i].width * emitters[i].height * 3);
    fix_pointers();
}

namespace {

// ------------------------------------------------------------------ math
struct Mat4 { double m[16]; };
Mat4 ident() { Mat4 r; for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0 : 0.0; return r; }
Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += a.m[4 * i + k] * b.m[4 * k + j]; r.m[4 * i + j] = s; } return r;
}
Mat4 inverse_affine(const Mat4 &a) {
    const double *m = a.m;
    double c00 = m[5] * m[10] - m[6] * m[9], c01 = m[6] * m[8] - m[4] * m[10], c02 = m[4] * m[9] - m[5] * m[8];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02, id = 1.0 / det;
    Mat4 r = ident();
    r.m[0] = c00 * id; r.m[1] = (m[2] * m[9] - m[1] * m[10]) * id; r.m[2] = (m[1] * m[6] - m[2] * m[5]) * id;
    r.m[4] = c01 * id; r.m[5] = (m[0] * m[10] - m[2] * m[8]) * id; r.m[6] = (m[2] * m[4] - m[0] * m[6]) * id;
    r.m[8] = c02 * id; r.m[9] = (m[1] * m[8] - m[0] * m[9]) * id; r.m[10] = (m[0] * m[5] - m[1] * m[4]) * id;
    for (int i = 0; i < 3; ++i) r.m[4 * i + 3] = -(r.m[4 * i] * m[3] + r.m[4 * i + 1] * m[7] + r.m[4 * i + 2] * m[11]);
    return r;
}
struct F3 { float x, y, z; };
// include/mitsuba/core/transform.h:296-309 / :261-290 in single precision
F3 xf_point(const float *m, F3 p) {
    return { fmaf(m[2], p.z, fmaf(m[1], p.y, fmaf(m[0], p.x, m[3]))), fmaf(m[6], p.z, fmaf(m[5], p.y, fmaf(m[4], p.x, m[7]))),
             fmaf(m[10], p.z, fmaf(m[9], p.y, fmaf(m[8], p.x, m[11]))) };
}
F3 xf_normal(const float *it, F3 n) {       // it = inverse transpose (row-major)
    F3 r = { fmaf(it[2], n.z, fmaf(it[1], n.y, it[0] * n.x)), fmaf(it[6], n.z, fmaf(it[5], n.y, it[4] * n.x)),
             fmaf(it[10], n.z, fmaf(it[9], n.y, it[8] * n.x)) };
    float il = 1.f / sqrtf(fmaf(r.z, r.z, fmaf(r.y, r.y, r.x * r.x)));
    return { r.x * il, r.y * il, r.z * il };
}
void to_float(const Mat4 &a, float *o) { for (int i = 0; i < 16; ++i) o[i] = (float) a.m[i]; }
void inv_transpose_float(const Mat4 &a, float *o) { Mat4 inv = inverse_affine(a); for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) o[4 * i + j] = (float) inv.m[4 * j + i]; }

float float_to_half_to_float(float f) {      // round-to-nearest-even through binary16
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = x & 0x80000000u, ax = x & 0x7fffffffu;
    if (ax >= 0x47800000u) { uint32_t r = sign | (ax > 0x7f800000u ? 0x7fc00000u : 0x7f800000u); float o; memcpy(&o, &r, 4); return o; }
    if (ax < 0x38800000u) {                  // subnormal half: quantum 2^-24
        float q = ldexpf(nearbyintf(ldexpf(fabsf(f), 24)), -24); return sign ? -q : q;
    }
    uint32_t rem = ax & 0x1fffu, base = ax & ~0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (base & 0x2000u))) base += 0x2000u;
    uint32_t r = sign | base; float o; memcpy(&o, &r, 4); return o;
}
float srgb_to_linear(float v) { return v <= 0.04045f ? v * (1.f / 12.92f) : powf((v + 0.055f) * (1.f / 1.055f), 2.4f); }

// ------------------------------------------------------------ object tree
struct Obj;
using ObjP = std::shared_ptr<Obj>;
struct Prop { std::string tag; const XmlNode *node = nullptr; Mat4 xform; };
struct Obj {
    std::string tag, type, id, name;
    std::map<std::string, Prop> props;
    std::vector<std::pair<std::string, ObjP>> children;   // (name, object) in document order
};

struct Loader {
    std::string base_dir;
    std::map<std::string, std::string> vars;
    std::map<std::string, ObjP> by_id;
    SceneStorage &S;
    std::map<const Obj *, int> bsdf_ix, medium_ix, tex_ix;
    explicit Loader(SceneStorage &s) : S(s) {}

    [[noreturn]] void fail(const std::string &m) { throw std::runtime_error(m); }

    std::string subst(const std::string &v) {
        std::string o; size_t i = 0;
        while (i < v.size()) {
            if (v[i] == '$') {
                size_t j = i + 1; while (j < v.size() && (isalnum((unsigned char) v[j]) || v[j] == '_')) ++j;
                std::string k = v.substr(i + 1, j - i - 1);
                auto it = vars.find(k);
                if (it == vars.end()) fail("undefined parameter \"$" + k + "\" (pass it as a define)");
                o += it->second; i = j;
            } else o += v[i++];
        }
        return o;
    }
    std::string attr(const XmlNode &n, const char *k, const char *def = nullptr) {
        const std::string *v = n.find(k);
        if (!v) { if (def) return def; fail("<" + n.tag + ">: missing attribute \"" + k + "\""); }
        return subst(*v);
    }
    static std::vector<double> parse_list(const std::string &s) {
        std::vector<double> r; const char *p = s.c_str();
        while (*p) {
            while (*p && (isspace((unsigned char) *p) || *p == ',')) ++p;
            if (!*p) break;
            char *e; double v = strtod(p, &e);
            if (e == p || (*e && !isspace((unsigned char) *e) && *e != ',')) throw std::runtime_error("could not parse number list \"" + s + "\"");
            r.push_back(v); p = e;
        }
        return r;
    }
    // src/core/parser.cpp:627-638: string::stof<double> (the whole string must be a number, trailing blanks allowed:
    // src/core/string.cpp:39-65), stored as double and narrowed when the plugin fetches a ScalarFloat
    static float parse_f32(const std::string &s) {
        char *e; double v = strtod(s.c_str(), &e);
        bool ok = e != s.c_str();
        for (const char *p = e; ok && *p; ++p) if (*p != ' ' && *p != '\t') ok = false;
        if (!ok) throw std::runtime_error("could not parse floating point value \"" + s + "\"");
        return (float) v;
    }

    void vec3_attr(const XmlNode &n, double def, double out[3]) {
        if (n.has("value")) { auto v = parse_list(attr(n, "value")); if (v.size() == 1) out[0] = out[1] = out[2] = v[0]; else if (v.size() == 3) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; } else fail("<" + n.tag + ">: expected 1 or 3 values"); return; }
        out[0] = n.has("x") ? strtod(attr(n, "x").c_str(), nullptr) : def;
        out[1] = n.has("y") ? strtod(attr(n, "y").c_str(), nullptr) : def;
        out[2] = n.has("z") ? strtod(attr(n, "z").c_str(), nullptr) : def;
    }

    // src/core/parser.cpp:457-561: each operation is left-multiplied
    Mat4 parse_transform(const XmlNode &n) {
        Mat4 T = ident();
        for (auto &c : n.children) {
            Mat4 M = ident();
            if (c->tag == "translate") { double v[3]; vec3_attr(*c, 0.0, v); M.m[3] = v[0]; M.m[7] = v[1]; M.m[11] = v[2]; }
            else if (c->tag == "scale") { double v[3]; vec3_attr(*c, 1.0, v); M.m[0] = v[0]; M.m[5] = v[1]; M.m[10] = v[2]; }
            else if (c->tag == "rotate") {
                double a[3]; a[0] = c->has("x") ? strtod(attr(*c, "x").c_str(), nullptr) : 0; a[1] = c->has("y") ? strtod(attr(*c, "y").c_str(), nullptr) : 0;
                a[2] = c->has("z") ? strtod(attr(*c, "z").c_str(), nullptr) : 0;
                if (c->has("value")) { auto v = parse_list(attr(*c, "value")); if (v.size() != 3) fail("<rotate>: expected 3 values"); a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; }
                double ang = strtod(attr(*c, "angle").c_str(), nullptr) * M_PI / 180.0, s = sin(ang), co = cos(ang), t = 1 - co;
                double x = a[0], y = a[1], z = a[2];
                M.m[0] = co + x * x * t; M.m[1] = x * y * t - z * s; M.m[2] = x * z * t + y * s;
                M.m[4] = y * x * t + z * s; M.m[5] = co + y * y * t; M.m[6] = y * z * t - x * s;
                M.m[8] = z * x * t - y * s; M.m[9] = z * y * t + x * s; M.m[10] = co + z * z * t;
            } else if (c->tag == "lookat") {       // include/mitsuba/core/transform.h:177-205
                auto o = parse_list(attr(*c, "origin")), tg = parse_list(attr(*c, "target")), up = parse_list(attr(*c, "up", "0,1,0"));
                if (o.size() != 3 || tg.size() != 3 || up.size() != 3) fail("<lookat>: expected 3-vectors");
                auto norm3 = [](double *v) { double l = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; };
                double dir[3] = { tg[0] - o[0], tg[1] - o[1], tg[2] - o[2] }; norm3(dir);
                double left[3] = { up[1] * dir[2] - up[2] * dir[1], up[2] * dir[0] - up[0] * dir[2], up[0] * dir[1] - up[1] * dir[0] }; norm3(left);
                double nup[3] = { dir[1] * left[2] - dir[2] * left[1], dir[2] * left[0] - dir[0] * left[2], dir[0] * left[1] - dir[1] * left[0] };
                for (int i = 0; i < 3; ++i) { M.m[4 * i] = left[i]; M.m[4 * i + 1] = nup[i]; M.m[4 * i + 2] = dir[i]; M.m[4 * i + 3] = o[i]; }
            } else if (c->tag == "matrix") {
                auto v = parse_list(attr(*c, "value"));
                if (v.size() == 16) for (int i = 0; i < 16; ++i) M.m[i] = v[i];
                else if (v.size() == 9) { M = ident(); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M.m[4 * i + j] = v[3 * i + j]; }
                else fail("<matrix>: expected 9 or 16 values");
            } else fail("unsupported transform operation <" + c->tag + ">");
            T = mul(M, T);
        }
        return T;
    }

    ObjP parse_object(const XmlNode &n) {
        auto o = std::make_shared<Obj>();
        o->tag = n.tag; o->type = n.has("type") ? attr(n, "type") : ""; o->id = n.has("id") ? attr(n, "id") : ""; o->name = n.has("name") ? attr(n, "name") : "";
        for (auto &c : n.children) {
            const std::string &t = c->tag;
            if (t == "float" || t == "integer" || t == "string" || t == "boolean" || t == "rgb" || t == "spectrum" || t == "point" || t == "vector") {
                Prop p; p.tag = t; p.node = c.get(); o->props[attr(*c, "name")] = p;
            } else if (t == "transform") {
                Prop p; p.tag = t; p.node = c.get(); p.xform = parse_transform(*c); o->props[attr(*c, "name")] = p;
            } else if (t == "ref") {
                std::string id = attr(*c, "id"); auto it = by_id.find(id);
                if (it == by_id.end()) fail("reference to unknown id \"" + id + "\"");
                o->children.emplace_back(c->has("name") ? attr(*c, "name") : "", it->second);
            } else if (t == "default") { std::string k = attr(*c, "name"); if (!vars.count(k)) vars[k] = attr(*c, "value"); }
            else if (t == "alias" || t == "include" || t == "path") fail("<" + t + "> is not supported");
            else {
                ObjP ch = parse_object(*c);
                if (!ch->id.empty()) by_id[ch->id] = ch;
                o->children.emplace_back(ch->name, ch);
            }
        }
        return o;
    }

    // ---------------------------------------------------- property getters
    bool has(const Obj &o, const char *k) { return o.props.count(k) != 0; }
    float get_float(const Obj &o, const char *k, float def) {
        auto it = o.props.find(k); if (it == o.props.end()) return def;
        if (it->second.tag != "float" && it->second.tag != "integer") fail("property \"" + std::string(k) + "\" has the wrong type");
        return parse_f32(attr(*it->second.node, "value"));
    }
    int get_int(const Obj &o, const char *k, int def) {
        auto it = o.props.find(k); if (it == o.props.end()) return def;
        return (int) strtol(attr(*it->second.node, "value").c_str(), nullptr, 10);
    }
    bool get_bool(const Obj &o, const char *k, bool def) {
        auto it = o.props.find(k); if (it == o.props.end()) return def;
        std::string v = attr(*it->second.node, "value"); for (auto &c : v) c = (char) tolower(c);
        if (v == "true") return true; if (v == "false") return false;
        fail("could not parse boolean value \"" + v + "\"");
    }
    std::string get_string(const Obj &o, const char *k, const char *def) {
        auto it = o.props.find(k); if (it == o.props.end()) { if (!def) fail("missing property \"" + std::string(k) + "\" of <" + o.tag + " type=\"" + o.type + "\">"); return def; }
        return attr(*it->second.node, "value");
    }
    void get_rgb(const Obj &o, const char *k, float def, float out[3]) {
        auto it = o.props.find(k); if (it == o.props.end()) { out[0] = out[1] = out[2] = def; return; }
        auto v = parse_list(attr(*it->second.node, "value"));
        if (v.size() == 1) out[0] = out[1] = out[2] = (float) v[0];
        else if (v.size() == 3) { out[0] = (float) v[0]; out[1] = (float) v[1]; out[2] = (float) v[2]; }
        else fail("property \"" + std::string(k) + "\": expected 1 or 3 values");
    }
    Mat4 get_xform(const Obj &o, const char *k) { auto it = o.props.find(k); return it == o.props.end() ? ident() : it->second.xform; }
    ObjP child(const Obj &o, const char *tag, const char *name = nullptr) {
        for (auto &c : o.children) if (c.second->tag == tag && (!name || c.first == name)) return c.second;
        return nullptr;
    }
    std::string resolve(const std::string &f) { if (!f.empty() && f[0] == '/') return f; return base_dir.empty() ? f : base_dir + "/" + f; }

    // --------------------------------------------------------- plugins
    int make_rgb_texture(const float c[3]) {
        lrt_texture_desc T{}; T.type = LRT_TEX_RGB; for (int i = 0; i < 3; ++i) T.color0[i] = T.color1[i] = c[i];
        T.to_uv[0] = T.to_uv[4] = T.to_uv[8] = 1.f; S.textures.push_back(T); S.texdata.emplace_back(); return (int) S.textures.size() - 1;
    }
    void to_uv(const Obj &o, float out[9]) {
        Mat4 M = get_xform(o, "to_uv");      // parsed as a 4x4; a 2D affine map lives in x/y + translation
        out[0] = (float) M.m[0]; out[1] = (float) M.m[1]; out[2] = (float) M.m[3];
        out[3] = (float) M.m[4]; out[4] = (float) M.m[5]; out[5] = (float) M.m[7];
        out[6] = 0.f; out[7] = 0.f; out[8] = 1.f;
    }
    int make_texture(const ObjP &o) {
        auto it = tex_ix.find(o.get()); if (it != tex_ix.end()) return it->second;
        lrt_texture_desc T{}; std::vector<float> data;
        T.to_uv[0] = T.to_uv[4] = T.to_uv[8] = 1.f;
        if (o->type == "checkerboard") {       // src/textures/checkerboard.cpp:58-62
            T.type = LRT_TEX_CHECKERBOARD; get_rgb(*o, "color0", .4f, T.color0); get_rgb(*o, "color1", .2f, T.color1); to_uv(*o, T.to_uv);
        } else if (o->type == "bitmap") {      // src/textures/bitmap.cpp:176-360
            T.type = LRT_TEX_BITMAP; to_uv(*o, T.to_uv);
            std::string ft = get_string(*o, "filter_type", "bilinear"), wm = get_string(*o, "wrap_mode", "repeat");
            if (ft != "bilinear" || wm != "repeat") fail("bitmap texture: only bilinear/repeat is supported");
            bool raw = get_bool(*o, "raw", false);
            Image im = read_image_rgb(resolve(get_string(*o, "filename", nullptr)));
            int ch = im.channels >= 3 ? 3 : 1;   // RGB[A] -> RGB, Y[A] -> Y
            T.width = im.width; T.height = im.height; T.channels = ch;
            data.resize((size_t) im.width * im.height * ch);
            bool to_half = im.bits_per_channel <= 16;   // Format::Auto -> fp16 storage for <= 2 bytes/channel
            for (size_t p = 0; p < (size_t) im.width * im.height; ++p)
                for (int c = 0; c < ch; ++c) {
                    float v = im.data[p * im.channels + c];
                    if (im.srgb && !raw) v = srgb_to_linear(v);
                    data[p * ch + c] = to_half ? float_to_half_to_float(v) : v;
                }
        } else if (o->type == "srgb" || o->type == "rgb") {
            float c[3]; get_rgb(*o, "color", .5f, c); if (has(*o, "value")) get_rgb(*o, "value", .5f, c);
            int ix = make_rgb_texture(c); tex_ix[o.get()] = ix; return ix;
        } else fail("unsupported texture type \"" + o->type + "\"");
        S.textures.push_back(T); S.texdata.push_back(std::move(data));
        int ix = (int) S.textures.size() - 1; tex_ix[o.get()] = ix; return ix;
    }
    int texture_or_rgb(const Obj &o, const char *name, float def) {
        for (auto &c : o.children) if (c.second->tag == "texture" && c.first == name) return make_texture(c.second);
        float c[3]; get_rgb(o, name, def, c); return make_rgb_texture(c);
    }
    float lookup_ior(const Obj &o, const char *k, const char *def) {   // include/mitsuba/render/ior.h
        static const struct { const char *n; float v; } tab[] = { { "vacuum", 1.0f }, { "helium", 1.000036f }, { "hydrogen", 1.000132f }, { "air", 1.000277f },
            { "carbon dioxide", 1.00045f }, { "water", 1.3330f }, { "acetone", 1.36f }, { "ethanol", 1.361f }, { "carbon tetrachloride", 1.461f },
            { "glycerol", 1.4729f }, { "benzene", 1.501f }, { "silicone oil", 1.52045f }, { "bromine", 1.661f }, { "water ice", 1.31f },
            { "fused quartz", 1.458f }, { "pyrex", 1.470f }, { "acrylic glass", 1.49f }, { "polypropylene", 1.49f }, { "bk7", 1.5046f },
            { "sodium chloride", 1.544f }, { "amber", 1.55f }, { "pet", 1.5750f }, { "diamond", 2.419f } };
        auto it = o.props.find(k);
        std::string name = def;
        if (it != o.props.end()) { if (it->second.tag == "float" || it->second.tag == "integer") return parse_f32(attr(*it->second.node, "value")); name = attr(*it->second.node, "value"); }
        for (auto &e : tab) if (name == e.n) return e.v;
        fail("unknown material \"" + name + "\"");
    }
    int make_bsdf(const ObjP &o) {
        auto it = bsdf_ix.find(o.get()); if (it != bsdf_ix.end()) return it->second;
        lrt_bsdf_desc B{}; B.reflectance = B.nested = B.texture = -1; B.eta = 1.f; B.scale = 1.f;
        if (o->type == "diffuse") {
            B.type = LRT_BSDF_DIFFUSE; B.reflectance = texture_or_rgb(*o, "reflectance", .5f);
            if (S.textures[B.reflectance].type == LRT_TEX_BITMAP) fail("unsupported: a bitmap texture as diffuse reflectance (bitmaps are supported as bump-map heights only)");
        }
        else if (o->type == "dielectric") {
            B.type = LRT_BSDF_DIELECTRIC;
            float ii = lookup_ior(*o, "int_ior", "bk7"), ei = lookup_ior(*o, "ext_ior", "air");
            if (ii < 0 || ei < 0) fail("The interior and exterior indices of refraction must be positive!");
            if (has(*o, "specular_reflectance") || has(*o, "specular_transmittance")) fail("dielectric: specular_reflectance/transmittance are not supported");
            B.eta = ii / ei;
        } else if (o->type == "bumpmap") {
            B.type = LRT_BSDF_BUMPMAP; B.scale = get_float(*o, "scale", 1.f);
            ObjP nb = child(*o, "bsdf"), nt = child(*o, "texture");
            if (!nb) fail("Exactly one BSDF child object must be specified."); if (!nt) fail("Exactly one Texture child object must be specified.");
            B.nested = make_bsdf(nb); B.texture = make_texture(nt);
            if (S.bsdfs[B.nested].type == LRT_BSDF_BUMPMAP) fail("nested bump maps are not supported");
        } else if (o->type == "null") B.type = LRT_BSDF_NULL;
        else if (o->type == "twosided") { ObjP nb = child(*o, "bsdf"); if (!nb) fail("twosided: missing nested bsdf"); fail("twosided BSDFs are not supported"); }
        else fail("unsupported bsdf type \"" + o->type + "\"");
        S.bsdfs.push_back(B); int ix = (int) S.bsdfs.size() - 1; bsdf_ix[o.get()] = ix; return ix;
    }
    int make_medium(const ObjP &o) {
        auto it = medium_ix.find(o.get()); if (it != medium_ix.end()) return it->second;
        lrt_medium_desc M{};
        if (o->type == "homogeneous") M.type = LRT_MEDIUM_HOMOGENEOUS;
        else if (o->type == "liver") M.type = LRT_MEDIUM_LIVER;
        else if (o->type == "parenchyma") M.type = LRT_MEDIUM_PARENCHYMA;
        else if (o->type == "glissonCapsule") M.type = LRT_MEDIUM_GLISSON;
        else if (o->type == "heterogeneous") M.type = LRT_MEDIUM_HETEROGENEOUS;
        else fail("unsupported medium type \"" + o->type + "\"");
        std::vector<float> grid;
        const bool parenchyma = M.type == LRT_MEDIUM_PARENCHYMA;
        // src/media/homogeneous.cpp:112-119, src/media/liver.cpp:139-141,193-194, src/media/parenchyma.cpp:140-151
        if (M.type == LRT_MEDIUM_HETEROGENEOUS) {                                 // src/media/heterogeneous.cpp:156-164
            ObjP vol = child(*o, "volume", "sigma_t");
            if (!vol) fail("heterogeneous medium: `sigma_t` must be a gridvolume (constant volumes: use a homogeneous medium)");
            load_grid_volume(*vol, M, grid);
            for (auto &c : o->children) if (c.second->tag == "volume" && c.first == "albedo") fail("unsupported: a volume as the albedo of a heterogeneous medium");
            M.sigma_t[0] = M.sigma_t[1] = M.sigma_t[2] = 1.f;
            get_rgb(*o, "albedo", .75f, M.albedo);
        } else { get_rgb(*o, "sigma_t", 1.f, M.sigma_t); get_rgb(*o, "albedo", .75f, M.albedo); }
        M.scale = get_float(*o, "scale", 1.f);
        M.has_spectral_extinction = get_bool(*o, "has_spectral_extinction", !parenchyma);
        M.sample_emitters = get_bool(*o, "sample_emitters", !parenchyma);
        M.phase = LRT_PHASE_ISOTROPIC; M.g = 0.f;
        if (ObjP ph = child(*o, "phase")) {
            if (ph->type == "hg") { M.phase = LRT_PHASE_HG; M.g = get_float(*ph, "g", 0.8f); if (!(M.g > -1.f && M.g < 1.f)) fail("The asymmetry parameter must lie in the interval (-1, 1)!"); }
            else if (ph->type != "isotropic") fail("unsupported phase function \"" + ph->type + "\"");
        }
        // ---- bio parameters of the 5-argument sample_interaction (docs/BIO_TRANSPORT_SPEC.md section 2)
        for (int l = 0; l < 4; ++l) for (int c = 0; c < 3; ++c) M.sigma_collagen[l][c] = M.sigma_elastin[l][c] = 1.f;
        static const float limits[4] = { 0.0065f, 0.0072f, 0.0083f, 0.01f };
        for (int l = 0; l < 4; ++l) M.layer_limit[l] = limits[l];
        for (int c = 0; c < 3; ++c) M.sigma_blood[c] = M.sigma_bile[c] = M.sigma_lipid_water[c] = 1.f;
        M.sigma_hepatocity = 1.f;
        if (M.type == LRT_MEDIUM_LIVER || M.type == LRT_MEDIUM_GLISSON) {        // liver.cpp:143-186, glissonCapsule.cpp:143-186
            for (int l = 0; l < 4; ++l) {
                const std::string n = std::to_string(l + 1);
                M.layer_limit[l] = get_float(*o, ("layer" + n + "Limit").c_str(), limits[l]);
                // the members are filled R <- "_R", G <- "_B", B <- "_G"; elastin layers 3 and 4 read straight
                const bool swap_e = l < 2;
                M.sigma_collagen[l][0] = get_float(*o, ("sigma_collagen" + n + "_R").c_str(), 1.f);
                M.sigma_collagen[l][1] = get_float(*o, ("sigma_collagen" + n + "_B").c_str(), 1.f);
                M.sigma_collagen[l][2] = get_float(*o, ("sigma_collagen" + n + "_G").c_str(), 1.f);
                M.sigma_elastin[l][0] = get_float(*o, ("sigma_elastin" + n + "_R").c_str(), 1.f);
                M.sigma_elastin[l][1] = get_float(*o, ("sigma_elastin" + n + (swap_e ? "_B" : "_G")).c_str(), 1.f);
                M.sigma_elastin[l][2] = get_float(*o, ("sigma_elastin" + n + (swap_e ? "_G" : "_B")).c_str(), 1.f);
            }
        }
        if (M.type == LRT_MEDIUM_LIVER || M.type == LRT_MEDIUM_PARENCHYMA) {     // liver.cpp:188-191, parenchyma.cpp:144-147
            get_rgb(*o, "sigma_blood", 1.f, M.sigma_blood); get_rgb(*o, "sigma_bile", 1.f, M.sigma_bile);
            get_rgb(*o, "sigma_lipid_water", 1.f, M.sigma_lipid_water);
            M.sigma_hepatocity = get_float(*o, "sigma_hepatocity", 1.f);
        }
        snprintf(M.id, sizeof(M.id), "%s", o->id.empty() ? ("medium" + std::to_string(S.media.size())).c_str() : o->id.c_str());
        S.media.push_back(M); S.meddata.resize(S.media.size()); S.meddata.back() = std::move(grid);
        int ix = (int) S.media.size() - 1; medium_ix[o.get()] = ix; return ix;
    }

    // src/volumes/grid.cpp:159-330 (one channel, trilinear, clamp) + src/render/volumegrid.cpp:29-83 (the .vol file)
    void load_grid_volume(const Obj &v, lrt_medium_desc &M, std::vector<float> &grid) {
        if (v.type != "gridvolume") fail("unsupported volume type \"" + v.type + "\" (supported: gridvolume)");
        if (get_string(v, "filter_type", "trilinear") != "trilinear" || get_string(v, "wrap_mode", "clamp") != "clamp") fail("gridvolume: only trilinear / clamp is supported");
        if (has(v, "max_value")) fail("gridvolume: max_value is not supported");
        std::string path = resolve(get_string(v, "filename", nullptr));
        FILE *f = fopen(path.c_str(), "rb"); if (!f) fail("cannot open \"" + path + "\"");
        unsigned char hdr[4]; int32_t meta[5]; float dims[6];
        bool ok = fread(hdr, 1, 4, f) == 4 && fread(meta, 4, 5, f) == 5 && fread(dims, 4, 6, f) == 6;
        if (!ok || hdr[0] != 'V' || hdr[1] != 'O' || hdr[2] != 'L') { fclose(f); fail("Invalid volume file!"); }
        if (hdr[3] != 3 || meta[0] != 1) { fclose(f); fail("volume file: only version 3 / Float32 data is supported"); }
        if (meta[4] != 1) { fclose(f); fail("unsupported: grid volumes with more than one channel"); }
        if (meta[1] < 1 || meta[2] < 1 || meta[3] < 1 || (int64_t) meta[1] * meta[2] * meta[3] > (1ll << 28)) { fclose(f); fail("volume file: invalid dimensions"); }
        const size_t n = (size_t) meta[1] * meta[2] * meta[3];
        grid.resize(n);
        ok = fread(grid.data(), 4, n, f) == n; fclose(f);
        if (!ok) fail("volume file: truncated data");
        M.grid_res[0] = meta[1]; M.grid_res[1] = meta[2]; M.grid_res[2] = meta[3];
        float mx = -INFINITY; for (float x : grid) mx = std::max(mx, x);
        M.grid_max = mx;
        // m_to_local = to_world^-1 (Volume base class); use_grid_bbox: the file's bounding box maps onto the unit cube first
        Mat4 TW = get_xform(v, "to_world"), TL = inverse_affine(TW);
        if (get_bool(v, "use_grid_bbox", false)) {                                // VolumeGrid::bbox_transform: scale(1 / extents) * translate(-min)
            Mat4 B = ident();
            for (int a = 0; a < 3; ++a) { double e = (double) dims[3 + a] - (double) dims[a]; B.m[5 * a] = 1.0 / e; B.m[4 * a + 3] = -(double) dims[a] / e; }
            TL = mul(B, TL);
        }
        float tl[16]; to_float(TL, tl); memcpy(M.grid_to_local, tl, sizeof(float) * 12);
        Mat4 TWL = inverse_affine(TL); float tw[16]; to_float(TWL, tw);           // update_bbox(): world-space bounds of the unit cube
        for (int a = 0; a < 3; ++a) { M.grid_bbox_min[a] = INFINITY; M.grid_bbox_max[a] = -INFINITY; }
        for (int c = 0; c < 8; ++c) {
            F3 p = xf_point(tw, { (float) (c >> 2 & 1), (float) (c >> 1 & 1), (float) (c & 1) });
            const float q[3] = { p.x, p.y, p.z };
            for (int a = 0; a < 3; ++a) { M.grid_bbox_min[a] = std::min(M.grid_bbox_min[a], q[a]); M.grid_bbox_max[a] = std::max(M.grid_bbox_max[a], q[a]); }
        }
    }

    uint32_t add_vertex(F3 p, F3 n, float u, float v) {
        S.positions.insert(S.positions.end(), { p.x, p.y, p.z }); S.normals.insert(S.normals.end(), { n.x, n.y, n.z });
        S.texcoords.insert(S.texcoords.end(), { u, v }); return (uint32_t) (S.positions.size() / 3 - 1);
    }

    void load_obj(const Obj &o, const float *tw, const float *it, lrt_shape_desc &sd, uint32_t base) {
        // src/shapes/obj.cpp:146-400
        std::string path = resolve(get_string(o, "filename", nullptr));
        bool flip_tc = get_bool(o, "flip_tex_coords", true), face_normals = get_bool(o, "face_normals", false);
        FILE *f = fopen(path.c_str(), "rb"); if (!f) fail("Error while loading OBJ file \"" + path + "\": file not found");
        std::vector<F3> vs, ns; std::vector<std::pair<float, float>> ts;
        struct Key { uint32_t a, b, c; bool operator<(const Key &k) const { return a != k.a ? a < k.a : (b != k.b ? b < k.b : c < k.c); } };
        std::map<Key, uint32_t> vmap; std::vector<Key> keys; std::vector<uint32_t> tris;
        char buf[1100];
        while (fgets(buf, sizeof(buf), f)) {
            const char *cur = buf; while (*cur == ' ' || *cur == '\t' || *cur == '\r') ++cur;
            if (cur[0] == 'v' && (cur[1] == ' ' || cur[1] == '\t')) {
                char *e; F3 p; cur += 2; p.x = strtof(cur, &e); cur = e; p.y = strtof(cur, &e); cur = e; p.z = strtof(cur, &e);
                vs.push_back(xf_point(tw, p));
            } else if (cur[0] == 'v' && cur[1] == 'n' && (cur[2] == ' ' || cur[2] == '\t')) {
                if (!face_normals) { char *e; F3 n; cur += 3; n.x = strtof(cur, &e); cur = e; n.y = strtof(cur, &e); cur = e; n.z = strtof(cur, &e); ns.push_back(xf_normal(it, n)); }
            } else if (cur[0] == 'v' && cur[1] == 't' && (cur[2] == ' ' || cur[2] == '\t')) {
                char *e; float u, v; cur += 3; u = strtof(cur, &e); cur = e; v = strtof(cur, &e); if (flip_tc) v = 1.f - v; ts.emplace_back(u, v);
            } else if (cur[0] == 'f' && (cur[1] == ' ' || cur[1] == '\t')) {
                cur += 2; size_t vi = 0; int ti = 0; Key key{ 0, 0, 0 }; uint32_t tri[3] = { 0, 0, 0 };
                for (;;) {
                    char *nx; uint32_t val = (uint32_t) strtoul(cur, &nx, 10);
                    if (cur == nx) break;
                    if (ti == 0) key.a = val; else if (ti == 1) key.b = val; else if (ti == 2) key.c = val; else fail("could not parse OBJ face");
                    while (*nx == '/') { ti++; nx++; }
                    if (*nx == ' ' || *nx == '\t' || *nx == '\0' || *nx == '\r' || *nx == '\n') {
                        ti = 0;
                        if (key.a == 0 || key.a > vs.size()) fail("OBJ: reference to invalid vertex");
                        uint32_t id; auto fi = vmap.find(key);
                        if (fi != vmap.end()) id = fi->second; else { id = (uint32_t) keys.size(); vmap[key] = id; keys.push_back(key); }
                        if (vi < 3) tri[vi] = id; else { tri[1] = tri[2]; tri[2] = id; }
                        vi++;
                        if (vi >= 3) tris.insert(tris.end(), { tri[0], tri[1], tri[2] });
                        key = Key{ 0, 0, 0 };
                    }
                    cur = nx;
                }
            }
        }
        fclose(f);
        bool has_n = !face_normals && !ns.empty(), has_t = !ts.empty();
        for (auto &k : keys) {
            F3 n = { 0, 0, 0 }; float u = 0, v = 0;
            if (has_n && k.c) { if (k.c > ns.size()) fail("OBJ: reference to invalid normal"); n = ns[k.c - 1]; }
            if (has_t && k.b) { if (k.b > ts.size()) fail("OBJ: reference to invalid texture coordinate"); u = ts[k.b - 1].first; v = ts[k.b - 1].second; }
            add_vertex(vs[k.a - 1], n, u, v);
        }
        for (uint32_t t : tris) S.faces.push_back(base + t);
        sd.n_faces = (uint32_t) (tris.size() / 3); sd.has_normals = has_n; sd.has_texcoords = has_t;
    }

    void make_shape(const ObjP &o) {
        lrt_shape_desc sd{}; sd.bsdf = sd.emitter = sd.interior_medium = sd.exterior_medium = -1;
        Mat4 TW = get_xform(*o, "to_world"); float tw[16], it[16]; to_float(TW, tw); inv_transpose_float(TW, it);
        memcpy(sd.to_world, tw, sizeof(tw));
        sd.flip_normals = get_bool(*o, "flip_normals", false);
        sd.first_face = (uint32_t) (S.faces.size() / 3);
        uint32_t base = (uint32_t) (S.positions.size() / 3);
        if (o->type == "rectangle") {           // src/shapes/rectangle.cpp:85-160
            sd.kind = LRT_SHAPE_RECTANGLE;
            F3 n = xf_normal(it, { 0.f, 0.f, 1.f });
            for (uint32_t i = 0; i < 4; ++i) { float xf = (float) (i & 1), yf = (float) ((i & 2) >> 1); add_vertex(xf_point(tw, { fmaf(xf, 2.f, -1.f), fmaf(yf, 2.f, -1.f), 0.f }), n, xf, yf); }
            for (uint32_t v : { 1u, 2u, 0u, 1u, 3u, 2u }) S.faces.push_back(base + v);
            sd.n_faces = 2; sd.has_normals = 1; sd.has_texcoords = 1;
        } else if (o->type == "cube") {         // src/shapes/cube.cpp:103-150
            sd.kind = LRT_SHAPE_MESH;
            static const float V[24][3] = { { 1, -1, -1 }, { 1, -1, 1 }, { -1, -1, 1 }, { -1, -1, -1 }, { 1, 1, -1 }, { -1, 1, -1 }, { -1, 1, 1 }, { 1, 1, 1 },
                { 1, -1, -1 }, { 1, 1, -1 }, { 1, 1, 1 }, { 1, -1, 1 }, { 1, -1, 1 }, { 1, 1, 1 }, { -1, 1, 1 }, { -1, -1, 1 },
                { -1, -1, 1 }, { -1, 1, 1 }, { -1, 1, -1 }, { -1, -1, -1 }, { 1, 1, -1 }, { 1, -1, -1 }, { -1, -1, -1 }, { -1, 1, -1 } };
            static const float N[6][3] = { { 0, -1, 0 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }, { -1, 0, 0 }, { 0, 0, -1 } };
            static const float UV[4][2] = { { 0, 1 }, { 1, 1 }, { 1, 0 }, { 0, 0 } };
            static const uint32_t T[12][3] = { { 0, 1, 2 }, { 3, 0, 2 }, { 4, 5, 6 }, { 7, 4, 6 }, { 8, 9, 10 }, { 11, 8, 10 }, { 12, 13, 14 }, { 15, 12, 14 },
                { 16, 17, 18 }, { 19, 16, 18 }, { 20, 21, 22 }, { 23, 20, 22 } };
            for (int i = 0; i < 24; ++i) add_vertex(xf_point(tw, { V[i][0], V[i][1], V[i][2] }), xf_normal(it, { N[i / 4][0], N[i / 4][1], N[i / 4][2] }), UV[i % 4][0], UV[i % 4][1]);
            for (auto &t : T) for (uint32_t v : t) S.faces.push_back(base + v);
            sd.n_faces = 12; sd.has_normals = 1; sd.has_texcoords = 1;
        } else if (o->type == "obj") { sd.kind = LRT_SHAPE_MESH; load_obj(*o, tw, it, sd, base); }
        else fail("unsupported shape type \"" + o->type + "\"");
        uint32_t shape_ix = (uint32_t) S.shapes.size();
        for (uint32_t i = 0; i < sd.n_faces; ++i) S.face_shape.push_back(shape_ix);
        for (auto &c : o->children) {
            const ObjP &ch = c.second;
            if (ch->tag == "bsdf") sd.bsdf = make_bsdf(ch);
            else if (ch->tag == "medium") { int m = make_medium(ch); if (c.first == "interior") sd.interior_medium = m; else if (c.first == "exterior") sd.exterior_medium = m; else fail("medium child of a shape must be named \"interior\" or \"exterior\""); }
            else if (ch->tag == "emitter") {
                if (ch->type != "area") fail("only area emitters can be attached to shapes");
                if (sd.kind != LRT_SHAPE_RECTANGLE) fail("area emitters are supported on rectangle shapes only");
                if (has(*ch, "to_world")) fail("Found a 'to_world' transformation -- this is not allowed. The area light inherits this transformation from its parent shape.");
                lrt_emitter_desc E{}; E.type = LRT_EMITTER_AREA; get_rgb(*ch, "radiance", 1.f, E.radiance); E.shape = (int) shape_ix; E.scale = 1.f;
                S.emitters.push_back(E); S.emdata.emplace_back(); sd.emitter = (int) S.emitters.size() - 1;
            } else fail("unsupported child <" + ch->tag + "> of a shape");
        }
        if (sd.bsdf < 0) {                      // default BSDF: diffuse, reflectance 0.5
            lrt_bsdf_desc B{}; B.type = LRT_BSDF_DIFFUSE; float c[3] = { .5f, .5f, .5f }; B.reflectance = make_rgb_texture(c); B.nested = B.texture = -1; B.eta = 1.f; B.scale = 1.f;
            S.bsdfs.push_back(B); sd.bsdf = (int) S.bsdfs.size() - 1;
        }
        S.shapes.push_back(sd);
    }

    void make_emitter(const ObjP &o) {
        lrt_emitter_desc E{}; E.shape = -1; E.scale = 1.f; std::vector<float> data;
        Mat4 I = ident(); to_float(I, E.to_world);
        if (o->type == "constant") { E.type = LRT_EMITTER_CONSTANT; get_rgb(*o, "radiance", 1.f, E.radiance); }
        else if (o->type == "envmap") {         // src/emitters/envmap.cpp:109-236
            E.type = LRT_EMITTER_ENVMAP; E.scale = get_float(*o, "scale", 1.f);
            if (get_bool(*o, "mis_compensation", false)) fail("envmap: mis_compensation is not supported");
            to_float(get_xform(*o, "to_world"), E.to_world);
            Image im = read_image_rgb(resolve(get_string(*o, "filename", nullptr)));
            if (im.width < 2 || im.height < 3) fail("the environment map resolution must be at least 2x3 pixels");
            E.width = im.width; E.height = im.height; data.resize((size_t) im.width * im.height * 3);
            for (size_t p = 0; p < (size_t) im.width * im.height; ++p)
                for (int c = 0; c < 3; ++c) { float v = im.data[p * im.channels + (im.channels >= 3 ? c : 0)]; if (im.srgb) v = srgb_to_linear(v); data[p * 3 + c] = v; }
        } else fail("unsupported emitter type \"" + o->type + "\"");
        for (auto &e : S.emitters) if (e.type != LRT_EMITTER_AREA) fail("Only one environment emitter can be specified per scene.");
        S.emitters.push_back(E); S.emdata.push_back(std::move(data));
    }

    void make_sensor(const ObjP &o) {
        if (o->type != "perspective") fail("unsupported sensor type \"" + o->type + "\"");
        lrt_sensor_desc &C = S.desc.sensor; lrt_film_desc &F = S.desc.film;
        ObjP film = child(*o, "film"), sampler = child(*o, "sampler");
        F = lrt_film_desc{}; F.width = 768; F.height = 576; F.rfilter = LRT_RFILTER_GAUSSIAN; F.rfilter_param = .5f;
        if (film) {
            if (film->type != "hdrfilm") fail("unsupported film type \"" + film->type + "\"");
            F.width = get_int(*film, "width", 768); F.height = get_int(*film, "height", 576);
            std::string pf = get_string(*film, "pixel_format", "rgb"); for (auto &c : pf) c = (char) tolower(c);
            if (pf == "rgba") F.has_alpha = 1; else if (pf != "rgb") fail("hdrfilm: only rgb / rgba pixel formats are supported");
            if (ObjP rf = child(*film, "rfilter")) {
                if (rf->type == "box") { F.rfilter = LRT_RFILTER_BOX; F.rfilter_param = .5f; }
                else if (rf->type == "gaussian") { F.rfilter = LRT_RFILTER_GAUSSIAN; F.rfilter_param = get_float(*rf, "stddev", .5f); }
                else if (rf->type == "tent") { F.rfilter = LRT_RFILTER_TENT; F.rfilter_param = get_float(*rf, "radius", 1.f); }
                else fail("unsupported reconstruction filter \"" + rf->type + "\"");
            }
            F.crop_offset_x = get_int(*film, "crop_offset_x", 0); F.crop_offset_y = get_int(*film, "crop_offset_y", 0);
            F.crop_width = get_int(*film, "crop_width", F.width); F.crop_height = get_int(*film, "crop_height", F.height);
            if (get_bool(*film, "sample_border", false)) fail("hdrfilm: sample_border is not supported");
        } else { F.crop_width = F.width; F.crop_height = F.height; }
        if (F.crop_width <= 0 || F.crop_height <= 0 || F.crop_offset_x < 0 || F.crop_offset_y < 0 || F.crop_offset_x + F.crop_width > F.width || F.crop_offset_y + F.crop_height > F.height)
            fail("Invalid crop window specification!");
        S.desc.sample_count = 4; S.desc.sampler_seed = 0; S.desc.sampler_type = LRT_SAMPLER_INDEPENDENT;
        if (sampler) {
            if (sampler->type != "independent" && sampler->type != "ldsampler") fail("unsupported sampler type \"" + sampler->type + "\"");
            S.desc.sample_count = (uint32_t) get_int(*sampler, "sample_count", 4); S.desc.sampler_seed = (uint32_t) get_int(*sampler, "seed", 0);
            if (sampler->type == "ldsampler") {            // ldsampler.cpp:83-93: sample_count is rounded up to a square power of two
                S.desc.sampler_type = LRT_SAMPLER_LD;
                uint32_t res = 2;
                while (res * res < S.desc.sample_count) { ++res; uint32_t p2 = 1; while (p2 < res) p2 <<= 1; res = p2; }
                S.desc.sample_count = res * res;
            }
        }
        // src/render/sensor.cpp:123-132,142-196 (parse_fov)
        C.near_clip = get_float(*o, "near_clip", 1e-2f); C.far_clip = get_float(*o, "far_clip", 1e4f);
        C.principal_point_offset_x = get_float(*o, "principal_point_offset_x", 0.f); C.principal_point_offset_y = get_float(*o, "principal_point_offset_y", 0.f);
        if (C.near_clip <= 0.f) fail("The 'near_clip' parameter must be greater than zero!");
        if (C.near_clip >= C.far_clip) fail("The 'near_clip' parameter must be smaller than 'far_clip'.");
        double aspect = F.width / (double) F.height, fov, result;
        std::string axis;
        if (has(*o, "fov") && has(*o, "focal_length")) fail("Please specify either a focal length ('focal_length') or a field of view ('fov')!");
        if (has(*o, "fov")) {
            fov = strtod(attr(*o->props["fov"].node, "value").c_str(), nullptr);
            axis = get_string(*o, "fov_axis", "x"); for (auto &c : axis) c = (char) tolower(c);
            if (axis == "smaller") axis = aspect > 1 ? "y" : "x"; else if (axis == "larger") axis = aspect > 1 ? "x" : "y";
        } else {
            std::string fl = get_string(*o, "focal_length", "50mm"); if (fl.size() > 2 && fl.substr(fl.size() - 2) == "mm") fl = fl.substr(0, fl.size() - 2);
            double value = strtod(fl.c_str(), nullptr);
            fov = 2.0 * (180.0 / M_PI) * atan(sqrt(double(36 * 36 + 24 * 24)) / (2.0 * value)); axis = "diagonal";
        }
        if (axis == "x") result = fov;
        else if (axis == "y") result = (180.0 / M_PI) * (2.0 * atan(tan(0.5 * fov * M_PI / 180.0) * aspect));
        else if (axis == "diagonal") { double diag = 2.0 * tan(0.5 * fov * M_PI / 180.0), width = diag / sqrt(1.0 + 1.0 / (aspect * aspect)); result = (180.0 / M_PI) * (2.0 * atan(width * 0.5)); }
        else fail("The 'fov_axis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!");
        if (result <= 0.0 || result >= 180.0) fail("The horizontal field of view must be in the range [0, 180]!");
        C.fov_x = (float) result;
        to_float(get_xform(*o, "to_world"), C.to_world);
        C.medium = -1;
        for (auto &c : o->children) if (c.second->tag == "medium") C.medium = make_medium(c.second);
    }

    void make_integrator(const ObjP &o) {
        lrt_integrator_desc &I = S.desc.integrator;
        if (o->type == "path") I.type = LRT_INTEGRATOR_PATH;
        else if (o->type == "volpath") I.type = LRT_INTEGRATOR_VOLPATH;
        else if (o->type == "prbvolpath") I.type = LRT_INTEGRATOR_PRBVOLPATH;
        else if (o->type == "biovolpath") I.type = LRT_INTEGRATOR_BIOVOLPATH;
        else if (o->type == "biovolpath06") I.type = LRT_INTEGRATOR_BIOVOLPATH06;
        else if (o->type == "volpathmis") I.type = LRT_INTEGRATOR_VOLPATHMIS;
        else fail("unsupported integrator \"" + o->type + "\" (supported: path, volpath, volpathmis, prbvolpath, biovolpath, biovolpath06)");
        // src/render/integrator.cpp:535-552
        I.max_depth = get_int(*o, "max_depth", -1); I.rr_depth = get_int(*o, "rr_depth", 5); I.hide_emitters = get_bool(*o, "hide_emitters", false);
        if (I.max_depth < 0 && I.max_depth != -1) fail("\"max_depth\" must be set to -1 (infinite) or a value >= 0");
        if (I.rr_depth <= 0) fail("\"rr_depth\" must be set to a value greater than zero!");
        int spass = get_int(*o, "samples_per_pass", -1);                      // src/render/integrator.cpp:22-38 (SamplingIntegrator)
        S.desc.samples_per_pass = spass > 0 ? (uint32_t) spass : 0u;
        S.desc.use_spectral_mis = get_bool(*o, "use_spectral_mis", true) ? 1u : 0u;     // src/integrators/volpathmis.cpp:47
    }

    void run(const std::string &text) {
        std::unique_ptr<XmlNode> root = xml_parse(text);
        if (root->tag != "scene") fail("root element must be <scene>");
        // defaults first (document order matters only among themselves)
        for (auto &c : root->children) if (c->tag == "default") { std::string k = attr(*c, "name"); if (!vars.count(k)) vars[k] = attr(*c, "value"); }
        S.desc = lrt_scene_desc{};
        S.desc.integrator = { LRT_INTEGRATOR_PATH, -1, 5, 0 }; S.desc.use_spectral_mis = 1;
        bool have_sensor = false;
        std::vector<ObjP> objs;
        for (auto &c : root->children) {
            if (c->tag == "default") continue;
            ObjP o = parse_object(*c);
            if (!o->id.empty()) by_id[o->id] = o;
            objs.push_back(o);
        }
        for (auto &o : objs) {
            if (o->tag == "integrator") make_integrator(o);
            else if (o->tag == "sensor") { if (have_sensor) fail("only one sensor is supported"); make_sensor(o); have_sensor = true; }
            else if (o->tag == "shape") make_shape(o);
            else if (o->tag == "emitter") make_emitter(o);
            else if (o->tag == "bsdf") make_bsdf(o);
            else if (o->tag == "medium") make_medium(o);
            else if (o->tag == "texture") make_texture(o);
            else fail("unsupported top-level element <" + o->tag + ">");
        }
        if (!have_sensor) fail("the scene has no sensor");
        S.fix_pointers();
    }
};

} // namespace

void load_scene_xml(const std::string &xml_text, const std::string &base_dir,
                    const std::vector<std::pair<std::string, std::string>> &defines, SceneStorage &out) {
    Loader L(out);
    L.base_dir = base_dir;
    for (auto &d : defines) L.vars[d.first] = d.second;
    L.run(xml_text);
}

} // namespace lrt
