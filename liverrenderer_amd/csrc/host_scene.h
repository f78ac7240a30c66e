// Host-side scene object behind the C ABI (include/liverrt.h).
#pragma once
#include "../../include/liverrt.h"
#include <string>
#include <vector>
#include <memory>

namespace lrt {

struct DeviceScene;   // device-resident data + wavefront workspace (device.hip)
struct MultiContext;  // RCCL communicators of a device list (device.hip)

// Owns every array the POD description points into.
struct SceneStorage {
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> faces, face_shape;
    std::vector<lrt_shape_desc> shapes;
    std::vector<lrt_bsdf_desc> bsdfs;
    std::vector<lrt_texture_desc> textures;
    std::vector<std::vector<float>> texdata;
    std::vector<lrt_medium_desc> media;
    std::vector<lrt_emitter_desc> emitters;
    std::vector<std::vector<float>> emdata;
    std::vector<std::vector<float>> meddata;   // heterogeneous media: grid values
    lrt_scene_desc desc{};
    void fix_pointers();                 // re-point desc at the vectors above
    void copy_from(const lrt_scene_desc &d);
};

// XML -> storage (loader.cpp).  Throws std::runtime_error.
void load_scene_xml(const std::string &xml_text, const std::string &base_dir,
                    const std::vector<std::pair<std::string, std::string>> &defines, SceneStorage &out);

} // namespace lrt

struct lrt_scene {
    lrt::SceneStorage st;
    lrt::DeviceScene *dev = nullptr;     // created lazily on first device call
    bool params_dirty = true;
    int dev_ordinal = -1;                  // HIP device the device image lives on
    lrt_render_stats stats{};
    // lrt_render_multi: one device image per entry of the last device list, and that list's communicators
    std::vector<lrt::DeviceScene *> multi; std::vector<int> multi_ids; lrt::MultiContext *multi_ctx = nullptr; bool multi_params_dirty = false;
};
