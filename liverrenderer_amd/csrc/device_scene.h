// Interface between the plain-C++ part of the library (capi.cpp) and the HIP
// translation unit (device.hip).  All functions throw std::runtime_error.
#pragma once
#include "../../include/liverrt.h"

namespace lrt {
struct DeviceScene;
DeviceScene *device_scene_create(const lrt_scene_desc &d, int device);
void device_scene_destroy(DeviceScene *d);
void device_scene_update_params(DeviceScene *D, const lrt_scene_desc &d);
void device_render(DeviceScene *D, const lrt_scene_desc &d, const lrt_render_opts *opts, float *film_raw, float *image, lrt_render_stats &stats);
void device_develop(DeviceScene *D, const float *film_raw, float *image, int on_device);
void device_render_samples(DeviceScene *D, const lrt_scene_desc &d, const lrt_render_opts *opts, uint64_t lane_begin, uint32_t n, float *out, lrt_render_stats &stats);
void device_trace(DeviceScene *D, const lrt_rays_soa *rays, const lrt_hits_soa *hits, uint32_t n, int any_hit);
void device_render_backward(DeviceScene *D, const lrt_scene_desc &d, const lrt_render_opts *opts, const float *grad_image, lrt_param_grads *out, lrt_render_stats &stats);
} // namespace lrt
