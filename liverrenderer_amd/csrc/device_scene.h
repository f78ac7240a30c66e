// Interface between the plain-C++ part of the library (capi.cpp) and the HIP
// translation unit (device.hip).  All functions throw std::runtime_error.
#pragma once
#include "../../include/liverrt.h"
#include <vector>

namespace lrt {
struct DeviceScene;
DeviceScene *device_scene_create(const lrt_scene_desc &d, int device);
void device_scene_destroy(DeviceScene *d);
void device_scene_update_params(DeviceScene *D, const lrt_scene_desc &d);
void device_render(DeviceScene *D, const lrt_scene_desc &d, const lrt_render_opts *opts, float *film_raw, float *image, lrt_render_stats &stats);
void device_develop(DeviceScene *D, const float *film_raw, float *image, int on_device);
void device_render_samples(DeviceScene *D, const lrt_scene_desc &d, const lrt_render_opts *opts, uint64_t lane_begin, uint32_t n, float *out, lrt_render_stats &stats);
void device_trace(DeviceScene *D, const lrt_rays_soa *rays, const lrt_hits_soa *hits, uint32_t n, int any_hit);
// network stage of the learned subsurface model (kernels_vae.h): host arrays in, host arrays out
void device_vae_scatter(const float *blob, uint32_t n, const float *in_pos, const float *in_dir, const float *poly, const float albedo[3], float g, float ior,
                        const float sigma_t[3], float fit_scale, uint32_t seed, float *out_pos, float *out_absorption, int device);
void device_render_backward(DeviceScene *D, const lrt_scene_desc &d, const lrt_render_opts *opts, const float *grad_image, lrt_param_grads *out, lrt_render_stats &stats);
void device_math_eval(int fn, const float *x, const float *y, uint32_t n, float *out, float *out2, int device);   // test hook: dmath.h on the device
// one process, several devices: tiles over the devices, one RCCL all-reduce of the film / of the 7 gradient doubles (device.hip)
struct MultiContext;
void multi_context_destroy(MultiContext *m);
void device_render_multi(std::vector<DeviceScene *> &devs, MultiContext *&ctx, const lrt_scene_desc &d, const lrt_render_opts *opts, float *film_raw, float *image, lrt_render_stats &stats);
void device_render_backward_multi(std::vector<DeviceScene *> &devs, MultiContext *&ctx, const lrt_scene_desc &d, const lrt_render_opts *opts, const float *grad_image, lrt_param_grads *out, lrt_render_stats &stats);
} // namespace lrt
