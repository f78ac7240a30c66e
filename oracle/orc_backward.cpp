/* orc_backward.cpp -- PRB adjoint of the CPU oracle (filled in below). TEST INFRASTRUCTURE ONLY. */
#include "orc_scene.h"
extern "C" int orc_render_backward(orc_scene *, const lrt_render_opts *, int, const float *, lrt_param_grads *) { return 1; }
