// Developer microbenchmark: does a second, independent path per lane (instruction-level parallelism inside a wave) raise the throughput
// of the proven-free medium trip's arithmetic at 4 waves per SIMD?  One "trip" = the draw / log / exp / sincos / division skeleton of
// volpath_iteration's in-medium branch.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I../../liverrenderer_amd/csrc t_ilp2.hip -o t_ilp2
#include "dmath.h"
#include <cstdio>
#include <vector>
using namespace lrt;

struct Path { PCG32 rng; float ox, oy, oz, dx, dy, dz, tp; };
DEV void trip(Path &p) {
    float u = p.rng.next();                                  // termination draw
    float q = fmin_(p.tp, .95f); if (u > q + 2.f) p.tp = 0.f;
    float t = 0.f + (-m_log(1.f - p.rng.next()) / 1.3f);     // free flight
    float e = m_exp(-t * 1.3f); p.tp = p.tp * (e / (e * 1.3f)) * 1.3f * 0.9f;
    p.ox = fma_(p.dx, t, p.ox); p.oy = fma_(p.dy, t, p.oy); p.oz = fma_(p.dz, t, p.oz);
    (void) p.rng.next();
    float sx, sy; p.rng.next2(sx, sy); float u3 = p.rng.next();
    if (!(1.f - u3 >= 1e-20f) || sx > 2.f || sy > 2.f) p.tp = 0.f;
    (void) p.rng.next();
    float a, b; p.rng.next2(a, b);
    float z = fma_(-2.f, b, 1.f), r = safe_sqrt(fma_(-z, z, 1.f)); float s, c; m_sincos(2.f * kPi * a, &s, &c);
    p.dx = r * c; p.dy = r * s; p.dz = z;
}
template <int STREAMS>
__global__ void __launch_bounds__(1024, 4) k(float *out, int iters) {
    Path p[STREAMS];
    for (int k = 0; k < STREAMS; ++k) { p[k].rng.ld_count = 0; p[k].rng.seed(threadIdx.x + 1024 * blockIdx.x, 7 + k); p[k].ox = p[k].oy = p[k].oz = 0.f; p[k].dx = 1.f; p[k].dy = p[k].dz = 0.f; p[k].tp = 1.f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < STREAMS; ++k) trip(p[k]);
    }
    float acc = 0.f; for (int k = 0; k < STREAMS; ++k) acc += p[k].ox + p[k].oy + p[k].oz + p[k].tp;
    out[threadIdx.x + 1024 * blockIdx.x] = acc;
}
int main() {
    float *out; hipMalloc((void **) &out, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) for (int S = 1; S <= 3; ++S) {
        hipEventRecord(e0);
        if (S == 1) k<1><<<256, 1024>>>(out, iters); else if (S == 2) k<2><<<256, 1024>>>(out, iters); else k<3><<<256, 1024>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("streams %d: %.3f ms  -> %.3f ns per trip per lane-stream, %.2f G trips/s\n", S, ms, ms * 1e6 / (iters * S) , 256.0 * 1024 * iters * S / ms * 1e-6);
    }
    return 0;
}
