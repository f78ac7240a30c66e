// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the path-record streams of k_render:
// three kernels that do nothing but load_state / store_state on a KNOWN number of records through the DPathStreams
// layout (five float4 streams + one uint2 stream = 88 B / record, one 64-record tile per wave visit, as k_render's tiles).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/dbg/t_stream_calib scripts/dbg/t_stream_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_fetch -- scripts/dbg/t_stream_calib
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/calib_write -- scripts/dbg/t_stream_calib
// The program prints the algorithmic bytes of each launch; scripts/calib_summary.py divides.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../liverrenderer_amd/csrc/kernels.h"
using namespace lrt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) calib_load(DPathStreams q, uint32_t n, float *sink) {
    float acc = 0.f;
    for (uint32_t t = blockIdx.x * 16u + (threadIdx.x >> 6); t * 64u < n; t += gridDim.x * 16u) {
        const uint32_t i = t * 64u + (threadIdx.x & 63u);
        if (i < n) { PathState s; load_state(q, i, s); acc += s.o.x + s.d.y + s.tp.z + s.res.x + s.lp.y + s.maxt + s.eta + s.last_pdf + u2f(s.flags) + u2f(s.lane) + (float) (uint32_t) s.rng_state + (float) (uint32_t) (s.rng_state >> 32); }
    }
    if (acc == 12345.678f) *sink = acc;
}
__global__ void __launch_bounds__(1024) calib_store(DPathStreams q, uint32_t n) {
    for (uint32_t t = blockIdx.x * 16u + (threadIdx.x >> 6); t * 64u < n; t += gridDim.x * 16u) {
        const uint32_t i = t * 64u + (threadIdx.x & 63u);
        if (i < n) { PathState s; s.o = s.d = s.tp = s.res = s.lp = V3((float) i); s.maxt = s.eta = s.last_pdf = 1.f; s.flags = i; s.lane = i; s.rng_state = i; store_state(q, i, s); }
    }
}
__global__ void __launch_bounds__(1024) calib_copy(DPathStreams a, DPathStreams b, uint32_t n) {
    for (uint32_t t = blockIdx.x * 16u + (threadIdx.x >> 6); t * 64u < n; t += gridDim.x * 16u) {
        const uint32_t i = t * 64u + (threadIdx.x & 63u);
        if (i < n) { PathState s; load_state(a, i, s); s.maxt += 1.f; store_state(b, i, s); }
    }
}

static DPathStreams alloc(uint32_t n) {
    DPathStreams q{};
    CK(hipMalloc((void **) &q.o_maxt, (size_t) n * 16)); CK(hipMalloc((void **) &q.d_eta, (size_t) n * 16)); CK(hipMalloc((void **) &q.tp_pdf, (size_t) n * 16));
    CK(hipMalloc((void **) &q.res_flags, (size_t) n * 16)); CK(hipMalloc((void **) &q.lp_lane, (size_t) n * 16)); CK(hipMalloc((void **) &q.rng, (size_t) n * 8));
    q.tdepth = nullptr;
    return q;
}

int main() {
    const uint32_t n = 64u << 20;                        // 67.1 M records = 5.9 GB per queue: far beyond L2 + Infinity Cache
    DPathStreams a = alloc(n), b = alloc(n);
    float *sink; CK(hipMalloc((void **) &sink, 4));
    calib_store<<<256, 1024>>>(a, n); calib_store<<<256, 1024>>>(b, n); CK(hipDeviceSynchronize());     // warm-up / initialise
    for (int rep = 0; rep < 3; ++rep) {
        calib_load<<<256, 1024>>>(a, n, sink);
        calib_store<<<256, 1024>>>(b, n);
        calib_copy<<<256, 1024>>>(a, b, n);
    }
    CK(hipDeviceSynchronize());
    printf("records %u  bytes_per_record 88  calib_load reads %.0f B  calib_store writes %.0f B  calib_copy reads %.0f B writes %.0f B\n",
           n, 88.0 * n, 88.0 * n, 88.0 * n, 88.0 * n);
    return 0;
}
