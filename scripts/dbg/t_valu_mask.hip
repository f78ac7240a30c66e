// Microbenchmark: VALU issue cost of a wave64 instruction as a function of the EXEC mask pattern on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o t_valu_mask t_valu_mask.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode 0: all lanes; 1: lanes 0-15; 2: lanes 0-31; 3: one lane per 16-lane group; 4: lanes 0-47; 5: even lanes; 6: lane 0 only
__global__ void __launch_bounds__(1024) k_mask(float *out, int iters, int mode, int dependent) {
    const uint32_t lane = threadIdx.x & 63u;
    bool on = true;
    if (mode == 1) on = lane < 16; else if (mode == 2) on = lane < 32; else if (mode == 3) on = (lane & 15u) == 0;
    else if (mode == 4) on = lane < 48; else if (mode == 5) on = (lane & 1u) == 0; else if (mode == 6) on = lane == 0;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 1.0000001f, c = 1e-7f;
    if (on) {
        if (dependent) {
            for (int i = 0; i < iters; ++i) {
                #pragma unroll
                for (int k = 0; k < 8; ++k) a0 = __builtin_fmaf(a0, m, c);
            }
        } else {
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    const int blocks = 256, threads_max = 1024, iters = 200000;
    float *out; CHECK(hipMalloc(&out, sizeof(float) * blocks * threads_max));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[] = { "all 64 lanes", "lanes 0-15", "lanes 0-31", "one lane per 16", "lanes 0-47", "even lanes", "lane 0 only" };
    for (int threads : { 1024, 256 }) for (int dep = 0; dep < 2; ++dep) for (int mode = 0; mode < 7; ++mode) {
        k_mask<<<blocks, threads>>>(out, 1000, mode, dep); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); k_mask<<<blocks, threads>>>(out, iters, mode, dep); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double waves_per_simd = threads / 64.0 / 4.0, instr = 8.0 * iters * waves_per_simd;          // VALU fma instructions per SIMD
        printf("%4d thr/WG (%.0f waves/SIMD) %s  %-16s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", threads, waves_per_simd,
               dep ? "dependent  " : "independent", names[mode], ms, ms * 1e-3 * 2.4e9 / instr);
        fflush(stdout);
    }
    return 0;
}
