// Micro-benchmark: same-address device-scope returning atomics from persistent waves (ticket counters).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(1024) k_ticket(unsigned *cnt, unsigned n_tiles, unsigned n_extra, unsigned long long *sink) {
    unsigned lane = threadIdx.x & 63u; unsigned long long acc = 0;
    for (;;) {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(&cnt[0], 1u);
        t = __shfl(t, 0);
        if (t >= n_tiles) break;
        for (unsigned k = 0; k < n_extra; ++k) { unsigned s = 0; if (lane == 0) s = atomicAdd(&cnt[16 * (k + 1)], 37u); acc += __shfl(s, 0); }
        acc += t;
    }
    if (acc == 0xdeadbeefull) sink[0] = acc;
}
int main(int argc, char **argv) {
    unsigned *cnt; unsigned long long *sink; hipMalloc(&cnt, 4096); hipMalloc(&sink, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (unsigned extra = 0; extra <= 3; extra += 3) for (unsigned n : { 8192u, 32768u, 131072u, 524288u }) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(cnt, 0, 4096);
            hipEventRecord(a); k_ticket<<<256, 1024>>>(cnt, n, extra, sink); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("tickets %7u  extra atomics/ticket %u : %.3f ms  (%.1f M tickets/s)\n", n, extra, best, n / best * 1e-3);
    }
    return 0;
}
