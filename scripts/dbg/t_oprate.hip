// Developer microbenchmark: issue cost of single VALU instructions on gfx950 at 4 waves per SIMD (one 1024-thread workgroup per CU), eight
// independent chains per lane, no memory access.  Prints SIMD cycles per wave-instruction (clock from hipDeviceAttributeClockRate).
// Build: hipcc --offload-arch=gfx950 -O3 t_oprate.hip -o t_oprate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, DECL, BODY, SINK)                                                              \
__global__ void __launch_bounds__(1024, 4) NAME(float *out, int iters, float fb, float fc, unsigned ub) { \
    DECL                                                                                            \
    for (int i = 0; i < iters; ++i) {                                                               \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) { REP8(BODY) }                                \
    }                                                                                               \
    out[threadIdx.x + 1024 * blockIdx.x] = SINK;                                                    \
}
#define FDECL float a[8]; for (int k = 0; k < 8; ++k) a[k] = fb * (float) (threadIdx.x + k);
#define FSINK (a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
#define UDECL unsigned uc = ub * 3u + threadIdx.x; unsigned long long msk = __ballot(threadIdx.x & 1); asm volatile("s_mov_b64 vcc, %0" : : "s"(msk) : "vcc"); unsigned a[8]; for (int k = 0; k < 8; ++k) a[k] = ub * (threadIdx.x + k);
#define USINK (float) (a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7])
#define LDECL unsigned long long a[8]; for (int k = 0; k < 8; ++k) a[k] = (unsigned long long) ub * (threadIdx.x + k);
#define LSINK (float) (unsigned) ((a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7]) >> 7)
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PDECL f32x2 a[8]; for (int k = 0; k < 8; ++k) a[k] = f32x2{ fb * (float) (threadIdx.x + k), fc }; f32x2 pb = { fb, fc }, pc = { fc, fb };
#define PSINK (a[0].x + a[1].y + a[2].x + a[3].y + a[4].x + a[5].y + a[6].x + a[7].y)

#define B_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(fb), "v"(fc));
#define B_ADD(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(fb));
#define B_MIN3(k) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(fb), "v"(fc));
#define B_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
#define B_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
#define B_CVT(k) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[k]));
#define B_DIVSCALE(k) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[k]) : "v"(fb) : "vcc");
#define B_DIVFIXUP(k) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(fb), "v"(fc));
#define B_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(ub));
#define B_MULHI(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(ub));
#define B_MUL24(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[k]) : "v"(ub));
#define B_MAD24(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[k]) : "v"(ub));
#define B_XOR(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(ub));
#define B_ALIGNBIT(k) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(a[k]) : "v"(ub));
#define B_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[k]) : "v"(ub));
#define B_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(ub) : );
#define B_CNDMASK_S(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(ub), "s"(msk));
#define B_CNDMASK_D(k) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[k]) : "v"(ub), "v"(uc));
#define B_CMPCND(k) a[k] = (a[k] < ub) ? a[k] + 1u : uc;
#define B_BFI(k) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(ub), "v"(uc));
#define B_CMP(k) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[k]), "v"(ub) : "vcc");
#define B_MAD64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[k]) : "v"(ub), "v"((unsigned) threadIdx.x) : "vcc");
#define B_LSHL64(k) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[k]));
#define B_ADD64(k) a[k] += 0x123456789abcdefull + ub;
#define B_MUL64(k) a[k] = a[k] * 6364136223846793005ull + 1442695040888963407ull;
#define B_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(pb), "v"(pc));
#define B_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(pb));
#define B_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(ub));

KERNEL(k_fma, FDECL, B_FMA, FSINK) KERNEL(k_add, FDECL, B_ADD, FSINK) KERNEL(k_min3, FDECL, B_MIN3, FSINK) KERNEL(k_rcp, FDECL, B_RCP, FSINK)
KERNEL(k_sqrt, FDECL, B_SQRT, FSINK) KERNEL(k_cvt, FDECL, B_CVT, FSINK) KERNEL(k_divscale, FDECL, B_DIVSCALE, FSINK) KERNEL(k_divfixup, FDECL, B_DIVFIXUP, FSINK)
KERNEL(k_mullo, UDECL, B_MULLO, USINK) KERNEL(k_mulhi, UDECL, B_MULHI, USINK) KERNEL(k_mul24, UDECL, B_MUL24, USINK) KERNEL(k_mad24, UDECL, B_MAD24, USINK)
KERNEL(k_xor, UDECL, B_XOR, USINK) KERNEL(k_alignbit, UDECL, B_ALIGNBIT, USINK) KERNEL(k_lshladd, UDECL, B_LSHLADD, USINK) KERNEL(k_cndmask, UDECL, B_CNDMASK, USINK)
KERNEL(k_cmp, UDECL, B_CMP, USINK) KERNEL(k_cndmask_s, UDECL, B_CNDMASK_S, USINK) KERNEL(k_cndmask_d, UDECL, B_CNDMASK_D, USINK) KERNEL(k_cmpcnd, UDECL, B_CMPCND, USINK) KERNEL(k_bfi, UDECL, B_BFI, USINK) KERNEL(k_mov, UDECL, B_MOV, USINK)
KERNEL(k_mad64, LDECL, B_MAD64, LSINK) KERNEL(k_lshl64, LDECL, B_LSHL64, LSINK) KERNEL(k_add64, LDECL, B_ADD64, LSINK) KERNEL(k_pcgmul64, LDECL, B_MUL64, LSINK)
KERNEL(k_pkfma, PDECL, B_PKFMA, PSINK) KERNEL(k_pkmul, PDECL, B_PKMUL, PSINK)


#define GROUP(NAME, HEAD, SEL)                                                                      \
__global__ void __launch_bounds__(1024, 4) NAME(float *out, int iters, float fb, float fc, unsigned ub) { \
    UDECL                                                                                           \
    for (int i = 0; i < iters; ++i) {                                                               \
        _Pragma("unroll") for (int u = 0; u < 8; ++u)                                               \
            asm volatile(HEAD "\n s_nop 1\n" SEL(0) SEL(1) SEL(2) SEL(3) SEL(4) SEL(5) SEL(6) SEL(7)  \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(ub), "v"(uc), "s"(msk) : "vcc", "s10", "s11"); \
    }                                                                                               \
    out[threadIdx.x + 1024 * blockIdx.x] = USINK;                                                   \
}
#define SEL_VCC(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define SEL_S(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[10:11]\n"
GROUP(g_vcmp_vcc, "v_cmp_lt_u32 vcc, %8, %9", SEL_VCC)
GROUP(g_vcmp_s, "v_cmp_lt_u32_e64 s[10:11], %8, %9", SEL_S)
GROUP(g_smov_vcc, "s_mov_b64 vcc, %10", SEL_VCC)
GROUP(g_smov_s, "s_mov_b64 s[10:11], %10", SEL_S)
GROUP(g_none_vcc, "s_nop 0", SEL_VCC)

#define SEL_E64VCC(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, vcc\n"
#define FMA2(k) "v_xor_b32 %" #k ", %" #k ", %9\n v_xor_b32 %" #k ", %" #k ", %8\n"
#define SEL_VCC_MIX(k) SEL_VCC(k) FMA2(k)
#define SEL_S_MIX(k) SEL_S(k) FMA2(k)
#define ONLY_XOR(k) "v_xor_b32 %" #k ", %" #k ", %9\n"
#define LATE_VCC(k) ONLY_XOR(k)
GROUP(g_e64vcc, "v_cmp_lt_u32 vcc, %8, %9", SEL_E64VCC)
GROUP(g_mix_vcc, "v_cmp_lt_u32 vcc, %8, %9", SEL_VCC_MIX)
GROUP(g_mix_s, "v_cmp_lt_u32_e64 s[10:11], %8, %9", SEL_S_MIX)
GROUP(g_late_vcc, "v_cmp_lt_u32 vcc, %8, %9", ONLY_XOR)
__global__ void __launch_bounds__(1024, 4) g_late1(float *out, int iters, float fb, float fc, unsigned ub) {
    UDECL
    for (int i = 0; i < iters; ++i) {
        _Pragma("unroll") for (int u = 0; u < 8; ++u)
            asm volatile("v_cmp_lt_u32 vcc, %8, %9\n" ONLY_XOR(0) ONLY_XOR(1) ONLY_XOR(2) ONLY_XOR(3) ONLY_XOR(4) ONLY_XOR(5) ONLY_XOR(6) "v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(ub), "v"(uc), "s"(msk) : "vcc", "s10", "s11");
    }
    out[threadIdx.x + 1024 * blockIdx.x] = USINK;
}
__global__ void __launch_bounds__(1024, 4) g_late1s(float *out, int iters, float fb, float fc, unsigned ub) {
    UDECL
    for (int i = 0; i < iters; ++i) {
        _Pragma("unroll") for (int u = 0; u < 8; ++u)
            asm volatile("v_cmp_lt_u32_e64 s[10:11], %8, %9\n" ONLY_XOR(0) ONLY_XOR(1) ONLY_XOR(2) ONLY_XOR(3) ONLY_XOR(4) ONLY_XOR(5) ONLY_XOR(6) "v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(ub), "v"(uc), "s"(msk) : "vcc", "s10", "s11");
    }
    out[threadIdx.x + 1024 * blockIdx.x] = USINK;
}

int main() {
    float *out; if (hipMalloc((void **) &out, 256 * 1024 * 4) != hipSuccess) return 1;
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    printf("clock %d kHz; 256 workgroups x 1024 threads (4 waves per SIMD), %d x 64 instructions per lane\n", khz, iters);
#define RUN(NAME, NOTE) for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); NAME<<<256, 1024>>>(out, iters, 1.0001f, 0.5f, 0x9e3779b9u); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
        if (rep) printf("%-12s %8.3f ms  %6.2f SIMD cycles per wave-instruction  %s\n", #NAME, ms, ms * 1e-3 * khz * 1e3 / (4.0 * iters * 64), NOTE); }
    RUN(k_fma, "") RUN(k_add, "") RUN(k_min3, "") RUN(k_rcp, "") RUN(k_sqrt, "") RUN(k_cvt, "") RUN(k_divscale, "") RUN(k_divfixup, "")
    RUN(k_mullo, "") RUN(k_mulhi, "") RUN(k_mul24, "") RUN(k_mad24, "") RUN(k_xor, "") RUN(k_alignbit, "") RUN(k_lshladd, "") RUN(k_cndmask, "") RUN(k_cmp, "") RUN(k_cndmask_s, "(mask in an SGPR pair)") RUN(k_cndmask_d, "(destination not a source)") RUN(k_cmpcnd, "(compiler: compare + add + select per step)") RUN(k_bfi, "") RUN(k_mov, "")
    RUN(k_mad64, "") RUN(k_lshl64, "") RUN(k_add64, "(compiler: add_co + addc = 2 instructions per step)") RUN(k_pcgmul64, "(compiler's 64-bit multiply-add by the PCG32 constants: one step, several instructions)")
    printf("groups: one mask write + 8 selects (cycles per instruction of the 9; s_nop not counted)\n");
#undef RUN
#define RUN(NAME, NOTE) for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); NAME<<<256, 1024>>>(out, iters, 1.0001f, 0.5f, 0x9e3779b9u); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
        if (rep) printf("%-12s %8.3f ms  %6.2f SIMD cycles per instruction  %s\n", #NAME, ms, ms * 1e-3 * khz * 1e3 / (4.0 * iters * 72), NOTE); }
    RUN(g_vcmp_vcc, "v_cmp -> vcc, 8 x v_cndmask vcc") RUN(g_vcmp_s, "v_cmp -> s[10:11], 8 x v_cndmask_e64") RUN(g_smov_vcc, "s_mov -> vcc") RUN(g_smov_s, "s_mov -> s[10:11]") RUN(g_none_vcc, "vcc untouched")
    RUN(g_e64vcc, "v_cmp -> vcc, 8 x v_cndmask_b32_e64 ... vcc (VOP3 encoding, vcc explicit)") RUN(g_late_vcc, "v_cmp -> vcc, 8 x v_xor (no select): baseline for the next")
    printf("the next: cycles for the whole group\n");
#undef RUN
#define RUN(NAME, N, NOTE) for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); NAME<<<256, 1024>>>(out, iters, 1.0001f, 0.5f, 0x9e3779b9u); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
        if (rep) printf("%-12s %8.3f ms  %6.2f SIMD cycles per group of %d  %s\n", #NAME, ms, ms * 1e-3 * khz * 1e3 / (4.0 * iters * 8), N, NOTE); }
    RUN(g_mix_vcc, 25, "v_cmp -> vcc, 8 x (v_cndmask vcc, 2 x v_xor)") RUN(g_mix_s, 25, "v_cmp -> s, 8 x (v_cndmask_e64 s, 2 x v_xor)") RUN(g_late1, 9, "v_cmp -> vcc, 7 x v_xor, v_cndmask vcc") RUN(g_late1s, 9, "v_cmp -> s, 7 x v_xor, v_cndmask_e64 s")
    return 0;
    return 0;
}
