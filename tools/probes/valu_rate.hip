// valu_rate.hip -- issue cost (cycles per wave instruction on one SIMD) of the instructions a conv epilogue's SiLU is made of, f32 and
// f16 forms, measured with s_memtime around 16 x 256 independent instructions, one wave per SIMD and four.  What it is for: whether an
// f16 form of the activation (v_exp_f16 / v_rcp_f16 on the already-converted output) would be cheaper than the f32 one.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/probes/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

#define PROBE(NAME, ASM)                                                                                   \
    __global__ void NAME(unsigned long long *out, float seed) {                                           \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        float b0, b1, b2, b3, b4, b5, b6, b7;                                                             \
        b0 = b1 = b2 = b3 = b4 = b5 = b6 = b7 = seed;                                                     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
        for (int i = 0; i < 256; ++i) {                                                                   \
            asm volatile(REP16(ASM) : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) \
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));       \
        }                                                                                                  \
        asm volatile("s_nop 0" ::: "memory");                                                             \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                        \
        if (b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 == 12345.678f) out[1] = 1;                              \
    }

// 8 independent destinations per group of 8 instructions; REP16 x 8 = 128 instructions per loop trip
PROBE(k_exp32, "v_exp_f32 %0, %8\n v_exp_f32 %1, %9\n v_exp_f32 %2, %10\n v_exp_f32 %3, %11\n v_exp_f32 %4, %12\n v_exp_f32 %5, %13\n v_exp_f32 %6, %14\n v_exp_f32 %7, %15\n")
PROBE(k_rcp32, "v_rcp_f32 %0, %8\n v_rcp_f32 %1, %9\n v_rcp_f32 %2, %10\n v_rcp_f32 %3, %11\n v_rcp_f32 %4, %12\n v_rcp_f32 %5, %13\n v_rcp_f32 %6, %14\n v_rcp_f32 %7, %15\n")
PROBE(k_exp16, "v_exp_f16 %0, %8\n v_exp_f16 %1, %9\n v_exp_f16 %2, %10\n v_exp_f16 %3, %11\n v_exp_f16 %4, %12\n v_exp_f16 %5, %13\n v_exp_f16 %6, %14\n v_exp_f16 %7, %15\n")
PROBE(k_rcp16, "v_rcp_f16 %0, %8\n v_rcp_f16 %1, %9\n v_rcp_f16 %2, %10\n v_rcp_f16 %3, %11\n v_rcp_f16 %4, %12\n v_rcp_f16 %5, %13\n v_rcp_f16 %6, %14\n v_rcp_f16 %7, %15\n")
PROBE(k_exp16_sdwa, "v_exp_f16_sdwa %0, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_exp_f16_sdwa %1, %9 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_exp_f16_sdwa %2, %10 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_exp_f16_sdwa %3, %11 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_exp_f16_sdwa %4, %12 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_exp_f16_sdwa %5, %13 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_exp_f16_sdwa %6, %14 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_exp_f16_sdwa %7, %15 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n")
PROBE(k_mul32, "v_mul_f32 %0, %8, %8\n v_mul_f32 %1, %9, %9\n v_mul_f32 %2, %10, %10\n v_mul_f32 %3, %11, %11\n v_mul_f32 %4, %12, %12\n v_mul_f32 %5, %13, %13\n v_mul_f32 %6, %14, %14\n v_mul_f32 %7, %15, %15\n")
PROBE(k_pkmul16, "v_pk_mul_f16 %0, %8, %8\n v_pk_mul_f16 %1, %9, %9\n v_pk_mul_f16 %2, %10, %10\n v_pk_mul_f16 %3, %11, %11\n v_pk_mul_f16 %4, %12, %12\n v_pk_mul_f16 %5, %13, %13\n v_pk_mul_f16 %6, %14, %14\n v_pk_mul_f16 %7, %15, %15\n")
PROBE(k_cvtpk, "v_cvt_pkrtz_f16_f32 %0, %8, %9\n v_cvt_pkrtz_f16_f32 %1, %9, %10\n v_cvt_pkrtz_f16_f32 %2, %10, %11\n v_cvt_pkrtz_f16_f32 %3, %11, %12\n v_cvt_pkrtz_f16_f32 %4, %12, %13\n v_cvt_pkrtz_f16_f32 %5, %13, %14\n v_cvt_pkrtz_f16_f32 %6, %14, %15\n v_cvt_pkrtz_f16_f32 %7, %15, %8\n")

template <typename K>
static void run(const char *name, K k, unsigned long long *d) {
    for (int waves : {1, 4}) {
        unsigned long long h[2] = {0, 0};
        hipMemset(d, 0, 16);
        k<<<dim3(1), dim3(256 * waves)>>>(d, 0.5f);        // 256 threads = one wave on each of the CU's four SIMDs
        k<<<dim3(1), dim3(256 * waves)>>>(d, 0.5f);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        // s_memtime counts shader-clock cycles (kernels.h, STAMP)
        printf("%-14s waves/SIMD %d  cycles %llu  per wave instruction %.2f cycles\n", name, waves, h[0], (double)h[0] / (256.0 * 128.0 * waves));
    }
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 16);
    run("v_mul_f32", k_mul32, d);
    run("v_exp_f32", k_exp32, d);
    run("v_rcp_f32", k_rcp32, d);
    run("v_exp_f16", k_exp16, d);
    run("v_rcp_f16", k_rcp16, d);
    run("v_exp_f16_sdwa", k_exp16_sdwa, d);
    run("v_pk_mul_f16", k_pkmul16, d);
    run("v_cvt_pkrtz", k_cvtpk, d);
    return 0;
}
