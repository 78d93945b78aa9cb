// bigtile_probe.hip -- what a ONE-wave-per-SIMD GEMM body with a 128 x 128 register tile per wave reaches on gfx950 (round 5: DESIGN.md section 11 prices the tile table by
// 4096 x (1 / (r BN) + 1 / BM) bytes of LDS-DMA per MFMA clock; the 8-wave tiles stop at 256 x 128 / 512 x 64 because a wave's accumulators must fit 256 VGPRs at two waves per
// SIMD.  At ONE wave per SIMD a wave has 512 registers -- 256 of them AGPRs, which is where 64 accumulator tiles of 16 x 16 fit).
//   workgroup = 4 waves = 256 rows x 256 columns of C, k-phases of 64: A (rows) and B (columns) arrive by LDS-DMA, 64 KiB per phase, two slots;
//   wave (wm, wn) owns rows 128 wm .. +128, columns 128 wn .. +128: per 32-deep step 8 + 8 fragments, 64 MFMAs (16 x 16 x 32 f16).
// Synthetic operands (K-contiguous rows, as conv.hip's 1x1 case), C is reduced to one float per lane (the probe times the k-loop, there is no epilogue).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/bin/bigtile_probe tools/probes/bigtile_probe.hip
// Usage: bigtile_probe [K=1024] [tiles_per_wg=4] [mode=-1 (all)] [shareB=0]     modes 0-2: hipcc's schedule (DMA + reads + MFMA; static LDS image; MFMA only), 3-5: the written-out schedule
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef _Float16 f16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ int swz16(int row) { return ((row >> 3) & 1) * 3; }
__device__ __forceinline__ void glds16(const f16 *src, unsigned char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

// LDS image of a phase: 64 pieces of 1 KiB; piece (g, kk) = 16 rows (row group g of 16; A: g = 0..15, B: g = 16..31) x 32 halves of k-half kk, rows of 64 bytes with the
// chunk XOR swizzle of conv_dev.h (conflict-free for ds_read_b128 by lanes (r = lane & 15, q = lane >> 4))
constexpr int SLOT = 64 * 1024;

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void bigtile(const f16 *__restrict__ A, const f16 *__restrict__ B, float *__restrict__ out, int K, int tiles,
                                                                                        unsigned long long *__restrict__ clk) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int ld_row = lane >> 2, ld_chunk = (lane & 3) ^ swz16(ld_row);
    const int r = lane & 15, q = lane >> 4, rd_off = r * 64 + ((q ^ swz16(r)) << 4);
    const int nph = K / 64;
    floatx4 total = {0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0, t1 = 0;
    if (lane == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int tile = 0; tile < tiles; ++tile) {
        const int m0 = ((blockIdx.x * tiles + tile) * 256) % 65536;      // rows of A / B this tile reads (65 536-row operands, wrapped)
        // piece p of a phase: wave w issues pieces 16 w .. 16 w + 15; p < 32: A row group p >> 1, k-half p & 1; else B
        auto issue = [&](int ph, int slot) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int p = wave * 16 + i, g = (p & 31) >> 1, kk = p & 1;
                const f16 *base = p < 32 ? A : B;
                glds16(base + (size_t)(m0 + g * 16 + ld_row) * K + ph * 64 + kk * 32 + ld_chunk * 8, lds + slot * SLOT + p * 1024);
            }
        };
        floatx4 acc[8][8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[t][u] = floatx4{0.f, 0.f, 0.f, 0.f};
        half8 fa[2][8], fb[2][8];
        auto read_frags = [&](int slot, int kk, half8 (&a)[8], half8 (&b)[8]) {
            const unsigned char *sa = lds + slot * SLOT + ((wm * 8) * 2 + kk) * 1024 + rd_off;
            const unsigned char *sb = lds + slot * SLOT + ((16 + wn * 8) * 2 + kk) * 1024 + rd_off;
#pragma unroll
            for (int t = 0; t < 8; ++t) a[t] = *(const half8 *)(sa + t * 2048);
#pragma unroll
            for (int u = 0; u < 8; ++u) b[u] = *(const half8 *)(sb + u * 2048);
        };
        auto mfmas = [&](const half8 (&a)[8], const half8 (&b)[8]) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u], a[t], acc[t][u], 0, 0, 0);
        };
        if (MODE == 0) { issue(0, 0); if (nph > 1) issue(1, 1); }
        if (MODE == 0) { if (nph > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
        if (MODE != 2) read_frags(0, 0, fa[0], fb[0]);
        else {
#pragma unroll
            for (int t = 0; t < 8; ++t) { fa[0][t] = half8{(f16)lane, 1, 2, 3, 4, 5, 6, 7}; fb[0][t] = fa[0][t]; fa[1][t] = fa[0][t]; fb[1][t] = fa[0][t]; }
        }
        for (int ph = 0; ph < nph; ++ph) {
            const int slot = ph & 1;
            // k-half 1's fragments travel under k-half 0's MFMAs
            if (MODE != 2) read_frags(slot, 1, fa[1], fb[1]);
            mfmas(fa[0], fb[0]);
            if (MODE != 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every read of this slot has returned ...
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // ... and this wave's pieces of phase ph + 1 have landed
            if (MODE != 2) __syncthreads();                                         // slot `slot` is free, slot ^ 1 is complete
            if (MODE == 0 && ph + 2 < nph) issue(ph + 2, slot);
            if (MODE != 2 && ph + 1 < nph) read_frags(slot ^ 1, 0, fa[0], fb[0]);  // the next phase's first fragments under k-half 1's MFMAs
            mfmas(fa[1], fb[1]);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int u = 0; u < 8; ++u) total += acc[t][u];
        __syncthreads();
    }
    if (lane == 0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); clk[blockIdx.x * 4 + wave] = t1 - t0; }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = total[0] + total[1] + total[2] + total[3];
}

// ---- the same body with the schedule written out: accumulators pinned in AGPRs (inline-asm MFMAs: hipcc's allocation of 256 accumulator registers shuffles them through
// v_accvgpr_mov -- 842 of the builtin form's instructions), one ds_read_b128 of the NEXT 32-deep step after every 4th MFMA, one LDS-DMA piece after every 8th ----
#define BT_MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a))
#define BT_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")

template <int MODE>      // 3: MFMA only; 4: + fragment reads; 5: + LDS-DMA
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void bigtile_asm(const f16 *__restrict__ A, const f16 *__restrict__ B, float *__restrict__ out, int K, int tiles,
                                                                                            unsigned long long *__restrict__ clk, int shareB) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int ld_row = lane >> 2, ld_chunk = (lane & 3) ^ swz16(ld_row);
    const int r = lane & 15, q = lane >> 4, rd_off = r * 64 + ((q ^ swz16(r)) << 4);
    const int nph = K / 64;
    floatx4 total = {0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0, t1 = 0;
    if (lane == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    // LDS byte addresses of this lane's fragments: [slot][kk] for A (rows) and B (columns); t / u advance by the 2 048-byte immediate
    unsigned ra[2][2], rb[2][2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            ra[sl][kk] = (unsigned)(sl * SLOT + ((wm * 8) * 2 + kk) * 1024 + rd_off);
            rb[sl][kk] = (unsigned)(sl * SLOT + ((16 + wn * 8) * 2 + kk) * 1024 + rd_off);
        }
    for (int tile = 0; tile < tiles; ++tile) {
        const int m0 = ((blockIdx.x * tiles + tile) * 256) % 65536;
        // this lane's source offsets (halves) of the wave's 16 pieces of a phase, without the phase's k offset
        unsigned src[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int p = wave * 16 + i, g = (p & 31) >> 1, kk = p & 1;
            const int row0 = (p >= 32 && shareB) ? 0 : m0;      // shareB: every workgroup multiplies by the SAME 256 rows of B -- a conv's weights, L2-resident
            src[i] = (unsigned)((row0 + g * 16 + ld_row) * K + kk * 32 + ld_chunk * 8);
        }
        const f16 *base = wave < 2 ? A : B;                // (pieces 0-31 are A's: waves 0, 1)
        auto issue_piece = [&](int i, int ph, int slot) __attribute__((always_inline)) {
            glds16(base + (size_t)src[i] + ph * 64, lds + slot * SLOT + (wave * 16 + i) * 1024);
        };
        floatx4 acc[8][8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc[t][u] = floatx4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[t][u])); }
        half8 fa[2][8], fb[2][8];
        if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i) issue_piece(i, 0, 0);
            if (nph > 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) issue_piece(i, 1, 1);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (MODE >= 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { BT_DSREAD(fa[0][j], ra[0][0], j * 2048); BT_DSREAD(fb[0][j], rb[0][0], j * 2048); }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) { fa[0][t] = half8{(f16)lane, 1, 2, 3, 4, 5, 6, 7}; fb[0][t] = fa[0][t]; fa[1][t] = fa[0][t]; fb[1][t] = fa[0][t]; }
        }
        // one phase with everything about it known at compile time (slot, whether a phase follows, whether phase ph + 2 exists): no branch, no select between the MFMAs
        auto phase = [&](auto slot_c, auto more_c, auto dma_c, int ph) __attribute__((always_inline)) {
            constexpr int slot = decltype(slot_c)::value;
            constexpr bool more = decltype(more_c)::value, dma = decltype(dma_c)::value;
            // k-half 0: MFMAs on set 0; set 1 <- (slot, k-half 1)
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = t * 8 + u;
                    if (MODE >= 4 && (idx & 3) == 0) {
                        const int j = idx >> 2;
                        if (j < 8) BT_DSREAD(fa[1][j], ra[slot][1], j * 2048); else BT_DSREAD(fb[1][j - 8], rb[slot][1], (j - 8) * 2048);
                    }
                    BT_MFMA(acc[t][u], fa[0][t], fb[0][u]);
                }
            if (MODE >= 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MODE == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE >= 4) __syncthreads();                 // slot is free (every wave's reads of it have returned), slot ^ 1 is complete
            // k-half 1: MFMAs on set 1; set 0 <- (slot ^ 1, k-half 0); the wave's 16 pieces of phase ph + 2 go into the freed slot, one per 4 MFMAs
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = t * 8 + u;
                    if (MODE >= 4 && more && (idx & 3) == 0) {
                        const int j = idx >> 2;
                        if (j < 8) BT_DSREAD(fa[0][j], ra[slot ^ 1][0], j * 2048); else BT_DSREAD(fb[0][j - 8], rb[slot ^ 1][0], (j - 8) * 2048);
                    }
                    if (MODE == 5 && dma && (idx & 3) == 2) issue_piece(idx >> 2, ph + 2, slot);
                    BT_MFMA(acc[t][u], fa[1][t], fb[1][u]);
                }
            if (MODE >= 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        using T = std::true_type; using F = std::false_type; using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        int ph = 0;
        for (; ph + 4 <= nph; ph += 2) { phase(S0{}, T{}, T{}, ph); phase(S1{}, T{}, T{}, ph + 1); }      // (nph even, >= 2)
        phase(S0{}, T{}, F{}, ph); phase(S1{}, F{}, F{}, ph + 1);
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int u = 0; u < 8; ++u) total += acc[t][u];
        __syncthreads();
    }
    if (lane == 0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); clk[blockIdx.x * 4 + wave] = t1 - t0; }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = total[0] + total[1] + total[2] + total[3];
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 1024, tiles = argc > 2 ? atoi(argv[2]) : 4;
    const int only = argc > 3 ? atoi(argv[3]) : -1;
    const int shareB = argc > 4 ? atoi(argv[4]) : 0;      // 1: the written-out kernels read ONE set of 256 B rows in every workgroup
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int G = prop.multiProcessorCount;
    const size_t rows = 65536 + 256;
    std::vector<f16> h(rows * K);
    unsigned s = 99;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (f16)(((s >> 9) & 0xFF) / 1024.f - 0.12f); }
    f16 *A, *B; float *out; unsigned long long *clk;
    CK(hipMalloc(&A, h.size() * 2)); CK(hipMalloc(&B, h.size() * 2)); CK(hipMalloc(&out, (size_t)G * 256 * 4)); CK(hipMalloc(&clk, (size_t)G * 4 * 8));
    CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data() + 128, (h.size() - 128) * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[6] = {"LDS-DMA + fragment reads + MFMA", "fragment reads + MFMA (static LDS image)", "MFMA only",
                            "written-out schedule: MFMA only", "written-out schedule: fragment reads + MFMA", "written-out schedule: LDS-DMA + reads + MFMA"};
    CK(hipFuncSetAttribute((const void *)bigtile_asm<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT));
    CK(hipFuncSetAttribute((const void *)bigtile_asm<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT));
    CK(hipFuncSetAttribute((const void *)bigtile_asm<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT));
    for (int mode = 0; mode < 6; ++mode) {
        if (only >= 0 && mode != only) continue;
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(bigtile<0>, dim3(G), dim3(256), 2 * SLOT, 0, A, B, out, K, tiles, clk);
            else if (mode == 1) hipLaunchKernelGGL(bigtile<1>, dim3(G), dim3(256), 2 * SLOT, 0, A, B, out, K, tiles, clk);
            else if (mode == 2) hipLaunchKernelGGL(bigtile<2>, dim3(G), dim3(256), 2 * SLOT, 0, A, B, out, K, tiles, clk);
            else if (mode == 3) hipLaunchKernelGGL(bigtile_asm<3>, dim3(G), dim3(256), 2 * SLOT, 0, A, B, out, K, tiles, clk, shareB);
            else if (mode == 4) hipLaunchKernelGGL(bigtile_asm<4>, dim3(G), dim3(256), 2 * SLOT, 0, A, B, out, K, tiles, clk, shareB);
            else hipLaunchKernelGGL(bigtile_asm<5>, dim3(G), dim3(256), 2 * SLOT, 0, A, B, out, K, tiles, clk, shareB);
        };
        CK(hipFuncSetAttribute((const void *)bigtile<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT));
        CK(hipFuncSetAttribute((const void *)bigtile<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT));
        CK(hipFuncSetAttribute((const void *)bigtile<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT));
        launch(); CK(hipDeviceSynchronize());
        const int iters = 10;
        CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = 2.0 * 256 * 256 * (double)K * tiles * G, us = ms * 1e3 / iters;
        std::vector<unsigned long long> hc((size_t)G * 4);
        CK(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0; for (auto c : hc) sum += (double)c;
        const double clk_per_phase = sum / hc.size() / ((double)tiles * (K / 64));      // s_memtime runs at 100 MHz: report as a ratio only
        printf("%-44s%s K %5d x %d tiles per workgroup, %d workgroups: %8.1f us  %7.1f TFLOP/s (%.0f %% of 2.5 PFLOP/s)   [s_memtime ticks per phase %.1f]\n", names[mode], (mode >= 3 && shareB) ? " [B shared]" : "", K, tiles, G, us,
               flop / us * 1e-6, flop / us * 1e-6 / 2500 * 100, clk_per_phase);
    }
    return 0;
}
