// bottleneck.hip -- a whole YOLOv8 Bottleneck (Conv3x3+SiLU -> Conv3x3+SiLU [+ x]) in ONE launch,
// the intermediate activation never leaving the CU ("C2f bottleneck pairs kept in LDS").
//
// Replaces two conv launches of the C2f modules ultralytics builds for the reference's
// `YOLO.predict` (/root/reference/src/detection/detector.py:100-111; SURVEY.md App. A:
// Bottleneck(c, c, shortcut) = Conv(c,c,3) -> Conv(c,c,3), + x iff shortcut).
//
// One 512-thread workgroup (8 wave64s, 2 per SIMD) owns a TH x TW tile of output pixels of one
// image:
//   0. the (TH+4) x (TW+4) input patch (2-pixel halo) is DMA'd into LDS once -- all 9 taps of
//      the first conv read it from there (9x less global->LDS traffic than an implicit GEMM
//      that refetches per tap); out-of-tensor pixels are fetched from a zero page;
//   1. conv1 as an implicit GEMM over the (TH+2) x (TW+2) halo-1 region, weights streamed
//      tap by tap through a 2-stage LDS ring; epilogue = bias + SiLU -> fp16 -> LDS (zero
//      where the position lies outside the image: that IS conv2's zero padding);
//   2. conv2 over the TH x TW tile from that LDS image; epilogue = bias + SiLU (+ residual)
//      -> NHWC global store.
// Channel counts 32 / 64 / 128 (the P2 / P3 / P4 bottlenecks of YOLOv8 n..m); LDS images are
// [64-channel plane][pixel][128 B] (or [pixel][64 B] for c = 32) with the same XOR swizzle as
// conv.hip, so every fragment read is a conflict-free ds_read_b128.
#include <algorithm>

#include "kernels.h"
#include "tile_math.h"

namespace rtmodt {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_b(float x) { return silu(x); }

__device__ __forceinline__ void dma16(const f16 *src, unsigned char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

struct BneckArgs {
    const f16 *in, *w1, *w2, *res, *zeros;
    const float *b1, *b2;
    f16 *out;
    int B, H, W;
    int in_Hp, in_Wp, in_cs;                 // input tensor has a 1-pixel border
    int out_Hp, out_Wp, out_cs, out_pad;
    int res_Hp, res_Wp, res_cs, res_pad;
    int kp;                                  // weight row stride (= 9 * CH)
    int tiles_x, tiles_y;
    // optional C2f.cv2 tail: a 1x1 conv over [t_in (K1 channels, the earlier chunks of the C2f concat) | this bottleneck's output];
    // only the tail's output is stored
    const f16 *t_in, *t_wt; const float *t_bias; f16 *t_out;
    int t_k1, t_cout, t_kp, t_act, t_in_Hp, t_in_Wp, t_in_cs, t_out_Hp, t_out_Wp, t_out_cs, t_out_pad;
};

constexpr int BN_THREADS = 512, BN_WAVES = 8;

// (STAMP: kernels.h)

template <int CH, int TH, int TW>
struct BneckGeom {
    static constexpr int CW = CH < 64 ? CH : 64;          // channels per LDS plane
    static constexpr int CB = CW * 2;                     // bytes per pixel row in a plane
    static constexpr int NCH = CH / CW;                   // planes
    static constexpr int RPP = CB == 128 ? 8 : 16;        // rows (pixels / couts) per 1-KiB DMA piece
    static constexpr int SUB = CW / 32;                   // MFMA k-substeps per (tap, plane)
    static constexpr int NT = CH / 16;                    // cout tiles
    static constexpr int PW = TW + 4, PH = TH + 4;        // input patch
    static constexpr int P_PIX = PW * PH, P_ROWS = (P_PIX + RPP - 1) / RPP * RPP;
    static constexpr int IW = TW + 2, IH = TH + 2;        // intermediate (halo 1)
    static constexpr int M1 = IW * IH, M1T = (M1 + 15) / 16, T_ROWS = M1T * 16;
    // (conv1's m-tiles wrap inside the halo-1 region, IW = TW + 2 wide in a PW-wide patch image: one jump of 2 rows per wrapping tile, the
    // only fragment reads left with a 2-way conflict -- 1.71 LDS cycles per group against 2.06 before; enumerating all PW columns instead
    // makes every read conflict-free at the price of 11 % more conv1 rows (MFMA + SiLU work, i.e. energy) and was measured not to change
    // the launch time either way: not kept)
    static constexpr int M2 = TW * TH, M2T = M2 / 16;
    static constexpr int TM1 = (M1T + BN_WAVES - 1) / BN_WAVES, TM2 = (M2T + BN_WAVES - 1) / BN_WAVES;
    static constexpr int W_PIECES = CH / RPP;             // DMA pieces per (tap, plane) of weights
    static constexpr int W_STEP = CH * CB;                // bytes of one weight unit: (tap, plane)
    static constexpr bool ALLW = CH == 32;                // a conv's whole weight matrix (18 KiB) sits in ONE LDS buffer: no ring, no barrier in the k-loop
    static constexpr int UPS = CH == 64 ? 3 : 1;          // units per ring stage: 24 KiB (one kernel row of c = 64), 16 KiB (c = 128: one (tap, plane))
    static constexpr int W_STAGE = UPS * W_STEP;
    static constexpr int PATCH_BYTES = NCH * P_ROWS * CB, T_BYTES = NCH * T_ROWS * CB;
    static constexpr int LDS_BYTES = PATCH_BYTES + T_BYTES + (ALLW ? 9 * NCH * W_STEP : 2 * W_STAGE);
    static_assert(M2 % 16 == 0, "tile must hold whole 16-pixel MFMA tiles");
    // tail variant (C2f.cv2 fused): y tile + 2 CH channels of the concat for the tile's pixels + the 1x1's weights, or its output tile
    static constexpr int LDS_TAIL = 152 * 1024;
};

// LDS swizzle (tile_math.h: swz_slot / swz_src, checked exhaustively on the host by tests/test_tile_math_cpu.py): conflict-free
// ds_read_b128 fragments from ANY start row.  Round 2's XOR by {0, 0, 3, 3}[R >> 2] / (R >> 1) & 7 was conflict-free only from
// rows that are multiples of 16 (what conv.hip's tiles read), and took ~2 LDS cycles per 16-lane group instead of 1 on the
// activation fragments of every tap here (38 % conflict cycles in the counters).
template <int CB>
__device__ __forceinline__ int plane_off(int R, int c16) { return swz_plane_off<CB>(R, c16); }

// One conv of the pair as an implicit GEMM from an LDS image.  src: planes [NCH][rows][CB];
// SW = row width of that image; my_pb[i] = LDS pixel index (tap 0,0) of this lane's row in the
// wave's i-th m-tile.  Weights stream through wring (2 stages); a stage holds UPS "units" (a unit = one (tap, plane):
// CH couts x CW channels), so there is one wait + barrier per UPS units -- with one unit per stage (the first version) a
// c = 32 conv met a barrier every 6 MFMAs per wave and a c = 64 one every 24; three taps per stage (c = 64) is a
// barrier per 72, all nine (c = 32) one per conv.
template <typename G, int TM, int SW, int SRC_ROWS>
__device__ __forceinline__ void gemm_from_lds(const unsigned char *src, unsigned char *wring, const f16 *w, int kp, const int (&my_pb)[TM],
                                              floatx4 (&acc)[TM][G::NT], int lane, int wave) {
    constexpr int CB = G::CB, NCH = G::NCH, RPP = G::RPP, SUB = G::SUB, NT = G::NT, UPS = G::UPS;
    constexpr int UNITS = 9 * NCH, STEPS = UNITS / UPS;
    static_assert(UNITS % UPS == 0, "units per stage must divide the 9 * planes units of a conv");
    const int r = lane & 15, q = lane >> 4;
    const int ld_row = CB == 128 ? lane >> 3 : lane >> 2, ld_slot = CB == 128 ? lane & 7 : lane & 3;
    auto issue = [&](int step) {
        unsigned char *dst = wring + (step & 1) * G::W_STAGE;
        for (int idx = wave; idx < UPS * G::W_PIECES; idx += BN_WAVES) {
            const int un = idx / G::W_PIECES, pc = idx - un * G::W_PIECES;
            const int unit = step * UPS + un, tap = unit / NCH, plane = unit - tap * NCH;
            int row = pc * RPP + ld_row;
            int c16 = swz_src<CB>(row, ld_slot);
            dma16(w + ((long)row * kp + tap * (NCH * G::CW) + plane * G::CW + c16 * 8), dst + un * G::W_STEP + pc * 1024);
        }
    };
    issue(0);
    for (int step = 0; step < STEPS; ++step) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // weights of `step` landed; everyone left step-1
        asm volatile("" ::: "memory");
        if (step + 1 < STEPS) issue(step + 1);
#pragma unroll
        for (int un = 0; un < UPS; ++un) {
            const int unit = step * UPS + un, tap = unit / NCH, plane = unit - tap * NCH;
            const int kh = tap / 3, kw = tap - kh * 3;
            const int toff = kh * SW + kw;
            const unsigned char *splane = src + plane * (SRC_ROWS * CB);
            const unsigned char *wst = wring + (step & 1) * G::W_STAGE + un * G::W_STEP;
#pragma unroll
            for (int kk = 0; kk < SUB; ++kk) {
                half8 fa[TM], fb[NT];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *(const half8 *)(splane + plane_off<CB>(my_pb[i] + toff, kk * 4 + q));
#pragma unroll
                for (int u = 0; u < NT; ++u) fb[u] = *(const half8 *)(wst + plane_off<CB>(u * 16 + r, kk * 4 + q));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int u = 0; u < NT; ++u) acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[u], fa[i], acc[i][u], 0, 0, 0);
            }
        }
    }
}

// c = 32: a conv's whole weight matrix is 18 KiB -- ONE LDS buffer holds it, loaded by DMA in one go (w1 together with the
// input patch, w2 underneath conv1's epilogue), and the k-loop over the nine taps meets no wait and no barrier.  (Measured
// with the phase stamps of tools/probes/kernel_probe.hip: with one (tap, plane) unit per ring stage a c = 32 conv spent
// 8000 clk in a k-loop of 860 clk of MFMA per wave -- a barrier every 6 MFMAs; weights in REGISTERS instead cost each of the
// 8 waves its own copy of the matrix through the global load path, 8x the bytes, and gave the gain back.)
template <typename G>
__device__ __forceinline__ void allw_issue(unsigned char *wbuf, const f16 *w, int kp, int lane, int wave) {
    constexpr int CB = G::CB, NCH = G::NCH, RPP = G::RPP;
    const int ld_row = CB == 128 ? lane >> 3 : lane >> 2, ld_slot = CB == 128 ? lane & 7 : lane & 3;
    for (int idx = wave; idx < 9 * NCH * G::W_PIECES; idx += BN_WAVES) {
        const int unit = idx / G::W_PIECES, pc = idx - unit * G::W_PIECES;
        const int tap = unit / NCH, plane = unit - tap * NCH;
        int row = pc * RPP + ld_row;
        int c16 = swz_src<CB>(row, ld_slot);
        dma16(w + ((long)row * kp + tap * (NCH * G::CW) + plane * G::CW + c16 * 8), wbuf + unit * G::W_STEP + pc * 1024);
    }
}
template <typename G, int TM, int SW, int SRC_ROWS>
__device__ __forceinline__ void gemm_allw(const unsigned char *src, const unsigned char *wbuf, const int (&my_pb)[TM], floatx4 (&acc)[TM][G::NT], int lane) {
    constexpr int CB = G::CB, NCH = G::NCH, SUB = G::SUB, NT = G::NT;
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int unit = 0; unit < 9 * NCH; ++unit) {
        const int tap = unit / NCH, plane = unit - tap * NCH;
        const int kh = tap / 3, kw = tap - kh * 3;
        const int toff = kh * SW + kw;
        const unsigned char *splane = src + plane * (SRC_ROWS * CB);
        const unsigned char *wst = wbuf + unit * G::W_STEP;
#pragma unroll
        for (int kk = 0; kk < SUB; ++kk) {
            half8 fa[TM], fb[NT];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const half8 *)(splane + plane_off<CB>(my_pb[i] + toff, kk * 4 + q));
#pragma unroll
            for (int u = 0; u < NT; ++u) fb[u] = *(const half8 *)(wst + plane_off<CB>(u * 16 + r, kk * 4 + q));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[u], fa[i], acc[i][u], 0, 0, 0);
        }
    }
}

template <int CH, int TH, int TW, int N2T = 0>   // N2T > 0: C2f.cv2 fused as a tail with 16 * N2T output channels, K1 = 2 * CH
__global__ __launch_bounds__(BN_THREADS) void bottleneck_fused(BneckArgs p) {
    using G = BneckGeom<CH, TH, TW>;
    constexpr int CB = G::CB, NCH = G::NCH, RPP = G::RPP, NT = G::NT;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *patch = lds;
    unsigned char *tbuf = lds + G::PATCH_BYTES;
    unsigned char *wring = tbuf + G::T_BYTES;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int tx = blockIdx.x % p.tiles_x, ty = (blockIdx.x / p.tiles_x) % p.tiles_y, b = blockIdx.x / (p.tiles_x * p.tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;

    STAMP(0);
    // ---- 0. input patch -> LDS (2-pixel halo; outside the bordered tensor: zero page) ----
    {
        const int ld_row = CB == 128 ? lane >> 3 : lane >> 2, ld_slot = CB == 128 ? lane & 7 : lane & 3;
        constexpr int PIECES = NCH * (G::P_ROWS / RPP);
        for (int pc = wave; pc < PIECES; pc += BN_WAVES) {
            const int plane = pc / (G::P_ROWS / RPP), pp = pc - plane * (G::P_ROWS / RPP);
            const int pix = pp * RPP + ld_row;
            const int py = pix / G::PW, px = pix - py * G::PW;
            const int gy = y0 - 2 + py, gx = x0 - 2 + px;
            const int c16 = swz_src<CB>(pix, ld_slot);
            const bool ok = pix < G::P_PIX && gy >= -1 && gy <= p.H && gx >= -1 && gx <= p.W;
            const f16 *srcp = ok ? p.in + (((long)(b * p.in_Hp + gy + 1) * p.in_Wp + gx + 1) * p.in_cs + plane * G::CW + c16 * 8) : p.zeros;
            dma16(srcp, patch + plane * (G::P_ROWS * CB) + pp * 1024);
        }
    }

    // ---- 1. conv1 over the halo-1 region ----
    {
        int pb[G::TM1];
        int mrow[G::TM1];
#pragma unroll
        for (int i = 0; i < G::TM1; ++i) {
            int t = wave + BN_WAVES * i;
            int m = t * 16 + r;
            mrow[i] = t < G::M1T ? m : -1;
            m = m < G::M1 ? m : G::M1 - 1;                      // padding rows of the last tile re-read a valid pixel
            int iy = m / G::IW, ix = m - iy * G::IW;
            pb[i] = iy * G::PW + ix;
        }
        floatx4 acc[G::TM1][NT];
#pragma unroll
        for (int i = 0; i < G::TM1; ++i)
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[i][u] = *(const floatx4 *)(p.b1 + u * 16 + q * 4);   // bias as the initial accumulator
        STAMP(1);
        if constexpr (G::ALLW) {
            allw_issue<G>(wring, p.w1, p.kp, lane, wave);       // in flight together with the patch
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                    // the input patch and w1 have landed
            STAMP(2);
            gemm_allw<G, G::TM1, G::PW, G::P_ROWS>(patch, wring, pb, acc, lane);
            __syncthreads();                                    // every wave is done with w1
            allw_issue<G>(wring, p.w2, p.kp, lane, wave);       // w2 arrives underneath the epilogue below
        } else {
            gemm_from_lds<G, G::TM1, G::PW, G::P_ROWS>(patch, wring, p.w1, p.kp, pb, acc, lane, wave);
        }
        STAMP(3);
        // epilogue 1: SiLU -> fp16 -> tbuf; positions outside the image are conv2's zero padding
#pragma unroll
        for (int i = 0; i < G::TM1; ++i) {
            const int m = mrow[i];
            if (m < 0 || m >= G::M1) continue;
            const int iy = m / G::IW, ix = m - iy * G::IW;
            const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
            const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int n = u * 16 + q * 4;
                half4 h = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                if (inside) { floatx4 sv = acc[i][u]; silu4(sv); h = half4{(f16)sv[0], (f16)sv[1], (f16)sv[2], (f16)sv[3]}; }
                const int plane = n / G::CW, cn = n - plane * G::CW;
                *(half4 *)(tbuf + plane * (G::T_ROWS * CB) + plane_off<CB>(m, cn >> 3) + (cn & 7) * 2) = h;
            }
        }
    }
    if constexpr (G::ALLW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // w2
    __syncthreads();                                            // tbuf complete; weight ring free again (ALLW: w2 landed)
    STAMP(4);

    // ---- 2. conv2 over the tile ----
    {
        int pb[G::TM2];
        int mrow[G::TM2];
#pragma unroll
        for (int i = 0; i < G::TM2; ++i) {
            int t = wave + BN_WAVES * i;
            int m = t * 16 + r;
            mrow[i] = t < G::M2T ? m : -1;
            m = m < G::M2 ? m : G::M2 - 1;
            int oy = m / TW, ox = m - oy * TW;
            pb[i] = oy * G::IW + ox;
        }
        floatx4 acc[G::TM2][NT];
#pragma unroll
        for (int i = 0; i < G::TM2; ++i)
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[i][u] = *(const floatx4 *)(p.b2 + u * 16 + q * 4);
        if constexpr (G::ALLW) gemm_allw<G, G::TM2, G::IW, G::T_ROWS>(tbuf, wring, pb, acc, lane);
        else gemm_from_lds<G, G::TM2, G::IW, G::T_ROWS>(tbuf, wring, p.w2, p.kp, pb, acc, lane, wave);
        STAMP(5);
        // epilogue 2: SiLU (+ residual) in fp32, one rounding, staged as a [pixel][channel] fp16 tile in the (dead) patch
        // region so that every lane then stores 16 bytes -- whole cache lines of the NHWC output per wave instruction
        constexpr int ROWB = CH * 2 + 16;
        static_assert(G::M2 * ROWB <= G::PATCH_BYTES, "output tile must fit the patch region");
#pragma unroll
        for (int i = 0; i < G::TM2; ++i) {
            const int m = mrow[i];
            if (m < 0) continue;
            const int oy = m / TW, ox = m - oy * TW;
            const int gy = y0 + oy, gx = x0 + ox;
            const bool live = gy < p.H && gx < p.W;
            const long rpix = (p.res && live) ? ((long)(b * p.res_Hp + gy + p.res_pad) * p.res_Wp + gx + p.res_pad) * p.res_cs : 0;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int n = u * 16 + q * 4;
                floatx4 v = acc[i][u];
                silu4(v);
                if (p.res && live) {
                    half4 rv = *(const half4 *)(p.res + rpix + n);
                    v[0] += (float)rv[0]; v[1] += (float)rv[1]; v[2] += (float)rv[2]; v[3] += (float)rv[3];
                }
                *(half4 *)(patch + m * ROWB + n * 2) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            }
        }
        if constexpr (N2T == 0) {
            __syncthreads();
            constexpr int CPR = CH / 8;
            for (int c = threadIdx.x; c < G::M2 * CPR; c += BN_THREADS) {
                const int m = c / CPR, k8 = c - m * CPR;
                const int oy = m / TW, ox = m - oy * TW;
                const int gy = y0 + oy, gx = x0 + ox;
                if (gy >= p.H || gx >= p.W) continue;
                const long opix = ((long)(b * p.out_Hp + gy + p.out_pad) * p.out_Wp + gx + p.out_pad) * p.out_cs;
                *(half8 *)(p.out + opix + k8 * 8) = *(const half8 *)(patch + m * ROWB + k8 * 16);
            }
        } else {
            // ---- 3. C2f.cv2 as a tail: out = act(W2 . [t_in (K1 = 2 CH channels from HBM) | y (CH channels, the tile above)] + b2) ----
            // LDS from here on: [y tile M2 x ROWB][A: K1/32 x M2T pieces][W2: K2/32 x N2T pieces]; conv1/conv2 buffers are dead
            constexpr int K1C = 2 * CH / 32, KYC = CH / 32, K2C = K1C + KYC, M2T = G::M2T;
            constexpr int A_OFF = (G::M2 * ROWB + 1023) / 1024 * 1024, W_OFF = A_OFF + K1C * M2T * 1024;
            constexpr int ROWO = N2T * 32 + 16;
            static_assert(W_OFF + K2C * N2T * 1024 <= G::LDS_TAIL && G::M2 * ROWO <= G::LDS_TAIL, "tail buffers");
            static_assert(M2T % BN_WAVES == 0, "pixel tiles must split over the waves");
            __syncthreads();                                   // y tile complete; tbuf / weight ring free
            STAMP(6);
            {
                const int ld_row = lane >> 2, ld_chunk = (lane & 3) ^ (((ld_row >> 3) & 1) * 3);
                for (int pi = wave; pi < K1C * M2T; pi += BN_WAVES) {       // earlier C2f chunks of this tile's pixels
                    const int kc = pi / M2T, mt = pi - kc * M2T;
                    const int m = mt * 16 + ld_row;
                    const int gy = min(y0 + m / TW, p.H - 1), gx = min(x0 + m % TW, p.W - 1);   // outside the image: any valid pixel (never stored)
                    dma16(p.t_in + (((long)(b * p.t_in_Hp + gy + 1) * p.t_in_Wp + gx + 1) * p.t_in_cs + kc * 32 + ld_chunk * 8), lds + A_OFF + pi * 1024);
                }
                for (int pi = wave; pi < K2C * N2T; pi += BN_WAVES) {       // the tail's weights, K order = concat order
                    const int kc = pi / N2T, u = pi - kc * N2T;
                    dma16(p.t_wt + ((long)(u * 16 + ld_row) * p.t_kp + kc * 32 + ld_chunk * 8), lds + W_OFF + pi * 1024);
                }
            }
            constexpr int TMT = M2T / BN_WAVES;
            floatx4 acc2[TMT][N2T], b2v[N2T];
#pragma unroll
            for (int u = 0; u < N2T; ++u) {
                b2v[u] = *(const floatx4 *)(p.t_bias + u * 16 + q * 4);
#pragma unroll
                for (int i = 0; i < TMT; ++i) acc2[i][u] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            STAMP(7);
            const int rd_off = r * 64 + ((q ^ (((r >> 3) & 1) * 3)) << 4);
#pragma unroll
            for (int kc = 0; kc < K2C; ++kc) {
                half8 fa[TMT], fb[N2T];
#pragma unroll
                for (int i = 0; i < TMT; ++i) {
                    const int mt = wave + BN_WAVES * i;
                    fa[i] = kc < K1C ? *(const half8 *)(lds + A_OFF + (kc * M2T + mt) * 1024 + rd_off)
                                     : *(const half8 *)(lds + (mt * 16 + r) * ROWB + ((kc - K1C) * 4 + q) * 16);
                }
#pragma unroll
                for (int u = 0; u < N2T; ++u) fb[u] = *(const half8 *)(lds + W_OFF + (kc * N2T + u) * 1024 + rd_off);
#pragma unroll
                for (int i = 0; i < TMT; ++i)
#pragma unroll
                    for (int u = 0; u < N2T; ++u) acc2[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[u], fa[i], acc2[i][u], 0, 0, 0);
            }
            __syncthreads();                                   // every wave is done with the y tile and the operands
            STAMP(8);
#pragma unroll
            for (int i = 0; i < TMT; ++i) {
                const int m = (wave + BN_WAVES * i) * 16 + r;
#pragma unroll
                for (int u = 0; u < N2T; ++u) {
                    floatx4 v = acc2[i][u] + b2v[u];
                    if (p.t_act) silu4(v);
                    *(half4 *)(lds + m * ROWO + (u * 16 + q * 4) * 2) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                }
            }
            __syncthreads();
            constexpr int CPR = N2T * 2;
            for (int c = threadIdx.x; c < G::M2 * CPR; c += BN_THREADS) {
                const int m = c / CPR, k8 = c - m * CPR;
                if (k8 * 8 >= p.t_cout) continue;
                const int gy = y0 + m / TW, gx = x0 + m % TW;
                if (gy >= p.H || gx >= p.W) continue;
                const long opix = ((long)(b * p.t_out_Hp + gy + p.t_out_pad) * p.t_out_Wp + gx + p.t_out_pad) * p.t_out_cs;
                *(half8 *)(p.t_out + opix + k8 * 8) = *(const half8 *)(lds + m * ROWO + k8 * 16);
            }
        }
    }
    STAMP(9);
}

template <int CH, int TH, int TW, int N2T = 0>
int launch_one(const BneckArgs &a0, int H, int W, int B, hipStream_t s) {
    using G = BneckGeom<CH, TH, TW>;
    constexpr int ROWB = CH * 2 + 16;
    constexpr int A_OFF = (G::M2 * ROWB + 1023) / 1024 * 1024, TAIL_END = A_OFF + (2 * CH / 32) * G::M2T * 1024 + (3 * CH / 32) * N2T * 1024;
    constexpr int TAIL_OUT = G::M2 * (N2T * 32 + 16);
    constexpr int LDS = N2T == 0 ? G::LDS_BYTES : std::max(G::LDS_BYTES, std::max(TAIL_END, TAIL_OUT));
    static_assert(LDS <= G::LDS_TAIL, "LDS budget");
    BneckArgs a = a0;
    a.tiles_x = cdiv(W, TW); a.tiles_y = cdiv(H, TH);
    static DynLdsSeen seen;
    RT_TRY(raise_dynamic_lds((const void *)bottleneck_fused<CH, TH, TW, N2T>, (size_t)LDS, seen));
    hipLaunchKernelGGL((bottleneck_fused<CH, TH, TW, N2T>), dim3(a.tiles_x * a.tiles_y * B), dim3(BN_THREADS), LDS, s, a);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

}  // namespace

bool bottleneck_supported(int c) { return c == 32 || c == 64 || c == 128; }

int launch_bottleneck(const BottleneckLaunch &l, hipStream_t s) {
    const TensorView &in = l.in, &out = l.out, &res = l.res;
    RT_CHECK(bottleneck_supported(l.c), RTMODT_E_UNSUPPORTED, "launch_bottleneck: %d channels", l.c);
    RT_CHECK(in.pad == 1 && in.c == l.c && out.c == l.c && in.H == out.H && in.W == out.W && l.kp == 9 * l.c, RTMODT_E_INVALID,
             "launch_bottleneck: shapes");
    RT_CHECK(in.coff % 8 == 0 && in.C % 8 == 0 && out.coff % 8 == 0 && out.C % 8 == 0 && l.zeros, RTMODT_E_INVALID, "launch_bottleneck: alignment");
    for (const TensorView *v : {&in, &out, &res})
        RT_CHECK(!v->base || (uintptr_t)v->base >= (1ull << 32), RTMODT_E_INVALID, "launch_bottleneck: view base %p is not a device address", (void *)v->base);
    BneckArgs a{};
    a.in = in.base + in.coff; a.w1 = l.w1; a.w2 = l.w2; a.b1 = l.b1; a.b2 = l.b2; a.zeros = l.zeros;
    a.out = out.base + out.coff;
    a.res = res.base ? res.base + res.coff : nullptr;
    a.B = l.B; a.H = in.H; a.W = in.W;
    a.in_Hp = in.H + 2; a.in_Wp = in.W + 2; a.in_cs = in.C;
    a.out_Hp = out.H + 2 * out.pad; a.out_Wp = out.W + 2 * out.pad; a.out_cs = out.C; a.out_pad = out.pad;
    a.res_Hp = res.H + 2 * res.pad; a.res_Wp = res.W + 2 * res.pad; a.res_cs = res.C; a.res_pad = res.pad;
    a.kp = l.kp;
    if (a.res) RT_CHECK(res.H == out.H && res.W == out.W && res.c == l.c && res.coff % 4 == 0, RTMODT_E_INVALID, "launch_bottleneck: residual shape");
    if (l.tail_wt) {                                       // C2f.cv2 fused as a tail (c = 32 -> 64 outputs, c = 64 -> 128 outputs)
        const TensorView &ti = l.tail_in, &to = l.tail_out;
        RT_CHECK((l.c == 32 || l.c == 64) && l.tail_cout == 2 * l.c && l.tail_kp == 3 * l.c && l.tail_bias, RTMODT_E_INVALID,
                 "launch_bottleneck: tail shapes (c %d, tail cout %d kp %d)", l.c, l.tail_cout, l.tail_kp);
        RT_CHECK(ti.base && ti.pad == 1 && ti.c == 2 * l.c && ti.H == in.H && ti.W == in.W && ti.coff % 8 == 0 && ti.C % 8 == 0 && to.base && to.H == in.H &&
                     to.W == in.W && to.c == l.tail_cout && to.coff % 8 == 0 && to.C % 8 == 0,
                 RTMODT_E_INVALID, "launch_bottleneck: tail views");
        for (const TensorView *v : {&ti, &to})
            RT_CHECK((uintptr_t)v->base >= (1ull << 32), RTMODT_E_INVALID, "launch_bottleneck: view base %p is not a device address", (void *)v->base);
        a.t_in = ti.base + ti.coff; a.t_wt = l.tail_wt; a.t_bias = l.tail_bias; a.t_out = to.base + to.coff;
        a.t_k1 = 2 * l.c; a.t_cout = l.tail_cout; a.t_kp = l.tail_kp; a.t_act = l.tail_act;
        a.t_in_Hp = ti.H + 2; a.t_in_Wp = ti.W + 2; a.t_in_cs = ti.C;
        a.t_out_Hp = to.H + 2 * to.pad; a.t_out_Wp = to.W + 2 * to.pad; a.t_out_cs = to.C; a.t_out_pad = to.pad;
        if (l.c == 32 && l.persistent32 && bottleneck32_tail_supported(l)) return launch_bottleneck32_tail(l, s);
        if (l.c == 32) return launch_one<32, 16, 16, 4>(a, in.H, in.W, l.B, s);
        return launch_one<64, 16, 16, 8>(a, in.H, in.W, l.B, s);
    }
    switch (l.c) {
        case 32: return launch_one<32, 16, 16>(a, in.H, in.W, l.B, s);
        case 64: return launch_one<64, 16, 16>(a, in.H, in.W, l.B, s);
        case 128: return launch_one<128, 8, 16>(a, in.H, in.W, l.B, s);
    }
    return fail(RTMODT_E_UNSUPPORTED, "launch_bottleneck: %d channels", l.c);
}

}  // namespace rtmodt
