// bneck32.hip -- the 32-channel C2f of YOLOv8s (layer 2: cv1's output [y0 | y1] -> Bottleneck(y1) -> cv2 over [y0 | y1 | y2]) behind its first
// 1x1 conv, in ONE launch of PERSISTENT workgroups for gfx950:
//     y2 = y1 + SiLU(conv3x3(SiLU(conv3x3(y1))))          (the Bottleneck, shortcut optional)
//     out = SiLU(W2 . [y0 | y1 | y2] + b)                  (C2f.cv2, 96 -> 64)
// Replaces bottleneck_fused<32, 16, 16, 4> (bottleneck.hip) for this shape -- SURVEY App. A row 2; what ultralytics runs behind
// /root/reference/src/detection/detector.py:100-111.  That kernel is one 152-KiB workgroup per CU and per tile: every tile starts with an empty LDS,
// waits for its patch, its weights (twice) and its concat chunks from HBM / L2, and runs nine barrier phases of ~31 000 clk around 2 900 clk of MFMA
// (profiles/r03).  Here, as in front.hip:
//   * a 256-thread workgroup owns an 8 x 16 tile and is PERSISTENT (tiles t, t + G, ...); 76 KiB of LDS -> two workgroups per CU that drift against
//     each other (one's SiLU epilogue under the other's MFMAs);
//   * the weights of both 3x3 convs live in REGISTERS (2 x 18 fragments per wave, loaded once per workgroup), cv2's in LDS (12 KiB, once);
//   * the next tile's 12 x 20 patch of y1 is requested by LDS-DMA while this tile is computed (two patch buffers), its y0 tile behind this tile's cv2;
//   * conv1 runs over the patch LINEARLY (20 positions per row, the two extra columns are junk): output position m reads patch rows m + 20 kh + kw, so
//     every LDS address is (one of a few per-lane registers) + (compile-time offset) -- with row pitches that are multiples of 4 the swizzle term
//     is an XOR with a constant.
// Arithmetic order and roundings are bottleneck_fused's (bias as the initial accumulator of the 3x3 convs, taps in (kh, kw) order, SiLU -> fp16 for the
// intermediate, SiLU + shortcut in fp32 -> fp16 for y2, cv2 over [y0 | y1 | y2] in concat order, + bias, SiLU): outputs are BIT-IDENTICAL to it.
#include "conv_dev.h"

namespace rtmodt {

namespace {

constexpr int B_TH = 8, B_TW = 16;
constexpr int B_PW = B_TW + 4, B_PH = B_TH + 4;           // 12 x 20 patch of y1 (2-pixel halo)
constexpr int B_M1 = (B_TH + 2) * B_PW;                   // conv1 runs over 10 rows x 20 positions (18 + 2 junk columns) = 200
constexpr int B_G1 = (B_M1 + 15) / 16;                    // 13 groups of 16 positions
constexpr int PATCH_ROWS = 256, PATCH_BYTES = PATCH_ROWS * 64;      // 240 patch rows + slack for the junk positions' reads (up to row 249)
constexpr int P0_OFF = 0, P1_OFF = PATCH_BYTES;
constexpr int Y0_OFF = 2 * PATCH_BYTES, Y0_BYTES = B_TH * B_TW * 64;
constexpr int TB_OFF = Y0_OFF + Y0_BYTES, TB_BYTES = 16 * 1024;      // conv1's output (208 rows x 64 B); later the output tile [128 px][128 B]
constexpr int Y2_OFF = TB_OFF + TB_BYTES, Y2_BYTES = B_TH * B_TW * 64;
constexpr int WT_OFF = Y2_OFF + Y2_BYTES, WT_BYTES = 12 * 1024;      // cv2's weights as fragments [cout tile 4][k chunk 3][lane]
constexpr int BT_OFF = WT_OFF + WT_BYTES, BT_BYTES = 512;            // biases [cout tile][q]: cv2's (4 tiles), then conv1's and conv2's (2 tiles each)
constexpr int B1_OFF = BT_OFF + 256, B2_OFF = BT_OFF + 384;
constexpr int B_LDS = BT_OFF + BT_BYTES;
static_assert(B_G1 * 16 <= TB_BYTES / 64 && B_G1 * 16 + 2 * B_PW + 2 <= PATCH_ROWS && B_TH * B_TW * 128 <= TB_BYTES, "LDS map");
static_assert(B_LDS <= 80 * 1024, "two workgroups per CU");

struct B32Args {
    const f16 *cat;          // the C2f concat tensor at channel 0 of [y0 | y1] (tail_in's slice); y1 = + c halves
    const f16 *zeros;
    const f16 *w1, *w2, *wt; const float *b1, *b2, *bt; int kp, kpt;
    f16 *out;                // cv2's output view (channel offset applied)
    int H, W, Hp, Wp, cs;    // image, padded tensor dims, pixel stride (halves) of the concat tensor
    int out_Hp, out_Wp, out_cs, out_pad;
    int tiles_x, tiles_y, n_tiles, shortcut, act;
};

#if defined(RTMODT_STAMP)
#define BST(k) do { if (t == (int)(blockIdx.x + gridDim.x)) STAMP(k); } while (0)
#else
#define BST(k)
#endif

__global__ __launch_bounds__(256, 2) void c2f32_fused(B32Args a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[B_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;

    // ---- weights: both 3x3 convs -> registers; cv2 -> LDS as fragments ----
    half8 w1f[2][9], w2f[2][9];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            w1f[u][tap] = *(const half8 *)(a.w1 + (size_t)(u * 16 + p) * a.kp + tap * 32 + q * 8);
            w2f[u][tap] = *(const half8 *)(a.w2 + (size_t)(u * 16 + p) * a.kp + tap * 32 + q * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {                          // fragment f = (cout tile ug, k chunk kc) = 3 * wave + i
        const int f = 3 * wave + i, ug = f / 3, kc = f - ug * 3;
        *(half8 *)(lds + WT_OFF + f * 1024 + lane * 16) = *(const half8 *)(a.wt + (size_t)(ug * 16 + p) * a.kpt + kc * 32 + q * 8);
    }
    if (tid < 16) *(floatx4 *)(lds + BT_OFF + tid * 16) = *(const floatx4 *)(a.bt + (tid >> 2) * 16 + (tid & 3) * 4);
    else if (tid < 24) *(floatx4 *)(lds + B1_OFF + (tid - 16) * 16) = *(const floatx4 *)(a.b1 + ((tid - 16) >> 2) * 16 + (tid & 3) * 4);
    else if (tid < 32) *(floatx4 *)(lds + B2_OFF + (tid - 24) * 16) = *(const floatx4 *)(a.b2 + ((tid - 24) >> 2) * 16 + (tid & 3) * 4);

    // ---- per-lane LDS addresses; 64-byte rows, chunk c of row R in slot c ^ 2 * bit 2 of R (tile_math.h swz_slot<64>) ----
    int rd3[3];                                            // fragment (chunk q) of row (multiple of 4) + p + kw; odd multiples of 4: ^ 32
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) rd3[kw] = (p + kw) * 64 + ((q ^ ((((p + kw) >> 2) & 1) << 1)) << 4);
    const int rd0 = rd3[0];
    const int wr = p * 64 + (((q >> 1) ^ (((p >> 2) & 1) << 1)) << 4) + (q & 1) * 8;                   // this lane's 8 bytes (cout tile 0) of row (multiple of 4 .. 16) + p; tile 1: ^ 32
    const int rs = (p + 2) * 64 + (((q >> 1) ^ ((((p + 2) >> 2) & 1) << 1)) << 4) + (q & 1) * 8;       // ... of row (multiple of 4) + p + 2: the shortcut's y1
    const int rot = 2 * (p >> 1);
    const int stw_b = p * 128 + (q & 1) * 8, stw_c = (q >> 1) + rot;      // output tile [pixel][128 B]: this lane's 8 bytes of cout tile ug, row p: stw_b + (((stw_c + 2 ug) & 7) << 4)
    const int st_px = tid >> 3;
    const int st_rd = TB_OFF + st_px * 128 + ((((tid & 7) + 2 * (st_px >> 1)) & 7) << 4);
    const int st_go = ((st_px >> 4) * a.out_Wp + (st_px & 15)) * a.out_cs + (tid & 7) * 8;

    // ---- DMA sources, tile-independent part: byte offset from the tile's origin pixel, flags (patch edge rows / columns) in the low 4 bits ----
    const int dchunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);       // the 16-byte chunk whose LDS slot this lane fills (rows = multiples of 16 + lane >> 2)
    int pt_off[4], y0_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = min((wave + 4 * i) * 16 + (lane >> 2), B_PH * B_PW - 1), py = R / B_PW, px = R - py * B_PW;
        pt_off[i] = (((py * a.Wp + px) * a.cs + 32 + dchunk * 8) * 2) | (py == 0 ? 1 : 0) | (py == B_PH - 1 ? 2 : 0) | (px == 0 ? 4 : 0) | (px == B_PW - 1 ? 8 : 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) y0_off[i] = (((wave + 4 * i) * a.Wp + (lane >> 2)) * a.cs + dchunk * 8) * 2;

    const int per_img = a.tiles_x * a.tiles_y;
    auto decode = [&](int t, int &b, int &y0, int &x0) {
        b = t / per_img;
        const int r = t - b * per_img, ty = r / a.tiles_x;
        y0 = ty * B_TH; x0 = (r - ty * a.tiles_x) * B_TW;
    };
    auto issue_patch = [&](int t, int buf) {               // y1 of the tile's 12 x 20 patch -> patch buffer `buf`; what lies outside the bordered tensor: zero page
        int b, y0, x0;
        decode(t, b, y0, x0);
        const char *org = (const char *)a.cat + ((long)(b * a.Hp + y0 - 1) * a.Wp + x0 - 1) * a.cs * 2;      // pixel (y0 - 2, x0 - 2) in padded coordinates
        const int edge = (y0 == 0 ? 1 : 0) | (y0 + B_TH == a.H ? 2 : 0) | (x0 == 0 ? 4 : 0) | (x0 + B_TW == a.W ? 8 : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char *src = (pt_off[i] & edge) ? (const char *)a.zeros : org + (pt_off[i] & ~15);
            glds16((const f16 *)src, lds + (buf ? P1_OFF : P0_OFF) + (wave + 4 * i) * 1024);
        }
    };
    auto issue_y0 = [&](int t) {                           // y0 of the tile's 8 x 16 pixels
        int b, y0, x0;
        decode(t, b, y0, x0);
        const char *org = (const char *)a.cat + ((long)(b * a.Hp + y0 + 1) * a.Wp + x0 + 1) * a.cs * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16((const f16 *)(org + y0_off[i]), lds + Y0_OFF + (wave + 4 * i) * 1024);
    };

    int t = blockIdx.x;
    if (t >= a.n_tiles) return;
    issue_patch(t, 0);
    issue_y0(t);
    bool first = true;
    int buf = 0;

    for (; t < a.n_tiles; t += gridDim.x, buf ^= 1) {
        int b, y0, x0;
        decode(t, b, y0, x0);
        const int PB = buf ? P1_OFF : P0_OFF;
        const bool more = t + (int)gridDim.x < a.n_tiles;
        BST(0);
        // oldest first: this tile's patch (4 pieces per wave) and y0 (2), then the previous tile's four stores, which may stay in flight
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        first = false;
        __syncthreads();                                   // P0: patch and y0 have landed; every wave has left the previous tile
        BST(1);
        if (more) issue_patch(t + gridDim.x, buf ^ 1);     // the other patch buffer was last read in the previous tile's cv2

        // ---- P1: conv1 over the linearised patch, groups wave, wave + 4, wave + 8 (+ 12 for wave 0); positions outside the image are conv2's zero padding.
        // (A hand-pipelined form -- fragments of group i + 2 and MFMAs of group i + 1 in front of the SiLU epilogue of group i -- needs a second fragment
        // set beside the 144 weight registers and spills 35 VGPRs: not kept.  Neither is the form with ONE cout tile's weights per wave (72 registers, the
        // pipeline then fits): every position fragment is then read by two waves, conv1's phase went 5 450 -> 6 350 clk, the launch 113 -> 119 us:
        // these 32-channel convs are bound by LDS fragment reads as soon as a fragment feeds fewer than two MFMAs -- gpurun_out/r05/bneck32_probe2.txt.) ----
        {
            const int ng = wave == 0 ? 4 : 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i >= ng) break;
                const int g = wave + 4 * i;
                floatx4 acc[2] = {*(const floatx4 *)(lds + B1_OFF + q * 16), *(const floatx4 *)(lds + B1_OFF + 64 + q * 16)};      // bias = the initial accumulator
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const half8 fr = *(const half8 *)(lds + PB + (rd3[kw] ^ ((kh & 1) << 5)) + (16 * g + B_PW * kh) * 64);
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[u][kh * 3 + kw], fr, acc[u], 0, 0, 0);
                    }
                const int m = 16 * g + p, iy = m / B_PW, ix = m - iy * B_PW;
                const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
                const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    silu4(acc[u]);
                    half4 h = {(f16)acc[u][0], (f16)acc[u][1], (f16)acc[u][2], (f16)acc[u][3]};
                    if (!inside) h = half4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                    *(half4 *)(lds + TB_OFF + (wr ^ (u << 5)) + g * 1024) = h;
                }
            }
        }
        BST(2);
        __syncthreads();                                   // conv1's image complete
        BST(3);

        // ---- P2: conv2 on output rows 2 wave, 2 wave + 1; SiLU, + shortcut (y1 from the patch) in fp32, one rounding -> y2 tile ----
        {
            const floatx4 b2a = *(const floatx4 *)(lds + B2_OFF + q * 16), b2b = *(const floatx4 *)(lds + B2_OFF + 64 + q * 16);
            floatx4 acc[2][2] = {{b2a, b2b}, {b2a, b2b}};
            const int rb = 2 * wave * B_PW * 64;           // byte offset of this wave's first row in a 20-wide image
#pragma unroll
            for (int s = 0; s < 4; ++s)                    // image row 2 wave + s serves kh = s of row j = 0 and kh = s - 1 of row j = 1
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const half8 fr = *(const half8 *)(lds + TB_OFF + rb + (rd3[kw] ^ ((s & 1) << 5)) + s * B_PW * 64);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int kh = s - j;
                        if (kh < 0 || kh > 2) continue;
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[j][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[u][kh * 3 + kw], fr, acc[j][u], 0, 0, 0);
                    }
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    floatx4 v = acc[j][u];
                    silu4(v);
                    if (a.shortcut) {
                        const half4 y1 = *(const half4 *)(lds + PB + rb + ((rs ^ ((j & 1) << 5)) ^ (u << 5)) + (j + 2) * B_PW * 64);
                        v[0] += (float)y1[0]; v[1] += (float)y1[1]; v[2] += (float)y1[2]; v[3] += (float)y1[3];
                    }
                    *(half4 *)(lds + Y2_OFF + (wr ^ (u << 5)) + (2 * wave + j) * 1024) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                }
        }
        BST(4);
        __syncthreads();                                   // y2 complete (the tail of a row reads both cout tiles); conv1's image is dead
        BST(5);

        // ---- P3: cv2 over [y0 | y1 | y2] for rows 2 wave, 2 wave + 1, all 64 couts (two cout tiles at a time: registers); weight fragments from LDS ----
        {
            const int rb = 2 * wave * B_PW * 64;
#pragma unroll
            for (int uh = 0; uh < 2; ++uh) {
                floatx4 acc[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[j][u] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < 3; ++kc) {
                    half8 fa[2], fw[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) fw[u] = *(const half8 *)(lds + WT_OFF + ((2 * uh + u) * 3 + kc) * 1024 + lane * 16);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        fa[j] = kc == 1 ? *(const half8 *)(lds + PB + rb + (rd3[2] ^ ((j & 1) << 5)) + (j + 2) * B_PW * 64)
                                        : *(const half8 *)(lds + (kc == 0 ? Y0_OFF : Y2_OFF) + rd0 + (2 * wave + j) * 1024);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[j][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[u], fa[j], acc[j][u], 0, 0, 0);
                }
                if (uh == 1) BST(6);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ug = 2 * uh + u;
                    const floatx4 bv = *(const floatx4 *)(lds + BT_OFF + (ug * 4 + q) * 16);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        floatx4 v = acc[j][u] + bv;
                        if (a.act) silu4(v);
                        *(half4 *)(lds + TB_OFF + stw_b + (((stw_c + 2 * ug) & 7) << 4) + (2 * wave + j) * 2048) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    }
                }
            }
        }
        BST(7);
        __syncthreads();                                   // the output tile is complete; y0 / y2 / this patch buffer are dead
        if (more) issue_y0(t + gridDim.x);

        // ---- P4: 16-byte NHWC stores ----
        f16 *const orow = a.out + ((size_t)(b * a.out_Hp + y0 + a.out_pad) * a.out_Wp + x0 + a.out_pad) * a.out_cs + st_go;
        const int row2 = 2 * a.out_Wp * a.out_cs;
#pragma unroll
        for (int k = 0; k < (B_TH * B_TW * 8) / 256; ++k) *(half8 *)(orow + k * row2) = *(const half8 *)(lds + st_rd + k * 4096);
        BST(8);
    }
}

}  // namespace

bool bottleneck32_tail_supported(const BottleneckLaunch &l) {
    const TensorView &in = l.in, &ti = l.tail_in, &to = l.tail_out;
    return l.c == 32 && l.tail_wt && l.tail_cout == 64 && l.tail_kp == 96 && l.kp == 288 && in.H % B_TH == 0 && in.W % B_TW == 0 && in.pad == 1 && ti.pad == 1 &&
           ti.base == in.base && ti.C == in.C && in.coff == ti.coff + 32 && ti.c == 64 && ti.coff % 8 == 0 && in.C % 8 == 0 &&
           (!l.res.base || (l.res.base == in.base && l.res.coff == in.coff && l.res.C == in.C && l.res.pad == in.pad)) &&
           to.base && to.H == in.H && to.W == in.W && to.c == 64 && to.coff % 8 == 0 && to.C % 8 == 0 &&
           (long)l.B * (in.H + 2) * (in.W + 2) * in.C < (1L << 30);
}

int launch_bottleneck32_tail(const BottleneckLaunch &l, hipStream_t s) {
    RT_CHECK(bottleneck32_tail_supported(l), RTMODT_E_UNSUPPORTED, "launch_bottleneck32_tail: shape");
    RT_CHECK(l.w1 && l.w2 && l.b1 && l.b2 && l.tail_bias && l.zeros, RTMODT_E_INVALID, "launch_bottleneck32_tail: null operand");
    for (const TensorView *v : {&l.in, &l.tail_in, &l.tail_out})
        RT_CHECK((uintptr_t)v->base >= (1ull << 32), RTMODT_E_INVALID, "launch_bottleneck32_tail: view base %p is not a device address", (void *)v->base);
    B32Args a{};
    a.cat = l.tail_in.base + l.tail_in.coff; a.zeros = l.zeros;
    a.w1 = l.w1; a.w2 = l.w2; a.wt = l.tail_wt; a.b1 = l.b1; a.b2 = l.b2; a.bt = l.tail_bias; a.kp = l.kp; a.kpt = l.tail_kp;
    const TensorView &to = l.tail_out;
    a.out = to.base + to.coff; a.out_Hp = to.H + 2 * to.pad; a.out_Wp = to.padded_w(); a.out_cs = to.C; a.out_pad = to.pad;
    a.H = l.in.H; a.W = l.in.W; a.Hp = l.in.H + 2; a.Wp = l.in.W + 2; a.cs = l.in.C;
    a.tiles_x = a.W / B_TW; a.tiles_y = a.H / B_TH; a.n_tiles = l.B * a.tiles_x * a.tiles_y;
    a.shortcut = l.res.base != nullptr; a.act = l.tail_act;
    static const int per_cu = rt_diag("FRONT_WGS") ? std::max(1, atoi(rt_diag("FRONT_WGS"))) : 2;      // (diagnostic builds: one workgroup per CU, as in front.hip)
    const int G = std::min(a.n_tiles, per_cu * device_cus());
    hipLaunchKernelGGL(c2f32_fused, dim3(G), dim3(256), 0, s, a);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

}  // namespace rtmodt
