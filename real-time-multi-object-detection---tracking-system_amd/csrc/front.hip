// front.hip -- the FRONT END of the YOLOv8 forward pass in ONE launch for gfx950:
//     letterbox + BGR->RGB + /255 + .half()  ->  layer 0 (Conv 3->32, 3x3 / s2)  ->  layer 1 (Conv 32->64, 3x3 / s2)  ->  2.cv1 (C2f's first 1x1, 64->64)
// with neither the stem's 320 x 320 x 32 output nor layer 1's 160 x 160 x 64 output ever leaving the CU.
//
// What it replaces (what ultralytics runs behind /root/reference/src/detection/detector.py:100-111): stem_fused (conv.hip) +
// conv_mfma_tail (layer 1 with 2.cv1 as its tail).  Those two launches were 18 % of the forward pass for 9 % of its FLOPs: the stem WROTE
// 205 MB per 32-frame step and layer 1 FETCHED it twice -- a tensor with exactly one consumer.  Here the only HBM traffic is the frames' bytes in
// (1.23 MB per frame) and 2.cv1's output (3.3 MB per frame).
//
// Work decomposition: one 256-thread workgroup (4 waves) per TILE of 8 x 16 layer-1 output pixels, persistent over tiles t, t + G, ...; ~63 KiB of LDS,
// so two workgroups share a CU and drift against each other -- while one runs a SiLU epilogue (VALU) the other feeds the matrix pipe.  Per tile:
//   T0  the tile's 35 x 68 source pixels (4 y0 - 3 ..., 4 x0 - 3 ...: the receptive field of 17 x 33 stem outputs) are in LDS as RAW BYTES -- whole aligned
//       16-byte chunks by LDS-DMA, requested while the PREVIOUS tile was being computed;
//   T1  bytes -> (R, G, B, 0) fp16 pixels (four pixels = twelve bytes per work item, v_alignbyte; 114 where the letterbox canvas has no image, 0 outside
//       the canvas: the stem's zero padding); then the DMA of the NEXT tile's bytes is issued;
//   T3  stem: 17 x 33 positions x 32 couts on v_mfma_f32_16x16x32_f16 (K re-indexed kh * 16 + kw * 4 + c as in conv.hip's stem), bias = initial
//       accumulator, SiLU, fp16 -> LDS.  Positions are kept in TWO planes (even / odd column) so that layer 1's stride-2 taps read 16 CONSECUTIVE rows;
//   T4  layer 1: 128 pixels x 64 couts x K = 288.  A wave owns 4 output rows x 32 couts; its 18 weight fragments live in REGISTERS for the whole
//       launch (loaded once per workgroup, not per tile); a stem row's fragment serves kh = 2 of one output row and kh = 0 of the next.  Epilogue:
//       + bias, SiLU, fp16 -> LDS tile [pixel][64 channels];
//   T5  2.cv1 on that tile (weights in registers), + bias, SiLU, fp16 -> LDS;  T6  16-byte NHWC stores, whole 128-byte lines per pixel.
// Arithmetic order and roundings are those of the two launches it replaces (same MFMA instruction, same k order, bias added where they add it, fp16
// at the same two places), so the stored 2.cv1 tensor is BIT-IDENTICAL to theirs (tests/test_gpu_detector.py::test_front_fused_is_bit_identical).
//
// Second source (`img`): the letterboxed RGB0 fp16 image tensor (frames that needed a resize went through preprocess.hip): its pixels are DMA'd
// straight into the pixel tile, T1 disappears.
#include <type_traits>

#include "conv_dev.h"

namespace rtmodt {

namespace {

constexpr int F_TH = 8, F_TW = 16;                      // layer-1 output tile
constexpr int F_SH = 2 * F_TH + 1, F_SW = 2 * F_TW + 1; // 17 x 33 stem positions feed it
constexpr int F_PH = 2 * F_SH + 1, F_PW = 2 * F_SW + 2; // 35 rows x 68 columns of source pixels (the last column only meets the stem's zero 4th tap)
constexpr int F_PIXROW = F_PW * 8;                      // bytes per pixel row (RGB0 fp16)
constexpr int F_RAWCH = 14, F_RAWROW = F_RAWCH * 16;    // raw bytes per row: 3 * 68 = 204 + up to 15 of misalignment -> 14 chunks
// Stem outputs live in two planes of 64-byte rows: EVEN columns 0, 2, ..., 32 (17 per stem row, row pitch 24) and ODD columns 1, ..., 31 (16 per row,
// pitch 16).  Layer 1's taps kw = 0 / 1 / 2 of output columns p = 0..15 then read rows (pitch * stem row) + p of the even / odd / even (+ 1) plane:
// sixteen CONSECUTIVE rows -- and with pitches that are multiples of 8 the swizzle term of a row depends on the lane only, so every address in this
// kernel is (one per-lane base register) + (compile-time offset): without that the hoisted address registers alone spill the kernel.
constexpr int F_EP = 24, F_OP = 16;
constexpr int RAW_OFF = 0, RAW_BYTES = 8 * 1024;                               // 35 x 14 = 490 chunks -> 8 wave instructions
constexpr int PIX_OFF = RAW_OFF + RAW_BYTES, PIX_BYTES = 19 * 1024;            // 35 x 544 = 19 040 -> 19 wave instructions (tensor source)
constexpr int SE_OFF = PIX_OFF + PIX_BYTES, SE_BYTES = F_SH * F_EP * 64;
constexpr int SO_OFF = SE_OFF + SE_BYTES, SO_BYTES = F_SH * F_OP * 64;
constexpr int W2_OFF = SO_OFF + SO_BYTES, W2_BYTES = 8 * 1024;                 // 2.cv1's weights as MFMA fragments [cout tile][k half][lane]
constexpr int B2_OFF = W2_OFF + W2_BYTES, B2_BYTES = 256;                      // its bias [cout tile][q]
constexpr int F_LDS = B2_OFF + B2_BYTES;
constexpr int Y1_OFF = PIX_OFF, Y2_OFF = SE_OFF;                               // layer 1's tile reuses the pixel tile, 2.cv1's the even stem plane
static_assert(F_PH * F_RAWROW <= RAW_BYTES && F_PH * F_PIXROW <= PIX_BYTES && F_TH * F_TW * 128 <= PIX_BYTES && F_TH * F_TW * 128 <= SE_BYTES, "LDS map");
static_assert(F_LDS <= 80 * 1024, "two workgroups per CU");

struct FrontArgs {
    FramePtrs frames; int frame0, pitch, top, left, new_h, new_w, frame_bytes;      // byte source (frames that need no resize)
    const f16 *img; int img_Hp, img_Wp;                   // tensor source: RGB0 fp16 with a 1-pixel zero border (nullptr: byte source)
    const f16 *zeros;                                     // >= 16 bytes of zeros in device memory
    const f16 *w0, *w1, *w2; const float *b0, *b1, *b2; int kp1, kp2;
    f16 *out; int out_Hp, out_Wp, out_cs, out_pad;        // 2.cv1's output view (channel offset applied)
    int in_h, in_w, tiles_x, tiles_y, n_tiles;
};

// 64-byte rows: chunk c of row R at slot c ^ 2 * bit 2 of R (tile_math.h swz_slot<64>: conflict-free ds_read_b128 from any 16 consecutive rows);
// 128-byte rows: slot (c + 2 * (R >> 1)) & 7 (swz_slot<128>)
__device__ __forceinline__ int s_off(int R, int c) { return R * 64 + (swz_slot<64>(R, c) << 4); }
__device__ __forceinline__ int y_off(int R, int c) { return R * 128 + (swz_slot<128>(R, c) << 4); }

// diagnostic build only (-DRTMODT_STAMP, tools/probes/front_probe.hip): phase stamps of a workgroup's SECOND tile (steady state)
#if defined(RTMODT_STAMP)
#define FST(k) do { if (t == (int)(blockIdx.x + gridDim.x)) STAMP(k); } while (0)
#else
#define FST(k)
#endif

template <bool TENSOR>
__global__ __launch_bounds__(256, 2) void front_fused(FrontArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[F_LDS];
    unsigned char *const raw = lds + RAW_OFF, *const pix = lds + PIX_OFF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int ch = wave & 1, ph = wave >> 1;               // layer 1 / 2.cv1: this wave's cout half (32 couts) and pixel half (4 rows)

    // ---- weights and biases: stem and layer 1 -> registers, 2.cv1 -> LDS (as fragments), once per workgroup ----
    half8 w0f[2][2], w1f[2][9];
    floatx4 b0v[2], b1v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) w0f[u][kk] = *(const half8 *)(a.w0 + (u * 16 + p) * 64 + kk * 32 + q * 8);
        b0v[u] = *(const floatx4 *)(a.b0 + u * 16 + q * 4);
        const int n = (2 * ch + u) * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) w1f[u][tap] = *(const half8 *)(a.w1 + (size_t)(n + p) * a.kp1 + tap * 32 + q * 8);
        b1v[u] = *(const floatx4 *)(a.b1 + n + q * 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {                          // fragment (cout tile ug, k half kc) = 2 * wave + i of 2.cv1's weights
        const int fr = 2 * wave + i, ug = fr >> 1, kc = fr & 1;
        *(half8 *)(lds + W2_OFF + fr * 1024 + lane * 16) = *(const half8 *)(a.w2 + (size_t)(ug * 16 + p) * a.kp2 + kc * 32 + q * 8);
    }
    if (tid < 16) *(floatx4 *)(lds + B2_OFF + tid * 16) = *(const floatx4 *)(a.b2 + (tid >> 2) * 16 + (tid & 3) * 4);

    // ---- per-lane LDS addresses (bytes from `lds`), tile-independent; everything else is a compile-time offset from one of these ----
    const int sw = ((p >> 2) & 1) << 1;                                                    // swizzle term of rows (multiple of 8) + p
    const int st_src = PIX_OFF + (4 * p + 2 * (q & 1)) * 8 + (q >> 1) * F_PIXROW;           // stem fragment of even column 2 p, kernel rows 0 / 1, stem row 0
    const int st_src2 = PIX_OFF + (4 * p + 2 * (q & 1)) * 8 + 2 * F_PIXROW;                 // ... kernel row 2
    const int st_dst = p * 64 + (((q >> 1) ^ sw) << 4) + (q & 1) * 8;                       // where (row pitch * g + p, cout tile 0) goes inside a plane; tile 1: ^ 32
    const int l1_e0 = SE_OFF + (8 * ph * F_EP + p) * 64 + ((q ^ sw) << 4);                  // layer-1 fragment, kw = 0, stem row 8 ph
    const int l1_e2 = SE_OFF + (8 * ph * F_EP + p + 1) * 64 + ((q ^ ((((p + 1) >> 2) & 1) << 1)) << 4);      // kw = 2
    const int l1_o = SO_OFF + (8 * ph * F_OP + p) * 64 + ((q ^ sw) << 4);                   // kw = 1
    const int rot = 2 * (p >> 1);                                                           // rotation of 128-byte rows (multiple of 16) + p
    int y_wr[2], y_rd[2];                                                                   // tile [pixel][64 channels]: this wave's writes (cout tile u) and reads (k half kc), row 4 ph
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        y_wr[u] = (64 * ph + p) * 128 + ((((2 * ch + u) * 2 + (q >> 1) + rot) & 7) << 4) + (q & 1) * 8;
        y_rd[u] = (64 * ph + p) * 128 + (((u * 4 + q + rot) & 7) << 4);
    }
    const int st_px = tid >> 3;                                                             // T6: pixel (+ 32 k) and chunk of this thread's stores
    const int st_rd = Y2_OFF + st_px * 128 + ((((tid & 7) + 2 * (st_px >> 1)) & 7) << 4);
    const int st_go = ((st_px >> 4) * a.out_Wp + (st_px & 15)) * a.out_cs + (tid & 7) * 8;

    const int per_img = a.tiles_x * a.tiles_y;
    auto decode = [&](int t, int &b, int &y0, int &x0) {
        b = t / per_img;
        const int r = t - b * per_img, ty = r / a.tiles_x;
        y0 = ty * F_TH; x0 = (r - ty * a.tiles_x) * F_TW;
    };
    // ---- source DMA.  Which (row, chunk) of a tile's source a lane fetches does not depend on the tile: worked out once. ----
    // byte source: 35 rows x 14 chunks = 490 chunks = 8 wave instructions (2 per wave); tensor source: 35 x 34 = 1 190 chunks = 19 (4 or 5 per wave)
    constexpr int NSRC = TENSOR ? 5 : 2, SRC_CH = TENSOR ? F_PW / 2 : F_RAWCH, SRC_INSTR = TENSOR ? 19 : 8;
    int src_rj[NSRC];
#pragma unroll
    for (int i = 0; i < NSRC; ++i) {
        const int g = (wave + 4 * i) * 64 + lane;
        src_rj[i] = min(g / SRC_CH, F_PH - 1) | ((g - (g / SRC_CH) * SRC_CH) << 8);
    }
    auto issue_src = [&](int t) {
        int b, y0, x0;
        decode(t, b, y0, x0);
        const int cy0 = 4 * y0 - 3, cx0 = 4 * x0 - 3;
        if constexpr (TENSOR) {
            // pixel rows straight from the image tensor: chunk j of row r = canvas pixels (cx0 + 2 j, + 1) of canvas row cy0 + r; what lies outside the
            // bordered tensor (rows / columns -3, -2 of the first tiles) comes from the zero page
            const f16 *img_b = a.img + (size_t)b * a.img_Hp * a.img_Wp * 4;
#pragma unroll
            for (int i = 0; i < NSRC; ++i) {
                if (wave + 4 * i >= SRC_INSTR) break;
                const int ty = cy0 + (src_rj[i] & 255) + 1, tx = cx0 + 2 * (src_rj[i] >> 8) + 1;      // padded tensor coordinates
                const f16 *src = ty >= 0 && tx >= 0 ? img_b + (ty * a.img_Wp + tx) * 4 : a.zeros;
                glds16(src, pix + (wave + 4 * i) * 1024);
            }
        } else {
            const uint8_t *f = a.frames.p[a.frame0 + b];
            const int mis = (int)((uintptr_t)f & 15);
            const uint8_t *flo = f - mis;
            const int fspan = (mis + a.frame_bytes + 15) & ~15;                 // (frame_bytes < 2^31: checked at launch)
            const int col0 = mis + 3 * (cx0 - a.left);
#pragma unroll
            for (int i = 0; i < NSRC; ++i) {
                const int sy = min(max(cy0 + (src_rj[i] & 255) - a.top, 0), a.new_h - 1);     // rows outside the image: any row of the frame (never read)
                int off = ((sy * a.pitch + col0) & ~15) + 16 * (src_rj[i] >> 8);              // the row's first needed byte, floor-aligned, + this lane's chunk
                off = min(max(off, 0), fspan - 16);                                           // chunks that hold no byte of the frame: clamped into its pages (never read)
                glds16((const f16 *)(flo + off), raw + (wave + 4 * i) * 1024);
            }
        }
    };

    // ---- stem building blocks for one group of 16 positions x 32 couts: fragments (two 16-byte reads), 2 x 2 MFMAs, SiLU + fp16 -> plane ----
    auto st_load = [&](int src, int src2, half8 &a0, half8 &a1) { a0 = *(const half8 *)(lds + src); a1 = *(const half8 *)(lds + src2); };
    auto st_mma = [&](const half8 &a0, const half8 &a1, floatx4 (&acc)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0f[u][0], a0, b0v[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0f[u][1], a1, acc[u], 0, 0, 0);
    };
    auto st_store = [&](floatx4 (&acc)[2], int dst, bool zero) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            silu4(acc[u]);
            half4 h = {(f16)acc[u][0], (f16)acc[u][1], (f16)acc[u][2], (f16)acc[u][3]};
            if (zero) h = half4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
            *(half4 *)(lds + (dst ^ (u * 32))) = h;
        }
    };

    int t = blockIdx.x;
    if (t >= a.n_tiles) return;
    issue_src(t);
    bool first = true;

    for (; t < a.n_tiles; t += gridDim.x) {
        int b, y0, x0;
        decode(t, b, y0, x0);
        const int cy0 = 4 * y0 - 3, cx0 = 4 * x0 - 3;
        FST(0);
        // this tile's source pieces are the OLDEST outstanding memory operations of the wave; the previous tile's four 16-byte stores were issued after
        // them and may stay in flight (vmcnt retires in issue order)
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        first = false;
        __syncthreads();                                   // T0: this tile's source has landed; every wave has left the previous tile
        FST(1);
        if constexpr (!TENSOR) {
            // ---- T1: bytes -> (R, G, B, 0) fp16; byte -> float -> * (1 / 255) -> half == the letterbox kernel's half(c / 255.f) for all 256 values ----
            const uint8_t *f = a.frames.p[a.frame0 + b];
            const int col0 = (int)((uintptr_t)f & 15) + 3 * (cx0 - a.left), pm = a.pitch & 15;
            // Branch-free, so that the (up to) three items of a thread are independent instruction streams the scheduler interleaves (LDS latency of one under
            // the conversions of another): the twelve bytes are always read (from inside the raw row, whatever they are), then every pixel picks its bytes,
            // the letterbox fill 114 (canvas outside the image) or 0 (outside the canvas: the stem's zero padding).  A tile whose 35 x 67 needed pixels all lie
            // inside the image (85 % of the tiles of a 640 x 640 frame; column 67 only meets the stem's zero tap and may hold any bytes) skips the masks --
            // 38 % of the masked form's instructions.
            const bool interior = cy0 >= a.top && cy0 + F_PH <= a.top + a.new_h && cx0 >= a.left && cx0 + F_PW - 1 <= a.left + a.new_w;      // (wave-uniform)
            auto convert = [&](auto masked_c) {
                constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
                for (int it = 0; it < (F_PH * (F_PW / 4) + 255) / 256; ++it) {
                    if (it * 256 + wave * 64 >= F_PH * (F_PW / 4)) break;            // (wave-uniform: the last pass has work for waves 0 and 1 only)
                    const int i = min(it * 256 + tid, F_PH * (F_PW / 4) - 1);        // (surplus lanes of the last wave repeat the last item)
                    const int r = i / (F_PW / 4), gq = i - r * (F_PW / 4);
                    const int cy = cy0 + r, sy = cy - a.top, cx = cx0 + 4 * gq;
                    const bool row_canvas = !MASKED || (cy >= 0 && cy < a.in_h), row_img = !MASKED || (row_canvas && sy >= 0 && sy < a.new_h);
                    // offset of canvas column cx inside the raw row: (first needed byte of the row) mod 16 + 12 bytes per group
                    const int o = (row_img ? (sy * pm + col0) & 15 : 0) + 12 * gq;
                    const unsigned *rp = (const unsigned *)(raw + r * F_RAWROW + (o & ~3));
                    const unsigned q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
                    const unsigned sh = (unsigned)(o & 3);
                    const unsigned d0 = __builtin_amdgcn_alignbyte(q1, q0, sh), d1 = __builtin_amdgcn_alignbyte(q2, q1, sh), d2 = __builtin_amdgcn_alignbyte(q3, q2, sh);
                    const unsigned px[4] = {d0, __builtin_amdgcn_alignbyte(d1, d0, 3), __builtin_amdgcn_alignbyte(d2, d1, 2), d2 >> 8};      // B | G << 8 | R << 16 (| junk << 24)
                    half4 h[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int x = cx + k;
                        const bool in_img = !MASKED || (row_img && x >= a.left && x < a.left + a.new_w), in_canvas = !MASKED || (row_canvas && x >= 0 && x < a.in_w);
                        const unsigned v = in_img ? px[k] : 0x727272u;
                        const float sc = in_canvas ? 1.0f / 255.0f : 0.0f;
                        h[k] = half4{(f16)((float)((v >> 16) & 255u) * sc), (f16)((float)((v >> 8) & 255u) * sc), (f16)((float)(v & 255u) * sc), (f16)0.f};
                    }
                    unsigned char *dst = pix + r * F_PIXROW + gq * 32;
                    *(half8 *)dst = half8{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
                    *(half8 *)(dst + 16) = half8{h[2][0], h[2][1], h[2][2], h[2][3], h[3][0], h[3][1], h[3][2], h[3][3]};
                }
            };
            if (interior) convert(std::false_type{}); else convert(std::true_type{});
            FST(2);
            __syncthreads();                               // the raw rows are free: the next tile's bytes travel under the rest of this tile
            if (t + (int)gridDim.x < a.n_tiles) issue_src(t + gridDim.x);
        }
        FST(3);

        // ---- T3: stem.  36 groups of 16 positions: stem row g's even columns 0 .. 30 (E g, 17 groups), its odd columns (O g, 17), and column 32 of
        // all rows (X0: rows 0 .. 15, X1: row 16).  Wave w: E g for g = w mod 4, O g for g = 3 - w mod 4, X0 -> wave 1, X1 -> wave 2: nine each.
        // Row / column -1 of the stem's output (first tiles only) is layer 1's zero padding. ----
        {
            const bool top = y0 == 0, lft = x0 == 0 && p == 0;
            const int eb = wave, ob = 3 - wave;            // first even / odd row of this wave
            // the ninth group: E 16 (wave 0), O 16 (wave 3), X0 (wave 1: column 32 of stem rows p), X1 (wave 2: column 32 of row 16; lanes p > 0 repeat it)
            int src9, src29, dst9;
            if (wave == 0) { src9 = st_src + 16 * 2 * F_PIXROW; src29 = st_src2 + 16 * 2 * F_PIXROW; dst9 = SE_OFF + st_dst + 16 * F_EP * 64; }
            else if (wave == 3) { src9 = st_src + 16 + 16 * 2 * F_PIXROW; src29 = st_src2 + 16 + 16 * 2 * F_PIXROW; dst9 = SO_OFF + st_dst + 16 * F_OP * 64; }
            else {
                const int row = wave == 1 ? p : 16;
                src9 = PIX_OFF + (64 + 2 * (q & 1)) * 8 + (2 * row + (q >> 1)) * F_PIXROW;
                src29 = PIX_OFF + (64 + 2 * (q & 1)) * 8 + (2 * row + 2) * F_PIXROW;
                dst9 = SE_OFF + (row * F_EP + 16) * 64 + ((q >> 1) << 4) + (q & 1) * 8;      // row pitch * row + 16: bit 2 of the row index is 0 -> no swizzle term
            }
            const bool z9 = (wave == 0 && lft) || (wave == 1 && top && p == 0);
            // Software pipeline over batches of two groups (E k = 0, 1 | E 2, 3 | O 0, 1 | O 2, 3 | the ninth): the MFMAs of batch b + 1 are issued BETWEEN the
            // SiLU instructions of batch b (one MFMA, then a slice of VALU work: sched_group_barrier), so the matrix pipe runs in the shadow of the
            // epilogue instead of in front of it -- stamped: the phase was its MFMA time + its VALU time, the waves of a SIMD taking turns.
            half8 a0[2][2], a1[2][2];
            floatx4 acc[2][2][2];
            auto gsrc = [&](int b, int j, int &s1, int &s2, int &dd, bool &zz) {      // group j of batch b (b < 4)
                const bool odd = b >= 2;
                const int k = 2 * (b & 1) + j, row = (odd ? ob : eb) + 4 * k;
                s1 = st_src + (odd ? 16 : 0) + row * 2 * F_PIXROW; s2 = st_src2 + (odd ? 16 : 0) + row * 2 * F_PIXROW;
                dd = (odd ? SO_OFF + row * F_OP * 64 : SE_OFF + row * F_EP * 64) + st_dst;
                zz = (top && row == 0) || (!odd && lft);
            };
            int s1, s2, dd[2][2]; bool zz[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { gsrc(0, j, s1, s2, dd[0][j], zz[0][j]); st_load(s1, s2, a0[0][j], a1[0][j]); }
#pragma unroll
            for (int j = 0; j < 2; ++j) st_mma(a0[0][j], a1[0][j], acc[0][j]);
#pragma unroll
            for (int b = 1; b < 4; ++b) {
                const int cur = b & 1, prev = cur ^ 1;
#pragma unroll
                for (int j = 0; j < 2; ++j) { gsrc(b, j, s1, s2, dd[cur][j], zz[cur][j]); st_load(s1, s2, a0[cur][j], a1[cur][j]); }
#pragma unroll
                for (int j = 0; j < 2; ++j) st_mma(a0[cur][j], a1[cur][j], acc[cur][j]);
#pragma unroll
                for (int j = 0; j < 2; ++j) st_store(acc[prev][j], dd[prev][j], zz[prev][j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {                  // 8 MFMAs of this batch among the ~80 VALU / transcendental instructions of the previous one
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x402, 9, 0);
                }
            }
            st_load(src9, src29, a0[0][0], a1[0][0]);
            st_mma(a0[0][0], a1[0][0], acc[0][0]);
#pragma unroll
            for (int j = 0; j < 2; ++j) st_store(acc[1][j], dd[1][j], zz[1][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x402, 18, 1);
            }
            st_store(acc[0][0], dst9, z9);
        }
        FST(4);
        __syncthreads();                                   // stem planes complete; the pixel tile is dead
        FST(5);

        // ---- T4: layer 1.  Output row 4 ph + i, tap (kh, kw) reads stem row 2 (4 ph + i) + kh, columns 2 p + kw ----
        floatx4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[i][u] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            half8 fr[3];
            fr[0] = *(const half8 *)(lds + l1_e0 + s * F_EP * 64);
            fr[1] = *(const half8 *)(lds + l1_o + s * F_OP * 64);
            fr[2] = *(const half8 *)(lds + l1_e2 + s * F_EP * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kh = s - 2 * i;
                if (kh < 0 || kh > 2) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[u][kh * 3 + kw], fr[kw], acc[i][u], 0, 0, 0);
            }
        }
        FST(6);
        // (an accumulator meets its taps as (kh, kw) = (0,0) (0,1) (0,2) (1,0) ... -- the k order of the launch this replaces)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                floatx4 v = acc[i][u] + b1v[u];
                silu4(v);
                *(half4 *)(lds + Y1_OFF + y_wr[u] + i * 2048) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            }
        FST(7);
        __syncthreads();                                   // layer 1's tile complete (both cout halves); the stem planes are dead
        FST(8);

        // ---- T5: 2.cv1 (1x1) on the tile; weight fragments and bias from LDS ----
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[i][u] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            half8 fa[4], fw[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) fw[u] = *(const half8 *)(lds + W2_OFF + ((2 * ch + u) * 2 + kc) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *(const half8 *)(lds + Y1_OFF + y_rd[kc] + i * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[u], fa[i], acc[i][u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const floatx4 b2v = *(const floatx4 *)(lds + B2_OFF + ((2 * ch + u) * 4 + q) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                floatx4 v = acc[i][u] + b2v;
                silu4(v);
                *(half4 *)(lds + Y2_OFF + y_wr[u] + i * 2048) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            }
        }
        FST(9);
        __syncthreads();                                   // 2.cv1's tile complete; nobody reads layer 1's tile (= the pixel tile's LDS) any more
        if constexpr (TENSOR) {                            // tensor source: the pixel tile doubled as layer 1's tile, so its next load starts only here
            if (t + (int)gridDim.x < a.n_tiles) issue_src(t + gridDim.x);
        }

        // ---- T6: 16-byte NHWC stores (8 chunks = one 128-byte line per pixel) ----
        f16 *const orow = a.out + ((size_t)(b * a.out_Hp + y0 + a.out_pad) * a.out_Wp + x0 + a.out_pad) * a.out_cs + st_go;
        const int row2 = 2 * a.out_Wp * a.out_cs;          // thread's pixels: rows 2 k + (tid >> 7) of the tile
#pragma unroll
        for (int k = 0; k < (F_TH * F_TW * 8) / 256; ++k) *(half8 *)(orow + k * row2) = *(const half8 *)(lds + st_rd + k * 4096);
        FST(10);
    }
}

}  // namespace

bool front_supported(int c0, int c1, int c2, int in_h, int in_w) {
    return c0 == 32 && c1 == 64 && c2 == 64 && in_h % (4 * F_TH) == 0 && in_w % (4 * F_TW) == 0;
}

int launch_front(const FrontLaunch &l, hipStream_t s) {
    RT_CHECK(front_supported(l.c0, l.c1, l.c2, l.in_h, l.in_w), RTMODT_E_UNSUPPORTED, "launch_front: channels %d / %d / %d at %dx%d", l.c0, l.c1, l.c2, l.in_w, l.in_h);
    RT_CHECK(l.w0 && l.w1 && l.w2 && l.b0 && l.b1 && l.b2 && l.zeros && l.out.base, RTMODT_E_INVALID, "launch_front: null operand");
    RT_CHECK(l.kp1 >= 9 * l.c0 && l.kp2 >= l.c1 && l.kp1 % 8 == 0 && l.kp2 % 8 == 0, RTMODT_E_INVALID, "launch_front: weight row strides %d / %d", l.kp1, l.kp2);
    const TensorView &o = l.out;
    RT_CHECK(o.H == l.in_h / 4 && o.W == l.in_w / 4 && o.c == l.c2 && o.coff % 8 == 0 && o.C % 8 == 0, RTMODT_E_INVALID, "launch_front: output view");
    RT_CHECK((uintptr_t)o.base >= (1ull << 32), RTMODT_E_INVALID, "launch_front: view base %p is not a device address", (void *)o.base);
    RT_CHECK(l.B >= 1 && (long)l.B * (o.H + 2 * o.pad) * o.padded_w() * o.C < (1L << 31), RTMODT_E_INVALID, "launch_front: batch %d", l.B);
    FrontArgs a{};
    a.zeros = l.zeros;
    a.w0 = l.w0; a.w1 = l.w1; a.w2 = l.w2; a.b0 = l.b0; a.b1 = l.b1; a.b2 = l.b2; a.kp1 = l.kp1; a.kp2 = l.kp2;
    a.out = o.base + o.coff; a.out_Hp = o.H + 2 * o.pad; a.out_Wp = o.padded_w(); a.out_cs = o.C; a.out_pad = o.pad;
    a.in_h = l.in_h; a.in_w = l.in_w;
    a.tiles_x = o.W / F_TW; a.tiles_y = o.H / F_TH; a.n_tiles = l.B * a.tiles_x * a.tiles_y;
    // two workgroups per CU (LDS), persistent over the tiles.  Diagnostic builds: RTMODT_FRONT_WGS=1 launches ONE per CU, leaving half of every CU's LDS and
    // registers to the launches of the engine's other stages (same-box A/B: profiles/r05/README.md)
    static const int per_cu = rt_diag("FRONT_WGS") ? std::max(1, atoi(rt_diag("FRONT_WGS"))) : 2;
    const int G = std::min(a.n_tiles, per_cu * device_cus());
    if (l.from_tensor) {
        const TensorView &im = l.img4;
        RT_CHECK(im.base && im.C == 4 && im.pad == 1 && im.H == l.in_h && im.W == l.in_w && (uintptr_t)im.base % 16 == 0, RTMODT_E_INVALID, "launch_front: image tensor must be 4-channel with a border");
        a.img = im.base; a.img_Hp = im.H + 2; a.img_Wp = im.W + 2;
        hipLaunchKernelGGL((front_fused<true>), dim3(G), dim3(256), 0, s, a);
    } else {
        const LetterboxGeom &g = l.g;
        RT_CHECK(!g.resize, RTMODT_E_INVALID, "launch_front: frames that need a resize go through the letterbox kernel (from_tensor)");
        RT_CHECK(l.frame0 >= 0 && l.frame0 + l.B <= 64, RTMODT_E_INVALID, "launch_front: frames %d..%d", l.frame0, l.frame0 + l.B);
        RT_CHECK((long)g.src_h * l.pitch < (1L << 31) && g.new_h >= 1 && g.new_w >= 1, RTMODT_E_INVALID, "launch_front: frame geometry");
        // the kernel reads source pixel (y - top, x - left) for every canvas pixel inside [top, top + new_h) x [left, left + new_w): that window must be the frame, inside the canvas
        RT_CHECK(g.new_h == g.src_h && g.new_w == g.src_w && g.top >= 0 && g.left >= 0 && g.top + g.new_h <= l.in_h && g.left + g.new_w <= l.in_w && l.pitch >= 3 * g.src_w,
                 RTMODT_E_INVALID, "launch_front: a %dx%d frame (pitch %d) at (%d, %d) does not lie inside the %dx%d canvas unscaled", g.src_w, g.src_h, l.pitch, g.left, g.top, l.in_w, l.in_h);
        for (int i = 0; i < l.B; ++i)
            RT_CHECK((uintptr_t)l.frames.p[l.frame0 + i] >= (1ull << 32), RTMODT_E_INVALID, "launch_front: frame %d (%p) is not a device address", l.frame0 + i, (const void *)l.frames.p[l.frame0 + i]);
        a.frames = l.frames; a.frame0 = l.frame0; a.pitch = l.pitch; a.top = g.top; a.left = g.left; a.new_h = g.new_h; a.new_w = g.new_w;
        a.frame_bytes = (g.src_h - 1) * l.pitch + 3 * g.src_w;
        hipLaunchKernelGGL((front_fused<false>), dim3(G), dim3(256), 0, s, a);
    }
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

}  // namespace rtmodt
