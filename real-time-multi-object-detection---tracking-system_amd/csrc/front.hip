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
#include "conv_dev.h"

namespace rtmodt {

namespace {

constexpr int F_TH = 8, F_TW = 16;                      // layer-1 output tile
constexpr int F_SH = 2 * F_TH + 1, F_SW = 2 * F_TW + 1; // 17 x 33 stem positions feed it
constexpr int F_PH = 2 * F_SH + 1, F_PW = 2 * F_SW + 2; // 35 rows x 68 columns of source pixels (the last column only meets the stem's zero 4th tap)
constexpr int F_PIXROW = F_PW * 8;                      // bytes per pixel row (RGB0 fp16)
constexpr int F_RAWCH = 14, F_RAWROW = F_RAWCH * 16;    // raw bytes per row: 3 * 68 = 204 + up to 15 of misalignment -> 14 chunks
constexpr int F_NE = (F_SW + 1) / 2, F_NO = F_SW / 2;   // even / odd stem columns of a row: 17 / 16
constexpr int F_EVEN = F_SH * F_NE, F_ODD = F_SH * F_NO;                       // 289 / 272 positions
constexpr int F_EG = (F_EVEN + 15) / 16, F_OG = (F_ODD + 15) / 16;             // 19 / 17 groups of 16
constexpr int RAW_OFF = 0, RAW_BYTES = 8 * 1024;                               // 35 x 14 = 490 chunks -> 8 wave instructions
constexpr int PIX_OFF = RAW_OFF + RAW_BYTES, PIX_BYTES = 19 * 1024;            // 35 x 544 = 19 040 -> 19 wave instructions (tensor source)
constexpr int SE_OFF = PIX_OFF + PIX_BYTES, SO_OFF = SE_OFF + F_EG * 16 * 64;  // stem planes, 64 B per position
constexpr int F_LDS = SO_OFF + F_OG * 16 * 64;
constexpr int Y1_OFF = PIX_OFF, Y2_OFF = SE_OFF;                               // layer 1's tile reuses the pixel tile, 2.cv1's the stem planes
static_assert(F_PH * F_RAWROW <= RAW_BYTES && F_PH * F_PIXROW <= PIX_BYTES && F_TH * F_TW * 128 <= PIX_BYTES && F_TH * F_TW * 128 <= F_EG * 16 * 64, "LDS map");
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

template <bool TENSOR>
__global__ __launch_bounds__(256, 2) void front_fused(FrontArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[F_LDS];
    unsigned char *const raw = lds + RAW_OFF, *const pix = lds + PIX_OFF, *const sE = lds + SE_OFF, *const sO = lds + SO_OFF;
    unsigned char *const y1 = lds + Y1_OFF, *const y2 = lds + Y2_OFF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int ch = wave & 1, ph = wave >> 1;               // layer 1 / 2.cv1: this wave's cout half (32 couts) and pixel half (4 rows)

    // ---- weights and biases of the three convs -> registers, once per workgroup ----
    half8 w0f[2][2], w1f[2][9], w2f[2][2];
    floatx4 b0v[2], b1v[2], b2v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) w0f[u][kk] = *(const half8 *)(a.w0 + (u * 16 + p) * 64 + kk * 32 + q * 8);
        b0v[u] = *(const floatx4 *)(a.b0 + u * 16 + q * 4);
        const int n = (2 * ch + u) * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) w1f[u][tap] = *(const half8 *)(a.w1 + (size_t)(n + p) * a.kp1 + tap * 32 + q * 8);
        b1v[u] = *(const floatx4 *)(a.b1 + n + q * 4);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) w2f[u][kc] = *(const half8 *)(a.w2 + (size_t)(n + p) * a.kp2 + kc * 32 + q * 8);
        b2v[u] = *(const floatx4 *)(a.b2 + n + q * 4);
    }

    const int per_img = a.tiles_x * a.tiles_y;
    auto decode = [&](int t, int &b, int &y0, int &x0) {
        b = t / per_img;
        const int r = t - b * per_img, ty = r / a.tiles_x;
        y0 = ty * F_TH; x0 = (r - ty * a.tiles_x) * F_TW;
    };
    // byte source: where row r of tile (b, y0, x0) starts inside its frame (relative to the frame's 16-byte aligned base `flo`), floor-aligned to 16
    auto raw_base = [&](const uint8_t *f, int sy, int cx0) -> long {
        const long row0 = (long)((uintptr_t)f & 15) + (long)sy * a.pitch;
        return (row0 + 3L * (cx0 - a.left)) & ~15L;
    };
    auto issue_src = [&](int t) {
        int b, y0, x0;
        decode(t, b, y0, x0);
        const int cy0 = 4 * y0 - 3, cx0 = 4 * x0 - 3;
        if constexpr (TENSOR) {
            // pixel rows straight from the image tensor: chunk j of row r = canvas pixels (cx0 + 2 j, + 1) of canvas row cy0 + r; what lies outside the
            // bordered tensor (rows / columns -3, -2 of the first tiles) comes from the zero page
#pragma unroll 1
            for (int k = wave; k < 19; k += 4) {
                const int g = k * 64 + lane, r = min(g / 34, F_PH - 1), j = g - (g / 34) * 34;
                const int ty = cy0 + r + 1, tx = cx0 + 2 * j + 1;              // padded tensor coordinates
                const bool ok = ty >= 0 && tx >= 0;
                const f16 *src = ok ? a.img + ((size_t)(b * a.img_Hp + ty) * a.img_Wp + tx) * 4 : a.zeros;
                glds16(src, pix + k * 1024);
            }
        } else {
            const uint8_t *f = a.frames.p[a.frame0 + b];
            const uint8_t *flo = f - ((uintptr_t)f & 15);
            const long fspan = (long)(((uintptr_t)f & 15) + a.frame_bytes + 15) & ~15L;
#pragma unroll 1
            for (int k = wave; k < 8; k += 4) {
                const int g = k * 64 + lane, r = min(g / F_RAWCH, F_PH - 1), j = g - (g / F_RAWCH) * F_RAWCH;
                const int sy = min(max(cy0 + r - a.top, 0), a.new_h - 1);      // rows outside the image: any row of the frame (never read)
                long off = raw_base(f, sy, cx0) + 16 * j;
                off = off < 0 ? 0 : (off > fspan - 16 ? fspan - 16 : off);     // chunks that hold no byte of the frame: clamped into its pages (never read)
                glds16((const f16 *)(flo + off), raw + k * 1024);
            }
        }
    };

    int t = blockIdx.x;
    if (t >= a.n_tiles) return;
    issue_src(t);

    for (; t < a.n_tiles; t += gridDim.x) {
        int b, y0, x0;
        decode(t, b, y0, x0);
        const int cy0 = 4 * y0 - 3, cx0 = 4 * x0 - 3;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // T0: this tile's source has landed; every wave has left the previous tile
        if constexpr (!TENSOR) {
            // ---- T1: bytes -> (R, G, B, 0) fp16; byte -> float -> * (1 / 255) -> half == the letterbox kernel's half(c / 255.f) for all 256 values ----
            const uint8_t *f = a.frames.p[a.frame0 + b];
            for (int i = tid; i < F_PH * (F_PW / 4); i += 256) {
                const int r = i / (F_PW / 4), gq = i - r * (F_PW / 4);
                const int cy = cy0 + r, sy = cy - a.top, cx = cx0 + 4 * gq;
                const bool row_canvas = cy >= 0 && cy < a.in_h, row_img = row_canvas && sy >= 0 && sy < a.new_h;
                unsigned d0, d1, d2;
                bool canvas[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) canvas[k] = row_canvas && cx + k >= 0 && cx + k < a.in_w;
                const unsigned char *rr = raw + r * F_RAWROW;
                const int o = row_img ? (int)((long)((uintptr_t)f & 15) + (long)sy * a.pitch + 3L * (cx - a.left) - raw_base(f, sy, cx0)) : 0;
                if (row_img && cx >= a.left && cx + 3 < a.left + a.new_w) {
                    const unsigned *rp = (const unsigned *)(rr + (o & ~3));
                    const unsigned q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
                    const unsigned sh = (unsigned)(o & 3);
                    d0 = __builtin_amdgcn_alignbyte(q1, q0, sh);
                    d1 = __builtin_amdgcn_alignbyte(q2, q1, sh);
                    d2 = __builtin_amdgcn_alignbyte(q3, q2, sh);
                } else {                                   // the group straddles an edge of the image or lies outside it: byte by byte
                    unsigned dd[3] = {0u, 0u, 0u};
#pragma unroll
                    for (int j = 0; j < 12; ++j) {
                        const int x = cx + j / 3;
                        const bool in_img = row_img && x >= a.left && x < a.left + a.new_w;
                        const unsigned v = in_img ? rr[o + j] : 114u;
                        dd[j >> 2] |= v << (8 * (j & 3));
                    }
                    d0 = dd[0]; d1 = dd[1]; d2 = dd[2];
                }
                auto cv = [&](unsigned byte, bool in) -> f16 { return (f16)((float)byte * (in ? 1.0f / 255.0f : 0.0f)); };
                half8 lo, hi;
                lo[0] = cv((d0 >> 16) & 255u, canvas[0]); lo[1] = cv((d0 >> 8) & 255u, canvas[0]); lo[2] = cv(d0 & 255u, canvas[0]); lo[3] = (f16)0.f;
                lo[4] = cv((d1 >> 8) & 255u, canvas[1]); lo[5] = cv(d1 & 255u, canvas[1]); lo[6] = cv(d0 >> 24, canvas[1]); lo[7] = (f16)0.f;
                hi[0] = cv(d2 & 255u, canvas[2]); hi[1] = cv(d1 >> 24, canvas[2]); hi[2] = cv((d1 >> 16) & 255u, canvas[2]); hi[3] = (f16)0.f;
                hi[4] = cv(d2 >> 24, canvas[3]); hi[5] = cv((d2 >> 16) & 255u, canvas[3]); hi[6] = cv((d2 >> 8) & 255u, canvas[3]); hi[7] = (f16)0.f;
                *(half8 *)(pix + r * F_PIXROW + gq * 32) = lo;
                *(half8 *)(pix + r * F_PIXROW + gq * 32 + 16) = hi;
            }
            __syncthreads();                               // the raw rows are free: the next tile's bytes travel under the rest of this tile
            if (t + (int)gridDim.x < a.n_tiles) issue_src(t + gridDim.x);
        }

        // ---- T3: stem over the two planes of positions ----
        for (int gi = wave; gi < F_EG + F_OG; gi += 4) {
            const bool odd = gi >= F_EG;
            const int ncol = odd ? F_NO : F_NE;
            const int idx = min((odd ? gi - F_EG : gi) * 16 + p, (odd ? F_ODD : F_EVEN) - 1);      // the last even group: rows past the plane re-read its last position
            const int syr = idx / ncol, sxr = 2 * (idx - syr * ncol) + (odd ? 1 : 0);
            const unsigned char *src = pix + (2 * syr) * F_PIXROW + (2 * sxr + 2 * (q & 1)) * 8;
            const half8 a0 = *(const half8 *)(src + (q >> 1) * F_PIXROW);      // kernel rows 0 / 1
            const half8 a1 = *(const half8 *)(src + 2 * F_PIXROW);             // kernel row 2 (k' >= 48 meets zero weights)
            const bool inside = 2 * y0 - 1 + syr >= 0 && 2 * x0 - 1 + sxr >= 0;   // row / column -1 of the stem's output = layer 1's zero padding
            unsigned char *dst = (odd ? sO : sE);
            const int R = (odd ? gi - F_EG : gi) * 16 + p;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                floatx4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0f[u][0], a0, b0v[u], 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0f[u][1], a1, acc, 0, 0, 0);
                silu4(acc);
                half4 h = {(f16)acc[0], (f16)acc[1], (f16)acc[2], (f16)acc[3]};
                if (!inside) h = half4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                *(half4 *)(dst + s_off(R, u * 2 + (q >> 1)) + (q & 1) * 8) = h;
            }
        }
        __syncthreads();                                   // stem planes complete; the pixel tile is dead

        // ---- T4: layer 1.  Output row 4 ph + i, tap (kh, kw) reads stem row 2 (4 ph + i) + kh, columns 2 p + kw ----
        floatx4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[i][u] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int syr = 8 * ph + s;
            half8 fr[3];
            fr[0] = *(const half8 *)(sE + s_off(syr * F_NE + p, q));
            fr[1] = *(const half8 *)(sO + s_off(syr * F_NO + p, q));
            fr[2] = *(const half8 *)(sE + s_off(syr * F_NE + p + 1, q));
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
        // NOTE on order: an accumulator meets its taps as (kh, kw) = (0,0) (0,1) (0,2) (1,0) ... -- the k order of the launch this replaces
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = (4 * ph + i) * 16 + p;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                floatx4 v = acc[i][u] + b1v[u];
                silu4(v);
                *(half4 *)(y1 + y_off(px, (2 * ch + u) * 2 + (q >> 1)) + (q & 1) * 8) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            }
        }
        __syncthreads();                                   // layer 1's tile complete (both cout halves); the stem planes are dead

        // ---- T5: 2.cv1 (1x1) on the tile ----
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[i][u] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            half8 fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *(const half8 *)(y1 + y_off((4 * ph + i) * 16 + p, kc * 4 + q));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[u][kc], fa[i], acc[i][u], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = (4 * ph + i) * 16 + p;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                floatx4 v = acc[i][u] + b2v[u];
                silu4(v);
                *(half4 *)(y2 + y_off(px, (2 * ch + u) * 2 + (q >> 1)) + (q & 1) * 8) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            }
        }
        __syncthreads();                                   // 2.cv1's tile complete; nobody reads layer 1's tile (= the pixel tile's LDS) any more
        if constexpr (TENSOR) {                            // tensor source: the pixel tile doubled as layer 1's tile, so its next load starts only here
            if (t + (int)gridDim.x < a.n_tiles) issue_src(t + gridDim.x);
        }

        // ---- T6: 16-byte NHWC stores (8 chunks = one 128-byte line per pixel) ----
        f16 *const orow = a.out + ((size_t)(b * a.out_Hp + y0 + a.out_pad) * a.out_Wp + x0 + a.out_pad) * a.out_cs;
#pragma unroll
        for (int k = 0; k < (F_TH * F_TW * 8) / 256; ++k) {
            const int e = k * 256 + tid, px = e >> 3, c = e & 7;
            const half8 v = *(const half8 *)(y2 + y_off(px, c));
            *(half8 *)(orow + ((size_t)(px >> 4) * a.out_Wp + (px & 15)) * a.out_cs + c * 8) = v;
        }
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
    const int G = std::min(a.n_tiles, 2 * device_cus());   // two workgroups per CU (LDS), persistent over the tiles
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
        a.frames = l.frames; a.frame0 = l.frame0; a.pitch = l.pitch; a.top = g.top; a.left = g.left; a.new_h = g.new_h; a.new_w = g.new_w;
        a.frame_bytes = (g.src_h - 1) * l.pitch + 3 * g.src_w;
        hipLaunchKernelGGL((front_fused<false>), dim3(G), dim3(256), 0, s, a);
    }
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

}  // namespace rtmodt
