// stem_l1.hip -- the first THREE convolutions of YOLOv8s in one launch:
//     layer 0  Conv(3 -> 32, 3x3, s2) + SiLU      (the stem; letterbox / BGR->RGB / /255 / .half() folded in as in conv.hip:stem_fused)
//     layer 1  Conv(32 -> 64, 3x3, s2) + SiLU
//     2.cv1    Conv(64 -> 64, 1x1) + SiLU         (optional tail, as conv.hip:epilogue_tail)
// What it replaces: the first three fused convs ultralytics runs for `YOLO.predict` at
// /root/reference/src/detection/detector.py:100-111.
//
// Why: at 16 frames the stem WRITES 105 MB of 320x320x32 activations that layer 1 reads straight back -- 210 MB of the step's
// traffic for 4 % of its FLOPs ("cut bytes, not flops").  Here the stem's output never leaves the CU.
//
// One persistent 512-thread workgroup (8 wave64s, two workgroups per CU) walks over SEGMENTS = (image, layer-1 output row,
// 80-pixel half of the row):
//   1. the 7 x 323 input pixels the segment depends on become fp16 (R, G, B, 0) pixels in LDS -- from the BGR bytes of the
//      frame (114 outside the image, 0 outside the canvas) or from the letterboxed image tensor;
//   2. stem: 3 rows x 161 pixels of layer 0 on the matrix cores (K re-indexed kh*16 + kw*4 + c, two 32-deep steps, weights
//      in registers), SiLU, fp16 -> LDS, ZERO where the position lies outside the 320x320 map (that is layer 1's padding);
//   3. layer 1: 80 pixels x 64 channels, nine taps read from that LDS image (stride-2 pixel fragments, 80-byte pixel pitch:
//      2-way bank conflicts at worst), each wave owns one 16-channel tile whose nine weight fragments live in REGISTERS;
//   4. the 1x1 tail on the fp16 tile in LDS (weights in registers as well);
//   5. 16-byte NHWC stores of the only tensor that leaves: 2.cv1's output (or layer 1's when there is no tail).
// All weights are register-resident for the lifetime of the workgroup (60 VGPRs), so a segment moves 7 KB of pixels in and
// 10 KB out -- nothing else.  fp32 accumulate, one rounding per layer output, exactly like the separate launches.
#include <algorithm>

#include "kernels.h"

namespace rtmodt {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_s(float x) { return silu(x); }

constexpr int SL_THREADS = 512, SL_WAVES = 8;
constexpr int SL_TW = 80;                       // layer-1 output pixels per segment
constexpr int SL_NSX = 2 * SL_TW + 1;           // stem pixels per row a segment needs (161)
constexpr int SL_NIX = 4 * SL_TW + 3;           // image pixels per row (323)
constexpr int SL_PIXW = 324;                    // half4 pixels per LDS image row (even: pixel pairs stay 16-byte aligned)
constexpr int SL_SROW = 162;                    // stem pixels per LDS row
constexpr int SL_SPITCH = 80;                   // bytes per stem pixel in LDS (64 + 16: stride-2 fragment reads 2-way at worst)
constexpr int SL_OPITCH = 144;                  // bytes per layer-1 pixel in LDS (128 + 16: conflict-free fragment reads)
constexpr int SL_PIX_BYTES = 7 * SL_PIXW * 8;   // 18144
constexpr int SL_STEM_BYTES = 3 * SL_SROW * SL_SPITCH;   // 38880
constexpr int SL_LDS = SL_PIX_BYTES + SL_STEM_BYTES;
static_assert(SL_TW * SL_OPITCH <= SL_PIX_BYTES && SL_TW * SL_OPITCH <= SL_STEM_BYTES, "the output tiles reuse the dead input buffers");

struct StemL1Args {
    FramePtrs frames;                        // bytes mode (img == nullptr)
    int frame0, pitch, top, left, new_h, new_w;
    const f16 *img;                          // tensor mode: [B][in_h + 2][in_w + 2][4] fp16 RGB0, zero border
    int in_h, in_w;                          // canvas (network input)
    int H0, W0;                              // stem output map
    int H1, W1;                              // layer-1 output map
    const f16 *w0; const float *b0;          // stem: [32][64] (k' = kh*16 + kw*4 + c), bias[32]
    const f16 *w1; const float *b1; int kp1; // layer 1: [64 -> 128 rows][kp1 >= 288], K order (kh, kw, cin)
    const f16 *wt; const float *bt; int kpt, t_cout, t_act;   // optional 1x1 tail [t_cout -> 128 rows][kpt >= 64]
    f16 *out; int out_Hp, out_Wp, out_cs, out_pad, out_c;      // the one tensor that is stored (view: base + channel offset applied)
    int B, segs_per_row, n_seg;
};

__global__ __launch_bounds__(SL_THREADS, 4) void stem_l1_fused(StemL1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    half4 *pix = (half4 *)lds;                                   // [7][SL_PIXW]
    unsigned char *stemo = lds + SL_PIX_BYTES;                   // [3][SL_SROW][SL_SPITCH]
    unsigned char *l1o = lds;                                    // [SL_TW][SL_OPITCH]  (the image pixels are dead by then)
    unsigned char *to = stemo;                                   // [SL_TW][SL_OPITCH]  (the stem rows are dead by then)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;

    // ---- layer 1's weights -> registers, once per workgroup (the stem's and the tail's, 24 VGPRs, are re-read from L2 per
    // segment right where they are used: keeping them too does not fit 128 VGPRs, i.e. two workgroups per CU) ----
    const int u1 = wave & 3, gs = wave >> 2;                     // this wave's 16-channel tile of layer 1 / of the tail; its pixel groups gs, gs + 2, gs + 4
    half8 wf1[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wf1[t] = *(const half8 *)(a.w1 + (long)(u1 * 16 + p) * a.kp1 + t * 32 + q * 8);
    const floatx4 bv1 = *(const floatx4 *)(a.b1 + u1 * 16 + q * 4);
    const bool tail = a.wt != nullptr;
    const bool tail_mine = tail && u1 * 16 < a.t_cout;

    // The 7 x 324 input pixels of a segment: 5 per thread, FETCHED into registers one segment ahead (the loads fly under the
    // previous segment's three matrix phases) and committed to LDS, converted, at the top of their own segment.
    constexpr int NPRE = (7 * SL_PIXW + SL_THREADS - 1) / SL_THREADS;
    uint2 pre[NPRE];
    auto fetch = [&](int seg) {
        const int xh = seg % a.segs_per_row, row = seg / a.segs_per_row;
        const int b = row / a.H1, oy = row - b * a.H1;
        const int iy0 = 4 * oy - 3, ix0 = 4 * (xh * SL_TW) - 3;   // image pixel of pix[0][0]
        const uint8_t *f = a.img ? nullptr : a.frames.p[a.frame0 + b];
        const int Wp = a.in_w + 2;
        const long ibase = (long)b * (a.in_h + 2) * Wp;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = tid + k * SL_THREADS;
            const int r = i / SL_PIXW, c = i - r * SL_PIXW;
            const int iy = iy0 + r, ix = ix0 + c;
            uint2 v = {0u, 0u};                                   // .y: 0 = the pixel is .x/.y's raw fp16 bits (tensor mode) or zero; 1 = 114 pad; 2 = BGR bytes in .x
            if (i < 7 * SL_PIXW && c < SL_NIX) {
                if (a.img) {
                    if (iy >= -1 && iy <= a.in_h && ix >= -1 && ix <= a.in_w) v = *(const uint2 *)(a.img + (ibase + (long)(iy + 1) * Wp + ix + 1) * 4);
                } else if (iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) {
                    const int sy = iy - a.top, sx = ix - a.left;
                    if (sy >= 0 && sy < a.new_h && sx >= 0 && sx < a.new_w) {
                        const uint8_t *s = f + (long)sy * a.pitch + sx * 3;
                        v = uint2{(unsigned)s[0] | ((unsigned)s[1] << 8) | ((unsigned)s[2] << 16), 2u};
                    } else {
                        v = uint2{0u, 1u};
                    }
                }
            }
            pre[k] = v;
        }
    };
    auto commit = [&]() {
        const float k255 = 1.0f / 255.0f;                          // (half)(byte * (1/255)) == (half)(byte / 255.f) for all 256 bytes (tests/test_oracle_yolo.py)
        const f16 pad = (f16)(114.0f * k255);
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = tid + k * SL_THREADS;
            if (i >= 7 * SL_PIXW) continue;
            uint2 v = pre[k];
            if (!a.img) {
                half4 h = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                if (v.y == 2u) h = half4{(f16)((float)((v.x >> 16) & 255u) * k255), (f16)((float)((v.x >> 8) & 255u) * k255), (f16)((float)(v.x & 255u) * k255), (f16)0.f};
                else if (v.y == 1u) h = half4{pad, pad, pad, (f16)0.f};
                pix[i] = h;
            } else {
                *(uint2 *)(pix + i) = v;
            }
        }
    };
    if ((int)blockIdx.x < a.n_seg) fetch(blockIdx.x);
    for (int seg = blockIdx.x; seg < a.n_seg; seg += gridDim.x) {
        const int xh = seg % a.segs_per_row, row = seg / a.segs_per_row;
        const int b = row / a.H1, oy = row - b * a.H1;
        const int ox0 = xh * SL_TW;
        const int sy0 = 2 * oy - 1, sx0 = 2 * ox0 - 1;            // stem pixel of stemo[0][0]

        // ---- 1. the segment's input pixels as fp16 (R, G, B, 0): pix[r][c] = image pixel (4 oy - 3 + r, 4 ox0 - 3 + c); column 323 exists only
        // as the zero-weight half of the last pixel pair and must be zero, never garbage ----
        commit();
        __syncthreads();
        if (seg + (int)gridDim.x < a.n_seg) fetch(seg + gridDim.x);      // next segment's pixels: in flight until the next commit

        // ---- 2. stem: 3 rows x 161 pixels, 16 pixels x 32 channels per item ----
        half8 wf0[2][2];
        floatx4 bv0[2];
        {
            const f16 *w0 = a.w0;
            asm volatile("" : "+v"(w0));                           // (a fresh address per segment: the loads stay inside the loop)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) wf0[u][kk] = *(const half8 *)(w0 + (u * 16 + p) * 64 + kk * 32 + q * 8);
                bv0[u] = *(const floatx4 *)(a.b0 + u * 16 + q * 4);
            }
        }
        for (int it = wave; it < 3 * 11; it += SL_WAVES) {
            const int sr = it / 11, g = it - sr * 11;
            const int ls = g * 16 + p, lsr = min(ls, SL_NSX - 1);
            const half4 *prow = pix + (2 * sr + (q >> 1)) * SL_PIXW + 2 * lsr + 2 * (q & 1);
            const half8 a0 = *(const half8 *)prow;                                        // kernel rows 0 / 1
            half8 a1 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
            if ((q >> 1) == 0) a1 = *(const half8 *)(pix + (2 * sr + 2) * SL_PIXW + 2 * lsr + 2 * (q & 1));   // kernel row 2; k' >= 48 meets zero weights
            const int sy = sy0 + sr, sx = sx0 + ls;
            const bool inside = sy >= 0 && sy < a.H0 && sx >= 0 && sx < a.W0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                floatx4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[u][0], a0, bv0[u], 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[u][1], a1, acc, 0, 0, 0);
                half4 h = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                if (inside) h = half4{(f16)silu_s(acc[0]), (f16)silu_s(acc[1]), (f16)silu_s(acc[2]), (f16)silu_s(acc[3])};
                if (ls < SL_NSX) *(half4 *)(stemo + (sr * SL_SROW + ls) * SL_SPITCH + (u * 16 + q * 4) * 2) = h;
            }
        }
        __syncthreads();

        // ---- 3. layer 1: this wave's 16 channels of pixel groups gs, gs + 2, gs + 4 ----
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {
            const int g = gs + 2 * gi;
            if (g < SL_TW / 16) {
                floatx4 acc1 = {0.f, 0.f, 0.f, 0.f};                  // bias after the sum, like conv.hip's epilogues (bit-identical to the separate launch)
                const unsigned char *base = stemo + (2 * (g * 16 + p)) * SL_SPITCH + q * 16;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int kh = t / 3, kw = t - kh * 3;
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[t], *(const half8 *)(base + (kh * SL_SROW + kw) * SL_SPITCH), acc1, 0, 0, 0);
                }
                acc1 = acc1 + bv1;
                *(half4 *)(l1o + (g * 16 + p) * SL_OPITCH + (u1 * 16 + q * 4) * 2) =
                    half4{(f16)silu_s(acc1[0]), (f16)silu_s(acc1[1]), (f16)silu_s(acc1[2]), (f16)silu_s(acc1[3])};
            }
        }
        __syncthreads();

        // ---- 4. the 1x1 tail on the tile ----
        const unsigned char *otile = l1o;
        if (tail) {
            if (tail_mine) {
                half8 wft[2];
                const f16 *wt = a.wt;
                asm volatile("" : "+v"(wt));
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) wft[kk] = *(const half8 *)(wt + (long)(u1 * 16 + p) * a.kpt + kk * 32 + q * 8);
                const floatx4 bvt = *(const floatx4 *)(a.bt + u1 * 16 + q * 4);
#pragma unroll
                for (int gi = 0; gi < 3; ++gi) {
                    const int g = gs + 2 * gi;
                    if (g < SL_TW / 16) {
                        floatx4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
                            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wft[kk], *(const half8 *)(l1o + (g * 16 + p) * SL_OPITCH + (kk * 4 + q) * 16), acc2, 0, 0, 0);
                        acc2 = acc2 + bvt;
                        if (a.t_act) { acc2[0] = silu_s(acc2[0]); acc2[1] = silu_s(acc2[1]); acc2[2] = silu_s(acc2[2]); acc2[3] = silu_s(acc2[3]); }
                        *(half4 *)(to + (g * 16 + p) * SL_OPITCH + (u1 * 16 + q * 4) * 2) = half4{(f16)acc2[0], (f16)acc2[1], (f16)acc2[2], (f16)acc2[3]};
                    }
                }
            }
            otile = to;
            __syncthreads();
        }

        // ---- 5. 16-byte NHWC stores ----
        const int cpr = a.out_c / 8;                               // 16-byte chunks per pixel
        for (int i = tid; i < SL_TW * cpr; i += SL_THREADS) {
            const int px = i / cpr, k8 = i - px * cpr;
            const int gx = ox0 + px;
            if (gx >= a.W1) continue;
            const long opix = ((long)(b * a.out_Hp + oy + a.out_pad) * a.out_Wp + gx + a.out_pad) * a.out_cs;
            *(half8 *)(a.out + opix + k8 * 8) = *(const half8 *)(otile + px * SL_OPITCH + k8 * 16);
        }
        __syncthreads();                                           // the buffers are rewritten by the next segment
    }
}

}  // namespace

bool stem_l1_supported(int c0, int c1, int tail_cout, int in_h, int in_w) {
    return c0 == 32 && c1 == 64 && (tail_cout == 0 || (tail_cout % 16 == 0 && tail_cout <= 64)) && in_w % 4 == 0 && in_h % 4 == 0 && (in_w / 4) % SL_TW == 0;
}

int launch_stem_l1(const StemL1Launch &l, hipStream_t s) {
    const int H0 = l.in_h / 2, W0 = l.in_w / 2, H1 = l.in_h / 4, W1 = l.in_w / 4;
    RT_CHECK(stem_l1_supported(32, 64, l.wt ? l.t_cout : 0, l.in_h, l.in_w), RTMODT_E_UNSUPPORTED, "launch_stem_l1: shape");
    RT_CHECK(l.w0 && l.b0 && l.w1 && l.b1 && l.kp1 >= 288 && l.out.base && l.out.H == H1 && l.out.W == W1 && l.out.coff % 8 == 0 && l.out.C % 8 == 0 &&
                 l.out.c == (l.wt ? l.t_cout : 64) && (!l.wt || (l.bt && l.kpt >= 64)),
             RTMODT_E_INVALID, "launch_stem_l1: operands");
    RT_CHECK((l.frames != nullptr) != (l.img4.base != nullptr), RTMODT_E_INVALID, "launch_stem_l1: exactly one source (frames or image tensor)");
    StemL1Args a{};
    if (l.frames) {
        RT_CHECK(!l.g.resize && l.frame0 >= 0 && l.frame0 + l.B <= 64 && (long)l.g.src_h * l.pitch < (1L << 31), RTMODT_E_INVALID, "launch_stem_l1: frames");
        a.frames = *l.frames; a.frame0 = l.frame0; a.pitch = l.pitch; a.top = l.g.top; a.left = l.g.left; a.new_h = l.g.new_h; a.new_w = l.g.new_w;
    } else {
        RT_CHECK(l.img4.C == 4 && l.img4.pad == 1 && l.img4.H == l.in_h && l.img4.W == l.in_w, RTMODT_E_INVALID, "launch_stem_l1: image tensor");
        a.img = l.img4.base;
    }
    a.in_h = l.in_h; a.in_w = l.in_w; a.H0 = H0; a.W0 = W0; a.H1 = H1; a.W1 = W1;
    a.w0 = l.w0; a.b0 = l.b0; a.w1 = l.w1; a.b1 = l.b1; a.kp1 = l.kp1;
    a.wt = l.wt; a.bt = l.bt; a.kpt = l.kpt; a.t_cout = l.t_cout; a.t_act = l.t_act;
    a.out = l.out.base + l.out.coff; a.out_Hp = l.out.H + 2 * l.out.pad; a.out_Wp = l.out.W + 2 * l.out.pad; a.out_cs = l.out.C; a.out_pad = l.out.pad; a.out_c = l.out.c;
    a.B = l.B; a.segs_per_row = W1 / SL_TW; a.n_seg = l.B * H1 * a.segs_per_row;
    static bool attr_set[64] = {};
    int dev = 0;
    RT_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        RT_HIP(hipFuncSetAttribute((const void *)stem_l1_fused, hipFuncAttributeMaxDynamicSharedMemorySize, SL_LDS));
        attr_set[dev] = true;
    }
    static const int persist = getenv("RTMODT_SL1_GRID") ? atoi(getenv("RTMODT_SL1_GRID")) : 2 * 256;      // experiment hook: workgroups in the grid
    const int grid = std::min(a.n_seg, persist > 0 ? persist : a.n_seg);   // two persistent workgroups per CU
    hipLaunchKernelGGL(stem_l1_fused, dim3(grid), dim3(SL_THREADS), SL_LDS, s, a);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

}  // namespace rtmodt
