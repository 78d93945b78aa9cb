// conv.hip -- the YOLOv8 forward pass's kernels for gfx950 (CDNA4, wave64).
//
// What they replace: the ATen/cuDNN kernels `ultralytics.YOLO.predict` runs for the call at
// /root/reference/src/detection/detector.py:100-111.
//
//  * conv_mfma  -- fused Conv(+folded BN)+bias+SiLU(+residual) as an implicit GEMM on the
//                  matrix cores (v_mfma_f32_16x16x32_f16, fp32 accumulate).  NHWC fp16,
//                  K order (kh, kw, cin).  Both operands are K-contiguous, so each 16-row x
//                  32-k block of either operand is ONE 1-KiB LDS-DMA piece
//                  (global_load_lds_dwordx4: per-lane global source, lane-linear LDS image)
//                  that a wave later reads back with one conflict-free ds_read_b128 at
//                  lane*16 -- no VGPR staging, no ds_write, no bank conflicts by construction.
//                  Two LDS stages, one barrier per 32-deep k-step; the DMA of step k+1 flies
//                  under the MFMAs of step k.  Weights are the MFMA "A" operand (rows =
//                  cout) and pixels the "B" operand (cols = pixel), so a lane ends up holding
//                  4 consecutive output channels of one pixel: an 8-byte NHWC store.
//                  Input halo comes from the tensors' zero border (kernels.h) -- no bounds
//                  checks in the k-loop.  Channel-slice views make C2f split/concat, the
//                  neck concats and the Detect-head fusion copy-free.
//  * stem_conv  -- 3->cout 3x3/s2 conv (K = 27, too thin for MFMA): one thread per output
//                  pixel, weights wave-uniform (scalar loads), fp32 accumulate.
//  * sppf_pool  -- the three chained 5x5 max-pools of SPPF as 5/9/13 windows from one LDS tile.
//  * upsample2  -- nearest 2x into a channel slice of the concat tensor.
#include "kernels.h"

namespace rtmodt {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const f16 *in;      // input tensor base + channel offset
    const f16 *wt;
    const float *bias;
    f16 *out;           // output tensor base + channel offset
    const f16 *res;     // residual tensor base + channel offset, or nullptr
    int in_Hp, in_Wp, in_cs, in_org;     // padded dims, pixel stride, (pad - ks/2) = top-left tap origin
    int out_Hp, out_Wp, out_cs, out_pad;
    int res_Hp, res_Wp, res_cs, res_pad;
    int Ho, Wo, M;                       // M = B*Ho*Wo
    int cin, cout, ks, stride, act, kp, K;
};

__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

template <int BM, int BN, int WM, int WN, bool GENERAL>
__global__ __launch_bounds__(256) void conv_mfma(ConvArgs p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BM % 64 == 0 && BN % 16 == 0, "tile shape");
    constexpr int NA = BM / 16, NB = BN / 16, NRB = NA + NB;
    constexpr int LA = NA / 4;                 // A pieces per wave per k-step
    constexpr int LB = (NB + 3) / 4;           // B pieces per wave per k-step (last may be absent)
    constexpr int STAGE = NRB * 1024;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int HoWo = p.Ho * p.Wo;

    // ---- loader set-up: element offsets of this lane's 16-byte chunk in each piece ----
    int a_off[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int m = m0 + (wave + 4 * i) * 16 + r;
        m = m < p.M ? m : p.M - 1;                       // tail rows re-read the last pixel (masked at store)
        int b = m / HoWo, rem = m - b * HoWo;
        int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_off[i] = ((b * p.in_Hp + oy * p.stride + p.in_org) * p.in_Wp + ox * p.stride + p.in_org) * p.in_cs;
    }
    int b_off[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        int n = n0 + (wave + 4 * i) * 16 + r;            // weights are zero-padded to a multiple of 128 rows
        b_off[i] = n * p.kp + q * 8;
    }

    floatx4 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.kp / 32;
    // uniform path: (kh, kw, c0) of the current k-step are wave-uniform scalars
    int kh = 0, kw = 0, c0 = 0;
    // general path: per-lane position of chunk q inside K
    int g_tap = 0, g_c = q * 8;
    if (GENERAL) {
        while (g_c >= p.cin) { g_c -= p.cin; ++g_tap; }
    }

    auto issue = [&](int kt, int stage) {
        unsigned char *sbase = lds + stage * STAGE;
        int tap_off;
        if (!GENERAL) {
            tap_off = (kh * p.in_Wp + kw) * p.in_cs + c0 + q * 8;
        } else {
            int t = g_tap < p.ks * p.ks ? g_tap : 0;     // K tail (zero weights): any valid address
            int th = p.ks == 3 ? (t * 11) >> 5 : 0;      // t / 3 for t < 9
            int tw = t - th * p.ks;
            tap_off = (th * p.in_Wp + tw) * p.in_cs + (g_tap < p.ks * p.ks ? g_c : 0);
        }
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const f16 *src = p.in + (a_off[i] + tap_off);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(sbase + (wave + 4 * i) * 1024),
                                             16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            if (wave + 4 * i < NB) {
                const f16 *src = p.wt + (b_off[i] + kt * 32);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)src,
                    (__attribute__((address_space(3))) void *)(sbase + (NA + wave + 4 * i) * 1024), 16, 0, 0);
            }
        }
        // advance to the next k-step
        if (!GENERAL) {
            c0 += 32;
            if (c0 >= p.cin) {
                c0 = 0;
                if (++kw == p.ks) { kw = 0; ++kh; }
            }
        } else {
            g_c += 32;
            if (g_c >= p.cin) { g_c -= p.cin; ++g_tap; }
            if (g_c >= p.cin) { g_c -= p.cin; ++g_tap; }
            if (g_c >= p.cin) { g_c -= p.cin; ++g_tap; }   // cin >= 16 (multiple of 8): at most 2, 3rd for safety
        }
    };

    const int wm = wave / WN, wn = wave % WN;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                                  // stage kt landed (vmcnt(0)) and stage kt-1 fully read
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const unsigned char *sbase = lds + (kt & 1) * STAGE;
        half8 fa[TM], fb[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) fa[t] = *(const half8 *)(sbase + (wm * TM + t) * 1024 + lane * 16);
#pragma unroll
        for (int u = 0; u < TN; ++u) fb[u] = *(const half8 *)(sbase + (NA + wn * TN + u) * 1024 + lane * 16);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u)
                acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[u], fa[t], acc[t][u], 0, 0, 0);
    }

    // ---- epilogue: D[row = cout (lane>>4)*4+j][col = pixel lane&15] ----
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        int m = m0 + (wm * TM + t) * 16 + r;
        if (m >= p.M) continue;
        int b = m / HoWo, rem = m - b * HoWo;
        int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        long opix = ((long)(b * p.out_Hp + oy + p.out_pad) * p.out_Wp + ox + p.out_pad) * p.out_cs;
        long rpix = 0;
        if (p.res) rpix = ((long)(b * p.res_Hp + oy + p.res_pad) * p.res_Wp + ox + p.res_pad) * p.res_cs;
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            int n = n0 + (wn * TN + u) * 16 + q * 4;
            if (n >= p.cout) continue;
            floatx4 bv = *(const floatx4 *)(p.bias + n);
            floatx4 v = acc[t][u] + bv;
            if (p.act) {
                v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]);
            }
            if (p.res) {
                half4 rv = *(const half4 *)(p.res + rpix + n);
                v[0] += (float)rv[0]; v[1] += (float)rv[1]; v[2] += (float)rv[2]; v[3] += (float)rv[3];
            }
            half4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            *(half4 *)(p.out + opix + n) = o;
        }
    }
}

TileShape tile_shape(int tile) {
    switch (tile) {
        case TILE_128x128: return {128, 128};
        case TILE_128x64: return {128, 64};
        case TILE_64x64: return {64, 64};
        case TILE_256x32: return {256, 32};
        case TILE_64x128: return {64, 128};
    }
    return {0, 0};
}

template <int BM, int BN, int WM, int WN>
static void launch_tile(const ConvArgs &a, bool general, hipStream_t s) {
    dim3 grid(cdiv(a.M, BM), cdiv(a.cout, BN));
    if (general)
        hipLaunchKernelGGL((conv_mfma<BM, BN, WM, WN, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((conv_mfma<BM, BN, WM, WN, false>), grid, dim3(256), 0, s, a);
}

int launch_conv(const ConvLaunch &c, hipStream_t s) {
    RT_CHECK(c.in.base && c.out.base && c.wt && c.bias, RTMODT_E_INVALID, "launch_conv: null operand");
    RT_CHECK(c.ks == 1 || c.ks == 3, RTMODT_E_INVALID, "launch_conv: kernel size %d", c.ks);
    RT_CHECK(c.cin % 8 == 0 && c.cout % 4 == 0, RTMODT_E_INVALID, "launch_conv: cin %d / cout %d granularity", c.cin, c.cout);
    RT_CHECK(c.in.pad >= c.ks / 2, RTMODT_E_INVALID, "launch_conv: input border %d < %d", c.in.pad, c.ks / 2);
    RT_CHECK(c.in.c == c.cin && c.out.c == c.cout, RTMODT_E_INVALID, "launch_conv: view/channel mismatch");
    RT_CHECK(c.in.coff % 8 == 0 && c.out.coff % 4 == 0 && c.in.C % 8 == 0 && c.out.C % 4 == 0, RTMODT_E_INVALID,
             "launch_conv: slice alignment");
    ConvArgs a;
    a.in = c.in.base + c.in.coff;
    a.wt = c.wt;
    a.bias = c.bias;
    a.out = c.out.base + c.out.coff;
    a.res = c.res.base ? c.res.base + c.res.coff : nullptr;
    a.in_Hp = c.in.H + 2 * c.in.pad; a.in_Wp = c.in.W + 2 * c.in.pad; a.in_cs = c.in.C; a.in_org = c.in.pad - c.ks / 2;
    a.Ho = (c.in.H + 2 * (c.ks / 2) - c.ks) / c.stride + 1;
    a.Wo = (c.in.W + 2 * (c.ks / 2) - c.ks) / c.stride + 1;
    RT_CHECK(a.Ho == c.out.H && a.Wo == c.out.W, RTMODT_E_INVALID, "launch_conv: output %dx%d != %dx%d", c.out.H, c.out.W, a.Ho, a.Wo);
    a.out_Hp = c.out.H + 2 * c.out.pad; a.out_Wp = c.out.W + 2 * c.out.pad; a.out_cs = c.out.C; a.out_pad = c.out.pad;
    a.res_Hp = c.res.H + 2 * c.res.pad; a.res_Wp = c.res.W + 2 * c.res.pad; a.res_cs = c.res.C; a.res_pad = c.res.pad;
    if (a.res) {
        RT_CHECK(c.res.H == c.out.H && c.res.W == c.out.W && c.res.c == c.cout && c.res.coff % 4 == 0 && c.res.C % 4 == 0,
                 RTMODT_E_INVALID, "launch_conv: residual shape");
    }
    a.M = c.B * a.Ho * a.Wo;
    a.cin = c.cin; a.cout = c.cout; a.ks = c.ks; a.stride = c.stride; a.act = c.act;
    a.K = c.ks * c.ks * c.cin;
    a.kp = c.kp;
    RT_CHECK(a.kp % 32 == 0 && a.kp >= a.K, RTMODT_E_INVALID, "launch_conv: kp %d for K %d", a.kp, a.K);
    // 32-bit element offsets inside the kernel
    RT_CHECK((long)c.B * a.in_Hp * a.in_Wp * a.in_cs < (1L << 31) && (long)c.B * a.out_Hp * a.out_Wp * a.out_cs < (1L << 31),
             RTMODT_E_INVALID, "launch_conv: tensor exceeds 2^31 elements");
    const bool general = (c.cin % 32) != 0;
    switch (c.tile) {
        case TILE_128x128: launch_tile<128, 128, 2, 2>(a, general, s); break;
        case TILE_128x64: launch_tile<128, 64, 2, 2>(a, general, s); break;
        case TILE_64x64: launch_tile<64, 64, 2, 2>(a, general, s); break;
        case TILE_256x32: launch_tile<256, 32, 4, 1>(a, general, s); break;
        case TILE_64x128: launch_tile<64, 128, 1, 4>(a, general, s); break;
        default: return fail(RTMODT_E_INVALID, "launch_conv: tile %d", c.tile);
    }
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

// ---------------------------------------------------------------------------------------
// stem: out[b,oy,ox,:] = silu(bias + sum_{kh,kw,c<3} w[(kh*3+kw)*3+c][:] * img[b, 2oy+kh-1, 2ox+kw-1, c])
// img is the letterboxed RGB0 fp16 image with a 1-pixel zero border (== conv zero padding).
// ---------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void stem_conv(const f16 *__restrict__ img, int Hp, int Wp, f16 *__restrict__ out,
                                                 int Ho, int Wo, int oHp, int oWp, int ocs, int opad,
                                                 const float *__restrict__ w, const float *__restrict__ bias, int total) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int b = idx / (Ho * Wo), rem = idx - b * (Ho * Wo);
    int oy = rem / Wo, ox = rem - oy * Wo;
    float x[27];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            half4 v = *(const half4 *)(img + ((long)(b * Hp + 2 * oy + kh) * Wp + 2 * ox + kw) * 4);
            x[(kh * 3 + kw) * 3 + 0] = (float)v[0];
            x[(kh * 3 + kw) * 3 + 1] = (float)v[1];
            x[(kh * 3 + kw) * 3 + 2] = (float)v[2];
        }
    f16 *o = out + ((long)(b * oHp + oy + opad) * oWp + ox + opad) * ocs;
#pragma unroll
    for (int c8 = 0; c8 < COUT; c8 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bias[c8 + j];
#pragma unroll
        for (int k = 0; k < 27; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(x[k], w[k * COUT + c8 + j], acc[j]);
        half8 hv;
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = (f16)silu_f(acc[j]);
        *(half8 *)(o + c8) = hv;
    }
}

int launch_stem(const TensorView &img4, const TensorView &out, const float *w, const float *bias, int B, int cout,
                hipStream_t s) {
    RT_CHECK(img4.C == 4 && img4.pad == 1, RTMODT_E_INVALID, "launch_stem: image tensor must be 4-channel with border");
    int Ho = (img4.H - 1) / 2 + 1, Wo = (img4.W - 1) / 2 + 1;
    RT_CHECK(Ho == out.H && Wo == out.W && out.c == cout && out.coff == 0 && out.C % 8 == 0, RTMODT_E_INVALID, "launch_stem: output shape");
    int total = B * Ho * Wo;
    dim3 grid(cdiv(total, 256));
#define STEM_CASE(NC)                                                                                                \
    case NC:                                                                                                         \
        hipLaunchKernelGGL((stem_conv<NC>), grid, dim3(256), 0, s, img4.base, img4.H + 2, img4.W + 2, out.base, Ho, Wo, \
                           out.H + 2 * out.pad, out.W + 2 * out.pad, out.C, out.pad, w, bias, total);                 \
        break;
    switch (cout) {
        STEM_CASE(16) STEM_CASE(32) STEM_CASE(48) STEM_CASE(64) STEM_CASE(80)
        default: return fail(RTMODT_E_UNSUPPORTED, "launch_stem: cout %d", cout);
    }
#undef STEM_CASE
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

// ---------------------------------------------------------------------------------------
// SPPF pools.  MaxPool(5,1,2) chained three times == windows 5 / 9 / 13 clipped to the image
// (-inf padding).  One workgroup per (image, 8-channel chunk): the HxW x 8ch tile sits in
// LDS, row maxima for the three radii are formed once, then column maxima.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ half8 hmax8(half8 a, half8 b) {
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = a[j] > b[j] ? a[j] : b[j];
    return o;
}

__global__ __launch_bounds__(256) void sppf_pool(const f16 *__restrict__ y, int Hp, int Wp, int cs, int pad, int H, int W,
                                                 f16 *__restrict__ o1, f16 *__restrict__ o2, f16 *__restrict__ o3, int ocs,
                                                 int chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *tile = (half8 *)smem;               // [H*W]
    half8 *r5 = tile + H * W;                  // row maxima radius 2
    half8 *r9 = r5 + H * W;                    // radius 4
    half8 *r13 = r9 + H * W;                   // radius 6
    int b = blockIdx.x / chunks, ch = (blockIdx.x % chunks) * 8;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        int yy = i / W, xx = i - yy * W;
        tile[i] = *(const half8 *)(y + ((long)(b * Hp + yy + pad) * Wp + xx + pad) * cs + ch);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H * W; i += 256) {
        int yy = i / W, xx = i - yy * W;
        half8 m = tile[i];
        for (int d = 1; d <= 2; ++d) {
            if (xx - d >= 0) m = hmax8(m, tile[i - d]);
            if (xx + d < W) m = hmax8(m, tile[i + d]);
        }
        r5[i] = m;
        for (int d = 3; d <= 4; ++d) {
            if (xx - d >= 0) m = hmax8(m, tile[i - d]);
            if (xx + d < W) m = hmax8(m, tile[i + d]);
        }
        r9[i] = m;
        for (int d = 5; d <= 6; ++d) {
            if (xx - d >= 0) m = hmax8(m, tile[i - d]);
            if (xx + d < W) m = hmax8(m, tile[i + d]);
        }
        r13[i] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H * W; i += 256) {
        int yy = i / W, xx = i - yy * W;
        half8 a = r5[i], bq = r9[i], c = r13[i];
        for (int d = 1; d <= 6; ++d) {
            if (yy - d >= 0) {
                if (d <= 2) a = hmax8(a, r5[i - d * W]);
                if (d <= 4) bq = hmax8(bq, r9[i - d * W]);
                c = hmax8(c, r13[i - d * W]);
            }
            if (yy + d < H) {
                if (d <= 2) a = hmax8(a, r5[i + d * W]);
                if (d <= 4) bq = hmax8(bq, r9[i + d * W]);
                c = hmax8(c, r13[i + d * W]);
            }
        }
        long opix = ((long)(b * Hp + yy + pad) * Wp + xx + pad) * ocs + ch;
        *(half8 *)(o1 + opix) = a;
        *(half8 *)(o2 + opix) = bq;
        *(half8 *)(o3 + opix) = c;
    }
}

int launch_sppf_pool(const TensorView &y, const TensorView &p1, const TensorView &p2, const TensorView &p3, int B,
                     hipStream_t s) {
    RT_CHECK(y.c % 8 == 0 && y.coff % 8 == 0 && p1.base == y.base && p2.base == y.base && p3.base == y.base, RTMODT_E_INVALID,
             "launch_sppf_pool: slices must live in one tensor");
    size_t smem = (size_t)y.H * y.W * 16 * 4;
    RT_CHECK(smem <= 150 * 1024, RTMODT_E_UNSUPPORTED, "launch_sppf_pool: %dx%d tile exceeds LDS", y.H, y.W);
    int chunks = y.c / 8;
    static size_t attr_bytes = 0;                      // raised once, outside graph capture (engine runs an eager pass first)
    if (smem > attr_bytes) {
        RT_HIP(hipFuncSetAttribute((const void *)sppf_pool, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_bytes = smem;
    }
    hipLaunchKernelGGL(sppf_pool, dim3(B * chunks), dim3(256), smem, s, y.base + y.coff, y.H + 2 * y.pad, y.W + 2 * y.pad, y.C,
                       y.pad, y.H, y.W, p1.base + p1.coff, p2.base + p2.coff, p3.base + p3.coff, y.C, chunks);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample2(const f16 *__restrict__ in, int iHp, int iWp, int ics, int ipad, int H, int W,
                                                 int chunks, f16 *__restrict__ out, int oHp, int oWp, int ocs, int opad,
                                                 long total) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;       // over (b, 2H, 2W, chunk)
    if (idx >= total) return;
    int ch = (int)(idx % chunks) * 8;
    long pix = idx / chunks;
    int ox = (int)(pix % (2 * W));
    long t = pix / (2 * W);
    int oy = (int)(t % (2 * H));
    int b = (int)(t / (2 * H));
    half8 v = *(const half8 *)(in + ((long)(b * iHp + (oy >> 1) + ipad) * iWp + (ox >> 1) + ipad) * ics + ch);
    *(half8 *)(out + ((long)(b * oHp + oy + opad) * oWp + ox + opad) * ocs + ch) = v;
}

int launch_upsample2(const TensorView &in, const TensorView &out, int B, hipStream_t s) {
    RT_CHECK(out.H == 2 * in.H && out.W == 2 * in.W && out.c == in.c && in.c % 8 == 0 && in.coff % 8 == 0 && out.coff % 8 == 0 &&
                 in.C % 8 == 0 && out.C % 8 == 0,
             RTMODT_E_INVALID, "launch_upsample2: shape");
    int chunks = in.c / 8;
    long total = (long)B * out.H * out.W * chunks;
    hipLaunchKernelGGL(upsample2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in.base + in.coff, in.H + 2 * in.pad,
                       in.W + 2 * in.pad, in.C, in.pad, in.H, in.W, chunks, out.base + out.coff, out.H + 2 * out.pad,
                       out.W + 2 * out.pad, out.C, out.pad, total);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

}  // namespace rtmodt
