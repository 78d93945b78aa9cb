// conv_dev.h -- device-side helpers shared by the conv kernels of conv.hip and conv_pp.hip: kernel arguments, the LDS-DMA
// primitive, counted waits, the epilogues (direct stores, through LDS, with a fused 1x1 tail) and the tap-reuse position map.
#pragma once

#include "kernels.h"
#include "tile_math.h"

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
    FastDiv d_howo, d_wo;                // m -> (image, row, column) of the output
    FastDiv d_hwp, d_wp;                 // tap-reuse kernel: division by Ho * rows_wq (the positions of one image it enumerates) and by rows_wq
    int rows_wq;                         // tap-reuse kernel: positions enumerated per image row = in_Wp - 1 (the right border column is skipped)
    int last_pos;                        // tap-reuse kernel: index of the last padded position of the input tensor (B * in_Hp * in_Wp - 1)
    f16 *out2;                           // optional second destination: nearest-2x upsampled copy (neck concat slice)
    int out2_Hp, out2_Wp, out2_cs, out2_pad;
    int cin, cout, ks, stride, act, kp, K;
    const f16 *in2;                      // optional half-resolution source of channels [0, split) (nearest-2x on the fly), 1x1 only
    int in2_Hp, in2_Wp, in2_cs, in2_pad, split;
    // optional fused TAIL: a 1x1 conv (+bias +SiLU) applied to this conv's output tile while it sits in LDS; only the
    // tail's output is stored (the tile kernels with BN == cout, TAIL instantiations)
    const f16 *t_wt; const float *t_bias; f16 *t_out;
    int t_cout, t_kp, t_act, t_out_Hp, t_out_Wp, t_out_cs, t_out_pad;
    int epi16;                           // epilogue through LDS with 16-byte NHWC stores (all channel offsets / strides % 8 == 0):
                                         // 1 = tile kernels, 2 = also the tap-reuse kernel
    int epi_prio;                        // experiment hook RTMODT_EPI_PRIO: 1 = a wave raises its issue priority for its epilogue, 2 = lowers it (main loops at 1)
    int wthru;                           // output stores are WRITE-THROUGH (sc1): the tile leaves the XCD's L2 while the kernel still runs,
                                         // instead of as one write-back of every dirty line at the kernel boundary
};

__device__ __forceinline__ float silu_f(float x) { return silu(x); }

// RTMODT_EPI_PRIO (ConvArgs::epi_prio): wave issue priority around an epilogue
__device__ __forceinline__ void prio_main(int mode) { if (mode == 2) __builtin_amdgcn_s_setprio(1); else if (mode == 1) __builtin_amdgcn_s_setprio(0); }
__device__ __forceinline__ void prio_epilogue(int mode) { if (mode == 1) __builtin_amdgcn_s_setprio(1); else if (mode == 2) __builtin_amdgcn_s_setprio(0); }

// Output stores.  A plain store leaves the line dirty in the XCD's L2 and the whole output is written back at the kernel
// boundary (B / ~6 TB/s with nothing else running); a write-through (sc1) store sends it on its way at once, under the
// rest of the kernel.  Values and addresses are identical either way.  Measured per launch at 16 frames (profiles/r02):
// the tile kernels' 16-byte stores gain 5-12 % (6.cv2 21.1 -> 18.5 us), the 8-byte stores of the accumulator layout LOSE
// 5-10 % written through (an 8-byte sc1 store costs 2.7x a 16-byte one per byte), so only mode 2 (A/B hook) sends those too.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16(f16 *base, long off, const half8 &v, int wt) {
    if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0xFFFFFFFF, 0x00020000), (unsigned)(off * 2), 0, 16);
    else *(half8 *)(base + off) = v;
}
__device__ __forceinline__ void store8(f16 *base, long off, const half4 &v, int wt) {
    if (wt > 1) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0xFFFFFFFF, 0x00020000), (unsigned)(off * 2), 0, 16);
    else *(half4 *)(base + off) = v;
}

// XCD-aware tile order (speed only, any placement is correct).  Workgroups are dealt round-robin
// over the 8 XCDs, each with its own 4 MiB L2; a 3x3 conv re-reads its input 9 times (taps) and
// once per cout tile, so the tiles an XCD works on should be NEIGHBOURS: the launch-linear id is
// remapped (bijectively) so that ids congruent mod 8 -- one XCD under round-robin placement --
// cover one contiguous run of pixel tiles with all their cout tiles.
__device__ __forceinline__ void xcd_tile(int gx, int gy, int bx, int by, int &mt, int &nt) {
    const int id = xcd_tile_id(gx * gy, bx + by * gx);          // tile_math.h
    mt = id / gy;
    nt = id - mt * gy;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// position of a 32-deep k-step inside K = (kh, kw, cin): wave-uniform when cin % 32 == 0
struct KPos {
    int kh = 0, kw = 0, c0 = 0;
    __device__ __forceinline__ void advance(int steps, int cin, int ks) {
        c0 += 32 * steps;
        while (c0 >= cin) {
            c0 -= cin;
            if (++kw == ks) { kw = 0; ++kh; }
        }
    }
};
// per-lane position of this lane's 8-channel chunk (general path: cin % 32 != 0, cin % 8 == 0)
struct KLane {
    int tap = 0, c = 0;
    __device__ __forceinline__ void advance(int halves, int cin) {
        c += halves;
        while (c >= cin) { c -= cin; ++tap; }
    }
    __device__ __forceinline__ int offset(int ks, int in_Wp, int in_cs) const {
        const bool in = tap < ks * ks;                   // K tail (zero weights): any valid address
        int t = in ? tap : 0;
        int th = ks == 3 ? (t * 11) >> 5 : 0;            // t / 3 for t < 9
        int tw = t - th * ks;
        return (th * in_Wp + tw) * in_cs + (in ? c : 0);
    }
};

// LDS-DMA piece = 16 rows x 32 k (64 B per row), stored ROW-MAJOR: DMA lane l fills bytes
// [16 l, 16 l + 16) = row l>>2, slot l&3, so every 16-lane quarter of the wave instruction
// touches 4 rows x 64 contiguous bytes (4 cache-line lookups, not 16).  The MFMA fragment of
// lane (r = lane&15, q = lane>>4) is k-chunk q of row r; a plain row-major image would make
// that ds_read_b128 2-way bank conflicted, so chunk q of row r is stored in slot
// q ^ swz(r), swz(r) = {0,0,3,3}[r>>2] (an involution applied on the SOURCE address, the LDS
// image itself stays lane-linear as the DMA requires) -- conflict-free for all four
// 16-lane groups of ds_read_b128.
__device__ __forceinline__ int swz16(int row) { return ((row >> 3) & 1) * 3; }
struct LaneMap {
    int ld_row, ld_chunk;    // DMA: which row / which 8-half k-chunk this lane fetches
    int rd_off;              // byte offset of this lane's fragment inside a piece
    __device__ __forceinline__ LaneMap(int lane) {
        ld_row = lane >> 2;
        ld_chunk = (lane & 3) ^ swz16(ld_row);
        int r = lane & 15, q = lane >> 4;
        rd_off = r * 64 + ((q ^ swz16(r)) << 4);
    }
};

__device__ __forceinline__ void glds16(const f16 *src, unsigned char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

// fused epilogue for one 16(pixel) x 16(cout) accumulator tile: lane holds pixel (lane&15),
// channels n .. n+3
__device__ __forceinline__ void store_tile(const ConvArgs &p, const floatx4 &acc, const floatx4 &bias, long opix, long rpix, int n, long opix2 = -1) {
    floatx4 v = acc + bias;
    if (p.act) silu4(v);
    if (p.res) {
        half4 rv = *(const half4 *)(p.res + rpix + n);
        v[0] += (float)rv[0]; v[1] += (float)rv[1]; v[2] += (float)rv[2]; v[3] += (float)rv[3];
    }
    half4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    store8(p.out, opix + n, o, p.wthru);
    if (opix2 >= 0) {                                    // Upsample(nearest, x2) + Concat folded into this epilogue
        const long row = (long)p.out2_Wp * p.out2_cs;
        store8(p.out2, opix2 + n, o, p.wthru);
        store8(p.out2, opix2 + p.out2_cs + n, o, p.wthru);
        store8(p.out2, opix2 + row + n, o, p.wthru);
        store8(p.out2, opix2 + row + p.out2_cs + n, o, p.wthru);
    }
}

__device__ __forceinline__ long upsampled_offset(const ConvArgs &p, int b, int oy, int ox) {
    return p.out2 ? ((long)(b * p.out2_Hp + 2 * oy + p.out2_pad) * p.out2_Wp + 2 * ox + p.out2_pad) * p.out2_cs : -1;
}

__device__ __forceinline__ bool pixel_offsets(const ConvArgs &p, int m, long &opix, long &rpix, long &opix2) {
    if (m >= p.M) return false;
    const int HoWo = p.Ho * p.Wo;
    int b = fdiv(m, p.d_howo), rem = m - b * HoWo;
    int oy = fdiv(rem, p.d_wo), ox = rem - oy * p.Wo;
    opix = ((long)(b * p.out_Hp + oy + p.out_pad) * p.out_Wp + ox + p.out_pad) * p.out_cs;
    rpix = p.res ? ((long)(b * p.res_Hp + oy + p.res_pad) * p.res_Wp + ox + p.res_pad) * p.res_cs : 0;
    opix2 = upsampled_offset(p, b, oy, ox);
    return true;
}

// the same pixel in the half-resolution source (1x1 stride-1 convs only)
__device__ __forceinline__ int input_offset_lo(const ConvArgs &p, int m) {
    m = m < p.M ? m : p.M - 1;
    const int HoWo = p.Ho * p.Wo;
    int b = fdiv(m, p.d_howo), rem = m - b * HoWo;
    int oy = fdiv(rem, p.d_wo), ox = rem - oy * p.Wo;
    return ((b * p.in2_Hp + (oy >> 1) + p.in2_pad) * p.in2_Wp + (ox >> 1) + p.in2_pad) * p.in2_cs;
}

__device__ __forceinline__ int input_offset(const ConvArgs &p, int m) {
    m = m < p.M ? m : p.M - 1;                           // tail rows re-read the last pixel (masked at store)
    const int HoWo = p.Ho * p.Wo;
    int b = fdiv(m, p.d_howo), rem = m - b * HoWo;
    int oy = fdiv(rem, p.d_wo), ox = rem - oy * p.Wo;
    return ((b * p.in_Hp + oy * p.stride + p.in_org) * p.in_Wp + ox * p.stride + p.in_org) * p.in_cs;
}


// Epilogue through LDS: the accumulator layout (lane = pixel r, 4 channels) stores 8 bytes per lane in 32-byte runs;
// staged as a [pixel][channel] fp16 tile instead, every lane then moves 16 bytes and a wave instruction covers whole
// cache lines of the NHWC output (and of the nearest-2x copy).  bias / SiLU / residual are applied on the way in, in
// fp32, exactly as store_tile does -- the stored values are identical.  `lds` = the (drained) stage buffers.
// pix(pm, opix, rpix, opix2) -> false for a row of the tile that is not an output pixel.
template <int BM, int BN, int TM, int TN, int NTHREADS = 256, typename PixFn>
__device__ __forceinline__ void epilogue_lds(const ConvArgs &p, const int n0, const floatx4 (&acc)[TM][TN], const floatx4 (&bv)[TN],
                                             unsigned char *lds, const int wm, const int wn, PixFn pix) {
    constexpr int ROWB = BN * 2 + 16;                      // +16: the b64 writes of a 16-pixel group land in distinct banks
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    __syncthreads();                                       // every wave is done reading the last stage
    STAMP(7);
    prio_epilogue(p.epi_prio);
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int pm = (wm * TM + t) * 16 + r;
        long opix = 0, rpix = 0, opix2 = 0;
        const bool live = p.res ? pix(pm, opix, rpix, opix2) : false;      // (only the residual needs the pixel's position here)
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            const int nl = (wn * TN + u) * 16 + q * 4;
            floatx4 v = acc[t][u] + bv[u];
            if (p.act) silu4(v);
            if (p.res && live && n0 + nl < p.cout) {
                half4 rv = *(const half4 *)(p.res + rpix + n0 + nl);
                v[0] += (float)rv[0]; v[1] += (float)rv[1]; v[2] += (float)rv[2]; v[3] += (float)rv[3];
            }
            *(half4 *)(lds + pm * ROWB + nl * 2) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        }
    }
    STAMP(8);
    prio_main(p.epi_prio);
    __syncthreads();
    STAMP(9);
    constexpr int CPR = BN / 8;                            // 16-byte chunks per pixel
    for (int c = threadIdx.x; c < BM * CPR; c += NTHREADS) {
        const int pm = c / CPR, k8 = c - pm * CPR, n = n0 + k8 * 8;
        if (n >= p.cout) continue;
        long opix, rpix, opix2;
        if (!pix(pm, opix, rpix, opix2)) continue;
        const unsigned char *src = lds + pm * ROWB + k8 * 16;
        if (n + 8 <= p.cout) {
            const half8 v = *(const half8 *)src;
            store16(p.out, opix + n, v, p.wthru);
            if (opix2 >= 0) {
                const long row = (long)p.out2_Wp * p.out2_cs;
                store16(p.out2, opix2 + n, v, p.wthru);
                store16(p.out2, opix2 + p.out2_cs + n, v, p.wthru);
                store16(p.out2, opix2 + row + n, v, p.wthru);
                store16(p.out2, opix2 + row + p.out2_cs + n, v, p.wthru);
            }
        } else {                                           // cout % 8 == 4: the last chunk is half a chunk
            const half4 v = *(const half4 *)src;
            store8(p.out, opix + n, v, p.wthru);
            if (opix2 >= 0) {
                const long row = (long)p.out2_Wp * p.out2_cs;
                store8(p.out2, opix2 + n, v, p.wthru);
                store8(p.out2, opix2 + p.out2_cs + n, v, p.wthru);
                store8(p.out2, opix2 + row + n, v, p.wthru);
                store8(p.out2, opix2 + row + p.out2_cs + n, v, p.wthru);
            }
        }
    }
}


// Epilogue with a fused 1x1 TAIL conv (conv -> C2f.cv1 pairs of the backbone): the fp16 tile of this conv's output
// (all cout channels of BM pixels: BN == cout) is staged in LDS exactly as epilogue_lds does, the tail's weights
// [t_cout][cout] arrive by DMA next to it, every wave multiplies its BM/64 pixel tiles by them, and the tail's
// output tile replaces the first one in LDS on its way to 16-byte stores.  The intermediate tensor never exists in
// HBM (one launch, one store and one load less per pair).  N2T = cout tiles of the tail (t_cout <= 16 * N2T).
template <int BM, int BN, int TM, int TN, int N2T, typename PixFn>
__device__ __forceinline__ void epilogue_tail(const ConvArgs &p, const floatx4 (&acc)[TM][TN], const floatx4 (&bv)[TN], unsigned char *lds,
                                              const int wm, const int wn, PixFn pix) {
    constexpr int ROWB = BN * 2 + 16, KC2 = BN / 32, TM2 = BM / 64;
    constexpr int W2_OFF = (BM * ROWB + 1023) / 1024 * 1024;
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const LaneMap lm(lane);
    __syncthreads();                                       // every wave is done reading the last stage
    for (int pi = wave; pi < N2T * KC2; pi += 4) {         // piece (k-chunk kc, cout tile u) of the tail's weights
        const int kc = pi / N2T, u = pi - kc * N2T;
        glds16(p.t_wt + ((u * 16 + lm.ld_row) * p.t_kp + kc * 32 + lm.ld_chunk * 8), lds + W2_OFF + pi * 1024);
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {                         // this conv's tile -> LDS (bias, SiLU in fp32, one rounding)
        const int pm = (wm * TM + t) * 16 + r;
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            const int nl = (wn * TN + u) * 16 + q * 4;
            floatx4 v = acc[t][u] + bv[u];
            if (p.act) silu4(v);
            *(half4 *)(lds + pm * ROWB + nl * 2) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        }
    }
    floatx4 acc2[TM2][N2T], b2[N2T];
#pragma unroll
    for (int u = 0; u < N2T; ++u) {
        b2[u] = *(const floatx4 *)(p.t_bias + u * 16 + q * 4);                 // padded to 128 entries; added after the sum like store_tile does
#pragma unroll
        for (int i = 0; i < TM2; ++i) acc2[i][u] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    wait_vmcnt<0>();
    __syncthreads();                                       // tile and weights complete
#pragma unroll
    for (int kc = 0; kc < KC2; ++kc) {
        half8 fa[TM2], fb[N2T];
#pragma unroll
        for (int i = 0; i < TM2; ++i) fa[i] = *(const half8 *)(lds + ((wave + 4 * i) * 16 + r) * ROWB + (kc * 4 + q) * 16);
#pragma unroll
        for (int u = 0; u < N2T; ++u) fb[u] = *(const half8 *)(lds + W2_OFF + (kc * N2T + u) * 1024 + lm.rd_off);
#pragma unroll
        for (int i = 0; i < TM2; ++i)
#pragma unroll
            for (int u = 0; u < N2T; ++u) acc2[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[u], fa[i], acc2[i][u], 0, 0, 0);
    }
    __syncthreads();                                       // everyone is done reading the first tile
    constexpr int ROWB2 = N2T * 32 + 16;
#pragma unroll
    for (int i = 0; i < TM2; ++i) {
        const int pm = (wave + 4 * i) * 16 + r;
#pragma unroll
        for (int u = 0; u < N2T; ++u) {
            floatx4 v = acc2[i][u] + b2[u];
            if (p.t_act) silu4(v);
            *(half4 *)(lds + pm * ROWB2 + (u * 16 + q * 4) * 2) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        }
    }
    __syncthreads();
    constexpr int CPR = N2T * 2;                           // 16-byte chunks per pixel of the tail's tile
    for (int c = threadIdx.x; c < BM * CPR; c += 256) {
        const int pm = c / CPR, k8 = c - pm * CPR, n = k8 * 8;
        if (n >= p.t_cout) continue;
        long opix;
        if (!pix(pm, opix)) continue;
        store16(p.t_out, opix + n, *(const half8 *)(lds + pm * ROWB2 + k8 * 16), p.wthru);
    }
}


template <int BM, int BN, int N2T>
constexpr int tail_lds_bytes() {
    constexpr int ROWB = BN * 2 + 16, W2_OFF = (BM * ROWB + 1023) / 1024 * 1024;
    constexpr int a = W2_OFF + N2T * (BN / 32) * 1024, b = BM * (N2T * 32 + 16);
    return a > b ? a : b;
}
__device__ __forceinline__ bool tail_pixel_offset(const ConvArgs &p, int m, long &opix) {
    if (m >= p.M) return false;
    const int HoWo = p.Ho * p.Wo;
    const int b = fdiv(m, p.d_howo), rem = m - b * HoWo;
    const int oy = fdiv(rem, p.d_wo), ox = rem - oy * p.Wo;
    opix = ((long)(b * p.t_out_Hp + oy + p.t_out_pad) * p.t_out_Wp + ox + p.t_out_pad) * p.t_out_cs;
    return true;
}

// tap-reuse kernels: enumerated position m -> index of that pixel in the padded input tensor (not clamped: positions past the last image map
// past the tensor and are clamped to its last -- border, zero -- position where they are used)
__device__ __forceinline__ int rows_pos(const ConvArgs &p, int m) {
    const int b = fdiv(m, p.d_hwp), rem = m - b * (p.Ho * p.rows_wq);
    const int oy = fdiv(rem, p.d_wp), x = rem - oy * p.rows_wq;
    return (b * p.in_Hp + oy) * p.in_Wp + x;
}

// ---- kernel entry points: one problem per launch, or a GROUP of independent problems that
// share a tile configuration.  Grouping turns the Detect head's 15 small launches into 3.
constexpr int MAX_GROUP = 6;
// Grouped launch: a 1-D grid holding the tiles of every problem back to back (problem z owns ids start[z] .. start[z+1],
// starts rounded up to 8 so that "id & 7" stays the XCD inside each problem), deepest-K problem first.
struct ConvGroupArgs { ConvArgs p[MAX_GROUP]; int start[MAX_GROUP + 1]; int gx[MAX_GROUP]; int n; };
template <int BM, int BN>
__device__ __forceinline__ int group_pick(const ConvGroupArgs &g, int &bx, int &by) {
    int z = 0, id = blockIdx.x;
    while (z + 1 < g.n && id >= g.start[z + 1]) ++z;
    id -= g.start[z];
    const int gx = g.gx[z], gy = (g.p[z].cout + BN - 1) / BN;
    if (id >= gx * gy) return -1;                          // alignment filler
    by = id / gx; bx = id - by * gx;
    return z;
}

// conv_pp.hip: n 3x3 / stride-1 problems (ConvArgs prepared for the tap-reuse enumeration) as one ping-pong launch with cout tile bn
// compute units of the current device (the persistent kernels size their grids by it), looked up once per device
static inline int device_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

int launch_conv3x3_pp(const ConvArgs *a, int n, int bn, hipStream_t s);
// ... one conv of any kernel size / stride without tap reuse (1x1, 3x3 stride 2) on the ping-pong tile kernel
int launch_conv_tile_pp(const ConvArgs &a, int bn, hipStream_t s);

}  // namespace rtmodt
