// tile_math.h -- the integer arithmetic the conv kernels share with their host-side tests: plain C++, no HIP types, so that
// tests/test_tile_math_cpu.py can compile it with g++ and check it exhaustively on the build box (no GPU).
#pragma once

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtmodt {

// Division by a launch constant: n / d == (n * mul) >> shift for every 0 <= n < 2^31 (shift = 31 + ceil(log2 d),
// mul = ceil(2^shift / d) < 2^32: the error term n * (mul * d - 2^shift) stays below 2^shift).  One 32x32->64 multiply and
// one shift instead of the ~30-instruction expansion of an integer division: the epilogues turn a pixel index into
// (image, row, column) once per 16-byte store.
struct FastDiv { unsigned mul, shift; };
inline FastDiv make_fastdiv(int d) {
    unsigned s = 0;
    while ((1u << s) < (unsigned)d) ++s;
    const unsigned long long p2 = 1ull << (31 + s);
    return FastDiv{(unsigned)((p2 + (unsigned)d - 1) / (unsigned)d), 31 + s};
}
RT_HD int fdiv(int n, const FastDiv &f) { return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.shift); }

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (launch-linear id & 7); XCD x works on the x-th
// contiguous eighth of the n tiles.  Returns the tile id of workgroup `lin`.
RT_HD int xcd_tile_id(int n, int lin) {
    const int q = n >> 3, r = n & 7, xcd = lin & 7, k = lin >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

// Persistent tile kernel (conv_mfma64_pt): which pixel tiles workgroup g of cout slice `slice` walks over.  `groups` workgroups
// per slice, n_mt pixel tiles.  XCD x owns the x-th contiguous run of the tile list; inside it the XCD's workgroups of this
// slice (rank j of Gx) take tiles run0 + j, run0 + j + Gx, ...  With fewer than 8 workgroups per slice every workgroup gets a
// run of its own, so that every run has an owner.
struct PtRun { int run0, run_n, j, Gx; };
RT_HD PtRun pt_run(int g, int slice, int groups, int n_mt) {
    const bool few = groups < 8;
    const int X = few ? groups : 8;
    const int xcd = few ? g : (g + slice * groups) & 7;
    const int g0 = few ? g : (xcd - slice * groups) & 7;           // the first workgroup of this slice on that XCD
    PtRun r;
    r.j = (g - g0) >> 3;
    r.Gx = few ? 1 : (groups - g0 + 7) >> 3;
    const int q = n_mt / X, rem = n_mt - q * X;
    r.run0 = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    r.run_n = q + (xcd < rem ? 1 : 0);
    return r;
}

// LDS swizzle of the fused Bottleneck's activation images (bottleneck.hip): where the 16-byte chunk c of row R sits inside its
// row, and back.  The MFMA fragment reads of its two convs start at ARBITRARY rows (pixel + tap offset), not at multiples of 16
// like the tile kernels', so the map has to be conflict-free for any 16 consecutive rows under ds_read_b128's lane groups
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: inside a group the k-chunk index q is 0 for r in [0, 4) u [12, 16) and 1 for
// r in [4, 12), or the other way round):
//   64-byte rows (c = 32; 4 rows per 256-byte bank row): slot = c ^ 2 * bit 2 of R        (an involution)
//   128-byte rows (2 rows per bank row):                  slot = (c + 2 * (R >> 1)) mod 8  (a rotation; its inverse on the DMA source)
template <int CB> RT_HD int swz_slot(int R, int c) { return CB == 128 ? ((c + 2 * (R >> 1)) & 7) : (c ^ (((R >> 2) & 1) << 1)); }
template <int CB> RT_HD int swz_src(int R, int slot) { return CB == 128 ? ((slot - 2 * (R >> 1)) & 7) : (slot ^ (((R >> 2) & 1) << 1)); }
// byte offset of 16-byte chunk c of row R inside a plane of 1-KiB DMA pieces (8 rows x 128 B or 16 rows x 64 B)
template <int CB> RT_HD int swz_plane_off(int R, int c) {
    return CB == 128 ? (R >> 3) * 1024 + (R & 7) * 128 + (swz_slot<CB>(R, c) << 4) : (R >> 4) * 1024 + (R & 15) * 64 + (swz_slot<CB>(R, c) << 4);
}

// ---- nms_kernel (postprocess.hip) ----
// k-th (0-based) set bit of w; k < popcount(w): binary search over the halves' popcounts
RT_HD int kth_set_bit(unsigned long long w, int k) {
    int b = 0;
    unsigned x = (unsigned)w;
    int c = __builtin_popcount(x);
    if (k >= c) { k -= c; b = 32; x = (unsigned)(w >> 32); }
    c = __builtin_popcount(x & 0xFFFFu); if (k >= c) { k -= c; b += 16; x >>= 16; }
    c = __builtin_popcount(x & 0xFFu);   if (k >= c) { k -= c; b += 8; x >>= 8; }
    c = __builtin_popcount(x & 0xFu);    if (k >= c) { k -= c; b += 4; x >>= 4; }
    c = __builtin_popcount(x & 0x3u);    if (k >= c) { k -= c; b += 2; x >>= 2; }
    if (k >= (int)(x & 1u)) b += 1;
    return b;
}

// Bitonic network, G consecutive strides (s0 << (G - 1), ..., s0) per barrier step: worker t of the step owns the 2^G keys
// i0 + e * s0, i0 = t with G zero bits inserted at bit log2(s0); the schedule of one k-phase (strides k/2 ... 1) takes up to three
// strides per step from the top.  bitonic_run applies the whole network to keys[0, P) in place (descending), one worker after the
// other -- the host-side statement of what nms_kernel's threads do between barriers (tests/native/tile_math_check.cpp).
RT_HD int bitonic_i0(int t, int ls, int G, int s0) { return ((t >> ls) << (ls + G)) | (t & (s0 - 1)); }
RT_HD int bitonic_group(int j) { return j >= 4 ? 3 : (j >= 2 ? 2 : 1); }        // strides taken by the step that starts at stride j
template <typename K>
inline void bitonic_run(K *keys, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        int j = k >> 1;
        while (j > 0) {
            const int g = bitonic_group(j), s0 = j >> (g - 1), E = 1 << g;
            int ls = 0;
            while ((1 << ls) < s0) ++ls;
            for (int t = 0; t < (P >> g); ++t) {
                const int i0 = bitonic_i0(t, ls, g, s0);
                const bool desc = (i0 & k) == 0;
                for (int h = E >> 1; h >= 1; h >>= 1)
                    for (int e = 0; e < E; ++e)
                        if ((e & h) == 0) {
                            K &x = keys[i0 + e * s0], &y = keys[i0 + (e | h) * s0];
                            if (desc ? (x < y) : (x > y)) { const K tmp = x; x = y; y = tmp; }
                        }
            }
            j >>= g;
        }
    }
}

// ---- conv3x3_pp (conv_pp.hip) ----
// The ping-pong kernel's DMA stream, per wave: what a wave issues in phase kw (= tap kw) of super-step s, in issue order, and when each piece is
// first read.  lb = weight pieces per wave and tap; wave 0 carries the 33rd strip piece.
//   kw 0:  B(s, kw 2) x lb            A(s+1) x 2 (+1 for wave 0)
//   kw 1:  B(s+1, kw 0) x lb          A(s+1) x 2
//   kw 2:  B(s+1, kw 1) x lb
// vmcnt retires in issue order, so "everything phase kw + 1 reads has landed" == "at most N pieces are outstanding", N = the pieces issued after the
// youngest piece phase kw + 1 reads:
//   kw 0 -> B(s, kw 1), issued in kw 2 of s - 1: this phase's pieces are younger               N = lb + 2 (+1)
//   kw 1 -> B(s, kw 2), the first lb of kw 0: kw 0's strip pieces and this phase's are younger   N = 2 (+1) + lb + 2
//   kw 2 -> A(s+1) and B(s+1, kw 0): only this phase's weight pieces are younger                 N = lb
// (tests/native/tile_math_check.cpp replays the stream and checks the three counts: never too loose, never tighter than needed)
struct PpIssue { int nB, nA; };
// la0 / la1 = strip pieces per wave in kw 0 / kw 1 (2 + 2 for the 256-position tiles, 4 + 4 for the 512-position one)
RT_HD PpIssue pp_issue(int kw, int lb, bool wave0, int la0 = 2, int la1 = 2) { return kw == 0 ? PpIssue{lb, la0 + (wave0 ? 1 : 0)} : (kw == 1 ? PpIssue{lb, la1} : PpIssue{lb, 0}); }
constexpr int pp_wait_count(int kw, int lb, bool wave0, int la0 = 2, int la1 = 2) { return kw == 0 ? lb + la0 + (wave0 ? 1 : 0) : (kw == 1 ? la0 + (wave0 ? 1 : 0) + lb + la1 : lb); }

// The ping-pong kernels' epilogue (conv_pp.hip: PpOut) forms element offsets with 24-bit multiplies: every factor -- the padded H x W x C of ONE image of
// the output tensor and of the shortcut tensor, and Ho x (positions enumerated per input row) -- must stay below 2^24, the enumerated positions below 2^30.
// Shared by the launch check (conv_pp.hip: pp_out_fits) and by the tuner's candidate filter (engine.hip: tile_legal), so that a tile the kernel would
// refuse is never timed nor taken from a cache (YOLOv8l's layer-2 concat tensor, 258 x 258 x 320 at 1024 pixels, exceeds it).
RT_HD bool pp_index_fits(long out_hwc, long res_hwc /* 0: no shortcut */, long ho_inwp, long m) {
    const long lim = 1L << 24;
    return out_hwc < lim && res_hwc < lim && ho_inwp < lim && m < (1L << 30);
}

// Which tiles each persistent workgroup of a (grouped) ping-pong launch runs: longest-processing-time-first inside each XCD's share.
// A launch holds the tiles of up to 6 problems back to back (launch-linear ids, problem z = ids start[z] .. start[z] + tiles[z]), sorted deepest K
// first; workgroup g of G (G % 8 == 0 or G < 8) sits, under round-robin placement, on XCD g & 7 and is given ids with the same residue mod 8 (the ids
// of one residue are one XCD's share of xcd_tile_id's tile order: neighbouring tiles meet in one L2).  With the static stride g, g + G, ... a launch
// whose problems differ in depth is badly balanced -- Detect stage 0 at 32 frames: 53 tiles of 72 taps x 64 channels, 205 of 36, 810 of 18; the
// workgroups that start with a deep tile then take as many more as everybody else (284 k clk against 159 k).  Here every id goes, in order of
// decreasing cost, to the least loaded workgroup of its XCD.  Returns T (entries per workgroup); table[g * T + j] = j-th id of workgroup g, or -1.
// cost[z] = relative cost of one tile of problem z.
inline int pp_lpt_schedule(int n, const int *start, const int *tiles, const long *cost, int G, int *table /* G * max_T */, int max_T) {
    const int X = G >= 8 ? 8 : 1;
    long load[1024] = {};
    int cnt[1024] = {};
    if (G > 1024) return -1;
    for (int i = 0; i < G * max_T; ++i) table[i] = -1;
    // problems are visited in order of decreasing cost (stable), their ids ascending
    int order[16];
    for (int z = 0; z < n; ++z) order[z] = z;
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && cost[order[j]] > cost[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    for (int oi = 0; oi < n; ++oi) {
        const int z = order[oi];
        for (int k = 0; k < tiles[z]; ++k) {
            const int id = start[z] + k, x = X == 8 ? (id & 7) : 0;
            int best = -1;
            for (int g = x; g < G; g += X)
                if (best < 0 || load[g] < load[best]) best = g;
            if (cnt[best] >= max_T) return -1;
            table[best * max_T + cnt[best]++] = id;
            load[best] += cost[z];
        }
    }
    int T = 0;
    for (int g = 0; g < G; ++g) T = cnt[g] > T ? cnt[g] : T;
    return T;
}

}  // namespace rtmodt
