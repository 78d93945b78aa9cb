// conv.hip -- the YOLOv8 forward pass's kernels for gfx950 (CDNA4, wave64).
//
// What they replace: the ATen/cuDNN kernels `ultralytics.YOLO.predict` runs for the call at
// /root/reference/src/detection/detector.py:100-111.
//
//  All convs are fused Conv(+folded BN)+bias+SiLU(+residual)(+nearest-2x copy) implicit GEMMs on the
//  matrix cores (v_mfma_f32_16x16x32_f16, fp32 accumulate), NHWC fp16, K order (kh, kw, cin).
//  Weights are the MFMA "A" operand (rows = cout) and pixels the "B" operand (cols = pixel), so a
//  lane ends up holding 4 consecutive output channels of one pixel.  Both operands are
//  K-contiguous and travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: per-lane global
//  source, lane-linear LDS image) in 1-KiB pieces of whole rows (16 x 64 B or 8 x 128 B), XOR
//  swizzled on the SOURCE address so that every fragment read is a conflict-free ds_read_b128;
//  counted s_waitcnt vmcnt(N) + raw s_barrier per k-step, never vmcnt(0) inside a ring.  Input halo
//  comes from the tensors' zero border (kernels.h) -- no bounds checks in the k-loop.  Channel-slice
//  views make C2f split/concat, the neck concats and the Detect-head fusion copy-free.
//
//  * conv_mfma      -- block tile BM x BN, 32-deep k-steps, NSTAGE-deep ring (any cin % 8 == 0).
//  * conv_mfma64    -- the same with 64-deep k-steps (cin % 64 == 0): one barrier per 64 of K.
//  * conv3x3_rows   -- 3x3 / stride 1 with TAP REUSE: the GEMM runs over padded positions so one
//                      strip of input rows in LDS serves the three horizontal taps.
//  * conv_mfma_wsk  -- the 4 waves split K over one tile (small-M layers), LDS reduction.
//  * epilogue_lds   -- bias / SiLU / residual in fp32, one rounding, fp16 tile staged in the drained
//                      stage buffers, 16-byte NHWC stores (tile kernels; the rows kernel stores directly).
//  * *_grp          -- up to 6 independent problems sharing a tile configuration in ONE launch (Detect head).
//  * xcd_tile       -- workgroup id remap so that the tiles one XCD (one L2) works on are neighbours.
//  * stem_mfma / stem_fused -- 3->cout 3x3/s2 conv with K re-indexed kh*16 + kw*4 + c (two MFMA
//                      k-steps of adjacent-pixel pairs); stem_fused builds those pairs from the BGR
//                      bytes of the frame itself (letterbox, BGR->RGB, /255, .half() folded in).
//  * sppf_pool      -- the three chained 5x5 max-pools of SPPF as 5/9/13 windows from one LDS tile.
//  * upsample2      -- nearest 2x into a channel slice of the concat tensor (only when the producer
//                      cannot fold the copy into its epilogue).
#include <algorithm>

#include "conv_dev.h"

namespace rtmodt {


// ---------------------------------------------------------------------------------------
// conv_mfma: block tile BM x BN shared by 4 waves (WM x WN), THREE LDS stages.  The DMA of
// k-steps kt+1 and kt+2 is in flight while kt is multiplied: each step waits with a COUNTED
// s_waitcnt vmcnt (never 0 inside the loop) and a raw s_barrier.
// ---------------------------------------------------------------------------------------
// wait until at most `steps` k-steps' worth of this wave's DMA pieces are still in flight
template <int L, int MAXSTEPS>
__device__ __forceinline__ void wait_steps(int steps) {
    if constexpr (MAXSTEPS == 0) {
        wait_vmcnt<0>();
    } else {
        if (steps >= MAXSTEPS) wait_vmcnt<L * MAXSTEPS>();
        else wait_steps<L, MAXSTEPS - 1>(steps);
    }
}

// the same with EXTRA younger operations (stores) that may stay outstanding
template <int L, int MAXSTEPS, int EXTRA>
__device__ __forceinline__ void wait_steps_plus(int steps) {
    if constexpr (MAXSTEPS == 0) {
        wait_vmcnt<EXTRA>();
    } else {
        static_assert(L * MAXSTEPS + EXTRA <= 63, "vmcnt is a 6-bit counter");
        if (steps >= MAXSTEPS) wait_vmcnt<L * MAXSTEPS + EXTRA>();
        else wait_steps_plus<L, MAXSTEPS - 1, EXTRA>(steps);
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE, bool GENERAL, int N2T = 0>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs &p, const int bx, const int by) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BM % 64 == 0 && BN % 16 == 0, "tile shape");
    constexpr int NA = BM / 16, NB = BN / 16, NRB = NA + NB;
    constexpr int LA = NA / 4;                 // A pieces per wave per k-step
    constexpr int LB = (NB + 3) / 4;           // B pieces per wave per k-step (last may be absent)
    constexpr int LBF = NB / 4, LBR = NB % 4;  // every wave issues LBF, waves < LBR one more
    constexpr int DEPTH = NSTAGE - 1;          // k-steps of DMA kept in flight ahead of the MFMAs
    constexpr int STAGE = NRB * 1024;
    static_assert(NSTAGE >= 2 && (LA + LBF + 1) * (DEPTH - 1) <= 63, "vmcnt is a 6-bit counter");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int TAIL_BYTES = N2T > 0 ? tail_lds_bytes<BM, BN, N2T>() : 0;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTAGE * STAGE > TAIL_BYTES ? NSTAGE * STAGE : TAIL_BYTES];
    static_assert(BM * (BN * 2 + 16) <= NSTAGE * STAGE, "the epilogue's fp16 tile must fit the stage buffers");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const LaneMap lm(lane);
    int mt, nt;
    xcd_tile((p.M + BM - 1) / BM, (p.cout + BN - 1) / BN, bx, by, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- loader set-up: element offsets of this lane's 16-byte chunk in each piece ----
    int a_off[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) a_off[i] = input_offset(p, m0 + (wave + 4 * i) * 16 + lm.ld_row);
    int b_off[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) b_off[i] = (n0 + (wave + 4 * i) * 16 + lm.ld_row) * p.kp + lm.ld_chunk * 8;   // weights zero-padded to 128 rows

    floatx4 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.kp / 32;
    KPos kp;
    KLane kl;
    if (GENERAL) kl.advance(lm.ld_chunk * 8, p.cin);

    auto issue = [&](int kt, int stage) {
        unsigned char *sbase = lds + stage * STAGE;
        const int tap_off = GENERAL ? kl.offset(p.ks, p.in_Wp, p.in_cs) : (kp.kh * p.in_Wp + kp.kw) * p.in_cs + kp.c0 + lm.ld_chunk * 8;
#pragma unroll
        for (int i = 0; i < LA; ++i) glds16(p.in + (a_off[i] + tap_off), sbase + (wave + 4 * i) * 1024);
#pragma unroll
        for (int i = 0; i < LB; ++i)
            if (i < LBF || wave < LBR) glds16(p.wt + (b_off[i] + kt * 32), sbase + (NA + wave + 4 * i) * 1024);
        if (GENERAL) kl.advance(32, p.cin); else kp.advance(1, p.cin, p.ks);
    };

    const int wm = wave / WN, wn = wave % WN;
    floatx4 bv[TN];                                        // bias fetched now, consumed after the k-loop (padded to 128 rows)
#pragma unroll
    for (int u = 0; u < TN; ++u) bv[u] = *(const floatx4 *)(p.bias + n0 + (wn * TN + u) * 16 + q * 4);
    STAMP(0);
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        if (i < nk) issue(i, i);
    STAMP(1);
    int rstage = 0, wstage = DEPTH % NSTAGE;              // stage read this step / stage refilled this step
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once only the pieces of the (up to DEPTH-1) later steps are outstanding
        const int ahead = min(DEPTH - 1, nk - 1 - kt);
        if (LBR != 0 && wave < LBR) wait_steps<LA + LBF + 1, DEPTH - 1>(ahead); else wait_steps<LA + LBF, DEPTH - 1>(ahead);
        __builtin_amdgcn_s_barrier();                     // everyone's stage kt landed; everyone is done reading stage kt-1
        asm volatile("" ::: "memory");
        if (kt < 3) STAMP(2 + kt);
        if (kt + DEPTH < nk) issue(kt + DEPTH, wstage);   // refills the stage read in step kt-1
        const unsigned char *sbase = lds + rstage * STAGE;
        rstage = rstage + 1 == NSTAGE ? 0 : rstage + 1;
        wstage = wstage + 1 == NSTAGE ? 0 : wstage + 1;
        half8 fa[TM], fb[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) fa[t] = *(const half8 *)(sbase + (wm * TM + t) * 1024 + lm.rd_off);
#pragma unroll
        for (int u = 0; u < TN; ++u) fb[u] = *(const half8 *)(sbase + (NA + wn * TN + u) * 1024 + lm.rd_off);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u)
                acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[u], fa[t], acc[t][u], 0, 0, 0);
    }

    STAMP(5);
    // ---- epilogue: D[row = cout (lane>>4)*4+j][col = pixel lane&15] ----
    if constexpr (N2T > 0) {
        epilogue_tail<BM, BN, TM, TN, N2T>(p, acc, bv, lds, wm, wn, [&](int pm, long &o) { return tail_pixel_offset(p, m0 + pm, o); });
        STAMP(6);
        return;
    }
    if (p.epi16) {
        epilogue_lds<BM, BN, TM, TN>(p, n0, acc, bv, lds, wm, wn, [&](int pm, long &o, long &rp, long &o2) { return pixel_offsets(p, m0 + pm, o, rp, o2); });
        STAMP(6);
        return;
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        long opix, rpix, opix2;
        if (!pixel_offsets(p, m0 + (wm * TM + t) * 16 + r, opix, rpix, opix2)) continue;
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            int n = n0 + (wn * TN + u) * 16 + q * 4;
            if (n < p.cout) store_tile(p, acc[t][u], bv[u], opix, rpix, n, opix2);
        }
    }
}

// ---------------------------------------------------------------------------------------
// conv_mfma64: the same block-tile kernel with 64-deep k-steps for layers whose cin is a
// multiple of 64.  A DMA piece is 8 rows x 128 B (whole cache lines: a 16-lane quarter of the
// wave instruction covers 2 rows = 2 lines), a stage holds two MFMA k-substeps, so there is
// one barrier per 64 of K.  Chunk c (0..7) of row r sits in slot c ^ ((r>>1)&7): conflict-free
// ds_read_b128 for both k-substeps.
// ---------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NSTAGE, int N2T = 0>
__device__ __forceinline__ void conv_mfma64_body(const ConvArgs &p, const int bx, const int by) {
    constexpr int NW = WM * WN;                             // waves per workgroup: the DMA path sustains ~5 B/clk PER WAVE (tools/probes/dma_probe),
    static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves");      // so the big tiles run 8 (16) waves to issue their operands twice (four times) as fast
    static_assert(N2T == 0 || NW == 4, "the fused tail is written for 4 waves");
    static_assert(BM % 32 == 0 && BN % 32 == 0, "tile shape");
    constexpr int NA = BM / 8, NB = BN / 8, NP = NA + NB;   // pieces per k-step
    constexpr int LA = NA / NW, LBp = NB / NW;               // per wave
    static_assert(NA % NW == 0 && NB % NW == 0, "pieces must split evenly over the waves");
    constexpr int DEPTH = NSTAGE - 1;
    constexpr int STAGE = NP * 1024;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    static_assert((LA + LBp) * (DEPTH > 1 ? DEPTH - 1 : 1) <= 63, "vmcnt is a 6-bit counter");
    constexpr int TAIL_BYTES = N2T > 0 ? tail_lds_bytes<BM, BN, N2T>() : 0;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTAGE * STAGE > TAIL_BYTES ? NSTAGE * STAGE : TAIL_BYTES];
    constexpr bool EPI_FITS = BM * (BN * 2 + 16) <= NSTAGE * STAGE;      // the epilogue's fp16 tile goes through the drained stage buffers when it fits them
    static_assert(EPI_FITS || N2T == 0, "a fused tail needs the tile in LDS");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    int mt, nt;
    xcd_tile((p.M + BM - 1) / BM, (p.cout + BN - 1) / BN, bx, by, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;
    // DMA lane map: row8 = lane>>3 of the piece, slot = lane&7
    const int ld_row8 = lane >> 3, ld_slot = lane & 7;
    // fragment read offsets inside a 16-row tile (two pieces) for the two k-substeps
    const int rd_base = (r >> 3) * 1024 + (r & 7) * 128;
    const int rd_off0 = rd_base + (((0 + q) ^ ((r >> 1) & 7)) << 4);
    const int rd_off1 = rd_base + (((4 + q) ^ ((r >> 1) & 7)) << 4);

    int a_off[LA], a_lo[LA], b_off[LBp];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int row = (wave + NW * i) * 8 + ld_row8;             // row inside the BM tile
        a_off[i] = input_offset(p, m0 + row) + ((ld_slot ^ ((row >> 1) & 7)) << 3);
        a_lo[i] = p.in2 ? input_offset_lo(p, m0 + row) + ((ld_slot ^ ((row >> 1) & 7)) << 3) : 0;
    }
#pragma unroll
    for (int i = 0; i < LBp; ++i) {
        int row = (wave + NW * i) * 8 + ld_row8;
        b_off[i] = (n0 + row) * p.kp + ((ld_slot ^ ((row >> 1) & 7)) << 3);
    }

    floatx4 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.kp / 64;
    int kh = 0, kw = 0, c0 = 0;
    auto issue = [&](int kt, int stage) {
        unsigned char *sbase = lds + stage * STAGE;
        const int tap_off = (kh * p.in_Wp + kw) * p.in_cs + c0;
        if (p.in2 && c0 < p.split) {                       // the upsampled half of a neck concat, read where it was produced
#pragma unroll
            for (int i = 0; i < LA; ++i) glds16(p.in2 + (a_lo[i] + c0), sbase + (wave + NW * i) * 1024);
        } else {
#pragma unroll
            for (int i = 0; i < LA; ++i) glds16(p.in + (a_off[i] + tap_off), sbase + (wave + NW * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < LBp; ++i) glds16(p.wt + (b_off[i] + kt * 64), sbase + (NA + wave + NW * i) * 1024);
        c0 += 64;
        if (c0 >= p.cin) { c0 = 0; if (++kw == p.ks) { kw = 0; ++kh; } }
    };

    const int wm = wave / WN, wn = wave % WN;
    floatx4 bv[TN];                                        // bias fetched now, consumed after the k-loop (padded to 128 rows)
#pragma unroll
    for (int u = 0; u < TN; ++u) bv[u] = *(const floatx4 *)(p.bias + n0 + (wn * TN + u) * 16 + q * 4);
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        if (i < nk) issue(i, i);
    int rstage = 0, wstage = DEPTH % NSTAGE;
    STAMP(0);
    {
    for (int kt = 0; kt < nk; ++kt) {
        if (kt == 2) STAMP(10);
        wait_steps<LA + LBp, DEPTH - 1>(min(DEPTH - 1, nk - 1 - kt));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt < 5) STAMP(1 + kt);
        if (kt + DEPTH < nk) issue(kt + DEPTH, wstage);
        if (kt == 2) STAMP(11);
        const unsigned char *sbase = lds + rstage * STAGE;
        rstage = rstage + 1 == NSTAGE ? 0 : rstage + 1;
        wstage = wstage + 1 == NSTAGE ? 0 : wstage + 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ro = kk ? rd_off1 : rd_off0;
            half8 fa[TM], fb[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) fa[t] = *(const half8 *)(sbase + (wm * TM + t) * 2048 + ro);
#pragma unroll
            for (int u = 0; u < TN; ++u) fb[u] = *(const half8 *)(sbase + NA * 1024 + (wn * TN + u) * 2048 + ro);
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[u], fa[t], acc[t][u], 0, 0, 0);
        }
    }
    }

    STAMP(6);
    if constexpr (N2T > 0) {
        epilogue_tail<BM, BN, TM, TN, N2T>(p, acc, bv, lds, wm, wn, [&](int pm, long &o) { return tail_pixel_offset(p, m0 + pm, o); });
        return;
    }
    if constexpr (EPI_FITS) {
        if (p.epi16) {
            epilogue_lds<BM, BN, TM, TN, NW * 64>(p, n0, acc, bv, lds, wm, wn, [&](int pm, long &o, long &rp, long &o2) { return pixel_offsets(p, m0 + pm, o, rp, o2); });
            return;
        }
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        long opix, rpix, opix2;
        if (!pixel_offsets(p, m0 + (wm * TM + t) * 16 + r, opix, rpix, opix2)) continue;
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            int n = n0 + (wn * TN + u) * 16 + q * 4;
            if (n < p.cout) store_tile(p, acc[t][u], bv[u], opix, rpix, n, opix2);
        }
    }
}

// ---------------------------------------------------------------------------------------
// conv3x3_rows: 3x3 / stride-1 convolution with TAP REUSE.  The generic kernels above fetch
// the input tile once per tap -- nine times -- and the global->LDS path (about 16 B/clk/CU of
// LDS-DMA) is what bounds them.  Here the GEMM's M dimension enumerates *padded* pixel
// positions (border columns and rows included, ~5 % junk at 80x80, masked at the store), which
// makes "the pixel one tap to the right" simply the next row of the A tile: one strip of
// BM + 2 consecutive padded pixels, loaded once per (kh, channel chunk), serves kw = 0, 1, 2 by
// reading LDS at row offsets 0, 1, 2.  A "super-step" = one strip + the weights of three taps
// + 3 x (BK/32) x TM x TN MFMAs: ~1.8x fewer DMA bytes at 128x64, ~2.4x at 256x32.
// Two LDS stages (a super-step is long enough to cover one DMA batch), vmcnt(0) + one barrier
// per super-step.  Requires stride 1, input border 1, cin % BK == 0.
// ---------------------------------------------------------------------------------------

template <int BM, int BN, int WM, int WN, bool K64>
__device__ __forceinline__ void conv3x3_rows_body(const ConvArgs &p, const int bx, const int by) {
    constexpr int NW = WM * WN;                     // 4 or 8 waves per workgroup (8: the DMA pieces of a super-step are issued by twice as many waves)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    constexpr int RP = K64 ? 8 : 16;               // rows per DMA piece
    constexpr int RB = K64 ? 128 : 64;             // bytes per row in LDS
    constexpr int BK = K64 ? 64 : 32;
    constexpr int NAS = BM / RP + 1;               // strip pieces (BM + RP rows >= BM + 2)
    constexpr int NBT = BN / RP;                   // weight pieces per tap
    constexpr int NP = NAS + 3 * NBT;
    constexpr int STAGE = NP * 1024;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int TILE_BYTES = 16 * RB;            // LDS bytes of a 16-row operand tile
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
    static_assert(BM * (BN * 2 + 16) <= 2 * STAGE, "the epilogue's fp16 tile must fit the stage buffers");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    // The GEMM rows enumerate padded positions m = (b * Ho + oy) * Wq + x with x in [0, Wq), Wq = in_Wp - 1: the image ROWS THAT HOLD
    // OUTPUTS (the top and bottom border rows are not enumerated) and every column but the right border (round 3: -3.6 / -7.0 / -13.2 % GEMM
    // rows at 80 / 40 / 20 pixels against all (H + 2) x (W + 2) positions).  Output (oy, ox) reads padded input (oy + kh, ox + kw), so
    // position m sits at padded index pos(m) = (b * in_Hp + oy) * in_Wp + x, and the strip gives tap kw the row pos(m + kw): the same
    // as pos(m) + kw inside a row; for the last real column (x = W - 1, kw = 2) it is the LEFT border pixel of the next enumerated row
    // instead of the right border pixel of this one -- both are zeros of the tensor's border (past the last image: clamped to the tensor's
    // last position, a border pixel too).  One junk column (x = W) per row is what tap reuse still costs.
    const int Mp = p.M;                            // = B * Ho * Wq
    int mt, nt;
    xcd_tile((Mp + BM - 1) / BM, (p.cout + BN - 1) / BN, bx, by, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;

    // row R of a piece-structured image: byte offset of k-chunk c (8 halves)
    auto lds_off = [](int R, int c) -> int {
        if (K64) return (R >> 3) * 1024 + (R & 7) * 128 + ((c ^ ((R >> 1) & 7)) << 4);
        return (R >> 4) * 1024 + (R & 15) * 64 + ((c ^ swz16(R)) << 4);
    };
    // DMA lane map
    const int ld_row = K64 ? lane >> 3 : lane >> 2;
    const int ld_slot = K64 ? lane & 7 : lane & 3;
    // fragment read offsets: A row r + kw of a 16-row tile (tile base is a multiple of 16 rows)
    int a_rd[3][K64 ? 2 : 1], b_rd[K64 ? 2 : 1];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int kk = 0; kk < (K64 ? 2 : 1); ++kk) a_rd[kw][kk] = lds_off(r + kw, kk * 4 + q);
#pragma unroll
    for (int kk = 0; kk < (K64 ? 2 : 1); ++kk) b_rd[kk] = lds_off(r, kk * 4 + q);

    // per-wave piece lists: strip pieces wave, wave+4, ...; weight pieces likewise over 3*NBT
    constexpr int LA = (NAS + NW - 1) / NW, LB = (3 * NBT + NW - 1) / NW;
    int a_off[LA], b_off[LB];
    const int last_pix = p.last_pos;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int R = (wave + NW * i) * RP + ld_row;                 // strip row
        int chunk = K64 ? (ld_slot ^ ((R >> 1) & 7)) : (ld_slot ^ swz16(R));
        // pixel index is clamped per tap row at issue time; keep row and chunk parts separate
        a_off[i] = rows_pos(p, m0 + R) | (chunk << 28);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        int pc = wave + NW * i;                                 // 0 .. 3*NBT-1
        int kw = pc / NBT, R = (pc - kw * NBT) * RP + ld_row;   // cout row inside the BN tile
        int chunk = K64 ? (ld_slot ^ ((R >> 1) & 7)) : (ld_slot ^ swz16(R));
        b_off[i] = (n0 + R) * p.kp + kw * p.cin + chunk * 8;
    }

    floatx4 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WN, wn = wave % WN;
    floatx4 bv[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) bv[u] = *(const floatx4 *)(p.bias + n0 + (wn * TN + u) * 16 + q * 4);

    const int cpk = p.cin / BK;                     // channel chunks per tap row
    const int ns = 3 * cpk;                         // super-steps: (kh, chunk)
    int kh = 0, c0 = 0;
    auto issue = [&](int stage) {
        unsigned char *sbase = lds + stage * STAGE;
        const int row_shift = kh * p.in_Wp;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            if (wave + NW * i < NAS) {
                int pix = min((a_off[i] & 0x0FFFFFFF) + row_shift, last_pix);
                glds16(p.in + ((long)pix * p.in_cs + c0 + (a_off[i] >> 28) * 8), sbase + (wave + NW * i) * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < LB; ++i)
            if (wave + NW * i < 3 * NBT) glds16(p.wt + (b_off[i] + kh * 3 * p.cin + c0), sbase + (NAS + wave + NW * i) * 1024);
        c0 += BK;
        if (c0 >= p.cin) { c0 = 0; ++kh; }
    };

    STAMP(0);
    issue(0);
    STAMP(1);
    for (int st = 0; st < ns; ++st) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                   // stage st landed; everyone finished stage st-1
        asm volatile("" ::: "memory");
        if (st < 3) STAMP(2 + st);
        if (st + 1 < ns) issue((st + 1) & 1);
        const unsigned char *sA = lds + (st & 1) * STAGE + wm * TM * TILE_BYTES;
        const unsigned char *sB = lds + (st & 1) * STAGE + NAS * 1024 + wn * TN * TILE_BYTES;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int kk = 0; kk < (K64 ? 2 : 1); ++kk) {
                half8 fa[TM], fb[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[t] = *(const half8 *)(sA + t * TILE_BYTES + a_rd[kw][kk]);
#pragma unroll
                for (int u = 0; u < TN; ++u) fb[u] = *(const half8 *)(sB + kw * NBT * 1024 + u * TILE_BYTES + b_rd[kk]);
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[u], fa[t], acc[t][u], 0, 0, 0);
            }
    }

    STAMP(5);
    // ---- epilogue: padded position -> (b, y, x); border rows/columns are junk ----
    const int HW = p.Ho * p.rows_wq;
    if (p.epi16 > 1) {                                     // measured: the extra LDS round trip costs this MFMA-heavier kernel more than its stores do
        epilogue_lds<BM, BN, TM, TN, NW * 64>(p, n0, acc, bv, lds, wm, wn, [&](int pm, long &opix, long &rpix, long &opix2) {
            const int m = m0 + pm;
            if (m >= Mp) return false;
            const int b = fdiv(m, p.d_hwp), rem = m - b * HW;
            const int oy = fdiv(rem, p.d_wp), ox = rem - oy * p.rows_wq;
            if (oy >= p.Ho || ox >= p.Wo) return false;
            opix = ((long)(b * p.out_Hp + oy + p.out_pad) * p.out_Wp + ox + p.out_pad) * p.out_cs;
            rpix = p.res ? ((long)(b * p.res_Hp + oy + p.res_pad) * p.res_Wp + ox + p.res_pad) * p.res_cs : 0;
            opix2 = upsampled_offset(p, b, oy, ox);
            return true;
        });
        STAMP(6);
        return;
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        int m = m0 + (wm * TM + t) * 16 + r;
        if (m >= Mp) continue;
        int b = fdiv(m, p.d_hwp), rem = m - b * HW;
        int oy = fdiv(rem, p.d_wp), ox = rem - oy * p.rows_wq;
        if (oy >= p.Ho || ox >= p.Wo) continue;
        long opix = ((long)(b * p.out_Hp + oy + p.out_pad) * p.out_Wp + ox + p.out_pad) * p.out_cs;
        long rpix = p.res ? ((long)(b * p.res_Hp + oy + p.res_pad) * p.res_Wp + ox + p.res_pad) * p.res_cs : 0;
        long opix2 = upsampled_offset(p, b, oy, ox);
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            int n = n0 + (wn * TN + u) * 16 + q * 4;
            if (n < p.cout) store_tile(p, acc[t][u], bv[u], opix, rpix, n, opix2);
        }
    }
    STAMP(6);
}

// ---------------------------------------------------------------------------------------
// conv_mfma_wsk: "wave-split-K" for the deep, small-M layers (P5: M = 400 per frame,
// K up to 4608).  The 4 waves of a workgroup all own the SAME BM x BN output tile but take
// every 4th k-step each, stage their operands in wave-private LDS (two stages per wave) and
// therefore never meet at a barrier inside the k-loop; the four partial accumulators are
// summed through LDS at the end and each wave finishes a quarter of the tile.  Four times
// the k-parallelism per CU exactly where a 64-wide tile leaves most CUs waiting on memory.
// ---------------------------------------------------------------------------------------
template <int BM, int BN, bool GENERAL>
__device__ __forceinline__ void conv_mfma_wsk_body(const ConvArgs &p, const int bx, const int by) {
    constexpr int NA = BM / 16, NB = BN / 16, NP = NA + NB;
    constexpr int WSTAGE = NP * 1024;
    constexpr int TM = NA, TN = NB;
    constexpr int STAGING = 4 * 2 * WSTAGE, REDUCE = 4 * TM * TN * 1024;
    constexpr int LDS_BYTES = STAGING > REDUCE ? STAGING : REDUCE;
    static_assert((TM * TN) % 4 == 0, "tile count must split over 4 waves");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const LaneMap lm(lane);
    int mt, nt;
    xcd_tile((p.M + BM - 1) / BM, (p.cout + BN - 1) / BN, bx, by, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;
    unsigned char *wbase = lds + wave * 2 * WSTAGE;

    int a_off[NA], b_off[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) a_off[i] = input_offset(p, m0 + i * 16 + lm.ld_row);
#pragma unroll
    for (int i = 0; i < NB; ++i) b_off[i] = (n0 + i * 16 + lm.ld_row) * p.kp + lm.ld_chunk * 8;

    floatx4 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.kp / 32;
    const int my_nk = (nk - wave + 3) / 4;               // k-steps wave, wave+4, ...
    KPos kp;
    KLane kl;
    if (GENERAL) kl.advance(lm.ld_chunk * 8 + 32 * wave, p.cin); else kp.advance(wave, p.cin, p.ks);

    auto issue = [&](int kt, int stage) {
        unsigned char *sbase = wbase + stage * WSTAGE;
        const int tap_off = GENERAL ? kl.offset(p.ks, p.in_Wp, p.in_cs) : (kp.kh * p.in_Wp + kp.kw) * p.in_cs + kp.c0 + lm.ld_chunk * 8;
#pragma unroll
        for (int i = 0; i < NA; ++i) glds16(p.in + (a_off[i] + tap_off), sbase + i * 1024);
#pragma unroll
        for (int i = 0; i < NB; ++i) glds16(p.wt + (b_off[i] + kt * 32), sbase + (NA + i) * 1024);
        if (GENERAL) kl.advance(128, p.cin); else kp.advance(4, p.cin, p.ks);
    };

    if (my_nk > 0) issue(wave, 0);
    for (int it = 0; it < my_nk; ++it) {
        if (it + 1 < my_nk) { issue(wave + 4 * (it + 1), (it + 1) & 1); wait_vmcnt<NP>(); } else { wait_vmcnt<0>(); }
        const unsigned char *sbase = wbase + (it & 1) * WSTAGE;
        half8 fa[TM], fb[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) fa[t] = *(const half8 *)(sbase + t * 1024 + lm.rd_off);
#pragma unroll
        for (int u = 0; u < TN; ++u) fb[u] = *(const half8 *)(sbase + (NA + u) * 1024 + lm.rd_off);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u)
                acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[u], fa[t], acc[t][u], 0, 0, 0);
        // the next issue() overwrites the stage just read: its LDS reads must have retired
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // ---- cross-wave reduction through LDS: partial[wave][tile][lane] (16 B per lane, lane-linear) ----
    __syncthreads();                                       // all staging reads done before the region is reused
    floatx4 *part = (floatx4 *)lds;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) part[(wave * TM * TN + t * TN + u) * 64 + lane] = acc[t][u];
    __syncthreads();
    constexpr int PER = TM * TN / 4;                       // tiles finished by each wave
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int tile = wave * PER + k, t = tile / TN, u = tile % TN;
        floatx4 v = part[(0 * TM * TN + tile) * 64 + lane] + part[(1 * TM * TN + tile) * 64 + lane] +
                    part[(2 * TM * TN + tile) * 64 + lane] + part[(3 * TM * TN + tile) * 64 + lane];
        long opix, rpix, opix2;
        if (!pixel_offsets(p, m0 + t * 16 + r, opix, rpix, opix2)) continue;
        int n = n0 + u * 16 + q * 4;
        if (n < p.cout) store_tile(p, v, *(const floatx4 *)(p.bias + n), opix, rpix, n, opix2);
    }
}

// ---------------------------------------------------------------------------------------
// conv_mfma64_pt: PERSISTENT-TILE variant of conv_mfma64_w8 (8 waves, 64-deep k-steps, BM = 128).  A workgroup owns one
// cout slice and walks over pixel tiles g, g + G, ...; the (tile, k-step) pairs form ONE stream of steps through the
// stage ring, so the first operands of the next tile are already in flight while this tile's epilogue runs -- in the plain
// tile kernels every workgroup starts with an empty ring (phase stamps: ~2 600 clk before the first MFMA of a tile, 15-35 %
// of a workgroup's life on the short-K layers).  The epilogue stores straight from the accumulators (the lane-pair
// exchange across lane pairs: 16-byte stores, no LDS round trip), so it does not need the stage buffers the prefetch is using.
// Two stages = 64 KiB at BN = 128: two workgroups per CU, as in the plain kernel.  Runs full tiles only (M % 128 == 0,
// cout % BN == 0), any kernel size / stride, no residual / second destination / half-resolution source.
// Same MFMA order per output as conv_mfma64 (k ascending), same swizzled piece layout.
// ---------------------------------------------------------------------------------------
template <int BN, int NSTAGE, int BM = 128, int NW = 8>
__global__ __launch_bounds__(NW * 64) void conv_mfma64_pt(ConvArgs p, int groups) {
    constexpr int WM = NW / 2, WN = 2;                      // 8 waves on 128 x BN (two workgroups per CU) or 16 waves on 256 x BN (one, three stages)
    static_assert(BM % (WM * 16) == 0 && (BM / 8) % NW == 0, "pixel tile must split over the waves");
    constexpr int NA = BM / 8, NB = BN / 8, LA = NA / NW, LBp = NB / NW, L = LA + LBp;
    constexpr int DEPTH = NSTAGE - 1;
    constexpr int STAGE = (NA + NB) * 1024;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    static_assert(TN % 2 == 0, "the epilogue pairs neighbouring cout tiles");
    static_assert(NB % NW == 0, "weight pieces must split evenly over the waves");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTAGE * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.y * BN, g = blockIdx.x;
    const int nk = p.kp / 64;
    const int n_mt = p.M / BM;
    // Which pixel tiles: workgroups go round-robin over the 8 XCDs, and XCD x works on the x-th contiguous eighth of the tile
    // list (as in xcd_tile: neighbouring tiles of a 3x3 conv share input rows, which should meet in ONE L2); inside that run
    // the XCD's workgroups (rank j of Gx) take tiles j, j + Gx, ... -- at any time the XCD is busy with Gx consecutive tiles.
    const PtRun run = pt_run(g, blockIdx.y, groups, n_mt);       // tile_math.h (checked exhaustively on the host)
    const int j = run.j, Gx = run.Gx, run0 = run.run0, run_n = run.run_n;
    const int n_my = j < run_n ? (run_n - j + Gx - 1) / Gx : 0;
    const int S = n_my * nk;                                // steps of this workgroup
    auto tile_m0 = [&](int ti) { return (run0 + j + ti * Gx) * BM; };
    const int ld_row8 = lane >> 3, ld_slot = lane & 7;
    const int rd_base = (r >> 3) * 1024 + (r & 7) * 128;
    const int rd_off0 = rd_base + (((0 + q) ^ ((r >> 1) & 7)) << 4);
    const int rd_off1 = rd_base + (((4 + q) ^ ((r >> 1) & 7)) << 4);

    int a_off[LA], a_lo[LA], b_off[LBp];
#pragma unroll
    for (int i = 0; i < LBp; ++i) {
        const int row = (wave + NW * i) * 8 + ld_row8;
        b_off[i] = (n0 + row) * p.kp + ((ld_slot ^ ((row >> 1) & 7)) << 3);
    }
    int is_tile = 0, is_kt = 0, is_stage = 0, kh = 0, kw = 0, c0 = 0;      // the next step to issue
    auto issue = [&]() {
        if (is_kt == 0) {
            const int m0 = tile_m0(is_tile);
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                const int row = (wave + NW * i) * 8 + ld_row8;
                a_off[i] = input_offset(p, m0 + row) + ((ld_slot ^ ((row >> 1) & 7)) << 3);
                a_lo[i] = p.in2 ? input_offset_lo(p, m0 + row) + ((ld_slot ^ ((row >> 1) & 7)) << 3) : 0;
            }
            kh = kw = c0 = 0;
        }
        unsigned char *sbase = lds + is_stage * STAGE;
        const int tap_off = (kh * p.in_Wp + kw) * p.in_cs + c0;
        if (p.in2 && c0 < p.split) {                       // the upsampled half of a neck concat, read where it was produced
#pragma unroll
            for (int i = 0; i < LA; ++i) glds16(p.in2 + (a_lo[i] + c0), sbase + (wave + NW * i) * 1024);
        } else {
#pragma unroll
            for (int i = 0; i < LA; ++i) glds16(p.in + (a_off[i] + tap_off), sbase + (wave + NW * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < LBp; ++i) glds16(p.wt + (b_off[i] + is_kt * 64), sbase + (NA + wave + NW * i) * 1024);
        c0 += 64;
        if (c0 >= p.cin) { c0 = 0; if (++kw == p.ks) { kw = 0; ++kh; } }
        if (++is_kt == nk) { is_kt = 0; ++is_tile; }
        is_stage = is_stage + 1 == NSTAGE ? 0 : is_stage + 1;
    };
    const int wm = wave / WN, wn = wave % WN;
    const int qe = q & ~1, odd = q & 1;
    floatx4 bv[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) bv[u] = *(const floatx4 *)(p.bias + n0 + (wn * TN + u) * 16 + q * 4);
    floatx4 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = floatx4{0.f, 0.f, 0.f, 0.f};
    STAMP(0);
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        if (i < S) issue();
    STAMP(1);
    int kt = 0, tile = 0, rstage = 0;
    // vmcnt counts the epilogue's global stores together with the LDS-DMAs, in issue order: for the DEPTH steps
    // after a tile's epilogue its stores are younger than the stage being waited for and may stay outstanding
    constexpr int NST = TM * TN / 2;
    int store_credit = 0;
    for (int s = 0; s < S; ++s) {
        const int ahead = min(DEPTH - 1, S - 1 - s);
        if (store_credit > 0) { wait_steps_plus<L, DEPTH - 1, NST>(ahead); --store_credit; }
        else wait_steps<L, DEPTH - 1>(ahead);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s < 4) STAMP(2 + s);
        // residual (Bottleneck shortcut): this tile's values are requested at the start of its LAST k-step, before the next
        // DMA pieces, so they are OLDER than everything the epilogue leaves in flight (vmcnt retires in issue order)
        half4 rv[TM][TN];
        if (p.res && kt == nk - 1) {
            const int m0r = tile_m0(tile);
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                long opix, rpix, opix2;
                pixel_offsets(p, m0r + (wm * TM + t) * 16 + r, opix, rpix, opix2);
#pragma unroll
                for (int u = 0; u < TN; ++u) rv[t][u] = *(const half4 *)(p.res + rpix + n0 + (wn * TN + u) * 16 + q * 4);
            }
        }
        const bool issued = s + DEPTH < S;
        if (issued) issue();                                // refills the stage read in step s - 1
        const unsigned char *sbase = lds + rstage * STAGE;
        rstage = rstage + 1 == NSTAGE ? 0 : rstage + 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ro = kk ? rd_off1 : rd_off0;
            half8 fa[TM], fb[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) fa[t] = *(const half8 *)(sbase + (wm * TM + t) * 2048 + ro);
#pragma unroll
            for (int u = 0; u < TN; ++u) fb[u] = *(const half8 *)(sbase + NA * 1024 + (wn * TN + u) * 2048 + ro);
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[u], fa[t], acc[t][u], 0, 0, 0);
        }
        if (++kt == nk) {                                   // the tile is complete: epilogue straight from the accumulators
            kt = 0;
            const int m0 = tile_m0(tile);
            ++tile;
            if (tile == 1) STAMP(6);
            prio_epilogue(p.epi_prio);
            if (p.res) {                                    // the residual loads have landed; the DMA pieces issued after them may stay in flight
                if (issued) wait_vmcnt<L>(); else wait_vmcnt<0>();
            }
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                long opix, rpix, opix2;
                pixel_offsets(p, m0 + (wm * TM + t) * 16 + r, opix, rpix, opix2);      // (every tile is full: always live)
#pragma unroll
                for (int u = 0; u < TN; u += 2) {
                    floatx4 v0 = acc[t][u] + bv[u], v1 = acc[t][u + 1] + bv[u + 1];
                    if (p.act) {
                        silu4(v0);
                        silu4(v1);
                    }
                    if (p.res) {
                        const half4 r0 = rv[t][u], r1 = rv[t][u + 1];
                        v0[0] += (float)r0[0]; v0[1] += (float)r0[1]; v0[2] += (float)r0[2]; v0[3] += (float)r0[3];
                        v1[0] += (float)r1[0]; v1[1] += (float)r1[1]; v1[2] += (float)r1[2]; v1[3] += (float)r1[3];
                    }
                    const half4 h0 = {(f16)v0[0], (f16)v0[1], (f16)v0[2], (f16)v0[3]}, h1 = {(f16)v1[0], (f16)v1[1], (f16)v1[2], (f16)v1[3]};
                    const half4 give = odd ? h0 : h1, keep = odd ? h1 : h0;
                    u32x2 gw = __builtin_bit_cast(u32x2, give);
                    gw[0] = (unsigned)__shfl_xor((int)gw[0], 16);
                    gw[1] = (unsigned)__shfl_xor((int)gw[1], 16);
                    const half4 got = __builtin_bit_cast(half4, gw);
                    const half4 lo = odd ? got : keep, hi = odd ? keep : got;
                    const half8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const int n = n0 + (wn * TN + u + odd) * 16 + qe * 4;
                    store16(p.out, opix + n, o, p.wthru);
                    acc[t][u] = floatx4{0.f, 0.f, 0.f, 0.f};
                    acc[t][u + 1] = floatx4{0.f, 0.f, 0.f, 0.f};
                }
            }
            store_credit = DEPTH;
            prio_main(p.epi_prio);
            if (tile == 1) STAMP(7);
        }
    }
    STAMP(9);
}


template <int BM, int BN, int WM, int WN, int NSTAGE, bool GENERAL>
__global__ __launch_bounds__(256) void conv_mfma(ConvArgs p) { conv_mfma_body<BM, BN, WM, WN, NSTAGE, GENERAL>(p, blockIdx.x, blockIdx.y); }
template <int BM, int BN, int WM, int WN, int NSTAGE, bool GENERAL>
__global__ __launch_bounds__(256) void conv_mfma_grp(ConvGroupArgs g) {
    int bx, by;
    const int z = group_pick<BM, BN>(g, bx, by);
    if (z < 0) return;
    conv_mfma_body<BM, BN, WM, WN, NSTAGE, GENERAL>(g.p[z], bx, by);
}
template <int BM, int BN, int WM, int WN, int NSTAGE, int N2T>
__global__ __launch_bounds__(256) void conv_mfma_tail(ConvArgs p) { conv_mfma_body<BM, BN, WM, WN, NSTAGE, false, N2T>(p, blockIdx.x, blockIdx.y); }
template <int BM, int BN, int WM, int WN, int NSTAGE, int N2T>
__global__ __launch_bounds__(256) void conv_mfma64_tail(ConvArgs p) { conv_mfma64_body<BM, BN, WM, WN, NSTAGE, N2T>(p, blockIdx.x, blockIdx.y); }
template <int BM, int BN, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(512) void conv_mfma64_w8(ConvArgs p) { conv_mfma64_body<BM, BN, WM, WN, NSTAGE>(p, blockIdx.x, blockIdx.y); }
template <int BM, int BN, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(256) void conv_mfma64(ConvArgs p) { conv_mfma64_body<BM, BN, WM, WN, NSTAGE>(p, blockIdx.x, blockIdx.y); }
template <int BM, int BN, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(256) void conv_mfma64_grp(ConvGroupArgs g) {
    int bx, by;
    const int z = group_pick<BM, BN>(g, bx, by);
    if (z < 0) return;
    conv_mfma64_body<BM, BN, WM, WN, NSTAGE>(g.p[z], bx, by);
}
template <int BM, int BN, int WM, int WN, bool K64>
__global__ __launch_bounds__(512) void conv3x3_rows_w8(ConvArgs p) { conv3x3_rows_body<BM, BN, WM, WN, K64>(p, blockIdx.x, blockIdx.y); }
template <int BM, int BN, int WM, int WN, bool K64>
__global__ __launch_bounds__(512) void conv3x3_rows_w8_grp(ConvGroupArgs g) {
    int bx, by;
    const int z = group_pick<BM, BN>(g, bx, by);
    if (z < 0) return;
    conv3x3_rows_body<BM, BN, WM, WN, K64>(g.p[z], bx, by);
}
template <int BM, int BN, int WM, int WN, bool K64>
__global__ __launch_bounds__(256) void conv3x3_rows(ConvArgs p) { conv3x3_rows_body<BM, BN, WM, WN, K64>(p, blockIdx.x, blockIdx.y); }
template <int BM, int BN, int WM, int WN, bool K64>
__global__ __launch_bounds__(256) void conv3x3_rows_grp(ConvGroupArgs g) {
    int bx, by;
    const int z = group_pick<BM, BN>(g, bx, by);
    if (z < 0) return;
    conv3x3_rows_body<BM, BN, WM, WN, K64>(g.p[z], bx, by);
}
template <int BM, int BN, bool GENERAL>
__global__ __launch_bounds__(256) void conv_mfma_wsk(ConvArgs p) { conv_mfma_wsk_body<BM, BN, GENERAL>(p, blockIdx.x, blockIdx.y); }
template <int BM, int BN, bool GENERAL>
__global__ __launch_bounds__(256) void conv_mfma_wsk_grp(ConvGroupArgs g) {
    int bx, by;
    const int z = group_pick<BM, BN>(g, bx, by);
    if (z < 0) return;
    conv_mfma_wsk_body<BM, BN, GENERAL>(g.p[z], bx, by);
}

const char *tile_name(int tile) {
    static const char *names[TILE_COUNT] = {"128x128s3", "128x64s3", "64x64s3", "256x32s3", "64x128s3", "wsk64x64", "wsk32x64", "wsk64x32",
                                            "k64:64x64s3", "rows:128x32", "rows64:64x64", "tail:128x64", "tail:k64:128x128",
                                            "k64:128x128s2/8w", "k64:128x128s3/8w", "k64:128x64s3/8w", "k64:256x64s2/8w", "rows:256x64/8w",
                                            "pt:128x128s2", "pt:128x64s2", "pp:256x128", "pp:256x64", "pp:256x192", "pp:512x64", "ppt:256x128"};
    return tile >= 0 && tile < TILE_COUNT ? names[tile] : "?";
}

bool tile_is_w8(int tile) { return tile >= TILE_K64_128x128_S2_W8 && tile <= TILE_K64_256x64_S2_W8; }
bool tile_is_pt(int tile) { return tile == TILE_PT_128x128_S2 || tile == TILE_PT_128x64_S2; }
bool tile_is_tail(int tile) { return tile == TILE_TAIL_128x64 || tile == TILE_TAIL_K64_128x128; }
bool tile_is_pp(int tile) { return tile >= TILE_PP_256x128 && tile <= TILE_PP_512x64; }
bool tile_is_ppt(int tile) { return tile == TILE_PPT_256x128; }
bool tile_is_rows(int tile) { return tile == TILE_ROWS_128x32 || tile == TILE_ROWS_K64_64x64 || tile == TILE_ROWS_256x64_W8 || tile_is_pp(tile); }
bool tile_needs_cin64(int tile) {
    return tile == TILE_K64_64x64_S3 || tile == TILE_ROWS_K64_64x64 || tile == TILE_TAIL_K64_128x128 || tile_is_w8(tile) || tile_is_pt(tile) || tile_is_pp(tile) || tile_is_ppt(tile);
}
// the 64-deep tile kernels (conv_mfma64_body, its persistent form, conv_tile_pp) know how to read channels [0, lo_c) from a half-resolution tensor
bool tile_reads_lo(int tile) { return tile == TILE_K64_64x64_S3 || tile_is_w8(tile) || tile_is_pt(tile) || tile_is_ppt(tile); }

TileShape tile_shape(int tile) {
    switch (tile) {
        case TILE_128x128: return {128, 128};
        case TILE_128x64: return {128, 64};
        case TILE_64x64: return {64, 64};
        case TILE_256x32: return {256, 32};
        case TILE_64x128: return {64, 128};
        case TILE_WSK_64x64: return {64, 64};
        case TILE_WSK_32x64: return {32, 64};
        case TILE_WSK_64x32: return {64, 32};
        case TILE_K64_64x64_S3: return {64, 64};
        case TILE_ROWS_128x32: return {128, 32};
        case TILE_ROWS_K64_64x64: return {64, 64};
        case TILE_TAIL_128x64: return {128, 64};
        case TILE_TAIL_K64_128x128: return {128, 128};
        case TILE_K64_128x128_S2_W8: case TILE_K64_128x128_S3_W8: return {128, 128};
        case TILE_K64_128x64_S3_W8: return {128, 64};
        case TILE_K64_256x64_S2_W8: return {256, 64};
        case TILE_ROWS_256x64_W8: return {256, 64};
        case TILE_PT_128x128_S2: return {128, 128};
        case TILE_PT_128x64_S2: return {128, 64};
        case TILE_PP_256x128: return {256, 128};
        case TILE_PP_256x64: return {256, 64};
        case TILE_PP_256x192: return {256, 192};
        case TILE_PP_512x64: return {512, 64};
        case TILE_PPT_256x128: return {256, 128};
    }
    return {0, 0};
}

struct LaunchPlan {
    const ConvArgs *a; int n; bool general;
    dim3 grid(int bm, int bn) const { return dim3(cdiv(a[0].M, bm), cdiv(a[0].cout, bn), 1); }      // single problem
    dim3 group(ConvGroupArgs &g, int bm, int bn) const {
        g.n = n; g.start[0] = 0;
        for (int i = 0; i < n; ++i) {
            g.p[i] = a[i];
            g.gx[i] = cdiv(a[i].M, bm);
            g.start[i + 1] = (int)align_up((size_t)g.start[i] + (size_t)g.gx[i] * cdiv(a[i].cout, bn), 8);
        }
        for (int i = n; i < MAX_GROUP; ++i) { g.start[i + 1] = g.start[n]; g.gx[i] = 1; }
        return dim3(g.start[n], 1, 1);
    }
};

template <int BM, int BN, int WM, int WN, int NSTAGE>
static void launch_tile(const LaunchPlan &l, hipStream_t s) {
    dim3 grid = l.grid(BM, BN);
    if (l.n == 1) {
        if (l.general) hipLaunchKernelGGL((conv_mfma<BM, BN, WM, WN, NSTAGE, true>), grid, dim3(256), 0, s, l.a[0]);
        else hipLaunchKernelGGL((conv_mfma<BM, BN, WM, WN, NSTAGE, false>), grid, dim3(256), 0, s, l.a[0]);
    } else {
        ConvGroupArgs g;
        grid = l.group(g, BM, BN);
        if (l.general) hipLaunchKernelGGL((conv_mfma_grp<BM, BN, WM, WN, NSTAGE, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((conv_mfma_grp<BM, BN, WM, WN, NSTAGE, false>), grid, dim3(256), 0, s, g);
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
static void launch_k64(const LaunchPlan &l, hipStream_t s) {
    dim3 grid = l.grid(BM, BN);
    if (l.n == 1) {
        hipLaunchKernelGGL((conv_mfma64<BM, BN, WM, WN, NSTAGE>), grid, dim3(256), 0, s, l.a[0]);
    } else {
        ConvGroupArgs g;
        grid = l.group(g, BM, BN);
        hipLaunchKernelGGL((conv_mfma64_grp<BM, BN, WM, WN, NSTAGE>), grid, dim3(256), 0, s, g);
    }
}

template <int BM, int BN, int WM, int WN, bool K64>
static void launch_rows_w8(const LaunchPlan &l, hipStream_t s) {
    dim3 grid = l.grid(BM, BN);
    if (l.n == 1) {
        hipLaunchKernelGGL((conv3x3_rows_w8<BM, BN, WM, WN, K64>), grid, dim3(512), 0, s, l.a[0]);
    } else {
        ConvGroupArgs g;
        grid = l.group(g, BM, BN);
        hipLaunchKernelGGL((conv3x3_rows_w8_grp<BM, BN, WM, WN, K64>), grid, dim3(512), 0, s, g);
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
static int launch_k64_w8(const LaunchPlan &l, hipStream_t s) {
    RT_CHECK(l.n == 1, RTMODT_E_INVALID, "launch_conv: the 8-wave tiles run single problems");
    hipLaunchKernelGGL((conv_mfma64_w8<BM, BN, WM, WN, NSTAGE>), l.grid(BM, BN), dim3(512), 0, s, l.a[0]);
    return RTMODT_OK;
}

template <int BN, int NSTAGE, int BM = 128, int NW = 8>
static int launch_pt(const LaunchPlan &l, hipStream_t s) {
    const ConvArgs &a = l.a[0];
    RT_CHECK(l.n == 1 && !l.general && a.cin % 64 == 0 && a.kp % 64 == 0 && a.M % BM == 0 && a.cout % BN == 0 && !a.out2 && a.epi16,
             RTMODT_E_INVALID, "launch_conv: the persistent tile runs one conv with cin %% 64 == 0, full tiles (M %% %d == 0, cout %% BN == 0), no second destination", BM);
    const int slices = a.cout / BN, n_mt = a.M / BM;
    constexpr int per_cu_lds = (160 * 1024) / (NSTAGE * (BM / 8 + BN / 8) * 1024);      // workgroups of this kernel that fit one CU's LDS
    const int per_cu = per_cu_lds;
    const int groups = std::max(1, std::min(n_mt, per_cu * device_cus() / slices));
    hipLaunchKernelGGL((conv_mfma64_pt<BN, NSTAGE, BM, NW>), dim3(groups, slices), dim3(NW * 64), 0, s, a, groups);
    return RTMODT_OK;
}

template <int BM, int BN, int WM, int WN, bool K64>
static void launch_rows(const LaunchPlan &l, hipStream_t s) {
    dim3 grid = l.grid(BM, BN);
    if (l.n == 1) {
        hipLaunchKernelGGL((conv3x3_rows<BM, BN, WM, WN, K64>), grid, dim3(256), 0, s, l.a[0]);
    } else {
        ConvGroupArgs g;
        grid = l.group(g, BM, BN);
        hipLaunchKernelGGL((conv3x3_rows_grp<BM, BN, WM, WN, K64>), grid, dim3(256), 0, s, g);
    }
}

template <int BM, int BN>
static void launch_wsk(const LaunchPlan &l, hipStream_t s) {
    dim3 grid = l.grid(BM, BN);
    if (l.n == 1) {
        if (l.general) hipLaunchKernelGGL((conv_mfma_wsk<BM, BN, true>), grid, dim3(256), 0, s, l.a[0]);
        else hipLaunchKernelGGL((conv_mfma_wsk<BM, BN, false>), grid, dim3(256), 0, s, l.a[0]);
    } else {
        ConvGroupArgs g;
        grid = l.group(g, BM, BN);
        if (l.general) hipLaunchKernelGGL((conv_mfma_wsk_grp<BM, BN, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((conv_mfma_wsk_grp<BM, BN, false>), grid, dim3(256), 0, s, g);
    }
}

static int make_args(const ConvLaunch &c, ConvArgs &a) {
    RT_CHECK(c.in.base && c.out.base && c.wt && c.bias, RTMODT_E_INVALID, "launch_conv: null operand");
    // a view that still holds an arena OFFSET instead of a device address (engine.hip rebases them once) must never reach a kernel
    for (const TensorView *v : {&c.in, &c.out, &c.res, &c.out2, &c.tail_out, &c.in_lo})
        RT_CHECK(!v->base || (uintptr_t)v->base >= (1ull << 32), RTMODT_E_INVALID, "launch_conv: view base %p is not a device address", (void *)v->base);
    RT_CHECK(c.ks == 1 || c.ks == 3, RTMODT_E_INVALID, "launch_conv: kernel size %d", c.ks);
    RT_CHECK(c.cin % 8 == 0 && c.cout % 4 == 0, RTMODT_E_INVALID, "launch_conv: cin %d / cout %d granularity", c.cin, c.cout);
    RT_CHECK(c.in.pad >= c.ks / 2, RTMODT_E_INVALID, "launch_conv: input border %d < %d", c.in.pad, c.ks / 2);
    RT_CHECK(c.in.c == c.cin && c.out.c == c.cout, RTMODT_E_INVALID, "launch_conv: view/channel mismatch");
    RT_CHECK(c.in.coff % 8 == 0 && c.out.coff % 4 == 0 && c.in.C % 8 == 0 && c.out.C % 4 == 0, RTMODT_E_INVALID,
             "launch_conv: slice alignment");
    a.in = c.in.base + c.in.coff;
    a.wt = c.wt;
    a.bias = c.bias;
    a.out = c.out.base + c.out.coff;
    a.res = c.res.base ? c.res.base + c.res.coff : nullptr;
    a.in_Hp = c.in.H + 2 * c.in.pad; a.in_Wp = c.in.padded_w(); a.in_cs = c.in.C; a.in_org = c.in.pad - c.ks / 2;
    a.Ho = (c.in.H + 2 * (c.ks / 2) - c.ks) / c.stride + 1;
    a.Wo = (c.in.W + 2 * (c.ks / 2) - c.ks) / c.stride + 1;
    RT_CHECK(a.Ho == c.out.H && a.Wo == c.out.W, RTMODT_E_INVALID, "launch_conv: output %dx%d != %dx%d", c.out.H, c.out.W, a.Ho, a.Wo);
    a.out_Hp = c.out.H + 2 * c.out.pad; a.out_Wp = c.out.padded_w(); a.out_cs = c.out.C; a.out_pad = c.out.pad;
    a.res_Hp = c.res.H + 2 * c.res.pad; a.res_Wp = c.res.W + 2 * c.res.pad; a.res_cs = c.res.C; a.res_pad = c.res.pad;
    if (a.res) {
        RT_CHECK(c.res.H == c.out.H && c.res.W == c.out.W && c.res.c == c.cout && c.res.coff % 4 == 0 && c.res.C % 4 == 0,
                 RTMODT_E_INVALID, "launch_conv: residual shape");
    }
    a.out2 = nullptr; a.out2_Hp = a.out2_Wp = a.out2_cs = a.out2_pad = 0;
    if (c.out2.base) {
        RT_CHECK(c.out2.H == 2 * c.out.H && c.out2.W == 2 * c.out.W && c.out2.c == c.cout && c.out2.coff % 4 == 0 && c.out2.C % 4 == 0,
                 RTMODT_E_INVALID, "launch_conv: upsampled destination shape");
        a.out2 = c.out2.base + c.out2.coff;
        a.out2_Hp = c.out2.H + 2 * c.out2.pad; a.out2_Wp = c.out2.W + 2 * c.out2.pad; a.out2_cs = c.out2.C; a.out2_pad = c.out2.pad;
    }
    a.in2 = nullptr; a.in2_Hp = a.in2_Wp = a.in2_cs = a.in2_pad = a.split = 0;
    if (c.in_lo.base) {
        RT_CHECK(c.ks == 1 && c.stride == 1 && c.in_lo.H * 2 == c.in.H && c.in_lo.W * 2 == c.in.W && c.in_lo.c == c.lo_c && c.lo_c % 64 == 0 && c.lo_c > 0 &&
                     c.lo_c < c.cin && c.cin % 64 == 0 && c.in_lo.coff % 8 == 0 && c.in_lo.C % 8 == 0,
                 RTMODT_E_INVALID, "launch_conv: half-resolution source shape");
        RT_CHECK((long)c.B * (c.in_lo.H + 2 * c.in_lo.pad) * (c.in_lo.W + 2 * c.in_lo.pad) * c.in_lo.C < (1L << 31), RTMODT_E_INVALID, "launch_conv: tensor exceeds 2^31 elements");
        a.in2 = c.in_lo.base + c.in_lo.coff;
        a.in2_Hp = c.in_lo.H + 2 * c.in_lo.pad; a.in2_Wp = c.in_lo.W + 2 * c.in_lo.pad; a.in2_cs = c.in_lo.C; a.in2_pad = c.in_lo.pad;
        a.split = c.lo_c;
    }
    a.M = c.B * a.Ho * a.Wo;
    a.d_howo = make_fastdiv(a.Ho * a.Wo); a.d_wo = make_fastdiv(a.Wo);
    a.d_hwp = make_fastdiv(a.in_Hp * a.in_Wp); a.d_wp = make_fastdiv(a.in_Wp);      // (d_hwp is re-made for the tap-reuse tiles: launch_conv_group)
    a.last_pos = 0; a.rows_wq = a.in_Wp;
    a.cin = c.cin; a.cout = c.cout; a.ks = c.ks; a.stride = c.stride; a.act = c.act;
    a.K = c.ks * c.ks * c.cin;
    a.kp = c.kp;
    a.t_wt = c.tail_wt; a.t_bias = c.tail_bias; a.t_out = c.tail_wt ? c.tail_out.base + c.tail_out.coff : nullptr;
    a.t_cout = c.tail_cout; a.t_kp = c.tail_kp; a.t_act = c.tail_act;
    a.t_out_Hp = c.tail_out.H + 2 * c.tail_out.pad; a.t_out_Wp = c.tail_out.padded_w(); a.t_out_cs = c.tail_out.C; a.t_out_pad = c.tail_out.pad;
    static const int wt_env = rt_diag("WT") ? atoi(rt_diag("WT")) : 0;
    a.wthru = wt_env;
    static const int prio_env = rt_diag("EPI_PRIO") ? atoi(rt_diag("EPI_PRIO")) : 0;      // experiment hook
    a.epi_prio = prio_env;
    a.epi16 = c.out.coff % 8 == 0 && c.out.C % 8 == 0 && (!c.out2.base || (c.out2.coff % 8 == 0 && c.out2.C % 8 == 0)) &&
              (!c.res.base || (c.res.coff % 4 == 0)) ? c.epilogue : 0;
    RT_CHECK(a.kp % 32 == 0 && a.kp >= a.K, RTMODT_E_INVALID, "launch_conv: kp %d for K %d", a.kp, a.K);
    // 32-bit element offsets inside the kernel
    RT_CHECK((long)c.B * a.in_Hp * a.in_Wp * a.in_cs < (1L << 31) && (long)c.B * a.out_Hp * a.out_Wp * a.out_cs < (1L << 31),
             RTMODT_E_INVALID, "launch_conv: tensor exceeds 2^31 elements");
    return RTMODT_OK;
}

// the ping-pong kernels' 24-bit index bound (tile_math.h: pp_index_fits) for a launch description -- the same quantities make_args / launch_conv_group
// hand to conv_pp.hip's own check, without needing device pointers: M is the largest enumeration either kernel uses (tap reuse: Ho x (in_Wp - 1) positions)
// ... and the shortcut's geometry: the 3x3 ping-pong kernel reads the shortcut at the OUTPUT's element offsets (conv_pp.hip works a tile's store offsets out once, in front of the
// k-loop): a channel slice of a tensor with the output's rows, pad and channel stride -- what a C2f block's concat buffer gives it -- is all it takes.
bool conv_pp_index_fits(const ConvLaunch &c) {
    if (c.res.base && (c.res.H != c.out.H || c.res.W != c.out.W || c.res.pad != c.out.pad || c.res.padded_w() != c.out.padded_w() || c.res.C != c.out.C)) return false;
    const long out_hwc = (long)(c.out.H + 2 * c.out.pad) * c.out.padded_w() * c.out.C;
    const long res_hwc = c.res.base ? (long)(c.res.H + 2 * c.res.pad) * c.res.padded_w() * c.res.C : 0L;
    const long in_wp = c.in.padded_w();
    return pp_index_fits(out_hwc, res_hwc, (long)c.out.H * in_wp, (long)c.B * c.out.H * std::max(in_wp, (long)c.out.W));
}

// n independent convolutions in ONE launch with tile configuration `tile`
int launch_conv_group(const ConvLaunch *c, int n, int tile, hipStream_t s) {
    RT_CHECK(n >= 1 && n <= MAX_GROUP, RTMODT_E_INVALID, "launch_conv_group: %d problems", n);
    ConvArgs a[MAX_GROUP];
    bool general = false;
    for (int i = 0; i < n; ++i) {
        RT_TRY(make_args(c[i], a[i]));
        general = general || (c[i].cin % 32) != 0;
        if (tile_needs_cin64(tile))
            RT_CHECK(c[i].cin % 64 == 0 && a[i].kp % 64 == 0, RTMODT_E_INVALID, "launch_conv: tile %s needs cin %% 64 == 0 (cin %d)", tile_name(tile), c[i].cin);
    }
    for (int i = 0; i < n; ++i)
        RT_CHECK(!c[i].in_lo.base || tile_reads_lo(tile), RTMODT_E_INVALID, "launch_conv: tile %s cannot read a half-resolution source", tile_name(tile));
    for (int i = 0; i < n; ++i)                            // weight rows and bias entries exist up to cout rounded up to 128
        RT_CHECK(tile_shape(tile).bn <= 128 || c[i].cout % tile_shape(tile).bn == 0, RTMODT_E_INVALID, "launch_conv: tile %s needs cout %% %d == 0 (cout %d)", tile_name(tile), tile_shape(tile).bn, c[i].cout);
    if (tile_is_rows(tile)) {
        const int bk = tile_needs_cin64(tile) ? 64 : 32;
        for (int i = 0; i < n; ++i) {
            RT_CHECK(c[i].ks == 3 && c[i].stride == 1 && c[i].in.pad == 1 && c[i].cin % bk == 0, RTMODT_E_INVALID,
                     "launch_conv: tile %s needs a 3x3 stride-1 conv on a bordered input with cin %% %d == 0", tile_name(tile), bk);
            a[i].rows_wq = a[i].in_Wp - 1;                   // the GEMM runs over the padded positions of the rows that hold outputs, right border column left out
            a[i].M = c[i].B * a[i].Ho * a[i].rows_wq;
            a[i].d_hwp = make_fastdiv(a[i].Ho * a[i].rows_wq); a[i].d_wp = make_fastdiv(a[i].rows_wq);
            a[i].last_pos = c[i].B * a[i].in_Hp * a[i].in_Wp - 1;
            RT_CHECK((long)(a[i].last_pos + 1) * a[i].in_cs < (1L << 31) && a[i].last_pos + 1 < (1 << 28), RTMODT_E_INVALID, "launch_conv: tensor exceeds 2^31 elements");
        }
    }
    if (tile_is_tail(tile)) {
        const TileShape ts = tile_shape(tile);
        const int n2max = ts.bn == 64 ? 64 : 128;
        RT_CHECK(n == 1 && !general && c[0].tail_wt && c[0].tail_bias && c[0].tail_out.base, RTMODT_E_INVALID, "launch_conv: tile %s needs one conv with a tail", tile_name(tile));
        RT_CHECK(c[0].cout == ts.bn && c[0].tail_kp == c[0].cout && c[0].tail_cout >= 8 && c[0].tail_cout <= n2max && c[0].tail_cout % 8 == 0,
                 RTMODT_E_INVALID, "launch_conv: tile %s cannot fuse cout %d -> tail %d", tile_name(tile), c[0].cout, c[0].tail_cout);
        RT_CHECK(!c[0].res.base && !c[0].out2.base && c[0].tail_out.H == c[0].out.H && c[0].tail_out.W == c[0].out.W && c[0].tail_out.c == c[0].tail_cout &&
                     c[0].tail_out.coff % 8 == 0 && c[0].tail_out.C % 8 == 0,
                 RTMODT_E_INVALID, "launch_conv: tile %s: tail output view", tile_name(tile));
    }
    // workgroups are dispatched in id order and problem 0 owns the first ids: the problems whose workgroups run longest
    // (deepest K) go first, so that the short ones fill the tail instead of the other way round (Detect stage 0 at 8
    // frames: 83 -> 61 us)
    std::stable_sort(a, a + n, [](const ConvArgs &x, const ConvArgs &y) { return x.K > y.K; });
    LaunchPlan l{a, n, general};
    switch (tile) {
        case TILE_128x128: launch_tile<128, 128, 2, 2, 3>(l, s); break;
        case TILE_128x64: launch_tile<128, 64, 2, 2, 3>(l, s); break;
        case TILE_64x64: launch_tile<64, 64, 2, 2, 3>(l, s); break;
        case TILE_256x32: launch_tile<256, 32, 4, 1, 3>(l, s); break;
        case TILE_64x128: launch_tile<64, 128, 1, 4, 3>(l, s); break;
        case TILE_WSK_64x64: launch_wsk<64, 64>(l, s); break;
        case TILE_WSK_32x64: launch_wsk<32, 64>(l, s); break;
        case TILE_WSK_64x32: launch_wsk<64, 32>(l, s); break;
        case TILE_K64_64x64_S3: launch_k64<64, 64, 2, 2, 3>(l, s); break;
        case TILE_ROWS_128x32: launch_rows<128, 32, 4, 1, false>(l, s); break;
        case TILE_ROWS_K64_64x64: launch_rows<64, 64, 2, 2, true>(l, s); break;
        case TILE_ROWS_256x64_W8: launch_rows_w8<256, 64, 4, 2, false>(l, s); break;
        case TILE_K64_128x128_S2_W8: RT_TRY((launch_k64_w8<128, 128, 4, 2, 2>(l, s))); break;
        case TILE_K64_128x128_S3_W8: RT_TRY((launch_k64_w8<128, 128, 4, 2, 3>(l, s))); break;
        case TILE_K64_128x64_S3_W8: RT_TRY((launch_k64_w8<128, 64, 4, 2, 3>(l, s))); break;
        case TILE_K64_256x64_S2_W8: RT_TRY((launch_k64_w8<256, 64, 8, 1, 2>(l, s))); break;
        case TILE_PP_256x128: RT_TRY(launch_conv3x3_pp(a, n, 128, s)); break;
        case TILE_PP_256x64: RT_TRY(launch_conv3x3_pp(a, n, 64, s)); break;
        case TILE_PP_256x192: RT_TRY(launch_conv3x3_pp(a, n, 192, s)); break;
        case TILE_PP_512x64: RT_TRY(launch_conv3x3_pp(a, n, 576, s)); break;
        case TILE_PPT_256x128: RT_CHECK(n == 1, RTMODT_E_INVALID, "launch_conv: the ping-pong tile kernel runs single problems"); RT_TRY(launch_conv_tile_pp(a[0], 128, s)); break;
        case TILE_PT_128x128_S2: RT_TRY((launch_pt<128, 2>(l, s))); break;
        case TILE_PT_128x64_S2: RT_TRY((launch_pt<64, 2>(l, s))); break;
        case TILE_TAIL_128x64: hipLaunchKernelGGL((conv_mfma_tail<128, 64, 2, 2, 3, 4>), l.grid(128, 64), dim3(256), 0, s, a[0]); break;
        case TILE_TAIL_K64_128x128: hipLaunchKernelGGL((conv_mfma64_tail<128, 128, 4, 1, 2, 8>), l.grid(128, 128), dim3(256), 0, s, a[0]); break;
        default: return fail(RTMODT_E_INVALID, "launch_conv: tile %d", tile);
    }
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

int launch_conv(const ConvLaunch &c, hipStream_t s) { return launch_conv_group(&c, 1, c.tile, s); }

// ---------------------------------------------------------------------------------------
// stem: 3->cout 3x3/s2 conv on the letterboxed RGB0 fp16 image (1-pixel zero border == the
// conv's zero padding).  K = 27 is re-indexed so that the matrix cores can take it straight
// from NHWC4 memory with 16-byte loads and no gather: k' = kh*16 + kw*4 + c with kw in 0..3,
// c in 0..3 (the 4th tap and the 4th channel meet zero weights), K' = 48 padded to 64 = two
// v_mfma_f32_16x16x32_f16 steps.  Lane (pixel p, chunk q) of step kk reads the two adjacent
// pixels (2ox + 2(q&1), +1) of input row 2oy + 2kk + (q>>1): 16 contiguous bytes.
// One wave owns 16 consecutive output pixels of a row per iteration; weights live in VGPRs.
// ---------------------------------------------------------------------------------------

// Stem epilogue: the 16 pixels x 16 NT channels a wave just computed are one contiguous run of the NHWC
// output (the stem's output tensor has exactly 16 NT channels per pixel), so they leave as ONE 16-byte
// store per lane (whole cache lines) after a transpose through a padded LDS tile, instead of NT
// 8-byte stores with 32-byte segments.  `tile`: this wave's 16 x (32 NT + 8) bytes.
template <int NT>
__device__ __forceinline__ void stem_store(unsigned char *tile, const floatx4 *acc, f16 *__restrict__ orow, int lane) {
    constexpr int ROW = 32 * NT + 8;                       // bytes per pixel in LDS (+8: conflict-free b64 writes)
    const int p = lane & 15, q = lane >> 4;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        floatx4 sv = acc[u]; silu4(sv); half4 h = {(f16)sv[0], (f16)sv[1], (f16)sv[2], (f16)sv[3]};
        *(half4 *)(tile + p * ROW + u * 32 + q * 8) = h;
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int CH16 = 2 * NT;                           // 16-byte chunks per pixel
#pragma unroll
    for (int i = 0; i < (16 * CH16 + 63) / 64; ++i) {
        const int idx = i * 64 + lane;
        if (16 * CH16 % 64 == 0 || idx < 16 * CH16) {
            const int px = idx / CH16, c = idx - px * CH16;
            *(half8 *)(orow + px * (16 * NT) + c * 8) = *(const half8 *)(tile + px * ROW + c * 16);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <int NT>
__global__ __launch_bounds__(256) void stem_mfma(const f16 *__restrict__ img, int Hp, int Wp, f16 *__restrict__ out, int Ho, int Wo,
                                                 int oHp, int oWp, int ocs, int opad, const f16 *__restrict__ wm,
                                                 const float *__restrict__ bias, int groups) {
    __shared__ __attribute__((aligned(16))) unsigned char otile[4 * 16 * (32 * NT + 8)];
    const int lane = threadIdx.x & 63;
    const int p = lane & 15, q = lane >> 4;
    half8 wf[NT][2];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) wf[u][kk] = *(const half8 *)(wm + (u * 16 + p) * 64 + kk * 32 + q * 8);
    floatx4 bv[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) bv[u] = *(const floatx4 *)(bias + u * 16 + q * 4);
    const int gpr = Wo >> 4;                                   // 16-pixel groups per output row
    const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * 256) >> 6;
    for (int g = wave_global; g < groups; g += n_waves) {
        int row = g / gpr, gx = g - row * gpr;
        int b = row / Ho, oy = row - b * Ho;
        int ox = gx * 16 + p;
        const int kh0 = q >> 1, kh1 = min(2 + (q >> 1), 2);     // k' >= 48 has zero weights: re-read row 2
        const f16 *base = img + ((long)(b * Hp + 2 * oy) * Wp + 2 * ox + 2 * (q & 1)) * 4;
        half8 a0 = *(const half8 *)(base + (long)kh0 * Wp * 4);
        half8 a1 = *(const half8 *)(base + (long)kh1 * Wp * 4);
        floatx4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u][0], a0, bv[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u][1], a1, acc[u], 0, 0, 0);
        }
        if (ocs == 16 * NT) {
            stem_store<NT>(otile + (threadIdx.x >> 6) * (16 * (32 * NT + 8)), acc, out + ((long)(b * oHp + oy + opad) * oWp + gx * 16 + opad) * ocs, lane);
        } else {
            f16 *o = out + ((long)(b * oHp + oy + opad) * oWp + ox + opad) * ocs + q * 4;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                floatx4 sv = acc[u]; silu4(sv); half4 h = {(f16)sv[0], (f16)sv[1], (f16)sv[2], (f16)sv[3]};
                *(half4 *)(o + u * 16) = h;
            }
        }
    }
}

int launch_stem(const TensorView &img4, const TensorView &out, const f16 *wm, const float *bias, int B, int cout, hipStream_t s) {
    RT_CHECK(img4.C == 4 && img4.pad == 1, RTMODT_E_INVALID, "launch_stem: image tensor must be 4-channel with border");
    int Ho = (img4.H - 1) / 2 + 1, Wo = (img4.W - 1) / 2 + 1;
    RT_CHECK(Ho == out.H && Wo == out.W && out.c == cout && out.coff == 0 && out.C % 4 == 0, RTMODT_E_INVALID, "launch_stem: output shape");
    RT_CHECK(Wo % 16 == 0 && img4.W % 2 == 0 && cout % 16 == 0 && cout <= 80, RTMODT_E_UNSUPPORTED, "launch_stem: Wo %d / cout %d", Wo, cout);
    int groups = B * Ho * (Wo / 16);
    dim3 grid(std::min(cdiv(groups, 4), 256 * 8));
#define STEM_CASE(NTILES)                                                                                              \
    case NTILES:                                                                                                       \
        hipLaunchKernelGGL((stem_mfma<NTILES>), grid, dim3(256), 0, s, img4.base, img4.H + 2, img4.W + 2, out.base, Ho, Wo, \
                           out.H + 2 * out.pad, out.W + 2 * out.pad, out.C, out.pad, wm, bias, groups);                 \
        break;
    switch (cout / 16) {
        STEM_CASE(1) STEM_CASE(2) STEM_CASE(3) STEM_CASE(4) STEM_CASE(5)
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
// stem_fused: letterbox + BGR->RGB + /255 + .half() (preprocess.hip) folded into the stem conv for
// frames that need no resize (source size == letterboxed size; pads allowed).  One workgroup owns
// one output row: the three letterboxed input rows it needs are assembled in LDS as BGR bytes --
// source bytes fetched with coalesced aligned dword pairs and byte-shifted into place, 114 where the
// canvas has no image -- and every MFMA fragment (two adjacent input pixels of one row) is then six
// LDS byte reads through a 256-entry fp16 table of c/255 (built on the host with IEEE division, so the
// values equal the letterbox kernel's bit for bit); zeros outside the canvas are the conv's padding.
// Saves the 4-channel fp16 image's round trip through HBM and one launch per step.
// ---------------------------------------------------------------------------------------
struct StemSrc { FramePtrs frames; int pitch, top, left, new_h, new_w, in_h, in_w, frame0, frame_bytes; };

template <int NT>
__global__ __launch_bounds__(256) void stem_fused(StemSrc src, const f16 *__restrict__ lut_g, f16 *__restrict__ out, int Ho, int Wo,
                                                  int oHp, int oWp, int ocs, int opad, const f16 *__restrict__ wm,
                                                  const float *__restrict__ bias, int rowb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    unsigned char *rows = fsm;                             // [3][rowb] letterboxed BGR bytes of input rows 2oy-1 .. 2oy+1
    f16 *lut = (f16 *)(fsm + 3 * rowb);                    // [256]
    half4 *pix = (half4 *)(lut + 256);                     // [3][in_w + 2] (R,G,B,0) with one zero pixel either side
    unsigned char *otile = (unsigned char *)(pix + 3 * (src.in_w + 2));   // [4 waves][16 x (32 NT + 8)]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, q = lane >> 4;
    const int b = blockIdx.x / Ho, oy = blockIdx.x - b * Ho;
    (void)lut_g;                                           // (the c/255 table of the first version; kept in the signature)
    STAMP(0);
    // ---- the three source rows, RAW: whole aligned 16-byte chunks by LDS-DMA (1 KiB per wave instruction, no VGPR trip,
    // no byte shifting here).  Row r of the canvas rows 2oy-1 .. 2oy+1 lands at raw + r * rowb with its first source byte
    // at offset mis[r] = (address of the source row) & 15.  Chunk addresses are clamped to the chunks that hold at least
    // one frame byte (a 16-byte aligned chunk never crosses a page), so nothing outside the frame's pages is touched;
    // clamped chunks only ever cover bytes no canvas pixel maps to. ----
    const uint8_t *f = src.frames.p[src.frame0 + b];
    const uint8_t *flo = f - ((uintptr_t)f & 15);
    const long fspan = (((uintptr_t)f & 15) + src.frame_bytes + 15) & ~15L;     // bytes from flo to the end of the last chunk
    const int ipr = rowb >> 10;                              // wave instructions per row
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int i = wave_u; i < 3 * ipr; i += 4) {
        const int r = i / ipr, part = i - r * ipr;
        const int y = 2 * oy - 1 + r, sy = y - src.top;
        if (y < 0 || y >= src.in_h || sy < 0 || sy >= src.new_h) continue;       // (wave-uniform) pad row or outside the canvas
        const long row0 = ((uintptr_t)f & 15) + (long)sy * src.pitch;            // source row's first byte, from flo
        long off = (row0 & ~15L) + (part * 64 + lane) * 16;
        off = off < 0 ? 0 : (off > fspan - 16 ? fspan - 16 : off);
        glds16((const f16 *)(flo + off), rows + r * rowb + part * 1024);
    }
    half8 wf[NT][2];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) wf[u][kk] = *(const half8 *)(wm + (u * 16 + p) * 64 + kk * 32 + q * 8);
    floatx4 bv[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) bv[u] = *(const floatx4 *)(bias + u * 16 + q * 4);
    STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(2);
    // ---- bytes -> (R, G, B, 0) fp16 pixels; canvas column x sits at index x + 1, zeros outside the canvas, 114 where the
    // canvas has no image ----
    // Four pixels = twelve source bytes per work item, read as four aligned dwords and byte-aligned with v_alignbyte;
    // byte -> float -> * (1/255) -> half equals the letterbox kernel's (half)(c / 255.f) for all 256 byte values (checked
    // exhaustively: tests/test_oracle_yolo.py), so no table is needed.
    const int pw = src.in_w + 2;
    const int qpr = src.in_w >> 2;                         // groups of four pixels per row
    for (int i = threadIdx.x; i < 3 * qpr; i += 256) {
        const int r = i / qpr, g = i - r * qpr;
        const int y = 2 * oy - 1 + r, sy = y - src.top;
        const bool in_canvas = y >= 0 && y < src.in_h, in_img = in_canvas && sy >= 0 && sy < src.new_h;
        unsigned d0 = 0x72727272u, d1 = 0x72727272u, d2 = 0x72727272u;
        if (in_img) {
            const int mis = (int)((((uintptr_t)f & 15) + (long)sy * src.pitch) & 15);
            const int o = mis + 3 * (4 * g - src.left);    // raw-row offset of this group's first byte
            const unsigned char *rr = rows + r * rowb;
            if (4 * g >= src.left && 4 * g + 3 < src.left + src.new_w) {
                const unsigned *rp = (const unsigned *)(rr + (o & ~3));
                const unsigned q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
                const unsigned sh = (unsigned)(o & 3);
                d0 = __builtin_amdgcn_alignbyte(q1, q0, sh);
                d1 = __builtin_amdgcn_alignbyte(q2, q1, sh);
                d2 = __builtin_amdgcn_alignbyte(q3, q2, sh);
            } else {                                       // group straddles an edge of the image: byte by byte
                unsigned dd[3] = {0u, 0u, 0u};
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const int x = 4 * g + j / 3;
                    const unsigned v = x >= src.left && x < src.left + src.new_w ? rr[o + j] : 114u;
                    dd[j >> 2] |= v << (8 * (j & 3));
                }
                d0 = dd[0]; d1 = dd[1]; d2 = dd[2];
            }
        }
        const float k = in_canvas ? 1.0f / 255.0f : 0.0f;
        auto cv = [&](unsigned byte) -> f16 { return (f16)((float)byte * k); };
        half4 *dst = pix + r * pw + 1 + 4 * g;
        dst[0] = half4{cv((d0 >> 16) & 255u), cv((d0 >> 8) & 255u), cv(d0 & 255u), (f16)0.f};
        dst[1] = half4{cv((d1 >> 8) & 255u), cv(d1 & 255u), cv(d0 >> 24), (f16)0.f};
        dst[2] = half4{cv(d2 & 255u), cv(d1 >> 24), cv((d1 >> 16) & 255u), (f16)0.f};
        dst[3] = half4{cv(d2 >> 24), cv((d2 >> 16) & 255u), cv((d2 >> 8) & 255u), (f16)0.f};
    }
    if (threadIdx.x < 6) pix[(threadIdx.x >> 1) * pw + (threadIdx.x & 1) * (src.in_w + 1)] = half4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    STAMP(3);
    __syncthreads();
    STAMP(4);

    auto pair = [&](int r, int x0) -> half8 {              // pixels (x0, x0 + 1) of canvas row 2oy - 1 + r as {R,G,B,0,R,G,B,0}
        if (r < 0) return half8{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        const half4 lo = pix[r * pw + x0 + 1], hi = pix[r * pw + x0 + 2];
        return half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    const int gpr = Wo >> 4;
    for (int gx = wave; gx < gpr; gx += 4) {
        const int ox = gx * 16 + p;
        const int x0 = 2 * ox + 2 * (q & 1) - 1;
        half8 a0 = pair(q >> 1, x0);                         // kernel rows 0 / 1
        half8 a1 = pair((q >> 1) ? -1 : 2, x0);              // kernel row 2; k' >= 48 meets zero weights
        floatx4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u][0], a0, bv[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u][1], a1, acc[u], 0, 0, 0);
        }
        if (ocs == 16 * NT) {
            stem_store<NT>(otile + wave * (16 * (32 * NT + 8)), acc, out + ((long)(b * oHp + oy + opad) * oWp + gx * 16 + opad) * ocs, lane);
        } else {
            f16 *o = out + ((long)(b * oHp + oy + opad) * oWp + ox + opad) * ocs + q * 4;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                floatx4 sv = acc[u]; silu4(sv); half4 h = {(f16)sv[0], (f16)sv[1], (f16)sv[2], (f16)sv[3]};
                *(half4 *)(o + u * 16) = h;
            }
        }
    }
    STAMP(5);
}

int launch_stem_fused(const FramePtrs &frames, int frame0, int pitch, const LetterboxGeom &g, int in_h, int in_w, const f16 *lut,
                      const TensorView &out, const f16 *wm, const float *bias, int B, int cout, hipStream_t s) {
    RT_CHECK(!g.resize, RTMODT_E_INVALID, "launch_stem_fused: frames that need a resize go through the letterbox kernel");
    int Ho = (in_h - 1) / 2 + 1, Wo = (in_w - 1) / 2 + 1;
    RT_CHECK(Ho == out.H && Wo == out.W && out.c == cout && out.coff == 0 && out.C % 4 == 0, RTMODT_E_INVALID, "launch_stem_fused: output shape");
    RT_CHECK(Wo % 16 == 0 && in_w % 2 == 0 && cout % 16 == 0 && cout <= 80, RTMODT_E_UNSUPPORTED, "launch_stem_fused: Wo %d / cout %d", Wo, cout);
    RT_CHECK(frame0 >= 0 && frame0 + B <= 64, RTMODT_E_INVALID, "launch_stem_fused: frames %d..%d", frame0, frame0 + B);
    RT_CHECK((long)g.src_h * pitch < (1L << 31), RTMODT_E_INVALID, "launch_stem_fused: frame too large");
    StemSrc src{frames, pitch, g.top, g.left, g.new_h, g.new_w, in_h, in_w, frame0, (g.src_h - 1) * pitch + 3 * g.src_w};
    const int rowb = (int)align_up((size_t)3 * in_w + 32, 1024);     // raw source row in LDS: whole 1 KiB DMA pieces, misalignment + read-ahead slack
    const int nt = cout / 16;
    const size_t smem = (size_t)3 * rowb + 512 + (size_t)3 * (in_w + 2) * 8 + (size_t)4 * 16 * (32 * nt + 8);
    dim3 grid(B * Ho);
#define STEMF_CASE(NTILES)                                                                                             \
    case NTILES:                                                                                                       \
        hipLaunchKernelGGL((stem_fused<NTILES>), grid, dim3(256), smem, s, src, lut, out.base, Ho, Wo, out.H + 2 * out.pad, \
                           out.W + 2 * out.pad, out.C, out.pad, wm, bias, rowb);                                        \
        break;
    switch (nt) {
        STEMF_CASE(1) STEMF_CASE(2) STEMF_CASE(3) STEMF_CASE(4) STEMF_CASE(5)
        default: return fail(RTMODT_E_UNSUPPORTED, "launch_stem_fused: cout %d", cout);
    }
#undef STEMF_CASE
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

// ---------------------------------------------------------------------------------------
// eight maxima as four packed instructions (v_pk_max_f16; the element-wise compare + select this replaces was 16 instructions and made the pooling kernel VALU-bound: 618 of its
// 1 900 instructions).  maxnum semantics; the activations are finite, and a signed zero's sign has no effect downstream.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half8 hmax8(half8 a, half8 b) {
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const half2v m = __builtin_elementwise_max(half2v{a[j], a[j + 1]}, half2v{b[j], b[j + 1]});
        o[j] = m[0]; o[j + 1] = m[1];
    }
    return o;
}

__global__ __launch_bounds__(512) void sppf_pool(const f16 *__restrict__ y, int Hp, int Wp, int cs, int pad, int H, int W,
                                                 f16 *__restrict__ o1, f16 *__restrict__ o2, f16 *__restrict__ o3, int ocs,
                                                 int chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *tile = (half8 *)smem;               // [H*W]
    half8 *r5 = tile + H * W;                  // row maxima radius 2
    half8 *r9 = r5 + H * W;                    // radius 4
    half8 *r13 = r9 + H * W;                   // radius 6
    // A workgroup owns 8 channels (16 B) of every pixel; the 8 workgroups that share a 128-byte line of the tensor are
    // placed on ONE XCD (workgroups go round-robin over the 8 XCDs), so their partial-line reads and writes meet in one L2
    // instead of eight.
    int work = blockIdx.x;
    if (chunks % 8 == 0 && (gridDim.x / 8) % 8 == 0) {
        const int xcd = work & 7, t = work >> 3;
        work = ((t >> 3) * 8 + xcd) * 8 + (t & 7);           // (line group, chunk within the line)
    }
    int b = work / chunks, ch = (work % chunks) * 8;
    for (int i = threadIdx.x; i < H * W; i += 512) {
        int yy = i / W, xx = i - yy * W;
        tile[i] = *(const half8 *)(y + ((long)(b * Hp + yy + pad) * Wp + xx + pad) * cs + ch);
    }
    __syncthreads();
    // max is idempotent: a neighbour outside the map is replaced by the nearest one inside (already in the window) -- no predicates
    for (int i = threadIdx.x; i < H * W; i += 512) {
        const int yy = i / W, xx = i - yy * W, lmax = xx, rmax = W - 1 - xx;
        half8 m = tile[i];
#pragma unroll
        for (int d = 1; d <= 6; ++d) {
            m = hmax8(m, hmax8(tile[i - min(d, lmax)], tile[i + min(d, rmax)]));
            if (d == 2) r5[i] = m;
            if (d == 4) r9[i] = m;
        }
        r13[i] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H * W; i += 512) {
        const int yy = i / W, xx = i - yy * W, umax = yy, dmax = H - 1 - yy;
        half8 a = r5[i], bq = r9[i], c = r13[i];
#pragma unroll
        for (int d = 1; d <= 6; ++d) {
            const int up = i - min(d, umax) * W, dn = i + min(d, dmax) * W;
            if (d <= 2) a = hmax8(a, hmax8(r5[up], r5[dn]));
            if (d <= 4) bq = hmax8(bq, hmax8(r9[up], r9[dn]));
            c = hmax8(c, hmax8(r13[up], r13[dn]));
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
    static DynLdsSeen seen;                            // raised once per device, outside graph capture (engine runs an eager pass first)
    RT_TRY(raise_dynamic_lds((const void *)sppf_pool, smem, seen));
    hipLaunchKernelGGL(sppf_pool, dim3(B * chunks), dim3(512), smem, s,      // 20x20 pixels at P5: one pass of 512 threads per phase
                       y.base + y.coff, y.H + 2 * y.pad, y.W + 2 * y.pad, y.C,
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
