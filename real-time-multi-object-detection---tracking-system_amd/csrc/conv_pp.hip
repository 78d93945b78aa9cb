// conv_pp.hip -- 3x3 / stride-1 convolution with the two halves of a workgroup in ANTI-PHASE ("ping-pong") for gfx950.
//
// What it replaces: the 3x3 Conv+BN+SiLU layers of the YOLOv8 forward pass behind /root/reference/src/detection/detector.py:100-111
// (C2f bottlenecks, Detect stages 0 and 1: 58 % of the net's FLOPs).  Same arithmetic as conv3x3_rows (conv.hip): implicit GEMM over
// padded pixel positions on v_mfma_f32_16x16x32_f16, fp32 accumulate, weights = MFMA A operand, pixels = B operand.
//
// Why another kernel (profiles/r03/pf/): in every tile kernel of conv.hip all waves of a workgroup do the same thing at
// the same time -- wait for the stage, issue the next DMA pieces, read fragments, multiply -- so the matrix pipe of a SIMD idles while its
// waves wait, issue and read (a k-step of 1 500-2 000 clk around 512 clk of MFMA; two workgroups per CU at a random phase reach ~60 %).
// Here ONE workgroup of 8 waves owns a 256-pixel x BN-cout tile and is cut into two halves (waves 0-3 / 4-7: one wave of each half on every
// SIMD) that run the same program ONE BARRIER INTERVAL APART:
//
//      interval      2j                    2j+1                  2j+2
//      half 0    R(j): read+issue      M(j): 32 x MFMA       R(j+1)
//      half 1    M(j-1)                R(j): read+issue      M(j)
//
// so in every interval one wave per SIMD feeds the matrix pipe while its partner reads its fragments (ds_read_b128), issues the LDS-DMA
// pieces of later phases and waits for its own older pieces (counted vmcnt).  A phase = one tap (kh, kw) x 64 input channels:
// 4 x TN MFMA tiles x 2 k-halves per wave.  Tap reuse as in conv3x3_rows: a strip of 256 + 8 padded positions per (kh, 64 channels) serves
// kw = 0, 1, 2 at row offsets 0, 1, 2.  LDS: two strip slots (33 KiB each) + a three-slot ring of per-tap weight slices (BN x 128 B).
// A workgroup is persistent: it walks over tiles id, id + G, ... of a (grouped) launch and the DMA stream runs on into the next tile
// (its first strip and two taps are in flight under this tile's last phases and epilogue).
//
// Order rules (cdna guide, "Read a staged buffer one phase AFTER the wait that retires it"):
//  * RAW  a wave waits for ITS pieces of phase j+1 inside R(j), in front of the barrier that ends R(j); every reader of phase j+1 is behind a
//         later barrier (half 0: the next one; half 1: the one after).
//  * WAR  a slot is refilled from R(j+1) of half 0 on; its last reader is R(j) of half 1, one interval earlier, and every R ends with
//         s_waitcnt lgkmcnt(0) IN FRONT of its barrier -- the reads have returned before any wave can issue into the slot.
//  * vmcnt retires in issue order, so inside a phase the weight pieces (needed one tap later, L2-hot) are issued BEFORE the strip pieces (needed a
//         whole super-step later, from the Infinity Cache / HBM).  Issue lists per phase and the resulting counts: tile_math.h (pp_issue, pp_wait_count),
//         checked by a host-side replay of the stream (tests/test_tile_math_cpu.py).
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "conv_dev.h"

namespace rtmodt {

// The DMA schedule of a super-step (issue lists per phase, counted waits): pp_issue / pp_wait_count in tile_math.h, replayed on the host by
// tests/native/tile_math_check.cpp.

// WIDE (the 64-cout layers: C2f bottlenecks at 80 x 80, Detect's box branch): 512 positions x 64 couts per workgroup instead of 256 x BN -- with
// 64 couts a 256-position tile gives a wave 64 x 32 outputs and a half 256 clk of MFMA per phase, against ~470 clk of reads + DMA issue; here every
// wave keeps its 64 x 64 outputs (waves 2 wm + wn: positions 256 wn + 64 wm, all 64 couts) and the halves their 512-clk phase, for 8 KiB of weights
// and 22 KiB of strip per phase.
template <int BN, bool WIDE = false>
struct PpCfg {
    static constexpr int BM = WIDE ? 512 : 256, NW = 8;
    static constexpr int NAP = BM / 8 + 1;          // strip pieces of 8 rows x 128 B: BM + 8 rows >= BM + 2
    static constexpr int ASLOT = NAP * 1024;
    static constexpr int NBP = BN / 8;              // weight pieces per tap
    static constexpr int BSLOT = NBP * 1024;
    static constexpr int B_OFF = 2 * ASLOT;
    static constexpr int LDS_BYTES = B_OFF + 3 * BSLOT;
    static constexpr int LB = NBP / NW;             // weight pieces per wave and tap
    static constexpr int TM = 4, TN = WIDE ? BN / 16 : BN / 32;      // a wave owns 64 pixels x BN / 2 couts (WIDE: x all BN couts)
    static constexpr int HB = WIDE ? BN : BN / 2;   // couts per wave
    static constexpr int LA0 = WIDE ? 4 : 2, LA1 = WIDE ? 4 : 2;      // strip pieces a wave issues in phases kw 0 / kw 1 (wave 0: one more in kw 0, piece NAP - 1)
    static_assert(NW * (LA0 + LA1) + 1 == NAP, "strip pieces");
    static_assert(!WIDE || BN == 64, "the wide form is for 64 couts");
    static_assert(NBP % NW == 0, "weight pieces must split evenly over the waves");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// Diagnostic build only (-DRTMODT_STAMP -DPP_FINE): lane 0 of waves 0 and 4 stamps the parts of ONE phase (first tile, super-step 1, kw 1) into slots
// 0-7 / 8-15 of the workgroup's stamp row: R start, pieces issued, fragments read, counted wait done, barrier passed, MFMAs issued, barrier passed
#if defined(RTMODT_STAMP) && defined(PP_FINE)
#define FSTAMP(k)                                                                                                   \
    do {                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if (fine && (threadIdx.x & 255) == 0) {                                                                     \
            unsigned long long t_;                                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
            g_stamps[(size_t)blockIdx.x * 16 + (threadIdx.x >> 8) * 8 + (k)] = t_;                                  \
        }                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    } while (0)
#undef STAMP
#define STAMP(k)
#else
#define FSTAMP(k)
#endif


// Where output position m (of the launch's position enumeration: `wq` positions per image row, of which the first Wo hold outputs) and channel n go:
// element offset into the output (and shortcut) tensor, and whether there is anything to store.  One multiply-high + shift per division (FastDiv),
// 24-bit multiplies (full rate; every factor < 2^24: checked at launch) and ONE combined predicate -- the straightforward form (nested conditions, 32-bit
// multiplies, 64-bit offsets) was half of the epilogue's 716 instructions per wave (profiles/r04/pp/README.md).  Since round 5 the 64- and 128-cout tiles call index()
// ONCE per lane and tile, in front of the k-loop, and step from there (profiles/r05/pp_ahead); the 192-cout tile (12 chunks per pixel: a lane's stores are not one
// chunk column) and maps narrower than a lane's step still call it per store.
struct PpOut {
    int M, HW, wq, Ho, Wo, cout;
    int o0, o1, o2, ocs;          // output: offset of (b, y, x) = o0 + b * o1 + y * o2 + x * ocs
    int r0, r1, r2, rcs;          // shortcut tensor, likewise
    FastDiv d_img, d_row;
    __device__ __forceinline__ bool index(int m, int n, int &opix, int &rpix) const {
        const int b = fdiv(m, d_img), rem = m - (int)__umul24((unsigned)b, (unsigned)HW);
        const int oy = fdiv(rem, d_row), ox = rem - (int)__umul24((unsigned)oy, (unsigned)wq);
        opix = o0 + (int)__umul24((unsigned)b, (unsigned)o1) + (int)__umul24((unsigned)oy, (unsigned)o2) + (int)__umul24((unsigned)ox, (unsigned)ocs) + n;
        rpix = r0 + (int)__umul24((unsigned)b, (unsigned)r1) + (int)__umul24((unsigned)oy, (unsigned)r2) + (int)__umul24((unsigned)ox, (unsigned)rcs) + n;
        return (m < M) & (oy < Ho) & (ox < Wo) & (n < cout);
    }
};
__device__ __forceinline__ PpOut pp_out(const ConvArgs &p, int rows_wq, const FastDiv &d_img, const FastDiv &d_row) {
    PpOut o;
    o.M = p.M; o.wq = rows_wq; o.HW = p.Ho * rows_wq; o.Ho = p.Ho; o.Wo = p.Wo; o.cout = p.cout;
    o.ocs = p.out_cs; o.o2 = p.out_Wp * p.out_cs; o.o1 = p.out_Hp * o.o2; o.o0 = p.out_pad * o.o2 + p.out_pad * p.out_cs;
    o.rcs = p.res_cs; o.r2 = p.res_Wp * p.res_cs; o.r1 = p.res_Hp * o.r2; o.r0 = p.res_pad * o.r2 + p.res_pad * p.res_cs;
    o.d_img = d_img; o.d_row = d_row;
    return o;
}
// the host-side condition of PpOut's 24-bit multiplies
static bool pp_out_fits(const ConvArgs &a) {
    return pp_index_fits((long)a.out_Hp * a.out_Wp * a.out_cs, a.res ? (long)a.res_Hp * a.res_Wp * a.res_cs : 0L, (long)a.Ho * a.in_Wp, a.M);
}

template <int N>
__device__ __forceinline__ void pp_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BN, bool WIDE>
__global__ __launch_bounds__(512) void conv3x3_pp(ConvGroupArgs g, int total_ids, const int *sched, int sched_T) {
    using C = PpCfg<BN, WIDE>;
    constexpr int BM = C::BM, NW = C::NW, LB = C::LB, TM = C::TM, TN = C::TN, LA0 = C::LA0, LA1 = C::LA1, XP = LA0 + LA1;      // XP: index of wave 0's extra piece
    constexpr bool BIAS_AHEAD = BN <= 128;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[C::LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int half = wave >> 2;                       // 0: waves 0-3, 1: waves 4-7 (one of each per SIMD)
    const int row0 = WIDE ? (wave & 1) * 256 + (wave >> 1) * 64 : (wave >> 1) * 64;      // first position of this wave's 64 inside the tile (half h = waves 4 h .. 4 h + 3)
    const int col0 = WIDE ? 0 : (wave & 1) * (BN / 2);                                   // first cout of this wave inside the tile
    const bool w0 = wave == 0;

    // LDS images: rows of 128 B (64 halves = 8 chunks of 16 B), chunk c of row R in slot (c + 2 (R >> 1)) & 7 -- a rotation that keeps
    // ds_read_b128 conflict-free from ANY 16 consecutive rows (tile_math.h swz_slot<128>; the taps start reads at rows +0, +1, +2)
    auto slot_of = [](int R, int c) { return ((c + 2 * (R >> 1)) & 7) << 4; };
    int a_rd[3][2], b_rd[2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a_rd[kw][kk] = row0 * 128 + (r + kw) * 128 + slot_of(r + kw, kk * 4 + q);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_rd[kk] = C::B_OFF + col0 * 128 + r * 128 + slot_of(r, kk * 4 + q);
    const int ld_row = lane >> 3, ld_slot = lane & 7;
    const int ld_chunk = ((ld_slot - 2 * (ld_row >> 1)) & 7) * 8;      // halves: the k-chunk whose slot this DMA lane fills (piece bases are multiples of 8 rows)

    // A tile's DMA sources are (wave-uniform base: tensor + the tap row / channel chunk of the step) + (32-bit per-lane byte offset, fixed for the
    // tile): global_load_lds takes the pair as SGPR base + VGPR offset, so a piece costs scalar adds and one vector instruction -- no per-piece 64-bit
    // VALU address chain beside the partner's MFMAs (profiles/r04/pp/run3.txt: ~108 clk per piece with per-lane 64-bit addresses, MFMA phase 670 clk for 512)
    struct Tile {
        const char *in, *wt;                          // of the tile's problem (wave-uniform)
        int z, m0, n0, ns, cin;
        unsigned row_bytes;                           // bytes between two padded input rows (the kh step)
        unsigned b_step;                              // bytes from weight piece j to j + 8: 64 cout rows
    };                                                // (wave-uniform values only: two of these live in SGPRs)
    // per lane, bytes, of the tile the DMA stream is feeding: its row of strip piece wave + 8 i (i = 4: piece 32, wave 0 only); its row of weight piece
    // `wave`.  Worked out when the stream moves on to a tile (lane_offsets), not kept per located tile: the 192-wide tile has no registers for a second set.
    unsigned ia_off[XP + 1], ib_off;
    const int G = gridDim.x;
    auto locate = [&](int id, Tile &t) -> bool {      // launch-linear id -> (problem, tile); false for the alignment fillers between problems
        int z = 0;
        while (z + 1 < g.n && id >= g.start[z + 1]) ++z;
        id -= g.start[z];
        const ConvArgs &p = g.p[z];
        const int gx = g.gx[z], gy = (p.cout + BN - 1) / BN;
        if (id >= gx * gy) return false;
        const int by = id / gx, bx = id - by * gx;
        int mt, nt;
        xcd_tile(gx, gy, bx, by, mt, nt);
        t.z = z; t.m0 = mt * BM; t.n0 = nt * BN; t.ns = 3 * (p.cin / 64);
        t.in = (const char *)p.in; t.wt = (const char *)p.wt; t.cin = p.cin; t.row_bytes = (unsigned)(p.in_Wp * p.in_cs) * 2u;
        t.b_step = (unsigned)(64 * p.kp) * 2u;
        return true;
    };
    auto lane_offsets = [&](const Tile &t) {
        const ConvArgs &p = g.p[t.z];
        // positions past the last image are clamped so that position + kh rows stays inside the tensor: two rows above its last pixel, on the (zero)
        // right border column for kh = 0, 1 and on the last (zero) pixel for kh = 2; no enumerated position of a real image reaches that far
        const int clamp = p.last_pos - 2 * p.in_Wp;
#pragma unroll
        for (int i = 0; i <= XP; ++i) ia_off[i] = (unsigned)(min(rows_pos(p, t.m0 + (wave + NW * i) * 8 + ld_row), clamp) * p.in_cs + ld_chunk) * 2u;
        ib_off = (unsigned)((t.n0 + wave * 8 + ld_row) * p.kp + ld_chunk) * 2u;
    };
    // the j-th tile of this workgroup: from the launch's schedule (pp_lpt_schedule, read through the SCALAR cache: a vector load would sit in the vmcnt
    // queue of the DMA stream), or -- no schedule -- the ids blockIdx.x, + G, ... with the alignment fillers between problems skipped.  false: no more.
    int walk = blockIdx.x;                            // static walk: the next id to try
    auto next_tile = [&](int j, Tile &t) -> bool {
        if (sched) {
            if (j >= sched_T) return false;
            int id;
            const int off = (blockIdx.x * sched_T + j) * 4;
            asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(id) : "s"(sched), "s"(off) : "memory");
            return id >= 0 && locate(id, t);
        }
        for (; walk < total_ids; walk += G)
            if (locate(walk, t)) { walk += G; return true; }
        return false;
    };
    auto dma = [&](const char *base, unsigned off, unsigned char *dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off), (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    // weight slice of tap (kh, kw), channels [c0, c0 + 64) -> ring slot `slot`
    auto issue_B = [&](const Tile &t, int kh, int kw, int c0, int slot) {
        const char *base = t.wt + (size_t)(((kh * 3 + kw) * t.cin + c0) * 2);
#pragma unroll
        for (int j = 0; j < LB; ++j) dma(base + (size_t)j * t.b_step, ib_off, lds + C::B_OFF + slot * C::BSLOT + (wave + NW * j) * 1024);
    };
    // strip piece wave + 8 i of (kh, c0) -> strip slot `slot`
    auto issue_A = [&](const Tile &t, int kh, int c0, int slot, int i) {
        const char *base = t.in + ((size_t)kh * t.row_bytes + (size_t)(c0 * 2));
        dma(base, ia_off[i], lds + slot * C::ASLOT + (wave + NW * i) * 1024);
    };

    Tile cur, nxt, iss;                               // the tile being multiplied, the one after it, the one the DMA stream is feeding
    int tj = 0;
    if (!next_tile(tj++, cur)) return;                // (the same for every wave of the workgroup)
    bool have_next = next_tile(tj++, nxt);
    iss = cur;
    lane_offsets(cur);

    // bias of the first tile: requested BEFORE the first DMA piece (older than all of them)
    floatx4 bnext[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) bnext[u] = *(const floatx4 *)(g.p[cur.z].bias + cur.n0 + col0 + u * 16 + q * 4);

    // ---- pipeline fill, in the order the steady state would have issued it: strip 0 (first part), tap 0, strip 0 (rest), tap 1 ----
#pragma unroll
    for (int i = 0; i < LA0; ++i) issue_A(cur, 0, 0, 0, i);
    if (w0) issue_A(cur, 0, 0, 0, XP);
    issue_B(cur, 0, 0, 0, 0);
#pragma unroll
    for (int i = LA0; i < XP; ++i) issue_A(cur, 0, 0, 0, i);
    issue_B(cur, 0, 1, 0, 1);
    STAMP(0);
    pp_wait<LB>();                                    // everything but tap 1
    __builtin_amdgcn_s_barrier();
    STAMP(1);
    int aslot = 0;
    int tile_no = 0;

    while (true) {
        const ConvArgs &p = g.p[cur.z];
        // the accumulators START at the bias (fp32; the MFMA chain then adds the products): no bias registers and no ordinary global load whose
        // destination registers the k-loop would have to wait for -- hipcc answers ANY pending VGPR-destination load with s_waitcnt vmcnt(0) at
        // the top of the loop, which drains the DMA ring every super-step
        floatx4 acc[TM][TN];
        if constexpr (!BIAS_AHEAD) {                  // (the 192-wide tile has no registers to carry the next tile's bias through the k-loop)
            if (tile_no > 0) {
#pragma unroll
                for (int u = 0; u < TN; ++u) bnext[u] = *(const floatx4 *)(p.bias + cur.n0 + col0 + u * 16 + q * 4);
            }
        }
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u) acc[t][u] = bnext[u];
        // WHERE this lane's 16-byte stores of the tile go is worked out HERE, in front of the k-loop, not inside the epilogue (round 5): the epilogue is the part of a tile's
        // life in which the matrix pipe idles and the vector ALU is the bottleneck (SiLU), and PpOut::index per store was a third of its instructions; in front of the k-loop
        // the same arithmetic runs while half 1 waits at the STAGGER barrier anyway / beside half 0's first DMA issue.  A lane's stores are positions m, m + KS, m + 2 KS, ...
        // (KS = 64 / CPR) of one 16-byte chunk column: ONE decomposition (two multiply-high divisions), then steps of KS positions with a row / image carry.
        // Element offsets in opx[], the store predicates as bits of okm; the shortcut tensor has the output tensor's geometry (checked at launch): same offsets.
        // (Measured and dropped, profiles/r05/pp_ahead2: one step per phase inside the k-loop's R intervals, +7..17 %; half 0's share inside its first R interval, +3 %.)
        constexpr int EP_CPR = C::HB / 8, EP_NST = EP_CPR, EP_KS = 64 / (EP_CPR > 0 ? EP_CPR : 1);
        constexpr bool EP_AHEAD = C::HB == 64 || C::HB == 32;
        int opx_tile[EP_AHEAD ? EP_NST : 1];
        unsigned okm = 0;
        if constexpr (EP_AHEAD) {
            const PpOut po = pp_out(p, p.rows_wq, p.d_hwp, p.d_wp);
            const int L = lane / EP_CPR, cc = lane - L * EP_CPR;
            const int n = cur.n0 + col0 + cc * 8;
            int m = cur.m0 + row0 + L;
            int b = fdiv(m, po.d_img), rem = m - (int)__umul24((unsigned)b, (unsigned)po.HW);
            int oy = fdiv(rem, po.d_row), x = rem - (int)__umul24((unsigned)oy, (unsigned)po.wq);
            const bool nok = n < po.cout;
            if (po.wq >= EP_KS) {
                int off = po.o0 + (int)__umul24((unsigned)b, (unsigned)po.o1) + (int)__umul24((unsigned)oy, (unsigned)po.o2) + (int)__umul24((unsigned)x, (unsigned)po.ocs) + n;
                const int dx = EP_KS * po.ocs, drow = po.o2 - po.wq * po.ocs, dimg = po.o1 - po.Ho * po.o2;      // a step along the row; what a row carry / an image carry adds
#pragma unroll
                for (int j = 0; j < EP_NST; ++j) {
                    opx_tile[j] = off;
                    okm |= (unsigned)((m < po.M) & (x < po.Wo) & nok) << j;
                    m += EP_KS; x += EP_KS; off += dx;
                    if (x >= po.wq) { x -= po.wq; off += drow; oy += 1; if (oy >= po.Ho) { oy = 0; off += dimg; } }
                }
            } else {                                      // rows shorter than the step (maps narrower than 8 / 16 pixels): the division per store
#pragma unroll
                for (int j = 0; j < EP_NST; ++j) {
                    int rp;
                    okm |= (unsigned)po.index(m + j * EP_KS, n, opx_tile[j], rp) << j;
                }
            }
        }
        if (half == 1) __builtin_amdgcn_s_barrier();  // STAGGER: half 1 runs one interval behind half 0
        int kh = 0, c0 = 0;
        for (int s = 0; s < cur.ns; ++s) {
            const bool last = s + 1 == cur.ns;
            const bool valid = !last || have_next;    // is there a super-step s + 1 (of this tile or of the next one) to prefetch?
            int nkh = kh, nc0 = c0 + 64;
            if (nc0 >= cur.cin) { nc0 = 0; ++nkh; }
            if (last) { nkh = 0; nc0 = 0; }
            const unsigned char *sA = lds + aslot * C::ASLOT;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const bool fine = tile_no == 0 && s == 1 && kw == 1; (void)fine;
                FSTAMP(0);
                // ---------------- R: DMA pieces of later phases, fragments of this phase, counted wait ----------------
                // (the pieces go first: the texture addresser works them off while the LDS serves the fragment reads -- issued the other way round
                // the two take turns: 16 reads ~250 clk, then ~90 clk per piece, profiles/r04/pp/)
                if (kw == 0) {
                    issue_B(iss, kh, 2, c0, 2);
                    if (last && have_next) { iss = nxt; lane_offsets(nxt); }        // from here on the stream feeds the next tile
                    if (valid) {
#pragma unroll
                        for (int i = 0; i < LA0; ++i) issue_A(iss, nkh, nc0, aslot ^ 1, i);
                        if (w0) issue_A(iss, nkh, nc0, aslot ^ 1, XP);
                    }
                } else if (kw == 1) {
                    if (valid) {
                        issue_B(iss, nkh, 0, nc0, 0);
#pragma unroll
                        for (int i = LA0; i < XP; ++i) issue_A(iss, nkh, nc0, aslot ^ 1, i);
                    }
                } else {
                    if (valid) issue_B(iss, nkh, 1, nc0, 1);
                }
                FSTAMP(1);
                half8 fa[2][TM], fb[2][TN];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int u = 0; u < TN; ++u) fb[kk][u] = *(const half8 *)(lds + kw * C::BSLOT + u * 2048 + b_rd[kk]);
#pragma unroll
                    for (int t = 0; t < TM; ++t) fa[kk][t] = *(const half8 *)(sA + t * 2048 + a_rd[kw][kk]);
                }
                FSTAMP(2);
                if (!valid) pp_wait<0>();             // the stream ends here: drain (once per workgroup)
                else if (kw == 0) { if (w0) pp_wait<pp_wait_count(0, LB, true, LA0, LA1)>(); else pp_wait<pp_wait_count(0, LB, false, LA0, LA1)>(); }
                else if (kw == 1) { if (w0) pp_wait<pp_wait_count(1, LB, true, LA0, LA1)>(); else pp_wait<pp_wait_count(1, LB, false, LA0, LA1)>(); }
                else pp_wait<pp_wait_count(2, LB, false, LA0, LA1)>();
                FSTAMP(3);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragments are in registers: the slots may be refilled behind the barrier
                __builtin_amdgcn_s_barrier();
                FSTAMP(4);
                // ---------------- M: the matrix pipe is this half's ----------------
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);                       // (keeps hipcc from moving MFMAs across the barriers; as a priority it measured +-1 %)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int t = 0; t < TM; ++t)
#pragma unroll
                        for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[kk][u], fa[kk][t], acc[t][u], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                FSTAMP(5);
                __builtin_amdgcn_s_barrier();
                FSTAMP(6);
                if (tile_no == 0 && s == 1) STAMP(2 + kw);      // (diagnostic build only) the three phases of the second super-step
            }
            aslot ^= 1; kh = nkh; c0 = nc0;
        }
        if (half == 0) __builtin_amdgcn_s_barrier();  // BALANCE: both halves run the epilogue together (two waves per SIMD share its VALU)
        if (tile_no == 0) STAMP(6);

        // ---- epilogue straight from the accumulators (conv3x3_rows' direct-store path): padded position -> (b, y, x); border positions are junk ----
        if (BIAS_AHEAD && have_next) {                // the next tile's bias travels under this epilogue; pinned behind the k-loop (see above)
            int nz = nxt.z, nn0 = nxt.n0;
            asm volatile("" : "+v"(nz), "+v"(nn0));
            nz = __builtin_amdgcn_readfirstlane(nz); nn0 = __builtin_amdgcn_readfirstlane(nn0);
#pragma unroll
            for (int u = 0; u < TN; ++u) bnext[u] = *(const floatx4 *)(g.p[nz].bias + nn0 + col0 + u * 16 + q * 4);
        }
        // Stores in the accumulator layout are 8 bytes per lane in 32-byte runs of 16 different cache lines per instruction (9 500 clk per tile,
        // 19 000 with the residual read the same way: profiles/r04/pp/run1_stamp.txt).  Each wave therefore turns its 64 x BN/2 outputs round by
        // round through 4 KiB of ITS OWN in the strip slot this tile has finished with (chunk XOR pixel swizzle, no barrier: LDS serves a wave's
        // operations in order) and stores 16 bytes per lane, whole 128-byte lines per pixel.  bias is in the accumulators; SiLU and the residual in
        // fp32, ONE rounding: with a residual the staging holds fp32 and the shortcut is added on the store side, where it is read 16 bytes per lane.
        {
#if defined(RTMODT_STAMP)
            const int ablate = p.epi_prio;            // diagnostic build: RTMODT_EPI_PRIO bit 0 = no SiLU, bit 1 = no global stores (results wrong, times only)
#else
            constexpr int ablate = 0;
#endif
            unsigned char *stg = lds + (aslot ^ 1) * C::ASLOT + wave * 4096;
            constexpr int HB = C::HB, CPR = HB / 8;                      // couts of this wave; 16-byte fp16 chunks per pixel
            // the problem's scalars, fetched in ONE batch here: read where they are used (inside the lanes' branches) hipcc fetches them from the kernel
            // argument segment again for every store, a scalar-memory round trip each -- 5 800 of the 8 000 clk this epilogue took (profiles/r04/pp/run5.txt)
            f16 *const e_out = p.out; const f16 *const e_res = p.res;
            const int e_act = p.act;
            const PpOut po = pp_out(p, p.rows_wq, p.d_hwp, p.d_wp);      // (only the 192-wide form still indexes inside the epilogue)
            if constexpr (EP_AHEAD) asm volatile("" ::"s"(e_out), "s"(e_res), "s"(e_act));
            else
            asm volatile("" ::"s"(e_out), "s"(e_res), "s"(e_act), "s"(po.M), "s"(po.HW), "s"(po.wq), "s"(po.Ho), "s"(po.Wo), "s"(po.cout), "s"(po.o0), "s"(po.o1), "s"(po.o2),
                         "s"(po.ocs), "s"(po.r0), "s"(po.r1), "s"(po.r2), "s"(po.rcs), "s"(po.d_img.mul), "s"(po.d_img.shift), "s"(po.d_row.mul), "s"(po.d_row.shift));
            if (!e_res) {
                constexpr int RS = HB <= 32 ? 64 : (HB <= 64 ? 128 : 256), CH = RS / 16, TR = 4096 / (16 * RS);
                static_assert(TM % TR == 0 && (TR * 16 * CPR) % 64 == 0, "staging rounds");
#pragma unroll
                for (int t0 = 0; t0 < TM; t0 += TR) {
                    if (e_act && !(ablate & 1)) {
#pragma unroll
                        for (int tt = 0; tt < TR; ++tt)
#pragma unroll
                            for (int u = 0; u < TN; ++u) silu4(acc[t0 + tt][u]);
                    }
#pragma unroll
                    for (int tt = 0; tt < TR; ++tt)
#pragma unroll
                        for (int u = 0; u < TN; ++u) {
                            const floatx4 v = acc[t0 + tt][u];
                            const int pl = tt * 16 + r, c = u * 2 + (q >> 1);
                            *(half4 *)(stg + pl * RS + ((c ^ (pl & (CH - 1))) << 4) + (q & 1) * 8) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                        }
#pragma unroll
                    for (int i = 0; i < TR * 16 * CPR / 64; ++i) {
                        const int e = i * 64 + lane, pl = e / CPR, c = e - pl * CPR;
                        const half8 v = *(const half8 *)(stg + pl * RS + ((c ^ (pl & (CH - 1))) << 4));
                        int opix, rpix;
                        bool ok;
                        if constexpr (EP_AHEAD) { const int sj = (t0 / TR) * (TR * 16 * CPR / 64) + i; opix = opx_tile[sj]; ok = ((okm >> sj) & 1u) & !(ablate & 2); }
                        else ok = po.index(cur.m0 + row0 + t0 * 16 + pl, cur.n0 + col0 + c * 8, opix, rpix) & !(ablate & 2);
                        if (ok) *(half8 *)(e_out + opix) = v;
                    }
                }
            } else if constexpr (HB <= 64) {
                constexpr int RS = HB * 4, CH = RS / 16, TR = 4096 / (16 * RS), NI = TR * 16 * CPR / 64;
                static_assert(TM % TR == 0 && (TR * 16 * CPR) % 64 == 0, "staging rounds");
                // a round's shortcut values are requested one round AHEAD (they travel under the SiLU work, and their registers are not the ones a store
                // still reads -- hipcc waits for a pending store before a load may overwrite its data registers); masked lanes read the tensor's first bytes
                half8 rv[2][NI]; int op[2][NI]; bool ok[2][NI];
                auto request = [&](int t0, half8 (&rvx)[NI], int (&opx)[NI], bool (&okx)[NI]) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int e = i * 64 + lane, pl = e / CPR, c = e - pl * CPR;
                        int rpix;
                        if constexpr (EP_AHEAD) { const int sj = (t0 / TR) * NI + i; opx[i] = opx_tile[sj]; rpix = opx[i]; okx[i] = (okm >> sj) & 1u; }      // (shortcut geometry == output geometry: launch check)
                        else okx[i] = po.index(cur.m0 + row0 + t0 * 16 + pl, cur.n0 + col0 + c * 8, opx[i], rpix);
                        rvx[i] = *(const half8 *)(e_res + (okx[i] ? rpix : 0));
                    }
                };
                request(0, rv[0], op[0], ok[0]);
#pragma unroll
                for (int t0 = 0; t0 < TM; t0 += TR) {
                    const int cb = (t0 / TR) & 1;
                    if (t0 + TR < TM) request(t0 + TR, rv[cb ^ 1], op[cb ^ 1], ok[cb ^ 1]);
                    if (e_act) {
#pragma unroll
                        for (int tt = 0; tt < TR; ++tt)
#pragma unroll
                            for (int u = 0; u < TN; ++u) silu4(acc[t0 + tt][u]);
                    }
#pragma unroll
                    for (int tt = 0; tt < TR; ++tt)
#pragma unroll
                        for (int u = 0; u < TN; ++u) {
                            const floatx4 v = acc[t0 + tt][u];
                            const int pl = tt * 16 + r, c = u * 4 + q;
                            *(floatx4 *)(stg + pl * RS + ((c ^ (pl & (CH - 1))) << 4)) = v;
                        }
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int e = i * 64 + lane, pl = e / CPR, c = e - pl * CPR;
                        const floatx4 lo = *(const floatx4 *)(stg + pl * RS + (((2 * c) ^ (pl & (CH - 1))) << 4));
                        const floatx4 hi = *(const floatx4 *)(stg + pl * RS + (((2 * c + 1) ^ (pl & (CH - 1))) << 4));
                        const half8 x = rv[cb][i];
                        const half8 o = {(f16)(lo[0] + (float)x[0]), (f16)(lo[1] + (float)x[1]), (f16)(lo[2] + (float)x[2]), (f16)(lo[3] + (float)x[3]),
                                         (f16)(hi[0] + (float)x[4]), (f16)(hi[1] + (float)x[5]), (f16)(hi[2] + (float)x[6]), (f16)(hi[3] + (float)x[7])};
                        if (ok[cb][i]) *(half8 *)(e_out + op[cb][i]) = o;
                    }
                }
            } else {
                // a shortcut with more than 64 couts per wave (the 192-wide tile): no epilogue exists for it.  launch_conv3x3_pp and tile_legal refuse the combination;
                // a caller that gets past both (a probe including this file directly) must not leave the output tensor silently unwritten
                __builtin_trap();
            }
        }
        if (have_next) __builtin_amdgcn_s_barrier();     // every wave is done with its staging area: the next tile's strips may land there
        if (tile_no == 0) STAMP(7);
        ++tile_no;
        if (!have_next) break;
        cur = nxt;
        have_next = next_tile(tj++, nxt);
    }
    STAMP(9);
}


// ---------------------------------------------------------------------------------------
// conv_tile_pp: the same two-halves-in-anti-phase schedule for the convs WITHOUT tap reuse -- 1x1 (C2f.cv1 / cv2, SPPF, incl. the neck layers that
// read their upsampled channels from the half-resolution tensor) and 3x3 / stride 2 (layers 5, 7, 16, 19): 256 pixels x BN couts per tile, one phase =
// one 64-deep k-step (kh, kw, 64 channels): 32 pixel pieces + BN / 8 weight pieces by LDS-DMA, 4 x TN x 2 MFMA tiles per wave.  Three-slot rings for
// both operands, the stream two phases ahead of the multiplication: phase j issues phase j + 2 into the slots phase j - 1 has just left and waits for
// ITS pieces of phase j + 1 (all but the ones just issued).  One problem per launch, persistent workgroups (tiles g, g + G, ...), the stream runs on
// into the next tile.  These layers have 3 - 36 phases per tile, so a tile's fill and epilogue weigh as much as its k-loop: what the kernel buys there
// is the cross-tile prefetch and the 16-byte line-wide stores of the shared epilogue.
// ---------------------------------------------------------------------------------------
template <int BN>
struct PtileCfg {
    static constexpr int BM = 256, NW = 8;
    static constexpr int NAP = BM / 8, ASLOT = NAP * 1024;          // 32 pieces of 8 pixels x 128 B
    static constexpr int NBP = BN / 8, BSLOT = NBP * 1024;
    static constexpr int NS = 3;                                    // ring slots per operand
    static constexpr int B_OFF = NS * ASLOT;
    static constexpr int LDS_BYTES = B_OFF + NS * BSLOT;
    static constexpr int LA = NAP / NW, LB = NBP / NW, L = LA + LB;
    static constexpr int TM = 4, TN = BN / 32;
    static_assert(NBP % NW == 0 && LDS_BYTES <= 160 * 1024, "tile");
};

template <int BN>
__global__ __launch_bounds__(512) void conv_tile_pp(ConvArgs p, int n_tiles) {
    using C = PtileCfg<BN>;
    constexpr int BM = C::BM, NW = C::NW, LA = C::LA, LB = C::LB, L = C::L, TM = C::TM, TN = C::TN, NS = C::NS;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[C::LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int half = wave >> 2, wm = wave >> 1, wn = wave & 1;
    auto slot_of = [](int R, int c) { return ((c + 2 * (R >> 1)) & 7) << 4; };
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        a_rd[kk] = wm * 64 * 128 + r * 128 + slot_of(r, kk * 4 + q);
        b_rd[kk] = C::B_OFF + wn * (BN / 2) * 128 + r * 128 + slot_of(r, kk * 4 + q);
    }
    const int ld_row = lane >> 3, ld_slot = lane & 7;
    const int ld_chunk = ((ld_slot - 2 * (ld_row >> 1)) & 7) * 8;

    const int gx = (p.M + BM - 1) / BM, gy = (p.cout + BN - 1) / BN;
    const int nph = p.ks * p.ks * (p.cin / 64);      // phases per tile (>= 2: checked at launch)
    const int G = gridDim.x;
    // Per-lane DMA offsets of a tile (bytes): its pixel rows in `in` / in the half-resolution source, its weight rows.  Kept as plain arrays and copied
    // element by element: a struct of them that is assigned under a branch, or passed by reference to a lambda that may see one of two, ends up in scratch
    // memory and is reloaded -- with s_waitcnt vmcnt(0) -- in every phase.
    unsigned ia_off[LA], ia_lo[LA], ib_off[LB];      // the tile the DMA stream is feeding
    unsigned na_off[LA], na_lo[LA], nb_off[LB];      // the tile after the one being multiplied
    int cur_m0, cur_n0, nxt_m0 = 0, nxt_n0 = 0;
    auto locate = [&](int id, int &m0, int &n0, unsigned (&ao)[LA], unsigned (&al)[LA], unsigned (&bo)[LB]) __attribute__((always_inline)) {
        const int by = id / gx, bx = id - by * gx;
        int mt, nt;
        xcd_tile(gx, gy, bx, by, mt, nt);
        m0 = mt * BM; n0 = nt * BN;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int m = m0 + (wave + NW * i) * 8 + ld_row;
            ao[i] = (unsigned)(input_offset(p, m) + ld_chunk) * 2u;
            al[i] = p.in2 ? (unsigned)(input_offset_lo(p, m) + ld_chunk) * 2u : 0u;
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) bo[j] = (unsigned)((n0 + (wave + NW * j) * 8 + ld_row) * p.kp + ld_chunk) * 2u;
    };
    auto dma = [&](const char *base, unsigned off, unsigned char *dst) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off), (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    // the DMA stream's cursor: (tile, k-step) of the next phase to issue
    struct Cursor { int kh, kw, c0, k; };
    auto issue = [&](Cursor &c, int slot) __attribute__((always_inline)) {
        const char *wb = (const char *)p.wt + (size_t)c.k * 128;
#pragma unroll
        for (int j = 0; j < LB; ++j) dma(wb, ib_off[j], lds + C::B_OFF + slot * C::BSLOT + (wave + NW * j) * 1024);
        // the upsampled half of a neck concat is read where it was produced (per-element select: two arrays behind one branch would again be indexed in memory)
        const bool lo = p.in2 && c.c0 < p.split;
        const char *ab = lo ? (const char *)p.in2 + (size_t)c.c0 * 2 : (const char *)p.in + (size_t)(((c.kh * p.in_Wp + c.kw) * p.in_cs + c.c0) * 2);
#pragma unroll
        for (int i = 0; i < LA; ++i) dma(ab, lo ? ia_lo[i] : ia_off[i], lds + slot * C::ASLOT + (wave + NW * i) * 1024);
        ++c.k; c.c0 += 64;
        if (c.c0 >= p.cin) { c.c0 = 0; if (++c.kw == p.ks) { c.kw = 0; ++c.kh; } }
    };

    int tid = blockIdx.x;
    if (tid >= n_tiles) return;
    // The DMA stream feeds the tile being multiplied, or -- for its last two phases -- already the next one (a tile has >= 3 phases, so the stream
    // never needs a tile that has not been located yet).
    locate(tid, cur_m0, cur_n0, ia_off, ia_lo, ib_off);
    bool have_next = tid + G < n_tiles;
    if (have_next) locate(tid + G, nxt_m0, nxt_n0, na_off, na_lo, nb_off);
    Cursor ic{0, 0, 0, 0};
    bool iss_ahead = false, stream_open = true;
    floatx4 bnext[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) bnext[u] = *(const floatx4 *)(p.bias + cur_n0 + (wn * TN + u) * 16 + q * 4);
    int islot = 0;                                   // ring slot of the next phase to issue
    auto issue_next = [&]() __attribute__((always_inline)) -> bool {      // issues one phase if the stream has one left
        if (!stream_open) return false;
        issue(ic, islot);
        islot = islot + 1 == NS ? 0 : islot + 1;
        if (ic.k == nph) {                           // that was the tile's last phase: on to the next tile, or the stream ends
            if (!iss_ahead && have_next) {
                ic = Cursor{0, 0, 0, 0}; iss_ahead = true;
#pragma unroll
                for (int i = 0; i < LA; ++i) { ia_off[i] = na_off[i]; ia_lo[i] = na_lo[i]; }
#pragma unroll
                for (int j = 0; j < LB; ++j) ib_off[j] = nb_off[j];
            } else stream_open = false;
        }
        return true;
    };
    issue_next();
    issue_next();
    pp_wait<L>();                                    // phase 0 has landed, phase 1 may be in flight
    __builtin_amdgcn_s_barrier();
    int rslot = 0;

    while (true) {
        floatx4 acc[TM][TN];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u) acc[t][u] = bnext[u];
        // the tile's store offsets, in front of the k-loop (as in conv3x3_pp; no pad columns among this kernel's positions, so a store is valid iff m < M and n < cout)
        constexpr int EP_CPR = BN / 16, EP_KS = 64 / EP_CPR;
        int opx_tile[EP_CPR];
        unsigned okm = 0;
        {
            const PpOut po = pp_out(p, p.Wo, p.d_howo, p.d_wo);
            const int L = lane / EP_CPR, cc = lane - L * EP_CPR;
            const int n = cur_n0 + wn * (BN / 2) + cc * 8;
            int m = cur_m0 + wm * TM * 16 + L;
            int b = fdiv(m, po.d_img), rem = m - (int)__umul24((unsigned)b, (unsigned)po.HW);
            int oy = fdiv(rem, po.d_row), x = rem - (int)__umul24((unsigned)oy, (unsigned)po.wq);
            const bool nok = n < po.cout;
            if (po.wq >= EP_KS) {
                int off = po.o0 + (int)__umul24((unsigned)b, (unsigned)po.o1) + (int)__umul24((unsigned)oy, (unsigned)po.o2) + (int)__umul24((unsigned)x, (unsigned)po.ocs) + n;
                const int dx = EP_KS * po.ocs, drow = po.o2 - po.wq * po.ocs, dimg = po.o1 - po.Ho * po.o2;
#pragma unroll
                for (int j = 0; j < EP_CPR; ++j) {
                    opx_tile[j] = off;
                    okm |= (unsigned)((m < po.M) & nok) << j;
                    m += EP_KS; x += EP_KS; off += dx;
                    if (x >= po.wq) { x -= po.wq; off += drow; oy += 1; if (oy >= po.Ho) { oy = 0; off += dimg; } }
                }
            } else {
#pragma unroll
                for (int j = 0; j < EP_CPR; ++j) {
                    int rp;
                    okm |= (unsigned)po.index(m + j * EP_KS, n, opx_tile[j], rp) << j;
                }
            }
        }
        if (half == 1) __builtin_amdgcn_s_barrier();  // STAGGER
        for (int j = 0; j < nph; ++j) {
            // ---------------- R ----------------
            const bool issued = issue_next();
            half8 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int u = 0; u < TN; ++u) fb[kk][u] = *(const half8 *)(lds + rslot * C::BSLOT + u * 2048 + b_rd[kk]);
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[kk][t] = *(const half8 *)(lds + rslot * C::ASLOT + t * 2048 + a_rd[kk]);
            }
            if (issued) pp_wait<L>(); else pp_wait<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // ---------------- M ----------------
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[kk][u], fa[kk][t], acc[t][u], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            rslot = rslot + 1 == NS ? 0 : rslot + 1;
        }
        if (half == 0) __builtin_amdgcn_s_barrier();  // BALANCE
        const int stg_slot = rslot == 0 ? NS - 1 : rslot - 1;      // the slot of the phase just multiplied: free until the next tile's first R refills it
        if (have_next) {
            int nn0 = nxt_n0;
            asm volatile("" : "+v"(nn0));
            nn0 = __builtin_amdgcn_readfirstlane(nn0);
#pragma unroll
            for (int u = 0; u < TN; ++u) bnext[u] = *(const floatx4 *)(p.bias + nn0 + (wn * TN + u) * 16 + q * 4);
        }
        {
            unsigned char *stg = lds + stg_slot * C::ASLOT + wave * 4096;
            constexpr int HB = BN / 2, CPR = HB / 8;
            constexpr int RS = HB <= 32 ? 64 : (HB <= 64 ? 128 : 256), CH = RS / 16, TR = 4096 / (16 * RS);
            static_assert(TM % TR == 0 && (TR * 16 * CPR) % 64 == 0, "staging rounds");
            f16 *const e_out = p.out;
            const int e_act = p.act;
            asm volatile("" ::"s"(e_out), "s"(e_act));
#pragma unroll
            for (int t0 = 0; t0 < TM; t0 += TR) {
                if (e_act) {
#pragma unroll
                    for (int tt = 0; tt < TR; ++tt)
#pragma unroll
                        for (int u = 0; u < TN; ++u) silu4(acc[t0 + tt][u]);
                }
#pragma unroll
                for (int tt = 0; tt < TR; ++tt)
#pragma unroll
                    for (int u = 0; u < TN; ++u) {
                        const floatx4 v = acc[t0 + tt][u];
                        const int pl = tt * 16 + r, c = u * 2 + (q >> 1);
                        *(half4 *)(stg + pl * RS + ((c ^ (pl & (CH - 1))) << 4) + (q & 1) * 8) = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    }
#pragma unroll
                for (int i = 0; i < TR * 16 * CPR / 64; ++i) {
                    const int e = i * 64 + lane, pl = e / CPR, c = e - pl * CPR;
                    const half8 v = *(const half8 *)(stg + pl * RS + ((c ^ (pl & (CH - 1))) << 4));
                    const int sj = (t0 / TR) * (TR * 16 * CPR / 64) + i;
                    (void)pl; (void)c;
                    if ((okm >> sj) & 1u) *(half8 *)(e_out + opx_tile[sj]) = v;
                }
            }
        }
        if (!have_next) break;
        __builtin_amdgcn_s_barrier();                 // every wave is done with its staging area
        cur_m0 = nxt_m0; cur_n0 = nxt_n0;
        tid += G;
        have_next = tid + G < n_tiles;
        if (have_next) locate(tid + G, nxt_m0, nxt_n0, na_off, na_lo, nb_off);
        iss_ahead = false;                            // (the stream is on this tile, two phases in)
    }
}

// ---- host side ----
template <int BN, bool WIDE = false>
static int launch_pp_bn(const ConvArgs *a, int n, hipStream_t s) {
    using C = PpCfg<BN, WIDE>;
    ConvGroupArgs g;
    g.n = n; g.start[0] = 0;
    for (int i = 0; i < n; ++i) {
        g.p[i] = a[i];
        g.gx[i] = cdiv(a[i].M, C::BM);
        g.start[i + 1] = (int)align_up((size_t)g.start[i] + (size_t)g.gx[i] * cdiv(a[i].cout, BN), 8);
    }
    for (int i = n; i < MAX_GROUP; ++i) { g.start[i + 1] = g.start[n]; g.gx[i] = 1; }
    const int total = g.start[n];
    // one persistent workgroup per CU (its LDS and registers leave no room for a second).  Diagnostic builds: RTMODT_PP_RESERVE=n leaves n CUs to the
    // launches of the engine's other stages for the whole life of this launch.
    static const int reserve = rt_diag("PP_RESERVE") ? atoi(rt_diag("PP_RESERVE")) : 0;
    int G = std::min(total, std::max(8, device_cus() - reserve));
    if (G >= 8) G &= ~7;                              // a workgroup's ids keep their residue mod 8: one XCD's share of the tile order
    // Balanced tile lists (pp_lpt_schedule) for launches whose workgroups run more than one tile of DIFFERENT problems: built once per launch shape and
    // device, kept for the life of the process (a few KiB each).  A shape first seen while its stream is being captured runs on the static stride.
    const int *sched = nullptr; int sched_T = 0;
    if (n > 1 && total > G && G <= 1024) {
        int dev = 0;
        RT_HIP(hipGetDevice(&dev));
        std::string key = std::to_string(dev) + ":" + std::to_string(BN) + (WIDE ? "w:" : ":") + std::to_string(G);
        for (int i = 0; i < n; ++i) key += ":" + std::to_string(g.gx[i] * cdiv(a[i].cout, BN)) + "x" + std::to_string(a[i].cin);
        static std::mutex mu;
        static std::map<std::string, std::pair<int *, int>> plans;
        std::lock_guard<std::mutex> lock(mu);
        auto it = plans.find(key);
        if (it == plans.end()) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (s) RT_HIP(hipStreamIsCapturing(s, &cs));
            if (cs == hipStreamCaptureStatusNone) {
                int tiles[MAX_GROUP]; long cost[MAX_GROUP];
                for (int i = 0; i < n; ++i) {
                    tiles[i] = g.gx[i] * cdiv(a[i].cout, BN);
                    cost[i] = (long)(9 * (a[i].cin / 64)) * (BN == 64 && !WIDE ? 1000 : (BN <= 128 ? 1250 : 1750)) + (BN == 64 && !WIDE ? 3500 : (BN <= 128 ? 5000 : 8000));      // phases x clk per phase + epilogue
                }
                const int max_T = total / (G >= 8 ? 8 : 1) + 1;      // a whole XCD share on one workgroup at worst
                std::vector<int> wide((size_t)G * max_T);
                const int T = pp_lpt_schedule(n, g.start, tiles, cost, G, wide.data(), max_T);
                std::pair<int *, int> plan{nullptr, 0};
                if (T > 0) {
                    std::vector<int> tab((size_t)G * T);
                    for (int w = 0; w < G; ++w)
                        for (int j = 0; j < T; ++j) tab[(size_t)w * T + j] = wide[(size_t)w * max_T + j];
                    RT_HIP(hipMalloc(&plan.first, tab.size() * sizeof(int)));
                    RT_HIP(hipMemcpy(plan.first, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
                    plan.second = T;
                }
                it = plans.emplace(key, plan).first;
            }
        }
        if (it != plans.end()) { sched = it->second.first; sched_T = it->second.second; }
        else if (rt_opt("TUNE_LOG")) fprintf(stderr, "[pp] grouped launch %s first seen while its stream is capturing: this capture keeps the static tile stride (no balanced schedule)\n", key.c_str());
    }
    hipLaunchKernelGGL((conv3x3_pp<BN, WIDE>), dim3(G), dim3(512), 0, s, g, total, sched, sched_T);
    return RTMODT_OK;
}

template <int BN>
static int launch_tile_pp_bn(const ConvArgs &a, hipStream_t s) {
    const int n_tiles = cdiv(a.M, 256) * cdiv(a.cout, BN);
    static const int reserve = rt_diag("PP_RESERVE") ? atoi(rt_diag("PP_RESERVE")) : 0;
    int G = std::min(n_tiles, std::max(8, device_cus() - reserve));
    if (G >= 8) G &= ~7;
    hipLaunchKernelGGL((conv_tile_pp<BN>), dim3(G), dim3(512), 0, s, a, n_tiles);
    return RTMODT_OK;
}

int launch_conv_tile_pp(const ConvArgs &a, int bn, hipStream_t s) {
    RT_CHECK(a.cin % 64 == 0 && a.kp == a.K && a.ks * a.ks * (a.cin / 64) >= 3 && !a.out2 && !a.res, RTMODT_E_INVALID,
             "launch_conv: the ping-pong tile kernel runs one conv with cin %% 64 == 0 and K >= 192, no residual or second destination "
             "(cin %d, ks %d, kp %d, K %d, residual %d, second destination %d)", a.cin, a.ks, a.kp, a.K, a.res != nullptr, a.out2 != nullptr);      // (a tail attached to the launch is ignored, as by every non-tail tile)
    RT_CHECK(a.cout % 8 == 0 && (uintptr_t)a.out % 16 == 0 && a.out_cs % 8 == 0, RTMODT_E_INVALID,
             "launch_conv: the ping-pong tile kernel stores 16 bytes per lane (cout, channel offsets and strides %% 8 == 0)");
    RT_CHECK(pp_out_fits(a), RTMODT_E_INVALID, "launch_conv: the ping-pong tile kernel indexes its output with 24-bit multiplies (padded H x W x C of a tensor < 2^24)");
    switch (bn) {
        case 128: return launch_tile_pp_bn<128>(a, s);
        case 64: return launch_tile_pp_bn<64>(a, s);
        default: return fail(RTMODT_E_INVALID, "launch_conv: ping-pong tile kernel with BN %d", bn);
    }
}

int launch_conv3x3_pp(const ConvArgs *a, int n, int bn, hipStream_t s) {
    for (int i = 0; i < n; ++i) {
        RT_CHECK(a[i].ks == 3 && a[i].stride == 1 && a[i].cin % 64 == 0 && a[i].kp % 64 == 0 && !a[i].in2 && !a[i].out2, RTMODT_E_INVALID,
                 "launch_conv: the ping-pong tile runs 3x3 stride-1 convs with cin %% 64 == 0, no second destination");
        // 16-byte stores (and residual loads): channel offsets and pixel strides in multiples of 8 halves
        RT_CHECK(a[i].cout % 8 == 0 && (uintptr_t)a[i].out % 16 == 0 && a[i].out_cs % 8 == 0 && (!a[i].res || ((uintptr_t)a[i].res % 16 == 0 && a[i].res_cs % 8 == 0)),
                 RTMODT_E_INVALID, "launch_conv: the ping-pong tile stores 16 bytes per lane (cout, channel offsets and strides %% 8 == 0)");
        RT_CHECK(!a[i].res || bn <= 128 || bn == 576, RTMODT_E_INVALID, "launch_conv: the 192-wide ping-pong tile takes no residual");
        RT_CHECK(!a[i].res || (a[i].res_Hp == a[i].out_Hp && a[i].res_Wp == a[i].out_Wp && a[i].res_cs == a[i].out_cs && a[i].res_pad == a[i].out_pad), RTMODT_E_INVALID,
                 "launch_conv: the ping-pong tile reads its shortcut at the output's element offsets (a channel slice of the same tensor geometry)");
        RT_CHECK(pp_out_fits(a[i]), RTMODT_E_INVALID, "launch_conv: the ping-pong tile indexes its output with 24-bit multiplies (padded H x W x C of a tensor < 2^24)");
    }
    switch (bn) {
        case 128: return launch_pp_bn<128>(a, n, s);
        case 64: return launch_pp_bn<64>(a, n, s);
        case 192: return launch_pp_bn<192>(a, n, s);
        case 512 + 64: return launch_pp_bn<64, true>(a, n, s);      // (bn 576 = the wide form: 512 positions x 64 couts)
        default: return fail(RTMODT_E_INVALID, "launch_conv: ping-pong tile with BN %d", bn);
    }
}

}  // namespace rtmodt
