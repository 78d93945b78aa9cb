// tracker.hip -- the reference's _ByteTrackCore (src/tracking/tracker.py:43-194) as ONE
// kernel launch per frame: one 1024-thread workgroup (16 wave64s) per video stream, the
// whole update -- hi/lo split, IoU, row arg-max, assignment, commit, second pass, spawn,
// ageing, expiry -- in LDS.  Integer results (ids, ages, time_since_update, list order)
// and float32 bit patterns are identical to the reference's NumPy evaluation.
//
//   iou_ref            <- _batch_iou          tracker.py:150-161  (never materialised as a
//                                              matrix here: fused into the row arg-max)
//   assoc_pass         <- _linear_assignment  tracker.py:182-194  greedy branch, in its
//                         order-free form: row i is matched iff it is the SMALLEST row among
//                         those whose first-arg-max column is j and whose value >= thresh
//                         (oracle/tracker_oracle.py:assign_greedy_parallel proves the equivalence)
//   assoc_lap          <- _linear_assignment  tracker.py:168-181  lap.lapjv branch (PARITY UNPINNED:
//                         `lap` is not installed anywhere this runs; restated from its published
//                         algorithm, see the comment at assoc_lap)
//   tracker_update     <- _ByteTrackCore.update tracker.py:58-141, _age_tracks :144-148
//
// Build with -ffp-contract=off: every float op below must round separately, exactly like
// NumPy's float32 ufuncs (no FMA contraction of `a + b - w*h`).
#include "kernels.h"

#include <climits>

namespace rtmodt {

#pragma clang fp contract(off)

__device__ __forceinline__ float iou_ref(const float4 a, const float4 b) {
    float x1 = fmaxf(a.x, b.x), y1 = fmaxf(a.y, b.y);
    float x2 = fminf(a.z, b.z), y2 = fminf(a.w, b.w);
    float w = fmaxf(0.0f, x2 - x1), h = fmaxf(0.0f, y2 - y1);
    float inter = w * h;
    float area_a = (a.z - a.x) * (a.w - a.y);
    float area_b = (b.z - b.x) * (b.w - b.y);
    float uni = (area_a + area_b) - inter;
    return inter / (uni + 1e-6f);
}

// ---------------------------------------------------------------------------------------
// Opt-in Kalman motion model (TrackerArgs::kalman; oracle/kalman_oracle.py states the algorithm and its provenance:
// ByteTrack's published 8-state constant-velocity filter -- the reference itself has none, tracker.py:99-104).
// F, H, Q, R and the initial P keep P block-diagonal, so the filter is four independent (position, velocity) pairs
// with covariance [[a, b], [b, c]] whose only coupling is the noise scale h.  float32, one rounding per operation in
// exactly the oracle's order (this file is built with FMA contraction off and correctly rounded division).
// ---------------------------------------------------------------------------------------
struct Kf { float4 pos, vel, pa, pb, pc; };
constexpr float KF_WP = 0.05f, KF_WV = 0.00625f;
__device__ __forceinline__ Kf kf_load(const float4 *kf, int Mc, int i) { return Kf{kf[i], kf[Mc + i], kf[2 * Mc + i], kf[3 * Mc + i], kf[4 * Mc + i]}; }
__device__ __forceinline__ void kf_store(float4 *kf, int Mc, int i, const Kf &k) {
    kf[i] = k.pos; kf[Mc + i] = k.vel; kf[2 * Mc + i] = k.pa; kf[3 * Mc + i] = k.pb; kf[4 * Mc + i] = k.pc;
}
__device__ __forceinline__ float4 xyxy_to_xyah(const float4 b) {
    const float w = b.z - b.x, h = b.w - b.y;
    return float4{b.x + w * 0.5f, b.y + h * 0.5f, w / fmaxf(h, 1e-6f), h};
}
__device__ __forceinline__ float4 xyah_to_xyxy(const float4 m) {
    const float w = m.z * m.w;
    const float x1 = m.x - w * 0.5f, y1 = m.y - m.w * 0.5f;
    return float4{x1, y1, x1 + w, y1 + m.w};
}
__device__ __forceinline__ Kf kf_initiate(const float4 z) {
    const float sp = (2.0f * KF_WP) * z.w, sv = (10.0f * KF_WV) * z.w;
    const float p2 = sp * sp, v2 = sv * sv;
    Kf k;
    k.pos = z; k.vel = float4{0.f, 0.f, 0.f, 0.f};
    k.pa = float4{p2, p2, 1e-2f * 1e-2f, p2};
    k.pb = float4{0.f, 0.f, 0.f, 0.f};
    k.pc = float4{v2, v2, 1e-5f * 1e-5f, v2};
    return k;
}
__device__ __forceinline__ void kf_predict1(float &p, const float v, float &a, float &b, float &c, const float qp, const float qv) {
    const float a0 = a, b0 = b, c0 = c;
    p = p + v;
    a = ((a0 + (b0 + b0)) + c0) + qp;
    b = b0 + c0;
    c = c0 + qv;
}
__device__ __forceinline__ void kf_predict(Kf &k) {
    const float h = k.pos.w;
    const float sp = KF_WP * h, sv = KF_WV * h;
    const float qp = sp * sp, qv = sv * sv;
    kf_predict1(k.pos.x, k.vel.x, k.pa.x, k.pb.x, k.pc.x, qp, qv);
    kf_predict1(k.pos.y, k.vel.y, k.pa.y, k.pb.y, k.pc.y, qp, qv);
    kf_predict1(k.pos.z, k.vel.z, k.pa.z, k.pb.z, k.pc.z, 1e-2f * 1e-2f, 1e-5f * 1e-5f);
    kf_predict1(k.pos.w, k.vel.w, k.pa.w, k.pb.w, k.pc.w, qp, qv);
}
__device__ __forceinline__ void kf_update1(float &p, float &v, float &a, float &b, float &c, const float z, const float r) {
    const float a0 = a, b0 = b, c0 = c;
    const float s = a0 + r;
    const float k0 = a0 / s, k1 = b0 / s;
    const float y = z - p;
    p = p + k0 * y;
    v = v + k1 * y;
    a = a0 - k0 * a0;
    b = b0 - k0 * b0;
    c = c0 - k1 * b0;
}
__device__ __forceinline__ void kf_update(Kf &k, const float4 z) {
    const float sp = KF_WP * k.pos.w;
    const float r = sp * sp;
    kf_update1(k.pos.x, k.vel.x, k.pa.x, k.pb.x, k.pc.x, z.x, r);
    kf_update1(k.pos.y, k.vel.y, k.pa.y, k.pb.y, k.pc.y, z.y, r);
    kf_update1(k.pos.z, k.vel.z, k.pa.z, k.pb.z, k.pc.z, z.z, 1e-1f * 1e-1f);
    kf_update1(k.pos.w, k.vel.w, k.pa.w, k.pb.w, k.pc.w, z.w, r);
}

constexpr int TRK_THREADS = 1024;
constexpr int TRK_WAVES = TRK_THREADS / 64;

// exclusive prefix of a per-thread flag over the workgroup (thread order); returns position,
// writes the total.  Two barriers.  wsum: LDS int[TRK_WAVES + 1].
__device__ __forceinline__ int block_scan_flag(bool flag, int *wsum, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long m = __ballot(flag);
    int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TRK_WAVES; ++w) {
        int v = wsum[w];
        if (w < wave) off += v;
        tot += v;
    }
    __syncthreads();
    total = tot;
    return off + within;
}

struct AssocSmem {
    float4 *tbox;      // [max_tracks] track boxes (state before this frame's commits)
    float4 *dbox;      // [max_dets]
    int *row_best;     // [max_tracks] first-arg-max column (position in the column list) or -1
    int *col_winner;   // [max_dets]   smallest passing row per column, INT_MAX = free
};

// One association pass.  rows: n_rows track indices (rows == nullptr -> identity);
// cols: n_cols detection indices.  Afterwards row r is matched iff
// row_best[r] >= 0 && col_winner[row_best[r]] == r.
// R = 2^k lanes cooperate on a row (R chosen so that all rows are in flight at once when
// they fit): with hundreds of live tracks R is 1 -- one lane per row, no cross-lane traffic,
// detection boxes read as LDS broadcasts; with a handful of tracks a whole wave shares a
// row and finishes with a wave64 butterfly (max value, then lowest column).
__device__ __forceinline__ void assoc_pass(const AssocSmem &s, const int *rows, int n_rows, const int *cols, int n_cols,
                                           float thresh) {
    for (int c = threadIdx.x; c < n_cols; c += TRK_THREADS) s.col_winner[c] = INT_MAX;
    __syncthreads();
    int R = 64;
    while (R > 1 && (n_rows * R > TRK_THREADS || (R >> 1) >= n_cols)) R >>= 1;
    const int groups = TRK_THREADS / R;
    const int gid = threadIdx.x / R, sub = threadIdx.x & (R - 1);
    for (int r = gid; r < n_rows; r += groups) {
        const float4 tb = s.tbox[rows ? rows[r] : r];
        float bv = -INFINITY;
        int bc = INT_MAX;
        for (int c = sub; c < n_cols; c += R) {
            float v = iou_ref(tb, s.dbox[cols[c]]);
            if (bc == INT_MAX || v > bv) { bv = v; bc = c; }        // increasing c: first maximum wins
        }
        for (int d = R >> 1; d >= 1; d >>= 1) {                      // butterfly inside the R-lane group
            float ov = __shfl_xor(bv, d);
            int oc = __shfl_xor(bc, d);
            bool take = (oc != INT_MAX) && (bc == INT_MAX || ov > bv || (ov == bv && oc < bc));
            if (take) { bv = ov; bc = oc; }
        }
        if (sub == 0) {
            bool ok = (bc != INT_MAX) && (bv >= thresh);
            s.row_best[r] = ok ? bc : -1;
            if (ok) atomicMin(&s.col_winner[bc], r);
        }
    }
    __syncthreads();
}


// ---------------------------------------------------------------------------------------
// assoc_lap: the `lap.lapjv(1 - iou, extend_cost=True, cost_limit=1 - thresh)` branch
// (tracker.py:168-181).  lap embeds the M x N cost in an (M+N)^2 matrix whose off-diagonal
// blocks are cost_limit/2 and whose lower-right block is 0, so matching (i, j) costs c_ij while
// leaving both unmatched costs cost_limit: the optimum is the MAXIMUM-GAIN matching of the
// bipartite graph of pairs with gain cost_limit - c_ij > 0 (c_ij = double(float32(1 - iou))).
// At match_thresh 0.8 that graph is almost a perfect set of isolated edges, so:
//   1. every lane group scans a row, counts its candidate columns (row degree) and bumps the
//      column degrees (LDS atomics);
//   2. an edge whose row AND column have degree 1 is a connected component by itself and is
//      matched outright (its gain is positive);
//   3. the remaining "contested" rows/columns (<= LAP_ROWS / LAP_COLS, <= LAP_EDGES edges) are
//      compacted into LDS and solved EXACTLY by one lane with the shortest-augmenting-path
//      (Hungarian / Jonker-Volgenant) method on the sparse graph: every row has a private dummy
//      column of cost 0 (= stay unmatched), real edges cost c_ij - cost_limit < 0, potentials in
//      double.  Only the source row's edges can have negative reduced cost, so the Dijkstra
//      scan is valid; only touched columns are ever visited or reset.
// A scene denser than the LDS budget raises the sticky error 2 (host: RTMODT_E_CAPACITY).
// Output convention = assoc_pass: row r matched iff row_best[r] >= 0 && col_winner[row_best[r]] == r.
// ---------------------------------------------------------------------------------------
constexpr int LAP_ROWS = 256, LAP_COLS = 256, LAP_EDGES = 2048;
struct LapSmem {
    int *colmap;                   // [n_cols capacity] column -> local index among contested columns (-1 none, -2 marked)
    double *ecost, *u, *v, *minv;  // [LAP_EDGES], [LAP_ROWS], [LAP_COLS], [LAP_COLS]
    int *hrow, *hcol, *estart, *ecol;         // [LAP_ROWS], [LAP_COLS], [LAP_ROWS + 1], [LAP_EDGES]
    int *p, *rm, *wayrow, *touched, *usedl;   // col -> row, row -> col, col -> row it was reached from, lists
    unsigned char *used;           // [LAP_COLS]
};
static size_t lap_smem_bytes(int Nc) {
    return (size_t)LAP_EDGES * 12 + (size_t)LAP_ROWS * (8 + 4 + 4 + 4) + (size_t)LAP_COLS * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 1) + (size_t)Nc * 4 + 64;
}
__device__ __forceinline__ LapSmem lap_carve(unsigned char *base, int Nc) {     // base 8-byte aligned
    LapSmem L;
    L.ecost = (double *)base;
    L.u = L.ecost + LAP_EDGES;
    L.v = L.u + LAP_ROWS;
    L.minv = L.v + LAP_COLS;
    L.colmap = (int *)(L.minv + LAP_COLS);
    L.hrow = L.colmap + Nc;
    L.hcol = L.hrow + LAP_ROWS;
    L.estart = L.hcol + LAP_COLS;
    L.ecol = L.estart + LAP_ROWS + 1;
    L.p = L.ecol + LAP_EDGES;
    L.rm = L.p + LAP_COLS;
    L.wayrow = L.rm + LAP_ROWS;
    L.touched = L.wayrow + LAP_COLS;
    L.usedl = L.touched + LAP_COLS;
    L.used = (unsigned char *)(L.usedl + LAP_COLS);
    return L;
}

// exact sparse assignment of the contested sub-problem, run by ONE lane
__device__ void lap_solve(const LapSmem &L, int nhr) {
    const double INF = __builtin_huge_val();
    for (int h0 = 0; h0 < nhr; ++h0) {
        int nt = 0, nu = 0, i0 = h0, jend = -1, drow = -1;
        double dmin = INF;
        bool to_dummy = false;
        while (true) {
            const double ui = L.u[i0];
            for (int e = L.estart[i0]; e < L.estart[i0 + 1]; ++e) {          // relax the real edges of row i0
                const int j = L.ecol[e];
                if (L.used[j]) continue;
                const double cur = L.ecost[e] - ui - L.v[j];
                if (L.minv[j] == INF) L.touched[nt++] = j;
                if (cur < L.minv[j]) { L.minv[j] = cur; L.wayrow[j] = i0; }
            }
            if (0.0 - ui < dmin) { dmin = 0.0 - ui; drow = i0; }             // ... and its dummy edge
            double delta = dmin;
            int j1 = -1;
            for (int t = 0; t < nt; ++t) {
                const int j = L.touched[t];
                if (!L.used[j] && L.minv[j] < delta) { delta = L.minv[j]; j1 = j; }
            }
            L.u[h0] += delta;
            for (int t = 0; t < nu; ++t) { const int j = L.usedl[t]; L.u[L.p[j]] += delta; L.v[j] -= delta; }
            for (int t = 0; t < nt; ++t) { const int j = L.touched[t]; if (!L.used[j]) L.minv[j] -= delta; }
            dmin -= delta;
            if (j1 < 0) { to_dummy = true; break; }
            if (L.p[j1] < 0) { jend = j1; break; }
            L.used[j1] = 1;
            L.usedl[nu++] = j1;
            i0 = L.p[j1];
        }
        if (to_dummy) {                                   // row drow gives up its column; shift the path back to h0
            int i = drow, jfree = L.rm[i];
            L.rm[i] = -1;
            while (i != h0) {
                const int j = jfree, ip = L.wayrow[j];
                jfree = L.rm[ip];
                L.p[j] = ip;
                L.rm[ip] = j;
                i = ip;
            }
        } else {
            int j = jend;
            while (true) {
                const int ip = L.wayrow[j], jn = L.rm[ip];
                L.p[j] = ip;
                L.rm[ip] = j;
                if (ip == h0) break;
                j = jn;
            }
        }
        for (int t = 0; t < nt; ++t) { const int j = L.touched[t]; L.minv[j] = INF; L.used[j] = 0; }
    }
}

// val(r, c): IoU (float32) of row r and column c of this pass
template <typename F>
__device__ __forceinline__ void assoc_lap(F val, int n_rows, int n_cols, double limit, int *row_best, int *col_winner, int *rowcand,
                                          const LapSmem &L, int *wsum, int *err) {
    const int tid = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = tid; c < n_cols; c += TRK_THREADS) { col_winner[c] = 0; L.colmap[c] = -1; }
    __syncthreads();
    // ---- 1. degrees ----
    int R = 64;
    while (R > 1 && (n_rows * R > TRK_THREADS || (R >> 1) >= n_cols)) R >>= 1;
    const int groups = TRK_THREADS / R;
    const int gid = threadIdx.x / R, sub = threadIdx.x & (R - 1);
    for (int r = gid; r < n_rows; r += groups) {
        int deg = 0, last = -1;
        for (int c = sub; c < n_cols; c += R) {
            const float v = val(r, c);
            if ((double)(1.0f - v) < limit) { ++deg; last = c; atomicAdd(&col_winner[c], 1); }
        }
        for (int d = R >> 1; d >= 1; d >>= 1) {
            deg += __shfl_xor(deg, d);
            last = max(last, __shfl_xor(last, d));
        }
        if (sub == 0) { row_best[r] = deg; rowcand[r] = last; }
    }
    __syncthreads();
    // ---- 2. isolated edges vs contested rows (ascending) ----
    int nhr = 0;
    for (int base = 0; base < n_rows; base += TRK_THREADS) {
        const int r = base + tid;
        bool hard = false;
        if (r < n_rows) {
            const int deg = row_best[r];
            hard = deg >= 2 || (deg == 1 && col_winner[rowcand[r]] != 1);
            if (deg != 1 || hard) rowcand[r] = -1;          // rowcand >= 0 from here on == isolated edge
        }
        int tot;
        const int pos = block_scan_flag(hard, wsum, tot);
        if (hard && nhr + pos < LAP_ROWS) L.hrow[nhr + pos] = r;
        nhr += tot;
    }
    bool dense = nhr > LAP_ROWS;
    if (dense) nhr = 0;
    __syncthreads();                                       // hrow[] is written after the scan's own barriers
    if (tid == 0) {
        int e = 0;
        for (int h = 0; h < nhr; ++h) { L.estart[h] = e; e += row_best[L.hrow[h]]; }
        L.estart[nhr] = e;
    }
    __syncthreads();
    int ne = L.estart[nhr];
    if (ne > LAP_EDGES) { dense = true; nhr = 0; ne = 0; }
    // ---- 3. edges of the contested rows, columns ascending: one wave per row ----
    for (int h = wave; h < nhr; h += TRK_WAVES) {
        const int r = L.hrow[h];
        int e0 = L.estart[h];
        for (int cb = 0; cb < n_cols; cb += 64) {
            const int c = cb + lane;
            bool f = false;
            double cost = 0.0;
            if (c < n_cols) { cost = (double)(1.0f - val(r, c)); f = cost < limit; }
            const unsigned long long m = __ballot(f);
            if (f) {
                const int e = e0 + __popcll(m & ((1ull << lane) - 1ull));
                L.ecol[e] = c;
                L.ecost[e] = cost - limit;
                L.colmap[c] = -2;
            }
            e0 += __popcll(m);
        }
    }
    __syncthreads();
    int nhc = 0;
    for (int base = 0; base < n_cols; base += TRK_THREADS) {
        const int c = base + tid;
        const bool f = c < n_cols && L.colmap[c] == -2;
        int tot;
        const int pos = block_scan_flag(f, wsum, tot);
        if (f && nhc + pos < LAP_COLS) { L.colmap[c] = nhc + pos; L.hcol[nhc + pos] = c; }
        nhc += tot;
    }
    if (nhc > LAP_COLS) { dense = true; nhr = 0; ne = 0; nhc = 0; }
    __syncthreads();
    for (int e = tid; e < ne; e += TRK_THREADS) L.ecol[e] = L.colmap[L.ecol[e]];
    for (int h = tid; h < nhr; h += TRK_THREADS) { L.u[h] = 0.0; L.rm[h] = -1; }
    for (int j = tid; j < nhc; j += TRK_THREADS) { L.v[j] = 0.0; L.minv[j] = __builtin_huge_val(); L.p[j] = -1; L.used[j] = 0; }
    for (int c = tid; c < n_cols; c += TRK_THREADS) col_winner[c] = INT_MAX;
    __syncthreads();
    for (int r = tid; r < n_rows; r += TRK_THREADS) {
        const int c = rowcand[r];
        row_best[r] = c;
        if (c >= 0) col_winner[c] = r;
    }
    __syncthreads();
    // ---- 4. the contested sub-problem ----
    if (tid == 0) {
        if (dense) *err = 2;
        lap_solve(L, nhr);
        for (int h = 0; h < nhr; ++h)
            if (L.rm[h] >= 0) {
                const int r = L.hrow[h], c = L.hcol[L.rm[h]];
                row_best[r] = c;
                col_winner[c] = r;
            }
    }
    __syncthreads();
}

// one frame of one stream (tracker.py:58-141); `didx` = which detection slot feeds it.  Every thread of the workgroup
// returns together (all exits are on workgroup-uniform conditions).
__device__ __forceinline__ void tracker_step(const TrackerArgs &a, const int sidx, const int didx, unsigned char *smem) {
    const int Mc = a.max_tracks, Nc = a.max_dets;
    // LDS carve
    float4 *tbox = (float4 *)smem;
    float4 *dbox = tbox + Mc;
    float *dconf = (float *)(dbox + Nc);
    int *dcls = (int *)(dconf + Nc);
    int *hi_idx = dcls + Nc;
    int *lo_idx = hi_idx + Nc;
    int *sp_idx = lo_idx + Nc;
    int *col_winner = sp_idx + Nc;
    int *row_best = col_winner + Nc;
    int *um_t = row_best + Mc;
    int *t_matched = um_t + Mc;
    int *wsum = t_matched + Mc;
    const bool lapjv = a.assign_mode == RTMODT_ASSIGN_LAPJV;
    LapSmem lap{};
    if (lapjv) lap = lap_carve((unsigned char *)(((uintptr_t)(wsum + TRK_WAVES + 1) + 7) & ~(uintptr_t)7), Nc);
    __shared__ int lap_err;
    if (threadIdx.x == 0) lap_err = 0;

    TrackerState st = a.states[sidx];
    long long *meta = (long long *)a.meta + (size_t)sidx * 8;
    __shared__ int act_cnt;
    if (threadIdx.x == 0) act_cnt = 0;
    const int cur = (int)meta[0];
    const int M = (int)meta[1];
    const long long next_id = meta[4];
    // explicit selects (a runtime-indexed pointer array would live in scratch)
    int64_t *c_ids = cur ? st.ids[1] : st.ids[0];   int64_t *n_ids = cur ? st.ids[0] : st.ids[1];
    float4 *c_box = cur ? st.box[1] : st.box[0];    float4 *n_box = cur ? st.box[0] : st.box[1];
    float *c_conf = cur ? st.conf[1] : st.conf[0];  float *n_conf = cur ? st.conf[0] : st.conf[1];
    int32_t *c_cls = cur ? st.cls[1] : st.cls[0];   int32_t *n_cls = cur ? st.cls[0] : st.cls[1];
    int32_t *c_age = cur ? st.age[1] : st.age[0];   int32_t *n_age = cur ? st.age[0] : st.age[1];
    int32_t *c_tsu = cur ? st.tsu[1] : st.tsu[0];   int32_t *n_tsu = cur ? st.tsu[0] : st.tsu[1];
    float4 *c_kf = cur ? st.kf[1] : st.kf[0];       float4 *n_kf = cur ? st.kf[0] : st.kf[1];
    const bool kalman = a.kalman && c_kf != nullptr;
    int n = a.det_n[didx];                                // detections are indexed by absolute stream too (+ the frame's slot offset)
    if (n > Nc) n = Nc;                                   // host rejects this; belt and braces
    const int tid = threadIdx.x;

    if (kalman)                                            // every track moves on by one frame; a track that was not matched in
        for (int i = tid; i < M; i += TRK_THREADS) {       // the previous frame (tsu >= 2) coasts with its height velocity zeroed
            Kf k = kf_load(c_kf, Mc, i);
            if (c_tsu[i] >= 2) k.vel.w = 0.f;
            kf_predict(k);
            kf_store(c_kf, Mc, i, k);
            tbox[i] = xyah_to_xyxy(k.pos);                 // association sees the PREDICTED boxes
        }
    if (n == 0) {                                         // tracker.py:70-73: age only, nothing expires
        for (int i = tid; i < M; i += TRK_THREADS) c_tsu[i] += 1;
        if (tid == 0) meta[3] = 0;
        return;
    }

    const float4 *gb = a.det_box + (size_t)didx * a.det_stride;
    const float *gc = a.det_conf + (size_t)didx * a.det_stride;
    const int32_t *gk = a.det_cls + (size_t)didx * a.det_stride;
    for (int i = tid; i < n; i += TRK_THREADS) { dbox[i] = gb[i]; dconf[i] = gc[i]; dcls[i] = gk[i]; }
    if (!kalman) for (int i = tid; i < M; i += TRK_THREADS) tbox[i] = c_box[i];
    __syncthreads();

    // ---- 1. hi / lo split (tracker.py:76-85), input order preserved ----
    int nh = 0;
    for (int base = 0; base < n; base += TRK_THREADS) {
        int i = base + tid;
        bool hi = i < n && dconf[i] >= a.track_thresh;
        int tot;
        int pos = block_scan_flag(hi, wsum, tot);
        if (i < n) {
            if (hi) hi_idx[nh + pos] = i;
            else lo_idx[(i - (nh + pos))] = i;            // lows before i = i - highs before i
        }
        nh += tot;
    }
    const int nl = n - nh;
    __syncthreads();

    AssocSmem as{tbox, dbox, row_best, col_winner};

    // ---- 2. first association: ALL tracks x high detections (tracker.py:91-106) ----
    const bool pass1 = M > 0 && nh > 0;
    if (pass1) {
        if (lapjv) assoc_lap([&](int r, int c) { return iou_ref(tbox[r], dbox[hi_idx[c]]); }, M, nh, a.cost_limit, row_best, col_winner, t_matched, lap, wsum, &lap_err);
        else assoc_pass(as, nullptr, M, hi_idx, nh, a.match_thresh);
        for (int i = tid; i < M; i += TRK_THREADS) {
            int c = row_best[i];
            bool matched = c >= 0 && col_winner[c] == i;
            t_matched[i] = matched;
            if (matched) {                                 // tracker.py:99-104
                int d = hi_idx[c];
                c_box[i] = dbox[d];
                c_conf[i] = dconf[d];
                c_cls[i] = dcls[d];
                c_age[i] += 1;
                c_tsu[i] = 0;
                if (kalman) { Kf k = kf_load(c_kf, Mc, i); kf_update(k, xyxy_to_xyah(dbox[d])); kf_store(c_kf, Mc, i, k); }
            }
        }
    } else {
        for (int i = tid; i < M; i += TRK_THREADS) t_matched[i] = 0;
        for (int c = tid; c < nh; c += TRK_THREADS) col_winner[c] = INT_MAX;
    }
    __syncthreads();

    // spawn list = unmatched high detections, ascending (tracker.py:89,126): taken now,
    // before pass 2 reuses col_winner
    int nsp = 0;
    for (int base = 0; base < nh; base += TRK_THREADS) {
        int c = base + tid;
        bool f = c < nh && col_winner[c] == INT_MAX;
        int tot;
        int pos = block_scan_flag(f, wsum, tot);
        if (f) sp_idx[nsp + pos] = hi_idx[c];
        nsp += tot;
    }
    // unmatched tracks, ascending (tracker.py:109)
    int num = 0;
    for (int base = 0; base < M; base += TRK_THREADS) {
        int i = base + tid;
        bool f = i < M && !t_matched[i];
        int tot;
        int pos = block_scan_flag(f, wsum, tot);
        if (f) um_t[num + pos] = i;
        num += tot;
    }
    __syncthreads();

    // ---- 3. second association: unmatched tracks x low detections, SAME threshold (tracker.py:110-123) ----
    if (num > 0 && nl > 0) {
        if (lapjv) assoc_lap([&](int r, int c) { return iou_ref(tbox[um_t[r]], dbox[lo_idx[c]]); }, num, nl, a.cost_limit, row_best, col_winner, t_matched, lap, wsum, &lap_err);
        else assoc_pass(as, um_t, num, lo_idx, nl, a.match_thresh);
        for (int r = tid; r < num; r += TRK_THREADS) {
            int c = row_best[r];
            if (c >= 0 && col_winner[c] == r) {
                int i = um_t[r], d = lo_idx[c];
                c_box[i] = dbox[d];
                c_conf[i] = dconf[d];
                c_cls[i] = dcls[d];
                c_age[i] += 1;
                c_tsu[i] = 0;
                if (kalman) { Kf k = kf_load(c_kf, Mc, i); kf_update(k, xyxy_to_xyah(dbox[d])); kf_store(c_kf, Mc, i, k); }
            }
        }
    }

    // ---- 4. spawn (tracker.py:126-135) ----
    int err = 0;
    if (M + nsp > Mc) { err = 1; nsp = Mc - M; }
    for (int k = tid; k < nsp; k += TRK_THREADS) {
        int d = sp_idx[k], i = M + k;
        c_ids[i] = next_id + k;
        c_box[i] = dbox[d];
        c_conf[i] = dconf[d];
        c_cls[i] = dcls[d];
        c_age[i] = 1;
        c_tsu[i] = 0;
        if (kalman) kf_store(c_kf, Mc, i, kf_initiate(xyxy_to_xyah(dbox[d])));
    }
    const int M2 = M + nsp;
    __syncthreads();                                      // state writes visible to the whole workgroup

    // ---- 5. age every track, drop tsu > track_buffer, stable (tracker.py:138-139, :144-148) ----
    const int nxt = cur ^ 1;
    int kept = 0, active = 0;
    for (int base = 0; base < M2; base += TRK_THREADS) {
        int i = base + tid;
        int tsu = 0;
        bool keep = false;
        if (i < M2) { tsu = c_tsu[i] + 1; keep = tsu <= a.track_buffer; }
        int tot;
        int pos = block_scan_flag(keep, wsum, tot);
        if (keep) {
            int o = kept + pos;
            n_ids[o] = c_ids[i];
            n_box[o] = c_box[i];
            n_conf[o] = c_conf[i];
            n_cls[o] = c_cls[i];
            n_age[o] = c_age[i];
            n_tsu[o] = tsu;
            if (kalman) kf_store(n_kf, Mc, o, kf_load(c_kf, Mc, i));
        }
        if (keep && tsu == 0) atomicAdd(&act_cnt, 1);      // tracker.py:141 -- never true (SURVEY finding 4)
        kept += tot;
    }
    __syncthreads();
    active = act_cnt;
    if (tid == 0) {
        meta[0] = nxt;
        meta[1] = kept;
        if (err) meta[2] = 1;
        else if (lap_err) meta[2] = lap_err;
        meta[3] = active;
        meta[4] = next_id + nsp;
    }
}

// One workgroup per stream.  A detector batch that holds n_frames CONSECUTIVE frames of every stream (slot f * frame_step + s)
// is consumed in ONE launch: the workgroup walks its stream's frames in order -- tracker.py:58-141 still sees them one at a
// time -- with a workgroup barrier (and its workgroup-scope fence) between frames: the state and meta words frame f wrote
// are what frame f + 1 reads.  n_frames == 1 is the ordinary per-frame update.
__global__ __launch_bounds__(TRK_THREADS) void tracker_update(TrackerArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int sidx = a.stream_base + blockIdx.x;
    for (int f = 0; f < a.n_frames; ++f) {
        tracker_step(a, sidx, sidx + f * a.frame_step, smem);
        __syncthreads();
    }
}

static size_t tracker_smem_bytes(int Mc, int Nc) {
    return (size_t)Mc * 16 + (size_t)Nc * 16 + (size_t)Nc * 4 * 6 + (size_t)Mc * 4 * 3 + (TRK_WAVES + 1) * 4 + 64;
}

int launch_tracker_update(const TrackerArgs &a, hipStream_t s) {
    size_t smem = tracker_smem_bytes(a.max_tracks, a.max_dets) + (a.assign_mode == RTMODT_ASSIGN_LAPJV ? lap_smem_bytes(a.max_dets) + 8 : 0);
    RT_CHECK(smem <= 150 * 1024, RTMODT_E_INVALID, "tracker: max_tracks %d / max_dets %d need %zu B of LDS (> 160 KiB)", a.max_tracks,
             a.max_dets, smem);
    static DynLdsSeen seen;                                // the attribute is per device
    RT_TRY(raise_dynamic_lds((const void *)tracker_update, smem, seen));
    hipLaunchKernelGGL(tracker_update, dim3(a.n_streams), dim3(TRK_THREADS), smem, s, a);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

// ---- standalone pieces for the fixture-level parity tests (G1, G2) ----
__global__ void iou_matrix_kernel(const float4 *__restrict__ a, int m, const float4 *__restrict__ b, int n, float *__restrict__ out) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)m * n) return;
    int i = (int)(idx / n), j = (int)(idx - (long)i * n);
    out[idx] = iou_ref(a[i], b[j]);
}

int launch_iou_matrix(const float4 *a, int m, const float4 *b, int n, float *out, hipStream_t s) {
    long total = (long)m * n;
    if (total == 0) return RTMODT_OK;
    hipLaunchKernelGGL(iou_matrix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, m, b, n, out);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

// greedy assignment on a caller-supplied matrix: one workgroup, wave per row, same
// arg-max / atomicMin scheme as assoc_pass (col_winner lives in global memory here).
__global__ __launch_bounds__(TRK_THREADS) void assign_greedy_kernel(const float *__restrict__ iou, int m, int n, float thresh,
                                                                    int32_t *__restrict__ row_to_col, int32_t *__restrict__ col_winner) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < n; c += TRK_THREADS) col_winner[c] = INT_MAX;
    __syncthreads();
    for (int r = wave; r < m; r += TRK_WAVES) {
        float bv = -INFINITY;
        int bc = INT_MAX;
        for (int c = lane; c < n; c += 64) {
            float v = iou[(long)r * n + c];
            if (bc == INT_MAX || v > bv) { bv = v; bc = c; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            float ov = __shfl_xor(bv, d);
            int oc = __shfl_xor(bc, d);
            bool take = (oc != INT_MAX) && (bc == INT_MAX || ov > bv || (ov == bv && oc < bc));
            if (take) { bv = ov; bc = oc; }
        }
        if (lane == 0) {
            bool ok = (bc != INT_MAX) && (bv >= thresh);
            row_to_col[r] = ok ? bc : -1;
            if (ok) atomicMin(&col_winner[bc], r);
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < m; r += TRK_THREADS) {
        int c = row_to_col[r];
        if (c >= 0 && col_winner[c] != r) row_to_col[r] = -1;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += TRK_THREADS) col_winner[c] = col_winner[c] != INT_MAX;   // -> col_used
}

int launch_assign_greedy(const float *iou, int m, int n, float thresh, int32_t *row_to_col, int32_t *col_used, hipStream_t s) {
    hipLaunchKernelGGL(assign_greedy_kernel, dim3(1), dim3(TRK_THREADS), 0, s, iou, m, n, thresh, row_to_col, col_used);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}


// lapjv branch on a caller-supplied matrix: one workgroup; LDS = row_best[m] rowcand[m] col_winner[n] + LapSmem
__global__ __launch_bounds__(TRK_THREADS) void assign_lapjv_kernel(const float *__restrict__ iou, int m, int n, double limit,
                                                                   int32_t *__restrict__ row_to_col, int32_t *__restrict__ col_used,
                                                                   int32_t *__restrict__ err_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *row_best = (int *)smem, *rowcand = row_best + m, *col_winner = rowcand + m, *wsum = col_winner + n;
    LapSmem L = lap_carve((unsigned char *)(((uintptr_t)(wsum + TRK_WAVES + 1) + 7) & ~(uintptr_t)7), n);
    __shared__ int err;
    if (threadIdx.x == 0) err = 0;
    __syncthreads();
    assoc_lap([&](int r, int c) { return iou[(long)r * n + c]; }, m, n, limit, row_best, col_winner, rowcand, L, wsum, &err);
    for (int r = threadIdx.x; r < m; r += TRK_THREADS) {
        const int c = row_best[r];
        row_to_col[r] = (c >= 0 && col_winner[c] == r) ? c : -1;
    }
    for (int c = threadIdx.x; c < n; c += TRK_THREADS) col_used[c] = col_winner[c] != INT_MAX;
    if (threadIdx.x == 0) *err_out = err;
}

int launch_assign_lapjv(const float *iou, int m, int n, double cost_limit, int32_t *row_to_col, int32_t *col_used, int32_t *err,
                        hipStream_t s) {
    size_t smem = (size_t)(2 * m + n + TRK_WAVES + 1) * 4 + 8 + lap_smem_bytes(n);
    RT_CHECK(smem <= 150 * 1024, RTMODT_E_INVALID, "assign_lapjv: %d x %d needs %zu B of LDS", m, n, smem);
    static DynLdsSeen seen;
    RT_TRY(raise_dynamic_lds((const void *)assign_lapjv_kernel, smem, seen));
    hipLaunchKernelGGL(assign_lapjv_kernel, dim3(1), dim3(TRK_THREADS), smem, s, iou, m, n, cost_limit, row_to_col, col_used, err);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

}  // namespace rtmodt
