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

__global__ __launch_bounds__(TRK_THREADS) void tracker_update(TrackerArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int sidx = a.stream_base + blockIdx.x;
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
    int n = a.det_n[sidx];                                // detections are indexed by absolute stream too
    if (n > Nc) n = Nc;                                   // host rejects this; belt and braces
    const int tid = threadIdx.x;

    if (n == 0) {                                         // tracker.py:70-73: age only, nothing expires
        for (int i = tid; i < M; i += TRK_THREADS) c_tsu[i] += 1;
        if (tid == 0) meta[3] = 0;
        return;
    }

    const float4 *gb = a.det_box + (size_t)sidx * a.det_stride;
    const float *gc = a.det_conf + (size_t)sidx * a.det_stride;
    const int32_t *gk = a.det_cls + (size_t)sidx * a.det_stride;
    for (int i = tid; i < n; i += TRK_THREADS) { dbox[i] = gb[i]; dconf[i] = gc[i]; dcls[i] = gk[i]; }
    for (int i = tid; i < M; i += TRK_THREADS) tbox[i] = c_box[i];
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
        assoc_pass(as, nullptr, M, hi_idx, nh, a.match_thresh);
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
        assoc_pass(as, um_t, num, lo_idx, nl, a.match_thresh);
        for (int r = tid; r < num; r += TRK_THREADS) {
            int c = row_best[r];
            if (c >= 0 && col_winner[c] == r) {
                int i = um_t[r], d = lo_idx[c];
                c_box[i] = dbox[d];
                c_conf[i] = dconf[d];
                c_cls[i] = dcls[d];
                c_age[i] += 1;
                c_tsu[i] = 0;
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
        meta[3] = active;
        meta[4] = next_id + nsp;
    }
}

static size_t tracker_smem_bytes(int Mc, int Nc) {
    return (size_t)Mc * 16 + (size_t)Nc * 16 + (size_t)Nc * 4 * 6 + (size_t)Mc * 4 * 3 + (TRK_WAVES + 1) * 4 + 64;
}

int launch_tracker_update(const TrackerArgs &a, hipStream_t s) {
    size_t smem = tracker_smem_bytes(a.max_tracks, a.max_dets);
    RT_CHECK(smem <= 150 * 1024, RTMODT_E_INVALID, "tracker: max_tracks %d / max_dets %d need %zu B of LDS (> 160 KiB)", a.max_tracks,
             a.max_dets, smem);
    static size_t attr_bytes = 0;
    if (smem > attr_bytes) {
        RT_HIP(hipFuncSetAttribute((const void *)tracker_update, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_bytes = smem;
    }
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

}  // namespace rtmodt
