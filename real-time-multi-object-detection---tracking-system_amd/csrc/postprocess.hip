// postprocess.hip -- Detect-head decode and non-max suppression for gfx950.
//
// What they replace (third-party code the reference reaches through
// /root/reference/src/detection/detector.py:100-111, restated in SURVEY.md App. B):
//   decode_kernel        <- ultralytics Detect inference path: DFL softmax(16).arange,
//                           dist2bbox(xywh) x stride, sigmoid(cls)                      (B.2)
//   candidates           <- non_max_suppression steps 1-4: amax(cls) > conf (strict, f32),
//                           first-max class, classes filter, xywh->xyxy, anchor order    (B.3)
//   nms_kernel           <- steps 5-7: stable descending-score order, boxes + cls*7680,
//                           torchvision.ops.nms (strict iou > thr, no eps), [:max_det];
//                           then scale_boxes + clip                                      (B.3, B.4)
// Survivor indices are integers and must equal the oracle's on identical inputs, so the
// IoU arithmetic is kept un-contracted (-ffp-contract=off) and IEEE-divided.
//
// NMS shape: one 256-thread workgroup per image.  Candidates are compacted in anchor
// order (wave64 ballot + prefix), sorted by a 64-bit key (score bits, then lower anchor
// index first) with an LDS bitonic network, then suppressed greedily: each kept box is
// compared against all later live boxes by the whole workgroup (bitmap in LDS), and the
// loop stops after max_det keeps -- O(max_det * n) instead of the n^2 mask.
#include "kernels.h"

#include <climits>

namespace rtmodt {

#pragma clang fp contract(off)

constexpr int PP_THREADS = 256;
constexpr int PP_WAVES = PP_THREADS / 64;
constexpr int SORT_LDS_MAX = 8192;          // keys sorted in LDS (64 KiB); beyond: global rank sort
constexpr int MAX_NMS = 30000;              // ultralytics max_nms
constexpr float MAX_WH = 7680.0f;           // ultralytics max_wh (per-class coordinate offset)

// ---------------------------------------------------------------------------------------
// decode: one thread per (image, anchor)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_kernel(DecodeArgs a) {
    long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)a.B * a.n_anchors) return;
    int b = (int)(gid / a.n_anchors), an = (int)(gid - (long)b * a.n_anchors);
    int l = 0, local = an;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int cnt = a.lvl[k].H * a.lvl[k].W;
        if (l == k && local >= cnt) { local -= cnt; l = k + 1; }
    }
    const HeadLevel L = l == 0 ? a.lvl[0] : (l == 1 ? a.lvl[1] : a.lvl[2]);
    const int no = 64 + a.nc;
    const f16 *p = L.ptr + ((long)b * L.H * L.W + local) * no;
    int gy = local / L.W, gx = local - gy * L.W;
    float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;

    float dist[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[16];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) { v[j] = (float)p[s * 16 + j]; mx = fmaxf(mx, v[j]); }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { v[j] = expf(v[j] - mx); sum += v[j]; }
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) d += (v[j] / sum) * (float)j;
        dist[s] = d;
    }
    float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
    float st = (float)L.stride;
    float cx = ((x1 + x2) / 2.0f) * st, cy = ((y1 + y2) / 2.0f) * st;
    float bw = (x2 - x1) * st, bh = (y2 - y1) * st;

    float best = -1.0f;
    int bj = 0;
    float *pr = a.pred ? a.pred + (long)b * (4 + a.nc) * a.n_anchors + an : nullptr;
    for (int j = 0; j < a.nc; ++j) {
        float x = (float)p[64 + j];
        float sg = 1.0f / (1.0f + expf(-x));
        if (sg > best) { best = sg; bj = j; }            // first maximum
        if (pr) pr[(long)(4 + j) * a.n_anchors] = sg;
    }
    if (pr) {
        pr[0] = cx; pr[(long)a.n_anchors] = cy; pr[2L * a.n_anchors] = bw; pr[3L * a.n_anchors] = bh;
    }
    bool allowed = (a.class_mask[bj >> 6] >> (bj & 63)) & 1ull;
    bool cand = best > a.conf && allowed;
    float hw = bw / 2.0f, hh = bh / 2.0f;                // xywh2xyxy
    a.box[gid] = make_float4(cx - hw, cy - hh, cx + hw, cy + hh);
    a.score[gid] = cand ? best : -1.0f;
    a.cls[gid] = bj;
}

int launch_decode(const DecodeArgs &a, hipStream_t s) {
    long total = (long)a.B * a.n_anchors;
    RT_CHECK(a.nc >= 1 && a.nc <= 128, RTMODT_E_UNSUPPORTED, "decode: nc %d", a.nc);
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

// pred (4+nc, A) float32 -> dense candidates
__global__ __launch_bounds__(256) void pred_candidates_kernel(const float *__restrict__ pred, int nc, int A, float conf, uint64_t m0,
                                                              uint64_t m1, float4 *__restrict__ box, float *__restrict__ score,
                                                              int32_t *__restrict__ cls) {
    int an = blockIdx.x * 256 + threadIdx.x;
    if (an >= A) return;
    float best = -INFINITY;
    int bj = 0;
    for (int j = 0; j < nc; ++j) {
        float v = pred[(long)(4 + j) * A + an];
        if (v > best) { best = v; bj = j; }
    }
    uint64_t mask = bj < 64 ? m0 : m1;
    bool cand = best > conf && ((mask >> (bj & 63)) & 1ull);
    float cx = pred[an], cy = pred[(long)A + an], bw = pred[2L * A + an], bh = pred[3L * A + an];
    float hw = bw / 2.0f, hh = bh / 2.0f;
    box[an] = make_float4(cx - hw, cy - hh, cx + hw, cy + hh);
    score[an] = cand ? best : -1.0f;
    cls[an] = bj;
}

int launch_pred_candidates(const float *pred, int nc, int A, float conf, const uint64_t class_mask[2], float4 *box, float *score,
                           int32_t *cls, hipStream_t s) {
    hipLaunchKernelGGL(pred_candidates_kernel, dim3(cdiv(A, 256)), dim3(256), 0, s, pred, nc, A, conf, class_mask[0], class_mask[1], box,
                       score, cls);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

// ---------------------------------------------------------------------------------------
// NMS
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool nms_overlaps(const float4 a, float area_a, const float4 b, double thr) {
    float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
    float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
    float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    float inter = w * h;
    float area_b = (b.z - b.x) * (b.w - b.y);
    float ovr = inter / ((area_a + area_b) - inter);
    return (double)ovr > thr;                            // torchvision CPU kernel compares against the double threshold
}

__device__ __forceinline__ int pp_scan_flag(bool flag, int *wsum, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long m = __ballot(flag);
    int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < PP_WAVES; ++w) {
        int v = wsum[w];
        if (w < wave) off += v;
        tot += v;
    }
    __syncthreads();
    total = tot;
    return off + within;
}

__global__ __launch_bounds__(PP_THREADS) void nms_kernel(NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *skeys = (unsigned long long *)smem;                  // [SORT_LDS_MAX]
    unsigned long long *removed = skeys + SORT_LDS_MAX;                      // [ceil(MAX_NMS/64)]
    int *sel = (int *)(removed + (MAX_NMS + 63) / 64);                        // [max_det]
    __shared__ int wsum[PP_WAVES + 1];
    __shared__ int s_next;

    const int b = blockIdx.x, tid = threadIdx.x;
    const int A = a.n_anchors;
    const float4 *box = a.box + (size_t)b * A;
    const float *score = a.score + (size_t)b * A;
    const int32_t *cls = a.cls + (size_t)b * A;
    unsigned long long *gkeys = (unsigned long long *)a.keys + (size_t)b * A;
    float4 *sbox = a.sbox + (size_t)b * A;
    int32_t *sidx = a.sidx + (size_t)b * A;

    // ---- 1. compaction in anchor order ----
    int n = 0;
    for (int base = 0; base < A; base += PP_THREADS) {
        int i = base + tid;
        float sc = i < A ? score[i] : -1.0f;
        bool f = sc >= 0.0f;
        int tot;
        int pos = pp_scan_flag(f, wsum, tot);
        if (f) gkeys[n + pos] = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        n += tot;
    }
    __syncthreads();

    // ---- 2. sort by (score desc, anchor asc) ----
    if (n <= SORT_LDS_MAX) {
        int P = 1;
        while (P < n) P <<= 1;
        for (int i = tid; i < P; i += PP_THREADS) skeys[i] = i < n ? gkeys[i] : 0ull;
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < P; i += PP_THREADS) {
                    int ixj = i ^ j;
                    if (ixj > i) {
                        unsigned long long x = skeys[i], y = skeys[ixj];
                        bool desc = (i & k) == 0;
                        if (desc ? (x < y) : (x > y)) { skeys[i] = y; skeys[ixj] = x; }
                    }
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < n; i += PP_THREADS) {
            unsigned idx = 0xFFFFFFFFu - (unsigned)(skeys[i] & 0xFFFFFFFFull);
            sidx[i] = (int)idx;
        }
    } else {
        // rank sort through global memory (keys are unique): only reachable above 640x640 input
        for (int i = tid; i < n; i += PP_THREADS) {
            unsigned long long k = gkeys[i];
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += gkeys[j] > k;
            sidx[rank] = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
        }
    }
    __syncthreads();
    if (n > MAX_NMS) n = MAX_NMS;                          // top-max_nms by confidence (B.3 step 5)
    for (int i = tid; i < n; i += PP_THREADS) {
        int idx = sidx[i];
        float4 bx = box[idx];
        float off = a.agnostic ? 0.0f : (float)cls[idx] * MAX_WH;
        sbox[i] = make_float4(bx.x + off, bx.y + off, bx.z + off, bx.w + off);
    }
    const int nwords = (n + 63) >> 6;
    for (int w = tid; w < nwords; w += PP_THREADS) removed[w] = 0ull;
    __syncthreads();

    // ---- 3. greedy suppression, stops after max_det keeps ----
    const double thr = (double)a.iou;
    int kept = 0, pos = 0;
    while (kept < a.max_det && pos < n) {
        if (tid == 0) s_next = INT_MAX;
        __syncthreads();
        const int w0 = pos >> 6;
        for (int wb = w0; wb < nwords; wb += PP_THREADS) {
            int w = wb + tid;
            int cand = INT_MAX;
            if (w < nwords) {
                unsigned long long avail = ~removed[w];
                if (w == w0) avail &= ~((1ull << (pos & 63)) - 1ull);
                int last = n - (w << 6);
                if (last < 64) avail &= (1ull << last) - 1ull;
                if (avail) cand = (w << 6) + __builtin_ctzll(avail);
            }
            if (cand != INT_MAX) atomicMin(&s_next, cand);
            __syncthreads();
            if (s_next != INT_MAX) break;
        }
        const int cur = s_next;
        if (cur == INT_MAX) break;
        if (tid == 0) sel[kept] = cur;
        ++kept;
        pos = cur + 1;
        const float4 cb = sbox[cur];
        const float carea = (cb.z - cb.x) * (cb.w - cb.y);
        for (int j = pos + tid; j < n; j += PP_THREADS) {
            if ((removed[j >> 6] >> (j & 63)) & 1ull) continue;
            if (nms_overlaps(cb, carea, sbox[j], thr)) atomicOr(&removed[j >> 6], 1ull << (j & 63));
        }
        __syncthreads();
    }
    __syncthreads();

    // ---- 4. outputs (+ scale_boxes / clip) ----
    for (int k = tid; k < kept; k += PP_THREADS) {
        int idx = sidx[sel[k]];
        float4 bx = box[idx];
        if (a.rescale) {
            bx.x = (bx.x - a.pad_x) / a.gain; bx.z = (bx.z - a.pad_x) / a.gain;
            bx.y = (bx.y - a.pad_y) / a.gain; bx.w = (bx.w - a.pad_y) / a.gain;
            bx.x = fminf(fmaxf(bx.x, 0.0f), a.src_w); bx.z = fminf(fmaxf(bx.z, 0.0f), a.src_w);
            bx.y = fminf(fmaxf(bx.y, 0.0f), a.src_h); bx.w = fminf(fmaxf(bx.w, 0.0f), a.src_h);
        }
        size_t o = (size_t)b * a.max_det + k;
        ((float4 *)a.out_xyxy)[o] = bx;
        a.out_conf[o] = score[idx];
        a.out_cls[o] = cls[idx];
        if (a.out_anchor) a.out_anchor[o] = idx;
    }
    if (tid == 0) a.out_n[b] = kept;
}

int launch_nms(const NmsArgs &a, hipStream_t s) {
    size_t smem = (size_t)SORT_LDS_MAX * 8 + (size_t)((MAX_NMS + 63) / 64) * 8 + (size_t)a.max_det * 4 + 16;
    RT_CHECK(smem <= 150 * 1024, RTMODT_E_INVALID, "nms: max_det %d too large", a.max_det);
    RT_HIP(hipFuncSetAttribute((const void *)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(nms_kernel, dim3(a.B), dim3(PP_THREADS), smem, s, a);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

}  // namespace rtmodt
