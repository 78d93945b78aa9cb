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
#include "tile_math.h"

#include <climits>
#include <type_traits>

namespace rtmodt {

#pragma clang fp contract(off)

// nms_kernel's workgroup: 1024 threads (one image per workgroup; the rank sort and the suppression sweep are per-thread loops over the
// candidates a thread owns, so their latency goes with candidates / threads -- profiles/r03/nms_phases/); 256 = rounds 1-2, kept as an A/B
// and test hook (RTMODT_NMS_THREADS=256)
constexpr int SORT_LDS_MAX = 8192;          // keys sorted in LDS (64 KiB); beyond: rank sort tiled through LDS
constexpr int SORT_TILE = 4096;             // chunk of that tiled rank sort
constexpr int MAX_NMS = 30000;              // ultralytics max_nms
constexpr int REMOVED_WORDS = ((MAX_NMS + 63) / 64 + 1) & ~1;   // 64-bit words of the removed bitmap, even: what follows it in LDS stays 16-byte aligned
constexpr float MAX_WH = 7680.0f;           // ultralytics max_wh (per-class coordinate offset)

// ---------------------------------------------------------------------------------------
// decode: FOUR lanes per (image, anchor).  Lane j of the quad owns box side j (16 DFL bins:
// softmax . arange) and a quarter of the classes; an anchor's 64+nc logits are contiguous in
// the NHWC head tensor, so a wave reads 16 anchors x (64+nc) halves as one dense run.  The
// quad combines through two wave shuffles (best class: larger score, then LOWER class id =
// first maximum).
// ---------------------------------------------------------------------------------------
// One anchor's logits (row `p`: 64 box halves, then nc class halves) -> box, best class and score, by the quad of lanes
// j = 0..3 that owns it (lane j: box side j and a quarter of the classes).  `pr`: optional pred dump of this anchor.
struct DecodeCommon { int nc; float conf; uint64_t class_mask[2]; };
__device__ __forceinline__ void decode_row(const f16 *p, const int j, const int gx, const int gy, const float st, const DecodeCommon &a, float *pr,
                                           const long pr_stride, float4 &box_out, float &score_out, int &cls_out) {
    // ---- this lane's box side ----
    float v[16];
    {
        typedef _Float16 half8 __attribute__((ext_vector_type(8)));
        half8 h0 = *(const half8 *)(p + j * 16), h1 = *(const half8 *)(p + j * 16 + 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = (float)h0[k]; v[8 + k] = (float)h1[k]; }
    }
    float mx = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, v[k]);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { v[k] = expf(v[k] - mx); sum += v[k]; }
    float dist = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) dist += (v[k] / sum) * (float)k;
    // gather l, t, r, b into every lane of the quad
    const int qbase = (threadIdx.x & 63) & ~3;
    float dl = __shfl(dist, qbase + 0), dt = __shfl(dist, qbase + 1), dr = __shfl(dist, qbase + 2), db = __shfl(dist, qbase + 3);
    float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
    float x1 = ax - dl, y1 = ay - dt, x2 = ax + dr, y2 = ay + db;
    float cx = ((x1 + x2) / 2.0f) * st, cy = ((y1 + y2) / 2.0f) * st;
    float bw = (x2 - x1) * st, bh = (y2 - y1) * st;

    // ---- this lane's quarter of the classes: ids [c0, c1) ----
    const int per = (a.nc + 3) >> 2;
    const int c0 = j * per, c1 = min(c0 + per, a.nc);
    float best = -1.0f;
    int bj = c0;
    // Fast path: the sigmoid is strictly increasing on the fp16 grid as long as it is far from saturating in
    // float32 (x < 8: neighbouring halves are >= 2^-8 apart and sigmoid' > 3e-4, far above an ulp), so the first
    // arg-max of the logits IS the first arg-max of the scores and one sigmoid per anchor replaces nc of them.
    // Anything else (a logit >= 8, the pred dump, a class count that does not split into half4s) takes the
    // score-by-score loop below; the quad decides together (xm is quad-uniform).
    bool slow = pr != nullptr || (per & 3) != 0 || c1 - c0 != per;
    if (!slow) {
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        float xb = -INFINITY;
        for (int k = 0; k < per; k += 4) {
            const half4 h = *(const half4 *)(p + 64 + c0 + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = (float)h[e];
                if (x > xb) { xb = x; bj = c0 + k + e; }
            }
        }
#pragma unroll
        for (int d = 1; d <= 2; d <<= 1) {
            float ob = __shfl_xor(xb, d);
            int oj = __shfl_xor(bj, d);
            if (ob > xb || (ob == xb && oj < bj)) { xb = ob; bj = oj; }
        }
        if (xb < 8.0f) best = 1.0f / (1.0f + expf(-xb));
        else slow = true;
    }
    if (slow) {
        best = -1.0f;
        bj = c0;
        for (int c = c0; c < c1; ++c) {
            float x = (float)p[64 + c];
            float sg = 1.0f / (1.0f + expf(-x));
            if (sg > best) { best = sg; bj = c; }              // first maximum inside the quarter
            if (pr) pr[(long)(4 + c) * pr_stride] = sg;
        }
#pragma unroll
        for (int d = 1; d <= 2; d <<= 1) {
            float ob = __shfl_xor(best, d);
            int oj = __shfl_xor(bj, d);
            if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
        }
    }
    if (pr && j == 0) { pr[0] = cx; pr[pr_stride] = cy; pr[2 * pr_stride] = bw; pr[3 * pr_stride] = bh; }
    const bool allowed = (a.class_mask[bj >> 6] >> (bj & 63)) & 1ull;
    const bool cand = best > a.conf && allowed;
    const float hw = bw / 2.0f, hh = bh / 2.0f;          // xywh2xyxy
    box_out = make_float4(cx - hw, cy - hh, cx + hw, cy + hh);
    score_out = cand ? best : -1.0f;
    cls_out = bj;
}

__global__ __launch_bounds__(256) void decode_kernel(DecodeArgs a) {
    const long gid = ((long)blockIdx.x * 256 + threadIdx.x) >> 2;
    const int j = threadIdx.x & 3;
    const long total = (long)a.B * a.n_anchors;
    const bool live = gid < total;
    const long g = live ? gid : total - 1;                  // dead quads shadow the last anchor (shuffles stay convergent)
    int b = (int)(g / a.n_anchors), an = (int)(g - (long)b * a.n_anchors);
    int l = 0, local = an;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int cnt = a.lvl[k].H * a.lvl[k].W;
        if (l == k && local >= cnt) { local -= cnt; l = k + 1; }
    }
    const HeadLevel L = l == 0 ? a.lvl[0] : (l == 1 ? a.lvl[1] : a.lvl[2]);
    const f16 *p = L.ptr + ((long)b * L.H * L.W + local) * a.no;
    int gy = local / L.W, gx = local - gy * L.W;
    float *pr = (a.pred && live) ? a.pred + (long)b * (4 + a.nc) * a.n_anchors + an : nullptr;
    const DecodeCommon dc{a.nc, a.conf, {a.class_mask[0], a.class_mask[1]}};
    float4 box; float score; int cls;
    decode_row(p, j, gx, gy, (float)L.stride, dc, pr, (long)a.n_anchors, box, score, cls);
    if (!live || j != 0) return;
    a.box[gid] = box;
    a.score[gid] = score;
    a.cls[gid] = cls;
}

int launch_decode(const DecodeArgs &a, hipStream_t s) {
    long total = (long)a.B * a.n_anchors;
    RT_CHECK(a.nc >= 1 && a.nc <= 128, RTMODT_E_UNSUPPORTED, "decode: nc %d", a.nc);
    RT_CHECK(a.no % 8 == 0 && a.no >= 64 + a.nc, RTMODT_E_INVALID, "decode: head row stride %d", a.no);
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((total * 4 + 255) / 256)), dim3(256), 0, s, a);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

// ---------------------------------------------------------------------------------------
// head_final: cv2.l.2 + cv3.l.2 (1x1, no activation) + decode.  One 512-thread workgroup per 128 anchors of one
// level of one image: the two input tiles and both weight matrices arrive by LDS-DMA (16-row x 64-byte pieces,
// XOR-swizzled on the source address like conv.hip), each wave multiplies one 16-anchor tile on the matrix
// cores, the logits are rounded to fp16 into an LDS row per anchor -- exactly what the head tensor would hold --
// and decode_row runs on them, four lanes per anchor.
// ---------------------------------------------------------------------------------------
typedef _Float16 hf_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf_half4 __attribute__((ext_vector_type(4)));
typedef float hf_floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int hf_swz(int row) { return ((row >> 3) & 1) * 3; }
__device__ __forceinline__ void hf_dma16(const f16 *src, unsigned char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

constexpr int HF_WAVES = 8, HF_ANCHORS = 16 * HF_WAVES;     // 8 waves: twice the DMA issue rate, weights amortised over 128 anchors

template <int KCC, int NTC>      // ccls / 32, ceil(nc / 16)
__global__ __launch_bounds__(64 * HF_WAVES) void head_final_kernel(HeadFinalArgs a) {
    constexpr int NW = HF_WAVES;
    constexpr int KCB = 2, NTB = 4;                       // cbox = 64: two 32-deep chunks; 64 box logits: four 16-wide tiles
    constexpr int XB = 0, XC = XB + KCB * NW * 1024, WB = XC + KCC * NW * 1024, WC = WB + KCB * NTB * 1024, END = WC + KCC * NTC * 1024;
    constexpr int ROWO = (64 + NTC * 16) * 2 + 16;
    constexpr int OUT = 0;                                 // the logit rows reuse the (consumed) input tiles
    static_assert(HF_ANCHORS * ROWO <= WB, "logit rows must fit the input-tile region");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[END];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int ld_row = lane >> 2, ld_chunk = (lane & 3) ^ hf_swz(ld_row), rd_off = r * 64 + ((q ^ hf_swz(r)) << 4);

    // ---- which level / image / tile ----
    int rest = blockIdx.x, l = 0, base_anchor = 0;
    for (; l < 2; ++l) {
        const int n = a.lvl[l].tiles * a.B;
        if (rest < n) break;
        rest -= n;
        base_anchor += a.lvl[l].H * a.lvl[l].W;
    }
    const HeadFinalLevel L = l == 0 ? a.lvl[0] : (l == 1 ? a.lvl[1] : a.lvl[2]);
    const int b = rest / L.tiles, tile = rest - b * L.tiles;
    const int HW = L.H * L.W, a0 = tile * HF_ANCHORS;

    // ---- operands -> LDS ----
    {
        const int an = min(a0 + wave * 16 + ld_row, HW - 1);                  // tail tiles re-read the last anchor (never stored)
        const f16 *xb = L.xb + ((long)b * HW + an) * a.cbox + ld_chunk * 8;
        const f16 *xc = L.xc + ((long)b * HW + an) * a.ccls + ld_chunk * 8;
#pragma unroll
        for (int kc = 0; kc < KCB; ++kc) hf_dma16(xb + kc * 32, lds + XB + (kc * NW + wave) * 1024);
#pragma unroll
        for (int kc = 0; kc < KCC; ++kc) hf_dma16(xc + kc * 32, lds + XC + (kc * NW + wave) * 1024);
        for (int pi = wave; pi < KCB * NTB; pi += NW) {
            const int kc = pi / NTB, u = pi - kc * NTB;
            hf_dma16(L.wb + ((u * 16 + ld_row) * a.cbox + kc * 32 + ld_chunk * 8), lds + WB + pi * 1024);
        }
        for (int pi = wave; pi < KCC * NTC; pi += NW) {
            const int kc = pi / NTC, u = pi - kc * NTC;
            hf_dma16(L.wc + ((u * 16 + ld_row) * a.ccls + kc * 32 + ld_chunk * 8), lds + WC + pi * 1024);
        }
    }
    hf_floatx4 bb[NTB], bc[NTC];
#pragma unroll
    for (int u = 0; u < NTB; ++u) bb[u] = *(const hf_floatx4 *)(L.bb + u * 16 + q * 4);
#pragma unroll
    for (int u = 0; u < NTC; ++u) bc[u] = *(const hf_floatx4 *)(L.bc + u * 16 + q * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- the two 1x1 convs: wave w owns anchors [16 w, 16 w + 16) ----
    hf_floatx4 accb[NTB], accc[NTC];
#pragma unroll
    for (int u = 0; u < NTB; ++u) accb[u] = hf_floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NTC; ++u) accc[u] = hf_floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KCB; ++kc) {
        const hf_half8 fa = *(const hf_half8 *)(lds + XB + (kc * NW + wave) * 1024 + rd_off);
#pragma unroll
        for (int u = 0; u < NTB; ++u)
            accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const hf_half8 *)(lds + WB + (kc * NTB + u) * 1024 + rd_off), fa, accb[u], 0, 0, 0);
    }
#pragma unroll
    for (int kc = 0; kc < KCC; ++kc) {
        const hf_half8 fa = *(const hf_half8 *)(lds + XC + (kc * NW + wave) * 1024 + rd_off);
#pragma unroll
        for (int u = 0; u < NTC; ++u)
            accc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const hf_half8 *)(lds + WC + (kc * NTC + u) * 1024 + rd_off), fa, accc[u], 0, 0, 0);
    }
    __syncthreads();                                       // every wave has consumed its input tiles: their space becomes the logit rows
    // bias (no activation), one rounding to fp16: the row the head tensor would hold
    unsigned char *row = lds + OUT + (wave * 16 + r) * ROWO;
#pragma unroll
    for (int u = 0; u < NTB; ++u) {
        const hf_floatx4 v = accb[u] + bb[u];
        *(hf_half4 *)(row + (u * 16 + q * 4) * 2) = hf_half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    }
#pragma unroll
    for (int u = 0; u < NTC; ++u) {
        const hf_floatx4 v = accc[u] + bc[u];
        *(hf_half4 *)(row + (64 + u * 16 + q * 4) * 2) = hf_half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    }
    __syncthreads();

    // ---- decode: four lanes per anchor ----
    const int al = threadIdx.x >> 2, j = threadIdx.x & 3;
    const int local = min(a0 + al, HW - 1);
    const bool live = a0 + al < HW;
    const int gy = local / L.W, gx = local - gy * L.W;
    const DecodeCommon dc{a.nc, a.conf, {a.class_mask[0], a.class_mask[1]}};
    float4 box; float score; int cls;
    decode_row((const f16 *)(lds + OUT + al * ROWO), j, gx, gy, (float)L.stride, dc, nullptr, 0, box, score, cls);
    if (live && j == 0) {
        const long gid = (long)b * a.n_anchors + base_anchor + local;
        a.box[gid] = box;
        a.score[gid] = score;
        a.cls[gid] = cls;
    }
    if (L.heads && live) {                                 // debug: the head rows themselves
        f16 *hp = L.heads + ((long)b * HW + local) * a.no;
        const f16 *sp = (const f16 *)(lds + OUT + al * ROWO);
        for (int c = j * 8; c < a.no; c += 32) *(hf_half8 *)(hp + c) = *(const hf_half8 *)(sp + c);
    }
}

bool head_final_supported(int cbox, int ccls, int nc) { return cbox == 64 && (ccls == 128 || ccls == 192) && nc >= 1 && nc <= 80 && nc % 4 == 0; }

int launch_head_final(const HeadFinalArgs &a, hipStream_t s) {
    RT_CHECK(head_final_supported(a.cbox, a.ccls, a.nc), RTMODT_E_UNSUPPORTED, "head_final: cbox %d ccls %d nc %d", a.cbox, a.ccls, a.nc);
    RT_CHECK(a.no % 8 == 0 && a.no >= 64 + a.nc && a.no <= 64 + 80, RTMODT_E_INVALID, "head_final: head row stride %d", a.no);
    int blocks = 0;
    for (int l = 0; l < 3; ++l) {
        RT_CHECK(a.lvl[l].tiles == cdiv(a.lvl[l].H * a.lvl[l].W, HF_ANCHORS), RTMODT_E_INVALID, "head_final: tiles of level %d", l);
        blocks += a.lvl[l].tiles * a.B;
    }
    if (a.ccls == 128) hipLaunchKernelGGL((head_final_kernel<4, 5>), dim3(blocks), dim3(64 * HF_WAVES), 0, s, a);
    else hipLaunchKernelGGL((head_final_kernel<6, 5>), dim3(blocks), dim3(64 * HF_WAVES), 0, s, a);
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

constexpr int SBOX_LDS_MAX = 2048;          // sorted boxes kept in LDS (32 KiB); beyond: global scratch
constexpr int RANK_LDS_MAX = 2048;          // the LDS rank sort's capacity (8 keys per thread x 256 threads); launch_nms picks the switch-over below it

// one barrier step of the bitonic network over G consecutive strides (s0 << (G - 1), ..., s0): 2^G keys per thread in registers
template <int G, int PP_THREADS>
__device__ __forceinline__ void bitonic_step(unsigned long long *keys, int P, int k, int s0, int ls) {
    constexpr int E = 1 << G;
    for (int t = threadIdx.x; t < (P >> G); t += PP_THREADS) {
        const int i0 = bitonic_i0(t, ls, G, s0);                            // G zero bits inserted at bit ls (tile_math.h)
        const bool desc = (i0 & k) == 0;
        unsigned long long v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = keys[i0 + e * s0];
#pragma unroll
        for (int h = E >> 1; h >= 1; h >>= 1) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((e & h) == 0) {
                    const unsigned long long x = v[e], y = v[e | h];
                    const bool sw = desc ? (x < y) : (x > y);
                    v[e] = sw ? y : x;
                    v[e | h] = sw ? x : y;
                }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) keys[i0 + e * s0] = v[e];
    }
}

// Greedy suppression over boxes sorted by descending score, 64 STILL-ALIVE sorted positions at a time
// (exactly torchvision's result: a box is kept iff no earlier KEPT box overlaps it):
//   G. the block = the next (up to) 64 positions after the cursor that no kept box has removed yet, gathered from the removed
//      bitmap by every wave for itself (one bitmap word per lane over a 4 096-position window, wave prefix sum of the
//      popcounts, member s = the (s - prefix)-th clear bit of its word): in a dense scene a few kept boxes remove thousands of
//      positions, and the blocks then step over them instead of walking them 64 at a time;
//   A. the 64 x 64 overlap matrix of the block is formed by the whole workgroup (threads / 64 threads per row,
//      4 or 16 columns each), rows land in LDS as 64-bit masks;
//   B. every wave resolves the block serially but entirely in registers -- lane i holds row i,
//      v_readlane with a scalar index fetches the row of each box as it is kept: no LDS traffic, no barrier, one short step per kept box;
//   C. the boxes kept in this block are applied to all later positions: each thread owns the
//      sorted positions t, t + threads, ... (alive bits in two registers) and publishes removals
//      with one LDS atomicOr each.
// Two barriers per block instead of one (or two) per kept box; stops at max_det keeps.
template <bool LDSBOX, int PP_THREADS>
__device__ __forceinline__ int greedy_nms(const float4 *lbox, const float4 *gbox, int n, int max_det, double thr,
                                          unsigned long long *removed, unsigned long long *rowmask, int *sel, int *members, float4 *mbox) {
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned long long alive0 = 0ull, alive1 = 0ull;
    for (int k = 0; k < 64; ++k) {
        if (tid + PP_THREADS * k < n) alive0 |= 1ull << k;
        if (tid + PP_THREADS * (64 + k) < n) alive1 |= 1ull << k;
    }
    auto box_at = [&](int j) -> float4 { return LDSBOX ? lbox[j] : gbox[j]; };
    constexpr int OWN = 8;                                  // my first OWN positions' boxes stay in registers for sweep C
    float4 own[OWN];
#pragma unroll
    for (int k = 0; k < OWN; ++k) own[k] = tid + PP_THREADS * k < n ? box_at(tid + PP_THREADS * k) : make_float4(0.f, 0.f, 0.f, 0.f);
    int kept = 0, cursor = 0;
    const int nwords = (n + 63) >> 6;
    while (cursor < n && kept < max_det) {
        // ---- G: the next <= 64 alive positions (identical in every wave) ----
        const int cw = cursor >> 6;
        unsigned long long al = 0ull;
        {
            const int w = cw + lane;
            if (w < nwords) {
                al = ~removed[w];
                if (w == cw) al &= ~0ull << (cursor & 63);
                const int valid = n - (w << 6);
                if (valid < 64) al &= (1ull << valid) - 1ull;
            }
        }
        const int pc = __popcll(al);
        int incl = pc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        const int window_end = min(n, (cw + 64) << 6);
        if (total == 0) { cursor = window_end; continue; }     // uniform; nothing was written
        const int cnt = min(64, total);
        int pos = 0;
        {
            int L = 0;                                         // the first lane whose inclusive count exceeds my slot
#pragma unroll
            for (int step = 32; step; step >>= 1) {
                const int v = __shfl(incl, L + step - 1);
                if (v <= lane) L += step;
            }
            L = min(L, 63);
            const unsigned long long wL = ((unsigned long long)(unsigned)__shfl((int)(unsigned)(al >> 32), L) << 32) | (unsigned)__shfl((int)(unsigned)al, L);
            const int eL = __shfl(incl - pc, L);
            if (lane < cnt) pos = ((cw + L) << 6) + kth_set_bit(wL, lane - eL);
        }
        const int cursor_next = total > 64 ? __builtin_amdgcn_readlane(pos, 63) + 1 : window_end;
        if (lane < cnt) { members[lane] = pos; mbox[lane] = box_at(pos); }      // every wave writes the same values; it reads back only its own
        // ---- A: overlap rows of the block ----
        {
            constexpr int TPR = PP_THREADS / 64, CPT = 64 / TPR;       // threads per row, columns per thread
            const int i = tid / TPR, j0 = (tid % TPR) * CPT;
            unsigned long long part = 0ull;
            if (i < cnt) {
                const float4 bi = mbox[i];
                const float ai = (bi.z - bi.x) * (bi.w - bi.y);
#pragma unroll 4
                for (int jj = 0; jj < CPT; ++jj) {
                    const int j = j0 + jj;
                    if (j > i && j < cnt && nms_overlaps(bi, ai, mbox[j], thr)) part |= 1ull << j;
                }
            }
#pragma unroll
            for (int d = 1; d < TPR; d <<= 1) part |= __shfl_xor(part, d);
            if (tid % TPR == 0) rowmask[i] = part;
        }
        __syncthreads();
        // ---- B: serial resolve in registers (identical in every wave): one step per KEPT box -- the lowest live member is kept
        //         and takes its row out of the live set (rows only hold later members)
        const unsigned long long mine = rowmask[lane];
        const unsigned lo = (unsigned)mine, hi = (unsigned)(mine >> 32);
        unsigned long long live = cnt == 64 ? ~0ull : (1ull << cnt) - 1ull;
        unsigned long long keepm = 0ull;
        while (live) {
            const int i = __builtin_ctzll(live);
            keepm |= 1ull << i;
            const unsigned long long row = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, i) << 32) |
                                           (unsigned)__builtin_amdgcn_readlane((int)lo, i);
            live &= ~(row | (1ull << i));
        }
        int nk = __popcll(keepm);
        if (kept + nk > max_det) {                            // keep[:max_det]
            int drop = kept + nk - max_det;
            while (drop--) keepm &= ~(1ull << (63 - __builtin_clzll(keepm)));
            nk = max_det - kept;
        }
        if (tid < 64 && ((keepm >> tid) & 1ull)) sel[kept + __popcll(keepm & ((1ull << tid) - 1ull))] = pos;
        kept += nk;
        if (kept >= max_det) break;
        // ---- C: apply this block's kept boxes to every later position I own ----
        {   // positions before the next block's cursor are settled: drop them from my pool
            int kmin = cursor_next > tid ? (cursor_next - tid + PP_THREADS - 1) / PP_THREADS : 0;
            if (kmin >= 64) { alive0 = 0ull; int k1 = kmin - 64; alive1 = k1 >= 64 ? 0ull : alive1 & ~((1ull << k1) - 1ull); }
            else alive0 &= ~((1ull << kmin) - 1ull);
        }
        for (unsigned long long km = keepm; km; km &= km - 1) {
            const float4 cb = mbox[__builtin_ctzll(km)];
            const float carea = (cb.z - cb.x) * (cb.w - cb.y);
#pragma unroll
            for (int k = 0; k < OWN; ++k)                                        // boxes in registers: no memory in this chain
                if (((alive0 >> k) & 1ull) && nms_overlaps(cb, carea, own[k], thr)) {
                    alive0 &= ~(1ull << k);
                    const int j = tid + PP_THREADS * k;
                    atomicOr(&removed[j >> 6], 1ull << (j & 63));
                }
            for (unsigned long long m = alive0 >> OWN; m; m &= m - 1) {
                int k = OWN + __builtin_ctzll(m);
                int j = tid + PP_THREADS * k;
                if (nms_overlaps(cb, carea, box_at(j), thr)) { alive0 &= ~(1ull << k); atomicOr(&removed[j >> 6], 1ull << (j & 63)); }
            }
            for (unsigned long long m = alive1; m; m &= m - 1) {
                int k = __builtin_ctzll(m);
                int j = tid + PP_THREADS * (64 + k);
                if (nms_overlaps(cb, carea, box_at(j), thr)) { alive1 &= ~(1ull << k); atomicOr(&removed[j >> 6], 1ull << (j & 63)); }
            }
        }
        __syncthreads();                                      // removals visible; rowmask, members and mbox reusable
        cursor = cursor_next;
    }
    return kept;
}

template <int PP_THREADS>
__global__ __launch_bounds__(PP_THREADS) void nms_kernel(NmsArgs a, int dbg_stop, int rank_max) {
    constexpr int PP_WAVES = PP_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: [0, 64 KiB) the sort keys; once sorted, their anchor halves are packed in place into lidx (the first 32 KiB) and the sorted,
    // class-offset boxes take the second 32 KiB
    unsigned long long *skeys = (unsigned long long *)smem;                  // [SORT_LDS_MAX] sort keys
    int *lidx = (int *)smem;                                                 // [SORT_LDS_MAX] sorted position -> anchor (after the sort)
    float4 *lbox = (float4 *)(smem + (size_t)SORT_LDS_MAX * 4);              // [SBOX_LDS_MAX] sorted, class-offset boxes
    static_assert((size_t)SORT_LDS_MAX * 4 + (size_t)SBOX_LDS_MAX * 16 <= (size_t)SORT_LDS_MAX * 8, "boxes alias the upper half of the key area");
    unsigned long long *removed = (unsigned long long *)(skeys + SORT_LDS_MAX);  // [REMOVED_WORDS] removed bitmap
    unsigned long long *rowmask = removed + REMOVED_WORDS;                        // [64] overlap rows of the block being resolved
    float4 *mbox = (float4 *)(rowmask + 64);                                  // [64] boxes of the block being resolved
    int *members = (int *)(mbox + 64);                                        // [64] their sorted positions
    int *sel = members + 64;                                                  // [max_det]
    __shared__ int wsum[PP_WAVES + 1];

    const int b = blockIdx.x, tid = threadIdx.x;
    const int A = a.n_anchors;
    const float4 *box = a.box + (size_t)b * A;
    const float *score = a.score + (size_t)b * A;
    const int32_t *cls = a.cls + (size_t)b * A;
    unsigned long long *gkeys = (unsigned long long *)a.keys + (size_t)b * A;
    float4 *gsbox = a.sbox + (size_t)b * A;
    int32_t *sidx = a.sidx + (size_t)b * A;

    // ---- 1. compaction in anchor order: thread t owns the contiguous anchors [t*per, (t+1)*per) ----
    const int per = (A + PP_THREADS - 1) / PP_THREADS;
    const int lo = min(tid * per, A), hi = min(lo + per, A);
    constexpr int REG_SCORES = PP_THREADS == 256 ? 40 : 12;   // 8400 anchors / 256 threads = 33, / 1024 = 9
    const bool in_regs = per <= REG_SCORES;                  // uniform
    float sc[REG_SCORES];
    int mine = 0;
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < REG_SCORES; ++k) sc[k] = lo + k < hi ? score[lo + k] : -1.0f;   // all loads in flight at once
#pragma unroll
        for (int k = 0; k < REG_SCORES; ++k) mine += sc[k] >= 0.0f;
    } else {
        for (int i0 = lo; i0 < hi; i0 += 8) {                // 8 independent loads in flight per trip
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = i0 + k < hi ? score[i0 + k] : -1.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) mine += v[k] >= 0.0f;
        }
    }
    int n, base;
    {   // block exclusive scan of the per-thread counts
        const int lane = tid & 63, wave = tid >> 6;
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < PP_WAVES; ++w) {
            int v = wsum[w];
            if (w < wave) off += v;
            tot += v;
        }
        base = off + incl - mine;
        n = tot;
    }
    if (dbg_stop == 1) return;
    const bool lds_sort = n <= SORT_LDS_MAX;
    auto put_key = [&](int o, float v, int anchor) {
        unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)anchor);
        if (lds_sort) skeys[o] = key; else gkeys[o] = key;
    };
    if (mine) {
        int o = base;
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < REG_SCORES; ++k)
                if (sc[k] >= 0.0f) put_key(o++, sc[k], lo + k);
        } else {
            for (int i0 = lo; i0 < hi; i0 += 8) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = i0 + k < hi ? score[i0 + k] : -1.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (v[k] >= 0.0f) put_key(o++, v[k], i0 + k);
            }
        }
    }
    __syncthreads();
    if (dbg_stop == 2) return;

    // ---- 2. sort by (score desc, anchor asc); keys are unique ----
    if (n <= rank_max) {
        // LDS rank sort: each thread ranks its <= 8 keys against all n (16-byte broadcast reads, no barriers)
        const int npad = (n + 1) & ~1;
        if (tid == 0 && (n & 1)) skeys[n] = 0ull;
        __syncthreads();
        unsigned long long mk[8];
        int rk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { int i = tid + PP_THREADS * e; mk[e] = i < n ? skeys[i] : ~0ull; rk[e] = 0; }
        const ulonglong2 *k2 = (const ulonglong2 *)skeys;
        const int E = (n + PP_THREADS - 1) / PP_THREADS;     // keys per thread actually in use (uniform)
        auto rank_loop = [&](auto ne) {
            constexpr int NE = decltype(ne)::value;
#pragma unroll 4
            for (int j = 0; j < (npad >> 1); ++j) {
                ulonglong2 kk = k2[j];
#pragma unroll
                for (int e = 0; e < NE; ++e) rk[e] += (kk.x > mk[e]) + (kk.y > mk[e]);
            }
        };
        if (E <= 1) rank_loop(std::integral_constant<int, 1>{});
        else if (E <= 2) rank_loop(std::integral_constant<int, 2>{});
        else if (E <= 4) rank_loop(std::integral_constant<int, 4>{});
        else rank_loop(std::integral_constant<int, 8>{});
        __syncthreads();                                      // everyone has read skeys
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (tid + PP_THREADS * e < n) skeys[rk[e]] = mk[e];
        __syncthreads();
    } else if (lds_sort) {
        int P = 1;
        while (P < n) P <<= 1;
        for (int i = n + tid; i < P; i += PP_THREADS) skeys[i] = 0ull;
        __syncthreads();
        // bitonic network, up to three consecutive strides (4s, 2s, s) per barrier: a thread takes the 8 keys i0 + e * s whose indices
        // differ only in those three bits into registers, runs the three compare-exchange steps there and writes them back
        // (8 192 keys: 35 barrier steps instead of 91, a third of the LDS traffic); the direction bit k lies above all three strides
        for (int k = 2; k <= P; k <<= 1) {
            int j = k >> 1;
            while (j > 0) {
                const int g = bitonic_group(j);                         // strides in this step (tile_math.h)
                const int s0 = j >> (g - 1), ls = __builtin_ctz(s0);     // the smallest of them
                if (g == 3) bitonic_step<3, PP_THREADS>(skeys, P, k, s0, ls);
                else if (g == 2) bitonic_step<2, PP_THREADS>(skeys, P, k, s0, ls);
                else bitonic_step<1, PP_THREADS>(skeys, P, k, s0, ls);
                __syncthreads();
                j >>= g;
            }
        }
    } else {
        // > 8192 candidates (a saturated head, a very low confidence threshold): rank sort with the keys TILED through LDS --
        // every thread ranks 8 of its keys at a time against chunks of SORT_TILE keys read as 16-byte broadcasts, as the
        // LDS rank sort above does.  (The first version compared against global memory key by key: 10 ms for 8 400 candidates.)
        const ulonglong2 *k2 = (const ulonglong2 *)skeys;
        for (int base0 = 0; base0 < n; base0 += PP_THREADS * 8) {         // uniform trip count: barriers inside
            unsigned long long mk[8];
            int rk[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int i = base0 + tid + PP_THREADS * e; mk[e] = i < n ? gkeys[i] : ~0ull; rk[e] = 0; }
            for (int c0 = 0; c0 < n; c0 += SORT_TILE) {
                const int cn = min(SORT_TILE, n - c0), cpad = (cn + 1) & ~1;
                __syncthreads();                                      // everyone is done with the previous chunk
                for (int i = tid; i < cpad; i += PP_THREADS) skeys[i] = i < cn ? gkeys[c0 + i] : 0ull;   // 0 ranks below every key
                __syncthreads();
#pragma unroll 4
                for (int j = 0; j < (cpad >> 1); ++j) {
                    const ulonglong2 kk = k2[j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) rk[e] += (kk.x > mk[e]) + (kk.y > mk[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (base0 + tid + PP_THREADS * e < n) sidx[rk[e]] = (int)(0xFFFFFFFFu - (unsigned)(mk[e] & 0xFFFFFFFFull));
        }
        __syncthreads();
    }
    if (dbg_stop == 3) return;
    if (n > MAX_NMS) n = MAX_NMS;                          // top-max_nms by confidence (B.3 step 5)
    if (lds_sort) {
        // keys -> anchors, in place: lidx[i] (bytes 4i..) lands on keys[i / 2]; round r reads its PP_THREADS keys, everybody
        // meets, then writes -- the bytes it writes lie below every key a later round still has to read
        for (int i0 = 0; i0 < n; i0 += PP_THREADS) {       // uniform trip count: a barrier inside
            const int i = i0 + tid;
            const unsigned long long key = i < n ? skeys[i] : 0ull;
            __syncthreads();
            if (i < n) lidx[i] = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        }
        __syncthreads();
    }
    const bool lds_box = lds_sort && n <= SBOX_LDS_MAX;
    for (int i = tid; i < n; i += PP_THREADS) {
        int idx = lds_sort ? lidx[i] : sidx[i];
        float4 bx = box[idx];
        float off = a.agnostic ? 0.0f : (float)cls[idx] * MAX_WH;
        float4 ob = make_float4(bx.x + off, bx.y + off, bx.z + off, bx.w + off);
        if (lds_box) lbox[i] = ob; else gsbox[i] = ob;
    }
    for (int w = tid; w < ((n + 63) >> 6); w += PP_THREADS) removed[w] = 0ull;
    __syncthreads();
    if (dbg_stop == 4) return;

    // ---- 3. greedy suppression ----
    const double thr = (double)a.iou;
    const int kept = lds_box ? greedy_nms<true, PP_THREADS>(lbox, gsbox, n, a.max_det, thr, removed, rowmask, sel, members, mbox)
                             : greedy_nms<false, PP_THREADS>(lbox, gsbox, n, a.max_det, thr, removed, rowmask, sel, members, mbox);
    __syncthreads();
    if (dbg_stop == 5) return;

    // ---- 4. outputs (+ scale_boxes / clip) ----
    for (int k = tid; k < kept; k += PP_THREADS) {
        int sp = sel[k];
        int idx = lds_sort ? lidx[sp] : sidx[sp];
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

// The hooks are resolved ONCE (per detector at create time, per call of the stand-alone rtmodt_nms_pred), not per launch
NmsPlan nms_plan_from_options() {
    NmsPlan p;
    if (const char *e = rt_opt("NMS_THREADS")) p.threads = atoi(e) == 256 ? 256 : 1024;            // test hook: the round-2 form of the kernel
    // LDS rank sort (no barriers, n^2 / threads 64-bit compares per thread) up to rank_max, the bitonic network (log^2 n barrier steps) above:
    // with 1024 threads they cross at ~500 candidates (profiles/r03/nms_phases/sort_crossover.txt: 1 000 candidates 25 vs 18 us,
    // 2 000: 74 vs 24); the 256-thread form keeps round 2's 2 048
    p.rank_max = p.threads == 1024 ? 512 : RANK_LDS_MAX;
    if (const char *e = rt_diag("NMS_RANK_MAX")) p.rank_max = atoi(e);
    p.rank_max = max(0, min(p.rank_max, RANK_LDS_MAX));
    if (const char *e = rt_diag("NMS_STOP")) p.stop = atoi(e);                                     // timing-only cuts after each phase
    return p;
}

int launch_nms(const NmsArgs &a, const NmsPlan &plan, hipStream_t s) {
    size_t smem = (size_t)SORT_LDS_MAX * 8 + (size_t)REMOVED_WORDS * 8 + 64 * 8 + 64 * 16 + 64 * 4 + (size_t)a.max_det * 4 + 16;
    RT_CHECK(smem <= 150 * 1024, RTMODT_E_INVALID, "nms: max_det %d too large", a.max_det);
    RT_CHECK((plan.threads == 1024 || plan.threads == 256) && plan.rank_max >= 0 && plan.rank_max <= RANK_LDS_MAX, RTMODT_E_INVALID,
             "nms: plan (%d threads, rank sort up to %d)", plan.threads, plan.rank_max);
    const bool wide = plan.threads == 1024;
    static DynLdsSeen seen[2];                             // raise the dynamic-LDS limit once per device and size, not per launch
    RT_TRY(raise_dynamic_lds(wide ? (const void *)nms_kernel<1024> : (const void *)nms_kernel<256>, smem, seen[wide]));
    if (wide) hipLaunchKernelGGL(nms_kernel<1024>, dim3(a.B), dim3(1024), smem, s, a, plan.stop, plan.rank_max);
    else hipLaunchKernelGGL(nms_kernel<256>, dim3(a.B), dim3(256), smem, s, a, plan.stop, plan.rank_max);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

}  // namespace rtmodt
