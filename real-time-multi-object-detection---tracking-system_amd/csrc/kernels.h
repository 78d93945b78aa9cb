// kernels.h -- host-side launch interface of the hand-written gfx950 kernels.
#pragma once

#include "common.h"

namespace rtmodt {

// Diagnostic build only (tools/probes/kernel_probe.hip, -DRTMODT_STAMP): lane 0 of every workgroup writes the shader clock at
// phase boundaries into a buffer of its own; no product build contains a stamp.
#ifdef RTMODT_STAMP
extern __device__ unsigned long long *g_stamps;
#define STAMP(k)                                                                                         \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (threadIdx.x == 0) {                                                                          \
            unsigned long long t_, r_;                                                                   \
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");          \
            unsigned long long *s_ = g_stamps + (size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 16;       \
            s_[(k)] = t_;                                                                                \
            if ((k) == 0) { s_[13] = r_; s_[12] = t_; }      /* shader clock / 100 MHz real-time pair at the first ... */ \
            s_[15] = r_; s_[14] = t_;                        /* ... and at the latest stamp: in-kernel clock = d(12,14) / d(13,15) x 100 MHz */ \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#else
#define STAMP(k)
#endif

// ---------------------------------------------------------------------------------------
// Activation tensors: fp16 NHWC with an optional 1-pixel zero border,
//   element (b, y, x, c) at ((b*(H+2*pad) + y + pad)*(W+2*pad) + x + pad)*C + c.
// The border is zeroed once at arena creation and never written, so a 3x3 conv reads its
// halo without bounds checks.  A "view" is a channel slice [coff, coff+c) of a tensor --
// that is how C2f split/concat and the neck concats exist without any copy kernel.
// ---------------------------------------------------------------------------------------
struct TensorView {
    f16 *base = nullptr;   // tensor base (border included), channel offset NOT applied
    int H = 0, W = 0;      // interior size
    int C = 0;             // channels of the underlying tensor (pixel stride)
    int pad = 0;           // border width (0 or 1)
    int coff = 0, c = 0;   // the slice
    int padded_w() const { return W + 2 * pad; }
};

// Tile configurations of the conv kernels (conv.hip, conv_pp.hip).  Round 4 pruned the table to what the autotuner picks somewhere between 1 and 32
// frames per launch on the n / s / m widths (profiles/r04/tuner_wins_*.txt: 23 of the 25 win somewhere) plus the tiles the general (cin % 32 != 0) path needs; the families that lost
// every A/B -- weight-stationary 1x1, software-pipelined k-loop, 16-wave and 144-KiB one-per-CU tiles, deeper 32-deep rings, the 4-wave 64-deep
// variants, the persistent and the 128-row 8-wave tap-reuse tiles -- are gone from the library (their measurements stay under profiles/r02, r03).
enum ConvTile {
    TILE_128x128 = 0, TILE_128x64 = 1, TILE_64x64 = 2, TILE_256x32 = 3, TILE_64x128 = 4,   // 4 waves share one tile, 32-deep k-steps, three stages (any cin % 8 == 0)
    TILE_WSK_64x64 = 5, TILE_WSK_32x64 = 6, TILE_WSK_64x32 = 7,                           // 4 waves split K over one tile (small-M layers)
    TILE_K64_64x64_S3 = 8,                                                                // 64-deep k-steps (cin % 64 == 0), 4 waves
    TILE_ROWS_128x32 = 9, TILE_ROWS_K64_64x64 = 10,                                       // 3x3/s1 tap-reuse kernel, 4 waves (32- / 64-deep chunks)
    TILE_TAIL_128x64 = 11, TILE_TAIL_K64_128x128 = 12,                                    // conv + fused 1x1 tail (BN == cout)
    TILE_K64_128x128_S2_W8 = 13, TILE_K64_128x128_S3_W8 = 14, TILE_K64_128x64_S3_W8 = 15, TILE_K64_256x64_S2_W8 = 16,   // 64-deep, 8 waves per workgroup
    TILE_ROWS_256x64_W8 = 17,                                                             // tap-reuse kernel, 8 waves
    TILE_PT_128x128_S2 = 18, TILE_PT_128x64_S2 = 19,                                      // PERSISTENT 64-deep tile kernel: a workgroup walks over pixel tiles, the ring keeps prefetching across tile boundaries
    // 3x3 / stride-1 PING-PONG kernel (conv_pp.hip): one persistent 8-wave workgroup per CU, its two halves one barrier interval apart (one reads +
    // issues DMA while the other multiplies), tap reuse; 256 positions x BN couts, or 512 positions x 64 couts (the wide form for 64-cout convs)
    TILE_PP_256x128 = 20, TILE_PP_256x64 = 21, TILE_PP_256x192 = 22, TILE_PP_512x64 = 23,
    TILE_PPT_256x128 = 24,                                                                // the ping-pong schedule without tap reuse (1x1, 3x3 stride 2): one conv per launch, cin % 64 == 0, K >= 192
    TILE_COUNT = 25
};
const char *tile_name(int tile);
bool tile_needs_cin64(int tile);
bool tile_is_rows(int tile);      // 3x3 stride-1 only, bordered input
bool tile_is_pp(int tile);        // ping-pong 3x3 / stride-1 kernel (conv_pp.hip): cin % 64 == 0, groups allowed
bool tile_is_ppt(int tile);       // ping-pong tile kernel without tap reuse (conv_pp.hip): one conv, cin % 64 == 0, K >= 192, no residual / second destination
bool tile_is_tail(int tile);      // runs ConvLaunch::tail_* as well; needs cout == the tile's BN
bool tile_reads_lo(int tile);     // can serve ConvLaunch::in_lo
bool tile_is_w8(int tile);        // 8-wave 64-deep tile kernel (conv_mfma64_w8): one conv per launch
bool tile_is_pt(int tile);        // persistent 64-deep tile kernel: one conv per launch, cin % 64 == 0, full tiles, no second destination
struct TileShape { int bm, bn; };
TileShape tile_shape(int tile);

struct ConvLaunch {
    TensorView in, out, res;      // res.base == nullptr -> no residual
    TensorView out2;              // optional: the output is ALSO written nearest-2x upsampled into this slice
    // optional (1x1 stride-1 only, 64-deep tiles): channels [0, lo_c) of the input are read from this HALF-resolution tensor,
    // nearest-2x upsampled on the fly (Upsample + Concat of the neck without the upsampled copy); `in` covers [lo_c, cin)
    TensorView in_lo; int lo_c = 0;
    const f16 *wt = nullptr;      // [cout_pad][kp] fp16, K order (kh, kw, cin), zero padded
    const float *bias = nullptr;  // [cout_pad]
    int B = 1;
    int cin = 0, cout = 0, ks = 1, stride = 1, act = 1;
    int kp = 0;                   // weight row stride (K rounded up to 32)
    int tile = TILE_128x128;
    // optional fused tail: a 1x1 stride-1 conv over this conv's output, which is then never stored (TILE_TAIL_* only)
    TensorView tail_out; const f16 *tail_wt = nullptr; const float *tail_bias = nullptr; int tail_cout = 0, tail_kp = 0, tail_act = 1;
    int epilogue = 1;             // 0: 8-byte stores from the accumulator layout; 1: 16-byte stores through LDS in the tile kernels; 2: also in the tap-reuse kernel
};

int launch_conv(const ConvLaunch &c, hipStream_t s);
bool conv_pp_index_fits(const ConvLaunch &c);   // the ping-pong kernels' 24-bit index bound (tile_math.h: pp_index_fits); part of the tuner's candidate filter
// n (<= 6) independent convolutions sharing one tile configuration, in ONE launch
int launch_conv_group(const ConvLaunch *c, int n, int tile, hipStream_t s);

// A whole Bottleneck (conv3x3+SiLU -> conv3x3+SiLU [+ residual]) in one launch, intermediate kept in LDS
// (bottleneck.hip).  c in {32, 64, 128}; w1/w2 in the conv layout above with kp == 9*c.
struct BottleneckLaunch {
    TensorView in, out, res;      // res.base == nullptr -> no shortcut
    const f16 *w1 = nullptr, *w2 = nullptr;
    const float *b1 = nullptr, *b2 = nullptr;
    const f16 *zeros = nullptr;   // >= 16 bytes of zeros in device memory
    int B = 1, c = 0, kp = 0;
    // optional C2f.cv2 tail (last Bottleneck of a C2f with n = 1, c in {32, 64}): out2 = act(W . [tail_in (2c channels) | this output] + b);
    // only tail_out is stored
    TensorView tail_in, tail_out; const f16 *tail_wt = nullptr; const float *tail_bias = nullptr; int tail_cout = 0, tail_kp = 0, tail_act = 1;
    int persistent32 = 1;         // c = 32 with the tail: the persistent two-workgroups-per-CU kernel of bneck32.hip where it applies (0: bottleneck_fused, the A/B and test baseline)
};
bool bottleneck_supported(int c);
int launch_bottleneck(const BottleneckLaunch &l, hipStream_t s);
// bneck32.hip: the c = 32 Bottleneck + C2f.cv2 tail (YOLOv8s layer 2) as persistent workgroups with cross-tile prefetch; bit-identical to bottleneck_fused
bool bottleneck32_tail_supported(const BottleneckLaunch &l);
int launch_bottleneck32_tail(const BottleneckLaunch &l, hipStream_t s);

// stem: 3x3 stride-2 conv on the 4-channel (RGB0) padded fp16 image, cout in {16,32,48,64,80};
// wm = [cout][64] fp16 in the k' = kh*16 + kw*4 + c order (zero where kw == 3, c == 3 or k' >= 48)
int launch_stem(const TensorView &img4, const TensorView &out, const f16 *wm, const float *bias, int B,
                int cout, hipStream_t s);
// the same conv reading the BGR uint8 frames themselves (letterbox folded in); only for frames that need no
// resize.  lut: 256 fp16 values c/255.  Frames frame0 .. frame0+B-1 of `frames`.
struct LetterboxGeom { int src_h, src_w, new_w, new_h, top, left, resize; };
// frames: up to 64 device pointers to BGR uint8 images with row pitch `pitch`, passed by value in
// the kernel arguments (no pointer table to upload, nothing to race with launch-ahead)
struct FramePtrs { const uint8_t *p[64]; };
int launch_stem_fused(const FramePtrs &frames, int frame0, int pitch, const LetterboxGeom &g, int in_h, int in_w, const f16 *lut,
                      const TensorView &out, const f16 *wm, const float *bias, int B, int cout, hipStream_t s);
// The net's front end in ONE launch (front.hip): [letterbox +] stem (3 -> c0, 3x3 / s2) -> layer 1 (c0 -> c1, 3x3 / s2) -> C2f.cv1 of layer 2 (1x1, c1 -> c2);
// only the last conv's output is stored.  Source: the frames' bytes (frames that need no resize) or the letterboxed RGB0 image tensor.
struct FrontLaunch {
    FramePtrs frames; int frame0 = 0, pitch = 0; LetterboxGeom g{};      // byte source
    TensorView img4; bool from_tensor = false;                           // tensor source
    const f16 *zeros = nullptr;                                          // >= 16 bytes of zeros in device memory
    const f16 *w0 = nullptr, *w1 = nullptr, *w2 = nullptr;               // stem [c0][64] (k' = kh*16 + kw*4 + c), layer 1 [c1][kp1] (kh, kw, cin), 2.cv1 [c2][kp2]
    const float *b0 = nullptr, *b1 = nullptr, *b2 = nullptr;
    int kp1 = 0, kp2 = 0;
    TensorView out;                                                      // 2.cv1's output view
    int B = 1, c0 = 0, c1 = 0, c2 = 0, in_h = 0, in_w = 0;
};
bool front_supported(int c0, int c1, int c2, int in_h, int in_w);
int launch_front(const FrontLaunch &l, hipStream_t s);
// SPPF: y -> (max5(y), max5(max5(y)), max5^3(y)) written to three channel slices of the same tensor
int launch_sppf_pool(const TensorView &y, const TensorView &p1, const TensorView &p2, const TensorView &p3, int B,
                     hipStream_t s);
// nearest 2x upsample of a view into a channel slice of a tensor twice the size
int launch_upsample2(const TensorView &in, const TensorView &out, int B, hipStream_t s);

// ---------------------------------------------------------------------------------------
// preprocess (preprocess.hip)
// ---------------------------------------------------------------------------------------
struct ResizeTables {          // device arrays, cv::resize INTER_LINEAR fixed-point tables
    const int32_t *xofs, *xa0, *xa1, *yofs, *yb0, *yb1;
};
int launch_letterbox(const FramePtrs &frames, int pitch, const LetterboxGeom &g, const ResizeTables &t,
                     const TensorView &img4, int B, hipStream_t s);

// ---------------------------------------------------------------------------------------
// postprocess (postprocess.hip)
// ---------------------------------------------------------------------------------------
struct HeadLevel { const f16 *ptr; int H, W, stride; };   // [B][H][W][64+nc] fp16, no border
struct DecodeArgs {
    HeadLevel lvl[3];
    int B, nc, n_anchors;
    int no;                    // halves per anchor row in the head tensors (64 + nc rounded up to 8)
    float conf;
    uint64_t class_mask[2];    // allowed classes (bit per class id < 128)
    // per-anchor dense outputs [B][A]
    float4 *box;               // xyxy in network-input pixels
    float *score;              // best class score, or -1 when not a candidate
    int32_t *cls;
    float *pred;               // optional full (4+nc) x A tensor per image (debug / parity), or nullptr
};
int launch_decode(const DecodeArgs &a, hipStream_t s);

// Detect's last 1x1 convs (cv2.l.2: cbox -> 64 DFL logits, cv3.l.2: ccls -> nc class logits, no activation) and the
// decode above in ONE launch: the (64 + nc)-logit rows of 64 anchors are produced in LDS (fp16, as the head tensor
// would hold them) and decoded there; the head tensor is written only when `heads` is given (debug).
struct HeadFinalLevel {
    const f16 *xb, *xc;            // [B][H*W][cbox], [B][H*W][ccls]: inputs of the two convs (no border)
    const f16 *wb, *wc;            // weights [64 -> 128 rows][cbox], [nc -> 128 rows][ccls]
    const float *bb, *bc;          // biases (128 entries)
    f16 *heads;                    // optional [B][H*W][no]
    int H, W, stride, tiles;       // tiles = ceil(H*W / 128) per image
};
struct HeadFinalArgs {
    HeadFinalLevel lvl[3];
    int B, nc, n_anchors, cbox, ccls, no;
    float conf; uint64_t class_mask[2];
    float4 *box; float *score; int32_t *cls;
};
bool head_final_supported(int cbox, int ccls, int nc);
int launch_head_final(const HeadFinalArgs &a, hipStream_t s);

struct NmsArgs {
    int B, n_anchors, max_det, agnostic;
    float iou;
    const float4 *box; const float *score; const int32_t *cls;     // dense per-anchor, [B][A]
    // scratch [B][A]
    uint64_t *keys; float4 *sbox; int32_t *sidx;
    // rescale to the source frame (scale_boxes): x = (x - pad_x)/gain clipped to [0, src]
    float gain, pad_x, pad_y, src_w, src_h; int rescale;
    // outputs [B][max_det]
    float *out_xyxy; float *out_conf; int32_t *out_cls; int32_t *out_anchor; int32_t *out_n;
};
// which form of the kernel runs: resolved from the options once per detector (nms_plan_from_options), never per launch
struct NmsPlan { int threads = 1024; int rank_max = 512; int stop = 0; };
NmsPlan nms_plan_from_options();
int launch_nms(const NmsArgs &a, const NmsPlan &plan, hipStream_t s);
// pred[(4+nc)][A] float32 -> dense per-anchor candidates (what decode emits), for rtmodt_nms_pred
int launch_pred_candidates(const float *pred, int nc, int n_anchors, float conf, const uint64_t class_mask[2],
                           float4 *box, float *score, int32_t *cls, hipStream_t s);

// ---------------------------------------------------------------------------------------
// tracker (tracker.hip)
// ---------------------------------------------------------------------------------------
struct TrackerState {          // one stream; device pointers; double-buffered (cur = buffer index)
    int64_t *ids[2]; float4 *box[2]; float *conf[2]; int32_t *cls[2]; int32_t *age[2]; int32_t *tsu[2];
    // opt-in Kalman motion model (TrackerArgs::kalman): [5][max_tracks] float4 per buffer =
    // mean (cx, cy, a, h), velocities, and the a / b / c entries of the four 2x2 covariance blocks
    float4 *kf[2];
};
struct TrackerArgs {
    int n_streams, stream_base, max_tracks, max_dets;   // grid = n_streams workgroups, stream index = stream_base + blockIdx.x
    float track_thresh, match_thresh; int track_buffer;
    int assign_mode;           // RTMODT_ASSIGN_GREEDY | RTMODT_ASSIGN_LAPJV
    int kalman;                // 1: constant-velocity Kalman predict / update per track (opt-in; the reference has none)
    double cost_limit;         // lapjv: 1 - match_thresh evaluated in double (tracker.py:170)
    TrackerState *states;      // device array [n_streams]
    int64_t *meta;             // device [n_streams][8]: {cur, n_tracks, err, n_active, next_id, 0, 0, 0}
    // detections: [n_streams][det_stride] boxes / conf / cls ; counts [n_streams]
    const float4 *det_box; const float *det_conf; const int32_t *det_cls; const int32_t *det_n; int det_stride;
    // frames per launch: stream s consumes the detection slots s, s + frame_step, ..., in order, inside its workgroup
    int n_frames = 1, frame_step = 0;
};
int launch_tracker_update(const TrackerArgs &a, hipStream_t s);
int launch_iou_matrix(const float4 *a, int m, const float4 *b, int n, float *out, hipStream_t s);
int launch_assign_greedy(const float *iou, int m, int n, float thresh, int32_t *row_to_col, int32_t *col_used,
                         hipStream_t s);
int launch_assign_lapjv(const float *iou, int m, int n, double cost_limit, int32_t *row_to_col, int32_t *col_used, int32_t *err,
                        hipStream_t s);

// the tracker's device-resident state (tracker_api.hip), consumed by the zone engine (zones.hip):
// states[n_streams], meta[n_streams][8] = {cur, n_tracks, err, n_active, next_id, ...}; `stream` is the HIP stream
// the tracker's most recent update was launched on
struct TrackerDeviceView { const TrackerState *states; const int64_t *meta; int n_streams, max_tracks, device; hipStream_t stream; };
int tracker_device_view(rtmodt_tracker *trk, TrackerDeviceView *out);

// device-resident results of a detector's last enqueue_batch (engine.hip), consumed by the tracker
struct DetOutputs { const float4 *box; const float *conf; const int32_t *cls; const int32_t *n; int stride, count, device; hipStream_t stream; };
int detector_outputs(rtmodt_detector *det, DetOutputs *out);

}  // namespace rtmodt
