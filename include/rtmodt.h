/* rtmodt.h -- C ABI of librtmodt_hip.so, the MI355X (gfx950) detect + track hot path.
 *
 * Drop-in boundary for the reference's two hot-path classes:
 *   src/detection/detector.py:54-135   class Detector          -> rtmodt_detector_*
 *   src/tracking/tracker.py:43-194     class _ByteTrackCore    -> rtmodt_tracker_*
 * The Python classes of the same names in the package call these entry points through
 * ctypes; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: plain C, opaque handles, host pointers unless a parameter says "device".
 * Every function returns 0 on success or a negative RTMODT_E_* code; the message for the
 * calling thread's last failure is rtmodt_last_error().  No exceptions cross the ABI.
 * The library owns all device memory (one static arena per handle).  Handles are not
 * thread-safe: one Detector + one tracker per stream group, used from one thread
 * (as the reference's single main loop does, tools/run_pipeline.py:121-166).
 */
#ifndef RTMODT_H
#define RTMODT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMODT_OK 0
#define RTMODT_E_INVALID (-1)   /* bad argument / unsupported configuration            */
#define RTMODT_E_IO (-2)        /* weight file missing or malformed                    */
#define RTMODT_E_HIP (-3)       /* HIP runtime failure (message carries hipGetErrorString) */
#define RTMODT_E_CAPACITY (-4)  /* more tracks / detections than the handle was sized for */
#define RTMODT_E_UNSUPPORTED (-5)

#define RTMODT_MEM_HOST 0
#define RTMODT_MEM_DEVICE 1

#define RTMODT_ASSIGN_GREEDY 0  /* tracker.py:182-194 (the branch taken when `lap` is absent) */
#define RTMODT_ASSIGN_LAPJV 1   /* tracker.py:168-181 (the branch taken when `lap` is importable): exact optimal assignment
                                 * under cost_limit = 1 - match_thresh; PARITY UNPINNED (no `lap` to run against) */

typedef struct rtmodt_detector rtmodt_detector;
typedef struct rtmodt_tracker rtmodt_tracker;
typedef struct rtmodt_zones rtmodt_zones;

/* ---- library / device ------------------------------------------------------------- */
const char *rtmodt_last_error(void);
const char *rtmodt_version(void);
/* "csrc_sha256=<digest of the kernel sources this binary was built from> diag=<0|1>": bench.py attaches a measured
 * roofline.traffic only to a library whose digest equals the one the counters were collected on */
const char *rtmodt_build_info(void);
/* The value of the run-time option RTMODT_<name> exactly as the library reads it (NULL when unset).  Options are read at
 * detector-create / autotune time only, from ONE table (csrc/common.h); a name outside it is RTMODT_E_INVALID -- never an abort. */
int rtmodt_option(const char *name, const char **value);
int rtmodt_device_count(int *count);
int rtmodt_synchronize(int device);                 /* replaces torch.cuda.synchronize() (latency_profiler.py:63,69) */
int rtmodt_device_alloc(int device, size_t bytes, void **out);
int rtmodt_device_free(int device, void *ptr);
/* Page-locked host memory for the frame source (the ring RTSPReader.read() copies out of,
 * src/ingestion/rtsp_reader.py:74-79): enqueue_batch copies host frames on its own HIP stream into a
 * per-slot staging area, so frames that live in such memory upload underneath the previous
 * batch's forward pass; pageable frames still work but their copy blocks the calling thread. */
int rtmodt_host_alloc(int device, size_t bytes, void **out);
int rtmodt_host_free(int device, void *ptr);
int rtmodt_memcpy_h2d(int device, void *dst_device, const void *src_host, size_t bytes);
int rtmodt_memcpy_d2h(int device, void *dst_host, const void *src_device, size_t bytes);

/* ---- detector: replaces Detector.__init__/detect/_parse (detector.py:59-129) ------- */
typedef struct rtmodt_det_cfg {
    const char *weight_path;   /* RTMODTW1 file (package weights.py); model scale/nc come from its header */
    int32_t in_w, in_h;        /* network input = letterbox target; multiples of 32 (detector.py:61, :102) */
    float conf;                /* detector.py:62  confidence=0.35  (strict >, float32)                      */
    float iou;                 /* detector.py:63  iou=0.45         (strict >)                               */
    const int32_t *classes;    /* detector.py:64  class filter, NULL = all                                 */
    int32_t n_classes;
    int32_t half;              /* detector.py:65,76  must be 1: the engine stores activations in fp16       */
    int32_t device;            /* detector.py:66  ordinal of "cuda:N"                                      */
    int32_t max_det;           /* detector.py:67  max_det=100                                              */
    int32_t agnostic;          /* detector.py:68  agnostic_nms                                             */
    int32_t batch;             /* frames per detect_batch call (streams batched on this GPU), >= 1         */
    int32_t max_src_w, max_src_h; /* largest source frame accepted (staging), 0 = in_w/in_h                */
    int32_t use_graph;         /* 1: replay the forward pass as one captured hipGraph                      */
    int32_t autotune;          /* 1: time every conv tile configuration at create and keep the fastest      */
    int32_t chains;            /* sub-batches that run as independent chains (stem -> graph -> decode) on their own streams;
                                * -1 / -2 = STAGED with S = 2 / 3 stages: the whole batch per launch, the net cut into S
                                * stages on S streams, stage 1 of batch t + 1 overlapping stage 2 of batch t (S arena
                                * copies; keep S + 1 batches in flight: enqueue t + S before fetching t).  Three stages
                                * take every hardware queue of the process: use them when the frames are already in
                                * device memory, two when they come from the host through the engine's copy stream;
                                * 0 = automatic: two stages for batch >= 2, the plain single-stream engine for batch 1;
                                * 1 = plain engine; n > 1 = n sub-batch chains */
    int32_t rect;              /* 1: minimal-rectangle letterbox of `predict` on a .pt model (LetterBox auto=True): the scale is
                                * min(S/h, S/w) with S = max(in_w, in_h) and in_w x in_h is the rectangle (1080p: 640 x 384) */
} rtmodt_det_cfg;

int rtmodt_detector_create(const rtmodt_det_cfg *cfg, rtmodt_detector **out);
void rtmodt_detector_destroy(rtmodt_detector *det);

/* One frame, synchronous: letterbox -> forward -> decode -> NMS -> rescale -> D2H.
 * bgr: H x W x 3 uint8, row pitch stride_bytes (host).  Outputs caller-allocated:
 * xyxy[max_det*4], conf[max_det], cls[max_det]; *n_out = number of detections. */
int rtmodt_detector_detect(rtmodt_detector *det, const uint8_t *bgr, int h, int w, int stride_bytes,
                           float *xyxy, float *conf, int32_t *cls, int32_t *n_out);

/* n <= cfg.batch frames of identical size in one pass (BASELINE configs 4-5: streams
 * batched per GPU).  frames[i] is a host or device pointer per mem_kind.  Outputs are
 * [n][max_det] blocks; n_out[n]. */
int rtmodt_detector_detect_batch(rtmodt_detector *det, const uint8_t *const *frames, int n, int h, int w,
                                 int stride_bytes, int mem_kind, float *xyxy, float *conf, int32_t *cls,
                                 int32_t *n_out);

/* Asynchronous halves of detect_batch for the throughput path: enqueue leaves the
 * detections on the device (consumable by rtmodt_tracker_update_from_detector on the same
 * HIP stream, no host round trip) and starts their copy to pinned host memory; fetch waits
 * for the OLDEST batch in flight and hands it out.  Up to two batches may be in flight
 * (enqueue t+1, then fetch t), which hides the host's per-step work behind the GPU. */
int rtmodt_detector_enqueue_batch(rtmodt_detector *det, const uint8_t *const *frames, int n, int h, int w,
                                  int stride_bytes, int mem_kind);
int rtmodt_detector_fetch(rtmodt_detector *det, float *xyxy, float *conf, int32_t *cls, int32_t *n_out);

/* Introspection used by the parity tests and bench.py */
int rtmodt_detector_info(rtmodt_detector *det, int32_t *scale_id, int32_t *nc, int32_t *n_anchors,
                         int32_t *n_convs, int64_t *conv_flops_per_frame, int64_t *arena_bytes);
/* Number of sub-batch chains this detector runs its batch as (see rtmodt_det_cfg.chains). */
int rtmodt_detector_chains(rtmodt_detector *det, int32_t *n_chains);
/* Stages the detector runs as (rtmodt_det_cfg.chains = -1 / -2 and a hardware queue found for each): 2 or 3, else 1. */
int rtmodt_detector_stages(rtmodt_detector *det, int32_t *n_stages);
/* Copies out, for frame `img` of the last batch: the letterboxed network input as fp16 NHWC(3)
 * [in_h*in_w*3] (may be NULL), the three Detect maps as fp16 [A_i*(64+nc)] concatenated
 * P3,P4,P5 (may be NULL) and the decoded pre-NMS tensor pred[(4+nc)*A] float32 (may be NULL). */
int rtmodt_detector_debug_fetch(rtmodt_detector *det, int img, uint16_t *input_f16, uint16_t *heads_f16, float *pred);
/* Output of fused conv `name` ("4.cv2", "22.cv3.0.1", ...) for frame img as fp16 NHWC [H*W*C];
 * shape returned through hwc[3]; out may be NULL to query the shape. */
int rtmodt_detector_debug_layer(rtmodt_detector *det, const char *name, int img, uint16_t *out, int32_t *hwc);
/* Per-launch device time of the last `iters` eager (non-graph) forwards, measured with HIP
 * events on the detector's stream: names[i] points into handle-owned storage. */
int rtmodt_detector_profile(rtmodt_detector *det, int iters, int max_entries, const char **names, float *ms,
                            int64_t *flops, int32_t *n_entries);
/* Device time (ms, HIP events on the detector's stream) of the batch the last fetch returned:
 * whole pass, and the letterbox + forward-graph part alone. */
int rtmodt_detector_last_timing(rtmodt_detector *det, float *total_ms, float *forward_ms);
/* The same batch split into the stages the reference's profiler names (latency_profiler.py:38):
 * preprocess = letterbox, inference = forward pass + decode, nms = NMS + rescale (device ms). */
int rtmodt_detector_stage_times(rtmodt_detector *det, float *preprocess_ms, float *inference_ms, float *nms_ms);
/* In-kernel shader clock (measurement aid, no reference counterpart): while enabled, one wave behind every batch's NMS reads the
 * shader-cycle counter against the constant 100 MHz counter for ~20 us; _read waits for the post-processing stream and returns
 * mean / min / max GHz over the samples since it was enabled (or read last).  The MFMA peak at THAT clock, not at the 2.4 GHz of
 * the data sheet, is what the matrix cores could have delivered during the run. */
int rtmodt_detector_clock_enable(rtmodt_detector *det, int on);
int rtmodt_detector_clock_read(rtmodt_detector *det, double *ghz_mean, double *ghz_min, double *ghz_max, int32_t *n_samples);

/* decode-free NMS on a caller-supplied pre-NMS tensor pred[(4+nc)*A] float32 (the layout
 * ultralytics' non_max_suppression receives): BASELINE config 2's
 * "NMS correctness" case.  dets out: xyxy in the tensor's own coordinates (no rescale). */
int rtmodt_nms_pred(int device, const float *pred, int nc, int n_anchors, float conf, float iou,
                    const int32_t *classes, int n_classes, int agnostic, int max_det,
                    float *xyxy, float *conf_out, int32_t *cls, int32_t *anchor_idx, int32_t *n_out);

/* letterbox + BGR->RGB + /255 alone (ultralytics LetterBox + cv2.resize INTER_LINEAR restated):
 * out fp16 NHWC [in_h*in_w*3]. */
int rtmodt_preprocess(int device, const uint8_t *bgr, int h, int w, int stride_bytes, int in_w, int in_h,
                      uint16_t *out_f16);

/* ---- tracker: replaces _ByteTrackCore (tracker.py:43-194) ---------------------------- */
/* n_streams independent tracker states updated by ONE launch (one workgroup per stream). */
int rtmodt_tracker_create(int device, float track_thresh, int track_buffer, float match_thresh,
                          int assign_mode, int max_tracks, int max_dets, int n_streams, rtmodt_tracker **out);
void rtmodt_tracker_destroy(rtmodt_tracker *trk);
/* RTMODT_ASSIGN_LAPJV only: the reference evaluates `cost_limit = 1 - thresh` in Python doubles
 * (tracker.py:170); match_thresh above is a float, so a caller that wants the identical limit
 * passes it here.  Default: 1.0 - (double)match_thresh. */
int rtmodt_tracker_set_cost_limit(rtmodt_tracker *trk, double cost_limit);

/* One frame for one stream (tracker.py:58-141).  *n_active_out = tracks with
 * time_since_update == 0 after the update -- always 0, as in the reference (SURVEY finding 4). */
int rtmodt_tracker_update(rtmodt_tracker *trk, int stream, const float *xyxy, const float *conf,
                          const int32_t *cls, int n, int32_t *n_active_out);
/* One frame for every stream: xyxy[n_streams][max_dets][4], conf/cls[n_streams][max_dets], n[n_streams]. */
int rtmodt_tracker_update_batch(rtmodt_tracker *trk, const float *xyxy, const float *conf, const int32_t *cls,
                                const int32_t *n, int32_t *n_active_out);
/* Consumes the device-resident detections of det's last enqueue_batch (stream i <- frame i),
 * asynchronously on det's HIP stream. */
int rtmodt_tracker_update_from_detector(rtmodt_tracker *trk, rtmodt_detector *det);
/* The same for frames [first_frame, first_frame + n_frames) of det's batch (stream i <- frame first_frame + i):
 * a batch that holds several CONSECUTIVE frames of every stream (frame-major: image f * n_streams + s) is
 * tracked by calling this once per f, in order -- tracker.py:58-141 still sees each stream's frames one at a time. */
int rtmodt_tracker_update_from_detector_frames(rtmodt_tracker *trk, rtmodt_detector *det, int first_frame, int n_frames);
/* A frame-major batch (image f * n_streams + s, starting at first_frame) of n_frames consecutive frames of n_streams streams
 * in ONE launch: stream s's workgroup walks over its n_frames detection slots in order, so every stream still sees
 * tracker.py:58-141 one frame at a time; state after the call == n_frames calls of _update_from_detector_frames. */
int rtmodt_tracker_update_from_detector_batch(rtmodt_tracker *trk, rtmodt_detector *det, int first_frame, int n_streams, int n_frames);
/* List-order snapshot of a stream's state = the reference's _core._tracks + _core._next_id
 * (the parity surface).  Arrays sized max_tracks; any may be NULL. */
int rtmodt_tracker_state(rtmodt_tracker *trk, int stream, int64_t *ids, float *xyxy, float *conf,
                         int32_t *cls, int32_t *age, int32_t *tsu, int32_t *n, int64_t *next_id);
int rtmodt_tracker_reset(rtmodt_tracker *trk, int stream);   /* stream < 0: all */
/* OPT-IN, no reference counterpart (the reference overwrites a matched track's box, tracker.py:99-104, and has no motion
 * model): ByteTrack's published 8-state constant-velocity Kalman filter over (cx, cy, a, h), batched inside the same
 * launch -- every track is predicted at the start of a frame, association runs on the predicted boxes, a matched track
 * is corrected with its detection, a new track is initiated from it.  `xyxy` in rtmodt_tracker_state keeps the
 * reference's meaning (the last matched detection).  Call before the first update.  oracle/kalman_oracle.py. */
int rtmodt_tracker_enable_kalman(rtmodt_tracker *trk);
/* Filter state in list order: mean[n][8] = (cx, cy, a, h, vx, vy, va, vh); cov[n][12] = per coordinate the (a, b, c)
 * entries of its 2x2 covariance block [[a, b], [b, c]] (the 8x8 covariance is block-diagonal by construction). */
int rtmodt_tracker_kalman_state(rtmodt_tracker *trk, int stream, float *mean, float *cov, int32_t *n);

/* _ByteTrackCore._batch_iou (tracker.py:150-161) alone: out[m*n] float32, bit-exact. */
int rtmodt_iou_matrix(int device, const float *a, int m, const float *b, int n, float *out);
/* _linear_assignment greedy branch (tracker.py:182-194) alone on a caller-supplied matrix:
 * row_to_col[m] (-1 = unmatched), col_used[n]. */
int rtmodt_assign_greedy(int device, const float *iou, int m, int n, float thresh, int32_t *row_to_col,
                         int32_t *col_used);
/* _linear_assignment lap.lapjv branch (tracker.py:168-181) alone: the optimal assignment of
 * cost = 1 - iou (float32) extended with cost_limit, i.e. the maximum-gain matching over pairs
 * with cost < cost_limit.  RTMODT_E_CAPACITY when more than 256 rows / 256 columns / 2048 pairs
 * are contested (share a row or column with another candidate pair). */
int rtmodt_assign_lapjv(int device, const float *iou, int m, int n, double cost_limit, int32_t *row_to_col,
                        int32_t *col_used);

/* ---- zone events: replaces ZoneEventEngine.process (src/events/zone_engine.py:82-132) -------- */
/* One polygon zone (zone_engine.py:50-58, :142-151).  `key` = index of the FIRST zone carrying the
 * same name: the reference keys its occupancy and cooldown dicts by zone name (:98-100, :105), so
 * same-named zones share their timers; distinct names -> key == own index. */
typedef struct rtmodt_zone_cfg {
    const int32_t *polygon_xy;  /* n_points x (x, y), int32 like np.array(cfg["polygon"], dtype=np.int32) (:143) */
    int32_t n_points;
    double dwell_time_sec;      /* default 2.0  (:147) */
    double cooldown_sec;        /* default 10.0 (:148) */
    int32_t key;
} rtmodt_zone_cfg;
/* n_streams independent ledgers (occupancy + cooldown per track id and zone), at most 32 zones /
 * 2048 polygon points.  max_idle_frames: a track id not passed for more than this many frames loses
 * its cooldown entries (the reference never drops them, zone_engine.py:76; pass INT64_MAX/2 to
 * mirror that until the 2 x max_tracks ledger fills -> RTMODT_E_CAPACITY). */
int rtmodt_zones_create(int device, const rtmodt_zone_cfg *zones, int n_zones, int n_streams, int max_tracks,
                        int max_events, int64_t max_idle_frames, rtmodt_zones **out);
void rtmodt_zones_destroy(rtmodt_zones *z);
/* process(tracks, frame_id) for one stream on a caller-supplied track list (any order, unique ids);
 * `now` = the reference's time.time() (:84).  Events come back in the reference's order (track
 * order, then zone order): ev_track = index into the caller's list, ev_zone = zone index,
 * ev_dwell = now - first_seen (the reference rounds it to 2 decimals when it builds the record,
 * :113), ev_centroid[2] = int((x1+x2)/2), int((y1+y2)/2) (:91-92).  Outputs sized max_events. */
int rtmodt_zones_process(rtmodt_zones *z, int stream, const int64_t *track_ids, const float *xyxy,
                         const int32_t *cls, int n, double now, int64_t frame_id, int32_t *ev_track,
                         int32_t *ev_zone, double *ev_dwell, int32_t *ev_centroid, int32_t *n_events);
/* The same for every stream of `trk` at once, straight on its device-resident state and on the
 * HIP stream its last update ran on (no host round trip for the tracks).  The tracks "passed" are
 * those with time_since_update == report_tsu after the update (1 = matched or spawned this frame;
 * 0 = what the reference's tracker returns, i.e. none -- SURVEY finding 4).  Outputs are
 * [n_streams][max_events] (+ [4] / [2] for xyxy / centroid), n_events[n_streams]. */
int rtmodt_zones_process_tracker(rtmodt_zones *z, rtmodt_tracker *trk, double now, int64_t frame_id,
                                 int report_tsu, int64_t *ev_track_id, int32_t *ev_zone, double *ev_dwell,
                                 float *ev_xyxy, int32_t *ev_centroid, int32_t *ev_cls, int32_t *n_events);
/* Ledger snapshot of one stream, rows in ascending track id (= _occupancy and _cooldown, :74-76):
 * occ_mask bit k <=> zone key k is in _occupancy[track]; first_seen / last_alert are
 * [rows][n_zones] indexed by key (last_alert 0.0 = no entry).  Arrays sized 2 x max_tracks rows. */
int rtmodt_zones_state(rtmodt_zones *z, int stream, int64_t *ids, uint32_t *occ_mask, double *first_seen,
                       double *last_alert, int32_t *n);

#ifdef __cplusplus
}
#endif
#endif /* RTMODT_H */
