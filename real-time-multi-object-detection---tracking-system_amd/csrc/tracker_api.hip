// tracker_api.hip -- the tracker half of include/rtmodt.h: handle lifetime, staging, and
// the three ways a frame's detections reach the kernel in tracker.hip (host arrays for one
// stream, host arrays for all streams, or the detector's device-resident outputs).
// Mirrors _ByteTrackCore's constructor and update() (/root/reference/src/tracking/tracker.py:46-141).
#include <cstring>
#include <vector>

#include "kernels.h"

using namespace rtmodt;

struct rtmodt_tracker {
    int device = 0;
    hipStream_t stream = nullptr;
    // An update fed from a detector runs on THAT detector's post-processing stream (ordered behind its NMS, no host hop).
    // The tracker never keeps the foreign stream handle: it records `foreign_done` there, and everything it later does on
    // its own stream (host-fed updates, state read-back, reset, destroy) waits for that event first.
    hipEvent_t foreign_done = nullptr;
    bool foreign_pending = false;
    bool kalman = false;                  // opt-in motion model (rtmodt_tracker_enable_kalman)
    char *kf_pool = nullptr;
    int S = 1, Mc = 0, Nc = 0;
    float track_thresh = 0.5f, match_thresh = 0.8f;
    int track_buffer = 30;
    int assign_mode = RTMODT_ASSIGN_GREEDY;
    double cost_limit = 0.2;
    // device
    char *pool = nullptr;                 // all state arrays
    TrackerState *d_states = nullptr;
    std::vector<TrackerState> h_states;
    int64_t *d_meta = nullptr;
    float4 *d_box = nullptr; float *d_conf = nullptr; int32_t *d_cls = nullptr; int32_t *d_n = nullptr;   // staging [S][Nc]
    // pinned host
    int64_t *h_meta = nullptr;
    int32_t *h_n = nullptr;
    char *h_state = nullptr;              // one stream's state arrays (rtmodt_tracker_state): 40 bytes per track
};

static int64_t init_meta_row[8] = {0, 0, 0, 0, 1, 0, 0, 0};   // cur, n_tracks, err, n_active, next_id (tracker.py:55)

// make the tracker's own stream wait for the last update that ran on a detector's stream
static int join_foreign(rtmodt_tracker *t) {
    if (t->foreign_pending) {
        RT_HIP(hipStreamWaitEvent(t->stream, t->foreign_done, 0));
        t->foreign_pending = false;
    }
    return RTMODT_OK;
}

namespace rtmodt {
int tracker_device_view(rtmodt_tracker *t, TrackerDeviceView *out) {
    RT_CHECK(t && out, RTMODT_E_INVALID, "null argument");
    *out = TrackerDeviceView{t->d_states, t->d_meta, t->S, t->Mc, t->device, t->stream};
    return join_foreign(t);                                // the caller's work on t->stream is ordered behind every update
}
}  // namespace rtmodt

extern "C" {

void rtmodt_tracker_destroy(rtmodt_tracker *t) {
    if (!t) return;
    hipSetDevice(t->device);
    if (t->foreign_done) hipEventSynchronize(t->foreign_done);      // an update may still be queued on a detector's stream
    if (t->stream) hipStreamSynchronize(t->stream);
    if (t->foreign_done) hipEventDestroy(t->foreign_done);
    hipFree(t->kf_pool);
    hipFree(t->pool); hipFree(t->d_states); hipFree(t->d_meta);
    hipFree(t->d_box); hipFree(t->d_conf); hipFree(t->d_cls); hipFree(t->d_n);
    hipHostFree(t->h_meta); hipHostFree(t->h_n); hipHostFree(t->h_state);
    if (t->stream) hipStreamDestroy(t->stream);
    delete t;
}

static int tracker_create_impl(rtmodt_tracker *t) {
    RT_HIP(hipSetDevice(t->device));
    RT_HIP(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    RT_HIP(hipEventCreateWithFlags(&t->foreign_done, hipEventDisableTiming));
    const size_t per_buf = (size_t)t->Mc * (8 + 16 + 4 + 4 + 4 + 4);
    const size_t total = per_buf * 2 * t->S;
    RT_HIP(hipMalloc((void **)&t->pool, total));
    RT_HIP(hipMemset(t->pool, 0, total));
    t->h_states.resize(t->S);
    char *p = t->pool;
    for (int s = 0; s < t->S; ++s)
        for (int b = 0; b < 2; ++b) {
            TrackerState &st = t->h_states[s];
            st.ids[b] = (int64_t *)p; p += (size_t)t->Mc * 8;
            st.box[b] = (float4 *)p; p += (size_t)t->Mc * 16;
            st.conf[b] = (float *)p; p += (size_t)t->Mc * 4;
            st.cls[b] = (int32_t *)p; p += (size_t)t->Mc * 4;
            st.age[b] = (int32_t *)p; p += (size_t)t->Mc * 4;
            st.tsu[b] = (int32_t *)p; p += (size_t)t->Mc * 4;
            st.kf[b] = nullptr;
        }
    RT_HIP(hipMalloc((void **)&t->d_states, sizeof(TrackerState) * t->S));
    RT_HIP(hipMemcpy(t->d_states, t->h_states.data(), sizeof(TrackerState) * t->S, hipMemcpyHostToDevice));
    RT_HIP(hipMalloc((void **)&t->d_meta, sizeof(int64_t) * 8 * t->S));
    RT_HIP(hipHostMalloc((void **)&t->h_meta, sizeof(int64_t) * 8 * t->S, hipHostMallocDefault));
    RT_HIP(hipHostMalloc((void **)&t->h_n, sizeof(int32_t) * t->S, hipHostMallocDefault));
    RT_HIP(hipHostMalloc((void **)&t->h_state, (size_t)t->Mc * 40, hipHostMallocDefault));
    RT_HIP(hipMalloc((void **)&t->d_box, sizeof(float4) * t->Nc * t->S));
    RT_HIP(hipMalloc((void **)&t->d_conf, sizeof(float) * t->Nc * t->S));
    RT_HIP(hipMalloc((void **)&t->d_cls, sizeof(int32_t) * t->Nc * t->S));
    RT_HIP(hipMalloc((void **)&t->d_n, sizeof(int32_t) * t->S));
    RT_HIP(hipMemset(t->d_n, 0, sizeof(int32_t) * t->S));
    for (int s = 0; s < t->S; ++s) memcpy(t->h_meta + 8 * s, init_meta_row, sizeof(init_meta_row));
    RT_HIP(hipMemcpy(t->d_meta, t->h_meta, sizeof(int64_t) * 8 * t->S, hipMemcpyHostToDevice));
    return RTMODT_OK;
}

int rtmodt_tracker_create(int device, float track_thresh, int track_buffer, float match_thresh, int assign_mode, int max_tracks,
                          int max_dets, int n_streams, rtmodt_tracker **out) {
    RT_CHECK(out, RTMODT_E_INVALID, "null argument");
    RT_CHECK(assign_mode == RTMODT_ASSIGN_GREEDY || assign_mode == RTMODT_ASSIGN_LAPJV, RTMODT_E_UNSUPPORTED,
             "assign_mode %d: 0 (greedy, tracker.py:182-194) or 1 (lapjv, tracker.py:168-181)", assign_mode);
    RT_CHECK(max_tracks >= 1 && max_tracks <= 4096 && max_dets >= 1 && max_dets <= 4096 && n_streams >= 1 && n_streams <= 4096,
             RTMODT_E_INVALID, "max_tracks %d / max_dets %d / n_streams %d out of range", max_tracks, max_dets, n_streams);
    rtmodt_tracker *t = new rtmodt_tracker();
    t->device = device; t->S = n_streams; t->Mc = max_tracks; t->Nc = max_dets;
    t->track_thresh = track_thresh; t->match_thresh = match_thresh; t->track_buffer = track_buffer;
    t->assign_mode = assign_mode; t->cost_limit = 1.0 - (double)match_thresh;
    int rc = tracker_create_impl(t);
    if (rc != RTMODT_OK) {
        std::string keep = last_error();
        rtmodt_tracker_destroy(t);
        last_error() = keep;
        return rc;
    }
    *out = t;
    return RTMODT_OK;
}

int rtmodt_tracker_set_cost_limit(rtmodt_tracker *t, double cost_limit) {
    RT_CHECK(t && cost_limit == cost_limit, RTMODT_E_INVALID, "bad argument");
    t->cost_limit = cost_limit;
    return RTMODT_OK;
}

static TrackerArgs make_args(rtmodt_tracker *t) {
    TrackerArgs a{};
    a.n_streams = t->S; a.stream_base = 0; a.max_tracks = t->Mc; a.max_dets = t->Nc;
    a.track_thresh = t->track_thresh; a.match_thresh = t->match_thresh; a.track_buffer = t->track_buffer;
    a.assign_mode = t->assign_mode; a.cost_limit = t->cost_limit; a.kalman = t->kalman ? 1 : 0;
    a.states = t->d_states; a.meta = t->d_meta;
    a.det_box = t->d_box; a.det_conf = t->d_conf; a.det_cls = t->d_cls; a.det_n = t->d_n; a.det_stride = t->Nc;
    return a;
}

static int check_sticky(rtmodt_tracker *t, int s, int64_t err) {
    RT_CHECK(err != 1, RTMODT_E_CAPACITY, "stream %d: more than max_tracks=%d live tracks", s, t->Mc);
    RT_CHECK(err != 2, RTMODT_E_CAPACITY, "stream %d: lapjv assignment too dense (more than 256 contested rows/columns or 2048 contested pairs)", s);
    RT_CHECK(err == 0, RTMODT_E_INVALID, "stream %d: tracker error %lld", s, (long long)err);
    return RTMODT_OK;
}

// reads back meta rows [s0, s0+cnt) after the launch; raises the sticky capacity error
static int finish(rtmodt_tracker *t, int s0, int cnt, int32_t *n_active_out) {
    RT_HIP(hipMemcpyAsync(t->h_meta + 8 * s0, t->d_meta + 8 * s0, sizeof(int64_t) * 8 * cnt, hipMemcpyDeviceToHost, t->stream));
    RT_HIP(hipStreamSynchronize(t->stream));
    for (int s = s0; s < s0 + cnt; ++s) {
        if (n_active_out) n_active_out[s - s0] = (int32_t)t->h_meta[8 * s + 3];
        RT_TRY(check_sticky(t, s, t->h_meta[8 * s + 2]));
    }
    return RTMODT_OK;
}

int rtmodt_tracker_update(rtmodt_tracker *t, int stream, const float *xyxy, const float *conf, const int32_t *cls, int n,
                          int32_t *n_active_out) {
    RT_CHECK(t && stream >= 0 && stream < t->S && n >= 0, RTMODT_E_INVALID, "bad argument");
    RT_CHECK(n == 0 || (xyxy && conf && cls), RTMODT_E_INVALID, "null detections");
    RT_CHECK(n <= t->Nc, RTMODT_E_CAPACITY, "%d detections > max_dets %d", n, t->Nc);
    RT_HIP(hipSetDevice(t->device));
    RT_TRY(join_foreign(t));
    t->h_n[stream] = n;
    if (n) {
        RT_HIP(hipMemcpyAsync(t->d_box + (size_t)stream * t->Nc, xyxy, (size_t)n * 16, hipMemcpyHostToDevice, t->stream));
        RT_HIP(hipMemcpyAsync(t->d_conf + (size_t)stream * t->Nc, conf, (size_t)n * 4, hipMemcpyHostToDevice, t->stream));
        RT_HIP(hipMemcpyAsync(t->d_cls + (size_t)stream * t->Nc, cls, (size_t)n * 4, hipMemcpyHostToDevice, t->stream));
    }
    RT_HIP(hipMemcpyAsync(t->d_n + stream, t->h_n + stream, 4, hipMemcpyHostToDevice, t->stream));
    TrackerArgs a = make_args(t);
    a.n_streams = 1; a.stream_base = stream;
    RT_TRY(launch_tracker_update(a, t->stream));
    return finish(t, stream, 1, n_active_out);
}

int rtmodt_tracker_update_batch(rtmodt_tracker *t, const float *xyxy, const float *conf, const int32_t *cls, const int32_t *n,
                                int32_t *n_active_out) {
    RT_CHECK(t && n, RTMODT_E_INVALID, "null argument");
    RT_HIP(hipSetDevice(t->device));
    RT_TRY(join_foreign(t));
    for (int s = 0; s < t->S; ++s) {
        RT_CHECK(n[s] >= 0 && n[s] <= t->Nc, RTMODT_E_CAPACITY, "stream %d: %d detections > max_dets %d", s, n[s], t->Nc);
        t->h_n[s] = n[s];
    }
    RT_HIP(hipMemcpyAsync(t->d_box, xyxy, (size_t)t->S * t->Nc * 16, hipMemcpyHostToDevice, t->stream));
    RT_HIP(hipMemcpyAsync(t->d_conf, conf, (size_t)t->S * t->Nc * 4, hipMemcpyHostToDevice, t->stream));
    RT_HIP(hipMemcpyAsync(t->d_cls, cls, (size_t)t->S * t->Nc * 4, hipMemcpyHostToDevice, t->stream));
    RT_HIP(hipMemcpyAsync(t->d_n, t->h_n, (size_t)t->S * 4, hipMemcpyHostToDevice, t->stream));
    RT_TRY(launch_tracker_update(make_args(t), t->stream));
    return finish(t, 0, t->S, n_active_out);
}

static int update_from_detector_slice(rtmodt_tracker *t, rtmodt_detector *det, int first, int count, int frames = 1) {
    RT_CHECK(t && det, RTMODT_E_INVALID, "null argument");
    DetOutputs o;
    RT_TRY(detector_outputs(det, &o));
    if (count < 0) count = o.count - first;
    RT_CHECK(o.device == t->device, RTMODT_E_INVALID, "tracker on device %d, detector on device %d", t->device, o.device);
    RT_CHECK(frames >= 1, RTMODT_E_INVALID, "n_frames %d", frames);
    RT_CHECK(first >= 0 && count >= 1 && (long)first + (long)count * frames <= o.count, RTMODT_E_INVALID, "frames [%d, %ld) outside the detector's batch of %d", first,
             (long)first + (long)count * frames, o.count);
    RT_CHECK(count <= t->S, RTMODT_E_INVALID, "%d frames > tracker streams %d", count, t->S);
    RT_CHECK(o.stride <= t->Nc, RTMODT_E_CAPACITY, "detector max_det %d > tracker max_dets %d", o.stride, t->Nc);
    RT_HIP(hipSetDevice(t->device));
    TrackerArgs a = make_args(t);
    a.n_streams = count;
    a.det_box = o.box + (size_t)first * o.stride; a.det_conf = o.conf + (size_t)first * o.stride; a.det_cls = o.cls + (size_t)first * o.stride;
    a.det_n = o.n + first; a.det_stride = o.stride;
    a.n_frames = frames; a.frame_step = count;
    // (host-fed updates are synchronous -- finish() waits for them -- so the detector's stream needs no event from ours)
    RT_TRY(launch_tracker_update(a, o.stream));          // same HIP stream as the detector's NMS: ordered, no host sync
    RT_HIP(hipEventRecord(t->foreign_done, o.stream));
    t->foreign_pending = true;
    return RTMODT_OK;
}

int rtmodt_tracker_update_from_detector(rtmodt_tracker *t, rtmodt_detector *det) { return update_from_detector_slice(t, det, 0, -1); }

int rtmodt_tracker_update_from_detector_frames(rtmodt_tracker *t, rtmodt_detector *det, int first_frame, int n_frames) {
    RT_CHECK(n_frames >= 1, RTMODT_E_INVALID, "n_frames %d", n_frames);
    return update_from_detector_slice(t, det, first_frame, n_frames);
}

int rtmodt_tracker_update_from_detector_batch(rtmodt_tracker *t, rtmodt_detector *det, int first_frame, int n_streams, int n_frames) {
    RT_CHECK(n_streams >= 1 && n_frames >= 1, RTMODT_E_INVALID, "n_streams %d, n_frames %d", n_streams, n_frames);
    return update_from_detector_slice(t, det, first_frame, n_streams, n_frames);
}

int rtmodt_tracker_state(rtmodt_tracker *t, int stream, int64_t *ids, float *xyxy, float *conf, int32_t *cls, int32_t *age,
                         int32_t *tsu, int32_t *n, int64_t *next_id) {
    RT_CHECK(t && stream >= 0 && stream < t->S, RTMODT_E_INVALID, "bad argument");
    RT_HIP(hipSetDevice(t->device));
    // ordered behind the most recent update (own stream, or -- through its event -- a detector's): meta first, then the arrays
    // of the buffer it names, all through pinned memory -- two stream syncs instead of a device sync and seven blocking copies
    RT_TRY(join_foreign(t));
    hipStream_t q = t->stream;
    int64_t *m = t->h_meta + 8 * stream;
    RT_HIP(hipMemcpyAsync(m, t->d_meta + 8 * stream, sizeof(int64_t) * 8, hipMemcpyDeviceToHost, q));
    RT_HIP(hipStreamSynchronize(q));
    RT_TRY(check_sticky(t, stream, m[2]));
    const int cur = (int)m[0], cnt = (int)m[1];
    const TrackerState &st = t->h_states[stream];
    if (n) *n = cnt;
    if (next_id) *next_id = m[4];
    if (cnt) {
        char *h = t->h_state;
        const size_t M = (size_t)t->Mc;
        char *h_ids = h, *h_box = h + M * 8, *h_conf = h + M * 24, *h_cls = h + M * 28, *h_age = h + M * 32, *h_tsu = h + M * 36;
        if (ids) RT_HIP(hipMemcpyAsync(h_ids, st.ids[cur], (size_t)cnt * 8, hipMemcpyDeviceToHost, q));
        if (xyxy) RT_HIP(hipMemcpyAsync(h_box, st.box[cur], (size_t)cnt * 16, hipMemcpyDeviceToHost, q));
        if (conf) RT_HIP(hipMemcpyAsync(h_conf, st.conf[cur], (size_t)cnt * 4, hipMemcpyDeviceToHost, q));
        if (cls) RT_HIP(hipMemcpyAsync(h_cls, st.cls[cur], (size_t)cnt * 4, hipMemcpyDeviceToHost, q));
        if (age) RT_HIP(hipMemcpyAsync(h_age, st.age[cur], (size_t)cnt * 4, hipMemcpyDeviceToHost, q));
        if (tsu) RT_HIP(hipMemcpyAsync(h_tsu, st.tsu[cur], (size_t)cnt * 4, hipMemcpyDeviceToHost, q));
        RT_HIP(hipStreamSynchronize(q));
        if (ids) memcpy(ids, h_ids, (size_t)cnt * 8);
        if (xyxy) memcpy(xyxy, h_box, (size_t)cnt * 16);
        if (conf) memcpy(conf, h_conf, (size_t)cnt * 4);
        if (cls) memcpy(cls, h_cls, (size_t)cnt * 4);
        if (age) memcpy(age, h_age, (size_t)cnt * 4);
        if (tsu) memcpy(tsu, h_tsu, (size_t)cnt * 4);
    }
    return RTMODT_OK;
}

int rtmodt_tracker_enable_kalman(rtmodt_tracker *t) {
    RT_CHECK(t, RTMODT_E_INVALID, "null argument");
    if (t->kalman) return RTMODT_OK;
    RT_HIP(hipSetDevice(t->device));
    RT_TRY(join_foreign(t));
    RT_HIP(hipMemcpyAsync(t->h_meta, t->d_meta, sizeof(int64_t) * 8 * t->S, hipMemcpyDeviceToHost, t->stream));
    RT_HIP(hipStreamSynchronize(t->stream));
    for (int s = 0; s < t->S; ++s)
        RT_CHECK(t->h_meta[8 * s + 1] == 0, RTMODT_E_INVALID, "stream %d already holds tracks: enable the Kalman model before the first update (or after reset)", s);
    const size_t per_buf = (size_t)t->Mc * 5 * sizeof(float4);
    RT_HIP(hipMalloc((void **)&t->kf_pool, per_buf * 2 * t->S));
    RT_HIP(hipMemset(t->kf_pool, 0, per_buf * 2 * t->S));
    for (int s = 0; s < t->S; ++s)
        for (int b = 0; b < 2; ++b) t->h_states[s].kf[b] = (float4 *)(t->kf_pool + per_buf * (2 * s + b));
    RT_HIP(hipMemcpy(t->d_states, t->h_states.data(), sizeof(TrackerState) * t->S, hipMemcpyHostToDevice));
    t->kalman = true;
    return RTMODT_OK;
}

int rtmodt_tracker_kalman_state(rtmodt_tracker *t, int stream, float *mean, float *cov, int32_t *n) {
    RT_CHECK(t && stream >= 0 && stream < t->S && n, RTMODT_E_INVALID, "bad argument");
    RT_CHECK(t->kalman, RTMODT_E_INVALID, "the Kalman model is not enabled on this tracker");
    RT_HIP(hipSetDevice(t->device));
    RT_TRY(join_foreign(t));
    int64_t *m = t->h_meta + 8 * stream;
    RT_HIP(hipMemcpyAsync(m, t->d_meta + 8 * stream, sizeof(int64_t) * 8, hipMemcpyDeviceToHost, t->stream));
    RT_HIP(hipStreamSynchronize(t->stream));
    RT_TRY(check_sticky(t, stream, m[2]));
    const int cur = (int)m[0], cnt = (int)m[1];
    *n = cnt;
    if (!cnt) return RTMODT_OK;
    std::vector<float4> buf((size_t)5 * t->Mc);
    RT_HIP(hipMemcpy(buf.data(), t->h_states[stream].kf[cur], buf.size() * sizeof(float4), hipMemcpyDeviceToHost));
    for (int i = 0; i < cnt; ++i) {
        const float4 pos = buf[i], vel = buf[t->Mc + i], pa = buf[2 * (size_t)t->Mc + i], pb = buf[3 * (size_t)t->Mc + i], pc = buf[4 * (size_t)t->Mc + i];
        if (mean) { float *o = mean + 8 * (size_t)i; o[0] = pos.x; o[1] = pos.y; o[2] = pos.z; o[3] = pos.w; o[4] = vel.x; o[5] = vel.y; o[6] = vel.z; o[7] = vel.w; }
        if (cov) {                                          // per coordinate (a, b, c) of [[a, b], [b, c]]
            float *o = cov + 12 * (size_t)i;
            o[0] = pa.x; o[1] = pb.x; o[2] = pc.x; o[3] = pa.y; o[4] = pb.y; o[5] = pc.y;
            o[6] = pa.z; o[7] = pb.z; o[8] = pc.z; o[9] = pa.w; o[10] = pb.w; o[11] = pc.w;
        }
    }
    return RTMODT_OK;
}

int rtmodt_tracker_reset(rtmodt_tracker *t, int stream) {
    RT_CHECK(t && stream < t->S, RTMODT_E_INVALID, "bad argument");
    RT_HIP(hipSetDevice(t->device));
    RT_HIP(hipDeviceSynchronize());
    int s0 = stream < 0 ? 0 : stream, s1 = stream < 0 ? t->S : stream + 1;
    for (int s = s0; s < s1; ++s) RT_HIP(hipMemcpy(t->d_meta + 8 * s, init_meta_row, sizeof(init_meta_row), hipMemcpyHostToDevice));
    return RTMODT_OK;
}

int rtmodt_iou_matrix(int device, const float *a, int m, const float *b, int n, float *out) {
    RT_CHECK(m >= 0 && n >= 0 && (m * (long)n == 0 || (a && b && out)), RTMODT_E_INVALID, "bad argument");
    if ((long)m * n == 0) return RTMODT_OK;
    RT_HIP(hipSetDevice(device));
    float4 *da = nullptr, *db = nullptr; float *dout = nullptr;
    int rc = RTMODT_OK;
    auto body = [&]() -> int {
        RT_HIP(hipMalloc((void **)&da, (size_t)m * 16)); RT_HIP(hipMalloc((void **)&db, (size_t)n * 16));
        RT_HIP(hipMalloc((void **)&dout, (size_t)m * n * 4));
        RT_HIP(hipMemcpy(da, a, (size_t)m * 16, hipMemcpyHostToDevice));
        RT_HIP(hipMemcpy(db, b, (size_t)n * 16, hipMemcpyHostToDevice));
        RT_TRY(launch_iou_matrix(da, m, db, n, dout, nullptr));
        RT_HIP(hipDeviceSynchronize());
        RT_HIP(hipMemcpy(out, dout, (size_t)m * n * 4, hipMemcpyDeviceToHost));
        return RTMODT_OK;
    };
    rc = body();
    hipFree(da); hipFree(db); hipFree(dout);
    return rc;
}

int rtmodt_assign_greedy(int device, const float *iou, int m, int n, float thresh, int32_t *row_to_col, int32_t *col_used) {
    RT_CHECK(m >= 1 && n >= 1 && iou && row_to_col && col_used, RTMODT_E_INVALID, "bad argument");
    RT_HIP(hipSetDevice(device));
    float *di = nullptr; int32_t *dr = nullptr, *dc = nullptr;
    auto body = [&]() -> int {
        RT_HIP(hipMalloc((void **)&di, (size_t)m * n * 4)); RT_HIP(hipMalloc((void **)&dr, (size_t)m * 4));
        RT_HIP(hipMalloc((void **)&dc, (size_t)n * 4));
        RT_HIP(hipMemcpy(di, iou, (size_t)m * n * 4, hipMemcpyHostToDevice));
        RT_TRY(launch_assign_greedy(di, m, n, thresh, dr, dc, nullptr));
        RT_HIP(hipDeviceSynchronize());
        RT_HIP(hipMemcpy(row_to_col, dr, (size_t)m * 4, hipMemcpyDeviceToHost));
        RT_HIP(hipMemcpy(col_used, dc, (size_t)n * 4, hipMemcpyDeviceToHost));
        return RTMODT_OK;
    };
    int rc = body();
    hipFree(di); hipFree(dr); hipFree(dc);
    return rc;
}

int rtmodt_assign_lapjv(int device, const float *iou, int m, int n, double cost_limit, int32_t *row_to_col, int32_t *col_used) {
    RT_CHECK(m >= 1 && n >= 1 && m <= 4096 && n <= 4096 && iou && row_to_col && col_used && cost_limit == cost_limit, RTMODT_E_INVALID, "bad argument");
    RT_HIP(hipSetDevice(device));
    float *di = nullptr; int32_t *dr = nullptr, *dc = nullptr, *de = nullptr;
    int32_t err = 0;
    auto body = [&]() -> int {
        RT_HIP(hipMalloc((void **)&di, (size_t)m * n * 4)); RT_HIP(hipMalloc((void **)&dr, (size_t)m * 4));
        RT_HIP(hipMalloc((void **)&dc, (size_t)n * 4)); RT_HIP(hipMalloc((void **)&de, 4));
        RT_HIP(hipMemcpy(di, iou, (size_t)m * n * 4, hipMemcpyHostToDevice));
        RT_TRY(launch_assign_lapjv(di, m, n, cost_limit, dr, dc, de, nullptr));
        RT_HIP(hipDeviceSynchronize());
        RT_HIP(hipMemcpy(row_to_col, dr, (size_t)m * 4, hipMemcpyDeviceToHost));
        RT_HIP(hipMemcpy(col_used, dc, (size_t)n * 4, hipMemcpyDeviceToHost));
        RT_HIP(hipMemcpy(&err, de, 4, hipMemcpyDeviceToHost));
        RT_CHECK(err == 0, RTMODT_E_CAPACITY, "assign_lapjv: too dense (more than 256 contested rows/columns or 2048 contested pairs)");
        return RTMODT_OK;
    };
    int rc = body();
    hipFree(di); hipFree(dr); hipFree(dc); hipFree(de);
    return rc;
}

}  // extern "C"
