/* Plain-C restatement of the reference tracker -- TEST INFRASTRUCTURE ONLY.
 *
 * Follows /root/reference/src/tracking/tracker.py:
 *   tro_update      <- _ByteTrackCore.update            (tracker.py:58-141)
 *   iou_f32         <- _ByteTrackCore._batch_iou        (tracker.py:150-161)
 *   assign_greedy   <- _linear_assignment, greedy branch (tracker.py:182-194)
 *   ageing          <- _ByteTrackCore._age_tracks       (tracker.py:144-148)
 *
 * It is checked bit-for-bit against the NumPy oracle (itself pinned to the
 * reference by tests/golden/tracker_*.npz) in tests/test_oracle_tracker_c.py, and
 * is the single-thread "port" timed as bench.py's tracker cpu_baseline.
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  Every float
 * operation is a separately rounded IEEE binary32 operation, as NumPy does it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float track_thresh, match_thresh;
    int track_buffer;
    int cap, n;
    int64_t next_id;
    int64_t *ids;
    float *xyxy, *conf;
    int32_t *cls, *age, *tsu;
    /* scratch */
    int *hi_idx, *lo_idx, *um_t, *row_best, *col_used, *row_matched;
    int scratch_cap;
} tro_t;

static float iou_f32(const float *a, const float *b) {
    float x1 = a[0] > b[0] ? a[0] : b[0];
    float y1 = a[1] > b[1] ? a[1] : b[1];
    float x2 = a[2] < b[2] ? a[2] : b[2];
    float y2 = a[3] < b[3] ? a[3] : b[3];
    float w = x2 - x1; if (!(w > 0.0f)) w = 0.0f;
    float h = y2 - y1; if (!(h > 0.0f)) h = 0.0f;
    float inter = w * h;
    float area_a = (a[2] - a[0]) * (a[3] - a[1]);
    float area_b = (b[2] - b[0]) * (b[3] - b[1]);
    float uni = (area_a + area_b) - inter;
    return inter / (uni + 1e-6f);
}

tro_t *tro_create(float track_thresh, int track_buffer, float match_thresh, int cap) {
    tro_t *t = (tro_t *)calloc(1, sizeof(tro_t));
    t->track_thresh = track_thresh; t->match_thresh = match_thresh; t->track_buffer = track_buffer;
    t->cap = cap; t->next_id = 1;
    t->ids = (int64_t *)malloc(sizeof(int64_t) * cap);
    t->xyxy = (float *)malloc(sizeof(float) * 4 * cap);
    t->conf = (float *)malloc(sizeof(float) * cap);
    t->cls = (int32_t *)malloc(sizeof(int32_t) * cap);
    t->age = (int32_t *)malloc(sizeof(int32_t) * cap);
    t->tsu = (int32_t *)malloc(sizeof(int32_t) * cap);
    t->scratch_cap = cap;
    t->hi_idx = (int *)malloc(sizeof(int) * cap); t->lo_idx = (int *)malloc(sizeof(int) * cap);
    t->um_t = (int *)malloc(sizeof(int) * cap); t->row_best = (int *)malloc(sizeof(int) * cap);
    t->col_used = (int *)malloc(sizeof(int) * cap); t->row_matched = (int *)malloc(sizeof(int) * cap);
    return t;
}

void tro_destroy(tro_t *t) {
    if (!t) return;
    free(t->ids); free(t->xyxy); free(t->conf); free(t->cls); free(t->age); free(t->tsu);
    free(t->hi_idx); free(t->lo_idx); free(t->um_t); free(t->row_best); free(t->col_used); free(t->row_matched);
    free(t);
}

/* One association pass: rows = tracks listed in rows[0..m), cols = detections
 * listed in cols[0..n) (indices into the frame's arrays).  Greedy in row order,
 * first arg-max, >= threshold in float32, no retry.  Writes row_matched[r] = the
 * matched col position or -1, col_used[c] = 1 if taken. */
static void assign_and_commit(tro_t *t, const int *rows, int m, const int *cols, int n,
                              const float *xyxy, const float *conf, const int32_t *cls) {
    for (int c = 0; c < n; ++c) t->col_used[c] = 0;
    for (int r = 0; r < m; ++r) {
        const float *tb = t->xyxy + 4 * rows[r];
        int best = 0; float bv = iou_f32(tb, xyxy + 4 * cols[0]);
        for (int c = 1; c < n; ++c) {
            float v = iou_f32(tb, xyxy + 4 * cols[c]);
            if (v > bv) { bv = v; best = c; }              /* first maximum wins */
        }
        t->row_matched[r] = -1;
        if (bv >= t->match_thresh && !t->col_used[best]) { t->col_used[best] = 1; t->row_matched[r] = best; }
    }
    /* commit AFTER the whole matrix is evaluated: the reference computes the IoU
     * matrix from the pre-update boxes (tracker.py:92-104) */
    for (int r = 0; r < m; ++r) {
        int c = t->row_matched[r];
        if (c < 0) continue;
        int ti = rows[r], di = cols[c];
        memcpy(t->xyxy + 4 * ti, xyxy + 4 * di, 4 * sizeof(float));
        t->conf[ti] = conf[di]; t->cls[ti] = cls[di]; t->age[ti] += 1; t->tsu[ti] = 0;
    }
}

/* returns number of tracks with tsu == 0 after the update (always 0), or -1 on overflow */
int tro_update(tro_t *t, const float *xyxy, const float *conf, const int32_t *cls, int n) {
    if (n == 0) {                                              /* tracker.py:70-73 */
        for (int i = 0; i < t->n; ++i) t->tsu[i] += 1;
        return 0;
    }
    if (n > t->scratch_cap) return -1;
    int nh = 0, nl = 0;
    for (int i = 0; i < n; ++i) {
        if (conf[i] >= t->track_thresh) t->hi_idx[nh++] = i; else t->lo_idx[nl++] = i;
    }
    int m = t->n, num = 0;
    int *rows_all = t->row_best;                                /* reuse as identity list */
    if (m > 0 && nh > 0) {
        for (int i = 0; i < m; ++i) rows_all[i] = i;
        assign_and_commit(t, rows_all, m, t->hi_idx, nh, xyxy, conf, cls);
        for (int i = 0; i < m; ++i) if (t->row_matched[i] < 0) t->um_t[num++] = i;
    } else {
        for (int i = 0; i < m; ++i) t->um_t[num++] = i;
        for (int c = 0; c < nh; ++c) t->col_used[c] = 0;
    }
    /* spawn list must be taken from pass 1's unmatched columns BEFORE pass 2 reuses col_used */
    int nspawn = 0;
    int *spawn = (int *)malloc(sizeof(int) * (nh > 0 ? nh : 1));
    for (int c = 0; c < nh; ++c) if (!t->col_used[c]) spawn[nspawn++] = t->hi_idx[c];
    if (num > 0 && nl > 0) {
        int *rows2 = (int *)malloc(sizeof(int) * num);
        memcpy(rows2, t->um_t, sizeof(int) * num);
        assign_and_commit(t, rows2, num, t->lo_idx, nl, xyxy, conf, cls);
        free(rows2);
    }
    if (t->n + nspawn > t->cap) { free(spawn); return -1; }
    for (int k = 0; k < nspawn; ++k) {                          /* tracker.py:126-135 */
        int di = spawn[k], ti = t->n++;
        t->ids[ti] = t->next_id++;
        memcpy(t->xyxy + 4 * ti, xyxy + 4 * di, 4 * sizeof(float));
        t->conf[ti] = conf[di]; t->cls[ti] = cls[di]; t->age[ti] = 1; t->tsu[ti] = 0;
    }
    free(spawn);
    int w = 0, active = 0;
    for (int i = 0; i < t->n; ++i) {                            /* tracker.py:138-139 */
        int tsu = t->tsu[i] + 1;
        if (tsu > t->track_buffer) continue;
        if (w != i) {
            t->ids[w] = t->ids[i]; memcpy(t->xyxy + 4 * w, t->xyxy + 4 * i, 4 * sizeof(float));
            t->conf[w] = t->conf[i]; t->cls[w] = t->cls[i]; t->age[w] = t->age[i];
        }
        t->tsu[w] = tsu;
        if (tsu == 0) ++active;
        ++w;
    }
    t->n = w;
    return active;
}

int tro_count(const tro_t *t) { return t->n; }
int64_t tro_next_id(const tro_t *t) { return t->next_id; }

void tro_state(const tro_t *t, int64_t *ids, float *xyxy, float *conf, int32_t *cls, int32_t *age, int32_t *tsu) {
    memcpy(ids, t->ids, sizeof(int64_t) * t->n);
    memcpy(xyxy, t->xyxy, sizeof(float) * 4 * t->n);
    memcpy(conf, t->conf, sizeof(float) * t->n);
    memcpy(cls, t->cls, sizeof(int32_t) * t->n);
    memcpy(age, t->age, sizeof(int32_t) * t->n);
    memcpy(tsu, t->tsu, sizeof(int32_t) * t->n);
}

/* standalone M x N IoU matrix (fixture G1) */
void tro_batch_iou(const float *a, int m, const float *b, int n, float *out) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) out[(size_t)i * n + j] = iou_f32(a + 4 * i, b + 4 * j);
}
