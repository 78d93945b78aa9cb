// zones.hip -- the reference's ZoneEventEngine.process (src/events/zone_engine.py:82-132) on the
// GPU: ONE launch per frame, one 256-thread workgroup per video stream, run either on track lists the
// host hands over (the reference's call shape) or straight on the tracker's device-resident state.
//
// The reference keeps two dicts keyed by track id: _occupancy {track -> {zone name -> first seen}} and
// the never-purged _cooldown {(track, zone name) -> last alert}.  Here both live in one LEDGER per
// stream: rows sorted by track id, each row = occupancy bit mask + first_seen[Z] + last_alert[Z]
// (indexed by the zone's KEY = the first zone of the same name, because the dicts are keyed by name
// and zones may share one).  A frame is:
//   1. every passed track finds its old row by binary search (ledger ids staged in LDS);
//   2. one lane per track walks the zones IN ORDER (the reference's loop :94, same-name zones
//      interact through the shared key): integer point-in-polygon (cv::pointPolygonTest's integer
//      branch restated, oracle/zone_oracle.py), occupancy / dwell / cooldown in double;
//   3. the new ledger = passed tracks + retained idle rows, merged by rank (prefix sums + binary
//      searches, no sort); rows idle for more than max_idle frames are dropped -- the reference never
//      drops a cooldown entry (a leak); with the device tracker a dead id never returns, so nothing
//      observable is lost;
//   4. events leave in (track order, zone order) through a block prefix sum.
// Times are doubles handed in by the caller (`now` = the reference's time.time() at :84).
#include <algorithm>
#include <cstring>
#include <vector>

#include "kernels.h"

namespace rtmodt {

constexpr int ZN_THREADS = 256, ZN_WAVES = ZN_THREADS / 64;
constexpr int ZN_MAX_ZONES = 32, ZN_MAX_POINTS = 2048;

struct ZoneTable {                 // device pointers
    const int2 *pts;               // all polygons' vertices, concatenated
    const int32_t *off;            // [Z + 1]
    const double *dwell, *cooldown;    // [Z]
    const int32_t *key;            // [Z] index of the first zone with the same name
    int Z, n_pts;
};

struct ZoneLedger {                // one stream, double-buffered (cur picks the buffer)
    int64_t *id[2]; int64_t *seen[2]; uint32_t *mask[2]; double *first[2]; double *alert[2];
};

struct ZoneArgs {
    ZoneTable zt;
    ZoneLedger *ledgers;           // [n_streams]
    int64_t *meta;                 // [n_streams][4]: cur, count, err, 0
    int cap, max_events, stream_base;
    int64_t max_idle;              // < 0: drop every row that is not in this frame's list
    double now; int64_t frame_id;
    // source A: staged lists [n_streams][max_tracks], sorted by id, `order` = the caller's index
    const int64_t *s_ids; const float4 *s_box; const int32_t *s_cls; const int32_t *s_order; const int32_t *s_n; int s_stride;
    // source B: the tracker's device state (s_ids == nullptr); passed tracks are those with tsu == report_tsu
    const TrackerState *t_states; const int64_t *t_meta; int report_tsu;
    // per-stream scratch [n_streams][max_tracks]
    int32_t *oldpos; uint32_t *evmask; double *evdwell;    // evdwell [..][Z]
    // events [n_streams][max_events]
    int64_t *ev_id; int32_t *ev_track; int32_t *ev_zone; double *ev_dwell; float4 *ev_box; int2 *ev_c; int32_t *ev_cls; int32_t *ev_n;
};

// cv::pointPolygonTest(contour int32, integer point, measureDist=false) >= 0  (oracle/zone_oracle.py:point_polygon_test)
__device__ __forceinline__ bool inside_or_on(const int2 *p, int total, int x, int y) {
    if (total == 0) return false;
    int counter = 0;
    int2 v = p[total - 1];
    for (int i = 0; i < total; ++i) {
        const int2 v0 = v;
        v = p[i];
        if ((v0.y <= y && v.y <= y) || (v0.y > y && v.y > y) || (v0.x < x && v.x < x)) {
            if (y == v.y && (x == v.x || (y == v0.y && ((v0.x <= x && x <= v.x) || (v.x <= x && x <= v0.x))))) return true;
            continue;
        }
        long long dist = (long long)(y - v0.y) * (v.x - v0.x) - (long long)(x - v0.x) * (v.y - v0.y);
        if (dist == 0) return true;
        if (v.y < v0.y) dist = -dist;
        counter += dist > 0;
    }
    return (counter & 1) != 0;
}

__device__ __forceinline__ int lower_bound_i64(const int64_t *a, int n, int64_t x) {     // first index with a[i] >= x
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// exclusive prefix of a per-thread count over the workgroup; two barriers
__device__ __forceinline__ int zn_block_scan(int v, int *wsum, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < ZN_WAVES; ++w) {
        const int s = wsum[w];
        if (w < wave) off += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return off + incl - v;
}

#pragma clang fp contract(off)

__global__ __launch_bounds__(ZN_THREADS) void zones_update(ZoneArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int sidx = a.stream_base + blockIdx.x, tid = threadIdx.x;
    const int Z = a.zt.Z, cap = a.cap;
    int64_t *old_id = (int64_t *)smem;                    // [cap]
    int *ret_pre = (int *)(old_id + cap);                 // [cap + 1] exclusive prefix of the retained flags
    int2 *pts = (int2 *)(ret_pre + cap + 1 + ((cap + 1) & 1));
    int *wsum = (int *)(pts + a.zt.n_pts);
    __shared__ int s_err;

    int64_t *meta = a.meta + (size_t)sidx * 4;
    const int cur = (int)meta[0], n_old = (int)meta[1], nxt = cur ^ 1;
    const ZoneLedger L = a.ledgers[sidx];
    const int64_t *o_id = cur ? L.id[1] : L.id[0];      int64_t *n_id = cur ? L.id[0] : L.id[1];
    const int64_t *o_seen = cur ? L.seen[1] : L.seen[0]; int64_t *n_seen = cur ? L.seen[0] : L.seen[1];
    const uint32_t *o_mask = cur ? L.mask[1] : L.mask[0]; uint32_t *n_mask = cur ? L.mask[0] : L.mask[1];
    const double *o_first = cur ? L.first[1] : L.first[0]; double *n_first = cur ? L.first[0] : L.first[1];
    const double *o_alert = cur ? L.alert[1] : L.alert[0]; double *n_alert = cur ? L.alert[0] : L.alert[1];

    // ---- this frame's list ----
    const int64_t *ids; const float4 *box; const int32_t *cls; const int32_t *order = nullptr; const int32_t *tsu = nullptr;
    int n;
    if (a.s_ids) {
        const size_t o = (size_t)sidx * a.s_stride;
        ids = a.s_ids + o; box = a.s_box + o; cls = a.s_cls + o; order = a.s_order + o; n = a.s_n[sidx];
    } else {
        const TrackerState st = a.t_states[sidx];
        const int64_t *tm = a.t_meta + (size_t)sidx * 8;
        const int tc = (int)tm[0];
        ids = tc ? st.ids[1] : st.ids[0]; box = tc ? st.box[1] : st.box[0]; cls = tc ? st.cls[1] : st.cls[0]; tsu = tc ? st.tsu[1] : st.tsu[0];
        n = (int)tm[1];
    }
    if (n > cap) n = cap;                                  // host sizes cap >= max_tracks

    if (tid == 0) s_err = 0;
    for (int i = tid; i < n_old; i += ZN_THREADS) old_id[i] = o_id[i];
    for (int i = tid; i < a.zt.n_pts; i += ZN_THREADS) pts[i] = a.zt.pts[i];
    __syncthreads();

    int32_t *oldpos = a.oldpos + (size_t)sidx * cap;
    uint32_t *evmask = a.evmask + (size_t)sidx * cap;
    double *evdwell = a.evdwell + (size_t)sidx * cap * Z;

    // ---- 1. which old rows are gone, which stay as idle rows ----
    for (int j = tid; j < n_old; j += ZN_THREADS) ret_pre[j] = 1;             // provisional: retained unless matched / expired
    __syncthreads();
    for (int i = tid; i < n; i += ZN_THREADS) {
        const int64_t id = ids[i];
        const int j = lower_bound_i64(old_id, n_old, id);
        const bool hit = j < n_old && old_id[j] == id;
        oldpos[i] = hit ? j : -1;
        if (hit) ret_pre[j] = 0;
    }
    __syncthreads();
    for (int j = tid; j < n_old; j += ZN_THREADS)
        if (ret_pre[j] && (a.max_idle < 0 || a.frame_id - o_seen[j] > a.max_idle)) ret_pre[j] = 0;
    __syncthreads();
    int n_ret = 0;
    for (int base = 0; base < n_old; base += ZN_THREADS) {                    // flags -> exclusive prefix, in place
        const int j = base + tid;
        const int f = j < n_old ? ret_pre[j] : 0;
        int tot;
        const int pos = zn_block_scan(f, wsum, tot);
        if (j < n_old) ret_pre[j] = f ? n_ret + pos : -(n_ret + pos) - 1;       // retained: rank; dropped: -(rank of next retained) - 1
        n_ret += tot;
    }
    if (tid == 0) ret_pre[n_old] = -n_ret - 1;
    __syncthreads();
    auto ranks_below = [&](int j) { const int v = ret_pre[j]; return v >= 0 ? v : -v - 1; };   // retained rows among old[0 .. j)
    const bool overflow = n + n_ret > cap;                                    // keep this frame's rows, drop the idle ones
    if (overflow && tid == 0) s_err = 1;

    // ---- 2. the reference's double loop: tracks (one lane each) x zones (in order) ----
    for (int i = tid; i < n; i += ZN_THREADS) {
        const int j = oldpos[i];
        const bool active = tsu ? tsu[i] == a.report_tsu : true;
        const int np = i + (overflow ? 0 : ranks_below(lower_bound_i64(old_id, n_old, ids[i])));
        uint32_t mask = j >= 0 ? o_mask[j] : 0u;
        uint32_t ev = 0u;
        double *nf = n_first + (size_t)np * Z, *na = n_alert + (size_t)np * Z;
        for (int z = 0; z < Z; ++z) {                                         // carry the old row over
            nf[z] = j >= 0 ? o_first[(size_t)j * Z + z] : 0.0;
            na[z] = j >= 0 ? o_alert[(size_t)j * Z + z] : 0.0;
        }
        if (active) {
            const float4 b = box[i];
            const int cx = (int)((b.x + b.z) / 2.0f), cy = (int)((b.y + b.w) / 2.0f);   // zone_engine.py:91-92
            for (int z = 0; z < Z; ++z) {
                const int k = a.zt.key[z];
                const int p0 = a.zt.off[z];
                if (inside_or_on(pts + p0, a.zt.off[z + 1] - p0, cx, cy)) {                // :95
                    if (!(mask >> k & 1u)) { mask |= 1u << k; nf[k] = a.now; }         // :99-100
                    const double dwell = a.now - nf[k];                                 // :102
                    if (dwell >= a.zt.dwell[z] && a.now - na[k] >= a.zt.cooldown[z]) {  // :104-107
                        ev |= 1u << z;
                        evdwell[(size_t)i * Z + z] = dwell;
                        na[k] = a.now;                                                  // :118
                    }
                } else {
                    mask &= ~(1u << k);                                                 // :121-123
                }
            }
        } else {
            mask = 0u;                                                                  // :126-128 (not passed this frame)
        }
        n_id[np] = ids[i];
        n_seen[np] = active ? a.frame_id : (j >= 0 ? o_seen[j] : a.frame_id);
        n_mask[np] = mask;
        evmask[i] = ev;
    }
    // idle rows move to their merged position
    if (!overflow)
        for (int j = tid; j < n_old; j += ZN_THREADS) {
            const int v = ret_pre[j];
            if (v < 0) continue;
            const int np = v + lower_bound_i64(ids, n, old_id[j]);
            n_id[np] = old_id[j];
            n_seen[np] = o_seen[j];
            n_mask[np] = 0u;                                                            // idle == not passed == purged occupancy
            for (int z = 0; z < Z; ++z) { n_first[(size_t)np * Z + z] = 0.0; n_alert[(size_t)np * Z + z] = o_alert[(size_t)j * Z + z]; }
        }
    __syncthreads();

    // ---- 3. events, in list order then zone order ----
    const size_t eo = (size_t)sidx * a.max_events;
    int n_ev = 0;
    for (int base = 0; base < n; base += ZN_THREADS) {
        const int i = base + tid;
        const uint32_t ev = i < n ? evmask[i] : 0u;
        int tot;
        int pos = n_ev + zn_block_scan(__popc(ev), wsum, tot);
        if (ev) {
            const float4 b = box[i];
            for (int z = 0; z < Z; ++z)
                if (ev >> z & 1u) {
                    if (pos < a.max_events) {
                        a.ev_id[eo + pos] = ids[i];
                        a.ev_track[eo + pos] = order ? order[i] : i;
                        a.ev_zone[eo + pos] = z;
                        a.ev_dwell[eo + pos] = evdwell[(size_t)i * Z + z];
                        a.ev_box[eo + pos] = b;
                        a.ev_c[eo + pos] = make_int2((int)((b.x + b.z) / 2.0f), (int)((b.y + b.w) / 2.0f));
                        a.ev_cls[eo + pos] = cls[i];
                    }
                    ++pos;
                }
        }
        n_ev += tot;
    }
    if (tid == 0) {
        if (n_ev > a.max_events) s_err = s_err ? s_err : 2;
        a.ev_n[sidx] = n_ev < a.max_events ? n_ev : a.max_events;
        meta[0] = nxt;
        meta[1] = n + (overflow ? 0 : n_ret);
        if (s_err) meta[2] = s_err;
    }
}

}  // namespace rtmodt

// ======================================================================================
// C ABI
// ======================================================================================
using namespace rtmodt;

struct rtmodt_zones {
    int device = 0, S = 1, Mc = 0, cap = 0, Z = 0, n_pts = 0, max_events = 0;
    int64_t max_idle = 0;
    hipStream_t stream = nullptr;
    char *pool = nullptr;                 // every device array below lives in this one allocation
    ZoneTable zt{};
    ZoneLedger *d_ledgers = nullptr;
    std::vector<ZoneLedger> h_ledgers;
    int64_t *d_meta = nullptr;
    int64_t *s_ids = nullptr; float4 *s_box = nullptr; int32_t *s_cls = nullptr, *s_order = nullptr, *s_n = nullptr;
    int32_t *oldpos = nullptr; uint32_t *evmask = nullptr; double *evdwell = nullptr;
    int64_t *ev_id = nullptr; int32_t *ev_track = nullptr, *ev_zone = nullptr, *ev_cls = nullptr, *ev_n = nullptr;
    double *ev_dwell = nullptr; float4 *ev_box = nullptr; int2 *ev_c = nullptr;
    char *h_pin = nullptr;                // pinned mirror of the event arrays + meta
    size_t ev_bytes = 0;
};

namespace {

struct Carver {
    char *base; size_t off = 0;
    template <typename T> T *take(size_t count) {
        off = align_up(off, 16);
        T *p = base ? (T *)(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

// lays out every device array; base == nullptr -> size only
size_t carve(rtmodt_zones *z, char *base) {
    Carver c{base};
    const size_t S = z->S, cap = z->cap, Z = z->Z, E = z->max_events;
    int2 *pts = c.take<int2>(std::max(z->n_pts, 1));
    int32_t *off = c.take<int32_t>(Z + 1);
    double *dw = c.take<double>(Z), *cd = c.take<double>(Z);
    int32_t *key = c.take<int32_t>(Z);
    z->zt = ZoneTable{pts, off, dw, cd, key, z->Z, z->n_pts};
    z->d_ledgers = c.take<ZoneLedger>(S);
    z->d_meta = c.take<int64_t>(S * 4);
    if (base) z->h_ledgers.assign(S, ZoneLedger{});
    for (size_t s = 0; s < S; ++s)
        for (int b = 0; b < 2; ++b) {
            int64_t *id = c.take<int64_t>(cap), *seen = c.take<int64_t>(cap);
            uint32_t *mask = c.take<uint32_t>(cap);
            double *first = c.take<double>(cap * Z), *alert = c.take<double>(cap * Z);
            if (base) { ZoneLedger &L = z->h_ledgers[s]; L.id[b] = id; L.seen[b] = seen; L.mask[b] = mask; L.first[b] = first; L.alert[b] = alert; }
        }
    z->s_ids = c.take<int64_t>(S * cap); z->s_box = c.take<float4>(S * cap); z->s_cls = c.take<int32_t>(S * cap);
    z->s_order = c.take<int32_t>(S * cap); z->s_n = c.take<int32_t>(S);
    z->oldpos = c.take<int32_t>(S * cap); z->evmask = c.take<uint32_t>(S * cap); z->evdwell = c.take<double>(S * cap * Z);
    const size_t ev0 = align_up(c.off, 16);
    z->ev_id = c.take<int64_t>(S * E); z->ev_dwell = c.take<double>(S * E); z->ev_box = c.take<float4>(S * E); z->ev_c = c.take<int2>(S * E);
    z->ev_track = c.take<int32_t>(S * E); z->ev_zone = c.take<int32_t>(S * E); z->ev_cls = c.take<int32_t>(S * E); z->ev_n = c.take<int32_t>(S);
    z->ev_bytes = align_up(c.off, 16) - ev0;
    return align_up(c.off, 16);
}

ZoneArgs make_args(rtmodt_zones *z, double now, int64_t frame_id) {
    ZoneArgs a{};
    a.zt = z->zt; a.ledgers = z->d_ledgers; a.meta = z->d_meta; a.cap = z->cap; a.max_events = z->max_events; a.stream_base = 0;
    a.max_idle = z->max_idle; a.now = now; a.frame_id = frame_id;
    a.oldpos = z->oldpos; a.evmask = z->evmask; a.evdwell = z->evdwell;
    a.ev_id = z->ev_id; a.ev_track = z->ev_track; a.ev_zone = z->ev_zone; a.ev_dwell = z->ev_dwell; a.ev_box = z->ev_box; a.ev_c = z->ev_c;
    a.ev_cls = z->ev_cls; a.ev_n = z->ev_n;
    return a;
}

int launch(rtmodt_zones *z, const ZoneArgs &a, int n_streams, hipStream_t s) {
    size_t smem = (size_t)z->cap * 8 + ((size_t)z->cap + 2) * 4 + (size_t)std::max(z->n_pts, 1) * 8 + (ZN_WAVES + 1) * 4 + 32;
    RT_CHECK(smem <= 150 * 1024, RTMODT_E_INVALID, "zones: capacity %d needs %zu B of LDS", z->cap, smem);
    static DynLdsSeen seen;
    RT_TRY(raise_dynamic_lds((const void *)zones_update, smem, seen));
    hipLaunchKernelGGL(zones_update, dim3(n_streams), dim3(ZN_THREADS), smem, s, a);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

// event arrays of streams [s0, s0 + cnt) -> pinned host mirror (same layout as the device block), then sync
struct EvHost { const int64_t *id; const double *dwell; const float4 *box; const int2 *c; const int32_t *track, *zone, *cls, *n; const int64_t *meta; };
int fetch_events(rtmodt_zones *z, hipStream_t s, EvHost &h) {
    char *d0 = (char *)z->ev_id;
    RT_HIP(hipMemcpyAsync(z->h_pin, d0, z->ev_bytes, hipMemcpyDeviceToHost, s));
    RT_HIP(hipMemcpyAsync(z->h_pin + z->ev_bytes, z->d_meta, sizeof(int64_t) * 4 * z->S, hipMemcpyDeviceToHost, s));
    RT_HIP(hipStreamSynchronize(s));
    auto at = [&](const void *p) { return z->h_pin + ((const char *)p - d0); };
    h.id = (const int64_t *)at(z->ev_id); h.dwell = (const double *)at(z->ev_dwell); h.box = (const float4 *)at(z->ev_box);
    h.c = (const int2 *)at(z->ev_c); h.track = (const int32_t *)at(z->ev_track); h.zone = (const int32_t *)at(z->ev_zone);
    h.cls = (const int32_t *)at(z->ev_cls); h.n = (const int32_t *)at(z->ev_n);
    h.meta = (const int64_t *)(z->h_pin + z->ev_bytes);
    return RTMODT_OK;
}

int check_err(rtmodt_zones *z, int s, int64_t err) {
    RT_CHECK(err != 1, RTMODT_E_CAPACITY, "zones stream %d: ledger full (%d rows): lower max_idle_frames or raise max_tracks", s, z->cap);
    RT_CHECK(err != 2, RTMODT_E_CAPACITY, "zones stream %d: more than max_events=%d events in one frame", s, z->max_events);
    RT_CHECK(err == 0, RTMODT_E_INVALID, "zones stream %d: error %lld", s, (long long)err);
    return RTMODT_OK;
}

}  // namespace

extern "C" {

void rtmodt_zones_destroy(rtmodt_zones *z) {
    if (!z) return;
    hipSetDevice(z->device);
    if (z->stream) hipStreamSynchronize(z->stream);
    hipFree(z->pool);
    hipHostFree(z->h_pin);
    if (z->stream) hipStreamDestroy(z->stream);
    delete z;
}

int rtmodt_zones_create(int device, const rtmodt_zone_cfg *zones, int n_zones, int n_streams, int max_tracks, int max_events,
                        int64_t max_idle_frames, rtmodt_zones **out) {
    RT_CHECK(out && (zones || n_zones == 0), RTMODT_E_INVALID, "null argument");
    RT_CHECK(n_zones >= 0 && n_zones <= ZN_MAX_ZONES, RTMODT_E_INVALID, "%d zones (at most %d)", n_zones, ZN_MAX_ZONES);
    RT_CHECK(n_streams >= 1 && n_streams <= 4096 && max_tracks >= 1 && max_tracks <= 4096 && max_events >= 1 && max_events <= (1 << 20),
             RTMODT_E_INVALID, "n_streams %d / max_tracks %d / max_events %d out of range", n_streams, max_tracks, max_events);
    std::vector<int2> pts;
    std::vector<int32_t> off(1, 0), key;
    std::vector<double> dw, cd;
    for (int i = 0; i < n_zones; ++i) {
        const rtmodt_zone_cfg &c = zones[i];
        RT_CHECK(c.n_points >= 0 && (c.n_points == 0 || c.polygon_xy), RTMODT_E_INVALID, "zone %d: bad polygon", i);
        RT_CHECK(c.key >= 0 && c.key <= i && zones[c.key].key == c.key, RTMODT_E_INVALID, "zone %d: key %d must name the first zone of the same name", i, c.key);
        RT_CHECK(c.dwell_time_sec == c.dwell_time_sec && c.cooldown_sec == c.cooldown_sec, RTMODT_E_INVALID, "zone %d: NaN time", i);
        for (int p = 0; p < c.n_points; ++p) pts.push_back(make_int2(c.polygon_xy[2 * p], c.polygon_xy[2 * p + 1]));
        off.push_back((int32_t)pts.size());
        key.push_back(c.key); dw.push_back(c.dwell_time_sec); cd.push_back(c.cooldown_sec);
    }
    RT_CHECK((int)pts.size() <= ZN_MAX_POINTS, RTMODT_E_INVALID, "%zu polygon points (at most %d)", pts.size(), ZN_MAX_POINTS);
    rtmodt_zones *z = new rtmodt_zones();
    z->device = device; z->S = n_streams; z->Mc = max_tracks; z->cap = 2 * max_tracks; z->Z = n_zones; z->n_pts = (int)pts.size();
    z->max_events = max_events; z->max_idle = max_idle_frames;
    auto body = [&]() -> int {
        RT_HIP(hipSetDevice(device));
        RT_HIP(hipStreamCreateWithFlags(&z->stream, hipStreamNonBlocking));
        const size_t total = carve(z, nullptr);
        RT_HIP(hipMalloc((void **)&z->pool, total));
        RT_HIP(hipMemset(z->pool, 0, total));
        carve(z, z->pool);
        RT_HIP(hipHostMalloc((void **)&z->h_pin, z->ev_bytes + sizeof(int64_t) * 4 * z->S, hipHostMallocDefault));
        if (!pts.empty()) RT_HIP(hipMemcpy((void *)z->zt.pts, pts.data(), pts.size() * sizeof(int2), hipMemcpyHostToDevice));
        RT_HIP(hipMemcpy((void *)z->zt.off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
        if (n_zones) {
            RT_HIP(hipMemcpy((void *)z->zt.dwell, dw.data(), dw.size() * 8, hipMemcpyHostToDevice));
            RT_HIP(hipMemcpy((void *)z->zt.cooldown, cd.data(), cd.size() * 8, hipMemcpyHostToDevice));
            RT_HIP(hipMemcpy((void *)z->zt.key, key.data(), key.size() * 4, hipMemcpyHostToDevice));
        }
        RT_HIP(hipMemcpy(z->d_ledgers, z->h_ledgers.data(), sizeof(ZoneLedger) * z->S, hipMemcpyHostToDevice));
        return RTMODT_OK;
    };
    int rc = body();
    if (rc != RTMODT_OK) {
        std::string keep = last_error();
        rtmodt_zones_destroy(z);
        last_error() = keep;
        return rc;
    }
    *out = z;
    return RTMODT_OK;
}

int rtmodt_zones_process(rtmodt_zones *z, int stream, const int64_t *track_ids, const float *xyxy, const int32_t *cls, int n, double now,
                         int64_t frame_id, int32_t *ev_track, int32_t *ev_zone, double *ev_dwell, int32_t *ev_centroid, int32_t *n_events) {
    RT_CHECK(z && stream >= 0 && stream < z->S && n >= 0 && n_events, RTMODT_E_INVALID, "bad argument");
    RT_CHECK(n == 0 || (track_ids && xyxy && cls), RTMODT_E_INVALID, "null tracks");
    RT_CHECK(n <= z->Mc, RTMODT_E_CAPACITY, "%d tracks > max_tracks %d", n, z->Mc);
    RT_CHECK(now == now, RTMODT_E_INVALID, "now is NaN");
    RT_HIP(hipSetDevice(z->device));
    // the ledger is sorted by id: hand the list over in id order, remember the caller's order
    std::vector<int32_t> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return track_ids[a] < track_ids[b]; });
    std::vector<int64_t> ids(n); std::vector<float4> box(n); std::vector<int32_t> kc(n);
    for (int i = 0; i < n; ++i) {
        const int p = perm[i];
        ids[i] = track_ids[p]; kc[i] = cls[p];
        box[i] = make_float4(xyxy[4 * p], xyxy[4 * p + 1], xyxy[4 * p + 2], xyxy[4 * p + 3]);
        RT_CHECK(i == 0 || ids[i] != ids[i - 1], RTMODT_E_INVALID, "track id %lld appears twice", (long long)ids[i]);
    }
    const size_t o = (size_t)stream * z->cap;
    if (n) {
        RT_HIP(hipMemcpyAsync(z->s_ids + o, ids.data(), (size_t)n * 8, hipMemcpyHostToDevice, z->stream));
        RT_HIP(hipMemcpyAsync(z->s_box + o, box.data(), (size_t)n * 16, hipMemcpyHostToDevice, z->stream));
        RT_HIP(hipMemcpyAsync(z->s_cls + o, kc.data(), (size_t)n * 4, hipMemcpyHostToDevice, z->stream));
        RT_HIP(hipMemcpyAsync(z->s_order + o, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice, z->stream));
    }
    RT_HIP(hipMemcpyAsync(z->s_n + stream, &n, 4, hipMemcpyHostToDevice, z->stream));
    RT_HIP(hipStreamSynchronize(z->stream));               // the vectors above are pageable and about to go away
    ZoneArgs a = make_args(z, now, frame_id);
    a.stream_base = stream;
    a.s_ids = z->s_ids; a.s_box = z->s_box; a.s_cls = z->s_cls; a.s_order = z->s_order; a.s_n = z->s_n; a.s_stride = z->cap;
    RT_TRY(launch(z, a, 1, z->stream));
    EvHost h;
    RT_TRY(fetch_events(z, z->stream, h));
    RT_TRY(check_err(z, stream, h.meta[4 * stream + 2]));
    const int ne = h.n[stream];
    const size_t eo = (size_t)stream * z->max_events;
    std::vector<int32_t> eord(ne);                          // the kernel emits in id order: back to the caller's (track, zone) order
    for (int e = 0; e < ne; ++e) eord[e] = e;
    std::stable_sort(eord.begin(), eord.end(), [&](int x, int y) {
        return h.track[eo + x] != h.track[eo + y] ? h.track[eo + x] < h.track[eo + y] : h.zone[eo + x] < h.zone[eo + y]; });
    for (int e = 0; e < ne; ++e) {
        const size_t k = eo + eord[e];
        if (ev_track) ev_track[e] = h.track[k];
        if (ev_zone) ev_zone[e] = h.zone[k];
        if (ev_dwell) ev_dwell[e] = h.dwell[k];
        if (ev_centroid) { ev_centroid[2 * e] = h.c[k].x; ev_centroid[2 * e + 1] = h.c[k].y; }
    }
    *n_events = ne;
    return RTMODT_OK;
}

int rtmodt_zones_process_tracker(rtmodt_zones *z, rtmodt_tracker *trk, double now, int64_t frame_id, int report_tsu, int64_t *ev_track_id,
                                 int32_t *ev_zone, double *ev_dwell, float *ev_xyxy, int32_t *ev_centroid, int32_t *ev_cls, int32_t *n_events) {
    RT_CHECK(z && trk && n_events && now == now, RTMODT_E_INVALID, "bad argument");
    TrackerDeviceView v;
    RT_TRY(tracker_device_view(trk, &v));
    RT_CHECK(v.device == z->device, RTMODT_E_INVALID, "zones on device %d, tracker on device %d", z->device, v.device);
    RT_CHECK(v.n_streams <= z->S && v.max_tracks <= z->cap, RTMODT_E_INVALID, "tracker (%d streams, %d tracks) larger than the zone engine (%d, %d)",
             v.n_streams, v.max_tracks, z->S, z->Mc);
    RT_HIP(hipSetDevice(z->device));
    ZoneArgs a = make_args(z, now, frame_id);
    a.t_states = v.states; a.t_meta = v.meta; a.report_tsu = report_tsu;
    a.max_idle = -1;                                        // a track the tracker dropped never comes back: its row goes too
    RT_TRY(launch(z, a, v.n_streams, v.stream));            // the stream the tracker's last update ran on: ordered after it
    EvHost h;
    RT_TRY(fetch_events(z, v.stream, h));
    for (int s = 0; s < v.n_streams; ++s) {
        RT_TRY(check_err(z, s, h.meta[4 * s + 2]));
        const int ne = h.n[s];
        const size_t eo = (size_t)s * z->max_events;
        for (int e = 0; e < ne; ++e) {
            const size_t k = eo + e;
            if (ev_track_id) ev_track_id[k] = h.id[k];
            if (ev_zone) ev_zone[k] = h.zone[k];
            if (ev_dwell) ev_dwell[k] = h.dwell[k];
            if (ev_xyxy) memcpy(ev_xyxy + 4 * k, &h.box[k], 16);
            if (ev_centroid) { ev_centroid[2 * k] = h.c[k].x; ev_centroid[2 * k + 1] = h.c[k].y; }
            if (ev_cls) ev_cls[k] = h.cls[k];
        }
        n_events[s] = ne;
    }
    return RTMODT_OK;
}

int rtmodt_zones_state(rtmodt_zones *z, int stream, int64_t *ids, uint32_t *occ_mask, double *first_seen, double *last_alert, int32_t *n) {
    RT_CHECK(z && stream >= 0 && stream < z->S && n, RTMODT_E_INVALID, "bad argument");
    RT_HIP(hipSetDevice(z->device));
    RT_HIP(hipDeviceSynchronize());
    int64_t m[4];
    RT_HIP(hipMemcpy(m, z->d_meta + 4 * stream, sizeof(m), hipMemcpyDeviceToHost));
    RT_TRY(check_err(z, stream, m[2]));
    const int cur = (int)m[0], cnt = (int)m[1];
    const ZoneLedger &L = z->h_ledgers[stream];
    *n = cnt;
    if (cnt) {
        if (ids) RT_HIP(hipMemcpy(ids, L.id[cur], (size_t)cnt * 8, hipMemcpyDeviceToHost));
        if (occ_mask) RT_HIP(hipMemcpy(occ_mask, L.mask[cur], (size_t)cnt * 4, hipMemcpyDeviceToHost));
        if (first_seen && z->Z) RT_HIP(hipMemcpy(first_seen, L.first[cur], (size_t)cnt * z->Z * 8, hipMemcpyDeviceToHost));
        if (last_alert && z->Z) RT_HIP(hipMemcpy(last_alert, L.alert[cur], (size_t)cnt * z->Z * 8, hipMemcpyDeviceToHost));
    }
    return RTMODT_OK;
}

}  // extern "C"
