// engine.hip -- host runtime behind the detector half of include/rtmodt.h.
//
// Plays the part `ultralytics.YOLO(...)` + `model.predict(...)` play for the reference's
// Detector (/root/reference/src/detection/detector.py:82-112): reads the fused-conv weight
// file, builds the YOLOv8 graph natively for one (scale, input size, batch), lays every
// activation out in ONE device arena (fp16 NHWC, zero borders, concat-free channel slices),
// captures the forward pass into a hipGraph and drives
//   letterbox -> [graph: stem, 60-odd MFMA convs, SPPF pools, upsamples, decode] -> NMS
// on a private HIP stream.  No PyTorch, no BLAS/MIOpen: only the kernels in this directory.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include "kernels.h"

namespace rtmodt {

static void segv_trace(int sig) {
    void *bt[64];
    int n = backtrace(bt, 64);
    const char msg[] = "[rtmodt] fatal signal, backtrace:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    _exit(128 + sig);
}
static void install_debug_handlers() {
    static bool done = false;
    if (done || !rt_opt("DEBUG")) return;
    done = true;
    signal(SIGSEGV, segv_trace);
    signal(SIGABRT, segv_trace);
}

std::string &last_error() {
    static thread_local std::string e;
    return e;
}

// rt_opt() met a name that common.h's table lacks: remembered (process-wide) until an entry point reports it
static std::atomic<bool> g_bad_option{false};
static char g_bad_option_name[64] = "";
void note_bad_option(const char *name) {
    if (!g_bad_option.exchange(true)) snprintf(g_bad_option_name, sizeof(g_bad_option_name), "%s", name ? name : "(null)");
}
static int bad_option_pending() {
    if (!g_bad_option.exchange(false)) return RTMODT_OK;
    return fail(RTMODT_E_INVALID, "option RTMODT_%s is not in csrc/common.h's table (a library bug: every option the library reads must be listed there)", g_bad_option_name);
}

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

// ------------------------------------------------------------------ weight file (package weights.py)
struct WeightRec {
    std::string name;
    int cin, cout, k, stride, act;
    std::vector<f16> w;      // [cout][k][k][cin]
    std::vector<float> b;    // [cout]
};

struct WeightFile {
    int scale_id = 1, nc = 80, reg_max = 16;
    std::map<std::string, WeightRec> recs;
    std::vector<std::string> order;
};

static int read_weight_file(const char *path, WeightFile &wf) {
    std::ifstream f(path, std::ios::binary);
    RT_CHECK(f.good(), RTMODT_E_IO, "No model found at %s", path);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    RT_CHECK(raw.size() >= 40 && memcmp(raw.data(), "RTMODTW1", 8) == 0, RTMODT_E_IO, "%s: not an RTMODTW1 weight file", path);
    const uint32_t *h = (const uint32_t *)(raw.data() + 8);
    RT_CHECK(h[0] == 1, RTMODT_E_IO, "%s: unsupported version %u", path, h[0]);
    wf.scale_id = (int)h[1]; wf.nc = (int)h[2]; wf.reg_max = (int)h[3];
    uint32_t n = h[4];
    RT_CHECK(raw.size() >= 40 + (size_t)n * 72, RTMODT_E_IO, "%s: truncated record table", path);
    for (uint32_t i = 0; i < n; ++i) {
        const char *r = raw.data() + 40 + (size_t)i * 72;
        WeightRec w;
        w.name = std::string(r, strnlen(r, 32));
        const uint32_t *u = (const uint32_t *)(r + 32);
        w.cin = u[0]; w.cout = u[1]; w.k = u[2]; w.stride = u[3]; w.act = u[4];
        uint64_t woff, boff;
        memcpy(&woff, r + 56, 8); memcpy(&boff, r + 64, 8);
        size_t nw = (size_t)w.cout * w.k * w.k * w.cin;
        RT_CHECK(woff + nw * 2 <= raw.size() && boff + (size_t)w.cout * 4 <= raw.size(), RTMODT_E_IO, "%s: record %s out of bounds", path,
                 w.name.c_str());
        w.w.resize(nw); w.b.resize(w.cout);
        memcpy(w.w.data(), raw.data() + woff, nw * 2);
        memcpy(w.b.data(), raw.data() + boff, (size_t)w.cout * 4);
        wf.order.push_back(w.name);
        wf.recs[w.name] = std::move(w);
    }
    return RTMODT_OK;
}

// ------------------------------------------------------------------ letterbox geometry (ultralytics LetterBox, App. B.1)
static inline int round_half_even(double v) { return (int)std::nearbyint(v); }   // == Python round()

struct LbHost { int new_w, new_h, top, left, resize; double gain; int pad_x, pad_y; };

// rect: LetterBox(auto=True) -- new_shape is the SQUARE S x S (S = the longer side of the rectangle the
// engine was built for), the pads are whatever is left of the rectangle (== the square's pads mod 32)
static LbHost letterbox_geometry(int h, int w, int in_h, int in_w, bool rect = false) {
    LbHost g;
    const int S = std::max(in_h, in_w);
    double r = rect ? std::min((double)S / h, (double)S / w) : std::min((double)in_h / h, (double)in_w / w);
    g.new_w = round_half_even(w * r); g.new_h = round_half_even(h * r);
    double dw = (in_w - g.new_w) / 2.0, dh = (in_h - g.new_h) / 2.0;
    g.top = round_half_even(dh - 0.1); g.left = round_half_even(dw - 0.1);
    g.resize = (g.new_w != w) || (g.new_h != h);
    // scale_boxes (App. B.4)
    g.gain = std::min((double)in_h / h, (double)in_w / w);
    g.pad_x = round_half_even((in_w - w * g.gain) / 2 - 0.1);
    g.pad_y = round_half_even((in_h - h * g.gain) / 2 - 0.1);
    return g;
}

// cv::resize INTER_LINEAR 8-bit tables (same arithmetic as oracle/yolo_oracle.py:_resize_coeffs)
static void build_resize_tables(int dst, int src, std::vector<int32_t> &ofs, std::vector<int32_t> &c0, std::vector<int32_t> &c1) {
    ofs.resize(dst); c0.resize(dst); c1.resize(dst);
    double scale = 1.0 / ((double)dst / src);
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f = f - (float)s;
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= src - 1) { s = src - 1; f = 0.f; }
        ofs[d] = s;
        c0[d] = (int)std::nearbyint((1.0f - f) * 2048.0f);
        c1[d] = (int)std::nearbyint(f * 2048.0f);
    }
}

// ------------------------------------------------------------------ graph description
struct Tensor { f16 *ptr = nullptr; int H, W, C, pad; size_t per_image; };

enum OpKind { OP_STEM, OP_CONV, OP_POOL, OP_UP, OP_GROUP, OP_BNECK };

struct Op {
    OpKind kind;
    std::string name;
    ConvLaunch conv;                 // OP_CONV
    std::vector<ConvLaunch> group;   // OP_GROUP: independent convs issued as one launch
    int group_tile = TILE_64x64;
    // OP_BNECK: a Bottleneck either as ONE fused launch (`bneck`) or as its two convs (`group[0]`, `group[1]`
    // with their own tiles); the autotuner keeps whichever is faster on this device and batch
    BottleneckLaunch bneck;
    bool fused = true;
    TensorView v[4];                 // STEM: in,out [, 2.cv1's output when the front end can run fused]; POOL: y,p1,p2,p3; UP: in,out
    const f16 *stem_w = nullptr; const float *stem_b = nullptr;
    // STEM: the front end as ONE launch (front.hip: stem -> layer 1 -> C2f.cv1 of layer 2; only v[2] is stored).  `front_ok`: the graph allows it;
    // `front_on`: it runs (the ops of layer 1 and 2.cv1 are then skipped)
    bool front_ok = false, front_on = false;
    const f16 *front_w1 = nullptr, *front_w2 = nullptr; const float *front_b1 = nullptr, *front_b2 = nullptr;
    int front_kp1 = 0, front_kp2 = 0, front_c1 = 0, front_c2 = 0;
    int64_t flops = 0;               // per frame
    int B = 1;                       // images this launch covers
    int head_level = -1;             // >= 0: belongs to the Detect branch of that level (independent of the other levels)
    // conv -> 1x1 pairs (backbone Conv -> C2f.cv1): the producer carries the 1x1 as conv.tail_*; when the tuner finds the
    // fused launch faster, `tail_on` runs it with `tail_tile` and the 1x1's own op is skipped
    bool tail_on = false, skip = false;
    int tail_tile = TILE_TAIL_128x64;
};

}  // namespace rtmodt

using namespace rtmodt;

constexpr int N_EXEC = 2;       // executable instances per captured graph of the plain / chained engines, alternating between successive batches (capture_chain)

struct rtmodt_detector {
    rtmodt_det_cfg cfg{};
    std::string weight_path;
    std::vector<int32_t> classes;
    int device = 0;
    hipStream_t stream = nullptr;
    int scale_id = 1, nc = 80, reg_max = 16;
    int B = 1, in_h = 640, in_w = 640, n_anchors = 0;
    bool rect = false;
    // arena
    char *arena = nullptr;
    size_t arena_bytes = 0;
    std::vector<Tensor> tensors;
    std::vector<Op> ops;                              // whole batch, one op per launch (eager path, profiler)
    // The batch is also cut into `n_chains` sub-batches whose op lists are captured as PARALLEL
    // branches of the forward graph (plus one branch per Detect level): the layers of this net
    // are small and latency-bound, so independent chains in flight hide each other's launch
    // gaps, first-load latency and epilogue tails.
    int n_chains = 1;
    // STAGED mode (cfg.chains = -1): the net is cut after SPPF into a front stage (stem .. layer 9) on the main stream and
    // a back stage (neck, Detect, decode) on a second stream with a hardware queue of its own; each stage runs the WHOLE
    // batch (the bigger GEMMs of a 16-frame launch) and the front of batch t + 1 overlaps the back of batch t.  The
    // stages of consecutive batches work in alternate copies of the activation arena (ring slot parity).
    bool pipe = false;
    size_t arena_stride = 0;                          // bytes between the two arena copies
    static constexpr int MAX_STAGES = 3;
    int n_stages = 1;                                 // 2: backbone | neck + Detect; 3: layers 0-6.m.1 | 6.cv2-16 | 18-22
    int stage_lo[MAX_STAGES + 1] = {0, 0, 0, 0};      // stage s runs ops [stage_lo[s], stage_lo[s + 1])
    int run_par = 0, last_par = 0;                    // arena copy the launches being issued use / the newest batch used
    std::vector<Op> par_ops[MAX_STAGES];              // d->ops shifted into each arena copy (one copy per stage)
    hipGraph_t pipe_graph[MAX_STAGES][MAX_STAGES] = {};       // [arena copy][stage]
    hipGraphExec_t pipe_exec[MAX_STAGES][MAX_STAGES][2] = {};   // [arena copy][stage][instance]: a copy's successive uses alternate instances
    hipStream_t stage_stream[MAX_STAGES] = {nullptr, nullptr, nullptr};      // [0] = the main stream
    bool chain_free_run = true;                       // chains never wait for each other (RTMODT_CHAIN_JOIN=1: join on the main stream per batch)
    std::vector<std::vector<Op>> chain_ops;
    std::vector<hipStream_t> aux_streams;             // fork targets during capture
    std::vector<hipStream_t> pad_streams;             // RTMODT_PAD_STREAMS test hook
    std::vector<hipEvent_t> aux_events;
    std::map<std::string, TensorView> layer_out;     // fused conv name -> output view
    std::vector<void *> dev_allocs;                   // weights etc.
    int img_t = -1;
    NmsPlan nms_plan;                                 // resolved at create time
    f16 *d_zeros = nullptr;                           // 256 zero bytes (DMA source for out-of-tensor halo pixels)
    int head_t[3] = {-1, -1, -1};
    // Detect's last 1x1 convs + decode as ONE launch (postprocess.hip: head_final); `stage2_op` = index of the grouped
    // launch it replaces (kept for the debug paths that want the head tensors themselves)
    bool head_final = false; int stage2_op = -1;
    HeadFinalArgs hf{};
    int64_t flops_per_frame = 0;
    // frames
    // host frames land in a per-ring-slot staging area through their own copy stream: the H2D of batch
    // t+1 runs under the forward pass of batch t when the caller's frames are in pinned memory
    uint8_t *stage = nullptr; size_t stage_per = 0;      // [RING_SLOTS][B][stage_per]
    hipStream_t copy_stream = nullptr;
    // RTMODT_ZERO_COPY=1: page-locked frames are read in place by the stem (opt-in: the stem then waits on PCIe with every CU's
    // wave slots taken, 11.8 k frames/s against 17.4 k for one DMA per step in front of stage 1 -- profiles/r02/README.md)
    bool zero_copy = false, last_in_place = false;
    int h2d_mode = 0;                                 // RTMODT_H2D: 0 = stream-ordered upload (events), 1 = host-synchronised upload (no events)
    FramePtrs fptrs{};
    // letterbox folded into the stem conv whenever the frames need no resize (conv.hip: stem_fused); the stem is
    // launched by enqueue_batch itself, never from the captured graph (its frame pointers change every batch)
    f16 *lut255 = nullptr;                           // c / 255 as fp16, c = 0..255
    bool stem_fuse = true, last_fused = false;
    LetterboxGeom last_lg{}; int last_pitch = 0;
    int tab_h = -1, tab_w = -1;
    int32_t *d_tab = nullptr; size_t tab_cap = 0;
    ResizeTables tabs{};
    // postprocess
    // decode's dense per-anchor outputs, one set per ring slot: NMS of batch t (post stream) reads
    // set t%2 while the forward pass of batch t+1 (main stream) fills the other one
    struct Dense { float4 *box = nullptr; float *score = nullptr; int32_t *cls = nullptr; };
    Dense dense[4];                                   // [RING_SLOTS]
    int cur_dense = 0;
    uint64_t batch_no = 0;                            // batches enqueued so far (graph instance / arena copy = parity)
    float *d_pred = nullptr;
    hipStream_t post_stream = nullptr;                // NMS + D2H + tracker run here, overlapped with the next forward
    uint64_t *d_keys = nullptr; float4 *d_sbox = nullptr; int32_t *d_sidx = nullptr;
    // results: a ring of RING_SLOTS batches may be in flight (enqueue t+1 [, t+2] before fetching t)
    struct Slot {
        float *o_xyxy = nullptr, *o_conf = nullptr; int32_t *o_cls = nullptr, *o_anchor = nullptr, *o_n = nullptr;   // device
        float *h_xyxy = nullptr, *h_conf = nullptr; int32_t *h_cls = nullptr, *h_n = nullptr;                        // pinned host
        hipEvent_t ev0 = nullptr, evp = nullptr, ev1 = nullptr, ev2 = nullptr, done = nullptr, decoded = nullptr;   // evp: letterbox done
        hipEvent_t copied = nullptr;       // this slot's frames have arrived in its staging area
        std::vector<hipEvent_t> chain_done; // [n_chains] (entry 0 unused): chain c has finished this slot's sub-batch
        hipEvent_t stage_done[2] = {nullptr, nullptr};   // staged mode: stage s of this slot's batch is done (s < last)
        bool staged = false;               // the staging area has been read by a letterbox launch (evp is meaningful)
        bool chained = false, joined = false;   // this slot's batch ran as sub-batch chains; the main stream has waited for all of them
        int n = 0;
    };
    static constexpr int RING_SLOTS = 4;               // staged engine with S stages: S + 1 batches in flight
    Slot slots[RING_SLOTS];
    int head = 0, n_pending = 0;          // next slot to fill; batches enqueued but not fetched
    int newest = -1, last_fetched = -1;
    uint64_t class_mask[2] = {~0ull, ~0ull};
    // graph
    std::vector<hipGraph_t> graphs; std::vector<hipGraphExec_t> graph_execs;   // one captured graph per sub-batch chain, N_EXEC executable instances of each
    std::vector<hipStream_t> chain_streams;           // [n_chains], chain 0 runs on `stream`
    std::vector<hipEvent_t> chain_fork, chain_join;
    bool want_pred = false;
    int last_h = 0, last_w = 0;
    // profile storage
    std::vector<std::string> prof_names;
    // in-kernel shader clock, sampled by one wave behind every batch's NMS while enabled (rtmodt_detector_clock_*)
    bool clock_on = false;
    unsigned long long *d_clock = nullptr;            // [CLOCK_SLOTS][4] = {s_memtime, s_memrealtime} at the start and the end of a sample
    uint64_t clock_n = 0;
    static constexpr int CLOCK_SLOTS = 4096;
};

namespace rtmodt {

// one wave: {s_memtime, s_memrealtime} now and again `ticks` 100-MHz ticks later (bounded: at most 4096 sleeps of ~64 x 64 clk)
__global__ void clock_sample_kernel(unsigned long long *out, unsigned ticks) {
    if (threadIdx.x != 0) return;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    t1 = t0; r1 = r0;
    for (int i = 0; i < 4096 && r1 - r0 < ticks; ++i) {
        __builtin_amdgcn_s_sleep(64);
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    }
    out[0] = t0; out[1] = r0; out[2] = t1; out[3] = r1;
}

struct Builder {
    rtmodt_detector *d;
    WeightFile *wf;
    size_t arena_used = 0;

    int T(int H, int W, int C, int pad) {
        Tensor t;
        t.H = H; t.W = W; t.C = C; t.pad = pad;
        t.per_image = (size_t)(H + 2 * pad) * (W + 2 * pad) * C;
        size_t bytes = align_up(t.per_image * d->B * sizeof(f16), 256);
        t.ptr = (f16 *)(uintptr_t)arena_used;            // offset for now; rebased after allocation
        arena_used += bytes + 256;                         // slack: tail tiles of the conv re-read within the tensor only, keep tensors apart
        d->tensors.push_back(t);
        return (int)d->tensors.size() - 1;
    }
    TensorView V(int t, int coff = 0, int c = -1) {
        const Tensor &x = d->tensors[t];
        TensorView v;
        v.base = x.ptr; v.H = x.H; v.W = x.W; v.C = x.C; v.pad = x.pad; v.coff = coff; v.c = c < 0 ? x.C : c;
        return v;
    }
};

static int upload(rtmodt_detector *d, const void *src, size_t bytes, void **out) {
    void *p = nullptr;
    RT_HIP(hipMalloc(&p, bytes));
    RT_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    d->dev_allocs.push_back(p);
    *out = p;
    return RTMODT_OK;
}

// the same launch restricted to images [b0, b0 + nb) of the batch
static Op sub_batch(const Op &op, int b0, int nb) {
    Op o = op;
    auto shift = [&](TensorView &v) {
        if (v.c) v.base += (size_t)b0 * (v.H + 2 * v.pad) * v.padded_w() * v.C;
    };
    o.B = nb;
    if (o.kind == OP_CONV) { shift(o.conv.in); shift(o.conv.out); shift(o.conv.res); shift(o.conv.out2); shift(o.conv.tail_out); shift(o.conv.in_lo); o.conv.B = nb; }
    else if (o.kind == OP_GROUP || o.kind == OP_BNECK) {
        for (auto &c : o.group) { shift(c.in); shift(c.out); shift(c.res); c.B = nb; }
        if (o.kind == OP_BNECK) { shift(o.bneck.in); shift(o.bneck.out); shift(o.bneck.res); shift(o.bneck.tail_in); shift(o.bneck.tail_out); o.bneck.B = nb; }
    }
    else for (auto &v : o.v) shift(v);
    return o;
}

// the same launch in the other copy of the arena
static Op shift_arena(const Op &op, size_t bytes) {
    Op o = op;
    auto shift = [&](TensorView &v) { if (v.c) v.base = (f16 *)((char *)v.base + bytes); };
    if (o.kind == OP_CONV) { shift(o.conv.in); shift(o.conv.out); shift(o.conv.res); shift(o.conv.out2); shift(o.conv.tail_out); shift(o.conv.in_lo); }
    else if (o.kind == OP_GROUP || o.kind == OP_BNECK) {
        for (auto &c : o.group) { shift(c.in); shift(c.out); shift(c.res); }
        if (o.kind == OP_BNECK) { shift(o.bneck.in); shift(o.bneck.out); shift(o.bneck.res); shift(o.bneck.tail_in); shift(o.bneck.tail_out); }
    }
    else for (auto &v : o.v) shift(v);
    return o;
}

static void set_chains(rtmodt_detector *d, int chains) {
    d->n_chains = chains;
    d->chain_ops.assign(chains, {});
    const int nb = d->B / chains;
    for (int c = 0; c < chains; ++c)
        for (auto &op : d->ops) d->chain_ops[c].push_back(sub_batch(op, c * nb, nb));
}

static int pick_tile(int M, int cout) {
    if (const char *e = rt_opt("TILE")) {
        int t = atoi(e);
        if (t >= 0 && t < TILE_COUNT) return t;
    }
    static const float eff[TILE_COUNT] = {1.0f, 0.85f, 0.62f, 0.75f, 0.85f, 0.5f, 0.4f, 0.4f};      // the rest (0): only reachable through the autotuner
    static const int occ[TILE_COUNT] = {3, 5, 8, 4, 5, 2, 3, 3};
    int best = 0;
    double best_cost = 1e30;
    for (int t = 0; t < TILE_COUNT; ++t) {
        if (eff[t] <= 0.f) continue;                       // only reachable through the autotuner
        TileShape ts = tile_shape(t);
        long nblk = (long)cdiv(M, ts.bm) * cdiv(cout, ts.bn);
        long slots = 256L * occ[t];
        double waves = std::ceil((double)nblk / slots);
        // partial last wave costs less than a full one when blocks are few
        double fill = (double)nblk / (waves * slots);
        double cost = waves * ts.bm * ts.bn / eff[t] * (0.5 + 0.5 * std::max(fill, 1.0 / occ[t]));
        if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return best;
}

// upload a fused conv (one or more records concatenated along cout) in the MFMA layout
static bool tile_legal(const ConvLaunch *c, int n, int t);
static int make_conv(rtmodt_detector *d, WeightFile &wf, const std::vector<std::string> &names, const std::string &op_name,
                     const TensorView &in, const TensorView &out, const TensorView *res, int cout_pad4,
                     std::vector<Op> *dst = nullptr) {
    std::vector<const WeightRec *> rs;
    for (auto &n : names) {
        auto it = wf.recs.find(n);
        RT_CHECK(it != wf.recs.end(), RTMODT_E_IO, "weight file lacks conv %s", n.c_str());
        rs.push_back(&it->second);
    }
    const WeightRec &r0 = *rs[0];
    int cout = 0;
    for (auto *r : rs) {
        RT_CHECK(r->cin == r0.cin && r->k == r0.k && r->stride == r0.stride && r->act == r0.act, RTMODT_E_IO, "cannot fuse %s", r->name.c_str());
        cout += r->cout;
    }
    int cout_eff = cout_pad4 ? (int)align_up(cout, 4) : cout;
    RT_CHECK(in.c == r0.cin, RTMODT_E_IO, "conv %s: graph expects cin %d, file has %d", op_name.c_str(), in.c, r0.cin);
    RT_CHECK(out.c == cout_eff, RTMODT_E_IO, "conv %s: graph expects cout %d, file has %d", op_name.c_str(), out.c, cout_eff);
    int K = r0.k * r0.k * r0.cin, kp = (int)align_up(K, 32), cp = (int)align_up(cout_eff, 128);
    std::vector<f16> w((size_t)cp * kp, (f16)0.0f);
    std::vector<float> b(cp, 0.f);
    int row = 0;
    for (auto *r : rs) {
        for (int o = 0; o < r->cout; ++o, ++row) {
            memcpy(&w[(size_t)row * kp], &r->w[(size_t)o * K], (size_t)K * sizeof(f16));
            b[row] = r->b[o];
        }
    }
    Op op;
    op.kind = OP_CONV; op.name = op_name;
    void *dw, *db;
    RT_TRY(upload(d, w.data(), w.size() * sizeof(f16), &dw));
    RT_TRY(upload(d, b.data(), b.size() * sizeof(float), &db));
    ConvLaunch &c = op.conv;
    c.in = in; c.out = out; if (res) c.res = *res;
    c.wt = (const f16 *)dw; c.bias = (const float *)db;
    c.B = d->B; c.cin = r0.cin; c.cout = cout_eff; c.ks = r0.k; c.stride = r0.stride; c.act = r0.act; c.kp = kp;
    if (const char *e = rt_opt("EPI16")) c.epilogue = atoi(e);      // A/B and test hook
    int M = d->B * out.H * out.W;
    c.tile = pick_tile(M, cout_eff);
    if (const char *e = rt_opt("TILE_K64")) {            // test hook: a 64-deep tile (incl. the 8-wave ones) wherever it is legal
        const int t = atoi(e);
        if (t >= 0 && t < TILE_COUNT && tile_needs_cin64(t) && !tile_is_rows(t) && !tile_is_tail(t) && c.cin % 64 == 0 && kp % 64 == 0 && !dst &&
            (!tile_is_pt(t) || (((long)d->B * out.H * out.W) % tile_shape(t).bm == 0 && cout_eff % tile_shape(t).bn == 0 && out.coff % 8 == 0 && out.C % 8 == 0)) &&
            (tile_shape(t).bn <= 128 || cout_eff % tile_shape(t).bn == 0) && (!tile_is_ppt(t) || tile_legal(&c, 1, t))) c.tile = t;
    }
    if (const char *e = rt_opt("TILE_3X3S1")) {          // test hook: force a tap-reuse tile wherever it is legal
        int t = atoi(e);
        if (t >= 0 && t < TILE_COUNT && tile_is_rows(t) && c.ks == 3 && c.stride == 1 && c.in.pad == 1 &&
            c.cin % (tile_needs_cin64(t) ? 64 : 32) == 0 && (!tile_is_pp(t) || tile_legal(&c, 1, t)))
            c.tile = t;
    }
    op.flops = 2LL * out.H * out.W * cout * K;
    (dst ? *dst : d->ops).push_back(op);
    int off = 0;
    for (auto *r : rs) {
        TensorView lv = out; lv.coff = out.coff + off; lv.c = r->cout;
        d->layer_out[r->name] = lv;
        off += r->cout;
    }
    return RTMODT_OK;
}

static int build_graph(rtmodt_detector *d, WeightFile &wf) {
    static const double SC[5][3] = {{0.33, 0.25, 1024}, {0.33, 0.50, 1024}, {0.67, 0.75, 768}, {1.00, 1.00, 512}, {1.00, 1.25, 512}};
    RT_CHECK(d->scale_id >= 0 && d->scale_id < 5, RTMODT_E_IO, "bad model scale id %d", d->scale_id);
    RT_CHECK(wf.reg_max == 16, RTMODT_E_UNSUPPORTED, "reg_max %d (only 16 built)", wf.reg_max);
    const double dep = SC[d->scale_id][0], wid = SC[d->scale_id][1], mx = SC[d->scale_id][2];
    auto ch = [&](int c) { return (int)(std::ceil(std::min((double)c, mx) * wid / 8.0) * 8); };
    auto rep = [&](int n) { return std::max((int)std::nearbyint(n * dep), 1); };   // Python round(): half-even
    Builder bld{d, &wf};
    auto T = [&](int H, int W, int C, int pad) { return bld.T(H, W, C, pad); };
    auto V = [&](int t, int coff = 0, int c = -1) { return bld.V(t, coff, c); };
    int rc = RTMODT_OK;
    auto conv = [&](const std::string &name, const TensorView &in, const TensorView &out, const TensorView *res = nullptr) {
        if (rc == RTMODT_OK) rc = make_conv(d, wf, {name}, name, in, out, res, 0);
    };
    auto c2f = [&](const std::string &i, const TensorView &in, int cout, int n, bool shortcut, const TensorView &out) {
        int c = cout / 2;
        int cat = T(in.H, in.W, (2 + n) * c, 1);
        conv(i + ".cv1", in, V(cat, 0, 2 * c));
        for (int j = 0; j < n; ++j) {
            int tmp = T(in.H, in.W, c, 1);
            std::string m = i + ".m." + std::to_string(j);
            TensorView r = V(cat, (1 + j) * c, c);
            static const bool no_fuse = rt_diag("NO_BNECK_FUSE") != nullptr;
            if (bottleneck_supported(c) && !no_fuse && rc == RTMODT_OK) {
                std::vector<Op> pair;
                rc = make_conv(d, wf, {m + ".cv1"}, m + ".cv1", V(cat, (1 + j) * c, c), V(tmp), nullptr, 0, &pair);
                if (rc == RTMODT_OK) rc = make_conv(d, wf, {m + ".cv2"}, m + ".cv2", V(tmp), V(cat, (2 + j) * c, c), shortcut ? &r : nullptr, 0, &pair);
                if (rc != RTMODT_OK) continue;
                Op op; op.kind = OP_BNECK; op.name = m + " (cv1+cv2)";
                op.group = {pair[0].conv, pair[1].conv};
                op.flops = pair[0].flops + pair[1].flops;
                BottleneckLaunch &b = op.bneck;
                b.in = V(cat, (1 + j) * c, c); b.out = V(cat, (2 + j) * c, c); if (shortcut) b.res = r;
                b.w1 = pair[0].conv.wt; b.b1 = pair[0].conv.bias; b.w2 = pair[1].conv.wt; b.b2 = pair[1].conv.bias;
                b.zeros = d->d_zeros; b.B = d->B; b.c = c; b.kp = pair[0].conv.kp;
                if (const char *e = rt_opt("BNECK")) op.fused = atoi(e) != 0;
                if (const char *e = rt_opt("BNECK32")) b.persistent32 = atoi(e) != 0;      // A/B and test hook: 0 = bottleneck_fused for the c = 32 C2f too
                d->ops.push_back(op);
                continue;
            }
            conv(m + ".cv1", V(cat, (1 + j) * c, c), V(tmp));
            conv(m + ".cv2", V(tmp), V(cat, (2 + j) * c, c), shortcut ? &r : nullptr);
        }
        conv(i + ".cv2", V(cat, 0, (2 + n) * c), out);
        // C2f with ONE Bottleneck of 32 / 64 channels run as a fused launch: cv2 can ride along as its tail (bottleneck.hip),
        // reading the two earlier chunks from the concat tensor and this Bottleneck's output from LDS
        if (rc == RTMODT_OK && n == 1 && (c == 32 || c == 64) && d->ops.size() >= 2 && !rt_diag("NO_BNECK_TAIL")) {
            Op &bn = d->ops[d->ops.size() - 2], &cv = d->ops.back();
            if (bn.kind == OP_BNECK && cv.kind == OP_CONV && cv.conv.cout == 2 * c && cv.conv.kp == 3 * c && !cv.conv.res.base && cv.conv.out.coff % 8 == 0 &&
                cv.conv.out.C % 8 == 0) {
                bn.bneck.tail_in = V(cat, 0, 2 * c); bn.bneck.tail_out = cv.conv.out; bn.bneck.tail_wt = cv.conv.wt; bn.bneck.tail_bias = cv.conv.bias;
                bn.bneck.tail_cout = cv.conv.cout; bn.bneck.tail_kp = cv.conv.kp; bn.bneck.tail_act = cv.conv.act;
                if (const char *e = rt_opt("BNECK_TAIL")) { bn.tail_on = atoi(e) != 0; cv.skip = bn.tail_on && bn.fused; }   // test hook (no autotune)
            }
        }
    };

    const int H = d->in_h, W = d->in_w;
    const int c1 = ch(64), c2 = ch(128), c3 = ch(256), c4 = ch(512), c5 = ch(1024);
    d->img_t = T(H, W, 4, 1);
    int t0 = T(H / 2, W / 2, c1, 1);
    {   // stem
        auto it = wf.recs.find("0");
        RT_CHECK(it != wf.recs.end(), RTMODT_E_IO, "weight file lacks conv 0");
        const WeightRec &r = it->second;
        RT_CHECK(r.cin == 3 && r.k == 3 && r.stride == 2 && r.cout == c1, RTMODT_E_IO, "stem conv shape mismatch");
        std::vector<f16> w(64 * (size_t)c1, (f16)0.0f);   // [cout][k' = kh*16 + kw*4 + c], zero elsewhere
        for (int o = 0; o < c1; ++o)
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw)
                    for (int c = 0; c < 3; ++c) w[(size_t)o * 64 + kh * 16 + kw * 4 + c] = r.w[(size_t)o * 27 + (kh * 3 + kw) * 3 + c];
        void *dw, *db;
        RT_TRY(upload(d, w.data(), w.size() * sizeof(f16), &dw));
        RT_TRY(upload(d, r.b.data(), r.b.size() * 4, &db));
        Op op; op.kind = OP_STEM; op.name = "0";
        op.v[0] = V(d->img_t); op.v[1] = V(t0); op.stem_w = (const f16 *)dw; op.stem_b = (const float *)db;
        op.flops = 2LL * (H / 2) * (W / 2) * c1 * 27;
        d->ops.push_back(op);
        d->layer_out["0"] = V(t0);
    }
    int t1 = T(H / 4, W / 4, c2, 1);
    conv("1", V(t0), V(t1));
    int t2 = T(H / 4, W / 4, c2, 1);
    c2f("2", V(t1), c2, rep(3), true, V(t2));
    int t3 = T(H / 8, W / 8, c3, 1);
    conv("3", V(t2), V(t3));
    // neck concat tensors; producers write straight into their slices
    int cat11 = T(H / 16, W / 16, c5 + c4, 1);
    int cat14 = T(H / 8, W / 8, c4 + c3, 1);
    int cat17 = T(H / 16, W / 16, c3 + c4, 1);
    int cat20 = T(H / 32, W / 32, c4 + c5, 1);
    TensorView out4 = V(cat14, c4, c3);
    c2f("4", V(t3), c3, rep(6), true, out4);
    int t5 = T(H / 16, W / 16, c4, 1);
    conv("5", out4, V(t5));
    TensorView out6 = V(cat11, c5, c4);
    c2f("6", V(t5), c4, rep(6), true, out6);
    int t7 = T(H / 32, W / 32, c5, 1);
    conv("7", out6, V(t7));
    int t8 = T(H / 32, W / 32, c5, 1);
    c2f("8", V(t7), c5, rep(3), true, V(t8));
    // SPPF
    int ch9 = c5 / 2;
    int cat9 = T(H / 32, W / 32, 4 * ch9, 1);
    conv("9.cv1", V(t8), V(cat9, 0, ch9));
    {
        Op op; op.kind = OP_POOL; op.name = "9.pool";
        op.v[0] = V(cat9, 0, ch9); op.v[1] = V(cat9, ch9, ch9); op.v[2] = V(cat9, 2 * ch9, ch9); op.v[3] = V(cat9, 3 * ch9, ch9);
        d->ops.push_back(op);
    }
    TensorView out9 = V(cat20, c4, c5);
    conv("9.cv2", V(cat9), out9);
    const bool fold_up = rt_diag("NO_UPFOLD") == nullptr;   // A/B switch: separate upsample2 launches instead
    // a conv that also writes the nearest-2x copy cannot run as the tail of the Bottleneck before it
    auto fold_into_last = [&](const TensorView &up) {
        d->ops.back().conv.out2 = up;
        if (tile_is_pt(d->ops.back().conv.tile) || tile_is_ppt(d->ops.back().conv.tile)) d->ops.back().conv.tile = TILE_K64_128x128_S2_W8;      // (test hook's forced tile: no second destination there)
        d->ops.back().skip = false;
        if (d->ops.size() >= 2) { Op &bn = d->ops[d->ops.size() - 2]; if (bn.kind == OP_BNECK) { bn.bneck.tail_wt = nullptr; bn.tail_on = false; } }
    };
    // Upsample + Concat of the neck (layers 10/11 and 13/14), cheapest first: (a) the consumer C2f.cv1 -- a 1x1 -- reads the
    // upsampled channels straight from the half-resolution tensor (no copy exists at all), (b) the producer's epilogue also
    // writes the nearest-2x copy into the concat slice, (c) separate upsample launches (RTMODT_NO_UPFOLD)
    const bool read_lo = fold_up && !(rt_opt("UP_READ") && atoi(rt_opt("UP_READ")) == 0);
    auto read_from_lo = [&](const std::string &cv1, const TensorView &lo, int lo_c) -> bool {
        for (auto &op : d->ops)
            if (op.kind == OP_CONV && op.name == cv1 && lo_c % 64 == 0 && op.conv.cin % 64 == 0 && op.conv.kp % 64 == 0 && op.conv.ks == 1) {
                op.conv.in_lo = lo; op.conv.lo_c = lo_c;
                if (!tile_reads_lo(op.conv.tile)) op.conv.tile = TILE_K64_128x64_S3_W8;
                return true;
            }
        return false;
    };
    const bool lo11 = read_lo && c5 % 64 == 0 && (c5 + c4) % 64 == 0, lo14 = read_lo && c4 % 64 == 0 && (c4 + c3) % 64 == 0;
    if (lo11) {}                                             // 12.cv1 is pointed at out9 once it exists (below)
    else if (fold_up) fold_into_last(V(cat11, 0, c5));        // layer 10 (Upsample) + 11 (Concat) folded into 9.cv2's epilogue
    else { Op op; op.kind = OP_UP; op.name = "10.up"; op.v[0] = out9; op.v[1] = V(cat11, 0, c5); d->ops.push_back(op); }
    TensorView out12 = V(cat17, c3, c4);
    c2f("12", V(cat11), c4, rep(3), false, out12);
    if (lo11) RT_CHECK(read_from_lo("12.cv1", out9, c5), RTMODT_E_INVALID, "12.cv1 cannot read the half-resolution source");
    if (lo14) {}
    else if (fold_up) fold_into_last(V(cat14, 0, c4));        // layer 13 (Upsample) + 14 (Concat) folded into 12.cv2's epilogue
    else { Op op; op.kind = OP_UP; op.name = "13.up"; op.v[0] = out12; op.v[1] = V(cat14, 0, c4); d->ops.push_back(op); }
    int t15 = T(H / 8, W / 8, c3, 1);
    c2f("15", V(cat14), c3, rep(3), false, V(t15));
    if (lo14) RT_CHECK(read_from_lo("15.cv1", out12, c4), RTMODT_E_INVALID, "15.cv1 cannot read the half-resolution source");
    conv("16", V(t15), V(cat17, 0, c3));
    int t18 = T(H / 16, W / 16, c4, 1);
    c2f("18", V(cat17), c4, rep(3), false, V(t18));
    conv("19", V(t18), V(cat20, 0, c4));
    int t21 = T(H / 32, W / 32, c5, 1);
    c2f("21", V(cat20), c5, rep(3), false, V(t21));
    RT_TRY(rc);
    // conv -> C2f.cv1 pairs of the backbone ("1" -> "2.cv1", "3" -> "4.cv1"): the conv's output tensor has no other
    // reader, so the 1x1 can run as a tail of the conv's launch (conv.hip: epilogue_tail) and the tensor never exists
    if (!rt_diag("NO_TAIL"))
        for (size_t i = 0; i + 1 < d->ops.size(); ++i) {
            Op &a = d->ops[i], &b = d->ops[i + 1];
            if (a.kind != OP_CONV || b.kind != OP_CONV || b.conv.ks != 1 || b.conv.stride != 1 || a.conv.res.base || a.conv.out2.base || b.conv.res.base || b.conv.out2.base) continue;
            if (a.name.find('.') != std::string::npos || b.name != std::to_string(atoi(a.name.c_str()) + 1) + ".cv1") continue;   // plain backbone Conv feeding the next C2f
            if (b.conv.in.base != a.conv.out.base || b.conv.in.coff != a.conv.out.coff || b.conv.in.c != a.conv.cout) continue;
            if ((a.conv.cout != 64 && a.conv.cout != 128) || a.conv.cin % 32 != 0 || b.conv.cout > a.conv.cout || b.conv.cout % 8 != 0 || b.conv.kp != a.conv.cout) continue;
            if (b.conv.out.coff % 8 != 0 || b.conv.out.C % 8 != 0) continue;
            a.conv.tail_out = b.conv.out; a.conv.tail_wt = b.conv.wt; a.conv.tail_bias = b.conv.bias;
            a.conv.tail_cout = b.conv.cout; a.conv.tail_kp = b.conv.kp; a.conv.tail_act = b.conv.act;
            a.tail_tile = a.conv.cout == 64 ? TILE_TAIL_128x64 : (a.conv.cin % 64 == 0 ? TILE_TAIL_K64_128x128 : TILE_TAIL_128x64);
            if (const char *e = rt_opt("TAIL")) {     // test hook (no autotune): force the fused launch on
                a.tail_on = atoi(e) != 0 && tile_shape(a.tail_tile).bn == a.conv.cout && !(tile_needs_cin64(a.tail_tile) && a.conv.cin % 64 != 0);
                b.skip = a.tail_on;
            }
        }
    // The front end as one launch (front.hip): stem -> "1" -> "2.cv1", whose two intermediate tensors have no other reader
    if (d->ops.size() >= 3 && !rt_diag("NO_FRONT")) {
        Op &st = d->ops[0], &l1 = d->ops[1], &cv = d->ops[2];
        if (st.kind == OP_STEM && l1.kind == OP_CONV && l1.name == "1" && cv.kind == OP_CONV && cv.name == "2.cv1" && l1.conv.ks == 3 && l1.conv.stride == 2 &&
            l1.conv.act == 1 && cv.conv.act == 1 && cv.conv.ks == 1 && cv.conv.stride == 1 && !l1.conv.res.base && !cv.conv.res.base && !l1.conv.out2.base && !cv.conv.out2.base &&
            l1.conv.in.base == st.v[1].base && l1.conv.cin == st.v[1].c && cv.conv.in.base == l1.conv.out.base && cv.conv.in.coff == l1.conv.out.coff && cv.conv.cin == l1.conv.cout &&
            front_supported(st.v[1].c, l1.conv.cout, cv.conv.cout, d->in_h, d->in_w) && cv.conv.out.coff % 8 == 0 && cv.conv.out.C % 8 == 0) {
            st.front_ok = true;
            st.v[2] = cv.conv.out;
            st.front_w1 = l1.conv.wt; st.front_b1 = l1.conv.bias; st.front_kp1 = l1.conv.kp; st.front_c1 = l1.conv.cout;
            st.front_w2 = cv.conv.wt; st.front_b2 = cv.conv.bias; st.front_kp2 = cv.conv.kp; st.front_c2 = cv.conv.cout;
            if (const char *e = rt_opt("FRONT")) {         // test hook (no autotune): force the fused front end on / off
                st.front_on = atoi(e) != 0;
                l1.skip = cv.skip = st.front_on;
            }
        }
    }
    // Detect head: the two first 3x3 convs of a level share their input -> one conv, cout = cbox + ccls
    const int cbox = std::max(16, std::max(c3 / 4, 64)), ccls = std::max(c3, std::min(d->nc, 100));
    const int nc4 = (int)align_up(d->nc, 4), no = 64 + (int)align_up(d->nc, 8);
    const int src[3] = {t15, t18, t21};
    std::vector<Op> hops;                                   // 5 convs per level: A, B2, B3, C2, C3
    for (int l = 0; l < 3; ++l) {
        const Tensor s = d->tensors[src[l]];               // BY VALUE: every T() below appends to d->tensors and may move it (a reference dangled as soon as the
                                                           // vector grew at this point -- YOLOv8l, whose deeper C2f modules allocate more tensors, got garbage head shapes)
        std::string L = std::to_string(l);
        int hA = T(s.H, s.W, cbox + ccls, 1);
        RT_TRY(make_conv(d, wf, {"22.cv2." + L + ".0", "22.cv3." + L + ".0"}, "22.cv2+cv3." + L + ".0", V(src[l]), V(hA), nullptr, 0, &hops));
        int hB2 = T(s.H, s.W, cbox, 0), hB3 = T(s.H, s.W, ccls, 0);
        RT_TRY(make_conv(d, wf, {"22.cv2." + L + ".1"}, "22.cv2." + L + ".1", V(hA, 0, cbox), V(hB2), nullptr, 0, &hops));
        RT_TRY(make_conv(d, wf, {"22.cv3." + L + ".1"}, "22.cv3." + L + ".1", V(hA, cbox, ccls), V(hB3), nullptr, 0, &hops));
        d->head_t[l] = T(s.H, s.W, no, 0);
        RT_TRY(make_conv(d, wf, {"22.cv2." + L + ".2"}, "22.cv2." + L + ".2", V(hB2), V(d->head_t[l], 0, 64), nullptr, 0, &hops));
        RT_TRY(make_conv(d, wf, {"22.cv3." + L + ".2"}, "22.cv3." + L + ".2", V(hB3), V(d->head_t[l], 64, nc4), nullptr, 1, &hops));
    }
    // Detect head launches: 0 = 15 separate convs; 1 = the two branches of a level share a launch
    // (9 launches); 2 = every level and branch of a stage in one launch (3 launches)
    int grouping = 2;
    if (const char *e = rt_diag("HEAD_GROUP")) grouping = atoi(e);
    auto add_group = [&](const std::string &name, std::initializer_list<int> idx) {
        Op g; g.kind = OP_GROUP; g.name = name;
        for (int i : idx) { g.group.push_back(hops[i].conv); g.flops += hops[i].flops; }
        bool same = true;
        for (auto &c : g.group) same = same && c.tile == g.group[0].tile;
        if (same && tile_is_rows(g.group[0].tile)) g.group_tile = g.group[0].tile;   // RTMODT_TILE_3X3S1 test hook
        d->ops.push_back(g);
    };
    if (grouping == 2) {
        add_group("22.stage0 (cv2+cv3 x3 levels)", {0, 5, 10});
        add_group("22.stage1 (6 convs)", {1, 2, 6, 7, 11, 12});
        add_group("22.stage2 (6 convs)", {3, 4, 8, 9, 13, 14});
    } else if (grouping == 1) {
        for (int l = 0; l < 3; ++l) {
            d->ops.push_back(hops[5 * l]);
            add_group("22.L" + std::to_string(l) + ".1 (cv2+cv3)", {5 * l + 1, 5 * l + 2});
            add_group("22.L" + std::to_string(l) + ".2 (cv2+cv3)", {5 * l + 3, 5 * l + 4});
        }
    } else {
        for (auto &h : hops) d->ops.push_back(h);
    }
    for (auto &op : d->ops) {
        op.B = d->B;
        if (op.kind == OP_CONV && op.name.rfind("22.", 0) == 0) {   // "22.cv2+cv3.<l>.0", "22.cv3.<l>.2": level = next-to-last field
            size_t last = op.name.rfind('.'), prev = op.name.rfind('.', last - 1);
            op.head_level = atoi(op.name.substr(prev + 1, last - prev - 1).c_str());
        }
    }
    d->n_anchors = 0;
    for (int l = 0; l < 3; ++l) d->n_anchors += d->tensors[d->head_t[l]].H * d->tensors[d->head_t[l]].W;
    d->flops_per_frame = 0;
    for (auto &op : d->ops) d->flops_per_frame += op.flops;

    // one arena for every activation; zeroed once (the zero borders are never written again)
    d->arena_bytes = bld.arena_used + 4096;
    // automatic (chains = 0): staged for every multi-frame batch -- measured against the best chain count at 1 / 2 / 4 / 8 /
    // 16 / 32 frames per launch set: +48 / +43 / +30 / +31 / +10 / +3 % throughput with three batches in flight; a single
    // frame run synchronously (detect(), batch 1) keeps the plain engine, whose latency is 7 % lower
    int stages = d->cfg.chains < 0 ? 1 - d->cfg.chains : (d->cfg.chains == 0 && d->B >= 2 ? 2 : 1);      // -1 -> 2 stages, -2 -> 3
    if (rt_opt("CHAINS")) stages = 1;
    if (const char *e = rt_diag("PIPE")) stages = atoi(e) != 0 ? 2 : 1;
    if (const char *e = rt_opt("STAGES")) stages = atoi(e);
    stages = d->cfg.use_graph ? std::max(1, std::min(stages, (int)rtmodt_detector::MAX_STAGES)) : 1;
    {   // stage boundaries by layer name: 2 stages cut after SPPF (52 % / 48 % of the kernel time), 3 stages at 35 % / 66 %
        // (sweeps of both sets of cuts on s @ 640.  Round 5: with the front end ~75 us shorter the three-stage cuts moved from {6.m.1, 16} to {6.cv2, 18.}: +3.2 ... 4.7 % on
        //  two boxes, nine pairs of cuts tried twice each -- profiles/r05/split3/)
        const char *cut2[] = {"12."}, *cut3[] = {"6.cv2", "18."};
        const char **cuts = stages == 3 ? cut3 : cut2;
        static std::string keep[2];                        // experiment hooks: RTMODT_SPLIT=<layer> (2 stages), RTMODT_SPLIT3=<layer>,<layer>
        if (const char *e = rt_diag("SPLIT")) { keep[0] = e; cut2[0] = keep[0].c_str(); }
        if (const char *e = rt_diag("SPLIT3")) {
            const std::string v = e; const size_t c = v.find(',');
            if (c != std::string::npos) { keep[0] = v.substr(0, c); keep[1] = v.substr(c + 1); cut3[0] = keep[0].c_str(); cut3[1] = keep[1].c_str(); }
        }
        d->stage_lo[0] = 0;
        for (int k = 1; k < stages; ++k) {
            d->stage_lo[k] = -1;
            for (size_t i = 0; i < d->ops.size(); ++i)
                if (d->ops[i].name.rfind(cuts[k - 1], 0) == 0) { d->stage_lo[k] = (int)i; break; }
            if (d->stage_lo[k] <= d->stage_lo[k - 1] + 1) { stages = 1; break; }
        }
        d->stage_lo[stages] = (int)d->ops.size();
    }
    d->n_stages = stages;
    d->pipe = stages > 1;
    d->arena_stride = align_up(d->arena_bytes, 4096);
    RT_HIP(hipMalloc((void **)&d->arena, d->arena_stride * d->n_stages));
    RT_HIP(hipMemset(d->arena, 0, d->arena_stride * d->n_stages));
    auto rebase = [&](TensorView &v) { if (v.base || v.c) v.base = (f16 *)(d->arena + (uintptr_t)v.base); };
    for (auto &t : d->tensors) t.ptr = (f16 *)(d->arena + (uintptr_t)t.ptr);
    for (auto &op : d->ops) {
        if (op.kind == OP_CONV) {
            rebase(op.conv.in); rebase(op.conv.out);
            if (op.conv.res.c) rebase(op.conv.res);
            if (op.conv.out2.c) rebase(op.conv.out2);
            if (op.conv.tail_out.c) rebase(op.conv.tail_out);
            if (op.conv.in_lo.c) rebase(op.conv.in_lo);
        } else if (op.kind == OP_GROUP || op.kind == OP_BNECK) {
            for (auto &c : op.group) { rebase(c.in); rebase(c.out); if (c.res.c) rebase(c.res); }
            if (op.kind == OP_BNECK) {
                rebase(op.bneck.in); rebase(op.bneck.out);
                if (op.bneck.res.c) rebase(op.bneck.res);
                if (op.bneck.tail_in.c) { rebase(op.bneck.tail_in); rebase(op.bneck.tail_out); }
            }
        } else {
            for (auto &v : op.v) if (v.c) rebase(v);
        }
    }
    for (auto &kv : d->layer_out) rebase(kv.second);
    for (size_t i = 0; i < d->ops.size(); ++i)
        if (d->ops[i].kind == OP_GROUP && d->ops[i].name.rfind("22.stage2", 0) == 0) d->stage2_op = (int)i;
    if (d->stage2_op >= 0 && !rt_opt("NO_HEAD_FINAL")) {
        const Op &g = d->ops[d->stage2_op];                // members: cv2.l.2, cv3.l.2 for l = 0, 1, 2
        const int cbox_in = g.group[0].cin, ccls_in = g.group[1].cin;
        bool ok = g.group.size() == 6 && head_final_supported(cbox_in, ccls_in, d->nc) && g.group[0].cout == 64;
        const int strides[3] = {8, 16, 32};
        for (int l = 0; ok && l < 3; ++l) {
            const ConvLaunch &cb = g.group[2 * l], &cc = g.group[2 * l + 1];
            ok = cb.in.pad == 0 && cc.in.pad == 0 && cb.in.coff == 0 && cc.in.coff == 0 && cb.in.C == cbox_in && cc.in.C == ccls_in && cb.kp == cbox_in &&
                 cc.kp == ccls_in && cb.act == 0 && cc.act == 0;
            const Tensor &t = d->tensors[d->head_t[l]];
            d->hf.lvl[l] = HeadFinalLevel{cb.in.base, cc.in.base, cb.wt, cc.wt, cb.bias, cc.bias, nullptr, t.H, t.W, strides[l], cdiv(t.H * t.W, 128)};
        }
        if (ok) {
            d->hf.B = d->B; d->hf.nc = d->nc; d->hf.n_anchors = d->n_anchors; d->hf.cbox = cbox_in; d->hf.ccls = ccls_in;
            d->hf.no = d->tensors[d->head_t[0]].C;
            d->head_final = true;
            d->ops[d->stage2_op].skip = true;
        }
    }

    // sub-batch chains (parallel graph branches)
    // measured on MI355X (s @ 640): 2 x 4 frames +4.5 % over 1 x 8, 2 x 8 +7 % over 1 x 16, 2 x 16 +8.6 % over 1 x 32; 4 x 8 -12 % against 1 x 32
    int chains = d->cfg.chains > 0 ? d->cfg.chains : (d->B >= 8 && d->B % 2 == 0 ? 2 : 1);
    if (const char *e = rt_opt("CHAINS")) chains = atoi(e);
    if (d->pipe) chains = 1;                               // the stages run the whole batch
    if (const char *e = rt_opt("CHAIN_JOIN")) d->chain_free_run = atoi(e) == 0;
    chains = std::max(1, std::min(chains, d->B));
    while (d->B % chains) --chains;
    set_chains(d, chains);
    return RTMODT_OK;
}

static int run_op_on(const Op &op, hipStream_t s) {
    switch (op.kind) {
        case OP_STEM: return launch_stem(op.v[0], op.v[1], op.stem_w, op.stem_b, op.B, op.v[1].c, s);
        case OP_CONV: {
            if (op.skip) return RTMODT_OK;  // runs as the tail of the previous launch
            if (!op.tail_on) return launch_conv(op.conv, s);
            ConvLaunch c = op.conv;
            c.tile = op.tail_tile;
            return launch_conv(c, s);
        }
        case OP_GROUP: return op.skip ? RTMODT_OK : launch_conv_group(op.group.data(), (int)op.group.size(), op.group_tile, s);
        case OP_BNECK:
            if (op.fused) {
                if (op.tail_on) return launch_bottleneck(op.bneck, s);       // with C2f.cv2 as its tail
                BottleneckLaunch b = op.bneck;
                b.tail_wt = nullptr;
                return launch_bottleneck(b, s);
            }
            RT_TRY(launch_conv(op.group[0], s));
            return launch_conv(op.group[1], s);
        case OP_POOL: return launch_sppf_pool(op.v[0], op.v[1], op.v[2], op.v[3], op.B, s);
        case OP_UP: return launch_upsample2(op.v[0], op.v[1], op.B, s);
    }
    return RTMODT_OK;
}
// (a failing launch check names the layer it belongs to: "launch_conv: ... [op 8.cv2]")
static int run_op(rtmodt_detector *d, const Op &op) {
    const int rc = run_op_on(op, d->stream);
    if (rc != RTMODT_OK) last_error() += " [op " + op.name + "]";
    return rc;
}

// Detect's tail for images [b0, b0 + nb) on stream `st`
static int run_decode_sub(rtmodt_detector *d, int b0, int nb, hipStream_t st) {
    const rtmodt_detector::Dense &dn = d->dense[d->cur_dense];
    const size_t a0 = (size_t)b0 * d->n_anchors;
    if (d->head_final && !d->want_pred) {                  // the two last convs of Detect and the decode in one launch
        HeadFinalArgs h = d->hf;
        h.conf = d->cfg.conf; h.class_mask[0] = d->class_mask[0]; h.class_mask[1] = d->class_mask[1];
        const size_t par = (size_t)d->run_par * d->arena_stride / sizeof(f16);
        for (int l = 0; l < 3; ++l) {
            const size_t hw = (size_t)h.lvl[l].H * h.lvl[l].W * b0;
            h.lvl[l].xb += hw * h.cbox + par; h.lvl[l].xc += hw * h.ccls + par;
        }
        h.B = nb;
        h.box = dn.box + a0; h.score = dn.score + a0; h.cls = dn.cls + a0;
        return launch_head_final(h, st);
    }
    DecodeArgs a{};
    const int strides[3] = {8, 16, 32};
    for (int l = 0; l < 3; ++l) {
        const Tensor &t = d->tensors[d->head_t[l]];
        a.lvl[l] = HeadLevel{t.ptr + t.per_image * b0 + (size_t)d->run_par * d->arena_stride / sizeof(f16), t.H, t.W, strides[l]};
    }
    a.B = nb; a.nc = d->nc; a.n_anchors = d->n_anchors; a.conf = d->cfg.conf;
    a.no = d->tensors[d->head_t[0]].C;
    a.class_mask[0] = d->class_mask[0]; a.class_mask[1] = d->class_mask[1];
    a.box = dn.box + a0; a.score = dn.score + a0; a.cls = dn.cls + a0;
    a.pred = d->want_pred ? d->d_pred + a0 * (4 + d->nc) : nullptr;
    return launch_decode(a, st);
}
static int run_decode(rtmodt_detector *d) { return run_decode_sub(d, 0, d->B, d->stream); }

// The net's first launch for one op list (whole batch, a sub-batch chain, or an arena copy): the stem -- straight from the
// frames' bytes (`from_bytes`: letterbox folded in) or from the letterboxed image tensor.  frame0: first frame of d->fptrs
// this op list covers.
static int front_launch(rtmodt_detector *d, const Op &op, int frame0, bool from_bytes, hipStream_t st) {
    FrontLaunch f;
    f.zeros = d->d_zeros;
    f.w0 = op.stem_w; f.b0 = op.stem_b; f.w1 = op.front_w1; f.b1 = op.front_b1; f.kp1 = op.front_kp1; f.w2 = op.front_w2; f.b2 = op.front_b2; f.kp2 = op.front_kp2;
    f.out = op.v[2]; f.B = op.B; f.c0 = op.v[1].c; f.c1 = op.front_c1; f.c2 = op.front_c2; f.in_h = d->in_h; f.in_w = d->in_w;
    if (from_bytes) { f.frames = d->fptrs; f.frame0 = frame0; f.pitch = d->last_pitch; f.g = d->last_lg; }
    else { f.from_tensor = true; f.img4 = op.v[0]; }
    return launch_front(f, st);
}
static int run_first(rtmodt_detector *d, const std::vector<Op> &ops, int frame0, bool from_bytes, hipStream_t st) {
    const Op &op = ops[0];
    RT_CHECK(op.kind == OP_STEM, RTMODT_E_INVALID, "op 0 is not the stem");
    if (op.front_on) return front_launch(d, op, frame0, from_bytes, st);
    if (from_bytes) return launch_stem_fused(d->fptrs, frame0, d->last_pitch, d->last_lg, d->in_h, d->in_w, d->lut255, op.v[1], op.stem_w,
                                             op.stem_b, op.B, op.v[1].c, st);
    return run_op_on(op, st);
}
static int run_stem_chain(rtmodt_detector *d, int c, bool fused, hipStream_t st) {
    return run_first(d, d->chain_ops[c], c * d->chain_ops[c][0].B, fused, st);
}
static int run_stems(rtmodt_detector *d, bool fused) {
    for (int c = 0; c < d->n_chains; ++c) RT_TRY(run_stem_chain(d, c, fused, d->stream));
    return RTMODT_OK;
}
// every launch after the stem, eagerly, on the main stream
static int forward_eager(rtmodt_detector *d) {
    for (auto &op : d->ops) if (op.kind != OP_STEM) RT_TRY(run_op(d, op));
    return run_decode(d);
}
// start-up / autotune: the whole net on whatever the image tensor holds
static int forward_eager_all(rtmodt_detector *d) {
    RT_TRY(run_first(d, d->ops, 0, false, d->stream));
    return forward_eager(d);
}

// Times every tile configuration of every MFMA conv on the device it will run on (HIP events,
// best of a few launches) and keeps the fastest: the GEMM shapes of this net are small and
// skinny, so the best tile depends on how M x N fills 256 CUs, not on a rule.
// best-of-3 time (ms per launch) of `launch` issued 4 times back to back on the detector's stream
// best-of-5 time (ms per launch) of `launch` issued 4 times back to back on the detector's stream (with 3 repetitions
// near-ties between tiles flipped from run to run and the bench moved by +-1 %)
template <typename F>
static int time_launch(rtmodt_detector *d, hipEvent_t e0, hipEvent_t e1, F &&launch, float &ms_out) {
    for (int w = 0; w < 2; ++w) RT_TRY(launch());
    float ms_min = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        RT_HIP(hipEventRecord(e0, d->stream));
        for (int k = 0; k < 4; ++k) RT_TRY(launch());
        RT_HIP(hipEventRecord(e1, d->stream));
        RT_HIP(hipEventSynchronize(e1));
        float ms = 0;
        RT_HIP(hipEventElapsedTime(&ms, e0, e1));
        ms_min = std::min(ms_min, ms);
    }
    ms_out = ms_min * 0.25f;
    return RTMODT_OK;
}

// may tile `t` run these n convs as one launch?  (the tuner's candidate filter; also applied to every cache hit)
static bool tile_legal(const ConvLaunch *c, int n, int t) {
    if (t < 0 || t >= TILE_COUNT || tile_is_tail(t)) return false;              // tail tiles only through tune_tails()
    bool cin64 = true, rows_ok = true;
    for (int i = 0; i < n; ++i) {
        cin64 = cin64 && c[i].cin % 64 == 0 && c[i].kp % 64 == 0;
        rows_ok = rows_ok && c[i].ks == 3 && c[i].stride == 1 && c[i].in.pad == 1 && c[i].cin % 32 == 0;
    }
    if (tile_needs_cin64(t) && !cin64) return false;
    if (tile_is_rows(t) && !rows_ok) return false;
    if (c[0].in_lo.base && !tile_reads_lo(t)) return false;
    if (tile_is_w8(t) && n != 1) return false;    // the 8-wave tiles have no group entry point
    for (int i = 0; i < n; ++i)                   // weights and bias are padded to 128 rows of cout: a wider tile must divide cout
        if (tile_shape(t).bn > 128 && c[i].cout % tile_shape(t).bn != 0) return false;
    if (tile_is_pt(t) && (n != 1 || c[0].out2.base || ((long)c[0].B * c[0].out.H * c[0].out.W) % tile_shape(t).bm != 0 ||
                          c[0].cout % tile_shape(t).bn != 0 || c[0].out.coff % 8 != 0 || c[0].out.C % 8 != 0)) return false;
    if (tile_is_ppt(t) && (n != 1 || c[0].out2.base || c[0].res.base || c[0].kp != c[0].ks * c[0].ks * c[0].cin || c[0].ks * c[0].ks * (c[0].cin / 64) < 3 ||
                           c[0].cout % 8 != 0 || c[0].out.coff % 8 != 0 || c[0].out.C % 8 != 0)) return false;
    if (tile_is_pp(t) || tile_is_ppt(t))         // 24-bit index arithmetic in their epilogues: larger tensors keep the other tiles (YOLOv8l / x above ~900 pixels)
        for (int i = 0; i < n; ++i)
            if (!conv_pp_index_fits(c[i])) return false;
    // the ping-pong 3x3 kernel stores (and reads the shortcut) 16 bytes per lane, has no second destination, and its 192-wide form no shortcut
    if (tile_is_pp(t))
        for (int i = 0; i < n; ++i)
            if (c[i].out2.base || c[i].in_lo.base || c[i].cout % 8 != 0 || c[i].out.coff % 8 != 0 || c[i].out.C % 8 != 0 ||
                (c[i].res.base && (c[i].res.coff % 8 != 0 || c[i].res.C % 8 != 0 || tile_shape(t).bn > 128))) return false;
    return true;
}
static bool tail_tile_legal(const ConvLaunch &c, int t) {
    return tile_is_tail(t) && tile_shape(t).bn == c.cout && !(tile_needs_cin64(t) && (c.cin % 64 != 0 || c.kp % 64 != 0)) && c.cin % 32 == 0;
}

// LDS bytes a workgroup of tile `t` holds (conv.hip's stage rings): two workgroups of DIFFERENT launches share a CU only if
// their LDS fits 160 KiB together -- which is what lets the stages of the staged engine overlap
static int tile_lds_kib(int t) {
    const TileShape ts = tile_shape(t);
    if (tile_is_ppt(t)) return 3 * (ts.bm / 8 + ts.bn / 8);                  // three-slot rings for both operands
    if (tile_is_pp(t)) return 2 * (ts.bm / 8 + 1) + 3 * (ts.bn / 8);      // two strip slots + three per-tap weight slots (conv_pp.hip)
    if (tile_is_rows(t)) { const int rp = tile_needs_cin64(t) ? 8 : 16; return 2 * (ts.bm / rp + 1 + 3 * (ts.bn / rp)); }
    if (t >= TILE_WSK_64x64 && t <= TILE_WSK_64x32) return std::max(8 * (ts.bm / 16 + ts.bn / 16), 4 * (ts.bm / 16) * (ts.bn / 16));
    if (tile_is_pt(t)) return 2 * (ts.bm / 8 + ts.bn / 8);
    const int stages = (t == TILE_K64_128x128_S2_W8 || t == TILE_K64_256x64_S2_W8) ? 2 : 3;
    const int rp = tile_needs_cin64(t) ? 8 : 16;
    return stages * (ts.bm / rp + ts.bn / rp);
}

// fastest tile for one conv (or one group of convs sharing a tile); returns its time
static int tune_conv(rtmodt_detector *d, hipEvent_t e0, hipEvent_t e1, const std::string &name, ConvLaunch *c, int n, int &tile_io,
                     float &best_ms) {
    best_ms = 1e30f;
    int best_tile = tile_io;
    for (int t = 0; t < TILE_COUNT; ++t) {
        if (!tile_legal(c, n, t)) continue;
        // A/B hook: RTMODT_TUNE_SKIP="22,23" keeps the listed tile ids out of the tuner (same-box comparisons of a tile family)
        static const std::string skip = rt_diag("TUNE_SKIP") ? std::string(",") + rt_diag("TUNE_SKIP") + "," : std::string();
        if (!skip.empty() && skip.find("," + std::to_string(t) + ",") != std::string::npos) continue;
        float ms;
        {   // a tile whose launch check refuses this shape is "not a candidate", not a failed create; anything else is an error
            const int trc = time_launch(d, e0, e1, [&]() { return launch_conv_group(c, n, t, d->stream); }, ms);
            if (trc == RTMODT_E_INVALID || trc == RTMODT_E_UNSUPPORTED) {
                if (rt_opt("TUNE_LOG")) fprintf(stderr, "[tune] %-28s %-16s refused: %s\n", name.c_str(), tile_name(t), last_error().c_str());
                continue;
            }
            RT_TRY(trc);
        }
        if (rt_opt("TUNE_LOG")) fprintf(stderr, "[tune] %-28s %-16s %8.2f us  (%d KiB LDS)\n", name.c_str(), tile_name(t), ms * 1e3f, tile_lds_kib(t));
        // the launches are timed ALONE, but in the staged engine they share the CUs with the other stages' launches: a tile
        // whose workgroup takes more than half the LDS keeps every other workgroup off its CU (experiment hook)
        static const float lds_penalty = rt_diag("TUNE_LDS_PENALTY") ? (float)atof(rt_diag("TUNE_LDS_PENALTY")) : 0.f;
        static const int lds_cap = rt_diag("TUNE_LDS_CAP") ? atoi(rt_diag("TUNE_LDS_CAP")) : 80;
        if (d->pipe && tile_lds_kib(t) > lds_cap) ms *= 1.f + lds_penalty;
        if (ms < best_ms) { best_ms = ms; best_tile = t; }
    }
    tile_io = best_tile;
    return RTMODT_OK;
}

// RTMODT_TUNE_CACHE=<file>: the tuner's decisions, one line per launch ("<key>\t<tile> <tile2> <fused>"), keyed by the
// launch's name and GEMM shape.  A process that finds its key there skips the timing runs -- start-up drops from
// seconds to milliseconds, and a profiler run replays exactly the configuration the bench measured.
static std::string tune_key(const rtmodt_detector *d, const Op &op) {
    char buf[256];
    const ConvLaunch &c = op.kind == OP_CONV ? op.conv : op.group[0];
    snprintf(buf, sizeof(buf), "%s|B%d|%dx%d|%d>%d|k%ds%d|n%zu", op.name.c_str(), op.B, c.in.H, c.in.W, c.cin, c.cout, c.ks, c.stride,
             op.kind == OP_CONV ? (size_t)1 : op.group.size());
    for (char *p = buf; *p; ++p) if (*p == ' ' || *p == '\t') *p = '_';
    (void)d;
    return buf;
}
struct TuneRec { int t0 = 0, t1 = 0, fused = 0; };
// first line of a cache file: the tile table it was written for (count + a hash of the tile names in id order); a file
// written by a build with another table is ignored, so an id can never silently select a different kernel
static std::string tune_cache_header() {
    uint64_t h = 1469598103934665603ull;
    for (int t = 0; t < TILE_COUNT; ++t)
        for (const char *p = tile_name(t); ; ++p) { h = (h ^ (unsigned char)*p) * 1099511628211ull; if (!*p) break; }
    char buf[96];
    snprintf(buf, sizeof(buf), "#rtmodt-tune tiles=%d table=%016llx", TILE_COUNT, (unsigned long long)h);
    return buf;
}
static std::map<std::string, TuneRec> tune_cache_read(const char *path) {
    std::map<std::string, TuneRec> m;
    std::ifstream f(path);
    std::string key;
    TuneRec r;
    std::string first;
    if (!std::getline(f, first) || first != tune_cache_header()) return m;
    while (f >> key >> r.t0 >> r.t1 >> r.fused)
        if (r.t0 >= 0 && r.t0 < TILE_COUNT && r.t1 >= 0 && r.t1 < TILE_COUNT) m[key] = r;
    return m;
}

static int autotune_ops(rtmodt_detector *d, std::vector<Op> &ops) {
    hipEvent_t e0, e1;
    RT_HIP(hipEventCreate(&e0)); RT_HIP(hipEventCreate(&e1));
    const char *cache_path = rt_opt("TUNE_CACHE");
    std::map<std::string, TuneRec> cache;
    if (cache_path) cache = tune_cache_read(cache_path);
    bool dirty = false;
    for (auto &op : ops) {
        if (op.kind != OP_CONV && op.kind != OP_GROUP && op.kind != OP_BNECK) continue;
        const std::string key = tune_key(d, op);
        auto hit = cache.find(key);
        if (hit != cache.end()) {                          // a hit must pass the same legality checks as a tuning candidate
            const TuneRec &r = hit->second;
            bool ok;
            if (op.kind == OP_CONV) ok = tile_legal(&op.conv, 1, r.t0) && (!op.conv.tail_wt || r.t1 == 0 || !r.fused || tail_tile_legal(op.conv, r.t1));
            else if (op.kind == OP_GROUP) ok = tile_legal(op.group.data(), (int)op.group.size(), r.t0);
            else ok = tile_legal(&op.group[0], 1, r.t0) && tile_legal(&op.group[1], 1, r.t1);
            if (!ok) { cache.erase(hit); hit = cache.end(); }
        }
        if (hit != cache.end()) {
            const TuneRec &r = hit->second;
            if (op.kind == OP_CONV) { op.conv.tile = r.t0; if (op.conv.tail_wt) { op.tail_tile = r.t1; op.tail_on = r.fused != 0; } }
            else if (op.kind == OP_GROUP) op.group_tile = r.t0;
            else { op.group[0].tile = r.t0; op.group[1].tile = r.t1; op.fused = r.fused != 0; op.tail_on = (r.fused == 2 || r.fused == 4) && op.bneck.tail_wt; }
            continue;
        }
        float ms;
        TuneRec r;
        if (op.kind == OP_CONV) {
            RT_TRY(tune_conv(d, e0, e1, op.name, &op.conv, 1, op.conv.tile, ms));
            r.t0 = op.conv.tile;
        } else if (op.kind == OP_GROUP) {
            RT_TRY(tune_conv(d, e0, e1, op.name, op.group.data(), (int)op.group.size(), op.group_tile, ms));
            r.t0 = op.group_tile;
        } else {
            float ms1, ms2, msf;
            RT_TRY(tune_conv(d, e0, e1, op.group[0].in.c ? op.name + ".cv1" : op.name, &op.group[0], 1, op.group[0].tile, ms1));
            RT_TRY(tune_conv(d, e0, e1, op.name + ".cv2", &op.group[1], 1, op.group[1].tile, ms2));
            BottleneckLaunch plain = op.bneck;
            plain.tail_wt = nullptr;
            RT_TRY(time_launch(d, e0, e1, [&]() { return launch_bottleneck(plain, d->stream); }, msf));
            if (rt_opt("TUNE_LOG")) fprintf(stderr, "[tune] %-28s fused %8.2f us vs two launches %8.2f us\n", op.name.c_str(), msf * 1e3f, (ms1 + ms2) * 1e3f);
            op.fused = msf < ms1 + ms2;
            if (const char *e = rt_opt("BNECK")) op.fused = atoi(e) != 0;      // A/B and test hook
            r.t0 = op.group[0].tile; r.t1 = op.group[1].tile; r.fused = op.fused;
        }
        cache[key] = r;
        dirty = true;
    }
    // conv -> 1x1 pairs: the fused launch (every legal tail tile) against the two tuned launches
    for (size_t i = 0; i + 1 < ops.size(); ++i) {
        Op &op = ops[i], &nx = ops[i + 1];
        if (op.kind != OP_CONV || !op.conv.tail_wt) continue;
        const std::string key = tune_key(d, op);
        auto hit = cache.find(key);
        const bool cached = hit != cache.end() && hit->second.t1 != 0;      // t1 == 0 (TILE_128x128) is never a tail tile: "not decided yet"
        if (!cached) {
            float ms_a, ms_b, best = 1e30f;
            RT_TRY(time_launch(d, e0, e1, [&]() { return launch_conv(op.conv, d->stream); }, ms_a));
            RT_TRY(time_launch(d, e0, e1, [&]() { return launch_conv(nx.conv, d->stream); }, ms_b));
            for (int t = TILE_TAIL_128x64; t <= TILE_TAIL_K64_128x128; ++t) {
                if (tile_shape(t).bn != op.conv.cout || (tile_needs_cin64(t) && (op.conv.cin % 64 != 0 || op.conv.kp % 64 != 0)) || op.conv.cin % 32 != 0) continue;
                ConvLaunch c = op.conv;
                c.tile = t;
                float ms;
                RT_TRY(time_launch(d, e0, e1, [&]() { return launch_conv(c, d->stream); }, ms));
                if (rt_opt("TUNE_LOG")) fprintf(stderr, "[tune] %-28s %-16s %8.2f us (+ %s)\n", op.name.c_str(), tile_name(t), ms * 1e3f, nx.name.c_str());
                if (ms < best) { best = ms; op.tail_tile = t; }
            }
            op.tail_on = best < ms_a + ms_b;
            if (rt_opt("TUNE_LOG")) fprintf(stderr, "[tune] %-28s with tail %8.2f us vs two launches %8.2f us\n", op.name.c_str(), best * 1e3f, (ms_a + ms_b) * 1e3f);
            TuneRec r; r.t0 = op.conv.tile; r.t1 = op.tail_tile; r.fused = op.tail_on;
            cache[key] = r;
            dirty = true;
        }
        if (const char *e = rt_opt("TAIL")) op.tail_on = atoi(e) != 0 && tile_shape(op.tail_tile).bn == op.conv.cout;   // A/B and test hook
        nx.skip = op.tail_on;
    }
    // fused Bottleneck + C2f.cv2 as its tail against the fused Bottleneck followed by cv2's own launch
    for (size_t i = 0; i + 1 < ops.size(); ++i) {
        Op &op = ops[i], &nx = ops[i + 1];
        if (op.kind != OP_BNECK || !op.bneck.tail_wt) continue;
        const std::string key = tune_key(d, op);
        auto hit = cache.find(key);
        const bool decided = hit != cache.end() && hit->second.fused >= 2;      // 2 / 4 = fused with tail, 3 = tail timed and rejected
        if (decided && hit->second.fused == 4) op.bneck.persistent32 = 0;
        if (!decided && op.fused) {
            float ms_plain, ms_cv2, ms_tail;
            BottleneckLaunch plain = op.bneck;
            plain.tail_wt = nullptr;
            RT_TRY(time_launch(d, e0, e1, [&]() { return launch_bottleneck(plain, d->stream); }, ms_plain));
            RT_TRY(time_launch(d, e0, e1, [&]() { return launch_conv(nx.conv, d->stream); }, ms_cv2));
            RT_TRY(time_launch(d, e0, e1, [&]() { return launch_bottleneck(op.bneck, d->stream); }, ms_tail));
            if (op.bneck.c == 32 && !rt_opt("BNECK32") && bottleneck32_tail_supported(op.bneck)) {      // two kernels for this shape: bneck32.hip's persistent form and bottleneck_fused
                BottleneckLaunch other = op.bneck;
                other.persistent32 = !op.bneck.persistent32;
                float ms_other;
                RT_TRY(time_launch(d, e0, e1, [&]() { return launch_bottleneck(other, d->stream); }, ms_other));
                if (rt_opt("TUNE_LOG")) fprintf(stderr, "[tune] %-28s with cv2 tail: persistent %8.2f us vs bottleneck_fused %8.2f us\n", op.name.c_str(), (op.bneck.persistent32 ? ms_tail : ms_other) * 1e3f, (op.bneck.persistent32 ? ms_other : ms_tail) * 1e3f);
                if (ms_other < ms_tail) { op.bneck.persistent32 = other.persistent32; ms_tail = ms_other; }
            }
            op.tail_on = ms_tail < ms_plain + ms_cv2;
            if (rt_opt("TUNE_LOG")) fprintf(stderr, "[tune] %-28s with cv2 tail %8.2f us vs fused + cv2 %8.2f us\n", op.name.c_str(), ms_tail * 1e3f, (ms_plain + ms_cv2) * 1e3f);
            TuneRec r; r.t0 = op.group[0].tile; r.t1 = op.group[1].tile; r.fused = op.tail_on ? (op.bneck.persistent32 ? 2 : 4) : 3;      // 2 / 4: with the tail on bneck32.hip's kernel / on bottleneck_fused
            cache[key] = r;
            dirty = true;
        }
        if (const char *e = rt_opt("BNECK_TAIL")) op.tail_on = atoi(e) != 0;      // A/B and test hook
        op.tail_on = op.tail_on && op.fused;
        nx.skip = op.tail_on;
    }
    // the fused front end (stem -> layer 1 -> 2.cv1 in one launch) against the launches it replaces, both reading the letterboxed image tensor (what
    // the tuner's eager passes run on; the byte source differs by the same letterbox work on either side)
    if (ops.size() >= 3 && ops[0].kind == OP_STEM && ops[0].front_ok) {
        Op &st = ops[0], &l1 = ops[1], &cv = ops[2];
        const std::string key = "front|" + tune_key(d, l1);
        auto hit = cache.find(key);
        if (hit != cache.end()) st.front_on = hit->second.fused != 0;
        else {
            float ms_old, ms_new;
            const bool keep1 = l1.skip, keep2 = cv.skip;
            l1.skip = false; cv.skip = l1.tail_on;         // (the pair as the tuner just left it)
            RT_TRY(time_launch(d, e0, e1, [&]() { RT_TRY(launch_stem(st.v[0], st.v[1], st.stem_w, st.stem_b, st.B, st.v[1].c, d->stream)); RT_TRY(run_op_on(l1, d->stream)); return run_op_on(cv, d->stream); }, ms_old));
            RT_TRY(time_launch(d, e0, e1, [&]() { return front_launch(d, st, 0, false, d->stream); }, ms_new));
            l1.skip = keep1; cv.skip = keep2;
            st.front_on = ms_new < ms_old;
            if (rt_opt("TUNE_LOG")) fprintf(stderr, "[tune] %-28s fused front end %8.2f us vs stem + layer 1 (+ 2.cv1) %8.2f us\n", "0 + 1 + 2.cv1", ms_new * 1e3f, ms_old * 1e3f);
            TuneRec r; r.fused = st.front_on;
            cache[key] = r;
            dirty = true;
        }
        if (const char *e = rt_opt("FRONT")) st.front_on = atoi(e) != 0;      // A/B and test hook
        if (st.front_on) l1.skip = cv.skip = true;
        else { l1.skip = false; cv.skip = l1.tail_on; }
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (cache_path && dirty) {                            // whole file rewritten through a rename: readers never see half a file
        const std::string tmp = std::string(cache_path) + ".tmp." + std::to_string((long)getpid());
        {
            std::ofstream f(tmp);
            f << tune_cache_header() << '\n';
            for (auto &kv : cache) f << kv.first << '\t' << kv.second.t0 << ' ' << kv.second.t1 << ' ' << kv.second.fused << '\n';
        }
        rename(tmp.c_str(), cache_path);
    }
    return RTMODT_OK;
}

static int autotune_tiles(rtmodt_detector *d) {
    RT_TRY(autotune_ops(d, d->ops));
    // the tuner's decisions travel from the op it timed to the copies that run
    auto adopt = [](Op &dst, const Op &src) {
        dst.conv.tile = src.conv.tile;
        dst.group_tile = src.group_tile;
        dst.fused = src.fused;
        dst.tail_on = src.tail_on; dst.tail_tile = src.tail_tile; dst.skip = src.skip; dst.front_on = src.front_on; dst.bneck.persistent32 = src.bneck.persistent32;
        for (size_t g = 0; g < dst.group.size(); ++g) dst.group[g].tile = src.group[g].tile;
    };
    if (d->n_chains > 1) {
        RT_TRY(autotune_ops(d, d->chain_ops[0]));          // the sub-batch GEMMs have their own best tiles
        for (int c = 1; c < d->n_chains; ++c)
            for (size_t i = 0; i < d->ops.size(); ++i) adopt(d->chain_ops[c][i], d->chain_ops[0][i]);
    } else {
        for (size_t i = 0; i < d->ops.size(); ++i) adopt(d->chain_ops[0][i], d->ops[i]);
    }
    // the tuning launches left stale activations; run one clean pass
    RT_TRY(forward_eager_all(d));
    RT_HIP(hipStreamSynchronize(d->stream));
    return RTMODT_OK;
}

// One hipGraph per sub-batch chain.  Inside a chain's graph the Detect branch of level 0 forks
// off as soon as layer 15 is done and that of level 1 after layer 18, so they run beside the
// (small, latency-bound) P4/P5 neck.  Forks/joins are only ever made against the capture's
// ORIGIN stream: HIP 7.0's EndCapture recurses forever when two non-origin streams wait on
// each other, which is why the chains are separate graphs launched on separate streams
// instead of branches of one capture.
static int capture_chain(rtmodt_detector *d, int c) {
    hipStream_t main = d->chain_streams[c];
    hipStream_t h0 = d->aux_streams[2 * c], h1 = d->aux_streams[2 * c + 1];
    const auto &ops = d->chain_ops[c];
    size_t ev = (size_t)c * 4;
    auto fork = [&](hipStream_t from, hipStream_t to) -> int {
        hipEvent_t e = d->aux_events[ev++];
        RT_HIP(hipEventRecord(e, from));
        RT_HIP(hipStreamWaitEvent(to, e, 0));
        return RTMODT_OK;
    };
    auto run_head = [&](int level, hipStream_t st) -> int {
        for (auto &op : ops) if (op.head_level == level) RT_TRY(run_op_on(op, st));
        return RTMODT_OK;
    };
    auto has_level = [&](int level) { for (auto &op : ops) if (op.head_level == level) return true; return false; };
    bool forked0 = false, forked1 = false;
    auto body = [&]() -> int {
        for (auto &op : ops) {
            if (op.head_level >= 0 || op.kind == OP_STEM) continue;   // the stem is launched by enqueue_batch (fresh frame pointers)
            RT_TRY(run_op_on(op, main));
            if (op.name == "15.cv2" && has_level(0)) { RT_TRY(fork(main, h0)); RT_TRY(run_head(0, h0)); forked0 = true; }
            if (op.name == "18.cv2" && has_level(1)) { RT_TRY(fork(main, h1)); RT_TRY(run_head(1, h1)); forked1 = true; }
        }
        RT_TRY(run_head(2, main));
        if (forked0) RT_TRY(fork(h0, main));               // joins
        if (forked1) RT_TRY(fork(h1, main));
        return RTMODT_OK;
    };
    RT_HIP(hipStreamBeginCapture(main, hipStreamCaptureModeRelaxed));
    int rc = body();
    hipError_t e = hipStreamEndCapture(main, &d->graphs[c]);
    RT_TRY(rc);
    RT_HIP(e);
    // N_EXEC executable instances per graph, alternating between successive batches: relaunching an instance whose previous launch is still
    // running makes the runtime wait for it before submitting.
    // (rocprofv3 --kernel-trace segfaulting inside hipGraphLaunch on this engine after 200-300 launches is NOT a property of the engine: a
    // stand-alone program that launches one captured graph of 45 spinning kernels 1 000 times dies with the same stack, and with the HIP
    // runtime's AQL packet capture for graphs switched off -- DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 -- the engine's 100-step trace completes:
    // tools/probes/graph_trace_repro.hip, profiles/r04/graph_trace_repro/.  Round 4 first blamed in-flight relaunches and rotated four
    // instances; the collection run crashed all the same.)
    for (int k = 0; k < N_EXEC; ++k) RT_HIP(hipGraphInstantiate(&d->graph_execs[N_EXEC * c + k], d->graphs[c], nullptr, nullptr, 0));
    return RTMODT_OK;
}

// One stream per chain (chain 0 runs on the main stream).  The runtime maps streams onto a few hardware queues in creation
// order (4 per process here: with one, two or five streams created in front of it, chain 1's stream shared the main or the
// post-processing stream's queue and the bench lost 20-30 %), and kernels of streams that share a queue run back to back.
// What else the process has created (torch, RCCL, other detectors) is not ours to know, so every candidate stream is
// PROBED: two 40 us spin kernels, one on the candidate and one on each stream it must overlap with, have to finish in
// the time of one.  Candidates that fail are destroyed at the end; if none passes, the detector falls back to one chain.
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();                  // 100 MHz constant clock: the loop ends after `ticks` whatever else happens
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
static int streams_overlap(hipStream_t a, hipStream_t b, hipEvent_t e0, hipEvent_t e1, hipEvent_t ej, bool &overlap, float &ms_out) {
    const long long ticks = 4000;                         // 40 us
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        RT_HIP(hipEventRecord(e0, a));
        RT_HIP(hipStreamWaitEvent(b, e0, 0));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, ticks);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, ticks);
        RT_HIP(hipEventRecord(ej, b));
        RT_HIP(hipStreamWaitEvent(a, ej, 0));
        RT_HIP(hipEventRecord(e1, a));
        RT_HIP(hipEventSynchronize(e1));
        float ms = 0;
        RT_HIP(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    ms_out = best;
    overlap = best < 0.082f;                              // measured: 62-68 us when the two spins overlap, 102 us when they queue up
    return RTMODT_OK;
}
static int ensure_chain_streams(rtmodt_detector *d) {
    const int want = d->pipe ? d->n_stages : d->n_chains;  // staged mode: the later stages' streams are found like a chain's
    if ((int)d->chain_streams.size() >= want || d->stage_stream[0]) return RTMODT_OK;
    hipEvent_t e0, e1, ej;
    RT_HIP(hipEventCreate(&e0)); RT_HIP(hipEventCreate(&e1)); RT_HIP(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    if (const char *e = rt_opt("PAD_STREAMS"))      // test hook: streams created ahead of ours shift the queue mapping
        for (int i = 0; i < atoi(e); ++i) { hipStream_t st; RT_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); d->pad_streams.push_back(st); }
    std::vector<hipStream_t> rejected;
    int rc = RTMODT_OK;
    while (rc == RTMODT_OK && (int)d->chain_streams.size() < want) {
        hipStream_t st = d->stream;
        if (!d->chain_streams.empty()) {
            st = nullptr;
            for (int attempt = 0; attempt < 12 && !st && rc == RTMODT_OK; ++attempt) {
                hipStream_t cand;
                RT_HIP(hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
                bool ok = true;
                const char *pe = rt_opt("CHAIN_PROBE");      // 0: trust the creation order (counter-collecting profilers serialise every kernel, the probe would see no overlap)
                if (pe && atoi(pe) == 0) { st = cand; break; }
                std::vector<hipStream_t> others = d->chain_streams;
                others.push_back(d->post_stream);
                // the copy stream's event waits would hold up a chain that shared its queue; only the third stage of the staged
                // engine may (main, two more stages and post-processing are all four queues this runtime gives a process)
                if (!(d->pipe && d->chain_streams.size() == 2)) others.push_back(d->copy_stream);
                for (hipStream_t o : others) {
                    bool ov = false; float ms = 0;
                    rc = streams_overlap(o, cand, e0, e1, ej, ov, ms);
                    if (rt_opt("TUNE_LOG")) fprintf(stderr, "[streams] candidate %d for chain %zu against stream %p: %.1f us -> %s\n", attempt, d->chain_streams.size(), (void *)o, ms * 1e3f, ov ? "own queue" : "shared queue");
                    ok = ok && ov && rc == RTMODT_OK;
                    if (!ok) break;
                }
                if (ok) st = cand; else rejected.push_back(cand);
            }
            if (!st) break;                               // no stream of this process runs beside ours
        }
        d->chain_streams.push_back(st);
        hipEvent_t a, b;
        RT_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming)); RT_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        d->chain_fork.push_back(a); d->chain_join.push_back(b);
    }
    for (hipStream_t st : rejected) hipStreamDestroy(st);
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(ej);
    RT_TRY(rc);
    if (d->pipe) {                                         // fewer stages when fewer queues: 3 -> 2 needs new boundaries, so -> plain
        if ((int)d->chain_streams.size() == d->n_stages) {
            for (int k = 0; k < d->n_stages; ++k) d->stage_stream[k] = d->chain_streams[k];
        } else {
            for (size_t k = 1; k < d->chain_streams.size(); ++k) hipStreamDestroy(d->chain_streams[k]);
            d->pipe = false; d->n_stages = 1;              // (the extra arena copies stay unused)
        }
        for (size_t k = 1; k < d->chain_fork.size(); ++k) { hipEventDestroy(d->chain_fork[k]); hipEventDestroy(d->chain_join[k]); }
        d->chain_streams.resize(1); d->chain_fork.resize(1); d->chain_join.resize(1);
        return RTMODT_OK;
    }
    if ((int)d->chain_streams.size() < d->n_chains) {      // fewer queues than chains: as many chains as found streams (dividing the batch)
        int c = (int)d->chain_streams.size();
        while (d->B % c) --c;
        for (size_t k = c; k < d->chain_streams.size(); ++k) hipStreamDestroy(d->chain_streams[k]);
        d->chain_streams.resize(c);
        set_chains(d, c);
    }
    return RTMODT_OK;
}

// staged mode: per arena copy, one graph for the front stage (captured on the main stream) and one for the back stage
static int capture_pipe(rtmodt_detector *d) {
    for (int par = 0; par < d->n_stages; ++par) {
        d->par_ops[par].clear();
        for (auto &op : d->chain_ops[0]) d->par_ops[par].push_back(shift_arena(op, par * d->arena_stride));
        for (int stage = 0; stage < d->n_stages; ++stage) {
            for (int k = 0; k < 2; ++k) if (d->pipe_exec[par][stage][k]) { hipGraphExecDestroy(d->pipe_exec[par][stage][k]); d->pipe_exec[par][stage][k] = nullptr; }
            if (d->pipe_graph[par][stage]) hipGraphDestroy(d->pipe_graph[par][stage]);
            hipStream_t st = d->stage_stream[stage];
            RT_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            int rc = RTMODT_OK;
            for (int i = d->stage_lo[stage]; i < d->stage_lo[stage + 1] && rc == RTMODT_OK; ++i)
                if (d->par_ops[par][i].kind != OP_STEM) rc = run_op_on(d->par_ops[par][i], st);      // the stem is launched by enqueue_batch (fresh frame pointers)
            hipError_t e = hipStreamEndCapture(st, &d->pipe_graph[par][stage]);
            RT_TRY(rc);
            RT_HIP(e);
            for (int k = 0; k < 2; ++k) RT_HIP(hipGraphInstantiate(&d->pipe_exec[par][stage][k], d->pipe_graph[par][stage], nullptr, nullptr, 0));
        }
    }
    return RTMODT_OK;
}

static int capture_graph(rtmodt_detector *d) {
    if (d->pipe) return capture_pipe(d);
    const int C = d->n_chains;
    for (auto g : d->graph_execs) if (g) hipGraphExecDestroy(g);
    for (auto g : d->graphs) if (g) hipGraphDestroy(g);
    d->graphs.assign(C, nullptr); d->graph_execs.assign(N_EXEC * C, nullptr);
    RT_TRY(ensure_chain_streams(d));
    bool forks = false;
    for (auto &op : d->ops) forks = forks || op.head_level >= 0;
    while ((int)d->aux_streams.size() < 2 * C) {
        hipStream_t st = nullptr;
        if (forks) RT_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        d->aux_streams.push_back(st);
    }
    while ((int)d->aux_events.size() < 4 * C) {
        hipEvent_t e;
        RT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        d->aux_events.push_back(e);
    }
    for (int c = 0; c < C; ++c) RT_TRY(capture_chain(d, c));
    return RTMODT_OK;
}

// stem output -> dense per-anchor candidates of a single-chain detector, on the main stream
static int forward_graphs(rtmodt_detector *d) {
    const int inst = (int)(d->batch_no % N_EXEC);   // successive batches, successive instances
    RT_HIP(hipGraphLaunch(d->graph_execs[inst], d->stream));
    return run_decode(d);
}

// Two (or more) sub-batch chains, each on its OWN stream from the stem to the decoded candidates: stem -> graph ->
// Detect's tail of images [c * nb, (c + 1) * nb).  Nothing orders one chain against another -- chain c of batch t + 1
// queues behind chain c of batch t only -- so the kernels of one chain fill the CUs the other's launch tails leave
// idle (measured on MI355X, YOLOv8s @ 640, 16 frames: two chains of 8 run 5-8 % faster than one of 16).  The
// post-processing stream waits for every chain of the slot.  `join_main`: the main stream waits for the other chains too
// (needed when the next batch's whole-batch letterbox would overwrite an image tensor a lagging chain still reads).
static int forward_chains(rtmodt_detector *d, rtmodt_detector::Slot &sl, bool host_frames, bool join_main) {
    const int C = d->n_chains, nb = d->B / C;
    const int inst = (int)(d->batch_no % N_EXEC);
    for (int c = 0; c < C; ++c) {
        hipStream_t st = d->chain_streams[c];
        if (c > 0) {
            if (host_frames) RT_HIP(hipStreamWaitEvent(st, sl.copied, 0));
            if (!d->last_fused) RT_HIP(hipStreamWaitEvent(st, sl.evp, 0));           // the whole-batch letterbox on the main stream
        }
        RT_TRY(run_stem_chain(d, c, d->last_fused, st));
        if (c == 0 && d->last_fused) RT_HIP(hipEventRecord(sl.evp, d->stream));
        RT_HIP(hipGraphLaunch(d->graph_execs[N_EXEC * c + inst], st));
        RT_TRY(run_decode_sub(d, c * nb, nb, st));
        if (c > 0) RT_HIP(hipEventRecord(sl.chain_done[c], st));
    }
    if (join_main) for (int c = 1; c < C; ++c) RT_HIP(hipStreamWaitEvent(d->stream, sl.chain_done[c], 0));
    return RTMODT_OK;
}

static int run_nms(rtmodt_detector *d, const LbHost &g, int h, int w, rtmodt_detector::Slot &sl) {
    NmsArgs a{};
    a.B = d->B; a.n_anchors = d->n_anchors; a.max_det = d->cfg.max_det; a.agnostic = d->cfg.agnostic; a.iou = d->cfg.iou;
    const rtmodt_detector::Dense &dn = d->dense[d->cur_dense];
    a.box = dn.box; a.score = dn.score; a.cls = dn.cls;
    a.keys = d->d_keys; a.sbox = d->d_sbox; a.sidx = d->d_sidx;
    a.gain = (float)g.gain; a.pad_x = (float)g.pad_x; a.pad_y = (float)g.pad_y; a.src_w = (float)w; a.src_h = (float)h; a.rescale = 1;
    a.out_xyxy = sl.o_xyxy; a.out_conf = sl.o_conf; a.out_cls = sl.o_cls; a.out_anchor = sl.o_anchor; a.out_n = sl.o_n;
    return launch_nms(a, d->nms_plan, d->post_stream);
}

int detector_outputs(rtmodt_detector *d, DetOutputs *o) {
    RT_CHECK(d && o, RTMODT_E_INVALID, "null argument");
    RT_CHECK(d->newest >= 0, RTMODT_E_INVALID, "detector has no enqueued batch");
    const rtmodt_detector::Slot &sl = d->slots[d->newest];
    o->box = (const float4 *)sl.o_xyxy; o->conf = sl.o_conf; o->cls = sl.o_cls; o->n = sl.o_n;
    o->stride = d->cfg.max_det; o->count = sl.n; o->device = d->device; o->stream = d->post_stream;
    return RTMODT_OK;
}

}  // namespace rtmodt

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

const char *rtmodt_last_error(void) { return last_error().c_str(); }
const char *rtmodt_version(void) { return "rtmodt-hip 0.1 (gfx950)"; }
int rtmodt_option(const char *name, const char **value) {
    RT_CHECK(name && value, RTMODT_E_INVALID, "null argument");
    *value = rt_opt(name);
    return bad_option_pending();
}

int rtmodt_device_count(int *count) {
    RT_CHECK(count, RTMODT_E_INVALID, "null argument");
    RT_HIP(hipGetDeviceCount(count));
    return RTMODT_OK;
}
int rtmodt_synchronize(int device) {
    RT_HIP(hipSetDevice(device));
    RT_HIP(hipDeviceSynchronize());
    return RTMODT_OK;
}
int rtmodt_device_alloc(int device, size_t bytes, void **out) {
    RT_CHECK(out, RTMODT_E_INVALID, "null argument");
    RT_HIP(hipSetDevice(device));
    RT_HIP(hipMalloc(out, bytes));
    return RTMODT_OK;
}
// page-locked host memory for the frame source: an H2D copy from it is a real asynchronous DMA
int rtmodt_host_alloc(int device, size_t bytes, void **out) {
    RT_CHECK(out && bytes > 0, RTMODT_E_INVALID, "bad argument");
    RT_HIP(hipSetDevice(device));
    RT_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return RTMODT_OK;
}
int rtmodt_host_free(int device, void *ptr) {
    RT_HIP(hipSetDevice(device));
    RT_HIP(hipHostFree(ptr));
    return RTMODT_OK;
}
int rtmodt_device_free(int device, void *ptr) {
    RT_HIP(hipSetDevice(device));
    RT_HIP(hipFree(ptr));
    return RTMODT_OK;
}
int rtmodt_memcpy_h2d(int device, void *dst, const void *src, size_t bytes) {
    RT_HIP(hipSetDevice(device));
    RT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return RTMODT_OK;
}
int rtmodt_memcpy_d2h(int device, void *dst, const void *src, size_t bytes) {
    RT_HIP(hipSetDevice(device));
    RT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return RTMODT_OK;
}

void rtmodt_detector_destroy(rtmodt_detector *d) {
    if (!d) return;
    hipSetDevice(d->device);
    hipDeviceSynchronize();
    for (auto g : d->graph_execs) if (g) hipGraphExecDestroy(g);
    for (auto g : d->graphs) if (g) hipGraphDestroy(g);
    for (int par = 0; par < rtmodt_detector::MAX_STAGES; ++par)
        for (int st = 0; st < rtmodt_detector::MAX_STAGES; ++st) {
            for (int k = 0; k < 2; ++k) if (d->pipe_exec[par][st][k]) hipGraphExecDestroy(d->pipe_exec[par][st][k]);
            if (d->pipe_graph[par][st]) hipGraphDestroy(d->pipe_graph[par][st]);
        }
    for (int k = 1; k < rtmodt_detector::MAX_STAGES; ++k) if (d->stage_stream[k]) hipStreamDestroy(d->stage_stream[k]);
    for (auto &dn : d->dense) { hipFree(dn.box); hipFree(dn.score); hipFree(dn.cls); }
    for (size_t c = 1; c < d->chain_streams.size(); ++c) hipStreamDestroy(d->chain_streams[c]);
    for (auto e : d->chain_fork) hipEventDestroy(e);
    for (auto e : d->chain_join) hipEventDestroy(e);
    if (d->post_stream) hipStreamDestroy(d->post_stream);
    if (d->d_clock) hipFree(d->d_clock);
    if (d->copy_stream) hipStreamDestroy(d->copy_stream);
    for (auto st : d->aux_streams) if (st) hipStreamDestroy(st);
    for (auto st : d->pad_streams) hipStreamDestroy(st);
    for (auto e : d->aux_events) hipEventDestroy(e);
    for (void *p : d->dev_allocs) hipFree(p);
    hipFree(d->arena); hipFree(d->stage); hipFree(d->d_tab); hipFree(d->d_zeros); hipFree(d->lut255);
    hipFree(d->d_pred);
    hipFree(d->d_keys); hipFree(d->d_sbox); hipFree(d->d_sidx);
    for (auto &sl : d->slots) {
        hipFree(sl.o_xyxy); hipFree(sl.o_conf); hipFree(sl.o_cls); hipFree(sl.o_anchor); hipFree(sl.o_n);
        hipHostFree(sl.h_xyxy); hipHostFree(sl.h_conf); hipHostFree(sl.h_cls); hipHostFree(sl.h_n);
        for (hipEvent_t e : {sl.ev0, sl.evp, sl.ev1, sl.ev2, sl.done, sl.decoded, sl.copied}) if (e) hipEventDestroy(e);
        for (hipEvent_t e : sl.chain_done) if (e) hipEventDestroy(e);
        for (hipEvent_t e : sl.stage_done) if (e) hipEventDestroy(e);
    }
    if (d->stream) hipStreamDestroy(d->stream);
    delete d;
}

static int detector_create_impl(const rtmodt_det_cfg *cfg, rtmodt_detector *d) {
    d->cfg = *cfg;
    d->weight_path = cfg->weight_path;
    d->cfg.weight_path = d->weight_path.c_str();
    if (cfg->classes && cfg->n_classes > 0) {
        d->classes.assign(cfg->classes, cfg->classes + cfg->n_classes);
        d->class_mask[0] = d->class_mask[1] = 0;
        for (int c : d->classes)
            if (c >= 0 && c < 128) d->class_mask[c >> 6] |= 1ull << (c & 63);
    }
    d->cfg.classes = nullptr;
    d->device = cfg->device;
    d->B = cfg->batch; d->in_h = cfg->in_h; d->in_w = cfg->in_w; d->rect = cfg->rect != 0;
    RT_CHECK(cfg->half == 1, RTMODT_E_UNSUPPORTED, "half=0: this engine stores activations in fp16 only");
    RT_CHECK(d->B >= 1 && d->B <= 64, RTMODT_E_INVALID, "batch %d out of range [1,64]", d->B);
    RT_CHECK(d->in_h % 32 == 0 && d->in_w % 32 == 0 && d->in_h >= 32 && d->in_w >= 32 && d->in_h <= 1280 && d->in_w <= 1280,
             RTMODT_E_INVALID, "input size %dx%d must be a multiple of 32 in [32,1280]", d->in_w, d->in_h);
    RT_CHECK(cfg->max_det >= 1 && cfg->max_det <= 4096, RTMODT_E_INVALID, "max_det %d out of range [1,4096]", cfg->max_det);
    WeightFile wf;
    RT_TRY(read_weight_file(d->weight_path.c_str(), wf));
    d->scale_id = wf.scale_id; d->nc = wf.nc; d->reg_max = wf.reg_max;
    RT_CHECK(d->nc >= 1 && d->nc <= 128, RTMODT_E_UNSUPPORTED, "nc %d (1..128 supported)", d->nc);
    RT_HIP(hipSetDevice(d->device));
    RT_HIP(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    if (const char *e = rt_diag("POST_PRIO")) {     // experiment hook: the post-processing stream (one workgroup per image / stream) at another priority
        int lo = 0, hi = 0;
        RT_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        RT_HIP(hipStreamCreateWithPriority(&d->post_stream, hipStreamNonBlocking, atoi(e) > 0 ? hi : lo));
    } else {
        RT_HIP(hipStreamCreateWithFlags(&d->post_stream, hipStreamNonBlocking));
    }
    RT_HIP(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
    RT_HIP(hipMalloc((void **)&d->d_zeros, 256));
    RT_HIP(hipMemset(d->d_zeros, 0, 256));
    if (const char *e = rt_diag("H2D")) d->h2d_mode = atoi(e);
    if (const char *e = rt_opt("ZERO_COPY")) d->zero_copy = atoi(e) != 0;
    d->nms_plan = nms_plan_from_options();
    RT_TRY(build_graph(d, wf));
    RT_TRY(ensure_chain_streams(d));                       // may fall back to one chain: before anything is sized by n_chains

    int msw = cfg->max_src_w > 0 ? cfg->max_src_w : d->in_w, msh = cfg->max_src_h > 0 ? cfg->max_src_h : d->in_h;
    d->cfg.max_src_w = msw; d->cfg.max_src_h = msh;
    d->stage_per = align_up((size_t)msw * msh * 3 + 64, 256);
    RT_HIP(hipMalloc((void **)&d->stage, d->stage_per * d->B * rtmodt_detector::RING_SLOTS));
    d->tab_cap = (size_t)(d->in_w + d->in_h) * 3;
    RT_HIP(hipMalloc((void **)&d->d_tab, d->tab_cap * sizeof(int32_t)));

    size_t BA = (size_t)d->B * d->n_anchors, BD = (size_t)d->B * cfg->max_det;
    for (auto &dn : d->dense) {
        RT_HIP(hipMalloc((void **)&dn.box, BA * sizeof(float4)));
        RT_HIP(hipMalloc((void **)&dn.score, BA * sizeof(float)));
        RT_HIP(hipMalloc((void **)&dn.cls, BA * sizeof(int32_t)));
    }
    RT_HIP(hipMalloc((void **)&d->d_pred, BA * (4 + d->nc) * sizeof(float)));
    RT_HIP(hipMalloc((void **)&d->d_keys, BA * sizeof(uint64_t)));
    RT_HIP(hipMalloc((void **)&d->d_sbox, BA * sizeof(float4)));
    RT_HIP(hipMalloc((void **)&d->d_sidx, BA * sizeof(int32_t)));
    for (auto &sl : d->slots) {
        RT_HIP(hipMalloc((void **)&sl.o_xyxy, BD * 4 * sizeof(float)));
        RT_HIP(hipMalloc((void **)&sl.o_conf, BD * sizeof(float)));
        RT_HIP(hipMalloc((void **)&sl.o_cls, BD * sizeof(int32_t)));
        RT_HIP(hipMalloc((void **)&sl.o_anchor, BD * sizeof(int32_t)));
        RT_HIP(hipMalloc((void **)&sl.o_n, d->B * sizeof(int32_t)));
        RT_HIP(hipMemset(sl.o_n, 0, d->B * sizeof(int32_t)));
        RT_HIP(hipHostMalloc((void **)&sl.h_xyxy, BD * 4 * sizeof(float), hipHostMallocDefault));
        RT_HIP(hipHostMalloc((void **)&sl.h_conf, BD * sizeof(float), hipHostMallocDefault));
        RT_HIP(hipHostMalloc((void **)&sl.h_cls, BD * sizeof(int32_t), hipHostMallocDefault));
        RT_HIP(hipHostMalloc((void **)&sl.h_n, d->B * sizeof(int32_t), hipHostMallocDefault));
        RT_HIP(hipEventCreate(&sl.ev0)); RT_HIP(hipEventCreate(&sl.evp)); RT_HIP(hipEventCreate(&sl.ev1)); RT_HIP(hipEventCreate(&sl.ev2));
        RT_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        RT_HIP(hipEventCreateWithFlags(&sl.decoded, hipEventDisableTiming));
        RT_HIP(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
        sl.chain_done.assign(d->n_chains, nullptr);
        for (int c = 1; c < d->n_chains; ++c) RT_HIP(hipEventCreate(&sl.chain_done[c]));
        for (auto &e : sl.stage_done) RT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }

    {   // c / 255 in fp16 exactly as the letterbox kernel computes it (IEEE float division, then round to half)
        f16 lut[256];
        for (int c = 0; c < 256; ++c) lut[c] = (f16)((float)c / 255.0f);
        RT_HIP(hipMalloc((void **)&d->lut255, sizeof(lut)));
        RT_HIP(hipMemcpy(d->lut255, lut, sizeof(lut), hipMemcpyHostToDevice));
        if (const char *e = rt_opt("STEM_FUSE")) d->stem_fuse = atoi(e) != 0;      // A/B and test hook
    }
    // one eager pass (also sets kernel attributes) before capturing the graph
    RT_TRY(forward_eager_all(d));
    RT_HIP(hipStreamSynchronize(d->stream));
    if (cfg->autotune) RT_TRY(autotune_tiles(d));
    if (cfg->use_graph) RT_TRY(capture_graph(d));
    return RTMODT_OK;
}

int rtmodt_detector_create(const rtmodt_det_cfg *cfg, rtmodt_detector **out) {
    RT_CHECK(cfg && out && cfg->weight_path, RTMODT_E_INVALID, "null argument");
    install_debug_handlers();
    rtmodt_detector *d = new rtmodt_detector();
    int rc = detector_create_impl(cfg, d);
    if (rc == RTMODT_OK) rc = bad_option_pending();
    if (rc != RTMODT_OK) {
        std::string keep = last_error();
        rtmodt_detector_destroy(d);
        last_error() = keep;
        return rc;
    }
    *out = d;
    return RTMODT_OK;
}

int rtmodt_detector_enqueue_batch(rtmodt_detector *d, const uint8_t *const *frames, int n, int h, int w, int stride_bytes,
                                  int mem_kind) {
    RT_CHECK(d && frames, RTMODT_E_INVALID, "null argument");
    RT_CHECK(n >= 1 && n <= d->B, RTMODT_E_INVALID, "n %d outside [1, batch %d]", n, d->B);
    RT_CHECK(h >= 1 && w >= 1 && stride_bytes >= w * 3, RTMODT_E_INVALID, "bad frame geometry %dx%d pitch %d", w, h, stride_bytes);
    RT_CHECK(d->n_pending < rtmodt_detector::RING_SLOTS, RTMODT_E_CAPACITY, "%d batches already in flight: fetch before enqueueing more",
             d->n_pending);
    RT_HIP(hipSetDevice(d->device));
    rtmodt_detector::Slot &sl = d->slots[d->head];
    LbHost g = letterbox_geometry(h, w, d->in_h, d->in_w, d->rect);
    RT_CHECK(g.new_w <= d->in_w && g.new_h <= d->in_h, RTMODT_E_INVALID, "a %dx%d frame does not fit the %dx%d rectangle this detector was built for", w, h, d->in_w, d->in_h);
    // Opt-in (RTMODT_ZERO_COPY=1): page-locked frames the device can address (rtmodt_host_alloc / hipHostMalloc) that need no
    // resize are not copied -- the stem conv reads their bytes in place, over PCIe, once.  The caller keeps such frames
    // unchanged until the batch is fetched.
    bool in_place = false;
    if (mem_kind == RTMODT_MEM_HOST && d->zero_copy && d->stem_fuse && !g.resize) {
        in_place = true;
        for (int i = 0; i < n && in_place; ++i) {
            hipPointerAttribute_t at{};
            if (hipPointerGetAttributes(&at, frames[i]) != hipSuccess) { (void)hipGetLastError(); in_place = false; break; }
            in_place = at.type == hipMemoryTypeHost && at.devicePointer != nullptr;
            if (in_place) d->fptrs.p[i] = (const uint8_t *)at.devicePointer;
        }
    }
    d->last_in_place = in_place;
    if (in_place) {
    } else if (mem_kind == RTMODT_MEM_HOST) {
        RT_CHECK((size_t)h * stride_bytes <= d->stage_per, RTMODT_E_CAPACITY, "frame %dx%d exceeds max_src %dx%d", w, h, d->cfg.max_src_w,
                 d->cfg.max_src_h);
        uint8_t *area = d->stage + d->stage_per * d->B * d->head;
        // frames that sit back to back in the caller's (page-locked) ring slot travel as ONE copy
        const size_t fbytes = (size_t)h * stride_bytes;
        bool contig = n > 1;
        for (int i = 1; i < n && contig; ++i) contig = frames[i] == frames[0] + fbytes * i;
        const size_t dst_step = contig ? fbytes : d->stage_per;
        for (int i = 0; i < n; ++i) d->fptrs.p[i] = area + dst_step * i;
        auto upload_on = [&](hipStream_t cs) -> int {
            if (contig) RT_HIP(hipMemcpyAsync(area, frames[0], fbytes * n, hipMemcpyHostToDevice, cs));
            else for (int i = 0; i < n; ++i) RT_HIP(hipMemcpyAsync(area + dst_step * i, frames[i], fbytes, hipMemcpyHostToDevice, cs));
            return RTMODT_OK;
        };
        if (d->h2d_mode == 1) {
            // host-synchronised upload: the DMA runs on the copy stream with NO event recorded on it and no stream waiting for
            // it, so no barrier packet of the copy ever sits in a hardware queue that a stage of the net shares; the host waits
            // for the DMA itself (the S + 1 batches in flight keep the device busy meanwhile).  The staging area of this ring
            // slot was last read by the stem of a batch that has been fetched.
            RT_TRY(upload_on(d->copy_stream));
            RT_HIP(hipStreamSynchronize(d->copy_stream));
        } else {
            // three stages leave the copy stream no hardware queue of its own (it shares the third stage's, and its event waits
            // then hold that stage up: 12.8 k frames/s): the uploads go through the main stream instead, in front of stage 1
            const bool on_main = d->pipe && d->n_stages == 3 && !(rt_diag("COPY_ON_MAIN") && atoi(rt_diag("COPY_ON_MAIN")) == 0);
            if (sl.staged && !on_main) {                       // the launches that last read this area are done
                RT_HIP(hipStreamWaitEvent(d->copy_stream, sl.evp, 0));
                if (sl.chained) for (int c = 1; c < d->n_chains; ++c) RT_HIP(hipStreamWaitEvent(d->copy_stream, sl.chain_done[c], 0));
            }
            RT_TRY(upload_on(on_main ? d->stream : d->copy_stream));
            if (!on_main) {
                RT_HIP(hipEventRecord(sl.copied, d->copy_stream));
                RT_HIP(hipStreamWaitEvent(d->stream, sl.copied, 0));
            }
        }
        sl.staged = true;
    } else {
        for (int i = 0; i < n; ++i) d->fptrs.p[i] = frames[i];
    }
    for (int i = n; i < d->B; ++i) d->fptrs.p[i] = d->fptrs.p[0];
    if (g.resize && (h != d->tab_h || w != d->tab_w)) {
        std::vector<int32_t> xo, x0, x1, yo, y0, y1;
        build_resize_tables(g.new_w, w, xo, x0, x1);
        build_resize_tables(g.new_h, h, yo, y0, y1);
        RT_CHECK((size_t)(g.new_w + g.new_h) * 3 <= d->tab_cap, RTMODT_E_INVALID, "resize table overflow");
        RT_HIP(hipStreamSynchronize(d->stream));
        int32_t *p = d->d_tab;
        const std::vector<int32_t> *src[6] = {&xo, &x0, &x1, &yo, &y0, &y1};
        const int32_t *dst[6];
        for (int k = 0; k < 6; ++k) {
            RT_HIP(hipMemcpy(p, src[k]->data(), src[k]->size() * 4, hipMemcpyHostToDevice));
            dst[k] = p; p += src[k]->size();
        }
        d->tabs = ResizeTables{dst[0], dst[1], dst[2], dst[3], dst[4], dst[5]};
        d->tab_h = h; d->tab_w = w;
    }
    LetterboxGeom lg{h, w, g.new_w, g.new_h, g.top, g.left, g.resize};
    TensorView img; img.base = d->tensors[d->img_t].ptr; img.H = d->in_h; img.W = d->in_w; img.C = 4; img.pad = 1; img.c = 4;
    d->cur_dense = d->head;                            // ring slot == dense set
    d->last_lg = lg; d->last_pitch = stride_bytes;
    d->last_fused = d->stem_fuse && !g.resize;
    const bool graphs = (!d->graph_execs.empty() || (d->pipe && d->pipe_exec[0][0][0])) && !d->want_pred;
    const bool chained = graphs && d->n_chains > 1;
    // a lagging chain of the previous batch may still be reading what the main stream is about to overwrite or time
    if (!chained || !d->last_fused || !d->chain_free_run)
        for (int k = 0; k < rtmodt_detector::RING_SLOTS; ++k)
            if (d->slots[k].chained && !d->slots[k].joined) {
                for (int c = 1; c < d->n_chains; ++c) RT_HIP(hipStreamWaitEvent(d->stream, d->slots[k].chain_done[c], 0));
                d->slots[k].joined = true;
            }
    if (graphs && d->pipe && d->batch_no >= (uint64_t)d->n_stages) {   // this arena copy was last read by the last stage of batch t - S
        const rtmodt_detector::Slot &old = d->slots[(d->head + rtmodt_detector::RING_SLOTS - d->n_stages) % rtmodt_detector::RING_SLOTS];
        RT_HIP(hipStreamWaitEvent(d->stream, old.ev1, 0));
    }
    RT_HIP(hipEventRecord(sl.ev0, d->stream));
    if (!d->last_fused) {
        if (graphs && d->pipe) img.base = (f16 *)((char *)img.base + (size_t)(d->batch_no % d->n_stages) * d->arena_stride);
        RT_TRY(launch_letterbox(d->fptrs, stride_bytes, lg, d->tabs, img, d->B, d->stream));
        if (chained) RT_HIP(hipEventRecord(sl.evp, d->stream));
    }
    sl.chained = chained; sl.joined = false;
    d->run_par = 0;
    if (graphs && d->pipe) {
        const int S = d->n_stages, par = (int)(d->batch_no % S), inst = (int)((d->batch_no / S) & 1);
        d->run_par = par;
        RT_TRY(run_first(d, d->par_ops[par], 0, d->last_fused, d->stream));
        RT_HIP(hipEventRecord(sl.evp, d->stream));
        for (int k = 0; k < S; ++k) {
            hipStream_t st = d->stage_stream[k];
            if (k > 0) RT_HIP(hipStreamWaitEvent(st, sl.stage_done[k - 1], 0));
            RT_HIP(hipGraphLaunch(d->pipe_exec[par][k][inst], st));
            if (k + 1 < S) RT_HIP(hipEventRecord(sl.stage_done[k], st));
        }
        hipStream_t last = d->stage_stream[S - 1];
        RT_TRY(run_decode_sub(d, 0, d->B, last));
        RT_HIP(hipEventRecord(sl.ev1, last));
        RT_HIP(hipEventRecord(sl.decoded, last));
    } else if (chained) {
        sl.joined = !d->last_fused || !d->chain_free_run;
        RT_TRY(forward_chains(d, sl, mem_kind == RTMODT_MEM_HOST, sl.joined));
    } else {
        RT_TRY(run_stems(d, d->last_fused));
        RT_HIP(hipEventRecord(sl.evp, d->stream));          // the frames have been consumed (letterbox [+ stem])
        if (graphs) RT_TRY(forward_graphs(d));
        else RT_TRY(forward_eager(d));
    }
    if (!(graphs && d->pipe)) {
        RT_HIP(hipEventRecord(sl.ev1, d->stream));
        RT_HIP(hipEventRecord(sl.decoded, d->stream));
    }
    d->last_par = d->run_par;
    // post-processing on its own stream: one small workgroup per image, latency-bound -- it runs
    // underneath the next batch's forward pass instead of in front of it
    RT_HIP(hipStreamWaitEvent(d->post_stream, sl.decoded, 0));
    if (chained && !sl.joined) for (int c = 1; c < d->n_chains; ++c) RT_HIP(hipStreamWaitEvent(d->post_stream, sl.chain_done[c], 0));
    RT_TRY(run_nms(d, g, h, w, sl));
    RT_HIP(hipEventRecord(sl.ev2, d->post_stream));
    if (d->clock_on) {                                    // one wave, ~20 us, behind the NMS of this batch (the forward pass of the next ones is running)
        hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(64), 0, d->post_stream, d->d_clock + 4 * (d->clock_n % rtmodt_detector::CLOCK_SLOTS), 2000u);
        RT_HIP(hipGetLastError());
        d->clock_n += 1;
    }
    // results travel to pinned host memory right behind the kernels; fetch() only waits for `done`
    const int md = d->cfg.max_det;
    RT_HIP(hipMemcpyAsync(sl.h_n, sl.o_n, n * sizeof(int32_t), hipMemcpyDeviceToHost, d->post_stream));
    RT_HIP(hipMemcpyAsync(sl.h_xyxy, sl.o_xyxy, (size_t)n * md * 4 * sizeof(float), hipMemcpyDeviceToHost, d->post_stream));
    RT_HIP(hipMemcpyAsync(sl.h_conf, sl.o_conf, (size_t)n * md * sizeof(float), hipMemcpyDeviceToHost, d->post_stream));
    RT_HIP(hipMemcpyAsync(sl.h_cls, sl.o_cls, (size_t)n * md * sizeof(int32_t), hipMemcpyDeviceToHost, d->post_stream));
    RT_HIP(hipEventRecord(sl.done, d->post_stream));
    sl.n = n;
    d->newest = d->head;
    d->head = (d->head + 1) % rtmodt_detector::RING_SLOTS;
    d->batch_no += 1;
    d->n_pending += 1;
    d->last_h = h; d->last_w = w;
    return RTMODT_OK;
}

// ---- in-kernel shader clock ------------------------------------------------------------------------------------------
// The chip lowers its shader clock under an MFMA-dense load, and the hwmon / DPM reading overstates what the kernels see
// (measured with phase stamps: 1.70-1.76 GHz inside the tap-reuse kernel against 2.05-2.13 GHz from hwmon).  While enabled,
// ONE wave launched on the post-processing stream behind every batch's NMS reads the shader-cycle counter (s_memtime) and the
// constant 100 MHz counter (s_memrealtime) about 20 us apart -- while the other stages' forward kernels run on the rest of the
// chip -- so that clock = d(memtime) / d(memrealtime) x 100 MHz is the clock those kernels ran at.  Bounded spin (s_sleep).
int rtmodt_detector_clock_enable(rtmodt_detector *d, int on) {
    RT_CHECK(d, RTMODT_E_INVALID, "null argument");
    RT_HIP(hipSetDevice(d->device));
    if (on && !d->d_clock) {
        RT_HIP(hipMalloc((void **)&d->d_clock, sizeof(unsigned long long) * 4 * rtmodt_detector::CLOCK_SLOTS));
        RT_HIP(hipMemset(d->d_clock, 0, sizeof(unsigned long long) * 4 * rtmodt_detector::CLOCK_SLOTS));
    }
    if (on && !d->clock_on) d->clock_n = 0;
    d->clock_on = on != 0;
    return RTMODT_OK;
}

int rtmodt_detector_clock_read(rtmodt_detector *d, double *ghz_mean, double *ghz_min, double *ghz_max, int32_t *n_samples) {
    RT_CHECK(d && d->d_clock, RTMODT_E_INVALID, "clock sampling was never enabled");
    RT_HIP(hipSetDevice(d->device));
    RT_HIP(hipStreamSynchronize(d->post_stream));
    const int n = (int)std::min<uint64_t>(d->clock_n, rtmodt_detector::CLOCK_SLOTS);
    std::vector<unsigned long long> h((size_t)4 * std::max(n, 1));
    if (n) RT_HIP(hipMemcpy(h.data(), d->d_clock, sizeof(unsigned long long) * 4 * n, hipMemcpyDeviceToHost));
    double sum = 0, lo = 1e30, hi = 0;
    int good = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long t0 = h[4 * i], r0 = h[4 * i + 1], t1 = h[4 * i + 2], r1 = h[4 * i + 3];
        if (r1 <= r0 || t1 <= t0) continue;
        const double ghz = (double)(t1 - t0) / (double)(r1 - r0) * 0.1;
        sum += ghz; lo = std::min(lo, ghz); hi = std::max(hi, ghz); ++good;
    }
    if (ghz_mean) *ghz_mean = good ? sum / good : 0.0;
    if (ghz_min) *ghz_min = good ? lo : 0.0;
    if (ghz_max) *ghz_max = good ? hi : 0.0;
    if (n_samples) *n_samples = good;
    d->clock_n = 0;
    return RTMODT_OK;
}

int rtmodt_detector_fetch(rtmodt_detector *d, float *xyxy, float *conf, int32_t *cls, int32_t *n_out) {
    RT_CHECK(d && n_out, RTMODT_E_INVALID, "null argument");
    RT_CHECK(d->n_pending > 0, RTMODT_E_INVALID, "fetch without a pending enqueue_batch");
    RT_HIP(hipSetDevice(d->device));
    const int idx = (d->head + rtmodt_detector::RING_SLOTS - d->n_pending) % rtmodt_detector::RING_SLOTS;   // oldest in flight
    rtmodt_detector::Slot &sl = d->slots[idx];
    RT_HIP(hipEventSynchronize(sl.done));
    d->n_pending -= 1;
    d->last_fetched = idx;
    const int n = sl.n, md = d->cfg.max_det;
    memcpy(n_out, sl.h_n, n * sizeof(int32_t));
    if (xyxy) memcpy(xyxy, sl.h_xyxy, (size_t)n * md * 4 * sizeof(float));
    if (conf) memcpy(conf, sl.h_conf, (size_t)n * md * sizeof(float));
    if (cls) memcpy(cls, sl.h_cls, (size_t)n * md * sizeof(int32_t));
    return RTMODT_OK;
}

int rtmodt_detector_detect_batch(rtmodt_detector *d, const uint8_t *const *frames, int n, int h, int w, int stride_bytes, int mem_kind,
                                 float *xyxy, float *conf, int32_t *cls, int32_t *n_out) {
    RT_CHECK(d && d->n_pending == 0, RTMODT_E_INVALID, "detect_batch with batches still in flight: fetch them first");
    RT_TRY(rtmodt_detector_enqueue_batch(d, frames, n, h, w, stride_bytes, mem_kind));
    return rtmodt_detector_fetch(d, xyxy, conf, cls, n_out);
}

int rtmodt_detector_detect(rtmodt_detector *d, const uint8_t *bgr, int h, int w, int stride_bytes, float *xyxy, float *conf,
                           int32_t *cls, int32_t *n_out) {
    const uint8_t *f[1] = {bgr};
    return rtmodt_detector_detect_batch(d, f, 1, h, w, stride_bytes, RTMODT_MEM_HOST, xyxy, conf, cls, n_out);
}

int rtmodt_detector_info(rtmodt_detector *d, int32_t *scale_id, int32_t *nc, int32_t *n_anchors, int32_t *n_convs,
                         int64_t *conv_flops_per_frame, int64_t *arena_bytes) {
    RT_CHECK(d, RTMODT_E_INVALID, "null argument");
    if (scale_id) *scale_id = d->scale_id;
    if (nc) *nc = d->nc;
    if (n_anchors) *n_anchors = d->n_anchors;
    if (n_convs) {
        int c = 0;
        for (auto &op : d->ops) c += (op.kind == OP_GROUP || op.kind == OP_BNECK) ? (int)op.group.size() : (op.kind == OP_CONV || op.kind == OP_STEM);
        *n_convs = c;
    }
    if (conv_flops_per_frame) *conv_flops_per_frame = d->flops_per_frame;
    if (arena_bytes) *arena_bytes = (int64_t)d->arena_bytes;
    return RTMODT_OK;
}

// dense copy of a channel-slice view of image `img`
int rtmodt_detector_stages(rtmodt_detector *d, int32_t *n_stages) {
    RT_CHECK(d && n_stages, RTMODT_E_INVALID, "null argument");
    *n_stages = d->pipe ? d->n_stages : 1;
    return RTMODT_OK;
}

int rtmodt_detector_chains(rtmodt_detector *d, int32_t *n_chains) {
    RT_CHECK(d && n_chains, RTMODT_E_INVALID, "null argument");
    *n_chains = d->n_chains;
    return RTMODT_OK;
}

static int fetch_view(rtmodt_detector *d, const TensorView &v, int img, uint16_t *out, int par = -1) {
    size_t per = (size_t)(v.H + 2 * v.pad) * (v.W + 2 * v.pad) * v.C;
    std::vector<uint16_t> tmp(per);
    if (par < 0) par = d->last_par;                        // the arena copy the newest batch ran in
    RT_HIP(hipMemcpy(tmp.data(), (const char *)(v.base + per * img) + (size_t)par * d->arena_stride, per * 2, hipMemcpyDeviceToHost));
    for (int y = 0; y < v.H; ++y)
        for (int x = 0; x < v.W; ++x)
            memcpy(out + ((size_t)y * v.W + x) * v.c, &tmp[((size_t)(y + v.pad) * (v.W + 2 * v.pad) + x + v.pad) * v.C + v.coff], (size_t)v.c * 2);
    return RTMODT_OK;
}

// head_final keeps the head rows in LDS: have it write them out this once -- the very rows it decodes, so a `pred`
// derived from them is what the detections came from
static int materialize_heads(rtmodt_detector *d) {
    HeadFinalArgs h = d->hf;
    h.conf = d->cfg.conf; h.class_mask[0] = d->class_mask[0]; h.class_mask[1] = d->class_mask[1];
    const rtmodt_detector::Dense &dn = d->dense[d->cur_dense];
    h.box = dn.box; h.score = dn.score; h.cls = dn.cls;
    const size_t par = (size_t)d->last_par * d->arena_stride / sizeof(f16);      // the arena copy the newest batch ran in
    for (int l = 0; l < 3; ++l) { h.lvl[l].xb += par; h.lvl[l].xc += par; h.lvl[l].heads = d->tensors[d->head_t[l]].ptr + par; }
    RT_TRY(launch_head_final(h, d->stream));
    RT_HIP(hipStreamSynchronize(d->stream));
    return RTMODT_OK;
}

int rtmodt_detector_debug_fetch(rtmodt_detector *d, int img, uint16_t *input_f16, uint16_t *heads_f16, float *pred) {
    RT_CHECK(d && img >= 0 && img < d->B, RTMODT_E_INVALID, "bad argument");
    RT_HIP(hipSetDevice(d->device));
    RT_HIP(hipDeviceSynchronize());
    if (d->newest >= 0) d->cur_dense = d->newest;
    if (input_f16) {
        if (d->last_fused) {                               // the fused stem never wrote the image tensor: letterbox the same frames now
            TensorView iv; iv.base = d->tensors[d->img_t].ptr; iv.H = d->in_h; iv.W = d->in_w; iv.C = 4; iv.pad = 1; iv.c = 4;
            RT_TRY(launch_letterbox(d->fptrs, d->last_pitch, d->last_lg, d->tabs, iv, d->B, d->stream));
            RT_HIP(hipStreamSynchronize(d->stream));
        }
        TensorView v; const Tensor &t = d->tensors[d->img_t];
        v.base = t.ptr; v.H = t.H; v.W = t.W; v.C = 4; v.pad = 1; v.coff = 0; v.c = 3;
        RT_TRY(fetch_view(d, v, img, input_f16, d->last_fused ? 0 : -1));      // (the re-run above wrote the first arena copy)
    }
    if (d->head_final && (heads_f16 || pred)) RT_TRY(materialize_heads(d));
    if (heads_f16) {
        uint16_t *o = heads_f16;
        for (int l = 0; l < 3; ++l) {
            const Tensor &t = d->tensors[d->head_t[l]];
            TensorView v; v.base = t.ptr; v.H = t.H; v.W = t.W; v.C = t.C; v.pad = 0; v.coff = 0; v.c = 64 + d->nc;
            RT_TRY(fetch_view(d, v, img, o));
            o += (size_t)t.H * t.W * (64 + d->nc);
        }
    }
    if (pred) {
        // re-run decode with the pred dump enabled (inputs are still resident)
        d->want_pred = true;
        d->run_par = d->last_par;
        int rc = run_decode(d);
        d->want_pred = false;
        d->run_par = 0;
        RT_TRY(rc);
        RT_HIP(hipStreamSynchronize(d->stream));
        size_t per = (size_t)(4 + d->nc) * d->n_anchors;
        RT_HIP(hipMemcpy(pred, d->d_pred + per * img, per * sizeof(float), hipMemcpyDeviceToHost));
    }
    return RTMODT_OK;
}

int rtmodt_detector_debug_layer(rtmodt_detector *d, const char *name, int img, uint16_t *out, int32_t *hwc) {
    RT_CHECK(d && name && img >= 0 && img < d->B, RTMODT_E_INVALID, "bad argument");
    auto it = d->layer_out.find(name);
    RT_CHECK(it != d->layer_out.end(), RTMODT_E_INVALID, "no fused conv named %s", name);
    const TensorView &v = it->second;
    // what is stored is decided by the ops that RUN: with sub-batch chains those were tuned on their own (autotune_tiles) and may fuse
    // a launch the whole-batch ops keep apart -- the tensor between them would then hold the tuner's stale data
    const std::vector<Op> &ops = d->n_chains > 1 ? d->chain_ops[0] : d->ops;
    if (d->head_final && strncmp(name, "22.cv", 5) == 0 && strlen(name) > 2 && strcmp(name + strlen(name) - 2, ".2") == 0) {
        RT_HIP(hipSetDevice(d->device));
        RT_HIP(hipDeviceSynchronize());
        if (d->newest >= 0) d->cur_dense = d->newest;
        RT_TRY(materialize_heads(d));                      // Detect's last convs live inside head_final
    }
    if (!ops.empty() && ops[0].kind == OP_STEM && ops[0].front_on && (strcmp(name, "0") == 0 || strcmp(name, "1") == 0))
        return fail(RTMODT_E_UNSUPPORTED, "%s lives in LDS only: the front end (stem, layer 1, 2.cv1) runs as one launch", name);
    for (auto &op : ops)                               // a conv whose 1x1 tail runs in the same launch stores only the tail's output
        if (op.kind == OP_CONV && op.tail_on && !op.skip && op.name == name)
            return fail(RTMODT_E_UNSUPPORTED, "%s is consumed in LDS by the 1x1 conv fused into its launch", name);
    for (auto &op : ops)                               // ... and so does a fused Bottleneck whose C2f.cv2 runs as its tail
        if (op.kind == OP_BNECK && op.fused && op.tail_on && op.name == std::string(name).substr(0, std::string(name).rfind('.')) + " (cv1+cv2)")
            return fail(RTMODT_E_UNSUPPORTED, "%s is consumed in LDS by the 1x1 conv fused into its launch", name);
    for (auto &op : ops)                               // the first conv of a fused Bottleneck never leaves the CU
        if (op.kind == OP_BNECK && op.fused && op.name == std::string(name).substr(0, std::string(name).rfind('.')) + " (cv1+cv2)" &&
            std::string(name).size() > 4 && std::string(name).compare(std::string(name).size() - 4, 4, ".cv1") == 0)
            return fail(RTMODT_E_UNSUPPORTED, "%s is the LDS-resident intermediate of a fused Bottleneck launch", name);
    if (hwc) { hwc[0] = v.H; hwc[1] = v.W; hwc[2] = v.c; }
    if (!out) return RTMODT_OK;
    RT_HIP(hipSetDevice(d->device));
    RT_HIP(hipDeviceSynchronize());                        // every chain stream
    return fetch_view(d, v, img, out);
}

int rtmodt_detector_profile(rtmodt_detector *d, int iters, int max_entries, const char **names, float *ms, int64_t *flops,
                            int32_t *n_entries) {
    RT_CHECK(d && iters >= 1 && n_entries, RTMODT_E_INVALID, "bad argument");
    RT_HIP(hipSetDevice(d->device));
    // with sub-batch chains the launches that run are those of one chain (its sub-batch), timed here alone on the device
    const std::vector<Op> &ops = d->n_chains > 1 ? d->chain_ops[0] : d->ops;
    d->run_par = 0;                                        // eager launches work in the first arena copy
    const int PB = d->B / d->n_chains;
    const int n = (int)ops.size() + 1;
    std::vector<hipEvent_t> ev((size_t)n + 1);
    for (auto &e : ev) RT_HIP(hipEventCreate(&e));
    std::vector<double> acc(n, 0.0);
    RT_HIP(hipDeviceSynchronize());
    for (int it = 0; it < iters; ++it) {
        RT_HIP(hipEventRecord(ev[0], d->stream));
        for (int i = 0; i < n - 1; ++i) {
            if (ops[i].kind == OP_STEM) RT_TRY(run_first(d, ops, 0, d->last_fused, d->stream));
            else RT_TRY(run_op(d, ops[i]));
            RT_HIP(hipEventRecord(ev[i + 1], d->stream));
        }
        RT_TRY(run_decode_sub(d, 0, PB, d->stream));
        RT_HIP(hipEventRecord(ev[n], d->stream));
        RT_HIP(hipStreamSynchronize(d->stream));
        for (int i = 0; i < n; ++i) {
            float t = 0;
            RT_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
            acc[i] += t;
        }
    }
    for (auto &e : ev) hipEventDestroy(e);
    d->prof_names.clear();
    for (auto &op : ops) {
        char buf[160];
        if (op.kind == OP_CONV && op.skip && ops[0].kind == OP_STEM && ops[0].front_on && (&op == &ops[1] || &op == &ops[2])) {
            snprintf(buf, sizeof(buf), "%s [runs inside the front-end launch]", op.name.c_str());
        } else if (op.kind == OP_STEM && op.front_on) {
            snprintf(buf, sizeof(buf), "%s [front end fused: %sstem + layer 1 + 2.cv1, one launch]", op.name.c_str(), d->last_fused ? "letterbox + " : "");
        } else if (op.kind == OP_CONV && op.skip) {
            snprintf(buf, sizeof(buf), "%s [runs as the tail of the previous launch]", op.name.c_str());
        } else if (op.kind == OP_CONV) {
            snprintf(buf, sizeof(buf), "%s [M=%d N=%d K=%d k%d s%d tile %s]", op.name.c_str(), PB * op.conv.out.H * op.conv.out.W,
                     op.conv.cout, op.conv.ks * op.conv.ks * op.conv.cin, op.conv.ks, op.conv.stride, tile_name(op.tail_on ? op.tail_tile : op.conv.tile));
        } else if (op.kind == OP_BNECK) {
            if (op.fused) snprintf(buf, sizeof(buf), "%s [fused bottleneck%s, c=%d, %dx%d]", op.name.c_str(), op.tail_on ? " + C2f.cv2 tail" : "", op.bneck.c, op.bneck.in.H, op.bneck.in.W);
            else snprintf(buf, sizeof(buf), "%s [two launches: %s, %s]", op.name.c_str(), tile_name(op.group[0].tile), tile_name(op.group[1].tile));
        } else if (op.kind == OP_GROUP && op.skip) {
            snprintf(buf, sizeof(buf), "%s [runs inside head_final]", op.name.c_str());
        } else if (op.kind == OP_GROUP) {
            snprintf(buf, sizeof(buf), "%s [group of %zu, tile %s]", op.name.c_str(), op.group.size(), tile_name(op.group_tile));
        } else if (op.kind == OP_STEM && d->last_fused) {
            snprintf(buf, sizeof(buf), "%s [letterbox + stem fused]", op.name.c_str());
        } else {
            snprintf(buf, sizeof(buf), "%s", op.name.c_str());
        }
        d->prof_names.push_back(buf);
    }
    d->prof_names.push_back(d->head_final ? "22.stage2 + decode [head_final: one launch]" : "decode");
    *n_entries = n;
    for (int i = 0; i < n && i < max_entries; ++i) {
        if (names) names[i] = d->prof_names[i].c_str();
        if (ms) ms[i] = (float)(acc[i] / iters);
        if (flops) flops[i] = i < n - 1 ? ops[i].flops * PB : 0;
    }
    return RTMODT_OK;
}

int rtmodt_detector_last_timing(rtmodt_detector *d, float *total_ms, float *forward_ms) {
    RT_CHECK(d && d->last_fetched >= 0, RTMODT_E_INVALID, "no batch has been fetched");
    RT_HIP(hipSetDevice(d->device));
    const rtmodt_detector::Slot &sl = d->slots[d->last_fetched];          // the batch fetch() returned last
    RT_HIP(hipEventSynchronize(sl.ev2));
    if (total_ms) RT_HIP(hipEventElapsedTime(total_ms, sl.ev0, sl.ev2));
    if (forward_ms) {
        RT_HIP(hipEventElapsedTime(forward_ms, sl.ev0, sl.ev1));
        if (sl.chained && !sl.joined)                         // free-running chains: the batch is decoded when the last of them is
            for (int c = 1; c < d->n_chains; ++c) {
                float t = 0;
                RT_HIP(hipEventElapsedTime(&t, sl.ev0, sl.chain_done[c]));
                *forward_ms = std::max(*forward_ms, t);
            }
    }
    if (rt_diag("DEBUG_GAPS")) {                    // idle time of the main stream between two batches
        static hipEvent_t base = nullptr;
        static float prev_end = -1.f;
        if (!base) { hipEventCreate(&base); hipEventRecord(base, d->stream); hipEventSynchronize(base); }
        float t0 = 0, t1 = 0, t2 = 0;
        if (hipEventElapsedTime(&t0, base, sl.ev0) == hipSuccess && hipEventElapsedTime(&t1, base, sl.ev1) == hipSuccess &&
            hipEventElapsedTime(&t2, base, sl.ev2) == hipSuccess) {
            if (prev_end >= 0) fprintf(stderr, "[rtmodt] main-stream idle before this batch %.3f ms; forward %.3f; decode_end -> nms_end %.3f\n", t0 - prev_end, t1 - t0, t2 - t1);
            prev_end = t1;
        }
    }
    return RTMODT_OK;
}

int rtmodt_detector_stage_times(rtmodt_detector *d, float *preprocess_ms, float *inference_ms, float *nms_ms) {
    RT_CHECK(d && d->last_fetched >= 0, RTMODT_E_INVALID, "no batch has been fetched");
    RT_HIP(hipSetDevice(d->device));
    const rtmodt_detector::Slot &sl = d->slots[d->last_fetched];
    RT_HIP(hipEventSynchronize(sl.ev2));
    if (preprocess_ms) RT_HIP(hipEventElapsedTime(preprocess_ms, sl.ev0, sl.evp));
    if (inference_ms) RT_HIP(hipEventElapsedTime(inference_ms, sl.evp, sl.ev1));
    if (nms_ms) RT_HIP(hipEventElapsedTime(nms_ms, sl.ev1, sl.ev2));
    return RTMODT_OK;
}

// ---- standalone pieces ----------------------------------------------------------------
int rtmodt_nms_pred(int device, const float *pred, int nc, int A, float conf, float iou, const int32_t *classes, int n_classes,
                    int agnostic, int max_det, float *xyxy, float *conf_out, int32_t *cls, int32_t *anchor_idx, int32_t *n_out) {
    RT_CHECK(pred && n_out && nc >= 1 && nc <= 128 && A >= 1 && max_det >= 1 && max_det <= 4096, RTMODT_E_INVALID, "bad argument");
    RT_HIP(hipSetDevice(device));
    uint64_t mask[2] = {~0ull, ~0ull};
    if (classes && n_classes > 0) {
        mask[0] = mask[1] = 0;
        for (int i = 0; i < n_classes; ++i)
            if (classes[i] >= 0 && classes[i] < 128) mask[classes[i] >> 6] |= 1ull << (classes[i] & 63);
    }
    struct Bufs {
        std::vector<void *> p;
        ~Bufs() { for (void *q : p) hipFree(q); }
        int get(size_t bytes, void **o) { RT_HIP(hipMalloc(o, bytes)); p.push_back(*o); return RTMODT_OK; }
    } bufs;
    float *dpred; float4 *box, *sbox; float *score; int32_t *dcls, *sidx; uint64_t *keys;
    float *oxy, *ocf; int32_t *ocl, *oan, *on;
    RT_TRY(bufs.get((size_t)(4 + nc) * A * 4, (void **)&dpred));
    RT_TRY(bufs.get((size_t)A * 16, (void **)&box)); RT_TRY(bufs.get((size_t)A * 16, (void **)&sbox));
    RT_TRY(bufs.get((size_t)A * 4, (void **)&score)); RT_TRY(bufs.get((size_t)A * 4, (void **)&dcls));
    RT_TRY(bufs.get((size_t)A * 4, (void **)&sidx)); RT_TRY(bufs.get((size_t)A * 8, (void **)&keys));
    RT_TRY(bufs.get((size_t)max_det * 16, (void **)&oxy)); RT_TRY(bufs.get((size_t)max_det * 4, (void **)&ocf));
    RT_TRY(bufs.get((size_t)max_det * 4, (void **)&ocl)); RT_TRY(bufs.get((size_t)max_det * 4, (void **)&oan));
    RT_TRY(bufs.get(4, (void **)&on));
    RT_HIP(hipMemcpy(dpred, pred, (size_t)(4 + nc) * A * 4, hipMemcpyHostToDevice));
    RT_TRY(launch_pred_candidates(dpred, nc, A, conf, mask, box, score, dcls, nullptr));
    NmsArgs a{};
    a.B = 1; a.n_anchors = A; a.max_det = max_det; a.agnostic = agnostic; a.iou = iou;
    a.box = box; a.score = score; a.cls = dcls; a.keys = keys; a.sbox = sbox; a.sidx = sidx;
    a.rescale = 0; a.gain = 1.f;
    a.out_xyxy = oxy; a.out_conf = ocf; a.out_cls = ocl; a.out_anchor = oan; a.out_n = on;
    RT_TRY(launch_nms(a, nms_plan_from_options(), nullptr));
    RT_HIP(hipDeviceSynchronize());
    int n = 0;
    RT_HIP(hipMemcpy(&n, on, 4, hipMemcpyDeviceToHost));
    *n_out = n;
    if (xyxy) RT_HIP(hipMemcpy(xyxy, oxy, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (conf_out) RT_HIP(hipMemcpy(conf_out, ocf, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (cls) RT_HIP(hipMemcpy(cls, ocl, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (anchor_idx) RT_HIP(hipMemcpy(anchor_idx, oan, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RTMODT_OK;
}

int rtmodt_preprocess(int device, const uint8_t *bgr, int h, int w, int stride_bytes, int in_w, int in_h, uint16_t *out_f16) {
    RT_CHECK(bgr && out_f16 && h >= 1 && w >= 1 && stride_bytes >= 3 * w && in_w >= 1 && in_h >= 1, RTMODT_E_INVALID, "bad argument");
    RT_HIP(hipSetDevice(device));
    struct Bufs {
        std::vector<void *> p;
        ~Bufs() { for (void *q : p) hipFree(q); }
        int get(size_t bytes, void **o) { RT_HIP(hipMalloc(o, bytes)); p.push_back(*o); return RTMODT_OK; }
    } bufs;
    uint8_t *dimg; f16 *dout; int32_t *dtab;
    size_t per = (size_t)(in_h + 2) * (in_w + 2) * 4;
    RT_TRY(bufs.get((size_t)h * stride_bytes, (void **)&dimg));
    RT_TRY(bufs.get(per * 2, (void **)&dout));
    RT_HIP(hipMemset(dout, 0, per * 2));
    RT_HIP(hipMemcpy(dimg, bgr, (size_t)h * stride_bytes, hipMemcpyHostToDevice));
    FramePtrs fp{};
    fp.p[0] = dimg;
    LbHost g = letterbox_geometry(h, w, in_h, in_w);
    ResizeTables tabs{};
    if (g.resize) {
        std::vector<int32_t> t[6];
        build_resize_tables(g.new_w, w, t[0], t[1], t[2]);
        build_resize_tables(g.new_h, h, t[3], t[4], t[5]);
        RT_TRY(bufs.get((size_t)(g.new_w + g.new_h) * 3 * 4, (void **)&dtab));
        const int32_t *dst[6]; int32_t *p = dtab;
        for (int k = 0; k < 6; ++k) {
            RT_HIP(hipMemcpy(p, t[k].data(), t[k].size() * 4, hipMemcpyHostToDevice));
            dst[k] = p; p += t[k].size();
        }
        tabs = ResizeTables{dst[0], dst[1], dst[2], dst[3], dst[4], dst[5]};
    }
    LetterboxGeom lg{h, w, g.new_w, g.new_h, g.top, g.left, g.resize};
    TensorView img; img.base = dout; img.H = in_h; img.W = in_w; img.C = 4; img.pad = 1; img.c = 4;
    RT_TRY(launch_letterbox(fp, stride_bytes, lg, tabs, img, 1, nullptr));
    RT_HIP(hipDeviceSynchronize());
    std::vector<uint16_t> tmp(per);
    RT_HIP(hipMemcpy(tmp.data(), dout, per * 2, hipMemcpyDeviceToHost));
    for (int y = 0; y < in_h; ++y)
        for (int x = 0; x < in_w; ++x)
            memcpy(out_f16 + ((size_t)y * in_w + x) * 3, &tmp[((size_t)(y + 1) * (in_w + 2) + x + 1) * 4], 6);
    return RTMODT_OK;
}

}  // extern "C"
