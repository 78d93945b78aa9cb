// kernel_probe.hip -- DIAGNOSTIC build of single kernels of librtmodt_hip.so with in-kernel phase stamps (s_memtime by lane 0
// of every workgroup, -DRTMODT_STAMP) on synthetic tensors of the benchmarked shapes; prints per-phase cycle shares and the
// workgroup timeline.  Read its SHARES, never its run time (the stamps fence the scheduler).  The modes that timed kernels the
// round-4 prune removed (pf, ws, l1lines) went with them: their results are in profiles/r03/.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRTMODT_STAMP -o /tmp/kernel_probe tools/probes/kernel_probe.hip && /tmp/kernel_probe
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../real-time-multi-object-detection---tracking-system_amd/csrc/bottleneck.hip"
#include "../../real-time-multi-object-detection---tracking-system_amd/csrc/conv.hip"
#include "../../real-time-multi-object-detection---tracking-system_amd/csrc/conv_pp.hip"

namespace rtmodt {
__device__ unsigned long long *g_stamps;
std::string &last_error() { static std::string e; return e; }
int fail(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; }
void note_bad_option(const char *) {}
}
using namespace rtmodt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run_bneck(int c, int HW, int B, bool tail) {
    const int C = 3 * c;                                        // the C2f concat tensor: [y0 | y1 | y2]
    const size_t per = (size_t)(HW + 2) * (HW + 2);
    f16 *cat, *outt, *w1, *w2, *wt, *zeros; float *b1, *b2, *bt;
    CK(hipMalloc(&cat, per * B * C * 2)); CK(hipMemset(cat, 0, per * B * C * 2));
    CK(hipMalloc(&outt, per * B * 2 * c * 2)); CK(hipMemset(outt, 0, per * B * 2 * c * 2));
    std::vector<f16> hw((size_t)128 * 9 * c);
    for (auto &v : hw) v = (f16)(((rand() % 200) - 100) * 1e-3f);
    CK(hipMalloc(&w1, hw.size() * 2)); CK(hipMemcpy(w1, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&w2, hw.size() * 2)); CK(hipMemcpy(w2, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&wt, (size_t)128 * 3 * c * 2)); CK(hipMemcpy(wt, hw.data(), (size_t)128 * 3 * c * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&zeros, 256)); CK(hipMemset(zeros, 0, 256));
    CK(hipMalloc(&b1, 512)); CK(hipMemset(b1, 0, 512)); CK(hipMalloc(&b2, 512)); CK(hipMemset(b2, 0, 512)); CK(hipMalloc(&bt, 512)); CK(hipMemset(bt, 0, 512));
    {   // random activations
        std::vector<f16> h(per * B * C);
        for (auto &v : h) v = (f16)(((rand() % 2000) - 1000) * 1e-3f);
        CK(hipMemcpy(cat, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    BottleneckLaunch l;
    auto view = [&](f16 *base, int Ct, int coff, int cc) { TensorView v; v.base = base; v.H = v.W = HW; v.C = Ct; v.pad = 1; v.coff = coff; v.c = cc; return v; };
    l.in = view(cat, C, c, c); l.out = view(cat, C, 2 * c, c); l.res = view(cat, C, c, c);
    l.w1 = w1; l.w2 = w2; l.b1 = b1; l.b2 = b2; l.zeros = zeros; l.B = B; l.c = c; l.kp = 9 * c;
    if (tail) { l.tail_in = view(cat, C, 0, 2 * c); l.tail_out = view(outt, 2 * c, 0, 2 * c); l.tail_wt = wt; l.tail_bias = bt; l.tail_cout = 2 * c; l.tail_kp = 3 * c; l.tail_act = 1; }
    const int tiles = ((HW + 15) / 16) * ((HW + 15) / 16) * B;
    unsigned long long *d_st;
    CK(hipMalloc(&d_st, (size_t)tiles * 16 * 8)); CK(hipMemset(d_st, 0, (size_t)tiles * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (launch_bottleneck(l, nullptr) != 0) return 1;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 5; ++i) launch_bottleneck(l, nullptr);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st((size_t)tiles * 16);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    printf("bottleneck c=%d %dx%d B=%d tail=%d: %.1f us per launch (stamped build), %d workgroups\n", c, HW, HW, B, (int)tail, ms * 1e3 / 5, tiles);
    const char *names[] = {"patch DMA issue", "wait patch (+barrier)", "conv1 k-loop", "epilogue 1 + barrier", "conv2 k-loop", "epilogue 2 staging + barrier", "tail: DMA issue + wait + barrier", "tail MFMA + barrier", "tail store / plain store"};
    const int lastk = 9;
    std::vector<std::vector<double>> d(lastk);
    unsigned long long t0 = ~0ull, t1 = 0;
    std::vector<double> life;
    for (int g = 0; g < tiles; ++g) {
        const unsigned long long *s = &st[(size_t)g * 16];
        unsigned long long prev = s[0];
        for (int k = 1; k <= lastk; ++k) { if (!s[k]) continue; d[k - 1].push_back((double)(s[k] - prev)); prev = s[k]; }
        t0 = std::min(t0, s[0]); t1 = std::max(t1, s[lastk]);
        life.push_back((double)(s[lastk] - s[0]));
    }
    auto med = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    double tot = med(life);
    printf("  kernel span %llu clk (s_memtime ticks); median workgroup lifetime %.0f clk\n", t1 - t0, tot);
    for (int k = 0; k < lastk; ++k) printf("  %-36s median %8.0f clk  %5.1f %%\n", names[k], med(d[k]), 100.0 * med(d[k]) / tot);
    // concurrency: workgroups alive at the middle of the kernel
    const unsigned long long mid = t0 + (t1 - t0) / 2;
    int alive = 0;
    for (int g = 0; g < tiles; ++g) alive += st[(size_t)g * 16] <= mid && st[(size_t)g * 16 + lastk] >= mid;
    printf("  workgroups alive at mid-kernel: %d (%.2f per CU)\n", alive, alive / 256.0);
    return 0;
}

// weight-stationary 1x1 conv (or any tile) on a [B][HW+2][HW+2][cin] tensor
static int run_conv1x1(int tile, int cin, int cout, int HW, int B) {
    const size_t per = (size_t)(HW + 2) * (HW + 2);
    f16 *in, *out, *w; float *bias;
    CK(hipMalloc(&in, per * B * cin * 2)); CK(hipMalloc(&out, per * B * cout * 2));
    CK(hipMemset(out, 0, per * B * cout * 2));
    std::vector<f16> h(per * B * cin);
    for (auto &v : h) v = (f16)(((rand() % 2000) - 1000) * 1e-3f);
    CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    std::vector<f16> hw((size_t)((cout + 127) / 128 * 128) * cin);
    for (auto &v : hw) v = (f16)(((rand() % 200) - 100) * 1e-3f);
    CK(hipMalloc(&w, hw.size() * 2)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&bias, 4096)); CK(hipMemset(bias, 0, 4096));
    ConvLaunch c;
    auto view = [&](f16 *base, int Ct) { TensorView v; v.base = base; v.H = v.W = HW; v.C = Ct; v.pad = 1; v.coff = 0; v.c = Ct; return v; };
    c.in = view(in, cin); c.out = view(out, cout); c.wt = w; c.bias = bias; c.B = B; c.cin = cin; c.cout = cout; c.ks = 1; c.stride = 1; c.act = 1; c.kp = cin; c.tile = tile;
    const int wgs = 1024;
    unsigned long long *d_st;
    CK(hipMalloc(&d_st, (size_t)wgs * 16 * 8)); CK(hipMemset(d_st, 0, (size_t)wgs * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (launch_conv(c, nullptr) != 0) return 1;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 5; ++i) launch_conv(c, nullptr);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("conv 1x1 %d -> %d, %dx%d x %d, tile %s: %.1f us per launch (stamped build)\n", cin, cout, HW, HW, B, tile_name(tile), ms * 1e3 / 5);
    if (tile_is_pt(tile)) {
        std::vector<unsigned long long> st((size_t)wgs * 16);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        const char *nm[] = {"issue weights + ring prologue", "wait for weights + stage 0 (+barrier)", "step 0", "step 1", "step 2", "step 3", "rest of tile 0 incl. epilogue", "", "remaining tiles"};
        const int idx[] = {0, 1, 2, 3, 4, 5, 6, 7, 9};
        for (int k = 1; k < 9; ++k) {
            std::vector<double> d;
            for (int g = 0; g < 256; ++g) { const unsigned long long *s = &st[(size_t)g * 16]; if (s[idx[k]] && s[idx[k - 1]]) d.push_back((double)(s[idx[k]] - s[idx[k - 1]])); }
            if (d.empty()) continue;
            std::sort(d.begin(), d.end());
            printf("  %-40s median %8.0f clk\n", nm[k - 1], d[d.size() / 2]);
        }
    }
    if (tile_needs_cin64(tile) && !tile_is_pt(tile) && !tile_is_pp(tile) && !tile_is_ppt(tile) && !tile_is_rows(tile)) {      // conv_mfma64_body: 0 = loop entry, 1..5 = k-steps 0..4 (after the barrier), 6 = loop exit, 9 = end
        std::vector<unsigned long long> st((size_t)wgs * 16);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        const char *nm[] = {"prologue -> step 0 ready", "step 0", "step 1", "step 2", "step 3", "rest of the k-loop", "epilogue: everybody out of the loop", "epilogue: bias + SiLU + LDS staging", "epilogue: barrier"};
        const int idx[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
        for (int k = 1; k < 10; ++k) {
            std::vector<double> d;
            for (int g = 0; g < 256; ++g) { const unsigned long long *s = &st[(size_t)g * 16]; if (s[idx[k]] && s[idx[k - 1]]) d.push_back((double)(s[idx[k]] - s[idx[k - 1]])); }
            if (d.empty()) continue;
            std::sort(d.begin(), d.end());
            printf("  %-40s median %8.0f clk\n", nm[k - 1], d[d.size() / 2]);
        }
        // inside step 2 (plain loop): 10 = top of the iteration, 3 = behind its barrier, 11 = behind its DMA issue, 4 = behind the next barrier
        const int pa[][2] = {{10, 3}, {3, 11}, {11, 4}};
        const char *pn[] = {"step 2: wait for the stage + barrier", "step 2: issue of the DMA pieces", "step 2: fragment reads + MFMAs + next wait"};
        for (int k = 0; k < 3; ++k) {
            std::vector<double> d;
            for (int g = 0; g < 256; ++g) { const unsigned long long *s = &st[(size_t)g * 16]; if (s[pa[k][0]] && s[pa[k][1]]) d.push_back((double)(s[pa[k][1]] - s[pa[k][0]])); }
            if (d.empty()) continue;
            std::sort(d.begin(), d.end());
            printf("  %-40s median %8.0f clk\n", pn[k], d[d.size() / 2]);
        }
    }
    hipFree(in); hipFree(out); hipFree(w); hipFree(bias); hipFree(d_st);
    return 0;
}

static int run_stem(int B) {
    const int H = 640, W = 640, Ho = 320, Wo = 320, cout = 32;
    uint8_t *frames; f16 *out, *w; float *bias;
    CK(hipMalloc(&frames, (size_t)B * H * W * 3));
    std::vector<uint8_t> hf((size_t)B * H * W * 3);
    for (auto &v : hf) v = (uint8_t)(rand() & 255);
    CK(hipMemcpy(frames, hf.data(), hf.size(), hipMemcpyHostToDevice));
    const size_t per = (size_t)(Ho + 2) * (Wo + 2);
    CK(hipMalloc(&out, per * B * cout * 2)); CK(hipMemset(out, 0, per * B * cout * 2));
    std::vector<f16> hw((size_t)cout * 64);
    for (auto &v : hw) v = (f16)(((rand() % 200) - 100) * 1e-2f);
    CK(hipMalloc(&w, hw.size() * 2)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&bias, 4096)); CK(hipMemset(bias, 0, 4096));
    FramePtrs fp{};
    for (int b = 0; b < B; ++b) fp.p[b] = frames + (size_t)b * H * W * 3;
    LetterboxGeom g{H, W, W, H, 0, 0, 0};
    TensorView o; o.base = out; o.H = Ho; o.W = Wo; o.C = cout; o.pad = 1; o.coff = 0; o.c = cout;
    const int wgs = B * Ho;
    unsigned long long *d_st;
    CK(hipMalloc(&d_st, (size_t)wgs * 16 * 8)); CK(hipMemset(d_st, 0, (size_t)wgs * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (launch_stem_fused(fp, 0, W * 3, g, H, W, nullptr, o, w, bias, B, cout, nullptr) != 0) return 1;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 5; ++i) launch_stem_fused(fp, 0, W * 3, g, H, W, nullptr, o, w, bias, B, cout, nullptr);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stem_fused %d frames: %.1f us per launch (stamped build), %d workgroups\n", B, ms * 1e3 / 5, wgs);
    std::vector<unsigned long long> st((size_t)wgs * 16);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    const char *nm[] = {"rows: loads + shift + LDS write", "barrier", "bytes -> fp16 pixels", "barrier", "MFMA + SiLU + transpose + store"};
    for (int k = 1; k < 6; ++k) {
        std::vector<double> d;
        for (int g2 = 0; g2 < wgs; ++g2) { const unsigned long long *s = &st[(size_t)g2 * 16]; if (s[k] && s[k - 1]) d.push_back((double)(s[k] - s[k - 1])); }
        std::sort(d.begin(), d.end());
        printf("  %-40s median %8.0f clk\n", nm[k - 1], d[d.size() / 2]);
    }
    {   std::vector<double> d; unsigned long long t0 = ~0ull, t1 = 0;
        for (int g2 = 0; g2 < wgs; ++g2) { const unsigned long long *s = &st[(size_t)g2 * 16]; d.push_back((double)(s[5] - s[0])); t0 = std::min(t0, s[0]); t1 = std::max(t1, s[5]); }
        std::sort(d.begin(), d.end());
        printf("  workgroup life median %.0f clk; launch span %.0f clk (s_memtime ticks at 100 MHz: x clock ratio)\n", d[d.size() / 2], (double)(t1 - t0));
    }
    return 0;
}

// a 3x3 conv (optionally with the fused 1x1 tail) on the 4-wave tile kernel conv_mfma_body: phase stamps 0..6
static int run_conv3(int tile, int cin, int cout, int stride, int HWin, int B, int tail_cout, int Win = 0) {
    if (Win == 0) Win = HWin;                                  // (non-square only for the surrogate experiments)
    const int HWo = (HWin - 1) / stride + 1, Wo = (Win - 1) / stride + 1;
    const size_t per_in = (size_t)(HWin + 2) * (Win + 2), per_out = (size_t)(HWo + 2) * (Wo + 2);
    f16 *in, *out, *w, *wt; float *bias;
    CK(hipMalloc(&in, per_in * B * cin * 2)); CK(hipMalloc(&out, per_out * B * std::max(cout, tail_cout) * 2));
    CK(hipMemset(out, 0, per_out * B * std::max(cout, tail_cout) * 2));
    std::vector<f16> h(per_in * B * cin);
    for (auto &v : h) v = (f16)(((rand() % 2000) - 1000) * 1e-3f);
    CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const int kp = 9 * cin;
    std::vector<f16> hw((size_t)128 * kp);
    for (auto &v : hw) v = (f16)(((rand() % 200) - 100) * 1e-3f);
    CK(hipMalloc(&w, hw.size() * 2)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&wt, (size_t)128 * 128 * 2)); CK(hipMemcpy(wt, hw.data(), (size_t)128 * 128 * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&bias, 4096)); CK(hipMemset(bias, 0, 4096));
    ConvLaunch c;
    auto view = [&](f16 *base, int Hh, int Ww, int Ct) { TensorView v; v.base = base; v.H = Hh; v.W = Ww; v.C = Ct; v.pad = 1; v.coff = 0; v.c = Ct; return v; };
    c.in = view(in, HWin, Win, cin); c.wt = w; c.bias = bias; c.B = B; c.cin = cin; c.cout = cout; c.ks = 3; c.stride = stride; c.act = 1; c.kp = kp; c.tile = tile;
    if (tail_cout) { c.out = view(out, HWo, Wo, cout); c.tail_out = view(out, HWo, Wo, tail_cout); c.tail_wt = wt; c.tail_bias = bias; c.tail_cout = tail_cout; c.tail_kp = cout; }
    else c.out = view(out, HWo, Wo, cout);
    const int wgs = 65536;
    unsigned long long *d_st;
    CK(hipMalloc(&d_st, (size_t)wgs * 16 * 8)); CK(hipMemset(d_st, 0, (size_t)wgs * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (launch_conv(c, nullptr) != 0) return 1;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 5; ++i) launch_conv(c, nullptr);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("conv 3x3/s%d %d -> %d (tail %d), %dx%d x %d, tile %s: %.1f us per launch (stamped build)\n", stride, cin, cout, tail_cout, HWin, Win, B, tile_name(tile), ms * 1e3 / 5);
    std::vector<unsigned long long> st((size_t)wgs * 16);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    const char *nm[] = {"prologue DMA issue", "wait step 0 (+barrier)", "step 0 -> barrier of step 1", "step 1 -> barrier of step 2", "rest of the k-loop", "epilogue (tail GEMM + stores)"};
    printf("  (k-steps: %d of 32, or %d super-steps of the tap-reuse kernel at 32 deep / %d at 64 deep)\n", kp / 32, 3 * (cin / 32), 3 * (cin / 64));
    int live = 0;
    for (int k = 1; k < 7; ++k) {
        std::vector<double> d;
        for (int g = 0; g < wgs; ++g) { const unsigned long long *s2 = &st[(size_t)g * 16]; if (s2[k] && s2[k - 1]) d.push_back((double)(s2[k] - s2[k - 1])); }
        if (d.empty()) continue;
        live = (int)d.size();
        std::sort(d.begin(), d.end());
        printf("  %-40s median %8.0f clk\n", nm[k - 1], d[d.size() / 2]);
    }
    {   const int seq[] = {5, 7, 8, 9, 6};
        const char *en[] = {"  epilogue: barrier", "  epilogue: bias + SiLU -> LDS", "  epilogue: barrier", "  epilogue: LDS -> 16-byte stores"};
        for (int k = 1; k < 5; ++k) {
            std::vector<double> d;
            for (int g = 0; g < wgs; ++g) { const unsigned long long *s2 = &st[(size_t)g * 16]; if (s2[seq[k]] && s2[seq[k - 1]]) d.push_back((double)(s2[seq[k]] - s2[seq[k - 1]])); }
            if (d.empty()) continue;
            std::sort(d.begin(), d.end());
            printf("  %-40s median %8.0f clk\n", en[k - 1], d[d.size() / 2]);
        }
    }
    {   std::vector<double> d;
        for (int g = 0; g < wgs; ++g) { const unsigned long long *s2 = &st[(size_t)g * 16]; if (s2[6] && s2[0]) d.push_back((double)(s2[6] - s2[0])); }
        std::sort(d.begin(), d.end());
        if (!d.empty()) printf("  workgroup life median %.0f clk, %d workgroups stamped\n", d[d.size() / 2], live);
        {   // in-kernel shader clock: s_memtime ticks per s_memrealtime tick (100 MHz), per workgroup, median
            std::vector<double> ghz;
            for (int gidx = 0; gidx < wgs; ++gidx) {
                const unsigned long long *s = &st[(size_t)gidx * 16];
                if (s[15] > s[13] && s[14] > s[12]) ghz.push_back((double)(s[14] - s[12]) / (double)(s[15] - s[13]) * 0.1);
            }
            std::sort(ghz.begin(), ghz.end());
            if (!ghz.empty()) printf("  in-kernel shader clock (s_memtime / s_memrealtime): median %.3f GHz (min %.3f, max %.3f)\n", ghz[ghz.size() / 2], ghz.front(), ghz.back());
        }
    }
    hipFree(in); hipFree(out); hipFree(w); hipFree(wt); hipFree(bias); hipFree(d_st);
    return 0;
}

// ---- energy per launch: one tile configuration run back to back for `secs` seconds while a host thread samples the GPU's hwmon
// package power and clock every 20 ms (the staged bench runs at the package power limit: of two tiles that take the same time
// the one that draws less is the better one there)
#include <thread>
#include <atomic>
#include <chrono>
#include <glob.h>
static std::string hwmon_dir() {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0) != hipSuccess) return "";
    for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
    std::string pat = std::string("/sys/bus/pci/devices/") + bdf + "/hwmon/hwmon*";
    glob_t g; std::string out;
    if (glob(pat.c_str(), 0, nullptr, &g) == 0 && g.gl_pathc > 0) out = g.gl_pathv[0];
    globfree(&g);
    return out;
}
static long read_long(const std::string &path) { FILE *f = fopen(path.c_str(), "r"); if (!f) return -1; long v = -1; if (fscanf(f, "%ld", &v) != 1) v = -1; fclose(f); return v; }
static int run_energy(int tile, int ks, int cin, int cout, int HW, int B, double secs) {
    const size_t per = (size_t)(HW + 2) * (HW + 2);
    f16 *in, *out, *w; float *bias;
    CK(hipMalloc(&in, per * B * cin * 2)); CK(hipMalloc(&out, per * B * cout * 2)); CK(hipMemset(out, 0, per * B * cout * 2));
    {   std::vector<f16> h(per * B * cin);
        for (auto &v : h) v = (f16)(((rand() % 2000) - 1000) * 1e-3f);
        CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice)); }
    const int kp = ks * ks * cin;
    std::vector<f16> hw((size_t)((cout + 127) / 128 * 128) * kp);
    for (auto &v : hw) v = (f16)(((rand() % 200) - 100) * 1e-3f);
    CK(hipMalloc(&w, hw.size() * 2)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&bias, 4096)); CK(hipMemset(bias, 0, 4096));
    unsigned long long *d_st;
    CK(hipMalloc(&d_st, (size_t)65536 * 16 * 8)); CK(hipMemset(d_st, 0, (size_t)65536 * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
    ConvLaunch c;
    auto view = [&](f16 *base, int Ct) { TensorView v; v.base = base; v.H = v.W = HW; v.C = Ct; v.pad = 1; v.coff = 0; v.c = Ct; return v; };
    c.in = view(in, cin); c.out = view(out, cout); c.wt = w; c.bias = bias; c.B = B; c.cin = cin; c.cout = cout; c.ks = ks; c.stride = 1; c.act = 1; c.kp = kp; c.tile = tile;
    if (launch_conv(c, nullptr) != 0) { printf("  (tile %s is not legal for this conv)\n", tile_name(tile)); hipFree(in); hipFree(out); hipFree(w); hipFree(bias); hipFree(d_st); return 0; }
    CK(hipDeviceSynchronize());
    const std::string dir = hwmon_dir();
    std::atomic<bool> stop{false};
    std::vector<long> pw, fq;
    std::thread sampler([&]() {
        while (!stop.load()) {
            if (!dir.empty()) { pw.push_back(read_long(dir + "/power1_input")); fq.push_back(read_long(dir + "/freq1_input")); }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    });
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    do {
        for (int i = 0; i < 200; ++i) launch_conv(c, nullptr);
        CK(hipDeviceSynchronize());
        launches += 200;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < secs);
    stop.store(true); sampler.join();
    // drop the first 30 % of the samples (ramp of the power reading)
    double p = 0, f = 0; int n = 0;
    for (size_t i = pw.size() * 3 / 10; i < pw.size(); ++i) if (pw[i] > 0 && fq[i] > 0) { p += pw[i] * 1e-6; f += fq[i] * 1e-6; ++n; }
    if (n) { p /= n; f /= n; }
    const double us = el / launches * 1e6;
    printf("%dx%d conv %d -> %d, %dx%d x %d, tile %-18s %7.1f us/launch  %6.0f W  %5.0f MHz  -> %7.1f mJ/launch (above the 290 W of an idle chip: %6.1f mJ)\n", ks, ks, cin, cout, HW, HW, B,
           tile_name(tile), us, p, f, p * us * 1e-3, (p - 290.0) * us * 1e-3);
    hipFree(in); hipFree(out); hipFree(w); hipFree(bias); hipFree(d_st);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "energy")) {
        const double secs = argc > 2 ? atof(argv[2]) : 1.5;
        for (int t : {TILE_K64_128x128_S2_W8, TILE_PT_128x128_S2, TILE_PPT_256x128, TILE_128x128, TILE_128x64, TILE_K64_128x128_S3_W8})
            if (run_energy(t, 1, 512, 256, 40, 32, secs)) return 1;                      // 6.cv2
        for (int t : {TILE_K64_128x128_S2_W8, TILE_PT_128x128_S2, TILE_128x64})
            if (run_energy(t, 1, 256, 128, 80, 32, secs)) return 1;                      // 4.cv2
        for (int t : {TILE_ROWS_256x64_W8, TILE_PP_256x128, TILE_PT_128x128_S2, TILE_K64_128x128_S2_W8})
            if (run_energy(t, 3, 128, 128, 40, 32, secs)) return 1;                      // 6.m / 12.m / 18.m
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "rows")) {                                  // 3x3 stride-1 convs on the tap-reuse kernel (stamps 0..6)
        const int t6[] = {TILE_ROWS_256x64_W8, TILE_ROWS_K64_64x64, TILE_PP_256x128};
        for (int t : t6) if (run_conv3(t, 128, 128, 1, 40, 32, 0)) return 1;      // 6.m / 12.m / 18.m
        for (int t : t6) if (run_conv3(t, 64, 64, 1, 80, 32, 0)) return 1;        // 4.m / 15.m
        for (int t : t6) if (run_conv3(t, 128, 64, 1, 80, 32, 0)) return 1;       // Detect cv2.0 at P3
        for (int t : t6) if (run_conv3(t, 256, 128, 1, 40, 32, 0)) return 1;      // Detect cv3.0 at P4
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "l1")) {
        if (run_conv3(TILE_TAIL_128x64, 32, 64, 2, 320, 32, 64)) return 1;       // layer 1 + 2.cv1 at 32 frames
        if (run_conv3(TILE_128x64, 32, 64, 2, 320, 32, 0)) return 1;             // layer 1 alone
        if (run_conv3(TILE_64x64, 32, 64, 2, 320, 32, 0)) return 1;
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "stem")) return run_stem(argc > 2 ? atoi(argv[2]) : 32);
    if (argc > 1 && !strcmp(argv[1], "pt")) {                                     // persistent tiles against the plain 8-wave tile kernel
        const int shapes[][3] = {{256, 256, 40}, {512, 256, 40}, {384, 128, 80}, {768, 512, 20}};
        for (auto &sh : shapes)
            for (int t : {TILE_K64_128x128_S2_W8, TILE_PT_128x128_S2, TILE_PT_128x64_S2})
                if (run_conv1x1(t, sh[0], sh[1], sh[2], 32)) return 1;
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "ksteps")) {                                 // k-step time of the 64-deep tile kernels by tile shape (stamped build)
        for (int t : {TILE_K64_128x128_S2_W8, TILE_K64_128x128_S3_W8, TILE_K64_256x64_S2_W8, TILE_K64_128x64_S3_W8, TILE_K64_64x64_S3})
            if (run_conv1x1(t, 768, 256, 40, 32)) return 1;
        return 0;
    }
    if (run_bneck(32, 160, 16, true)) return 1;
    if (run_bneck(32, 160, 16, false)) return 1;
    if (run_bneck(64, 80, 16, false)) return 1;
    if (run_bneck(64, 80, 16, true)) return 1;
    return 0;
}
