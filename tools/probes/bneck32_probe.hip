// bneck32_probe.hip -- development harness of csrc/bneck32.hip: the c = 32 Bottleneck + C2f.cv2 tail as persistent workgroups against bottleneck_fused<32, 16, 16, 4>,
// bit for bit, and timed.   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DRTMODT_STAMP] -o tools/probes/bin/bneck32_probe tools/probes/bneck32_probe.hip
// Usage: bneck32_probe [B=32] [iters=20]   |   bneck32_probe fuzz [cases=60] [seed]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../real-time-multi-object-detection---tracking-system_amd/csrc/bottleneck.hip"
#include "../../real-time-multi-object-detection---tracking-system_amd/csrc/bneck32.hip"

namespace rtmodt {
#ifdef RTMODT_STAMP
__device__ unsigned long long *g_stamps;
#endif
std::string &last_error() { static std::string e; return e; }
int fail(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; }
void note_bad_option(const char *) {}
}
using namespace rtmodt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define RT(x) do { int r_ = (x); if (r_ != RTMODT_OK) { printf("%s -> %d\n", #x, r_); exit(1); } } while (0)
static unsigned rng_state = 4242;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xFFFF) / 65536.f - 0.5f; }
template <typename T> static T *dev(const std::vector<T> &h) { T *p; CK(hipMalloc(&p, h.size() * sizeof(T))); CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, iters = argc > 2 ? atoi(argv[2]) : 20;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef RTMODT_STAMP
    unsigned long long *ds; CK(hipMalloc(&ds, (size_t)(1 << 15) * 16 * 8)); CK(hipMemset(ds, 0, (size_t)(1 << 15) * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ds, sizeof(ds)));
#endif
    std::vector<f16> hw1(128 * 288), hw2(128 * 288), hwt(128 * 96);
    for (auto &v : hw1) v = (f16)(0.4f * frand());
    for (auto &v : hw2) v = (f16)(0.4f * frand());
    for (auto &v : hwt) v = (f16)(0.6f * frand());
    std::vector<float> hb1(128), hb2(128), hbt(128);
    for (auto *b : {&hb1, &hb2, &hbt}) for (auto &v : *b) v = 0.3f * frand();
    f16 *w1 = dev(hw1), *w2 = dev(hw2), *wt = dev(hwt); float *b1 = dev(hb1), *b2 = dev(hb2), *bt = dev(hbt);
    std::vector<f16> z(128, (f16)0.f); f16 *zeros = dev(z);
    int bad = 0;
    if (argc > 1 && !strcmp(argv[1], "fuzz")) {
        // fuzz N [seed]: random maps (H % 8 == 0, W % 16 == 0 up to 176 x 208), 1-6 images, the concat tensor 96-160 channels wide with the module's slices at a random
        // 8-aligned offset, the output a 64-channel slice of a wider tensor whose other channels must come back untouched -- persistent kernel against bottleneck_fused, bit for bit
        const int N = argc > 2 ? atoi(argv[2]) : 60;
        if (argc > 3) rng_state = (unsigned)atoi(argv[3]);
        auto rnd = [&](int n) { rng_state = rng_state * 1664525u + 1013904223u; return (int)((rng_state >> 10) % (unsigned)n); };
        int ran = 0;
        for (int k = 0; k < N; ++k) {
            const int H = 8 * (1 + rnd(22)), W = 16 * (1 + rnd(13)), Bn = 1 + rnd(6), coff = 8 * rnd(5), C = coff + 96 + 8 * rnd(5), ooff = 8 * rnd(4), OC = ooff + 64 + 8 * rnd(3);
            const bool shortcut = rnd(4) != 0;
            const size_t per = (size_t)(H + 2) * (W + 2), ncat = per * Bn * C, nout = per * Bn * OC;
            std::vector<f16> hcat(ncat, (f16)0.f), hguard(nout);
            for (int b = 0; b < Bn; ++b)
                for (int y = 1; y <= H; ++y)
                    for (int x = 1; x <= W; ++x) { f16 *px = &hcat[((size_t)(b * (H + 2) + y) * (W + 2) + x) * C]; for (int c2 = 0; c2 < C; ++c2) px[c2] = (f16)(2.f * frand()); }
            for (auto &v : hguard) v = (f16)(1000.f + (float)rnd(7));
            f16 *cat_o = dev(hcat), *cat_n = dev(hcat), *ref = dev(hguard), *out = dev(hguard);
            auto view = [&](f16 *base, int Ct, int co, int cc) { TensorView v; v.base = base; v.H = H; v.W = W; v.C = Ct; v.pad = 1; v.coff = co; v.c = cc; return v; };
            auto mk = [&](f16 *cat, f16 *dst, int persistent) {
                BottleneckLaunch l;
                l.in = view(cat, C, coff + 32, 32); l.out = view(cat, C, coff + 64, 32); if (shortcut) l.res = view(cat, C, coff + 32, 32);
                l.w1 = w1; l.w2 = w2; l.b1 = b1; l.b2 = b2; l.zeros = zeros; l.B = Bn; l.c = 32; l.kp = 288;
                l.tail_in = view(cat, C, coff, 64); l.tail_wt = wt; l.tail_bias = bt; l.tail_cout = 64; l.tail_kp = 96; l.tail_act = 1;
                l.tail_out = view(dst, OC, ooff, 64); l.persistent32 = persistent;
                return l;
            };
            const BottleneckLaunch lo = mk(cat_o, ref, 0), ln = mk(cat_n, out, 1);
            if (!bottleneck32_tail_supported(ln)) { printf("case %d %dx%d B %d: not supported\n", k, H, W, Bn); ++bad; continue; }
            RT(launch_bottleneck(lo, st)); RT(launch_bottleneck(ln, st));
            CK(hipStreamSynchronize(st));
            std::vector<f16> hr(nout), ho(nout), co(ncat), cn(ncat);
            CK(hipMemcpy(hr.data(), ref, nout * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho.data(), out, nout * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(co.data(), cat_o, ncat * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(cn.data(), cat_n, ncat * 2, hipMemcpyDeviceToHost));
            size_t diff = 0, guard = 0, nz = 0;
            for (size_t i = 0; i < nout; ++i) {
                const size_t ch = i % OC, pix = (i / OC) % per, y = pix / (W + 2), x = pix % (W + 2);
                const bool inside = ch >= (size_t)ooff && ch < (size_t)ooff + 64 && y >= 1 && y <= (size_t)H && x >= 1 && x <= (size_t)W;
                if (memcmp(&hr[i], &ho[i], 2) != 0) ++diff;
                if (!inside && memcmp(&ho[i], &hguard[i], 2) != 0) ++guard;
                nz += inside && (float)ho[i] != 0.f;
            }
            const size_t cdiff = memcmp(co.data(), cn.data(), ncat * 2) != 0;      // the concat tensor too: both forms write the Bottleneck's output slice into it
            if (diff || guard || cdiff || !nz) { ++bad; printf("case %d: %dx%d B %d C %d+%d out %d+%d shortcut %d: %zu halves differ, %zu guard halves written, concat tensor %s, %zu non-zero\n", k, H, W, Bn, coff, C, ooff, OC, (int)shortcut, diff, guard, cdiff ? "DIFFERS" : "equal", nz); }
            ++ran;
            CK(hipFree(cat_o)); CK(hipFree(cat_n)); CK(hipFree(ref)); CK(hipFree(out));
        }
        printf("bneck32 fuzz: %d cases run, %d FAILURES\n", ran, bad);
        return bad ? 1 : 0;
    }
    struct Case { int H, W; bool shortcut; };
    for (const Case &c : {Case{160, 160, true}, Case{160, 160, false}, Case{96, 160, true}, Case{80, 80, true}, Case{8, 16, true}}) {
        const int H = c.H, W = c.W, C = 96;
        const size_t per = (size_t)(H + 2) * (W + 2), ncat = per * B * C, nout = per * B * 64;
        std::vector<f16> hcat(ncat, (f16)0.f);
        for (int b = 0; b < B; ++b)
            for (int y = 1; y <= H; ++y)
                for (int x = 1; x <= W; ++x) { f16 *px = &hcat[((size_t)(b * (H + 2) + y) * (W + 2) + x) * C]; for (int k = 0; k < 64; ++k) px[k] = (f16)(2.f * frand()); }
        f16 *cat = dev(hcat), *ref, *out;
        CK(hipMalloc(&ref, nout * 2)); CK(hipMalloc(&out, nout * 2)); CK(hipMemset(ref, 0, nout * 2)); CK(hipMemset(out, 0, nout * 2));
        auto view = [&](f16 *base, int Ct, int coff, int cc) { TensorView v; v.base = base; v.H = H; v.W = W; v.C = Ct; v.pad = 1; v.coff = coff; v.c = cc; return v; };
        BottleneckLaunch l;
        l.in = view(cat, C, 32, 32); l.out = view(cat, C, 64, 32); if (c.shortcut) l.res = view(cat, C, 32, 32);
        l.w1 = w1; l.w2 = w2; l.b1 = b1; l.b2 = b2; l.zeros = zeros; l.B = B; l.c = 32; l.kp = 288;
        l.tail_in = view(cat, C, 0, 64); l.tail_wt = wt; l.tail_bias = bt; l.tail_cout = 64; l.tail_kp = 96; l.tail_act = 1;
        BottleneckLaunch lo = l, ln = l;
        lo.tail_out = view(ref, 64, 0, 64); lo.persistent32 = 0;
        ln.tail_out = view(out, 64, 0, 64); ln.persistent32 = 1;
        if (!bottleneck32_tail_supported(ln)) { printf("%dx%d: not supported\n", H, W); bad++; continue; }
        RT(launch_bottleneck(lo, st)); RT(launch_bottleneck(ln, st));
        CK(hipStreamSynchronize(st));
        std::vector<f16> hr(nout), ho(nout);
        CK(hipMemcpy(hr.data(), ref, nout * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho.data(), out, nout * 2, hipMemcpyDeviceToHost));
        size_t diff = 0, nz = 0, first = (size_t)-1; double maxd = 0;
        for (size_t i = 0; i < nout; ++i) {
            if (memcmp(&hr[i], &ho[i], 2) != 0) { if (!diff) first = i; ++diff; maxd = std::max(maxd, (double)std::fabs((float)hr[i] - (float)ho[i])); }
            nz += (float)hr[i] != 0.f;
        }
        auto tm = [&](const BottleneckLaunch &x) { RT(launch_bottleneck(x, st)); CK(hipEventRecord(e0, st)); for (int i = 0; i < iters; ++i) RT(launch_bottleneck(x, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f / iters; };
        const float us_old = tm(lo), us_new = tm(ln);
        printf("%3dx%3d B %2d shortcut %d: %zu / %zu halves differ (max |d| %.4g; %zu non-zero)   bottleneck_fused %7.1f us   persistent %7.1f us\n", H, W, B, (int)c.shortcut, diff, nout, maxd, nz, us_old, us_new);
        if (diff) { const size_t i = first, pp = (size_t)(W + 2) * 64; printf("    first difference: image %zu row %zu col %zu channel %zu: %g vs %g\n", i / (per * 64), (i % (per * 64)) / pp, (i % pp) / 64, i % 64, (float)hr[i], (float)ho[i]); }
        bad += diff != 0 || nz == 0;
#ifdef RTMODT_STAMP
        {
            const int G = 512;
            CK(hipStreamSynchronize(st)); CK(hipMemset(ds, 0, (size_t)G * 16 * 8));
            RT(launch_bottleneck(ln, st)); CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> hs((size_t)G * 16);
            CK(hipMemcpy(hs.data(), ds, hs.size() * 8, hipMemcpyDeviceToHost));
            double sum[9] = {0}; int cnt = 0;
            for (int g = 0; g < G; ++g) { const unsigned long long *r = &hs[(size_t)g * 16]; if (!r[8] || !r[0]) continue; for (int k = 1; k <= 8; ++k) sum[k] += (double)r[k] - (double)r[k - 1]; ++cnt; }
            if (cnt) printf("    stamps (%d wgs): wait+barrier %.0f | issue+conv1 %.0f | barrier %.0f | conv2 %.0f | barrier %.0f | cv2 mfma %.0f | cv2 epilogue %.0f | barrier+stores %.0f clk; total %.0f\n", cnt,
                            sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt, sum[7] / cnt, sum[8] / cnt, (sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[7] + sum[8]) / cnt);
        }
#endif
        CK(hipFree(cat)); CK(hipFree(ref)); CK(hipFree(out));
    }
    printf(bad ? "FAILED\n" : "all bit-identical\n");
    return bad ? 1 : 0;
}
