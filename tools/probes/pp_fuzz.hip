// pp_fuzz.hip -- random-shape parity of the ping-pong kernels (csrc/conv_pp.hip) against the general 64x64 tile of conv.hip, outside the handful of
// shapes YOLOv8 n / s / m produce: non-square maps, channel SLICES of wider tensors for input / output / shortcut, cout not a multiple of the tile,
// tiny and ragged M, activation on / off, grouped launches of unequal problems.  Every output tensor sits between two guard zones that must come back
// untouched (an out-of-range store shows up as a damaged guard, not as a fault somewhere else).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/pp_fuzz tools/probes/pp_fuzz.hip && /tmp/pp_fuzz [cases] [seed] [all]   ("all": every tile of the table)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../real-time-multi-object-detection---tracking-system_amd/csrc/conv.hip"
#include "../../real-time-multi-object-detection---tracking-system_amd/csrc/conv_pp.hip"

namespace rtmodt {
std::string &last_error() { static std::string e; return e; }
int fail(int code, const char *fmt, ...) { char b[512]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof(b), fmt, ap); va_end(ap); last_error() = b; return code; }
void note_bad_option(const char *) {}
}
using namespace rtmodt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static unsigned long long rng = 88172645463325252ull;
static unsigned rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 11); }
static int pick(std::initializer_list<int> l) { return *(l.begin() + rnd() % l.size()); }
static int range(int lo, int hi) { return lo + (int)(rnd() % (unsigned)(hi - lo + 1)); }
static float frand() { return (rnd() & 0xFFFF) / 65536.f - 0.5f; }

constexpr size_t GUARD = 32768;                       // halves on either side of an output tensor
constexpr unsigned short GUARD_BITS = 0x7bcd;

struct Problem {
    int cin, cout, H, W, B, ks, stride, act; bool res;
    int in_C, in_off, out_C, out_off, res_C, res_off;          // the tensors the slices live in
    f16 *in = nullptr, *out_alloc = nullptr, *resbuf = nullptr, *w = nullptr; float *bias = nullptr;
    size_t out_elems = 0;
    ConvLaunch c;
    f16 *out() const { return out_alloc + GUARD; }
};

static void make(Problem &P) {
    const int IH = P.H * P.stride, IW = P.W * P.stride;
    const size_t in_elems = (size_t)P.B * (IH + 2) * (IW + 2) * P.in_C;
    std::vector<f16> h(in_elems, (f16)0.f);
    for (int b = 0; b < P.B; ++b)
        for (int y = 1; y <= IH; ++y)
            for (int x = 1; x <= IW; ++x) {
                f16 *px = &h[(((size_t)b * (IH + 2) + y) * (IW + 2) + x) * P.in_C];
                for (int c = 0; c < P.in_C; ++c) px[c] = (f16)(2.f * frand());       // (channels outside the slice hold data too: reading them would show)
            }
    CK(hipMalloc(&P.in, in_elems * 2 + 256)); CK(hipMemcpy(P.in, h.data(), in_elems * 2, hipMemcpyHostToDevice));
    P.out_elems = (size_t)P.B * (P.H + 2) * (P.W + 2) * P.out_C;
    CK(hipMalloc(&P.out_alloc, (P.out_elems + 2 * GUARD) * 2));
    if (P.res) {
        const size_t re = (size_t)P.B * (P.H + 2) * (P.W + 2) * P.res_C;
        std::vector<f16> hr(re);
        for (auto &v : hr) v = (f16)frand();
        CK(hipMalloc(&P.resbuf, re * 2 + 256)); CK(hipMemcpy(P.resbuf, hr.data(), re * 2, hipMemcpyHostToDevice));
    }
    const int cp = (P.cout + 127) / 128 * 128, K = P.ks * P.ks * P.cin;
    std::vector<f16> hw((size_t)cp * K, (f16)0.f);
    const float sc = 2.0f / std::sqrt((float)K);
    for (int n = 0; n < P.cout; ++n)
        for (int k = 0; k < K; ++k) hw[(size_t)n * K + k] = (f16)(sc * frand());
    CK(hipMalloc(&P.w, hw.size() * 2)); CK(hipMemcpy(P.w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hb(cp, 0.f);
    for (int n = 0; n < P.cout; ++n) hb[n] = 0.5f * frand();
    CK(hipMalloc(&P.bias, hb.size() * 4)); CK(hipMemcpy(P.bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    auto view = [&](f16 *base, int Ct, int off, int c, int hh, int ww) { TensorView v; v.base = base; v.H = hh; v.W = ww; v.C = Ct; v.pad = 1; v.coff = off; v.c = c; return v; };
    ConvLaunch &c = P.c;
    c.in = view(P.in, P.in_C, P.in_off, P.cin, IH, IW); c.out = view(P.out(), P.out_C, P.out_off, P.cout, P.H, P.W);
    if (P.res) c.res = view(P.resbuf, P.res_C, P.res_off, P.cout, P.H, P.W);
    c.wt = P.w; c.bias = P.bias; c.B = P.B; c.cin = P.cin; c.cout = P.cout; c.ks = P.ks; c.stride = P.stride; c.act = P.act; c.kp = K;
}
static void drop(Problem &P) { (void)hipFree(P.in); (void)hipFree(P.out_alloc); (void)hipFree(P.w); (void)hipFree(P.bias); if (P.resbuf) (void)hipFree(P.resbuf); }

// run the problems as one launch of `tile`; outputs (whole tensors, borders and foreign channels included) into `got`; false when the tile refuses
static bool run(std::vector<Problem> &ps, int tile, std::vector<std::vector<f16>> &got, bool &guards_ok) {
    std::vector<ConvLaunch> cl;
    for (auto &p : ps) { p.c.tile = tile; cl.push_back(p.c); }
    for (auto &p : ps) {
        std::vector<unsigned short> init(p.out_elems + 2 * GUARD, GUARD_BITS);
        for (size_t i = 0; i < p.out_elems; ++i) init[GUARD + i] = 0x3c00;       // 1.0 everywhere inside: what a kernel must not touch keeps it
        CK(hipMemcpy(p.out_alloc, init.data(), init.size() * 2, hipMemcpyHostToDevice));
    }
    if (launch_conv_group(cl.data(), (int)cl.size(), tile, nullptr) != 0) return false;
    CK(hipDeviceSynchronize());
    got.clear(); guards_ok = true;
    for (auto &p : ps) {
        std::vector<unsigned short> all(p.out_elems + 2 * GUARD);
        CK(hipMemcpy(all.data(), p.out_alloc, all.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < GUARD; ++i) guards_ok = guards_ok && all[i] == GUARD_BITS && all[GUARD + p.out_elems + i] == GUARD_BITS;
        std::vector<f16> o(p.out_elems);
        memcpy(o.data(), all.data() + GUARD, p.out_elems * 2);
        got.push_back(o);
    }
    return true;
}

int main(int argc, char **argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 100;
    if (argc > 2) rng ^= (unsigned long long)atoll(argv[2]) * 0x9E3779B97F4A7C15ull;
    const bool all_tiles = argc > 3 && !strcmp(argv[3], "all");      // every tile of the table (tails excepted: they need a second conv) instead of the ping-pong ones
    const int pp_tiles[] = {TILE_PP_256x128, TILE_PP_256x64, TILE_PP_256x192, TILE_PP_512x64}, ppt_tiles[] = {TILE_PPT_256x128};
    int failures = 0, ran = 0, refused = 0;
    for (int k = 0; k < cases; ++k) {
        const int kind = (int)(rnd() % 4);              // 0, 1: one 3x3 / s1 conv; 2: a group of 2-3 of them; 3: 1x1 or 3x3 / s2 (ppt)
        const int n = kind == 2 ? range(2, 3) : 1;
        std::vector<Problem> ps(n);
        std::string desc;
        for (auto &P : ps) {
            P.cin = pick({64, 64, 128, 128, 192, 256, 320});
            P.cout = kind == 3 ? pick({8, 24, 64, 72, 128, 136, 192, 256}) : pick({8, 16, 64, 64, 72, 128, 128, 136, 192, 192, 200, 256, 384});
            P.H = range(1, 44); P.W = range(1, 44);
            if (rnd() % 5 == 0) { P.H = range(60, 90); P.W = range(60, 90); }
            P.B = range(1, 6);
            P.ks = kind == 3 ? pick({1, 1, 3}) : 3; P.stride = kind == 3 && P.ks == 3 ? 2 : 1;
            P.act = rnd() % 4 != 0;
            P.res = kind != 3 && rnd() % 3 == 0;
            P.in_off = pick({0, 0, 8, 64}); P.in_C = P.in_off + P.cin + pick({0, 0, 8, 56});
            P.out_off = pick({0, 0, 8, 40}); P.out_C = P.out_off + P.cout + pick({0, 0, 8, 24});
            P.res_off = pick({0, 8}); P.res_C = P.res_off + P.cout + pick({0, 16});
            if (rnd() % 3 != 0 && P.res_off + P.cout <= P.out_C) P.res_C = P.out_C;      // two in three: a slice of a tensor shaped like the output's (the only shortcut the ping-pong tiles take since round 5)
            make(P);
            char b[160];
            snprintf(b, sizeof(b), "[%d->%d %dx%d B%d k%ds%d act%d res%d in %d+%d/%d out %d+%d/%d] ", P.cin, P.cout, P.H, P.W, P.B, P.ks, P.stride, P.act, (int)P.res, P.in_off, P.cin, P.in_C,
                     P.out_off, P.cout, P.out_C);
            desc += b;
        }
        std::vector<std::vector<f16>> ref, got;
        bool g_ok = true;
        if (!run(ps, TILE_64x64, ref, g_ok) || !g_ok) { printf("case %d %s: REFERENCE tile failed (%s)\n", k, desc.c_str(), last_error().c_str()); ++failures; for (auto &p : ps) drop(p); continue; }
        int every[TILE_COUNT], ne = 0;
        for (int t = 0; t < TILE_COUNT; ++t) if (!tile_is_tail(t) && t != TILE_64x64) every[ne++] = t;
        const int *tiles = all_tiles ? every : (kind == 3 ? ppt_tiles : pp_tiles);
        const int nt = all_tiles ? ne : (kind == 3 ? 1 : 4);
        for (int ti = 0; ti < nt; ++ti) {
            if (!run(ps, tiles[ti], got, g_ok)) { ++refused; continue; }
            ++ran;
            size_t bad = 0; double md = 0, mr = 0;
            for (size_t i = 0; i < ps.size(); ++i) {
                for (size_t e = 0; e < ref[i].size(); ++e) mr = std::max(mr, (double)std::fabs((float)ref[i][e]));
                const double tol = 2e-3 * mr + 2e-3;
                for (size_t e = 0; e < ref[i].size(); ++e) {
                    const double d = std::fabs((double)(float)got[i][e] - (double)(float)ref[i][e]);
                    if (!(d <= tol)) ++bad;
                    md = std::max(md, d);
                }
            }
            if (bad || !g_ok) { ++failures; printf("case %d %s tile %s: %zu outside tolerance (max|d| %.4g, max|ref| %.3g), guards %s\n", k, desc.c_str(), tile_name(tiles[ti]), bad, md, mr, g_ok ? "intact" : "DAMAGED"); }
        }
        for (auto &p : ps) drop(p);
        if (k % 20 == 19) { printf("... %d cases, %d tile runs, %d refused, %d failures\n", k + 1, ran, refused, failures); fflush(stdout); }
    }
    printf("pp_fuzz: %d cases, %d tile runs compared, %d refused by the launchers, %d FAILURES\n", cases, ran, refused, failures);
    return failures ? 1 : 0;
}
