// front_probe.hip -- development harness of the fused front end (csrc/front.hip): stem + layer 1 + 2.cv1 in one launch against the two launches it
// replaces (stem_fused / stem_mfma + conv_mfma_tail), bit for bit, and timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/bin/front_probe tools/probes/front_probe.hip
// Usage: front_probe [B=32] [iters=20]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../real-time-multi-object-detection---tracking-system_amd/csrc/conv.hip"
#include "../../real-time-multi-object-detection---tracking-system_amd/csrc/front.hip"

namespace rtmodt {
#ifdef RTMODT_STAMP
__device__ unsigned long long *g_stamps;
#endif
std::string &last_error() { static std::string e; return e; }
int fail(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; }
void note_bad_option(const char *) {}
int launch_conv3x3_pp(const ConvArgs *, int, int, hipStream_t) { return RTMODT_E_UNSUPPORTED; }      // (conv_pp.hip is not part of this probe)
int launch_conv_tile_pp(const ConvArgs &, int, hipStream_t) { return RTMODT_E_UNSUPPORTED; }
}
using namespace rtmodt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define RT(x) do { int r_ = (x); if (r_ != RTMODT_OK) { printf("%s -> %d\n", #x, r_); exit(1); } } while (0)

static unsigned rng_state = 12345;
static unsigned urand() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static float frand() { return (urand() & 0xFFFF) / 65536.f - 0.5f; }

template <typename T> static T *dev(const std::vector<T> &h) { T *p; CK(hipMalloc(&p, h.size() * sizeof(T))); CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }

struct Net {
    f16 *w0, *w1, *w2; float *b0, *b1, *b2; f16 *zeros;
    int kp1 = 288, kp2 = 64;
};
static Net make_net() {
    Net n;
    std::vector<f16> w0(32 * 64, (f16)0.f);
    for (int o = 0; o < 32; ++o)
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw)
                for (int c = 0; c < 3; ++c) w0[o * 64 + kh * 16 + kw * 4 + c] = (f16)(1.2f * frand());
    std::vector<float> b0(128), b1(128), b2(128);
    for (auto *b : {&b0, &b1, &b2}) for (auto &v : *b) v = 0.3f * frand();
    std::vector<f16> w1(128 * 288, (f16)0.f), w2(128 * 64, (f16)0.f);
    for (int o = 0; o < 64; ++o) for (int k = 0; k < 288; ++k) w1[o * 288 + k] = (f16)(0.35f * frand());
    for (int o = 0; o < 64; ++o) for (int k = 0; k < 64; ++k) w2[o * 64 + k] = (f16)(0.5f * frand());
    n.w0 = dev(w0); n.w1 = dev(w1); n.w2 = dev(w2); n.b0 = dev(b0); n.b1 = dev(b1); n.b2 = dev(b2);
    std::vector<f16> z(128, (f16)0.f); n.zeros = dev(z);
    return n;
}

struct Case { const char *name; int in_h, in_w, src_h, src_w, top, left, pitch_extra, ptr_off; };

int main(int argc, char **argv) {
    const int B0 = argc > 1 && std::string(argv[1]) != "fuzz" ? atoi(argv[1]) : 32, iters = argc > 2 && std::string(argv[1]) != "fuzz" ? atoi(argv[2]) : (argc > 1 && std::string(argv[1]) == "fuzz" ? 1 : 20);
    const int B = B0;
    Net net = make_net();
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef RTMODT_STAMP
    const int STAMP_ROWS = 1 << 15;                          // every kernel of this build stamps: rows for the largest grid (the stem: B x 320 workgroups)
    unsigned long long *ds; CK(hipMalloc(&ds, (size_t)STAMP_ROWS * 16 * 8)); CK(hipMemset(ds, 0, (size_t)STAMP_ROWS * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ds, sizeof(ds)));
#endif
    const Case cases[] = {
        {"640x640 aligned", 640, 640, 640, 640, 0, 0, 0, 0},
        {"640x640 misaligned base +1, pitch +5", 640, 640, 640, 640, 0, 0, 5, 1},
        {"600x632 frame in a 640x640 canvas (top 20, left 4)", 640, 640, 600, 632, 20, 4, 0, 0},
        {"609x637 frame, top 15 left 1, misaligned +7 pitch +3", 640, 640, 609, 637, 15, 1, 3, 7},
        {"384x640 rectangle (1080p's minimal rectangle)", 384, 640, 384, 640, 0, 0, 0, 0},
        {"320x320", 320, 320, 320, 320, 0, 0, 0, 3},
    };
    int bad = 0;
    // `front_probe fuzz [n]`: n random geometries instead of the six named ones -- canvas sizes (multiples of 32 x 64 up to 704), frames smaller than the canvas at
    // random letterbox offsets, odd pitches and misaligned bases, 1-3 images: every case bit-identical to the two launches, byte and tensor source
    std::vector<Case> all(cases, cases + sizeof(cases) / sizeof(cases[0]));
    static char names[512][96];
    const bool fuzz = argc > 1 && std::string(argv[1]) == "fuzz";
    if (fuzz) {
        all.clear();
        const int n = argc > 2 ? atoi(argv[2]) : 60;
        for (int i = 0; i < n && i < 512; ++i) {
            const int in_h = 32 * (1 + urand() % 22), in_w = 64 * (1 + urand() % 11);
            const int src_h = std::max(1, in_h - (int)(urand() % 3 == 0 ? urand() % std::min(in_h, 70) : 0)), src_w = std::max(1, in_w - (int)(urand() % 3 == 0 ? urand() % std::min(in_w, 70) : 0));
            const int top = (in_h - src_h) > 0 ? urand() % (in_h - src_h + 1) : 0, left = (in_w - src_w) > 0 ? urand() % (in_w - src_w + 1) : 0;
            snprintf(names[i], sizeof(names[i]), "fuzz %3d: %dx%d frame in %dx%d at (%d, %d)", i, src_h, src_w, in_h, in_w, top, left);
            all.push_back(Case{names[i], in_h, in_w, src_h, src_w, top, left, (int)(urand() % 9), (int)(urand() % 16)});
        }
    }
    const int Bf = fuzz ? 1 + (int)(urand() % 3) : B;
    for (const Case &c : all) {
        const int B = Bf;
        const int H = c.in_h, W = c.in_w, H0 = H / 2, W0 = W / 2, H1 = H / 4, W1 = W / 4;
        const int pitch = 3 * c.src_w + c.pitch_extra;
        const size_t fbytes = (size_t)c.src_h * pitch + 64;
        std::vector<uint8_t> hf(fbytes * B + 64);
        for (auto &v : hf) v = (uint8_t)(urand() & 255);
        uint8_t *dfr = dev(hf);
        FramePtrs fp{};
        for (int b = 0; b < B; ++b) fp.p[b] = dfr + c.ptr_off + fbytes * b;
        for (int b = B; b < 64; ++b) fp.p[b] = fp.p[0];
        LetterboxGeom g{c.src_h, c.src_w, c.src_w, c.src_h, c.top, c.left, 0};
        // tensors: stem output t0 [B][H0+2][W0+2][32], concat tensor of layer 2 [B][H1+2][W1+2][96] (2.cv1 writes channels 0..63), twice (reference / fused)
        const size_t n0 = (size_t)B * (H0 + 2) * (W0 + 2) * 32, n2 = (size_t)B * (H1 + 2) * (W1 + 2) * 96, n1 = (size_t)B * (H1 + 2) * (W1 + 2) * 64;
        f16 *t0, *t1, *ref, *out; CK(hipMalloc(&t0, n0 * 2)); CK(hipMalloc(&t1, n1 * 2)); CK(hipMalloc(&ref, n2 * 2)); CK(hipMalloc(&out, n2 * 2));
        CK(hipMemset(t0, 0, n0 * 2)); CK(hipMemset(t1, 0, n1 * 2)); CK(hipMemset(ref, 0, n2 * 2)); CK(hipMemset(out, 0, n2 * 2));
        TensorView v0; v0.base = t0; v0.H = H0; v0.W = W0; v0.C = 32; v0.pad = 1; v0.coff = 0; v0.c = 32;
        TensorView v1; v1.base = t1; v1.H = H1; v1.W = W1; v1.C = 64; v1.pad = 1; v1.coff = 0; v1.c = 64;
        TensorView vr; vr.base = ref; vr.H = H1; vr.W = W1; vr.C = 96; vr.pad = 1; vr.coff = 0; vr.c = 64;
        TensorView vo = vr; vo.base = out;
        ConvLaunch cl;
        cl.in = v0; cl.out = v1; cl.wt = net.w1; cl.bias = net.b1; cl.B = B; cl.cin = 32; cl.cout = 64; cl.ks = 3; cl.stride = 2; cl.act = 1; cl.kp = 288;
        cl.tile = TILE_TAIL_128x64; cl.tail_out = vr; cl.tail_wt = net.w2; cl.tail_bias = net.b2; cl.tail_cout = 64; cl.tail_kp = 64; cl.tail_act = 1;
        FrontLaunch fl;
        fl.frames = fp; fl.frame0 = 0; fl.pitch = pitch; fl.g = g; fl.zeros = net.zeros;
        fl.w0 = net.w0; fl.w1 = net.w1; fl.w2 = net.w2; fl.b0 = net.b0; fl.b1 = net.b1; fl.b2 = net.b2; fl.kp1 = 288; fl.kp2 = 64;
        fl.out = vo; fl.B = B; fl.c0 = 32; fl.c1 = 64; fl.c2 = 64; fl.in_h = H; fl.in_w = W;
        auto run_ref = [&]() { RT(launch_stem_fused(fp, 0, pitch, g, H, W, nullptr, v0, net.w0, net.b0, B, 32, st)); RT(launch_conv(cl, st)); };
        auto run_new = [&]() { RT(launch_front(fl, st)); };
        run_ref(); run_new();
        CK(hipStreamSynchronize(st));
        std::vector<f16> hr(n2), ho(n2);
        CK(hipMemcpy(hr.data(), ref, n2 * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho.data(), out, n2 * 2, hipMemcpyDeviceToHost));
        size_t diff = 0, nz = 0; double maxd = 0; size_t first = (size_t)-1;
        for (size_t i = 0; i < n2; ++i) {
            if (memcmp(&hr[i], &ho[i], 2) != 0) { if (!diff) first = i; ++diff; maxd = std::max(maxd, (double)std::fabs((float)hr[i] - (float)ho[i])); }
            nz += (float)hr[i] != 0.f;
        }
        auto tm = [&](auto &&fn) { fn(); CK(hipEventRecord(e0, st)); for (int i = 0; i < iters; ++i) fn(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f / iters; };
        const float us_ref = tm(run_ref), us_new = tm(run_new);
#ifdef RTMODT_STAMP
        {                                                      // phase stamps of every workgroup's second tile: mean clocks between consecutive stamps
            const int G = 2 * 256;
            CK(hipStreamSynchronize(st));
            CK(hipMemset(ds, 0, (size_t)G * 16 * 8));
            run_new();
            CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> hs((size_t)G * 16);
            CK(hipMemcpy(hs.data(), ds, hs.size() * 8, hipMemcpyDeviceToHost));
            double sum[11] = {0}; int cnt = 0; double ghz = 0;
            for (int g = 0; g < G; ++g) {
                const unsigned long long *r = &hs[(size_t)g * 16];
                if (!r[10] || !r[0]) continue;
                for (int k = 1; k <= 10; ++k) sum[k] += (double)(r[k] ? r[k] : r[k - 1]) - (double)(r[k - 1] ? r[k - 1] : r[k]);
                if (r[15] > r[13]) ghz += (double)(r[14] - r[12]) / (double)(r[15] - r[13]) * 0.1;
                ++cnt;
            }
            printf("    stamps (%d workgroups, ~%.2f GHz): wait+barrier %.0f | convert %.0f | barrier+issue %.0f | stem %.0f | barrier %.0f | L1 mfma %.0f | L1 epilogue %.0f | barrier %.0f | tail %.0f | barrier+stores %.0f  clk; tile total %.0f\n",
                   cnt, cnt ? ghz / cnt : 0, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt, sum[7] / cnt, sum[8] / cnt, sum[9] / cnt, sum[10] / cnt,
                   (sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[7] + sum[8] + sum[9] + sum[10]) / cnt);
        }
#endif
        printf("%-56s B %2d  bytes source:  %zu / %zu halves differ (max |d| %.4g, first at %zu; %zu non-zero)   two launches %7.1f us   fused %7.1f us\n", c.name, B, diff, n2, maxd, first, nz, us_ref, us_new);
        bad += diff != 0 || nz == 0;
        if (diff) {
            const size_t per = (size_t)(H1 + 2) * (W1 + 2) * 96; const size_t i = first;
            printf("    first difference: image %zu, padded row %zu, col %zu, channel %zu: %g vs %g\n", i / per, (i % per) / ((W1 + 2) * 96), (i % ((W1 + 2) * 96)) / 96, i % 96, (float)hr[i], (float)ho[i]);
        }
        // ---- tensor source: the RGB0 fp16 image a letterbox launch would have written (same values as the byte path: half(c / 255.f)) ----
        {
            std::vector<f16> himg((size_t)B * (H + 2) * (W + 2) * 4, (f16)0.f);
            for (int b = 0; b < B; ++b)
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x) {
                        f16 *px = &himg[(((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * 4];
                        const int sy = y - c.top, sx = x - c.left;
                        if (sy >= 0 && sy < c.src_h && sx >= 0 && sx < c.src_w) {
                            const uint8_t *s = &hf[c.ptr_off + fbytes * b + (size_t)sy * pitch + 3 * sx];
                            px[0] = (f16)(s[2] / 255.f); px[1] = (f16)(s[1] / 255.f); px[2] = (f16)(s[0] / 255.f);
                        } else px[0] = px[1] = px[2] = (f16)(114 / 255.f);
                    }
            f16 *dimg = dev(himg);
            CK(hipMemset(out, 0, n2 * 2));
            FrontLaunch ft = fl;
            ft.from_tensor = true; ft.img4.base = dimg; ft.img4.H = H; ft.img4.W = W; ft.img4.C = 4; ft.img4.pad = 1; ft.img4.coff = 0; ft.img4.c = 4;
            RT(launch_front(ft, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(ho.data(), out, n2 * 2, hipMemcpyDeviceToHost));
            size_t d2 = 0;
            for (size_t i = 0; i < n2; ++i) d2 += memcmp(&hr[i], &ho[i], 2) != 0;
            const float us_t = tm([&]() { RT(launch_front(ft, st)); });
            printf("%-56s       tensor source: %zu halves differ                                                                        fused %7.1f us\n", "", d2, us_t);
            bad += d2 != 0;
            CK(hipFree(dimg));
        }
        CK(hipFree(dfr)); CK(hipFree(t0)); CK(hipFree(t1)); CK(hipFree(ref)); CK(hipFree(out));
    }
    printf(bad ? "FAILED\n" : "all bit-identical\n");
    return bad ? 1 : 0;
}
