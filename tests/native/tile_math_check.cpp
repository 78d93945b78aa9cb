// Host-side check of the integer arithmetic the conv kernels use (csrc/tile_math.h), compiled with g++ by
// tests/test_tile_math_cpu.py.  Exit code 0 and a line "ok ..." on success; the first counter-example otherwise.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "tile_math.h"
using namespace rtmodt;

static int check_fastdiv() {
    long checked = 0;
    std::mt19937_64 rng(1234);
    auto one = [&](int n, int d) -> bool {
        const FastDiv f = make_fastdiv(d);
        if (fdiv(n, f) != n / d) { printf("fastdiv: %d / %d = %d, got %d (mul %u shift %u)\n", n, d, n / d, fdiv(n, f), f.mul, f.shift); return false; }
        ++checked;
        return true;
    };
    for (int d = 1; d <= (1 << 20); d += (d < 70000 ? 1 : 257)) {           // every divisor a feature map can produce, then a stride
        const int top = 2147483647;
        const int ns[] = {0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, top, top - 1, top / d * d, top / d * d - 1};
        for (int n : ns) if (n >= 0 && !one(n, d)) return 1;
        for (int k = 0; k < 6; ++k) {
            const long long m = (long long)(rng() % (unsigned long long)(top / d + 1)) * d;      // a multiple of d and its neighbours
            for (long long n : {m - 1, m, m + 1}) if (n >= 0 && n <= top && !one((int)n, d)) return 1;
            if (!one((int)(rng() & 0x7FFFFFFF), d)) return 1;
        }
    }
    for (int d : {102400, 409600, 6400, 1600, 400, 25600, 215168, 56448, 15488, 82, 42, 22, 162, 322, 2147483647, 1073741824, 1073741825})
        for (int k = 0; k < 200000; ++k) if (!one((int)(rng() & 0x7FFFFFFF), d)) return 1;
    printf("ok fastdiv %ld cases\n", checked);
    return 0;
}

static int check_xcd_tile() {
    for (int n = 1; n <= 5000; ++n) {
        std::vector<int> seen(n, 0);
        for (int lin = 0; lin < n; ++lin) {
            const int id = xcd_tile_id(n, lin);
            if (id < 0 || id >= n || seen[id]++) { printf("xcd_tile_id: n %d lin %d -> %d\n", n, lin, id); return 1; }
        }
        // one XCD's tiles are one contiguous run, visited in order
        for (int x = 0; x < 8 && x < n; ++x) {
            int prev = -1;
            for (int lin = x; lin < n; lin += 8) { const int id = xcd_tile_id(n, lin); if (prev >= 0 && id != prev + 1) { printf("xcd run broken: n %d xcd %d\n", n, x); return 1; } prev = id; }
        }
    }
    printf("ok xcd_tile_id\n");
    return 0;
}

// the fused Bottleneck's LDS swizzle: a bijection per row, src the inverse of slot, and every ds_read_b128 lane group of an MFMA
// fragment read (16 consecutive rows from ANY start row, k-chunk q per lane) lands in 16 distinct 16-byte bank slots
template <int CB>
static int check_swizzle_cb() {
    const int NC = CB / 16;
    for (int R = 0; R < 4096; ++R) {
        int seen = 0;
        for (int c = 0; c < NC; ++c) {
            const int s = swz_slot<CB>(R, c);
            if (s < 0 || s >= NC || (seen >> s & 1) || swz_src<CB>(R, s) != c) { printf("swizzle<%d>: row %d chunk %d -> slot %d\n", CB, R, c, s); return 1; }
            seen |= 1 << s;
        }
    }
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    for (int R0 = 0; R0 < 2048; ++R0)
        for (int kk = 0; kk < NC / 4; ++kk)
            for (auto &g : groups) {
                int seen = 0;
                for (int lane : g) {
                    const int r = lane & 15, q = lane >> 4;
                    const int slot = (swz_plane_off<CB>(R0 + r, kk * 4 + q) >> 4) & 15;      // 16-byte slot of the 256-byte bank row
                    if (seen >> slot & 1) { printf("swizzle<%d>: bank conflict from row %d, k-substep %d, lane %d\n", CB, R0, kk, lane); return 1; }
                    seen |= 1 << slot;
                }
            }
    return 0;
}
static int check_swizzle() {
    if (check_swizzle_cb<64>() || check_swizzle_cb<128>()) return 1;
    printf("ok swizzle\n");
    return 0;
}

static int check_pt_run() {
    long cfgs = 0;
    for (int n_mt = 1; n_mt <= 700; ++n_mt)
        for (int slices = 1; slices <= 5; ++slices)
            for (int per_cu = 1; per_cu <= 3; ++per_cu) {
                int groups = per_cu * 256 / slices;
                if (groups > n_mt) groups = n_mt;
                if (groups < 1) groups = 1;
                for (int slice = 0; slice < slices; ++slice) {
                    std::vector<int> seen(n_mt, 0);
                    for (int g = 0; g < groups; ++g) {
                        const PtRun r = pt_run(g, slice, groups, n_mt);
                        if (r.Gx < 1 || r.j < 0 || r.j >= r.Gx || r.run0 < 0 || r.run0 + r.run_n > n_mt) { printf("pt_run: bad run n_mt %d groups %d slice %d g %d\n", n_mt, groups, slice, g); return 1; }
                        const int n_my = r.j < r.run_n ? (r.run_n - r.j + r.Gx - 1) / r.Gx : 0;      // as the kernel computes it
                        for (int t = 0; t < n_my; ++t) {
                            const int tile = r.run0 + r.j + t * r.Gx;
                            if (tile >= r.run0 + r.run_n || seen[tile]++) { printf("pt_run: tile %d twice or outside (n_mt %d groups %d slice %d g %d)\n", tile, n_mt, groups, slice, g); return 1; }
                        }
                    }
                    for (int t = 0; t < n_mt; ++t) if (!seen[t]) { printf("pt_run: tile %d has no owner (n_mt %d groups %d slices %d slice %d)\n", t, n_mt, groups, slices, slice); return 1; }
                    ++cfgs;
                }
            }
    printf("ok pt_run %ld configurations, every pixel tile of every slice owned exactly once\n", cfgs);
    return 0;
}

// nms_kernel's integer helpers: the k-th set bit of a word, and the bitonic schedule (up to three strides per step) as a sorting network
static int check_nms_math() {
    std::mt19937_64 rng(99);
    for (int it = 0; it < 200000; ++it) {
        unsigned long long w = rng();
        if (it % 3 == 0) w &= rng();
        if (it % 5 == 0) w |= rng();
        if (it < 64) w = 1ull << it;
        if (it == 64) w = ~0ull;
        int k = 0;
        for (int b = 0; b < 64; ++b)
            if ((w >> b) & 1ull) { if (kth_set_bit(w, k) != b) { printf("kth_set_bit(%llx, %d) = %d, want %d\n", w, k, kth_set_bit(w, k), b); return 1; } ++k; }
    }
    for (int P = 2; P <= 16384; P <<= 1)
        for (int rep = 0; rep < (P <= 1024 ? 20 : 3); ++rep) {
            std::vector<unsigned long long> keys(P), ref;
            for (auto &v : keys) v = rep == 0 ? rng() % 7 : rng();          // many equal keys in the first repetition
            if (rep == 1) for (int i = 0; i < P; ++i) keys[i] = i;            // ascending input
            ref = keys;
            bitonic_run(keys.data(), P);
            std::sort(ref.begin(), ref.end(), [](unsigned long long a, unsigned long long b) { return a > b; });
            if (keys != ref) { printf("bitonic_run: P %d rep %d not sorted descending\n", P, rep); return 1; }
        }
    printf("ok nms math\n");
    return 0;
}

// conv3x3_pp's DMA stream replayed per wave: after the counted wait at the end of R(kw) every piece the NEXT phase reads has retired (vmcnt retires
// in issue order: "at most N outstanding" == "all but the N youngest done"), and the wait is exact (the youngest piece it covers IS one the next
// phase reads -- a smaller count would stall on pieces nobody needs yet).  Also the fragment reads of the three taps (rows r + kw of the rotated
// 128-byte-row image) are bank-conflict free under ds_read_b128's lane groups.
static int check_pp_schedule() {
    long cases = 0;
    for (int la = 2; la <= 4; la += 2)
    for (int lb = 1; lb <= 4; ++lb)
        for (int w0 = 0; w0 < 2; ++w0) {
            std::vector<int> need;                       // need[i] = global phase index (3 s + kw) that first reads piece i; issue order
            const int S = 6;
            // fill: strip 0 (first part), tap 0, strip 0 (rest), tap 1 -- then the fill's wait (all but tap 1)
            for (int i = 0; i < la + w0; ++i) need.push_back(0);
            for (int i = 0; i < lb; ++i) need.push_back(0);
            for (int i = 0; i < la; ++i) need.push_back(0);
            for (int i = 0; i < lb; ++i) need.push_back(1);
            auto check_wait = [&](int n_out, int next_phase, const char *where) -> int {
                const int done = (int)need.size() - n_out;                      // pieces [0, done) have retired
                for (int i = done; i < (int)need.size(); ++i)
                    if (need[i] <= next_phase) { printf("pp schedule: lb %d wave0 %d %s: piece %d (first read in phase %d) may still be in flight before phase %d\n", lb, w0, where, i, need[i], next_phase); return 1; }
                if (done > 0 && need[done - 1] > next_phase) { printf("pp schedule: lb %d wave0 %d %s: waits for piece %d, first read only in phase %d (next phase %d)\n", lb, w0, where, done - 1, need[done - 1], next_phase); return 1; }
                return 0;
            };
            if (check_wait(lb, 0, "fill")) return 1;
            for (int s = 0; s < S; ++s)
                for (int kw = 0; kw < 3; ++kw) {
                    const PpIssue is = pp_issue(kw, lb, w0 != 0, la, la);
                    const int nb_need = kw == 0 ? 3 * s + 2 : (kw == 1 ? 3 * (s + 1) : 3 * (s + 1) + 1);
                    for (int i = 0; i < is.nB; ++i) need.push_back(nb_need);
                    for (int i = 0; i < is.nA; ++i) need.push_back(3 * (s + 1));
                    if (check_wait(pp_wait_count(kw, lb, w0 != 0, la, la), 3 * s + kw + 1, kw == 0 ? "kw 0" : (kw == 1 ? "kw 1" : "kw 2"))) return 1;
                    ++cases;
                }
        }
    // bank conflicts of the fragment reads: ds_read_b128 serves 4 groups of 16 lanes, bank row = 256 B = 16 slots of 16 B
    const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    for (int kw = 0; kw < 3; ++kw)
        for (int kk = 0; kk < 2; ++kk)
            for (int hi = 0; hi < 2; ++hi)
                for (int gi = 0; gi < 2; ++gi) {
                    int used[16] = {};
                    for (int li = 0; li < 16; ++li) {
                        const int lane = grp[gi][li] + 32 * hi, r = lane & 15, q = lane >> 4, R = r + kw;
                        const int addr = R * 128 + (swz_slot<128>(R, kk * 4 + q) << 4);
                        if (used[(addr >> 4) & 15]++) { printf("pp swizzle: tap %d k-half %d: bank conflict in lane group %d\n", kw, kk, gi + 2 * hi); return 1; }
                    }
                }
    printf("ok pp schedule %ld phases replayed, fragment reads conflict-free\n", cases);
    return 0;
}

// pp_lpt_schedule: every launch-linear tile id of every problem appears in exactly one workgroup's list, ids keep their residue mod 8 (one XCD's
// share), lists are dense prefixes, and no workgroup carries more than the lightest one plus one tile of the costliest problem it could still take
static int check_pp_lpt() {
    struct Case { int n; int tiles[6]; long cost[6]; int G; };
    const Case cases[] = {
        {3, {53, 205, 810}, {126000, 63000, 31500}, 256},       // Detect stage 0 at 32 frames, 192-wide tile
        {3, {159, 615, 2430}, {75500, 39500, 21500}, 256},      // ... 64-wide tile
        {6, {53, 53, 205, 205, 810, 810}, {24000, 24000, 24000, 24000, 24000, 24000}, 256},
        {2, {3, 40}, {1000, 10}, 8},
        {1, {5}, {7}, 5},
        {2, {1000, 24}, {10, 900}, 64},
    };
    for (const Case &c : cases) {
        int start[7] = {0};
        for (int z = 0; z < c.n; ++z) start[z + 1] = (start[z] + c.tiles[z] + 7) / 8 * 8;
        const int total = start[c.n], max_T = total / (c.G >= 8 ? 8 : 1) + 1;      // (what the launcher passes: a whole XCD share on one workgroup at worst)
        std::vector<int> table((size_t)c.G * max_T);
        const int T = pp_lpt_schedule(c.n, start, c.tiles, c.cost, c.G, table.data(), max_T);
        if (T <= 0) { printf("pp_lpt_schedule: no schedule (G %d)\n", c.G); return 1; }
        std::vector<int> seen(total, 0);
        std::vector<long> load(c.G, 0);
        long maxcost = 0;
        for (int z = 0; z < c.n; ++z) maxcost = std::max(maxcost, c.cost[z]);
        for (int g = 0; g < c.G; ++g) {
            bool ended = false;
            for (int j = 0; j < max_T; ++j) {
                const int id = table[(size_t)g * max_T + j];
                if (id < 0) { ended = true; continue; }
                if (ended || j >= T) { printf("pp_lpt_schedule: hole in the list of workgroup %d\n", g); return 1; }
                if (c.G >= 8 && (id & 7) != (g & 7)) { printf("pp_lpt_schedule: id %d left its XCD share (workgroup %d)\n", id, g); return 1; }
                int z = 0;
                while (z + 1 < c.n && id >= start[z + 1]) ++z;
                if (id - start[z] >= c.tiles[z] || seen[id]++) { printf("pp_lpt_schedule: id %d is a filler or appears twice\n", id); return 1; }
                load[g] += c.cost[z];
            }
        }
        for (int z = 0; z < c.n; ++z)
            for (int k = 0; k < c.tiles[z]; ++k) if (!seen[start[z] + k]) { printf("pp_lpt_schedule: id %d has no owner\n", start[z] + k); return 1; }
        for (int x = 0; x < (c.G >= 8 ? 8 : 1); ++x) {
            long lo = -1, hi = 0;
            for (int g = x; g < c.G; g += (c.G >= 8 ? 8 : 1)) { lo = lo < 0 ? load[g] : std::min(lo, load[g]); hi = std::max(hi, load[g]); }
            if (hi - lo > maxcost) { printf("pp_lpt_schedule: XCD share %d unbalanced: %ld .. %ld (costliest tile %ld)\n", x, lo, hi, maxcost); return 1; }
        }
    }
    printf("ok pp lpt schedule\n");
    return 0;
}

// pp_index_fits: whenever it accepts a tensor, PpOut's 24-bit multiplies (conv_pp.hip) give the exact element offset of every corner position; and the
// case ADVICE r04 names -- YOLOv8l's layer-2 concat tensor (5 x 64 channels) at 1024 / 1280 pixels -- is refused (so is every tensor above 2^24 elements per image).
static unsigned umul24(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)) & 0xFFFFFFFFull); }
static int check_pp_index() {
    const double widths[5] = {0.25, 0.5, 0.75, 1.0, 1.25};
    const int reps[5] = {1, 1, 2, 3, 3};                      // Bottlenecks of the layer-2 C2f
    long accepted = 0, refused = 0;
    for (int sc = 0; sc < 5; ++sc)
        for (int size = 320; size <= 1280; size += 32)
            for (int level = 0; level < 4; ++level) {          // the C2f concat tensors at strides 4, 8, 16, 32
                const int H = size >> (2 + level), c = (int)(std::min(64 << (level + 1), sc >= 3 ? 512 : 1024) * widths[sc] / 8 + 0.999) * 8 / 2;
                const int n = level == 0 || level == 3 ? reps[sc] : 2 * reps[sc], C = (2 + n) * c;
                const int Hp = H + 2, Wp = H + 2, B = 32;
                const long hwc = (long)Hp * Wp * C;
                const bool fits = pp_index_fits(hwc, hwc, (long)H * Wp, (long)B * H * Wp);
                if (hwc >= (1L << 24) && fits) { printf("pp_index_fits accepts %d x %d x %d\n", Hp, Wp, C); return 1; }
                if (!fits) { ++refused; continue; }
                ++accepted;
                const int o2 = Wp * C, o1 = Hp * o2, o0 = o2 + C, wq = Wp - 1, HW = H * wq;
                const FastDiv d_img = make_fastdiv(HW), d_row = make_fastdiv(wq);
                const int ms[6] = {0, wq - 1, HW - 1, HW, (B - 1) * HW + (H - 1) * wq + H - 1, B * HW - 1};
                for (int m : ms) {
                    const int b = fdiv(m, d_img), rem = m - (int)umul24((unsigned)b, (unsigned)HW);
                    const int oy = fdiv(rem, d_row), ox = rem - (int)umul24((unsigned)oy, (unsigned)wq);
                    const int opix = o0 + (int)umul24((unsigned)b, (unsigned)o1) + (int)umul24((unsigned)oy, (unsigned)o2) + (int)umul24((unsigned)ox, (unsigned)C);
                    const long eb = m / HW, er = m % HW, ey = er / wq, ex = er % wq;
                    const long exact = (long)o0 + eb * o1 + ey * o2 + ex * C;
                    if (exact < (1L << 31) && (long)opix != exact) { printf("PpOut index: m %d of %d x %d x %d: %d != %ld\n", m, Hp, Wp, C, opix, exact); return 1; }
                }
            }
    const long l1024 = 258L * 258 * 320, l1280 = 322L * 322 * 320, s640 = 162L * 162 * 96;
    if (pp_index_fits(l1024, 0, 256L * 258, 256L * 258) || pp_index_fits(l1280, 0, 320L * 322, 320L * 322) || !pp_index_fits(s640, s640, 160L * 162, 32L * 160 * 162)) {
        printf("pp_index_fits: the YOLOv8l layer-2 tensors must be refused, YOLOv8s @ 640 accepted\n");
        return 1;
    }
    printf("ok pp index bound (%ld tensors accepted and exact, %ld refused)\n", accepted, refused);
    return 0;
}

int main() { return check_pp_index() || check_fastdiv() || check_xcd_tile() || check_pt_run() || check_swizzle() || check_nms_math() || check_pp_schedule() || check_pp_lpt(); }
