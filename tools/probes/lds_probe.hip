// lds_probe.hip -- cycles per ds_read_b128 / ds_write_b64 wave instruction for the per-lane address patterns of front.hip / bneck32.hip (is a layout conflict-free
// on gfx950, or only under the lane-group model of tile_math.h?).  One workgroup of 4 waves per CU, every wave issues N back-to-back LDS operations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/bin/lds_probe tools/probes/lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(const int *offs, unsigned long long *out, int n) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[64 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 1024; i += 256) ((float *)lds)[i] = (float)i;
    __syncthreads();
    const int off = offs[lane];
    half8 acc = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) { const half8 v = *(const volatile half8 *)(lds + off + ((i * 8 + k) & 7) * 4096); acc += v; }
            else if (MODE == 1) { *(volatile half4 *)(lds + off + ((i * 8 + k) & 7) * 4096) = half4{acc[0], acc[1], acc[2], acc[3]}; }
            else { *(volatile half8 *)(lds + off + ((i * 8 + k) & 7) * 4096) = acc; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc[0] == (_Float16)12345.f) out[0] = 1;
}

int main() {
    struct Pat { const char *name; int mode; std::vector<int> off; };
    std::vector<Pat> pats;
    auto add = [&](const char *name, int mode, auto fn) { Pat p{name, mode, std::vector<int>(64)}; for (int l = 0; l < 64; ++l) p.off[l] = fn(l & 15, l >> 4, l); pats.push_back(p); };
    auto sw64 = [](int R, int c) { return R * 64 + ((c ^ (((R >> 2) & 1) << 1)) << 4); };
    auto sw128 = [](int R, int c) { return R * 128 + (((c + 2 * (R >> 1)) & 7) << 4); };
    add("b128 read  linear lane*16", 0, [&](int p, int q, int l) { return l * 16; });
    add("b128 read  64-B rows swz, rows p, chunk q (L1 / conv frags)", 0, [&](int p, int q, int l) { return sw64(p, q); });
    add("b128 read  64-B rows swz, rows p+1, chunk q", 0, [&](int p, int q, int l) { return sw64(p + 1, q); });
    add("b128 read  64-B rows swz, rows p+3, chunk q", 0, [&](int p, int q, int l) { return sw64(p + 3, q); });
    add("b128 read  64-B rows NO swizzle, rows p, chunk q", 0, [&](int p, int q, int l) { return p * 64 + q * 16; });
    add("b128 read  128-B rows rot, rows p, chunk q (Y tile)", 0, [&](int p, int q, int l) { return sw128(p, q); });
    add("b128 read  128-B rows rot, rows p, chunk 4+q", 0, [&](int p, int q, int l) { return sw128(p, 4 + q); });
    add("b128 read  stem pixel fragment (4p + 2(q&1))*8 + (q>>1)*544", 0, [&](int p, int q, int l) { return (4 * p + 2 * (q & 1)) * 8 + (q >> 1) * 544; });
    add("b128 read  stem pixel fragment, kernel row 2 (all q same row)", 0, [&](int p, int q, int l) { return (4 * p + 2 * (q & 1)) * 8; });
    add("b128 read  store side: px = l>>3, chunk l&7, 128-B rows rot", 0, [&](int p, int q, int l) { return sw128(l >> 3, l & 7); });
    add("b64 write  64-B rows swz, row p, chunk (q>>1), +8(q&1) (stem / conv1 epilogue)", 1, [&](int p, int q, int l) { return sw64(p, q >> 1) + (q & 1) * 8; });
    add("b64 write  128-B rows rot, row p, chunk (q>>1) (Y tile epilogue)", 1, [&](int p, int q, int l) { return sw128(p, q >> 1) + (q & 1) * 8; });
    add("b64 write  linear lane*8", 1, [&](int p, int q, int l) { return l * 8; });
    add("b128 write linear lane*16", 2, [&](int p, int q, int l) { return l * 16; });
    add("b128 write convert: lane stride 32 B", 2, [&](int p, int q, int l) { return l * 32; });
    int *doff; unsigned long long *dout;
    CK(hipMalloc(&doff, 256)); CK(hipMalloc(&dout, 256 * 4 * 8));
    const int n = 256;
    for (auto &p : pats) {
        CK(hipMemcpy(doff, p.off.data(), 256, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {
            if (p.mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, doff, dout, n);
            else if (p.mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, doff, dout, n);
            else hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, doff, dout, n);
            CK(hipDeviceSynchronize());
        }
        std::vector<unsigned long long> h(256 * 4);
        CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
        double s = 0; for (auto v : h) s += (double)v;
        // 4 waves per CU share one LDS: per-CU cycles per wave instruction = wave's elapsed / (n * 8) / 4 waves issuing concurrently
        printf("%-86s %6.2f clk per wave instruction per CU (4 waves concurrently: elapsed / ops / 4)\n", p.name, s / h.size() / (n * 8) / 4.0);
    }
    return 0;
}
