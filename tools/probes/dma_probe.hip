// dma_probe.hip -- how many bytes per clock and CU the global -> LDS path (global_load_lds_dwordx4) sustains from
// L2-resident data, as a function of the pieces kept in flight and the workgroups per CU.  No MFMA, no barrier:
// this is the ceiling the conv kernels' operand traffic runs into.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_probe tools/probes/dma_probe.hip && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// each wave streams `steps` x L pieces of 1 KiB (64 lanes x 16 B, rows of ROWB bytes) through a ring of DEPTH x L slots
template <int L, int DEPTH, int ROWB>
__global__ __launch_bounds__(256) void probe(const unsigned char *src, size_t region, int steps, unsigned *sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t start = (size_t)blockIdx.x * 4096;        // every address below is (start + ...) % span with span + 16 <= region
    const size_t span = region - 4096;
    unsigned char *ring = lds + wave * (DEPTH * L * 1024);
    constexpr int RPP = 1024 / ROWB;                       // rows per piece
    const int row = lane / (ROWB / 16), chunk = lane % (ROWB / 16);
    size_t off = (size_t)wave * 65536;
    auto issue = [&](int slot) {
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const unsigned char *p = src + (start + off + (size_t)(i * RPP + row) * 256 + chunk * 16) % span;   // rows 256 B apart (like a pixel stride)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                             (__attribute__((address_space(3))) void *)(ring + (slot * L + i) * 1024), 16, 0, 0);
        }
        off += (size_t)L * RPP * 256;
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d);
    int slot = DEPTH - 1;
    for (int s = 0; s < steps; ++s) {
        issue(slot);
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        wait_vm<L *(DEPTH - 1)>();
    }
    wait_vm<0>();
    if (lds[threadIdx.x * 16] == 123 && sink) sink[0] = 1;
}

template <int L, int DEPTH, int ROWB>
int run(const unsigned char *src, size_t region, int wgs_per_cu, unsigned *sink) {
    const int steps = 2000, grid = 256 * wgs_per_cu;
    const size_t smem = (size_t)4 * DEPTH * L * 1024;
    CK(hipFuncSetAttribute((const void *)probe<L, DEPTH, ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<L, DEPTH, ROWB>), dim3(grid), dim3(256), smem, 0, src, region, 50, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<L, DEPTH, ROWB>), dim3(grid), dim3(256), smem, 0, src, region, steps, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * 4 * (steps + DEPTH - 1) * L * 1024;
    printf("rows %3d B  pieces/step %d  depth %d  WGs/CU %d  LDS %3zu KB/WG : %7.2f TB/s = %5.1f B/clk/CU (2.4 GHz, 256 CUs)\n", ROWB, L, DEPTH, wgs_per_cu,
           smem / 1024, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 2.4e9 / 256);
    return 0;
}

int main() {
    const size_t region = 24u << 20;                        // 24 MiB: resident in the 8 x 4 MiB L2s only partly -> also run 8 MiB
    unsigned char *src; unsigned *sink;
    for (size_t reg : {(size_t)(8u << 20), region}) {
        CK(hipMalloc((void **)&src, reg)); CK(hipMemset(src, 1, reg)); CK(hipMalloc((void **)&sink, 4));
        printf("-- source region %zu MiB\n", reg >> 20);
        for (int w : {1, 2, 4}) {
            if (run<1, 2, 128>(src, reg, w, sink)) return 1;
            if (run<2, 2, 128>(src, reg, w, sink)) return 1;
            if (run<4, 2, 128>(src, reg, w, sink)) return 1;
            if (run<4, 4, 128>(src, reg, w, sink)) return 1;
            if (run<8, 2, 128>(src, reg, w, sink)) return 1;
            if (run<4, 4, 64>(src, reg, w, sink)) return 1;
        }
        CK(hipFree(src)); CK(hipFree(sink));
    }
    return 0;
}
