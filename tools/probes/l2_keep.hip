// l2_keep.hip -- does an XCD's L2 keep a producer kernel's output for the NEXT kernel on the same stream?
// Kernel W: workgroup b writes a pointer-chase ring into chunk (b >> 3) of region (b & 7) (one 2 MiB region per XCD slot under
// round-robin placement).  Kernel R: workgroup b chases the ring of region ((b + shift) & 7): shift 0 = the lines this XCD's L2
// just wrote, shift 1 = lines another XCD wrote.  Lane 0 times the dependent loads with s_memtime and reports its XCC_ID.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/l2_keep tools/probes/l2_keep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int LINES = 512, CHUNK = LINES * 128, REGION = 32 * CHUNK;   // 64 KiB per workgroup, 2 MiB per region

__global__ void wr(unsigned *buf) {
    const int b = blockIdx.x, region = b & 7, chunk = b >> 3;
    unsigned *p = buf + ((size_t)region * REGION + (size_t)chunk * CHUNK) / 4;
    for (int i = threadIdx.x; i < LINES; i += blockDim.x) p[i * 32] = (unsigned)((i * 5 + 1) & (LINES - 1));
}
__global__ void rd(const unsigned *buf, int shift, unsigned long long *out) {
    const int b = blockIdx.x, region = (b + shift) & 7, chunk = b >> 3;
    const unsigned *p = buf + ((size_t)region * REGION + (size_t)chunk * CHUNK) / 4;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long t0, t1;
        unsigned idx = 0;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int i = 0; i < LINES; ++i) idx = __builtin_nontemporal_load(p + idx * 32) & (LINES - 1);   // dependent chain
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(idx) : "memory");
        out[b * 2] = (t1 - t0) + (idx == 12345 ? 1 : 0);
        out[b * 2 + 1] = xcc & 15;
    }
}
int main() {
    unsigned *buf; unsigned long long *out;
    CK(hipMalloc(&buf, (size_t)8 * REGION)); CK(hipMalloc(&out, 256 * 16));
    std::vector<unsigned long long> h(512);
    for (int mode = 0; mode < 4; ++mode) {
        const int shift = mode & 1, sync_between = mode >> 1;
        std::vector<double> lat;
        int same = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(wr, dim3(256), dim3(256), 0, nullptr, buf);
            if (sync_between) CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(rd, dim3(256), dim3(64), 0, nullptr, buf, shift, out);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost));
            for (int b = 0; b < 256; ++b) { lat.push_back((double)h[b * 2] / LINES); same += (int)h[b * 2 + 1] == (b & 7); }
        }
        std::sort(lat.begin(), lat.end());
        printf("reader chases region of workgroup id + %d (%s), %s: median %.0f, p10 %.0f, p90 %.0f s_memtime ticks per dependent load; XCC_ID == id & 7 in %d of %d workgroups\n",
               shift, shift ? "another XCD's lines" : "this XCD's own lines", sync_between ? "host sync between the kernels" : "back to back on one stream",
               lat[lat.size() / 2], lat[lat.size() / 10], lat[lat.size() * 9 / 10], same, 5 * 256);
    }
    {   // reference points: the same chase twice in ONE kernel pair without a writer in between (second pass: L2-warm if L2 survives, else cold)
        hipLaunchKernelGGL(rd, dim3(256), dim3(64), 0, nullptr, buf, 0, out);
        hipLaunchKernelGGL(rd, dim3(256), dim3(64), 0, nullptr, buf, 0, out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost));
        std::vector<double> lat;
        for (int b = 0; b < 256; ++b) lat.push_back((double)h[b * 2] / LINES);
        std::sort(lat.begin(), lat.end());
        printf("read after read (no writer in between): median %.0f ticks per load\n", lat[128]);
    }
    return 0;
}
