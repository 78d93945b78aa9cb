// graph_trace_repro.hip -- is `rocprofv3 --kernel-trace` crashing inside hipGraphLaunch a property of the profiler or of this engine?
// A stand-alone program with NO engine code: one captured graph of N kernel nodes (each spins for ~`us` microseconds), launched L times
// on one non-blocking stream with at most D launches in flight (the plain engine's pattern: one big graph per step, S + 1 steps queued,
// an event wait on the oldest), optionally rotating over E executables of the same graph.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_trace_repro tools/probes/graph_trace_repro.hip
//   rocprofv3 --kernel-trace --stats -d /tmp/gt -- /tmp/graph_trace_repro 45 330 4 1 60
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args { float *p; int n; long spin; int pad[24]; };       // ~128 bytes of kernel arguments, like a conv launch

__global__ void spin_kernel(Args a) {
    const long t0 = wall_clock64();
    float v = a.p[threadIdx.x % a.n];
    while (wall_clock64() - t0 < a.spin) v = v * 1.0001f + 1.f;
    if (v == 12345.678f) a.p[0] = v;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 45, L = argc > 2 ? atoi(argv[2]) : 330, D = argc > 3 ? atoi(argv[3]) : 4;
    const int E = argc > 4 ? atoi(argv[4]) : 1, us = argc > 5 ? atoi(argv[5]) : 40;
    float *p;
    CK(hipMalloc(&p, 4096));
    CK(hipMemset(p, 0, 4096));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
        Args a{p, 1024, (long)us * 100, {}};                   // wall_clock64: 100 MHz
        hipLaunchKernelGGL(spin_kernel, dim3(256 + (i % 7) * 64), dim3(256), 0, s, a);
    }
    CK(hipStreamEndCapture(s, &g));
    std::vector<hipGraphExec_t> ex(E);
    for (auto &e : ex) CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
    std::vector<hipEvent_t> ev(D);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int l = 0; l < L; ++l) {
        if (l >= D) CK(hipEventSynchronize(ev[l % D]));          // the launch D steps back has finished
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, Args{p, 1024, 500, {}});      // an eager launch in front of the graph (the stem)
        CK(hipGraphLaunch(ex[l % E], s));
        CK(hipEventRecord(ev[l % D], s));
        if (l % 50 == 0) { printf("launch %d\n", l); fflush(stdout); }
    }
    CK(hipStreamSynchronize(s));
    printf("done: %d nodes x %d launches, %d in flight, %d executables\n", N, L, D, E);
    return 0;
}
