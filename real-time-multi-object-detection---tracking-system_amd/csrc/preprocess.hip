// preprocess.hip -- frame -> network input, one kernel.
//
// Replaces what ultralytics' predictor does before the model for the call at
// /root/reference/src/detection/detector.py:100-111: LetterBox
// (cv2.resize INTER_LINEAR + copyMakeBorder(114)), BGR->RGB, HWC uint8 -> float, /255,
// .half().  Output goes straight into the engine's input tensor: fp16 NHWC with 4 channels
// (R, G, B, 0) and the 1-pixel zero border the stem conv reads as its padding.
//
// The resize is OpenCV's 8-bit fixed-point bilinear, restated bit for bit (11-bit
// coefficients; horizontal pass x2048 in int32; vertical pass
// (((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2); the coefficient tables are built on the
// host (engine.hip: build_resize_tables) exactly as oracle/yolo_oracle.py:_resize_coeffs does.
#include "kernels.h"

namespace rtmodt {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void letterbox_kernel(FramePtrs frames, int pitch, LetterboxGeom g,
                                                        ResizeTables t, f16 *__restrict__ out, int in_h, int in_w, long total) {
    long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    int x = (int)(gid % in_w);
    long r = gid / in_w;
    int y = (int)(r % in_h);
    int b = (int)(r / in_h);
    int sy = y - g.top, sx = x - g.left;
    int c0 = 114, c1 = 114, c2 = 114;                       // B, G, R
    if (sy >= 0 && sy < g.new_h && sx >= 0 && sx < g.new_w) {
        const uint8_t *f = frames.p[b];
        if (!g.resize) {
            const uint8_t *p = f + (long)sy * pitch + sx * 3;
            c0 = p[0]; c1 = p[1]; c2 = p[2];
        } else {
            int xi = t.xofs[sx], a0 = t.xa0[sx], a1 = t.xa1[sx];
            int yi = t.yofs[sy], b0 = t.yb0[sy], b1 = t.yb1[sy];
            int xj = min(xi + 1, g.src_w - 1), yj = min(yi + 1, g.src_h - 1);
            const uint8_t *p00 = f + (long)yi * pitch + xi * 3, *p01 = f + (long)yi * pitch + xj * 3;
            const uint8_t *p10 = f + (long)yj * pitch + xi * 3, *p11 = f + (long)yj * pitch + xj * 3;
            int v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                int h0 = p00[c] * a0 + p01[c] * a1;
                int h1 = p10[c] * a0 + p11[c] * a1;
                int o = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                v[c] = min(max(o, 0), 255);
            }
            c0 = v[0]; c1 = v[1]; c2 = v[2];
        }
    }
    half4 o = {(f16)((float)c2 / 255.0f), (f16)((float)c1 / 255.0f), (f16)((float)c0 / 255.0f), (f16)0.0f};
    *(half4 *)(out + (((long)b * (in_h + 2) + y + 1) * (in_w + 2) + x + 1) * 4) = o;
}

int launch_letterbox(const FramePtrs &frames, int pitch, const LetterboxGeom &g, const ResizeTables &t,
                     const TensorView &img4, int B, hipStream_t s) {
    RT_CHECK(img4.C == 4 && img4.pad == 1, RTMODT_E_INVALID, "letterbox: image tensor must be 4-channel with border");
    RT_CHECK(B >= 1 && B <= 64, RTMODT_E_INVALID, "letterbox: batch %d", B);
    long total = (long)B * img4.H * img4.W;
    hipLaunchKernelGGL(letterbox_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames, pitch, g, t, img4.base,
                       img4.H, img4.W, total);
    RT_HIP(hipGetLastError());
    return RTMODT_OK;
}

}  // namespace rtmodt
