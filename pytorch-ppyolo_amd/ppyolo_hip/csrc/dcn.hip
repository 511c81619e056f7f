// DCNv2 bilinear gather for gfx950: replaces the ~60 index/elementwise ATen ops of the
// reference's DCNv2.forward (model/custom_layers.py:565-662).
//
// One WAVE per (output pixel, filter tap): the 64 lanes share one sampling position (the
// position, the 4 corner addresses and the 4 bilinear weights are computed once per wave in
// uniform registers), and sweep the C channels of the 4 corner pixels with 16-byte loads
// (NHWC: a corner pixel's C channels are one contiguous, fully coalesced run).  The blended,
// mask-modulated samples are written as the (tap, c)-ordered row of the "columns" matrix
// [N*Ho*Wo][9*C].  The model's forward pass no longer goes through this matrix -- dcn_fused.hip builds the same
// samples inside the contraction kernel -- this kernel remains as the stand-alone gather (parity of the sampling
// arithmetic against the reference's own offsets, tests/test_gpu_ops.py; columns for a weight gradient).
//
// Arithmetic follows the reference bit-for-bit (no fp contraction in this file):
//   * coordinates live in a zero-padded frame of size (H+2p+1) x (W+2p+1)   (:571-574)
//   * pos = (window origin + tap offset) + learned offset, clamped to [0, H+2p-1] (:612-615)
//   * the image index is folded into the fp32 row coordinate (y + n*(H+2p+1)) BEFORE floor, so
//     for n > 0 the fractional part carries that rounding                  (:626-633, :650-651)
//   * value = w1*v1 + w2*v2 + w3*v3 + w4*v4 (left to right), then * sigmoid(mask) (:654-660)
#include <math.h>

#include "common.h"
#pragma clang fp contract(off)

namespace {

__global__ void __launch_bounds__(256) dcn_sample_kernel(const float *__restrict__ x, int x_ld,
                                                         const float *__restrict__ om, int om_ld,
                                                         float *__restrict__ cols, int N, int H, int W,
                                                         int C, int Ho, int Wo, int stride, int pad) {
    const int lane = threadIdx.x & 63;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long total = (long long)N * Ho * Wo * 9;
    if (wave_id >= total) return;
    const int tap = (int)(wave_id % 9);
    const long long m = wave_id / 9;
    const int wo = (int)(m % Wo);
    const int ho = (int)((m / Wo) % Ho);
    const int n = (int)(m / ((long long)Wo * Ho));
    const int kh = tap / 3, kw = tap - kh * 3;

    const float *o = om + m * om_ld;
    const float off_y = o[2 * tap], off_x = o[2 * tap + 1];
    const float ml = o[18 + tap];
    const float mask = 1.0f / (1.0f + expf(-ml));

    const int Hp = H + 2 * pad + 1;
    float py = ((float)(ho * stride + pad) + (float)(kh - 1)) + off_y;
    float px = ((float)(wo * stride + pad) + (float)(kw - 1)) + off_x;
    py = fminf(fmaxf(py, 0.0f), (float)(H + 2 * pad) - 1.0f);
    px = fminf(fmaxf(px, 0.0f), (float)(W + 2 * pad) - 1.0f);
    const float row0 = (float)n * (float)Hp;
    py = py + row0;
    const float y1f = floorf(py), x1f = floorf(px);
    const float lh = py - y1f, lw = px - x1f;
    const float hh = 1.0f - lh, hw = 1.0f - lw;
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    // back to un-padded image coordinates of image n
    const int y1 = (int)y1f - n * Hp - pad, x1 = (int)x1f - pad;
    const int y2 = y1 + 1, x2 = x1 + 1;
    const bool y1ok = (unsigned)y1 < (unsigned)H, y2ok = (unsigned)y2 < (unsigned)H;
    const bool x1ok = (unsigned)x1 < (unsigned)W, x2ok = (unsigned)x2 < (unsigned)W;
    const float *img = x + (long long)n * H * W * x_ld;
    const float *p11 = img + ((long long)y1 * W + x1) * x_ld;
    const float *p12 = img + ((long long)y1 * W + x2) * x_ld;
    const float *p21 = img + ((long long)y2 * W + x1) * x_ld;
    const float *p22 = img + ((long long)y2 * W + x2) * x_ld;
    float *dst = cols + (m * 9 + tap) * C;
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane * 4; c < C; c += 256) {
        const floatx4 v1 = (y1ok && x1ok) ? *reinterpret_cast<const floatx4 *>(p11 + c) : zero;
        const floatx4 v2 = (y1ok && x2ok) ? *reinterpret_cast<const floatx4 *>(p12 + c) : zero;
        const floatx4 v3 = (y2ok && x1ok) ? *reinterpret_cast<const floatx4 *>(p21 + c) : zero;
        const floatx4 v4 = (y2ok && x2ok) ? *reinterpret_cast<const floatx4 *>(p22 + c) : zero;
        floatx4 r;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float t = w1 * v1[u] + w2 * v2[u];
            t = t + w3 * v3[u];
            t = t + w4 * v4[u];
            r[u] = t * mask;
        }
        *reinterpret_cast<floatx4 *>(dst + c) = r;
    }
}

}  // namespace

extern "C" int ppy_dcnv2_sample_f32(const float *x, int x_ld, const float *offset_mask, int om_ld, float *cols,
                                    int N, int H, int W, int C, int Ho, int Wo, int stride, int pad,
                                    void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && offset_mask && cols);
    PPY_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && x_ld >= C && x_ld % 4 == 0 && om_ld >= 27);
    PPY_CHECK_ARG(stride > 0 && pad >= 0);
    PPY_CHECK_ARG(Ho == (H + 2 * pad - 2) / stride && Wo == (W + 2 * pad - 2) / stride);  // reference :567-568
    PPY_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)cols & 15) == 0);
    const long long waves = (long long)N * Ho * Wo * 9;
    const long long blocks = (waves + 3) / 4;
    hipLaunchKernelGGL(dcn_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x_ld,
                       offset_mask, om_ld, cols, N, H, W, C, Ho, Wo, stride, pad);
    return ppy_launch_status();
}
