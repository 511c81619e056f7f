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

// Backward of the sampling: given d loss / d columns, (a) scatter into the four corners of every sample (d x: float atomics --
// sampling positions are data dependent, several samples hit one pixel; the reference's CUDA extension does the same,
// external/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:197-262), (b) the gradient of the two learned offsets and of the mask LOGIT
// of every (output pixel, tap), reduced over the channels inside the wave (deterministic; :264-327).  What autograd
// computes for the reference's pure-PyTorch DCNv2.forward: floor() has no gradient, the fractional parts lh = y - floor(y)
// have gradient 1, clamp() passes the gradient where the position was inside [0, H+2p-1] (bounds included).
__global__ void __launch_bounds__(256) dcn_sample_bwd_kernel(const float *__restrict__ x, int x_ld, const float *__restrict__ om,
                                                             int om_ld, const float *__restrict__ dcols, float *__restrict__ dx,
                                                             int dx_ld, float *__restrict__ dom, int dom_ld, int N, int H, int W,
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
    const float ymax = (float)(H + 2 * pad) - 1.0f, xmax = (float)(W + 2 * pad) - 1.0f;
    const float py0 = ((float)(ho * stride + pad) + (float)(kh - 1)) + off_y;
    const float px0 = ((float)(wo * stride + pad) + (float)(kw - 1)) + off_x;
    const float gate_y = (py0 >= 0.0f && py0 <= ymax) ? 1.0f : 0.0f;      // clamp backward
    const float gate_x = (px0 >= 0.0f && px0 <= xmax) ? 1.0f : 0.0f;
    float py = fminf(fmaxf(py0, 0.0f), ymax);
    float px = fminf(fmaxf(px0, 0.0f), xmax);
    py = py + (float)n * (float)Hp;
    const float y1f = floorf(py), x1f = floorf(px);
    const float lh = py - y1f, lw = px - x1f;
    const float hh = 1.0f - lh, hw = 1.0f - lw;
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    const int y1 = (int)y1f - n * Hp - pad, x1 = (int)x1f - pad;
    const int y2 = y1 + 1, x2 = x1 + 1;
    const bool ok11 = (unsigned)y1 < (unsigned)H && (unsigned)x1 < (unsigned)W, ok12 = (unsigned)y1 < (unsigned)H && (unsigned)x2 < (unsigned)W;
    const bool ok21 = (unsigned)y2 < (unsigned)H && (unsigned)x1 < (unsigned)W, ok22 = (unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W;
    const long long pix = ((long long)n * H + y1) * W + x1;
    const float *p11 = x + pix * x_ld, *p12 = p11 + x_ld, *p21 = p11 + (long long)W * x_ld, *p22 = p21 + x_ld;
    float *q11 = dx + pix * dx_ld, *q12 = q11 + dx_ld, *q21 = q11 + (long long)W * dx_ld, *q22 = q21 + dx_ld;
    const float *g = dcols + (m * 9 + tap) * C;
    float s_lh = 0.f, s_lw = 0.f, s_mask = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float gc = g[c];
        const float v1 = ok11 ? p11[c] : 0.f, v2 = ok12 ? p12[c] : 0.f, v3 = ok21 ? p21[c] : 0.f, v4 = ok22 ? p22[c] : 0.f;
        s_mask += gc * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
        const float gm = gc * mask;                     // d loss / d (blend)
        if (ok11) atomicAdd(q11 + c, gm * w1);
        if (ok12) atomicAdd(q12 + c, gm * w2);
        if (ok21) atomicAdd(q21 + c, gm * w3);
        if (ok22) atomicAdd(q22 + c, gm * w4);
        // d blend / d lh = hw (v3 - v1) + lw (v4 - v2);  d blend / d lw = hh (v2 - v1) + lh (v4 - v3)
        s_lh += gm * (hw * (v3 - v1) + lw * (v4 - v2));
        s_lw += gm * (hh * (v2 - v1) + lh * (v4 - v3));
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
        s_lh += __shfl_xor(s_lh, sft);
        s_lw += __shfl_xor(s_lw, sft);
        s_mask += __shfl_xor(s_mask, sft);
    }
    if (lane == 0) {
        float *d = dom + m * dom_ld;
        d[2 * tap] = s_lh * gate_y;
        d[2 * tap + 1] = s_lw * gate_x;
        d[18 + tap] = s_mask * (mask * (1.0f - mask));
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

extern "C" size_t ppy_dcnv2_backward_workspace_bytes(int N, int H, int W, int C, int K, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || stride <= 0 || pad < 0) return 0;
    const int Ho = (H + 2 * pad - 2) / stride, Wo = (W + 2 * pad - 2) / stride;
    if (Ho <= 0 || Wo <= 0) return 0;
    const size_t cols = ((size_t)N * Ho * Wo * 9 * C * sizeof(float) + 255) / 256 * 256;
    const size_t wg = ppy_conv2d_wgrad_workspace_bytes(N, Ho, Wo, 9 * C, K, 1, 1, 1, 0);
    const size_t dg = ppy_conv2d_dgrad_workspace_bytes(N, Ho, Wo, 9 * C, K, 1, 1, 1, 0, -1, 0);
    return cols + ((wg > dg ? wg : dg) + 255) / 256 * 256;
}

// d x (sampling path only: the caller adds conv_offset's own data gradient), d offset_mask (raw 27 channels: 18 offsets, 9 mask
// LOGITS), d w -- arg order after dcn_v2_backward of the reference's extension (external/DCNv2/src/dcn_v2.h:41-55): input,
// weight, offset+mask, grad_output.
extern "C" int ppy_dcnv2_backward_f32(const float *x, int x_ld, const float *w_krsc, const float *offset_mask, int om_ld,
                                      const float *dy, int dy_ld, float *dx, int dx_ld, float *d_offset_mask, int dom_ld,
                                      float *dw_krsc, int N, int H, int W, int C, int K, int stride, int pad, void *ws,
                                      size_t ws_bytes, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && w_krsc && offset_mask && dy && dx && d_offset_mask && dw_krsc);
    PPY_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && K > 0 && stride > 0 && pad >= 0);
    PPY_CHECK_ARG(x_ld >= C && x_ld % 4 == 0 && dx_ld >= C && om_ld >= 27 && dom_ld >= 27 && dy_ld >= K);
    const int Ho = (H + 2 * pad - 2) / stride, Wo = (W + 2 * pad - 2) / stride;
    PPY_CHECK_ARG(Ho > 0 && Wo > 0);
    const size_t need = ppy_dcnv2_backward_workspace_bytes(N, H, W, C, K, stride, pad);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255) != 0) return PPY_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *cols = (float *)ws;
    const size_t cols_bytes = ((size_t)N * Ho * Wo * 9 * C * sizeof(float) + 255) / 256 * 256;
    void *rest = (char *)ws + cols_bytes;
    const size_t rest_bytes = ws_bytes - cols_bytes;
    // (1) the columns again, (2) d w = dy^T . columns, (3) d columns = dy . w (over the columns buffer), (4) back through the sampling
    int rc = ppy_dcnv2_sample_f32(x, x_ld, offset_mask, om_ld, cols, N, H, W, C, Ho, Wo, stride, pad, stream);
    if (rc != PPY_OK) return rc;
    rc = ppy_conv2d_wgrad_f32(cols, 9 * C, dy, dy_ld, dw_krsc, N, Ho, Wo, 9 * C, K, 1, 1, 1, 0, nullptr, nullptr, rest, rest_bytes, stream);
    if (rc != PPY_OK) return rc;
    rc = ppy_conv2d_dgrad_f32(dy, dy_ld, w_krsc, cols, 9 * C, N, Ho, Wo, 9 * C, K, 1, 1, 1, 0, -1, 0, nullptr, rest, rest_bytes, stream);
    if (rc != PPY_OK) return rc;
    if (hipMemset2DAsync(dx, (size_t)dx_ld * 4, 0, (size_t)C * 4, (size_t)N * H * W, st) != hipSuccess) return PPY_ERR_LAUNCH;
    const long long waves = (long long)N * Ho * Wo * 9;
    hipLaunchKernelGGL(dcn_sample_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, x, x_ld, offset_mask, om_ld, cols,
                       dx, dx_ld, d_offset_mask, dom_ld, N, H, W, C, Ho, Wo, stride, pad);
    return ppy_launch_status();
}
