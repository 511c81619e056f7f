// Training-step kernels around the convolutions (BASELINE config 5): everything the
// reference's train.py:416-443 step does besides F.conv2d and its backward (csrc/conv_bwd.hip) -- BatchNorm on batch
// statistics forward / backward (torch.nn.BatchNorm2d in training mode: the reference never calls .eval() before its
// loop), the activation derivative, nearest x2 upsample backward (model/head.py:396-397), SPP max-pool backward
// (model/custom_layers.py:281-290), DropBlock (custom_layers.py:303-342) and the SGD-momentum update with L2
// (train.py:271-280).  All tensors fp32 NHWC (pixel stride ld); all of it HBM-bound elementwise / reduction work:
// 16-byte accesses, one or two passes over the data, deterministic reductions (fixed slice order, no float atomics).
#include "common.h"

#include <cstdlib>

namespace {

constexpr int BN_CH = 64;         // channels per workgroup of the statistics kernels (one wave = one pixel x 64 channels)
constexpr int BN_MAX_SLICES = 256;

struct BnArgs {
    const float *x, *dy, *y;
    float *out;
    int x_ld, dy_ld, y_ld, out_ld;
    int P, C, slices, pix_per_slice, act;
    const float *mean, *invstd, *gamma, *beta, *sum_dz, *sum_dzx, *res;
    int res_ld;
    float *part;
};

// ---- forward statistics: per channel (n, mean, M2) of a pixel slice.  256 threads = 16 pixel lanes x 16 groups of 4 channels
// (16-byte loads).  Two passes over the slice -- its mean first, then the squared deviations from it (the slice is a few dozen
// KB, so the second pass is served by L2) -- which is as accurate as Welford's update without a division per element; slices
// are merged with Chan's formula.  (Measured alternative: ONE pass with every thread shifting by its own first sample and a
// Chan merge of the 16 pixel lanes -- 16.07 instead of 16.13 ms per R50 step, the second pass costs that little, and noisier
// gradients in tests/test_gpu_train_step.py; not kept.)
__global__ void __launch_bounds__(256) bn_stats_partial_kernel(const BnArgs p) {
    __shared__ float s_1[16][BN_CH], s_mean[BN_CH];
    const int cg = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * BN_CH + cg * 4;
    const int p0 = blockIdx.y * p.pix_per_slice, p1 = min(p0 + p.pix_per_slice, p.P);
    const float n = (float)(p1 - p0);
    const bool vec = c + 3 < p.C && (p.x_ld & 3) == 0;
    for (int pass = 0; pass < 2; ++pass) {
        floatx4 a = {0.f, 0.f, 0.f, 0.f}, m = {0.f, 0.f, 0.f, 0.f};
        if (pass == 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = s_mean[cg * 4 + e];
        if (c < p.C) {
            if (vec) {
                for (int q = p0 + pl; q < p1; q += 16) {
                    const floatx4 v = *reinterpret_cast<const floatx4 *>(p.x + (long long)q * p.x_ld + c) - m;
                    a += pass ? v * v : v;
                }
            } else {
                for (int e = 0; e < 4; ++e)
                    if (c + e < p.C)
                        for (int q = p0 + pl; q < p1; q += 16) {
                            const float v = p.x[(long long)q * p.x_ld + c + e] - m[e];
                            a[e] += pass ? v * v : v;
                        }
            }
        }
        __syncthreads();                                   // (pass 1: everybody has read s_mean / finished with s_1)
#pragma unroll
        for (int e = 0; e < 4; ++e) s_1[pl][cg * 4 + e] = a[e];
        __syncthreads();
        if (threadIdx.x < BN_CH) {
            float t = 0.f;
#pragma unroll
            for (int l = 0; l < 16; ++l) t += s_1[l][threadIdx.x];
            const int cc = blockIdx.x * BN_CH + threadIdx.x;
            if (pass == 0) {
                s_mean[threadIdx.x] = t / n;
            } else if (cc < p.C) {
                float *o = p.part + ((long long)blockIdx.y * p.C + cc) * 3;
                o[0] = n;
                o[1] = s_mean[threadIdx.x];
                o[2] = t;
            }
        }
        __syncthreads();
    }
}

// Chan's merge of two (n, mean, M2) triples
__device__ __forceinline__ void chan_merge(float &n, float &mean, float &m2, float nb, float mb, float qb) {
    if (nb > 0.f) {
        const float d = mb - mean, nt = n + nb;
        mean += d * (nb / nt);
        m2 += qb + d * d * (n * nb / nt);
        n = nt;
    }
}
// 16 channels x 16 lanes per workgroup: lane l merges the slices l, l+16, ... in order, then a 4-level tree over the lanes
// (fixed order: run-to-run identical); mean, 1/sqrt(biased var + eps); running statistics as torch.nn.BatchNorm2d does:
// running = (1 - momentum) * running + momentum * {mean, UNBIASED var}
constexpr int FIN_CH = 16;
// body of bn_stats_final_kernel for the channel block `cb` (16 channels); all 256 threads call
__device__ __forceinline__ void bn_stats_final_body(const float *part, int C, int slices, float eps, float momentum, float *mean_out,
                                                    float *invstd_out, float *running_mean, float *running_var, int cb,
                                                    float (&s_n)[16][FIN_CH], float (&s_m)[16][FIN_CH], float (&s_q)[16][FIN_CH]) {
    const int cl = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int c = cb * FIN_CH + cl;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (c < C)
        for (int s = l; s < slices; s += 16) {
            const float *o = part + ((long long)s * C + c) * 3;
            chan_merge(n, mean, m2, o[0], o[1], o[2]);
        }
    s_n[l][cl] = n; s_m[l][cl] = mean; s_q[l][cl] = m2;
    __syncthreads();
    for (int w = 8; w > 0; w >>= 1) {
        if (l < w) {
            chan_merge(n, mean, m2, s_n[l + w][cl], s_m[l + w][cl], s_q[l + w][cl]);
            s_n[l][cl] = n; s_m[l][cl] = mean; s_q[l][cl] = m2;
        }
        __syncthreads();
    }
    if (l != 0 || c >= C) return;
    const float var = m2 / n;
    mean_out[c] = mean;
    invstd_out[c] = 1.0f / sqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
}
__global__ void __launch_bounds__(256) bn_stats_final_kernel(const float *part, int C, int slices, float eps, float momentum,
                                                             float *mean_out, float *invstd_out, float *running_mean,
                                                             float *running_var) {
    __shared__ float s_n[16][FIN_CH], s_m[16][FIN_CH], s_q[16][FIN_CH];
    bn_stats_final_body(part, C, slices, eps, momentum, mean_out, invstd_out, running_mean, running_var, (int)blockIdx.x, s_n, s_m, s_q);
}

// Round 3: "the last workgroup finalises".  A reduction kernel whose partial results are combined by a second, tiny launch (6 us of
// launch latency for a few KB: 132 such launches per training step) takes a ticket per column block when its partials are
// written (release fence, device-scope atomic); the workgroup that draws the last ticket (acquire fence) runs the second launch's
// body itself -- the same code, the same order, so the results are bit-identical to the two-launch form.  atomicInc wraps the
// counter back to 0 with the last ticket: the counters need no reset.  One set per reduction kind; launches of one kind must
// not overlap on a device (the training step is one stream).
// MEASURED AND NOT KEPT (opt-in with PPY_BN_FUSE_FINAL=1; tests/test_gpu_train_ops.py keeps both forms bit-identical): the R50vd
// training step 12.36 -> 14.12 ms.  The 56 second launches it removes cost 0.33 ms; the device-scope release fence that every one
// of the thousands of reduction workgroups now executes (an L2 write-back on a part with eight non-coherent L2s) costs 2 ms.
// A 6 us launch is cheaper than a fence per workgroup.
struct FinArgs {
    float eps, momentum;
    float *mean, *invstd, *running_mean, *running_var;      // forward statistics
    float *dbeta, *dgamma;                                   // backward sums
    int fuse;
};
__device__ unsigned g_tickets[3][512];      // [merge | slice statistics | backward sums][column block]
__device__ __forceinline__ bool last_ticket(int kind, int col, unsigned count) {
    __shared__ unsigned s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicInc(&g_tickets[kind][col], count - 1) == count - 1 ? 1u : 0u;
    __syncthreads();
    const bool last = s_last != 0;
    if (last) __threadfence();
    return last;
}

// First level of a long list of slices (the convolution epilogues write one per wave row-tile: thousands for the 152x152 maps):
// workgroup (channel group, g) merges the slices [g * per_group, (g + 1) * per_group) into one triple, same order as the final kernel
__global__ void __launch_bounds__(256) bn_stats_merge_kernel(const float *part, int C, int slices, int per_group, float *out, const FinArgs fin) {
    __shared__ float s_n[16][FIN_CH], s_m[16][FIN_CH], s_q[16][FIN_CH];
    const int cl = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int c = blockIdx.x * FIN_CH + cl;
    const int s0 = blockIdx.y * per_group, s1 = min(s0 + per_group, slices);
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (c < C)
        for (int s = s0 + l; s < s1; s += 16) {
            const float *o = part + ((long long)s * C + c) * 3;
            chan_merge(n, mean, m2, o[0], o[1], o[2]);
        }
    s_n[l][cl] = n; s_m[l][cl] = mean; s_q[l][cl] = m2;
    __syncthreads();
    for (int w = 8; w > 0; w >>= 1) {
        if (l < w) {
            chan_merge(n, mean, m2, s_n[l + w][cl], s_m[l + w][cl], s_q[l + w][cl]);
            s_n[l][cl] = n; s_m[l][cl] = mean; s_q[l][cl] = m2;
        }
        __syncthreads();
    }
    if (l == 0 && c < C) {
        float *o = out + ((long long)blockIdx.y * C + c) * 3;
        o[0] = n; o[1] = mean; o[2] = m2;
    }
    if (!fin.fuse || !last_ticket(0, (int)blockIdx.x, gridDim.y)) return;
    bn_stats_final_body(out, C, (int)gridDim.y, fin.eps, fin.momentum, fin.mean, fin.invstd, fin.running_mean, fin.running_var, (int)blockIdx.x,
                        s_n, s_m, s_q);
}

// ---- forward apply: y = act((x - mean) * invstd * gamma + beta [+ res]); 4 channels per element of work, the per-channel
// parameters as 16-byte loads (L1 / L2 resident).  A workgroup owns a run of pixels (at most two images), so that the optional
// per-image max|y| (the operand scale of a following f16x2 convolution, amax_track2 in common.h) costs one or two atomics per wave.
__global__ void __launch_bounds__(256) bn_apply_kernel(const BnArgs p, int pix_per_block, int hw, float *amax_out) {
    const int c4 = p.C >> 2;
    const long long q0 = (long long)blockIdx.x * pix_per_block;
    const long long q1 = q0 + pix_per_block < p.P ? q0 + pix_per_block : p.P;
    const long long total = (q1 - q0) * c4;
    const int n_lo = (int)(q0 / hw), n_hi = (int)((q1 - 1) / hw);
    const long long bnd = (long long)(n_lo + 1) * hw;
    float amx_lo = 0.f, amx_hi = 0.f;
    // (element i = (pixel, channel group): advanced by 256 per iteration without a division in the loop)
    const int dq = 256 / c4, dr = 256 - dq * c4;
    int cg = (int)(threadIdx.x % c4);
    long long q = q0 + threadIdx.x / c4;
    for (long long i = threadIdx.x; i < total; i += 256) {
        const int c = cg * 4;
        const floatx4 v = *reinterpret_cast<const floatx4 *>(p.x + q * p.x_ld + c);
        floatx4 r = {0.f, 0.f, 0.f, 0.f};
        if (p.res) r = *reinterpret_cast<const floatx4 *>(p.res + q * p.res_ld + c);
        const floatx4 mu = *reinterpret_cast<const floatx4 *>(p.mean + c), is = *reinterpret_cast<const floatx4 *>(p.invstd + c);
        const floatx4 ga = *reinterpret_cast<const floatx4 *>(p.gamma + c), be = *reinterpret_cast<const floatx4 *>(p.beta + c);
        floatx4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = ppy_apply_act((v[k] - mu[k]) * (is[k] * ga[k]) + be[k] + r[k], p.act);
        *reinterpret_cast<floatx4 *>(p.out + q * p.out_ld + c) = o;
        const float rmx = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
        amx_lo = fmaxf(amx_lo, q < bnd ? rmx : 0.0f);
        amx_hi = fmaxf(amx_hi, q < bnd ? 0.0f : rmx);
        q += dq;
        cg += dr;
        if (cg >= c4) { cg -= c4; ++q; }
    }
    if (amax_out) amax_track2(amx_lo, amx_hi, n_lo, n_hi, amax_out, blockIdx.x * 4 + (threadIdx.x >> 6));
}

__device__ __forceinline__ float act_grad(float y, int act) {      // derivative from the OUTPUT: y > 0 <=> pre-activation > 0
    if (act == PPY_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == PPY_ACT_LEAKY) return y > 0.f ? 1.f : 0.1f;
    return 1.f;
}

// ---- backward reduction: per channel sum(dz), sum(dz * xhat), dz = dy * act'(y), xhat = (x - mean) * invstd
// body of bn_bwd_final_kernel for the channel block `cb` (16 channels); all 256 threads call
__device__ __forceinline__ void bn_bwd_final_body(const float *part, int C, int slices, float *dbeta, float *dgamma, int cb,
                                                  float (&s_a)[16][FIN_CH], float (&s_b)[16][FIN_CH]) {
    const int cl = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int c = cb * FIN_CH + cl;
    float a = 0.f, b = 0.f;
    if (c < C)
        for (int s = l; s < slices; s += 16) {
            a += part[((long long)s * C + c) * 2];
            b += part[((long long)s * C + c) * 2 + 1];
        }
    s_a[l][cl] = a; s_b[l][cl] = b;
    __syncthreads();
    for (int w = 8; w > 0; w >>= 1) {
        if (l < w) {
            s_a[l][cl] += s_a[l + w][cl];
            s_b[l][cl] += s_b[l + w][cl];
        }
        __syncthreads();
    }
    if (l == 0 && c < C) {
        dbeta[c] = s_a[0][cl];
        dgamma[c] = s_b[0][cl];
    }
    __syncthreads();
}
__global__ void __launch_bounds__(256) bn_bwd_partial_kernel(const BnArgs p, const FinArgs fin) {
    // a thread = 4 channels (16-byte loads of dy, y, x) x one of 16 pixel lanes (round 4; a thread per channel with 4-byte loads left the
    // three tensors at 3.4 TB/s); the 16 lanes are added in a fixed order
    __shared__ float s_a[16][BN_CH], s_b[16][BN_CH];
    __shared__ float f_a[16][FIN_CH], f_b[16][FIN_CH];
    const int cg = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * BN_CH + cg * 4;
    const int p0 = blockIdx.y * p.pix_per_slice, p1 = min(p0 + p.pix_per_slice, p.P);
    floatx4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (c + 3 < p.C) {           // (C % 4 == 0: the entry point checks it)
        const floatx4 mu = *reinterpret_cast<const floatx4 *>(p.mean + c), is = *reinterpret_cast<const floatx4 *>(p.invstd + c);
        for (int q = p0 + pl; q < p1; q += 16) {
            const floatx4 dy = *reinterpret_cast<const floatx4 *>(p.dy + (long long)q * p.dy_ld + c);
            const floatx4 y = *reinterpret_cast<const floatx4 *>(p.y + (long long)q * p.y_ld + c);
            const floatx4 x = *reinterpret_cast<const floatx4 *>(p.x + (long long)q * p.x_ld + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dz = dy[e] * act_grad(y[e], p.act);
                a[e] += dz;
                b[e] += dz * ((x[e] - mu[e]) * is[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s_a[pl][cg * 4 + e] = a[e];
        s_b[pl][cg * 4 + e] = b[e];
    }
    __syncthreads();
    if (threadIdx.x < BN_CH) {
        const int cc = blockIdx.x * BN_CH + threadIdx.x;
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            ta += s_a[l][threadIdx.x];
            tb += s_b[l][threadIdx.x];
        }
        if (cc < p.C) {
            p.part[((long long)blockIdx.y * p.C + cc) * 2] = ta;
            p.part[((long long)blockIdx.y * p.C + cc) * 2 + 1] = tb;
        }
    }
    if (!fin.fuse || !last_ticket(2, (int)blockIdx.x, gridDim.y)) return;
#pragma unroll 1
    for (int g = 0; g < BN_CH / FIN_CH; ++g)
        bn_bwd_final_body(p.part, p.C, (int)gridDim.y, fin.dbeta, fin.dgamma, (int)blockIdx.x * (BN_CH / FIN_CH) + g, f_a, f_b);
}
__global__ void __launch_bounds__(256) bn_bwd_final_kernel(const float *part, int C, int slices, float *dbeta, float *dgamma) {
    __shared__ float s_a[16][FIN_CH], s_b[16][FIN_CH];
    bn_bwd_final_body(part, C, slices, dbeta, dgamma, (int)blockIdx.x, s_a, s_b);
}
// dx = gamma * invstd * (dz - (sum_dz + xhat * sum_dzx) / P); a workgroup owns a run of pixels, as bn_apply_kernel, so that the
// optional per-image max|dx| (operand scale of an f16x2 weight gradient) costs one or two atomics per wave
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const BnArgs p, int pix_per_block, int hw, float *amax_out) {
    const int c4 = p.C >> 2;
    const long long q0 = (long long)blockIdx.x * pix_per_block;
    const long long q1 = q0 + pix_per_block < p.P ? q0 + pix_per_block : p.P;
    const long long total = (q1 - q0) * c4;
    const int n_lo = (int)(q0 / hw), n_hi = (int)((q1 - 1) / hw);
    const long long bnd = (long long)(n_lo + 1) * hw;
    const float invP = 1.0f / (float)p.P;
    float amx_lo = 0.f, amx_hi = 0.f;
    const int dq = 256 / c4, dr = 256 - dq * c4;
    int cg = (int)(threadIdx.x % c4);
    long long q = q0 + threadIdx.x / c4;
    for (long long i = threadIdx.x; i < total; i += 256) {
        const int c = cg * 4;
        const floatx4 x = *reinterpret_cast<const floatx4 *>(p.x + q * p.x_ld + c);
        const floatx4 dy = *reinterpret_cast<const floatx4 *>(p.dy + q * p.dy_ld + c);
        const floatx4 y = *reinterpret_cast<const floatx4 *>(p.y + q * p.y_ld + c);
        const floatx4 mu = *reinterpret_cast<const floatx4 *>(p.mean + c), isv = *reinterpret_cast<const floatx4 *>(p.invstd + c);
        const floatx4 ga = *reinterpret_cast<const floatx4 *>(p.gamma + c), sa = *reinterpret_cast<const floatx4 *>(p.sum_dz + c);
        const floatx4 sb = *reinterpret_cast<const floatx4 *>(p.sum_dzx + c);
        floatx4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (x[k] - mu[k]) * isv[k];
            const float dz = dy[k] * act_grad(y[k], p.act);
            o[k] = ga[k] * isv[k] * (dz - (sa[k] + xh * sb[k]) * invP);
        }
        *reinterpret_cast<floatx4 *>(p.out + q * p.out_ld + c) = o;
        const float rmx = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
        amx_lo = fmaxf(amx_lo, q < bnd ? rmx : 0.0f);
        amx_hi = fmaxf(amx_hi, q < bnd ? 0.0f : rmx);
        q += dq;
        cg += dr;
        if (cg >= c4) { cg -= c4; ++q; }
    }
    if (amax_out) amax_track2(amx_lo, amx_hi, n_lo, n_hi, amax_out, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// ---- activation backward alone (convolutions with bias and no BatchNorm have none in PP-YOLO; kept for completeness)
__global__ void __launch_bounds__(256) act_bwd_kernel(const float *dy, int dy_ld, const float *y, int y_ld, float *dx, int dx_ld,
                                                      long long P, int C, int act) {
    const int c4 = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P * c4) return;
    const int c = (int)(i % c4) * 4;
    const long long q = i / c4;
    const floatx4 g = *reinterpret_cast<const floatx4 *>(dy + q * dy_ld + c);
    const floatx4 v = *reinterpret_cast<const floatx4 *>(y + q * y_ld + c);
    floatx4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = g[k] * act_grad(v[k], act);
    *reinterpret_cast<floatx4 *>(dx + q * dx_ld + c) = o;
}

// ---- nearest x2 upsample backward: dx[n,h,w,:] = sum of the 2x2 block of dy   (accumulate != 0: dx += ...)
__global__ void __launch_bounds__(256) upsample2x_bwd_kernel(const float *dy, int dy_ld, float *dx, int dx_ld, int N, int H, int W, int C,
                                                             int accumulate) {
    const int c4 = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * H * W * c4) return;
    const int c = (int)(i % c4) * 4;
    long long q = i / c4;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H), n = (int)(q / H);
    const float *s = dy + (((long long)n * 2 * H + 2 * h) * (2 * W) + 2 * w) * dy_ld + c;
    const floatx4 a = *reinterpret_cast<const floatx4 *>(s), b = *reinterpret_cast<const floatx4 *>(s + dy_ld);
    const floatx4 d = *reinterpret_cast<const floatx4 *>(s + 2LL * W * dy_ld), e = *reinterpret_cast<const floatx4 *>(s + (2LL * W + 1) * dy_ld);
    float *o = dx + (((long long)n * H + h) * W + w) * dx_ld + c;
    floatx4 r = (a + b) + (d + e);
    if (accumulate) r += *reinterpret_cast<const floatx4 *>(o);
    *reinterpret_cast<floatx4 *>(o) = r;
}

// ---- AvgPool2d(2, 2) backward (vd shortcut, reference model/resnet_vd.py:29-33): every input pixel of a complete 2x2 window gets a
// quarter of the window's gradient; the odd last row / column (floor mode) gets none.
__global__ void __launch_bounds__(256) avgpool2x2_bwd_kernel(const float *dy, int dy_ld, float *dx, int dx_ld, int N, int H, int W, int C) {
    const int c4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * H * W * c4) return;
    const int c = (int)(i % c4) * 4;
    long long q = i / c4;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H), n = (int)(q / H);
    floatx4 r = {0.f, 0.f, 0.f, 0.f};
    if ((h >> 1) < Ho && (w >> 1) < Wo) {
        r = *reinterpret_cast<const floatx4 *>(dy + (((long long)n * Ho + (h >> 1)) * Wo + (w >> 1)) * dy_ld + c);
        r *= 0.25f;
    }
    *reinterpret_cast<floatx4 *>(dx + (((long long)n * H + h) * W + w) * dx_ld + c) = r;
}

// ---- MaxPool2d(3, 2, 1) backward (stem, reference model/resnet_vd.py:103): torch routes a window's gradient to its FIRST maximum in
// (h, w) scan order (implicit -inf padding).  Gather form, deterministic: an input pixel collects the gradients of the at most
// four windows that contain it and whose first maximum it is.
__global__ void __launch_bounds__(256) maxpool3x3s2_bwd_kernel(const float *x, int x_ld, const float *dy, int dy_ld, float *dx, int dx_ld,
                                                               int N, int H, int W, int C, int Ho, int Wo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * H * W * C) return;
    const int c = (int)(i % C);
    long long q = i / C;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H), n = (int)(q / H);
    const float *xn = x + (long long)n * H * W * x_ld + c;
    const float v = xn[((long long)h * W + w) * x_ld];
    float acc = 0.f;
    // windows (ho, wo) with 2*ho - 1 <= h <= 2*ho + 1
    for (int ho = (h + 1) / 2 - ((h & 1) ? 1 : 0) - 0; ho <= (h + 1) / 2; ++ho) {
        if (ho < 0 || ho >= Ho || h < 2 * ho - 1 || h > 2 * ho + 1) continue;
        for (int wo = (w + 1) / 2 - ((w & 1) ? 1 : 0); wo <= (w + 1) / 2; ++wo) {
            if (wo < 0 || wo >= Wo || w < 2 * wo - 1 || w > 2 * wo + 1) continue;
            // is (h, w) the first maximum of this window?
            bool first = true;
            for (int r = 0; r < 3 && first; ++r) {
                const int hh = 2 * ho - 1 + r;
                if ((unsigned)hh >= (unsigned)H) continue;
                for (int t = 0; t < 3; ++t) {
                    const int ww = 2 * wo - 1 + t;
                    if ((unsigned)ww >= (unsigned)W) continue;
                    const float u = xn[((long long)hh * W + ww) * x_ld];
                    const bool before = hh < h || (hh == h && ww < w);
                    if (u > v || (before && u == v) || (u != u && before)) { first = false; break; }      // (a NaN earlier in the scan wins, like torch)
                }
            }
            if (first && v == v) acc += dy[(((long long)n * Ho + ho) * Wo + wo) * dy_ld + c];
        }
    }
    dx[(((long long)n * H + h) * W + w) * dx_ld + c] = acc;
}

// ---- zero insertion: up[n, i*s, j*s, :] = dy[n, i, j, :], zeros elsewhere -- turns the data gradient of a stride-s convolution into
// the stride-1 one of the upsampled gradient (ppy_conv2d_dgrad_f32 is stride 1).
__global__ void __launch_bounds__(256) zero_insert_kernel(const float *dy, int dy_ld, float *up, int up_ld, int N, int Ho, int Wo, int C,
                                                          int H1, int W1, int s) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * H1 * W1 * C) return;
    const int c = (int)(i % C);
    long long q = i / C;
    const int w = (int)(q % W1);
    q /= W1;
    const int h = (int)(q % H1), n = (int)(q / H1);
    float v = 0.f;
    if (h % s == 0 && w % s == 0 && h / s < Ho && w / s < Wo) v = dy[(((long long)n * Ho + h / s) * Wo + w / s) * dy_ld + c];
    up[(((long long)n * H1 + h) * W1 + w) * up_ld + c] = v;
}

// ---- SPP backward.  Forward (custom_layers.py:281-290): cat([x, pool5(x), pool9(x), pool13(x)]), stride 1, implicit -inf
// padding.  torch's max_pool2d routes a window's gradient to its FIRST maximum in (h, w) scan order.  Deterministic gather
// form: kernel 1 records every window's argmax position, kernel 2 lets each input element collect the gradients of the
// windows (at most k*k) that chose it, in a fixed order, on top of the identity branch.
__global__ void __launch_bounds__(256) spp_argmax_kernel(const float *x, int x_ld, int N, int H, int W, int C, int k, short *arg) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * H * W * C) return;
    const int c = (int)(i % C);
    long long q = i / C;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H), n = (int)(q / H);
    const int r = k / 2;
    // torch's CPU kernel: maxval = -inf, maxindex = first element of the window; take v when (v > maxval) || isnan(v)
    const int h0 = max(h - r, 0), w0 = max(w - r, 0);
    float best = -__builtin_huge_valf();
    int bi = h0 * W + w0;
    for (int hh = h0; hh <= min(h + r, H - 1); ++hh)
        for (int ww = w0; ww <= min(w + r, W - 1); ++ww) {
            const float v = x[(((long long)n * H + hh) * W + ww) * x_ld + c];
            if (v > best || v != v) {
                best = v;
                bi = hh * W + ww;
            }
        }
    arg[i] = (short)bi;
}
__global__ void __launch_bounds__(256) spp_bwd_kernel(const float *dy, int dy_ld, const short *arg5, const short *arg9, const short *arg13,
                                                      float *dx, int dx_ld, int N, int H, int W, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * H * W * C) return;
    const int c = (int)(i % C);
    long long q = i / C;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H), n = (int)(q / H);
    const int me = h * W + w;
    float g = dy[(((long long)n * H + h) * W + w) * dy_ld + c];                       // identity branch: channels [0, C)
    const short *args[3] = {arg5, arg9, arg13};
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int r = 2 + 2 * b;                                                      // 5 -> 2, 9 -> 4, 13 -> 6
        for (int hh = max(h - r, 0); hh <= min(h + r, H - 1); ++hh)
            for (int ww = max(w - r, 0); ww <= min(w + r, W - 1); ++ww) {
                const long long o = (((long long)n * H + hh) * W + ww);
                if (args[b][o * C + c] == me) g += dy[o * dy_ld + (b + 1) * C + c];
            }
    }
    dx[(((long long)n * H + h) * W + w) * dx_ld + c] = g;
}

// Round 3: ONE launch when an image's map fits the LDS, and 86 window visits per element instead of 550.  A workgroup owns
// (image, SPP_CG channels) and keeps everything in LDS.  The pools are nested -- a 9x9 window is the union of the nine 5x5
// windows centred at offsets {-2, 0, 2}^2 from its centre, a 13x13 window the union of nine 9x9 windows (centres outside the
// image add nothing: their in-image part lies inside a neighbour's) -- and "the FIRST maximum in scan order" composes: the
// first position of the big window attaining its maximum is the first position of every sub-window that contains it, so it is
// the smallest index among the sub-windows' own argmaxes with the maximal value.  Forward: argmax of the 5x5 windows (25
// visits), then which of its nine sub-windows a 9x9 / 13x13 window selects (9 + 9 visits).  Backward the other way round:
// a 13x13 window's gradient goes to the 9x9 sub-window it selected, that one's total to its 5x5 sub-window, that one's to its
// argmax -- 9 + 9 + 25 visits, every sum in a fixed order.  (NaN inputs: the 5x5 level keeps torch's rule; above it NaN maxima
// never win -- the four-launch form below is NaN-faithful at every level.)  Two earlier fused forms that visited all
// 25 + 81 + 169 window elements twice from LDS took 1055 us (16 channels per workgroup: one wave per SIMD) and 416 us (4
// channels) against 388 us for the four launches: the visits, not where they are served from, are the cost.
constexpr int SPP_CG = 4;
__global__ void __launch_bounds__(256) spp_bwd_fused_kernel(const float *x, int x_ld, const float *dy, int dy_ld, float *dx, int dx_ld,
                                                            int H, int W, int C) {
    extern __shared__ float spp_smem[];
    const int HW = H * W, n = blockIdx.y, c0 = blockIdx.x * SPP_CG, E = HW * SPP_CG;
    float *sx = spp_smem, *m5 = sx + E, *m9 = m5 + E, *g9 = m9 + E, *g5 = g9 + E;      // [HW][SPP_CG] each
    short *a5 = reinterpret_cast<short *>(g5 + E), *a9 = a5 + E, *s9 = a9 + E, *s13 = s9 + E;
    for (int i = threadIdx.x; i < E; i += 256) {
        const int c = i % SPP_CG, q = i / SPP_CG;
        sx[i] = c0 + c < C ? x[((long long)n * HW + q) * x_ld + c0 + c] : 0.f;
    }
    __syncthreads();
    // ---- 5x5: torch's CPU kernel -- maxval = -inf, index = first element of the window; take v when v > maxval or isnan(v)
    for (int i = threadIdx.x; i < E; i += 256) {
        const int c = i % SPP_CG, q = i / SPP_CG;
        const int h = q / W, w = q - h * W;
        const int h0 = max(h - 2, 0), h1 = min(h + 2, H - 1);
        float best = -__builtin_huge_valf();
        int bi = h0 * W + max(w - 2, 0);
        for (int hh = h0; hh <= h1; ++hh) {
            float v[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) v[j] = sx[(hh * W + min(max(w - 2 + j, 0), W - 1)) * SPP_CG + c];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int ww = w - 2 + j;
                if ((unsigned)ww < (unsigned)W && (v[j] > best || v[j] != v[j])) {
                    best = v[j];
                    bi = hh * W + ww;
                }
            }
        }
        m5[i] = best;
        a5[i] = (short)bi;
    }
    __syncthreads();
    // ---- 9x9 from the nine 5x5 sub-windows, 13x13 from the nine 9x9 ones: larger value, then smaller (= earlier) argmax index
    for (int lvl = 0; lvl < 2; ++lvl) {
        const float *mv = lvl ? m9 : m5;
        const short *av = lvl ? a9 : a5;
        for (int i = threadIdx.x; i < E; i += 256) {
            const int c = i % SPP_CG, q = i / SPP_CG;
            const int h = q / W, w = q - h * W;
            float best = -__builtin_huge_valf();
            int bi = 0x7fffffff, bc = q;
#pragma unroll
            for (int dh = -2; dh <= 2; dh += 2)
#pragma unroll
                for (int dw = -2; dw <= 2; dw += 2) {
                    const int hh = h + dh, ww = w + dw;
                    if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) {
                        const int o = hh * W + ww;
                        const float v = mv[o * SPP_CG + c];
                        const int ai = av[o * SPP_CG + c];
                        if (v > best || (v == best && ai < bi) || bi == 0x7fffffff) {
                            best = v;
                            bi = ai;
                            bc = o;
                        }
                    }
                }
            if (lvl == 0) {
                m9[i] = best;
                a9[i] = (short)bi;
                s9[i] = (short)bc;
            } else {
                s13[i] = (short)bc;
            }
        }
        __syncthreads();
    }
    // ---- backward: 13x13 -> its 9x9 sub-window -> its 5x5 sub-window -> the argmax
    const float *dyn = dy + (long long)n * HW * dy_ld + c0;
    for (int lvl = 1; lvl >= 0; --lvl) {
        const short *sel = lvl ? s13 : s9;
        float *gout = lvl ? g9 : g5;
        for (int i = threadIdx.x; i < E; i += 256) {
            const int c = i % SPP_CG, q = i / SPP_CG;
            const int h = q / W, w = q - h * W;
            const bool live = c0 + c < C;
            float g = live ? dyn[(long long)q * dy_ld + (lvl ? 2 : 1) * C + c] : 0.f;      // the 9x9 / 5x5 branch's own gradient
#pragma unroll
            for (int dh = -2; dh <= 2; dh += 2)
#pragma unroll
                for (int dw = -2; dw <= 2; dw += 2) {
                    const int hh = h - dh, ww = w - dw;                                  // the window whose sub-window (dh, dw) is centred here
                    if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) {
                        const int o = hh * W + ww;
                        if (sel[o * SPP_CG + c] == q) g += lvl ? (live ? dyn[(long long)o * dy_ld + 3 * C + c] : 0.f) : g9[o * SPP_CG + c];
                    }
                }
            gout[i] = g;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < E; i += 256) {
        const int c = i % SPP_CG, q = i / SPP_CG;
        if (c0 + c >= C) continue;
        const int h = q / W, w = q - h * W;
        float g = dyn[(long long)q * dy_ld + c];                                          // identity branch
        const int h0 = max(h - 2, 0), h1 = min(h + 2, H - 1);
        for (int hh = h0; hh <= h1; ++hh) {
            short a[5];
            float gv[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int o = hh * W + min(max(w - 2 + j, 0), W - 1);
                a[j] = a5[o * SPP_CG + c];
                gv[j] = g5[o * SPP_CG + c];
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int ww = w - 2 + j;
                if ((unsigned)ww < (unsigned)W && a[j] == q) g += gv[j];
            }
        }
        dx[((long long)n * HW + q) * dx_ld + c0 + c] = g;
    }
}

// ---- DropBlock.  mask (1 = keep) is given; y = x * mask * (numel / sum(mask)).  The mask itself: seeds = u < gamma with a
// counter-based uniform u (one Philox-style hash per element), mask = 1 - maxpool3x3(seeds) with zero padding
// (custom_layers.py:330-336); sum(mask) by a deterministic two-level reduction.
__device__ __forceinline__ unsigned mix32(unsigned long long key) {       // splitmix64 finaliser, high 32 bits
    key += 0x9E3779B97F4A7C15ull;
    key = (key ^ (key >> 30)) * 0xBF58476D1CE4E5B9ull;
    key = (key ^ (key >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)((key ^ (key >> 31)) >> 32);
}
__global__ void __launch_bounds__(256) dropblock_mask_kernel(float *mask, int N, int H, int W, int C, float gamma, unsigned long long seed,
                                                             float *block_sums) {
    __shared__ float red[4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float m = 0.f;
    if (i < (long long)N * H * W * C) {
        const int c = (int)(i % C);
        long long q = i / C;
        const int w = (int)(q % W);
        q /= W;
        const int h = (int)(q % H), n = (int)(q / H);
        bool hit = false;
        for (int hh = max(h - 1, 0); hh <= min(h + 1, H - 1); ++hh)
            for (int ww = max(w - 1, 0); ww <= min(w + 1, W - 1); ++ww) {
                const unsigned long long id = ((((unsigned long long)n * H + hh) * W + ww) * C + c);
                const float u = (float)(mix32(seed ^ (id * 0xD1342543DE82EF95ull)) >> 8) * (1.0f / 16777216.0f);
                hit = hit || (u < gamma);
            }
        m = hit ? 0.f : 1.f;
        mask[i] = m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// Round 3: the same mask from ONE hash per element instead of nine (the hashes -- 64-bit multiplies -- were the kernel's whole
// time: 72 us per level).  A workgroup owns (image, DB_TH rows, DB_CG channels): the seeds of its rows and a one-row / one-column
// halo go to LDS as bytes (the halo rows are hashed again by the neighbour: 1.5x instead of 9x), then every element ORs its 3x3
// neighbourhood from LDS.  Same hash of the same element id: bit-identical masks; the block sums are counts (exact in fp32).
constexpr int DB_TH = 4, DB_CG = 32;
__global__ void __launch_bounds__(256) dropblock_mask_tile_kernel(float *mask, int N, int H, int W, int C, float gamma,
                                                                  unsigned long long seed, float *block_sums) {
    extern __shared__ unsigned char s_seed[];      // [DB_TH + 2][W + 2][DB_CG]
    __shared__ float red[4];
    const int cgs = (C + DB_CG - 1) / DB_CG, hbs = (H + DB_TH - 1) / DB_TH;
    int b = blockIdx.x;
    const int cgi = b % cgs;
    b /= cgs;
    const int hb = b % hbs, n = b / hbs;
    const int c0 = cgi * DB_CG, h0 = hb * DB_TH, PW = W + 2;
    for (int i = threadIdx.x; i < (DB_TH + 2) * PW * DB_CG; i += 256) {
        const int c = i % DB_CG, q = i / DB_CG;
        const int pw = q % PW, ph = q / PW;
        const int hh = h0 - 1 + ph, ww = pw - 1;
        unsigned char sd = 0;
        if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W && c0 + c < C) {
            const unsigned long long id = ((((unsigned long long)n * H + hh) * W + ww) * C + c0 + c);
            const float u = (float)(mix32(seed ^ (id * 0xD1342543DE82EF95ull)) >> 8) * (1.0f / 16777216.0f);
            sd = u < gamma ? 1 : 0;
        }
        s_seed[i] = sd;
    }
    __syncthreads();
    float m = 0.f;
    for (int i = threadIdx.x; i < DB_TH * W * DB_CG; i += 256) {
        const int c = i % DB_CG, q = i / DB_CG;
        const int w = q % W, hl = q / W;
        if (h0 + hl < H && c0 + c < C) {
            unsigned hit = 0;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) hit |= s_seed[((hl + dh) * PW + (w + dw)) * DB_CG + c];
            const float v = hit ? 0.f : 1.f;
            mask[(((long long)n * H + h0 + hl) * W + w) * C + c0 + c] = v;
            m += v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(256) sum_blocks_kernel(const float *block_sums, int n, float numel, float *scale_out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += block_sums[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) scale_out[0] = numel / red[0];
}
// y = x * mask * scale[0]  (forward on activations, backward on gradients: the same map)
__global__ void __launch_bounds__(256) dropblock_apply_kernel(const float *x, int x_ld, const float *mask, const float *scale, float *y,
                                                              int y_ld, long long P, int C) {
    const int c4 = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P * c4) return;
    const int c = (int)(i % c4) * 4;
    const long long q = i / c4;
    const floatx4 v = *reinterpret_cast<const floatx4 *>(x + q * x_ld + c);
    const floatx4 m = *reinterpret_cast<const floatx4 *>(mask + q * C + c);
    const float s = scale[0];
    floatx4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = v[k] * m[k] * s;          // (x * mask) * scale: the reference's order up to the division
    *reinterpret_cast<floatx4 *>(y + q * y_ld + c) = o;
}

// ---- SGD with momentum and L2 (torch.optim.SGD: d = g + wd * p; v = mu * v + d; p -= lr * v), first step: v = d
__global__ void __launch_bounds__(256) sgd_kernel(float *p, const float *g, float *v, long long n, float lr, float mu, float wd, int first) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float d = g[i] + wd * p[i];
    const float nv = first ? d : mu * v[i] + d;
    v[i] = nv;
    p[i] -= lr * nv;
}

// ---- small plumbing kernels of the training graph
// dst += src (a tensor with two consumers receives its second gradient)
__global__ void __launch_bounds__(256) add_inplace_kernel(float *dst, int dst_ld, const float *src, int src_ld, long long P, int C) {
    const int c4 = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P * c4) return;
    const int c = (int)(i % c4) * 4;
    const long long q = i / c4;
    floatx4 a = *reinterpret_cast<const floatx4 *>(dst + q * dst_ld + c);
    a += *reinterpret_cast<const floatx4 *>(src + q * src_ld + c);
    *reinterpret_cast<floatx4 *>(dst + q * dst_ld + c) = a;
}
// nearest x2 upsample forward (model/head.py:396-397): y[n, 2h+i, 2w+j, :] = x[n, h, w, :]
__global__ void __launch_bounds__(256) upsample2x_kernel(const float *x, int x_ld, float *y, int y_ld, int N, int H, int W, int C) {
    const int c4 = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * H * W * c4) return;
    const int c = (int)(i % c4) * 4;
    long long q = i / c4;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H), n = (int)(q / H);
    const floatx4 v = *reinterpret_cast<const floatx4 *>(x + (((long long)n * H + h) * W + w) * x_ld + c);
    float *o = y + (((long long)n * 2 * H + 2 * h) * (2 * W) + 2 * w) * y_ld + c;
    *reinterpret_cast<floatx4 *>(o) = v;
    *reinterpret_cast<floatx4 *>(o + y_ld) = v;
    *reinterpret_cast<floatx4 *>(o + 2LL * W * y_ld) = v;
    *reinterpret_cast<floatx4 *>(o + (2LL * W + 1) * y_ld) = v;
}
// per-channel sum over the pixels (gradient of a convolution bias): partial sums per slice, then ordered combine
__global__ void __launch_bounds__(256) chan_sum_partial_kernel(const float *dy, int dy_ld, int P, int C, int pix_per_slice, float *part) {
    __shared__ float s_a[4][BN_CH];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = blockIdx.x * BN_CH + cl;
    const int p0 = blockIdx.y * pix_per_slice, p1 = min(p0 + pix_per_slice, P);
    float a = 0.f;
    if (c < C)
        for (int q = p0 + pl; q < p1; q += 4) a += dy[(long long)q * dy_ld + c];
    s_a[pl][cl] = a;
    __syncthreads();
    if (pl == 0 && c < C) part[(long long)blockIdx.y * C + c] = (s_a[0][cl] + s_a[1][cl]) + (s_a[2][cl] + s_a[3][cl]);
}
// 16 channels x 16 lanes per workgroup: lane l adds the slices l, l + 16, ... in order, then a 4-level tree over the lanes (fixed
// order: run-to-run identical).  (One thread per channel walking up to 256 slices took 25 us per call.)
__global__ void __launch_bounds__(256) chan_sum_final_kernel(const float *part, int C, int slices, float *out) {
    __shared__ float s_a[16][16];
    const int cl = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float a = 0.f;
    if (c < C)
        for (int s = l; s < slices; s += 16) a += part[(long long)s * C + c];
    s_a[l][cl] = a;
    __syncthreads();
    for (int w = 8; w > 0; w >>= 1) {
        if (l < w) s_a[l][cl] += s_a[l + w][cl];
        __syncthreads();
    }
    if (l == 0 && c < C) out[c] = s_a[0][cl];
}

// ---- ExponentialMovingAverage of the trainable parameters (reference model/EMA.py:29-44): numpy's float32 arithmetic,
// shadow = decay * shadow + (1 - decay) * param with three separate roundings (no fused multiply-add)
__global__ void __launch_bounds__(256) ema_kernel(float *shadow, const float *param, long long n, float decay, float one_minus) {
#pragma clang fp contract(off)      // (HIP's __fmul_rn / __fadd_rn are plain operators: without this the sum contracts into an fma)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = decay * shadow[i], b = one_minus * param[i];
    shadow[i] = a + b;
}

static int slices_for(int P, int C) {
    const int cb = ceil_div(C, BN_CH);
    int sl = ceil_div(1024, cb);                       // ~4 workgroups per CU
    const int maxsl = P / 128 > 0 ? P / 128 : 1;
    if (sl > maxsl) sl = maxsl;
    if (sl > BN_MAX_SLICES) sl = BN_MAX_SLICES;
    return sl < 1 ? 1 : sl;
}
static inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }
static inline bool fuse_final() {      // opt-in (PPY_BN_FUSE_FINAL=1), read per call: measured SLOWER, see last_ticket
    const char *e = getenv("PPY_BN_FUSE_FINAL");
    return e && e[0] == '1';
}

}  // namespace

extern "C" size_t ppy_bn_train_workspace_bytes(int P, int C) { return (size_t)slices_for(P, C) * C * 3 * sizeof(float); }

extern "C" int ppy_bn_train_stats_f32(const float *x, int x_ld, int P, int C, float eps, float momentum, float *mean, float *invstd,
                                      float *running_mean, float *running_var, void *ws, size_t ws_bytes, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && mean && invstd && P > 0 && C > 0 && x_ld >= C);
    if (!ws || ws_bytes < ppy_bn_train_workspace_bytes(P, C)) return PPY_ERR_WORKSPACE;
    BnArgs p = {};
    p.x = x; p.x_ld = x_ld; p.P = P; p.C = C; p.part = (float *)ws;
    p.slices = slices_for(P, C);
    p.pix_per_slice = ceil_div(P, p.slices);
    p.slices = ceil_div(P, p.pix_per_slice);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(ceil_div(C, BN_CH), p.slices), dim3(256), 0, st, p);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(ceil_div(C, FIN_CH)), dim3(256), 0, st, (const float *)ws, C, p.slices, eps, momentum, mean,
                       invstd, running_mean, running_var);
    return ppy_launch_status();
}

// The second half of ppy_bn_train_stats_f32 for statistics that come out of a convolution's epilogue
// (ppy_conv2d_train_fwd_f32): partials [slices][C][3] = (n, mean, M2) per slice and channel -> mean, 1 / sqrt(biased var + eps),
// running statistics.  `partials` is used as scratch behind its first slices * C * 3 floats when there are many slices
// (needs (slices + ceil(slices / 64)) * C * 3 floats in all).
extern "C" int ppy_bn_train_stats_merge_f32(float *partials, size_t partials_bytes, int slices, int C, float eps, float momentum, float *mean,
                                            float *invstd, float *running_mean, float *running_var, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(partials && mean && invstd && slices > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    const float *src = partials;
    if (slices > 256) {
        const int per_group = 64, groups = ceil_div(slices, per_group);
        float *lvl = partials + (size_t)slices * C * 3;
        if (partials_bytes < ((size_t)slices + groups) * C * 3 * sizeof(float)) return PPY_ERR_WORKSPACE;
        FinArgs fin = {};
        fin.eps = eps; fin.momentum = momentum; fin.mean = mean; fin.invstd = invstd; fin.running_mean = running_mean; fin.running_var = running_var;
        fin.fuse = fuse_final() && ceil_div(C, FIN_CH) <= 512;
        hipLaunchKernelGGL(bn_stats_merge_kernel, dim3(ceil_div(C, FIN_CH), groups), dim3(256), 0, st, (const float *)partials, C, slices,
                           per_group, lvl, fin);
        if (fin.fuse) return ppy_launch_status();
        src = lvl;
        slices = groups;
    } else if (partials_bytes < (size_t)slices * C * 3 * sizeof(float)) {
        return PPY_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(ceil_div(C, FIN_CH)), dim3(256), 0, st, src, C, slices, eps, momentum, mean, invstd,
                       running_mean, running_var);
    return ppy_launch_status();
}

extern "C" int ppy_bn_train_apply_f32(const float *x, int x_ld, const float *mean, const float *invstd, const float *gamma,
                                      const float *beta, const float *residual, int res_ld, float *y, int y_ld, int P, int C, int act,
                                      int pixels_per_image, float *amax_out, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && mean && invstd && gamma && beta && y && P > 0 && C > 0 && C % 4 == 0 && x_ld >= C && y_ld >= C);
    PPY_CHECK_ARG(x_ld % 4 == 0 && y_ld % 4 == 0 && (!residual || (res_ld >= C && res_ld % 4 == 0)));
    PPY_CHECK_ARG((((uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0);      // 16-byte parameter loads
    PPY_CHECK_ARG(!amax_out || (pixels_per_image > 0 && P % pixels_per_image == 0));
    BnArgs p = {};
    p.x = x; p.x_ld = x_ld; p.out = y; p.out_ld = y_ld; p.P = P; p.C = C; p.act = act;
    p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta; p.res = residual; p.res_ld = res_ld;
    const int hw = amax_out ? pixels_per_image : P;
    int ppb = ceil_div(P, 4096);                      // ~16 workgroups per CU
    const int floor_ppb = ceil_div(4096, C);          // ... of at least 1024 16-byte elements each
    if (ppb < floor_ppb) ppb = floor_ppb;
    if (ppb > hw) ppb = hw;                           // a workgroup's pixels lie in at most two images
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ceil_div(P, ppb)), dim3(256), 0, (hipStream_t)stream, p, ppb, hw, amax_out);
    return ppy_launch_status();
}

extern "C" int ppy_bn_train_bwd_f32(const float *x, int x_ld, const float *y, int y_ld, const float *dy, int dy_ld, const float *mean,
                                    const float *invstd, const float *gamma, float *dx, int dx_ld, float *dgamma, float *dbeta, int P,
                                    int C, int act, int pixels_per_image, float *amax_dx, void *ws, size_t ws_bytes, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && y && dy && mean && invstd && gamma && dx && dgamma && dbeta && P > 0 && C > 0 && C % 4 == 0);
    PPY_CHECK_ARG(x_ld >= C && y_ld >= C && dy_ld >= C && dx_ld >= C && x_ld % 4 == 0 && y_ld % 4 == 0 && dy_ld % 4 == 0 && dx_ld % 4 == 0);
    PPY_CHECK_ARG((((uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)gamma | (uintptr_t)dgamma | (uintptr_t)dbeta) & 15) == 0);
    if (!ws || ws_bytes < ppy_bn_train_workspace_bytes(P, C)) return PPY_ERR_WORKSPACE;
    BnArgs p = {};
    p.x = x; p.x_ld = x_ld; p.y = y; p.y_ld = y_ld; p.dy = dy; p.dy_ld = dy_ld; p.out = dx; p.out_ld = dx_ld;
    p.P = P; p.C = C; p.act = act; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.part = (float *)ws;
    p.slices = slices_for(P, C);
    p.pix_per_slice = ceil_div(P, p.slices);
    p.slices = ceil_div(P, p.pix_per_slice);
    p.sum_dz = dbeta; p.sum_dzx = dgamma;
    hipStream_t st = (hipStream_t)stream;
    FinArgs fin = {};
    fin.dbeta = dbeta; fin.dgamma = dgamma;
    fin.fuse = fuse_final() && ceil_div(C, BN_CH) <= 512;
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(ceil_div(C, BN_CH), p.slices), dim3(256), 0, st, p, fin);
    if (!fin.fuse)
        hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(ceil_div(C, FIN_CH)), dim3(256), 0, st, (const float *)ws, C, p.slices, dbeta, dgamma);
    const int hw = amax_dx ? pixels_per_image : P;
    if (amax_dx && (pixels_per_image <= 0 || P % pixels_per_image != 0)) return PPY_ERR_BAD_ARG;
    int ppb = ceil_div(P, 4096);
    const int floor_ppb = ceil_div(4096, C);
    if (ppb < floor_ppb) ppb = floor_ppb;
    if (ppb > hw) ppb = hw;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ceil_div(P, ppb)), dim3(256), 0, st, p, ppb, hw, amax_dx);
    return ppy_launch_status();
}

extern "C" int ppy_act_bwd_f32(const float *dy, int dy_ld, const float *y, int y_ld, float *dx, int dx_ld, long long P, int C, int act,
                               void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(dy && y && dx && P > 0 && C > 0 && C % 4 == 0 && dy_ld % 4 == 0 && y_ld % 4 == 0 && dx_ld % 4 == 0);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, dy_ld, y, y_ld, dx, dx_ld, P, C,
                       act);
    return ppy_launch_status();
}

extern "C" int ppy_upsample2x_bwd_f32(const float *dy, int dy_ld, float *dx, int dx_ld, int N, int H, int W, int C, int accumulate,
                                      void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && dy_ld >= C && dx_ld >= C && dy_ld % 4 == 0 && dx_ld % 4 == 0);
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(blocks_for((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, dy_ld,
                       dx, dx_ld, N, H, W, C, accumulate);
    return ppy_launch_status();
}

extern "C" int ppy_avgpool2x2_bwd_f32(const float *dy, int dy_ld, float *dx, int dx_ld, int N, int H, int W, int C, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(dy && dx && N > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0 && dy_ld >= C && dx_ld >= C && dy_ld % 4 == 0 && dx_ld % 4 == 0);
    hipLaunchKernelGGL(avgpool2x2_bwd_kernel, dim3(blocks_for((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, dy_ld,
                       dx, dx_ld, N, H, W, C);
    return ppy_launch_status();
}

extern "C" int ppy_maxpool3x3s2_bwd_f32(const float *x, int x_ld, const float *dy, int dy_ld, float *dx, int dx_ld, int N, int H, int W,
                                        int C, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && x_ld >= C && dy_ld >= C && dx_ld >= C);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(blocks_for((long long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, x, x_ld, dy,
                       dy_ld, dx, dx_ld, N, H, W, C, Ho, Wo);
    return ppy_launch_status();
}

extern "C" int ppy_zero_insert_f32(const float *dy, int dy_ld, float *up, int up_ld, int N, int Ho, int Wo, int C, int H1, int W1,
                                   int stride, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(dy && up && N > 0 && Ho > 0 && Wo > 0 && C > 0 && stride > 0 && dy_ld >= C && up_ld >= C);
    PPY_CHECK_ARG(H1 >= (Ho - 1) * stride + 1 && W1 >= (Wo - 1) * stride + 1);
    hipLaunchKernelGGL(zero_insert_kernel, dim3(blocks_for((long long)N * H1 * W1 * C)), dim3(256), 0, (hipStream_t)stream, dy, dy_ld, up,
                       up_ld, N, Ho, Wo, C, H1, W1, stride);
    return ppy_launch_status();
}

extern "C" size_t ppy_spp_bwd_workspace_bytes(int N, int H, int W, int C) { return (size_t)3 * N * H * W * C * sizeof(short); }

extern "C" int ppy_spp_bwd_f32(const float *x, int x_ld, const float *dy, int dy_ld, float *dx, int dx_ld, int N, int H, int W, int C,
                               void *ws, size_t ws_bytes, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && x_ld >= C && dy_ld >= 4 * C && dx_ld >= C && H * W < 32768);
    if (!ws || ws_bytes < ppy_spp_bwd_workspace_bytes(N, H, W, C)) return PPY_ERR_WORKSPACE;
    const long long n = (long long)N * H * W * C;
    short *a5 = (short *)ws, *a9 = a5 + n, *a13 = a9 + n;
    hipStream_t st = (hipStream_t)stream;
    const int lds = H * W * SPP_CG * (int)(5 * sizeof(float) + 4 * sizeof(short));
    if (lds <= 150 * 1024 && N <= 65535) {
        static PpyLdsAttr attr;
        if (ppy_lds_attr(attr, (const void *)spp_bwd_fused_kernel, lds) != PPY_OK) return PPY_ERR_LAUNCH;
        hipLaunchKernelGGL(spp_bwd_fused_kernel, dim3((C + SPP_CG - 1) / SPP_CG, N), dim3(256), lds, st, x, x_ld, dy, dy_ld, dx, dx_ld, H, W, C);
        return ppy_launch_status();
    }
    hipLaunchKernelGGL(spp_argmax_kernel, dim3(blocks_for(n)), dim3(256), 0, st, x, x_ld, N, H, W, C, 5, a5);
    hipLaunchKernelGGL(spp_argmax_kernel, dim3(blocks_for(n)), dim3(256), 0, st, x, x_ld, N, H, W, C, 9, a9);
    hipLaunchKernelGGL(spp_argmax_kernel, dim3(blocks_for(n)), dim3(256), 0, st, x, x_ld, N, H, W, C, 13, a13);
    hipLaunchKernelGGL(spp_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, st, dy, dy_ld, a5, a9, a13, dx, dx_ld, N, H, W, C);
    return ppy_launch_status();
}

static long long dropblock_tiles(int N, int H, int W, int C) {
    return (long long)N * ((H + DB_TH - 1) / DB_TH) * ((C + DB_CG - 1) / DB_CG);
}
extern "C" size_t ppy_dropblock_workspace_bytes(int N, int H, int W, int C) {
    const long long a = blocks_for((long long)N * H * W * C), b = dropblock_tiles(N, H, W, C);
    return ((size_t)(a > b ? a : b) + 1) * sizeof(float);
}
extern "C" int ppy_dropblock_mask_f32(float *mask, float *scale_out, int N, int H, int W, int C, int block_size, float keep_prob,
                                      unsigned long long seed, void *ws, size_t ws_bytes, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(mask && scale_out && N > 0 && H >= block_size && W > 0 && C > 0 && block_size == 3 && keep_prob > 0.f && keep_prob <= 1.f);
    if (!ws || ws_bytes < ppy_dropblock_workspace_bytes(N, H, W, C)) return PPY_ERR_WORKSPACE;
    // gamma from the HEIGHT only, like the reference (custom_layers.py:306-325)
    const float h = (float)H, bs = (float)block_size;
    const float gamma = (h * h * (1.0f - keep_prob)) / (bs * bs * ((h - bs + 1.0f) * (h - bs + 1.0f)));
    const long long n = (long long)N * H * W * C;
    hipStream_t st = (hipStream_t)stream;
    const int lds = (DB_TH + 2) * (W + 2) * DB_CG;
    const long long tiles = dropblock_tiles(N, H, W, C);
    if (lds <= 64 * 1024 && tiles < (1LL << 31)) {
        hipLaunchKernelGGL(dropblock_mask_tile_kernel, dim3((unsigned)tiles), dim3(256), lds, st, mask, N, H, W, C, gamma, seed, (float *)ws);
        hipLaunchKernelGGL(sum_blocks_kernel, dim3(1), dim3(256), 0, st, (const float *)ws, (int)tiles, (float)n, scale_out);
        return ppy_launch_status();
    }
    hipLaunchKernelGGL(dropblock_mask_kernel, dim3(blocks_for(n)), dim3(256), 0, st, mask, N, H, W, C, gamma, seed, (float *)ws);
    hipLaunchKernelGGL(sum_blocks_kernel, dim3(1), dim3(256), 0, st, (const float *)ws, (int)blocks_for(n), (float)n, scale_out);
    return ppy_launch_status();
}
extern "C" int ppy_dropblock_apply_f32(const float *x, int x_ld, const float *mask, const float *scale, float *y, int y_ld, long long P,
                                       int C, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && mask && scale && y && P > 0 && C > 0 && C % 4 == 0 && x_ld >= C && y_ld >= C && x_ld % 4 == 0 && y_ld % 4 == 0);
    hipLaunchKernelGGL(dropblock_apply_kernel, dim3(blocks_for(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, x_ld, mask, scale, y, y_ld,
                       P, C);
    return ppy_launch_status();
}

extern "C" int ppy_sgd_momentum_f32(float *param, const float *grad, float *velocity, long long n, float lr, float momentum,
                                    float weight_decay, int first_step, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(param && grad && velocity && n > 0);
    hipLaunchKernelGGL(sgd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, velocity, n, lr, momentum,
                       weight_decay, first_step);
    return ppy_launch_status();
}

extern "C" int ppy_add_inplace_f32(float *dst, int dst_ld, const float *src, int src_ld, long long P, int C, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(dst && src && P > 0 && C > 0 && C % 4 == 0 && dst_ld >= C && src_ld >= C && dst_ld % 4 == 0 && src_ld % 4 == 0);
    hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks_for(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, dst, dst_ld, src, src_ld, P, C);
    return ppy_launch_status();
}

extern "C" int ppy_upsample2x_f32(const float *x, int x_ld, float *y, int y_ld, int N, int H, int W, int C, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && x_ld >= C && y_ld >= C && x_ld % 4 == 0 && y_ld % 4 == 0);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(blocks_for((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, x_ld, y,
                       y_ld, N, H, W, C);
    return ppy_launch_status();
}

extern "C" int ppy_channel_sum_f32(const float *dy, int dy_ld, int P, int C, float *out, void *ws, size_t ws_bytes, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(dy && out && P > 0 && C > 0 && dy_ld >= C);
    if (!ws || ws_bytes < ppy_bn_train_workspace_bytes(P, C)) return PPY_ERR_WORKSPACE;
    int sl = slices_for(P, C);
    const int pps = ceil_div(P, sl);
    sl = ceil_div(P, pps);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chan_sum_partial_kernel, dim3(ceil_div(C, BN_CH), sl), dim3(256), 0, st, dy, dy_ld, P, C, pps, (float *)ws);
    hipLaunchKernelGGL(chan_sum_final_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, st, (const float *)ws, C, sl, out);
    return ppy_launch_status();
}

extern "C" int ppy_ema_update_f32(float *shadow, const float *param, long long n, float decay, float one_minus_decay, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(shadow && param && n > 0);
    hipLaunchKernelGGL(ema_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, shadow, param, n, decay, one_minus_decay);
    return ppy_launch_status();
}
