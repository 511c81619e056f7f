// Backward of the convolution for the training step (BASELINE config 5): what torch
// autograd computes for the F.conv2d of Conv2dUnit.forward (reference model/custom_layers.py:243-253) under
// all_loss.backward() (reference train.py:441).  With the reference's freeze_at = 5 only the head trains, and every head
// convolution is 1x1 or 3x3 with stride 1 (model/head.py:146-231, :334-364).
//
//   dgrad:  dx[n,h,w,c] = sum_{k,r,s} dy[n, h+pad-r, w+pad-s, k] * w[k,r,s,c]
//           = the FORWARD convolution of dy with the flipped / transposed weights w'[c][r'][s'][k] = w[k][R-1-r'][S-1-s'][c]
//           and pad' = R-1-pad: one re-layout kernel (weights change every step, so it is per call), then the implicit-GEMM
//           kernels of conv_igemm.hip / conv_x3.hip (exact bf16x3 split of w', done here as well) do the work.
//   wgrad:  dw[k,r,s,c] = sum_{n,ho,wo} dy[n,ho,wo,k] * x[n, ho*stride+r-pad, wo*stride+s-pad, c]
//           a GEMM whose REDUCTION index is the pixel: both operands are pixel-major in HBM (NHWC), which is exactly the
//           operand order of v_mfma_f32_32x32x2_f32 -- lane l supplies A[i = l%32][kk = l/32] = dy[pixel kk][k0 + i] and
//           B[kk][j = l%32] = x[pixel kk'][c0 + j]: the 32 lanes of a half-wave read 128 contiguous bytes, no
//           transposition anywhere.  One workgroup = one 128x128 (k, c) tile of ONE tap over a slice of the pixels;
//           the slices are combined in a fixed order by a second kernel (deterministic, no atomics).  Exact fp32.
#include "conv_shared.h"

#include <cstdlib>

namespace {

// w[K][R][S][C] -> wt[C][R][S][Kp] with both taps flipped and zero rows for k in [K, Kp); also ones[C], zeros[C]
__global__ void __launch_bounds__(256) dgrad_weights_kernel(const float *w, float *wt, float *ones, float *zeros, int K, int Kp,
                                                            int R, int S, int C) {
    const long long total = (long long)C * R * S * Kp;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < C) {
        ones[i] = 1.0f;
        zeros[i] = 0.0f;
    }
    if (i >= total) return;
    const int k = (int)(i % Kp);
    long long t = i / Kp;
    const int s = (int)(t % S);
    t /= S;
    const int r = (int)(t % R);
    const int c = (int)(t / R);
    wt[i] = k < K ? w[(((long long)k * R + (R - 1 - r)) * S + (S - 1 - s)) * C + c] : 0.0f;
}

// The same re-layout AND the f16x2 split of w' in one launch (round 3; it was dgrad_weights_kernel + split_weights_f16_kernel,
// 23 pairs of ~8 us launches per training step): one workgroup per row c of w' -- kred = R*S*Kp values, at most DW_MAXV per
// thread, kept in registers between the maximum and the split.  Same arithmetic as conv_x3.hip's split_weights_f16_kernel
// (power-of-two scale that puts max|w'[c,:]| into [2^13, 2^14), two RNE fp16 terms, chunk-major planes [plane][kred/32][C][32]):
// bit-identical planes and scales.
constexpr int DW_MAXV = 40;
__global__ void __launch_bounds__(256) dgrad_weights_f16_kernel(const float *w, float *wt, float *ones, float *zeros,
                                                                unsigned short *planes, float *scale_out, int K, int Kp, int R, int S,
                                                                int C) {
    __shared__ float smax[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int kred = R * S * Kp;
    float v[DW_MAXV];
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < DW_MAXV; ++j) {
        const int i = tid + 256 * j;
        float x = 0.f;
        if (i < kred) {
            const int k = i % Kp;
            const int t = i / Kp;
            const int s2 = t % S, r2 = t / S;
            if (k < K) x = w[(((long long)k * R + (R - 1 - r2)) * S + (S - 1 - s2)) * C + c];
        }
        v[j] = x;
        mx = fmaxf(mx, fabsf(x));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) smax[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    int f = 267 - e;
    f = f < 103 ? 103 : (f > 167 ? 167 : f);
    const float sw = __uint_as_float((unsigned)f << 23), inv = __uint_as_float((unsigned)(254 - f) << 23);
    const long long n = (long long)C * kred;
#pragma unroll
    for (int j = 0; j < DW_MAXV; ++j) {
        const int i = tid + 256 * j;
        if (i < kred) {
            wt[(long long)c * kred + i] = v[j];
            const float x = v[j] * sw;
            const _Float16 h0 = (_Float16)x;
            const _Float16 h1 = (_Float16)(x - (float)h0);
            const long long o = PPY_X3_BBLOCK ? ((long long)(i >> 5) * C + c) * 32 + (i & 31) : (long long)c * kred + i;
            planes[o] = __builtin_bit_cast(unsigned short, h0);
            planes[n + o] = __builtin_bit_cast(unsigned short, h1);
        }
    }
    if (tid == 0) {
        scale_out[c] = inv;      // (ones[c] * inv)
        ones[c] = 1.0f;
        zeros[c] = 0.0f;
    }
}

// ---- the same products for ALL trainable weights of a step (ppy_train_prepare_weights_f16x2): three launches.
// prep_fwd_kernel: one workgroup per row k of some tensor = conv_x3.hip's split_weights_f16_kernel with scale = 1.
// prep_colmax_kernel / prep_dgrad_kernel: workgroup (tensor, 32 consecutive c, row split); a unit = (tap', 32 k): the 32 x 32
// tile w[k][R-1-r'][S-1-s'][c] is read as 32 rows of 128 contiguous bytes, turned in LDS and leaves as 2 KB of contiguous plane
// bytes per plane (and 32 x 128 B of the fp32 copy) -- dgrad_weights_f16_kernel reads the same tensor with one 4-byte load
// per line.  Same scale rule and the same two roundings: bit-identical planes.
__device__ __forceinline__ int prep_find(const PpyWeightPrep *d, int count, int id, bool rows) {
    int t = 0;
    while (t + 1 < count && (rows ? d[t + 1].row0 : d[t + 1].blk0) <= id) ++t;
    return t;
}
__global__ void __launch_bounds__(256) prep_fwd_kernel(const PpyWeightPrep *desc, int count) {
    __shared__ float smax[4];
    const int t = prep_find(desc, count, (int)blockIdx.x, true);
    const PpyWeightPrep d = desc[t];
    const int k = (int)blockIdx.x - d.row0, tid = threadIdx.x, K = d.K;
    const long long kred = (long long)d.R * d.S * d.C;
    const float *row = d.w + (long long)k * kred;
    unsigned short *out = (unsigned short *)d.fwd_planes;
    float mx = 0.f;
    for (long long i = tid; i < kred; i += 256) mx = fmaxf(mx, fabsf(row[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) smax[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    int f = 267 - e;
    f = f < 103 ? 103 : (f > 167 ? 167 : f);
    const float sw = __uint_as_float((unsigned)f << 23), inv = __uint_as_float((unsigned)(254 - f) << 23);
    const long long n = (long long)K * kred;
    for (long long i = tid; i < kred; i += 256) {
        const float v = row[i] * sw;
        const _Float16 h0 = (_Float16)v;
        const _Float16 h1 = (_Float16)(v - (float)h0);
        const long long o = PPY_X3_BBLOCK ? ((i >> 5) * K + k) * 32 + (i & 31) : (long long)k * kred + i;
        out[o] = __builtin_bit_cast(unsigned short, h0);
        out[n + o] = __builtin_bit_cast(unsigned short, h1);
    }
    if (tid == 0) d.fwd_scale[k] = inv;
}
// units of a (tensor, split): u = split, split + PPY_PREP_SPLIT, ... < R * S * (Kp / 32); unit u = (tap' = u / kchunks, kc = u % kchunks)
__global__ void __launch_bounds__(256) prep_colmax_kernel(const PpyWeightPrep *desc, int count, float *colmax) {
    __shared__ float red[8][32];
    const int t = prep_find(desc, count, (int)blockIdx.x, false);
    const PpyWeightPrep d = desc[t];
    const int local = (int)blockIdx.x - d.blk0, cgrp = local / PPY_PREP_SPLIT, sp = local - cgrp * PPY_PREP_SPLIT;
    const int c = cgrp * 32 + (threadIdx.x & 31), kl = threadIdx.x >> 5;
    const int kchunks = (d.K + 31) / 32, units = d.R * d.S * kchunks;
    float mx = 0.f;
    if (d.dgrad_planes)
        for (int u = sp; u < units; u += PPY_PREP_SPLIT) {
            const int tap = u / kchunks, kc = u - tap * kchunks;       // (max over all taps: the flip does not matter here)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kc * 32 + kl + 8 * j;
                if (k < d.K) mx = fmaxf(mx, fabsf(d.w[((long long)k * d.R * d.S + tap) * d.C + c]));
            }
        }
    red[kl][threadIdx.x & 31] = mx;
    __syncthreads();
    if (threadIdx.x < 32) {
#pragma unroll
        for (int j = 1; j < 8; ++j) mx = fmaxf(mx, red[j][threadIdx.x]);
        colmax[(long long)blockIdx.x * 32 + threadIdx.x] = mx;
    }
}
__global__ void __launch_bounds__(256) prep_dgrad_kernel(const PpyWeightPrep *desc, int count, const float *colmax) {
    __shared__ float tile[32][33], ssw[32];
    const int t = prep_find(desc, count, (int)blockIdx.x, false);
    const PpyWeightPrep d = desc[t];
    if (!d.dgrad_planes) return;
    const int local = (int)blockIdx.x - d.blk0, cgrp = local / PPY_PREP_SPLIT, sp = local - cgrp * PPY_PREP_SPLIT;
    const int c0 = cgrp * 32, tid = threadIdx.x;
    const int Kp = (d.K + 31) / 32 * 32, kchunks = Kp / 32, units = d.R * d.S * kchunks;
    if (tid < 32) {
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < PPY_PREP_SPLIT; ++j) mx = fmaxf(mx, colmax[((long long)(d.blk0 + cgrp * PPY_PREP_SPLIT + j)) * 32 + tid]);
        const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
        int f = 267 - e;
        f = f < 103 ? 103 : (f > 167 ? 167 : f);
        ssw[tid] = __uint_as_float((unsigned)f << 23);
        if (sp == 0) d.dgrad_scale[c0 + tid] = __uint_as_float((unsigned)(254 - f) << 23);
    }
    __syncthreads();
    unsigned short *planes = (unsigned short *)d.dgrad_planes;
    const long long n = (long long)d.C * d.R * d.S * Kp;
    const int lc = tid & 31, kl = tid >> 5;              // read role: column (channel) lc, rows kl + 8 j
    const int wc = tid >> 3, kq = (tid & 7) * 4;         // write role: channel wc, four consecutive k
    for (int u = sp; u < units; u += PPY_PREP_SPLIT) {
        const int tap = u / kchunks, kc = u - tap * kchunks;
        const int r2 = tap / d.S, s2 = tap - r2 * d.S;
        const int src_tap = (d.R - 1 - r2) * d.S + (d.S - 1 - s2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kc * 32 + kl + 8 * j;
            tile[kl + 8 * j][lc] = k < d.K ? d.w[((long long)k * d.R * d.S + src_tap) * d.C + c0 + lc] : 0.f;
        }
        __syncthreads();
        const float sw = ssw[wc];
        float v[4];
        unsigned short h0[4], h1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = tile[kq + j][wc];
            const float x = v[j] * sw;
            const _Float16 a = (_Float16)x;
            const _Float16 b = (_Float16)(x - (float)a);
            h0[j] = __builtin_bit_cast(unsigned short, a);
            h1[j] = __builtin_bit_cast(unsigned short, b);
        }
        // row c of w' has kred' = R*S*Kp values, index i = tap * Kp + kc * 32 + kk: chunk i >> 5 = u, column kk
        const long long o = PPY_X3_BBLOCK ? ((long long)u * d.C + c0 + wc) * 32 + kq : (long long)(c0 + wc) * d.R * d.S * Kp + (long long)u * 32 + kq;
        typedef __attribute__((ext_vector_type(4))) unsigned short ushortx4;
        *reinterpret_cast<ushortx4 *>(planes + o) = ushortx4{h0[0], h0[1], h0[2], h0[3]};
        *reinterpret_cast<ushortx4 *>(planes + n + o) = ushortx4{h1[0], h1[1], h1[2], h1[3]};
        *reinterpret_cast<floatx4 *>(d.dgrad_wt + ((long long)(c0 + wc) * d.R * d.S + tap) * Kp + kc * 32 + kq) = floatx4{v[0], v[1], v[2], v[3]};
        __syncthreads();
    }
}

// dy [P][ld] -> padded [P][Kp] (zero channels beyond K): only when K % 32 != 0 (the 258-channel output convolutions)
__global__ void __launch_bounds__(256) pad_channels_kernel(const float *src, int ld, float *dst, int K, int Kp, long long P) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P * Kp) return;
    const int k = (int)(i % Kp);
    const long long p = i / Kp;
    dst[i] = k < K ? src[p * ld + k] : 0.0f;
}

struct WgradArgs {
    const float *x, *dy;
    float *out;              // dw (one slice) or the partial sums [slices][K][R][S][C]
    int x_ld, dy_ld;
    int N, H, W, C, Ho, Wo, K, R, S, stride, pad;
    int P, pix_per_slice, tiles_c, tiles_kc;
    const float *amax_x, *amax_dy;      // f16x2 kernel: tracked per-image maxima of x and dy (N * AMAX_SLOTS slots each)
};

// Workgroup -> (tile, tap, pixel slice).  The (k tile, c tile, tap) workgroups of ONE pixel slice read the same rows of dy and
// x, and the hardware deals consecutive workgroup ids round-robin over the 8 XCDs (8 private L2s): PMC (round 3) shows the
// weight-gradient launches pulling 4 GB per step through the fabric for 0.6 GB of operands.  PPY_WGRAD_XCD=1 (build time) gives
// XCD x a contiguous range of slice-major virtual ids (conv_x3.hip's order), so that a slice's workgroups share one L2 --
// measured and NOT kept: 99.3 instead of 93.3 us per launch on the R50 head (the fabric reads are served by the Infinity Cache
// and were not what the loop waits for; eight slices in flight spread the CUs' requests better than one).  Default: plain order.
#ifndef PPY_WGRAD_XCD
#define PPY_WGRAD_XCD 0
#endif
__device__ __forceinline__ void wgrad_block(int tiles, int taps, int &tile, int &tap, int &slice) {
    int v = (int)blockIdx.x;
    if (PPY_WGRAD_XCD) {
        const int nb = (int)gridDim.x, qd = nb >> 3, r = nb & 7;
        const int xcd = v & 7, idx = v >> 3;
        v = xcd * qd + min(xcd, r) + idx;
    }
    const int tt = tiles * taps;
    slice = v / tt;
    const int w = v - slice * tt;
    tap = w / tiles;
    tile = w - tap * tiles;
}

constexpr int WG_TK = 128, WG_TC = 128;       // workgroup tile: output channels x input channels
constexpr int PIX = 16;                       // pixels per pipeline step (8 MFMA depths of 2)

__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgradArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wk = wave >> 1, wc = wave & 1;                   // 2 x 2 waves, 64 x 64 each
    int tile, tap, slice;
    wgrad_block(p.tiles_kc, p.R * p.S, tile, tap, slice);
    const int tk = tile / p.tiles_c, tc = tile - tk * p.tiles_c;
    const int r = tap / p.S, s = tap - r * p.S;
    const int k_base = tk * WG_TK + wk * 64, c_base = tc * WG_TC + wc * 64;
    const int half = lane >> 5, l32 = lane & 31;
    const int p_begin = slice * p.pix_per_slice;
    const int p_end = min(p_begin + p.pix_per_slice, p.P);

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const bool kok[2] = {k_base + l32 < p.K, k_base + 32 + l32 < p.K};
    const bool cok[2] = {c_base + l32 < p.C, c_base + 32 + l32 < p.C};
    const int hw = p.Ho * p.Wo;
    // this lane walks the pixels p_begin + half, +2, +4, ...: (n, ho, wo) kept incrementally
    int pix = p_begin + half;
    int n = pix / hw, rem = pix - n * hw;
    int ho = rem / p.Wo, wo = rem - ho * p.Wo;

    float a[2][PIX / 2], b[2][PIX / 2];
    auto fetch = [&](float (&fa)[2][PIX / 2], float (&fb)[2][PIX / 2]) {
#pragma unroll
        for (int q = 0; q < PIX / 2; ++q) {
            const bool pok = pix < p_end;
            const int hi = ho * p.stride + r - p.pad, wi = wo * p.stride + s - p.pad;
            const bool inside = pok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const float *dyp = p.dy + (long long)pix * p.dy_ld + k_base + l32;
            const float *xp = p.x + (((long long)n * p.H + hi) * p.W + wi) * p.x_ld + c_base + l32;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i][q] = (pok && kok[i]) ? dyp[32 * i] : 0.0f;
                fb[i][q] = (inside && cok[i]) ? xp[32 * i] : 0.0f;
            }
            pix += 2;
            wo += 2;
            while (wo >= p.Wo) {
                wo -= p.Wo;
                if (++ho == p.Ho) {
                    ho = 0;
                    ++n;
                }
            }
        }
    };
    fetch(a, b);
    for (int p0 = p_begin; p0 < p_end; p0 += PIX) {
        float na[2][PIX / 2], nb[2][PIX / 2];
        fetch(na, nb);                                      // next step's operands in flight under this step's MFMAs
#pragma unroll
        for (int q = 0; q < PIX / 2; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < PIX / 2; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i][q] = na[i][q];
                b[i][q] = nb[i][q];
            }
    }
    // accumulator element e of tile (i, j): row k = (e&3) + 8*(e>>2) + 4*half, column c = l32
    float *out = p.out + (long long)slice * p.K * p.R * p.S * p.C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c_base + 32 * j + l32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k_base + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (k < p.K && c < p.C) out[(((long long)k * p.R + r) * p.S + s) * p.C + c] = acc[i][j][e];
            }
        }
#endif
}

// ---- wgrad on the 16-bit MFMA: exact 3-term bf16 split of both operands, the 6 leading products (as conv_x3.hip's bf16x3).
// v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE reduction indices per lane, and the reduction index here is the pixel, which
// is the strided one in NHWC -- so a step of 32 pixels goes through LDS transposed: every thread loads 4 pixels x 4 channels
// of dy and of x (16-byte loads, the padding taps of x as zeros), splits the 32 values once (not once per consuming wave) and
// stores [plane][channel][pixel] bf16 with 8-byte LDS writes; a fragment is then one 16-byte LDS read per plane
// (row pitch 80 B: conflict-free for the 16-lane groups of ds_read_b128).  One LDS buffer, two barriers per step; two
// workgroups per CU overlap each other's store / MFMA phases.  Tile 128 (k) x 128 (c) of one tap over a pixel slice.
// (Measured and not kept: global loads TWO steps ahead in a second register set -- 102 instead of 96 us per launch on the R50 head,
// the loop is bound by the split + LDS store + barrier phase, not by the latency of the loads.)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned uintx4_t;
typedef __attribute__((ext_vector_type(2))) unsigned uintx2_t;
constexpr int X3_PIX = 32;                 // pixels per step
constexpr int X3_PITCH = 80;               // bytes per channel row: 32 pixels x 2 B + 16 B

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {      // RNE, a -> low half
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float floatx2_t;
    const floatx2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// 4 pixels of one channel -> its three bf16 terms, each as 4 x bf16 = 8 bytes
__device__ __forceinline__ void split4(const float (&v)[4], uintx2_t (&out)[3]) {
    float r[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const unsigned lo = pack_bf16(r[0], r[1]), hi = pack_bf16(r[2], r[3]);
        out[t] = uintx2_t{lo, hi};
        r[0] -= __uint_as_float(lo << 16);
        r[1] -= __uint_as_float(lo & 0xffff0000u);
        r[2] -= __uint_as_float(hi << 16);
        r[3] -= __uint_as_float(hi & 0xffff0000u);
    }
}

// 4 pixels of one channel, scaled by a power of two into the fp16 range -> two fp16 terms (the residual of the SCALED value is
// exact: conv_x3.hip), each as 4 x fp16 = 8 bytes
__device__ __forceinline__ void split4_f16(const float (&v)[4], float sc, uintx2_t (&out)[2]) {
    const unsigned lo = cvt_pk_f16(v[0] * sc, v[1] * sc), hi = cvt_pk_f16(v[2] * sc, v[3] * sc);
    out[0] = uintx2_t{lo, hi};
    out[1] = uintx2_t{cvt_pk_f16(fmaf(v[0], sc, -f16_lo(lo)), fmaf(v[1], sc, -f16_hi(lo))),
                      cvt_pk_f16(fmaf(v[2], sc, -f16_lo(hi)), fmaf(v[3], sc, -f16_hi(hi)))};
}
// the power of two that puts max|tensor| (the maximum over its per-image tracked maxima) into [2^13, 2^14), and its inverse
__device__ __forceinline__ float tensor_scale(const float *amax, int slots, int lane, float *inv) {
    float mx = 0.0f;
    for (int i = lane; i < slots; i += 64) mx = fmaxf(mx, fabsf(amax[(long long)i * AMAX_STRIDE]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    int f = 267 - e;
    f = f < 103 ? 103 : (f > 167 ? 167 : f);
    *inv = __uint_as_float((unsigned)(254 - f) << 23);
    return __uint_as_float((unsigned)f << 23);
}

// F16 = false: bf16x3 (3 planes, 6 products).  F16 = true: f16x2 (2 planes, 3 products on v_mfma_f32_32x32x16_f16); the sum
// runs over the pixels of ALL images, so each operand gets ONE scale, from the maximum over its per-image maxima.
template <bool F16>
__global__ void __launch_bounds__(256, 2) conv_wgrad_x3_kernel(const WgradArgs p) {      // (three f16x2 workgroups per CU -- 40 KB of LDS each -- measured no faster: 17.55 vs 17.31 ms per step)
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NP = F16 ? 2 : 3;
    constexpr int X3_TILE = NP * 128 * X3_PITCH;                // one operand: NP planes x 128 channels
    __shared__ __attribute__((aligned(16))) char smem[2 * X3_TILE];
    char *sA = smem, *sB = smem + X3_TILE;                      // dy (k rows), x (c rows)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wc = wave & 1;
    int tile, tap, slice;
    wgrad_block(p.tiles_kc, p.R * p.S, tile, tap, slice);
    const int tk = tile / p.tiles_c, tc = tile - tk * p.tiles_c;
    const int r = tap / p.S, s = tap - r * p.S;
    const int k0 = tk * WG_TK, c0 = tc * WG_TC;
    const int p_begin = slice * p.pix_per_slice;
    const int p_end = min(p_begin + p.pix_per_slice, p.P);
    // loader role: pixel quad pq (4 pixels), channel group cg (4 channels)
    const int pq = tid & 7, cg = tid >> 3;
    const int hw = p.Ho * p.Wo;
    const bool a_ok = k0 + cg * 4 < p.K, b_ok = c0 + cg * 4 < p.C;          // (whole float4 groups: K, C are multiples of 4 or the
                                                                            //  buffers are zero beyond them up to a multiple of 4)
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    floatx4 ra[4], rb[4];
    auto fetch = [&](int pbase) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pix = pbase + pq * 4 + q;
            ra[q] = rb[q] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (pix < p_end) {
                if (a_ok) ra[q] = *reinterpret_cast<const floatx4 *>(p.dy + (long long)pix * p.dy_ld + k0 + cg * 4);
                const int n = pix / hw, rem = pix - n * hw;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                const int hi = ho * p.stride + r - p.pad, wi = wo * p.stride + s - p.pad;
                if (b_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    rb[q] = *reinterpret_cast<const floatx4 *>(p.x + (((long long)n * p.H + hi) * p.W + wi) * p.x_ld + c0 + cg * 4);
            }
        }
    };
    float s_dy = 1.0f, s_x = 1.0f, inv_dy = 1.0f, inv_x = 1.0f;
    if constexpr (F16) {
        s_dy = tensor_scale(p.amax_dy, p.N * AMAX_SLOTS, tid & 63, &inv_dy);
        s_x = tensor_scale(p.amax_x, p.N * AMAX_SLOTS, tid & 63, &inv_x);
    }
    auto stage = [&]() {        // registers -> NP 16-bit planes, transposed: [plane][channel][pixel]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float va[4] = {ra[0][e], ra[1][e], ra[2][e], ra[3][e]}, vb[4] = {rb[0][e], rb[1][e], rb[2][e], rb[3][e]};
            uintx2_t ta[NP], tb[NP];
            if constexpr (F16) {
                split4_f16(va, s_dy, ta);
                split4_f16(vb, s_x, tb);
            } else {
                split4(va, ta);
                split4(vb, tb);
            }
            const int row = cg * 4 + e;
#pragma unroll
            for (int t = 0; t < NP; ++t) {
                *reinterpret_cast<uintx2_t *>(sA + (t * 128 + row) * X3_PITCH + pq * 8) = ta[t];
                *reinterpret_cast<uintx2_t *>(sB + (t * 128 + row) * X3_PITCH + pq * 8) = tb[t];
            }
        }
    };
    const int frow = lane & 31, fh = lane >> 5;
    fetch(p_begin);
    for (int p0 = p_begin; p0 < p_end; p0 += X3_PIX) {
        __syncthreads();                        // everybody has finished reading the previous step
        stage();
        __syncthreads();
        fetch(p0 + X3_PIX);                     // next step's global loads fly under this step's MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uintx4_t fa[2][NP], fb[2][NP];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < NP; ++t) {
                    fa[i][t] = *reinterpret_cast<const uintx4_t *>(sA + (t * 128 + wk * 64 + i * 32 + frow) * X3_PITCH + ks * 32 + fh * 16);
                    fb[i][t] = *reinterpret_cast<const uintx4_t *>(sB + (t * 128 + wc * 64 + i * 32 + frow) * X3_PITCH + ks * 32 + fh * 16);
                }
            // the six (bf16x3) / three (f16x2: pieces {0, 1}) leading products, smallest first
            constexpr int ta_[6] = {2, 1, 0, 1, 0, 0}, tb_[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t = F16 ? 3 : 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (F16)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][ta_[t]]),
                                                                              __builtin_bit_cast(f16x8, fb[j][tb_[t]]), acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[i][ta_[t]]),
                                                                               __builtin_bit_cast(bf16x8_t, fb[j][tb_[t]]), acc[i][j], 0, 0, 0);
                    }
        }
    }
    float *out = p.out + (long long)slice * p.K * p.R * p.S * p.C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wc * 64 + 32 * j + frow;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wk * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (k < p.K && c < p.C) out[(((long long)k * p.R + r) * p.S + s) * p.C + c] = F16 ? acc[i][j][e] * (inv_dy * inv_x) : acc[i][j][e];
            }
        }
#endif
}

// ---- 3x3 / stride 1 / pad 1 weight gradient with ALL NINE TAPS in one workgroup (round 3; f16x2 only).
// What bounds the kernel above is bytes through the CU's vector memory path: a step delivers 32 KB (128 channels x 32 pixels of
// dy and of x) for 24 MFMAs per wave -- ~910 cycles of delivery at the ~36 B/clk a CU sustains against 768 cycles of MFMA, per
// workgroup, two workgroups per CU -- and the nine taps of a 3x3 layer are nine workgroups that each fetch and split the same
// dy tile and overlapping windows of x.  Here a workgroup owns 128 output channels x 32 input channels x 9 taps:
//   * dy tile split ONCE for nine taps; x: thread (row shift r, pixel quad, channel group) loads the SIX pixels wo0-1 .. wo0+4
//     of row ho+r-1, splits them once, and builds the three column-shifted quads of the taps (r, 0..2) in registers (the middle
//     one with two v_alignbit per plane) -- 34 KB per step for 54 MFMAs per wave: 2.4x fewer bytes per MFMA, 1.4x fewer
//     operand splits;
//   * so that a quad never straddles an image row the reduction runs over VIRTUAL pixels of rows padded to a multiple of 4
//     (19 -> 20: the padding pixels carry dy = 0);
//   * same LDS operand layout ([plane][channel][pixel], 80-byte pitch), same three products in the same order, 144 accumulator
//     registers per lane, 66.5 KB of LDS: two workgroups per CU.
constexpr int W9_TK = 128, W9_TC = 32;
constexpr int W9_SA = 2 * W9_TK * X3_PITCH, W9_SBT = 2 * W9_TC * X3_PITCH;      // dy planes; x planes of ONE tap
struct Wgrad9Args {
    const float *x, *dy;
    float *out;
    int x_ld, dy_ld;
    int N, H, W, Wp, C, K;
    int Pv, pix_per_slice, tiles_c;
    const float *amax_x, *amax_dy;
};

__global__ void __launch_bounds__(256, 2) conv_wgrad9_kernel(const Wgrad9Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem9[];
    char *sA = smem9, *sB = smem9 + W9_SA;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, slice = blockIdx.y;
    const int tk = tile / p.tiles_c, tc = tile - tk * p.tiles_c;
    const int k0 = tk * W9_TK, c0 = tc * W9_TC;
    const int p_begin = slice * p.pix_per_slice, p_end = min(p_begin + p.pix_per_slice, p.Pv);
    const int hwp = p.H * p.Wp;
    // loader roles.  dy: pixel quad pq, channel group cg (4 of the 128 channels).  x (threads 0 .. 191): row shift xr, pixel quad
    // xpq, channel group xcg (4 of the 32 channels)
    const int pq = tid & 7, cg = tid >> 3;
    const int xr = tid >> 6, xpq = lane & 7, xcg = lane >> 3;
    const bool a_ok = k0 + cg * 4 < p.K, x_thread = tid < 192;

    floatx16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    floatx4 ra[4], rb[6];
    auto fetch = [&](int vbase) {
        {   // dy: the four pixels of this thread's quad
            const int v0 = vbase + pq * 4;
            const int n = v0 / hwp, rem = v0 - n * hwp;
            const int ho = rem / p.Wp, wo0 = rem - ho * p.Wp;
            const float *row = p.dy + (((long long)n * p.H + ho) * p.W + wo0) * p.dy_ld + k0 + cg * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ra[q] = floatx4{0.f, 0.f, 0.f, 0.f};
                if (a_ok && v0 < p_end && wo0 + q < p.W) ra[q] = *reinterpret_cast<const floatx4 *>(row + (long long)q * p.dy_ld);
            }
        }
        if (x_thread) {
            const int v0 = vbase + xpq * 4;
            const int n = v0 / hwp, rem = v0 - n * hwp;
            const int ho = rem / p.Wp, wo0 = rem - ho * p.Wp;
            const int yi = ho + xr - 1;
            const bool rok = v0 < p_end && (unsigned)yi < (unsigned)p.H;
            const float *row = p.x + (((long long)n * p.H + yi) * p.W + wo0) * p.x_ld + c0 + xcg * 4;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                rb[j] = floatx4{0.f, 0.f, 0.f, 0.f};
                if (rok && (unsigned)(wo0 + j - 1) < (unsigned)p.W) rb[j] = *reinterpret_cast<const floatx4 *>(row + (long long)(j - 1) * p.x_ld);
            }
        }
    };
    float s_dy, s_x, inv_dy, inv_x;
    s_dy = tensor_scale(p.amax_dy, p.N * AMAX_SLOTS, lane, &inv_dy);
    s_x = tensor_scale(p.amax_x, p.N * AMAX_SLOTS, lane, &inv_x);
    auto stage = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float va[4] = {ra[0][e], ra[1][e], ra[2][e], ra[3][e]};
            uintx2_t ta[2];
            split4_f16(va, s_dy, ta);
            const int row = cg * 4 + e;
            *reinterpret_cast<uintx2_t *>(sA + (0 * W9_TK + row) * X3_PITCH + pq * 8) = ta[0];
            *reinterpret_cast<uintx2_t *>(sA + (1 * W9_TK + row) * X3_PITCH + pq * 8) = ta[1];
        }
        if (x_thread) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned P[3], Q[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float u0 = rb[2 * j][e], u1 = rb[2 * j + 1][e];
                    P[j] = cvt_pk_f16(u0 * s_x, u1 * s_x);
                    Q[j] = cvt_pk_f16(fmaf(u0, s_x, -f16_lo(P[j])), fmaf(u1, s_x, -f16_hi(P[j])));
                }
                const int row = xcg * 4 + e;
                char *b0 = sB + (xr * 3) * W9_SBT + row * X3_PITCH + xpq * 8;
                // tap (xr, 0): pixels wo-1 = loaded 0..3; (xr, 1): loaded 1..4; (xr, 2): loaded 2..5
                *reinterpret_cast<uintx2_t *>(b0) = uintx2_t{P[0], P[1]};
                *reinterpret_cast<uintx2_t *>(b0 + W9_TC * X3_PITCH) = uintx2_t{Q[0], Q[1]};
                *reinterpret_cast<uintx2_t *>(b0 + W9_SBT) = uintx2_t{__builtin_amdgcn_alignbit(P[1], P[0], 16), __builtin_amdgcn_alignbit(P[2], P[1], 16)};
                *reinterpret_cast<uintx2_t *>(b0 + W9_SBT + W9_TC * X3_PITCH) =
                    uintx2_t{__builtin_amdgcn_alignbit(Q[1], Q[0], 16), __builtin_amdgcn_alignbit(Q[2], Q[1], 16)};
                *reinterpret_cast<uintx2_t *>(b0 + 2 * W9_SBT) = uintx2_t{P[1], P[2]};
                *reinterpret_cast<uintx2_t *>(b0 + 2 * W9_SBT + W9_TC * X3_PITCH) = uintx2_t{Q[1], Q[2]};
            }
        }
    };
    const int frow = lane & 31, fh = lane >> 5;
    fetch(p_begin);
    for (int v = p_begin; v < p_end; v += X3_PIX) {
        __syncthreads();                        // everybody has finished reading the previous step
        stage();
        __syncthreads();
        fetch(v + X3_PIX);                      // next step's global loads fly under this step's MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uintx4_t fa[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
                fa[t] = *reinterpret_cast<const uintx4_t *>(sA + (t * W9_TK + wave * 32 + frow) * X3_PITCH + ks * 32 + fh * 16);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                uintx4_t fb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    fb[t] = *reinterpret_cast<const uintx4_t *>(sB + tap * W9_SBT + (t * W9_TC + frow) * X3_PITCH + ks * 32 + fh * 16);
                // the three leading products, smallest first (as conv_wgrad_x3_kernel<true>): lo x hi, hi x lo, hi x hi
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[1]), __builtin_bit_cast(f16x8, fb[0]), acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0]), __builtin_bit_cast(f16x8, fb[1]), acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0]), __builtin_bit_cast(f16x8, fb[0]), acc[tap], 0, 0, 0);
            }
        }
    }
    float *out = p.out + (long long)slice * p.K * 9 * p.C;
    const int c = c0 + frow;
    const float inv = inv_dy * inv_x;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = k0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
            if (k < p.K && c < p.C) out[((long long)k * 9 + tap) * p.C + c] = acc[tap][e] * inv;
        }
#endif
}

// virtual pixels (rows padded to a multiple of 4) and slice count of the nine-tap kernel: ~2 workgroups per CU in one round,
// at least 8 steps per slice, at most 64 slices (the partial sums are written and read once each)
static int wgrad9_slices(int K, int C, int Pv) {
    const int tiles = ceil_div(K, W9_TK) * (C / W9_TC);
    const char *e = getenv("PPY_WGRAD9_WGS");            // (experiments: the workgroup count the slices aim at)
    const int target = e && atoi(e) > 0 ? atoi(e) : 512;
    int sl = ceil_div(target, tiles);
    const int maxsl = Pv / 256 > 0 ? Pv / 256 : 1;
    if (sl > maxsl) sl = maxsl;
    if (sl > 64) sl = 64;
    return sl < 1 ? 1 : sl;
}
static bool wgrad9_applies(int C, int R, int S, int stride, int pad) {
    const char *sw = getenv("PPY_WGRAD9");      // A/B switch, read per call (tests compare the two kernels)
    const bool off = sw && sw[0] == '0';
    return !off && R == 3 && S == 3 && stride == 1 && pad == 1 && C % W9_TC == 0;
}

// dw[i] = sum over the slices in index order (a fixed summation order: run-to-run identical results)
__global__ void __launch_bounds__(256) wgrad_combine_kernel(const float *part, float *dw, long long n, int slices) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if ((n & 3) == 0) {       // (n % 4 == 0 whenever C % 4 == 0: slices stay 16-byte aligned)
        // (four slices requested at a time, added in index order: the same sums with a quarter of the round trips)
        floatx4 v = *reinterpret_cast<const floatx4 *>(part + i);
        int sl = 1;
        for (; sl + 3 < slices; sl += 4) {
            floatx4 u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = *reinterpret_cast<const floatx4 *>(part + (long long)(sl + k) * n + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[0] += u[k][0]; v[1] += u[k][1]; v[2] += u[k][2]; v[3] += u[k][3]; }
        }
        for (; sl < slices; ++sl) {
            const floatx4 u = *reinterpret_cast<const floatx4 *>(part + (long long)sl * n + i);
            v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
        }
        *reinterpret_cast<floatx4 *>(dw + i) = v;
    } else {
        for (long long t = i; t < min(i + 4, n); ++t) {
            float v = part[t];
            for (int sl = 1; sl < slices; ++sl) v += part[sl * n + t];
            dw[t] = v;
        }
    }
}

static int wgrad_slices(int K, int C, int R, int S, int P) {
    // enough workgroups for ~3 rounds over the 256 CUs, at least 256 pixels per slice
    const int tiles = ceil_div(K, WG_TK) * ceil_div(C, WG_TC) * R * S;
    int sl = ceil_div(768, tiles);
    const int maxsl = P / 256 > 0 ? P / 256 : 1;
    if (sl > maxsl) sl = maxsl;
    if (sl > 64) sl = 64;
    return sl < 1 ? 1 : sl;
}

static inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

static bool bwd_geometry(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, Geometry *g) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0) return false;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    if (Ho <= 0 || Wo <= 0 || (long long)N * Ho * Wo > 0x7fffffffLL / 4) return false;
    g->Ho = Ho; g->Wo = Wo; g->M = N * Ho * Wo; g->Kred = R * S * C; g->chunks = 0;
    return true;
}

}  // namespace

extern "C" size_t ppy_conv2d_dgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int cfg,
                                        int splitk) {
    Geometry g;          // (of the layer itself: C need not be a multiple of 32 here -- CoordConv layers have C = 514 -- K is padded)
    if (stride != 1 || !bwd_geometry(N, H, W, C, K, R, S, stride, pad, &g)) return 0;
    const int Kp = (K + 31) / 32 * 32;
    size_t b = align256((size_t)C * R * S * Kp * 4) + align256((size_t)C * 4) * 2;      // w', ones, zeros
    b += align256((size_t)C * R * S * Kp * 6);                                          // three bf16 planes of w' (or two fp16 planes)
    b += align256((size_t)C * 4);                                                       // f16x2: epilogue scale with the weight scale folded in
    if (Kp != K) b += align256((size_t)g.M * Kp * 4);                                   // channel-padded dy
    return b + align256(ppy_conv2d_workspace_bytes(N, g.Ho, g.Wo, Kp, C, R, S, 1, R - 1 - pad, cfg, splitk));
}

extern "C" int ppy_conv2d_dgrad_f32(const float *dy, int dy_ld, const float *w_krsc, float *dx, int dx_ld, int N, int H,
                                    int W, int C, int K, int R, int S, int stride, int pad, int cfg, int splitk,
                                    const float *amax_dy, void *ws, size_t ws_bytes, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(dy && w_krsc && dx && N > 0 && C > 0 && K > 0 && dy_ld >= K && dx_ld >= C);
    if (stride != 1) return PPY_ERR_UNSUPPORTED;           // (the trainable head has no strided convolution)
    Geometry g;
    if (!bwd_geometry(N, H, W, C, K, R, S, stride, pad, &g)) return PPY_ERR_BAD_ARG;
    const int Kp = (K + 31) / 32 * 32;
    PPY_CHECK_ARG(R - 1 - pad >= 0 && (Kp != K || (dy_ld % 4 == 0 && ((uintptr_t)dy & 15) == 0)));      // (a padded copy is aligned by construction)
    const size_t need = ppy_conv2d_dgrad_workspace_bytes(N, H, W, C, K, R, S, stride, pad, cfg, splitk);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255) != 0) return PPY_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char *base = (char *)ws;
    float *wt = (float *)base;
    base += align256((size_t)C * R * S * Kp * 4);
    float *ones = (float *)base;
    base += align256((size_t)C * 4);
    float *zeros = (float *)base;
    base += align256((size_t)C * 4);
    void *planes = base;
    base += align256((size_t)C * R * S * Kp * 6);
    float *scale_f16 = (float *)base;
    base += align256((size_t)C * 4);
    const long long total = (long long)C * R * S * Kp;
    int rc;
    // amax_dy (tracked per-image maxima of dy, as ppy_bn_train_bwd_f32 records them): the f16x2 kernels -- w' as two fp16 planes
    // scaled per output channel (= input channel of the layer), dy scaled per image; else the exact bf16x3 split
    if (amax_dy && R * S * Kp <= 256 * DW_MAXV) {
        hipLaunchKernelGGL(dgrad_weights_f16_kernel, dim3(C), dim3(256), 0, st, w_krsc, wt, ones, zeros, (unsigned short *)planes,
                           scale_f16, K, Kp, R, S, C);
        rc = ppy_launch_status();
        if (rc != PPY_OK) return rc;
    } else {
        hipLaunchKernelGGL(dgrad_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_krsc, wt, ones, zeros,
                           K, Kp, R, S, C);
        rc = ppy_launch_status();
        if (rc != PPY_OK) return rc;
        rc = amax_dy ? ppy_conv2d_split_weights_f16x2(wt, C, (long long)R * S * Kp, ones, planes, scale_f16, stream)
                     : ppy_conv2d_split_weights_bf16x3(wt, total, planes, stream);
        if (rc != PPY_OK) return rc;
    }
    const float *src = dy;
    int src_ld = dy_ld;
    if (Kp != K) {
        float *padded = (float *)base;
        base += align256((size_t)g.M * Kp * 4);
        const long long n = (long long)g.M * Kp;
        hipLaunchKernelGGL(pad_channels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, dy_ld, padded, K, Kp,
                           (long long)g.M);
        rc = ppy_launch_status();
        if (rc != PPY_OK) return rc;
        src = padded;
        src_ld = Kp;
    }
    const size_t rest = ws_bytes - (size_t)(base - (char *)ws);
    // PPY_DGRAD_FP32=1 (read once per process; bench.py's `value_fp32_exact` leg of the training step): the same contraction on the
    // exact-fp32 MFMA kernels -- no operand planes, the library's own tile choice
    static const bool force_fp32 = getenv("PPY_DGRAD_FP32") && getenv("PPY_DGRAD_FP32")[0] == '1';
    if (force_fp32)
        return ppy_conv2d_bn_act_f32(src, src_ld, wt, nullptr, nullptr, ones, nullptr, zeros, nullptr, 0, nullptr, nullptr, dx, dx_ld, N, g.Ho, g.Wo,
                                     Kp, C, R, S, 1, R - 1 - pad, PPY_ACT_NONE, 0, -1, 0, nullptr, nullptr, base, rest, stream);
    // dy is [N, Ho, Wo, K]; for stride 1 the forward convolution with pad' = R-1-pad maps it back onto [N, H, W, C]
    return ppy_conv2d_bn_act_f32(src, src_ld, wt, amax_dy ? nullptr : planes, amax_dy ? planes : nullptr, ones, amax_dy ? scale_f16 : nullptr,
                                 zeros, nullptr, 0, nullptr, nullptr, dx, dx_ld, N, g.Ho, g.Wo, Kp, C, R, S, 1, R - 1 - pad, PPY_ACT_NONE, 0,
                                 cfg, splitk, amax_dy, nullptr, base, rest, stream);
}

extern "C" int ppy_train_prepare_weights_f16x2(const PpyWeightPrep *desc_dev, int count, int rows_total, int blocks_total, float *colmax,
                                               size_t colmax_bytes, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(desc_dev && count > 0 && rows_total > 0 && blocks_total >= 0);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(prep_fwd_kernel, dim3(rows_total), dim3(256), 0, st, desc_dev, count);
    if (blocks_total > 0) {
        if (!colmax || colmax_bytes < (size_t)blocks_total * 32 * sizeof(float)) return PPY_ERR_WORKSPACE;
        hipLaunchKernelGGL(prep_colmax_kernel, dim3(blocks_total), dim3(256), 0, st, desc_dev, count, colmax);
        hipLaunchKernelGGL(prep_dgrad_kernel, dim3(blocks_total), dim3(256), 0, st, desc_dev, count, (const float *)colmax);
    }
    return ppy_launch_status();
}

extern "C" int ppy_conv2d_dgrad_prepared_f32(const float *dy, int dy_ld, const float *wt, const void *planes, const float *scale_f16x2,
                                             const float *ones, const float *zeros, float *dx, int dx_ld, int N, int H, int W, int C, int K,
                                             int R, int S, int pad, int cfg, int splitk, const float *amax_dy, void *ws, size_t ws_bytes,
                                             void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(dy && wt && planes && scale_f16x2 && ones && zeros && dx && amax_dy && N > 0 && C > 0 && K > 0 && dy_ld >= K && dx_ld >= C);
    Geometry g;
    if (!bwd_geometry(N, H, W, C, K, R, S, 1, pad, &g)) return PPY_ERR_BAD_ARG;
    const int Kp = (K + 31) / 32 * 32;
    PPY_CHECK_ARG(R - 1 - pad >= 0 && (Kp != K || (dy_ld % 4 == 0 && ((uintptr_t)dy & 15) == 0)));
    hipStream_t st = (hipStream_t)stream;
    char *base = (char *)ws;
    const float *src = dy;
    int src_ld = dy_ld;
    if (Kp != K) {
        const size_t pad_bytes = align256((size_t)g.M * Kp * 4);
        if (!ws || ws_bytes < pad_bytes || ((uintptr_t)ws & 255) != 0) return PPY_ERR_WORKSPACE;
        float *padded = (float *)base;
        base += pad_bytes;
        const long long n = (long long)g.M * Kp;
        hipLaunchKernelGGL(pad_channels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, dy_ld, padded, K, Kp, (long long)g.M);
        const int rc = ppy_launch_status();
        if (rc != PPY_OK) return rc;
        src = padded;
        src_ld = Kp;
    }
    const size_t rest = ws ? ws_bytes - (size_t)(base - (char *)ws) : 0;
    return ppy_conv2d_bn_act_f32(src, src_ld, wt, nullptr, planes, ones, scale_f16x2, zeros, nullptr, 0, nullptr, nullptr, dx, dx_ld, N, g.Ho,
                                 g.Wo, Kp, C, R, S, 1, R - 1 - pad, PPY_ACT_NONE, 0, cfg, splitk, amax_dy, nullptr, ws ? base : nullptr, rest, stream);
}

extern "C" size_t ppy_conv2d_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0) return 0;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    int sl = wgrad_slices(K, C, R, S, N * Ho * Wo);
    if (wgrad9_applies(C, R, S, stride, pad)) {
        const int s9 = wgrad9_slices(K, C, N * H * ((W + 3) / 4 * 4));
        if (s9 > sl) sl = s9;
    }
    return sl > 1 ? (size_t)sl * K * R * S * C * 4 : 0;
}

extern "C" int ppy_conv2d_wgrad_f32(const float *x, int x_ld, const float *dy, int dy_ld, float *dw_krsc, int N, int H, int W,
                                    int C, int K, int R, int S, int stride, int pad, const float *amax_x, const float *amax_dy,
                                    void *ws, size_t ws_bytes, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && dy && dw_krsc && N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0);
    PPY_CHECK_ARG(x_ld >= C && dy_ld >= K);
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    PPY_CHECK_ARG(Ho > 0 && Wo > 0 && (long long)N * Ho * Wo < (1LL << 31));
    WgradArgs p;
    p.x = x; p.dy = dy; p.x_ld = x_ld; p.dy_ld = dy_ld;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.K = K; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.P = N * Ho * Wo;
    p.amax_x = amax_x; p.amax_dy = amax_dy;
    const int sl = wgrad_slices(K, C, R, S, p.P);
    const size_t need = ppy_conv2d_wgrad_workspace_bytes(N, H, W, C, K, R, S, stride, pad);
    if (need && (!ws || ws_bytes < need || ((uintptr_t)ws & 15) != 0)) return PPY_ERR_WORKSPACE;
    // the bf16x3 kernel reads whole float4 channel groups: pixel strides and base pointers 16-byte aligned, and the last group
    // of a tensor whose channel count is not a multiple of 4 must be readable up to the multiple (what lies there only
    // reaches output rows / columns that are not stored) -- true for this library's own buffers (pixel stride = channels
    // rounded up to 32); PPY_WGRAD_FP32=1 forces the exact-fp32 kernel
    static const bool force_fp32 = getenv("PPY_WGRAD_FP32") && getenv("PPY_WGRAD_FP32")[0] == '1';
    const bool x3 = !force_fp32 && x_ld % 4 == 0 && dy_ld % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 &&
                    (C % 4 == 0 || x_ld >= (C + 3) / 4 * 4) && (K % 4 == 0 || dy_ld >= (K + 3) / 4 * 4);
    const int gran = x3 ? X3_PIX : PIX;
    p.pix_per_slice = ceil_div(ceil_div(p.P, sl), gran) * gran;
    const int slices = ceil_div(p.P, p.pix_per_slice);
    p.tiles_c = ceil_div(C, WG_TC);
    p.tiles_kc = ceil_div(K, WG_TK) * p.tiles_c;
    p.out = slices > 1 ? (float *)ws : dw_krsc;
    hipStream_t st = (hipStream_t)stream;
    PPY_CHECK_ARG((long long)p.tiles_kc * R * S * slices < (1LL << 31));
    const dim3 grid(p.tiles_kc * R * S * slices);
    if (x3 && amax_x && amax_dy && wgrad9_applies(C, R, S, stride, pad)) {
        Wgrad9Args q;
        q.x = x; q.dy = dy; q.x_ld = x_ld; q.dy_ld = dy_ld; q.N = N; q.H = H; q.W = W; q.Wp = (W + 3) / 4 * 4; q.C = C; q.K = K;
        q.Pv = N * H * q.Wp;
        q.amax_x = amax_x; q.amax_dy = amax_dy;
        const int s9 = wgrad9_slices(K, C, q.Pv);
        q.pix_per_slice = ceil_div(ceil_div(q.Pv, s9), X3_PIX) * X3_PIX;
        const int slices9 = ceil_div(q.Pv, q.pix_per_slice);
        q.tiles_c = C / W9_TC;
        q.out = slices9 > 1 ? (float *)ws : dw_krsc;
        static PpyLdsAttr attr9;
        const int lds9 = W9_SA + 9 * W9_SBT;
        if (ppy_lds_attr(attr9, (const void *)conv_wgrad9_kernel, lds9) != PPY_OK) return PPY_ERR_LAUNCH;
        hipLaunchKernelGGL(conv_wgrad9_kernel, dim3(ceil_div(K, W9_TK) * q.tiles_c, slices9), dim3(256), lds9, st, q);
        int rc9 = ppy_launch_status();
        if (rc9 != PPY_OK) return rc9;
        if (slices9 > 1) {
            const long long n = (long long)K * 9 * C;
            hipLaunchKernelGGL(wgrad_combine_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, st, (const float *)ws, dw_krsc, n, slices9);
            rc9 = ppy_launch_status();
        }
        return rc9;
    } else if (x3 && amax_x && amax_dy) {
        hipLaunchKernelGGL(conv_wgrad_x3_kernel<true>, grid, dim3(256), 0, st, p);
    } else if (x3) {
        hipLaunchKernelGGL(conv_wgrad_x3_kernel<false>, grid, dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 0, st, p);
    }
    int rc = ppy_launch_status();
    if (rc != PPY_OK) return rc;
    if (slices > 1) {
        const long long n = (long long)K * R * S * C;
        hipLaunchKernelGGL(wgrad_combine_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, st, (const float *)ws, dw_krsc,
                           n, slices);
        rc = ppy_launch_status();
    }
    return rc;
}
