// Stem conv (NCHW -> NHWC) and the pooling kernels of the PP-YOLO backbone / SPP.
// All HBM/L2-bound: 16-byte coalesced NHWC accesses, no MFMA.
#include <math.h>
#include <cstdlib>

#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------
// stage1_conv1_1 (reference model/resnet_vd.py:100): 3x3 s2 p1 conv, 3 -> K channels, reading
// the caller's NCHW tensor.  One thread = one output pixel x KT channels; the 27 input taps are
// held in registers, the 27*KT weights come through the scalar cache (uniform addresses).
// K-sum order: (c, r, s) ascending, fp32 fma chain.
template <int KT>
__global__ void __launch_bounds__(256) stem_conv_kernel(const float *__restrict__ x,
                                                        const float *__restrict__ w,
                                                        const float *__restrict__ scale,
                                                        const float *__restrict__ shift, float *y,
                                                        int y_ld, int N, int H, int W, int Ho, int Wo,
                                                        int K, int act, float *amax_out) {
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * Ho * Wo;
    float amx = 0.0f;
    if (pix < total) {
    const int k0 = blockIdx.y * KT;
    const int wo = (int)(pix % Wo);
    const int ho = (int)((pix / Wo) % Ho);
    const int n = (int)(pix / ((long long)Wo * Ho));
    float in[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int hi = ho * 2 - 1 + r, wi = wo * 2 - 1 + s;
                float v = 0.f;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                    v = x[(((long long)n * 3 + c) * H + hi) * W + wi];
                in[c * 9 + r * 3 + s] = v;
            }
    float *o = y + pix * y_ld + k0;
#pragma unroll
    for (int kk = 0; kk < KT; kk += 4) {
        floatx4 r4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + kk + u;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 27; ++t) acc = fmaf(in[t], w[k * 27 + t], acc);
            r4[u] = ppy_apply_act(fmaf(acc, scale[k], shift[k]), act);
            amx = fmaxf(amx, fabsf(r4[u]));
        }
        *reinterpret_cast<floatx4 *>(o + kk) = r4;
    }
    }
    if (amax_out) {      // tracked per-image max|y| (f16x2 consumers)
        const long long pc = pix < total ? pix : total - 1;
        amax_track(amx, (int)(pc / ((long long)Wo * Ho)), amax_out, blockIdx.x * 4 + (threadIdx.x >> 6));
    }
}

// Round 3: the same operator organised around an output ROW SEGMENT (64 pixels x 32 channels per workgroup).  The thread-per-
// pixel form above reads its 27 taps with stride-2 4-byte loads (half of every line unused, twice: K / 16 = 2 grid rows) and
// stores 16-byte pieces 128 B apart -- 64 lines per wave instruction: 67-71 us for 130 MB = 1.9 TB/s.  Here the 3 x 3 input
// rows of the segment are copied into LDS as whole lines (even / odd columns apart, so that the stride-2 taps read
// consecutive words), wave w computes channels 8w .. 8w+7 of the 64 pixels (weights uniform per wave: scalar loads), and the
// 64 x 32 outputs leave through LDS as 1 KB of contiguous bytes per wave instruction.  Same fma chain, (c, r, s) ascending:
// bit-identical to the kernel above (tests/test_gpu_ops.py).
constexpr int ST_TW = 64, ST_OLD = 36;
__global__ void __launch_bounds__(256) stem_conv_row_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ scale, const float *__restrict__ shift,
                                                            float *y, int y_ld, int N, int H, int W, int Ho, int Wo, int K, int act,
                                                            float *amax_out, int tiles_x) {
    __shared__ float s_in[3][3][2][ST_TW + 2];
    __shared__ __attribute__((aligned(16))) float s_out[ST_TW][ST_OLD];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x % tiles_x, ho = (blockIdx.x / tiles_x) % Ho, n = blockIdx.x / (tiles_x * Ho);
    const int wo0 = tile * ST_TW;
    // ---- input: rows 2 ho - 1 .. 2 ho + 1 of the three channels, columns 2 wo0 - 1 .. 2 wo0 + 127
    for (int i = tid; i < 9 * (2 * ST_TW + 1); i += 256) {
        const int j = i % (2 * ST_TW + 1), cr = i / (2 * ST_TW + 1);
        const int c = cr / 3, r = cr - 3 * c;
        const int hi = 2 * ho - 1 + r, wi = 2 * wo0 - 1 + j;
        float v = 0.f;
        if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) v = x[(((long long)n * 3 + c) * H + hi) * W + wi];
        // j even: odd input column 2 (wo0 + j / 2) - 1 (taps s = 0 of pixel j / 2, s = 2 of pixel j / 2 - 1); j odd: the even column (s = 1)
        s_in[c][r][(j & 1) ^ 1][j >> 1] = v;
    }
    __syncthreads();
    const int px = tid & 63, wv = tid >> 6;
    float in[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            in[c * 9 + r * 3 + 0] = s_in[c][r][1][px];
            in[c * 9 + r * 3 + 1] = s_in[c][r][0][px];
            in[c * 9 + r * 3 + 2] = s_in[c][r][1][px + 1];
        }
    const int k0 = blockIdx.y * 32 + wv * 8;
    const bool live = wo0 + px < Wo;
    float amx = 0.f;
    floatx4 o4[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = __builtin_amdgcn_readfirstlane(k0 + u);
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) acc = fmaf(in[t], w[k * 27 + t], acc);
        const float v = ppy_apply_act(fmaf(acc, scale[k], shift[k]), act);
        o4[u >> 2][u & 3] = v;
        amx = fmaxf(amx, live ? fabsf(v) : 0.f);
    }
    *reinterpret_cast<floatx4 *>(&s_out[px][wv * 8]) = o4[0];
    *reinterpret_cast<floatx4 *>(&s_out[px][wv * 8 + 4]) = o4[1];
    __syncthreads();
    // ---- output: 64 pixels x 128 B, contiguous in HBM (pixel stride y_ld floats)
    float *yrow = y + (((long long)n * Ho + ho) * Wo + wo0) * y_ld + blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = (tid >> 3) + 32 * i, c4 = (tid & 7) * 4;
        if (wo0 + q < Wo) *reinterpret_cast<floatx4 *>(yrow + (long long)q * y_ld + c4) = *reinterpret_cast<const floatx4 *>(&s_out[q][c4]);
    }
    if (amax_out) amax_track(amx, n, amax_out, blockIdx.x * 4 + wv);      // tracked per-image max|y| (f16x2 consumers)
}

// Round 4: the same operator on the matrix pipe.  The row kernel above spends 83 us on 1.28 GFLOP and 130 MB -- 12 160 small
// workgroups, each two barriers and 216 dependent fma per thread deep.  Here the 27-deep reduction (padded to 32) is TWO k-steps of
// v_mfma_f32_32x32x16_bf16 on exactly-split operands (three bf16 terms per fp32 value, six partial products, fp32 accumulate:
// conv_x3.hip's bf16x3 scheme -- no scaling, no range question for an arbitrary input image), a persistent workgroup of four
// waves walks over tiles of 2 output rows x 64 pixels, wave w owning 32 pixels of a row: the five input rows of a tile are copied
// to LDS as whole lines (even / odd columns apart, as above), a lane gathers its pixel's 8 taps per k-group from there, and the
// accumulator layout (lane = channel, 16 pixel rows per lane) stores 128 contiguous bytes per half-wave: whole lines, no LDS
// transposition.  Weights: 32 x 27 -> three bf16 planes of B fragments in 24 VGPRs, built once per workgroup.
// Error: the bf16x3 split is exact, the six products drop a 2^-24 tail like the fp32 fma chain (tests/test_gpu_ops.py).
constexpr int SM_TR = 2, SM_TC = 64;
__global__ void __launch_bounds__(256) stem_conv_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             float *y, int y_ld, int N, int H, int W, int Ho, int Wo, int act,
                                                             float *amax_out, int tiles_x, int tiles_y) {
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float floatx2_t;
    typedef __attribute__((ext_vector_type(16))) float floatx16_t;
    typedef __attribute__((ext_vector_type(4))) unsigned uint4_t;
    constexpr int NR = 2 * SM_TR + 1, LD = SM_TC + 2;
    __shared__ float s_in[3][NR][2][LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 5, fr = lane & 31;
    auto pk = [](float a, float b) -> unsigned {
        const floatx2_t v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    };
    // 8 consecutive-k values -> three bf16x8 operands (exact: the third term holds what two roundings left)
    auto split8 = [&](const float (&v)[8], uint4_t (&out)[3]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = v[2 * q], b = v[2 * q + 1];
            const unsigned P0 = pk(a, b);
            const float ra = a - __uint_as_float(P0 << 16), rb = b - __uint_as_float(P0 & 0xffff0000u);
            const unsigned P1 = pk(ra, rb);
            const float sa = ra - __uint_as_float(P1 << 16), sb = rb - __uint_as_float(P1 & 0xffff0000u);
            out[0][q] = P0;
            out[1][q] = P1;
            out[2][q] = pk(sa, sb);
        }
    };
    // B fragments: column = channel fr, k = 16 ks + 8 g + i  (k = 9 c + 3 r + s; k >= 27: zero)
    uint4_t bfr[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 16 * ks + 8 * g + i;
            v[i] = k < 27 ? w[fr * 27 + k] : 0.f;
        }
        split8(v, bfr[ks]);
    }
    const float sc = scale[fr], sh = shift[fr];
    const int row = wv >> 1, px = (wv & 1) * 32 + fr;        // this lane's pixel of the tile (A fragment row fr)
    const long long ntiles = (long long)N * tiles_y * tiles_x;
    int run_n = -1;
    float run_mx = 0.f;
    // the 3 x NR input rows of 2 SM_TC + 1 columns of a tile: half a workgroup per row, a lane per column -- no division per element (the
    // flat index walk this replaces spent more VALU cycles on i % 129 and i / 129 than the tile spends on its MFMAs: 50 -> 41 us).
    // (Requesting them one tile ahead into registers measured SLOWER, 48 us: eight workgroups per CU already hide the latency.)
    static_assert(2 * SM_TC == 128, "a half workgroup = one input row");
    constexpr int NIT = (3 * NR + 1) / 2;
    const int half = tid >> 7, jcol = tid & 127;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y), n = (int)(t / ((long long)tiles_x * tiles_y));
        const int ho0 = ty * SM_TR, wo0 = tx * SM_TC;
        __syncthreads();                                     // (the previous tile's gathers are done)
        {
            const int wi = 2 * wo0 - 1 + jcol;
            const bool wok = (unsigned)wi < (unsigned)W;
            float pre[NIT], pre_l = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {               // loads first (all in flight), LDS writes after
                const int cr = 2 * it + half;
                const int c = (cr >= NR) + (cr >= 2 * NR), r = cr - NR * c;
                const int hi = 2 * ho0 - 1 + r;
                pre[it] = 0.f;
                if (cr < 3 * NR && wok && (unsigned)hi < (unsigned)H) pre[it] = x[(((long long)n * 3 + c) * H + hi) * W + wi];
            }
            if (tid < 3 * NR) {                              // the last column (j = 2 SM_TC) of row `tid`
                const int c = (tid >= NR) + (tid >= 2 * NR), r = tid - NR * c;
                const int hi = 2 * ho0 - 1 + r, wl = 2 * wo0 - 1 + 2 * SM_TC;
                if ((unsigned)hi < (unsigned)H && (unsigned)wl < (unsigned)W) pre_l = x[(((long long)n * 3 + c) * H + hi) * W + wl];
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int cr = 2 * it + half;
                const int c = (cr >= NR) + (cr >= 2 * NR), r = cr - NR * c;
                if (cr < 3 * NR) s_in[c][r][(jcol & 1) ^ 1][jcol >> 1] = pre[it];      // j even: odd input column (taps s = 0 / s = 2 of the pixel before)
            }
            if (tid < 3 * NR) {
                const int c = (tid >= NR) + (tid >= 2 * NR), r = tid - NR * c;
                s_in[c][r][1][SM_TC] = pre_l;
            }
        }
        __syncthreads();
        floatx16_t acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 16 * ks + 8 * g + i;
                const int c = k / 9, r = (k - 9 * c) / 3, sx = k - 9 * c - 3 * r;
                v[i] = k < 27 ? s_in[c][2 * row + r][sx == 1 ? 0 : 1][px + (sx == 2 ? 1 : 0)] : 0.f;
            }
            uint4_t a[3];
            split8(v, a);
            // partial products, smallest first (conv_x3.hip): (a2,b0) (a1,b1) (a0,b2) (a1,b0) (a0,b1) (a0,b0)
            constexpr int ta[6] = {2, 1, 0, 1, 0, 0}, tb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ta[q]]), __builtin_bit_cast(bf16x8_t, bfr[ks][tb[q]]),
                                                              acc, 0, 0, 0);
        }
        // accumulator element e of this lane: pixel (e & 3) + 8 (e >> 2) + 4 g of the wave's 32, channel fr
        if (n != run_n) {
            if (amax_out && run_n >= 0) amax_track(run_mx, run_n, amax_out, (int)blockIdx.x * 4 + wv);
            run_n = n;
            run_mx = 0.f;
        }
        // (stores straight from the accumulator layout: a half-wave writes one whole 128-byte pixel per instruction; going through
        // an LDS patch for 16-byte stores measured the same 50-52 us and costs 18 KB of LDS, i.e. two workgroups per CU)
        const int ho = ho0 + row;
        float *yrow = y + (((long long)n * Ho + ho) * Wo) * y_ld + fr;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int wo = wo0 + (wv & 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
            const float o = ppy_apply_act(fmaf(acc[e], sc, sh), act);
            if (ho < Ho && wo < Wo) {
                yrow[(long long)wo * y_ld] = o;
                run_mx = fmaxf(run_mx, fabsf(o));
            }
        }
    }
    if (amax_out && run_n >= 0) amax_track(run_mx, run_n, amax_out, (int)blockIdx.x * 4 + wv);
}

// ---------------------------------------------------------------------------------------
// MaxPool2d(3, 2, 1): implicit -inf padding == skip out-of-range taps.
__global__ void __launch_bounds__(256) maxpool3x3s2_kernel(const float *__restrict__ x, int x_ld,
                                                           float *y, int y_ld, int N, int H, int W,
                                                           int C4, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long pix = i / C4;
        const int wo = (int)(pix % Wo);
        const int ho = (int)((pix / Wo) % Ho);
        const int n = (int)(pix / ((long long)Wo * Ho));
        floatx4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = ho * 2 - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = wo * 2 - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                const floatx4 v = *reinterpret_cast<const floatx4 *>(
                    x + (((long long)n * H + hi) * W + wi) * x_ld + c4 * 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) m[u] = fmaxf(m[u], v[u]);
            }
        }
        *reinterpret_cast<floatx4 *>(y + pix * y_ld + c4 * 4) = m;
    }
}

// AvgPool2d(2, 2, 0): ((x00 + x01) + x10 + x11) / 4 in torch's accumulation order.
__global__ void __launch_bounds__(256) avgpool2x2_kernel(const float *__restrict__ x, int x_ld,
                                                         float *y, int y_ld, int N, int H, int W,
                                                         int C4, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long pix = i / C4;
        const int wo = (int)(pix % Wo);
        const int ho = (int)((pix / Wo) % Ho);
        const int n = (int)(pix / ((long long)Wo * Ho));
        const float *p = x + (((long long)n * H + 2 * ho) * W + 2 * wo) * x_ld + c4 * 4;
        const floatx4 a = *reinterpret_cast<const floatx4 *>(p);
        const floatx4 b = *reinterpret_cast<const floatx4 *>(p + x_ld);
        const floatx4 c = *reinterpret_cast<const floatx4 *>(p + (long long)W * x_ld);
        const floatx4 d = *reinterpret_cast<const floatx4 *>(p + (long long)(W + 1) * x_ld);
        floatx4 r;
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = (((a[u] + b[u]) + c[u]) + d[u]) * 0.25f;
        *reinterpret_cast<floatx4 *>(y + pix * y_ld + c4 * 4) = r;
    }
}

// ---------------------------------------------------------------------------------------
// SPP: max-pool 5 / 9 / 13 (stride 1, "same"; reference model/custom_layers.py:275-290).  pool9 = pool5(pool5), pool13 =
// pool5(pool9) (max is associative and the -inf border makes the cascade exact), each pool5 separable.
// One workgroup = one image x SPP_CC channels; the H*W*SPP_CC tile and one temp live in LDS.  Round 3: a thread owns UNITS of
// (pixel, 4 channels) -- 16-byte global and LDS accesses, the pixel coordinates of a unit computed once -- and a workgroup takes
// 16 channels (64 contiguous bytes per pixel): 46 -> us on the 19x19x512 map of the R50vd head (it was 4-byte accesses of
// 32-byte pixel segments with two integer divisions per element and pass).
constexpr int SPP_THREADS = 512;

template <int CC>      // channels per workgroup: 16 (maps up to 29x29 in 160 KB of LDS), 8 beyond
__global__ void __launch_bounds__(SPP_THREADS) spp_kernel(const float *__restrict__ x, int x_ld, float *y5,
                                                          float *y9, float *y13, int y_ld, int H, int W,
                                                          int C) {
    constexpr int G = CC / 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int units = H * W * G;
    floatx4 *t0 = reinterpret_cast<floatx4 *>(sm), *t1 = t0 + units, *t2 = t1 + units;
    const int n = blockIdx.y, c0 = blockIdx.x * CC;
    const int tid = threadIdx.x;
    const long long img = (long long)n * H * W;
    const floatx4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int u0 = tid; u0 < units; u0 += 4 * SPP_THREADS) {      // (four requests in flight per thread: a 19x19 map is ONE round trip, not three)
        floatx4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = min(u0 + k * SPP_THREADS, units - 1);
            v[k] = *reinterpret_cast<const floatx4 *>(x + (img + u / G) * x_ld + c0 + (u % G) * 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (u0 + k * SPP_THREADS < units) t0[u0 + k * SPP_THREADS] = v[k];
    }
    __syncthreads();
    float *outs[3] = {y5, y9, y13};
    floatx4 *src = t0, *dst = t2;
    for (int lvl = 0; lvl < 3; ++lvl) {
        for (int u = tid; u < units; u += SPP_THREADS) {          // horizontal
            const int pix = u / G, w = pix % W;
            floatx4 m = ninf;
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                const floatx4 v = (unsigned)(w + d) < (unsigned)W ? src[u + d * G] : ninf;
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
            t1[u] = m;
        }
        __syncthreads();
        for (int u = tid; u < units; u += SPP_THREADS) {          // vertical, and out
            const int pix = u / G, h = pix / W;
            floatx4 m = ninf;
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                const floatx4 v = (unsigned)(h + d) < (unsigned)H ? t1[u + d * W * G] : ninf;
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
            dst[u] = m;
            *reinterpret_cast<floatx4 *>(outs[lvl] + (img + pix) * y_ld + c0 + (u % G) * 4) = m;
        }
        floatx4 *t = src;
        src = dst;
        dst = t;
        __syncthreads();
    }
}

int grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" int ppy_stem_conv3x3s2_nchw_f32(const float *x_nchw, const float *w_kcrs, const float *scale,
                                           const float *shift, float *y, int y_ld, int N, int H, int W,
                                           int K, int act, float *amax_out, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x_nchw && w_kcrs && scale && shift && y);
    PPY_CHECK_ARG(N > 0 && H > 0 && W > 0 && K > 0 && K % 16 == 0 && y_ld >= K && y_ld % 4 == 0);
    PPY_CHECK_ARG(((uintptr_t)y & 15) == 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * Ho * Wo;
    const char *sw = getenv("PPY_STEM_OLD");      // A/B switch, read per call: tests compare the two forms bit for bit
    const bool old_form = sw && sw[0] == '1';
    const int tiles_x = (Wo + ST_TW - 1) / ST_TW;
    if (K % 32 == 0 && !old_form && (long long)N * Ho * tiles_x < (1LL << 31)) {
        hipLaunchKernelGGL(stem_conv_row_kernel, dim3((unsigned)(N * Ho * tiles_x), K / 32), dim3(256), 0, (hipStream_t)stream, x_nchw,
                           w_kcrs, scale, shift, y, y_ld, N, H, W, Ho, Wo, K, act, amax_out, tiles_x);
        return ppy_launch_status();
    }
    dim3 grid((unsigned)((total + 255) / 256), K / 16);
    hipLaunchKernelGGL(stem_conv_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, x_nchw, w_kcrs, scale,
                       shift, y, y_ld, N, H, W, Ho, Wo, K, act, amax_out);
    return ppy_launch_status();
}

// The stem on the bf16 MFMA (stem_conv_mfma_kernel; K = 32 only -- both backbones').  Same arguments as
// ppy_stem_conv3x3s2_nchw_f32; results agree with its fp32 fma chain to fp32 rounding, not bit for bit.
extern "C" int ppy_stem_conv3x3s2_nchw_x3_f32(const float *x_nchw, const float *w_kcrs, const float *scale,
                                              const float *shift, float *y, int y_ld, int N, int H, int W,
                                              int K, int act, float *amax_out, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x_nchw && w_kcrs && scale && shift && y);
    PPY_CHECK_ARG(N > 0 && H > 0 && W > 0 && y_ld >= K && y_ld % 4 == 0 && ((uintptr_t)y & 15) == 0);
    if (K != 32) return PPY_ERR_UNSUPPORTED;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int tiles_x = (Wo + SM_TC - 1) / SM_TC, tiles_y = (Ho + SM_TR - 1) / SM_TR;
    const long long ntiles = (long long)N * tiles_x * tiles_y;
    const unsigned grid = (unsigned)(ntiles < 2048 ? ntiles : 2048);      // eight workgroups of four waves per CU
    hipLaunchKernelGGL(stem_conv_mfma_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w_kcrs, scale, shift, y, y_ld,
                       N, H, W, Ho, Wo, act, amax_out, tiles_x, tiles_y);
    return ppy_launch_status();
}

extern "C" int ppy_maxpool3x3s2_f32(const float *x, int x_ld, float *y, int y_ld, int N, int H, int W, int C,
                                    void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    PPY_CHECK_ARG(x_ld >= C && y_ld >= C && x_ld % 4 == 0 && y_ld % 4 == 0);
    PPY_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, x_ld,
                       y, y_ld, N, H, W, C / 4, Ho, Wo);
    return ppy_launch_status();
}

extern "C" int ppy_avgpool2x2_f32(const float *x, int x_ld, float *y, int y_ld, int N, int H, int W, int C,
                                  void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && y && N > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0);
    PPY_CHECK_ARG(x_ld >= C && y_ld >= C && x_ld % 4 == 0 && y_ld % 4 == 0);
    PPY_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0);
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(avgpool2x2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, x_ld, y,
                       y_ld, N, H, W, C / 4, Ho, Wo);
    return ppy_launch_status();
}

extern "C" int ppy_spp_f32(const float *x, int x_ld, float *y5, float *y9, float *y13, int y_ld, int N, int H,
                           int W, int C, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && y5 && y9 && y13 && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
    PPY_CHECK_ARG(x_ld >= C && y_ld >= C && x_ld % 4 == 0 && y_ld % 4 == 0);
    PPY_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y5 & 15) == 0 && ((uintptr_t)y9 & 15) == 0 && ((uintptr_t)y13 & 15) == 0);
    const int cc = (C % 16 == 0 && (size_t)3 * H * W * 16 * sizeof(float) <= 160 * 1024) ? 16 : 8;
    const size_t lds = (size_t)3 * H * W * cc * sizeof(float);
    if (lds > 160 * 1024) return PPY_ERR_UNSUPPORTED;
    static PpyLdsAttr attr16, attr8;
    if (cc == 16) {
        if (ppy_lds_attr(attr16, reinterpret_cast<const void *>(spp_kernel<16>), 160 * 1024) != PPY_OK) return PPY_ERR_LAUNCH;
        hipLaunchKernelGGL(spp_kernel<16>, dim3(C / 16, N), dim3(SPP_THREADS), lds, (hipStream_t)stream, x, x_ld, y5, y9, y13, y_ld, H, W, C);
    } else {
        if (ppy_lds_attr(attr8, reinterpret_cast<const void *>(spp_kernel<8>), 160 * 1024) != PPY_OK) return PPY_ERR_LAUNCH;
        hipLaunchKernelGGL(spp_kernel<8>, dim3(C / 8, N), dim3(SPP_THREADS), lds, (hipStream_t)stream, x, x_ld, y5, y9, y13, y_ld, H, W, C);
    }
    return ppy_launch_status();
}
