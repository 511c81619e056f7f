// Implicit-GEMM convolution on fp32 MFMA for gfx950 (MI355X).
//
// Replaces Conv2dUnit.forward of the reference (model/custom_layers.py:243-253): conv +
// eval-BatchNorm affine + activation, plus the residual add / nearest-x2 upsample /
// CoordConv terms that surround it in model/resnet_vd.py and model/head.py.
//
// GEMM view:  M = N*Ho*Wo output pixels, Ncol = K output channels, Kred = R*S*C.
//   A[m][(r,s,c)] = x[n, ho*stride+r-pad, wo*stride+s-pad, c]   (NHWC: c contiguous, never
//                                                                 materialised: im2col-free)
//   B[k][(r,s,c)] = w[k][r][s][c]                                (KRSC)
// The reduction is walked in chunks of BK=32 channels of ONE filter tap (C % 32 == 0), so an
// A-tile row is 128 contiguous bytes of one input pixel -> 16-byte coalesced loads, and a
// padding tap is a zero row.  Chunk order is (channel-chunk outer, tap inner) so the 9 taps of
// a 3x3 re-read the same input lines back-to-back (L1/L2 hits).
//
// Workgroup = 256 threads = 4 waves (one per SIMD).  Tiles are staged global -> VGPR -> LDS
// (double-buffered, one barrier per chunk); each wave owns a WM x WN sub-tile built from
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/SIMD; dependent-accumulator latency == issue
// interval, so one accumulator chain per 32x32 tile already saturates the pipe).
//
// LDS layout: row-major [rows][36] floats (32 + 4 pad): ds_read_b128 of 16 distinct rows at
// one column hits 16 distinct 16-byte slots (144-byte stride), and the 128-byte row written by
// 8 consecutive lanes with ds_write_b128 is contiguous -> both conflict-free.
// K-slot trick: within an 8-wide k group lane-half h reads floats [4h, 4h+4); the t-th MFMA of
// the group contracts k = {t, 4+t}.  A and B use the same permutation, so the sum is unchanged
// and every LDS read is a b128.
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 36;

struct ConvArgs {
    const float *x, *w, *scale, *shift, *res, *posb;
    float *y, *part;
    int x_ld, res_ld, y_ld;
    int N, H, W, C, Ho, Wo, K, R, S, stride, pad, act, ups;
    int M, Kred, cchunks, chunks_total, chunks_per_split;
};

__device__ __forceinline__ void epilogue_store(const ConvArgs &p, int m, int col, float v,
                                               float sc, float sh) {
    const int hw = p.Ho * p.Wo;
    if (p.posb) v += p.posb[(long long)(m % hw) * p.K + col];
    v = fmaf(v, sc, sh);
    if (p.res) v += p.res[(long long)m * p.res_ld + col];
    v = ppy_apply_act(v, p.act);
    if (!p.ups) {
        p.y[(long long)m * p.y_ld + col] = v;
    } else {
        const int n = m / hw, rem = m - n * hw;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const long long W2 = 2LL * p.Wo;
        float *o = p.y + (((long long)n * 2 * p.Ho + 2 * ho) * W2 + 2 * wo) * p.y_ld + col;
        o[0] = v;
        o[p.y_ld] = v;
        o[W2 * p.y_ld] = v;
        o[(W2 + 1) * p.y_ld] = v;
    }
}

template <int BM, int BN, int WM, int WN, bool SPLIT>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvArgs p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    constexpr int A_PER = BM / 32, B_PER = BN / 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;                     // [2][BM][LDS_LD]
    float *sB = smem + 2 * BM * LDS_LD;   // [2][BN][LDS_LD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tiles_n = (p.K + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int kc_begin = split * p.chunks_per_split;
    const int kc_end = min(kc_begin + p.chunks_per_split, p.chunks_total);

    // ---- per-thread loader coordinates: row = (tid>>3) + 32*j, 16-byte column tid&7 ----
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
    long long a_base[A_PER];
    int a_hi0[A_PER], a_wi0[A_PER];
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int m = m0 + lrow + 32 * j;
        if (m < p.M) {
            const int n = m / hw, rem = m - n * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            a_hi0[j] = ho * p.stride - p.pad;
            a_wi0[j] = wo * p.stride - p.pad;
            a_base[j] = (((long long)n * p.H + a_hi0[j]) * p.W + a_wi0[j]) * p.x_ld + lc4;
        } else {
            a_hi0[j] = -(1 << 20);   // every tap fails the bounds test -> zero rows
            a_wi0[j] = -(1 << 20);
            a_base[j] = 0;
        }
    }
    const float *b_row[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
        const int k = min(n0 + lrow + 32 * j, p.K - 1);   // clamp: rows >= K are masked at store
        b_row[j] = p.w + (long long)k * p.Kred + lc4;
    }

    floatx4 ra[A_PER], rb[B_PER];
    const int RS = p.R * p.S;

    auto load_tiles = [&](int kc) {
        const int cc = kc / RS, tap = kc - cc * RS;
        const int r = tap / p.S, s = tap - r * p.S;
        const int coff = cc * BK;
        const long long tap_off = (long long)(r * p.W + s) * p.x_ld + coff;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int hi = a_hi0[j] + r, wi = a_wi0[j] + s;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const floatx4 *>(p.x + a_base[j] + tap_off);
            ra[j] = v;
        }
        const int woff = tap * p.C + coff;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) rb[j] = *reinterpret_cast<const floatx4 *>(b_row[j] + woff);
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < A_PER; ++j)
            *reinterpret_cast<floatx4 *>(sA + (buf * BM + lrow + 32 * j) * LDS_LD + lc4) = ra[j];
#pragma unroll
        for (int j = 0; j < B_PER; ++j)
            *reinterpret_cast<floatx4 *>(sB + (buf * BN + lrow + 32 * j) * LDS_LD + lc4) = rb[j];
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag_off = (lane & 31) * LDS_LD + (lane >> 5) * 4;
    auto compute = [&](int buf) {
        const float *a_ptr = sA + (buf * BM + wm * WM) * LDS_LD + frag_off;
        const float *b_ptr = sB + (buf * BN + wn * WN) * LDS_LD + frag_off;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            floatx4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * LDS_LD + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const floatx4 *>(b_ptr + j * 32 * LDS_LD + q * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop: register-staged double buffering, one barrier per chunk ----
    if (kc_begin < kc_end) {
        load_tiles(kc_begin);
        store_tiles(0);
        __syncthreads();
        int cur = 0;
        for (int kc = kc_begin; kc < kc_end; ++kc) {
            const bool more = kc + 1 < kc_end;
            if (more) load_tiles(kc + 1);     // global loads in flight under the MFMAs
            compute(cur);
            if (more) store_tiles(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue: lane l holds channel (l&31) of 16 pixels per 32x32 tile ----
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WN + j * 32 + (lane & 31);
        const bool colok = col < p.K;
        float sc = 1.f, sh = 0.f;
        if (!SPLIT && colok) {
            sc = p.scale[col];
            sh = p.shift[col];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int m = m0 + wm * WM + i * 32 + row;
                if (colok && m < p.M) {
                    if (SPLIT)
                        p.part[((long long)split * p.M + m) * p.K + col] = acc[i][j][e];
                    else
                        epilogue_store(p, m, col, acc[i][j][e], sc, sh);
                }
            }
        }
    }
}

// Deterministic split-K combine (fixed z order) + the same epilogue.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ConvArgs p, int splits) {
    const long long total = (long long)p.M * p.K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / p.K), col = (int)(i - (long long)m * p.K);
        float v = p.part[i];
        for (int z = 1; z < splits; ++z) v += p.part[(long long)z * total + i];
        epilogue_store(p, m, col, v, p.scale[col], p.shift[col]);
    }
}

struct TileCfg {
    int bm, bn, wm, wn;
};
constexpr TileCfg kCfgs[] = {
    {128, 128, 64, 64},  // 0
    {128, 64, 64, 32},   // 1
    {64, 128, 32, 64},   // 2
    {64, 64, 32, 32},    // 3
    {256, 32, 64, 32},   // 4
    {128, 32, 32, 32},   // 5
    {32, 128, 32, 32},   // 6
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

template <int BM, int BN, int WM, int WN>
int launch_cfg(const ConvArgs &p, int splits, hipStream_t stream) {
    const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.K, BN);
    dim3 grid(tiles, splits), block(256);
    if (splits > 1) {
        auto k = conv_igemm_kernel<BM, BN, WM, WN, true>;
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return PPY_ERR_LAUNCH;
            attr_done = true;
        }
        hipLaunchKernelGGL(k, grid, block, lds, stream, p);
        const long long total = (long long)p.M * p.K;
        const int rgrid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rgrid), dim3(256), 0, stream, p, splits);
    } else {
        auto k = conv_igemm_kernel<BM, BN, WM, WN, false>;
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return PPY_ERR_LAUNCH;
            attr_done = true;
        }
        hipLaunchKernelGGL(k, grid, block, lds, stream, p);
    }
    return ppy_launch_status();
}

struct Geometry {
    int Ho, Wo, M, Kred, chunks;
};

bool conv_geometry(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, Geometry *g) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0)
        return false;
    if (C % BK != 0) return false;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return false;
    const long long M = (long long)N * Ho * Wo;
    if (M > 0x7fffffffLL / 4) return false;
    g->Ho = Ho;
    g->Wo = Wo;
    g->M = (int)M;
    g->Kred = R * S * C;
    g->chunks = R * S * (C / BK);
    return true;
}

// Cost model (cycles on one CU) used when the caller does not force a configuration.
// A work item (tile x split) costs BM*BN/4 MFMA-cycles per chunk on a CU (4 SIMDs x 64
// FLOP/clk); items are dealt round-robin to 256 CUs.  Small tiles pay more LDS/L1 traffic per
// FLOP (eff), split-K pays a combine pass through HBM plus a kernel boundary.
void pick_config(const Geometry &g, int K, int *cfg_out, int *split_out) {
    static const double eff[kNumCfgs] = {1.00, 0.95, 0.95, 0.88, 0.85, 0.80, 0.80};
    static const int split_opts[] = {1, 2, 3, 4, 6, 8, 9, 12, 16, 18};
    double best = 1e30;
    int bc = 3, bs = 1;
    for (int c = 0; c < kNumCfgs; ++c) {
        const TileCfg &t = kCfgs[c];
        // do not pick tiles much wider than the problem
        if (t.bn > 32 && t.bn / 2 >= K) continue;
        const long long tiles = (long long)ceil_div(g.M, t.bm) * ceil_div(K, t.bn);
        for (int s : split_opts) {
            const int per = ceil_div(g.chunks, s);
            if (s > 1 && per < 4) break;
            if ((long long)(s - 1) * per >= g.chunks) continue;   // empty trailing split
            const long long items = tiles * s;
            const double rounds = (double)((items + 255) / 256);
            double cyc = rounds * (t.bm * t.bn / 4.0) * (per + 1.5) / eff[c];
            if (s > 1) cyc += 4000.0 + (double)(s + 1) * g.M * K * 4.0 / 2500.0;
            if (cyc < best) {
                best = cyc;
                bc = c;
                bs = s;
            }
        }
    }
    *cfg_out = bc;
    *split_out = bs;
}

}  // namespace

extern "C" int ppy_conv2d_num_configs(void) { return kNumCfgs; }

extern "C" int ppy_conv2d_pick(int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                               int *cfg_out, int *splitk_out) {
    Geometry g;
    if (!conv_geometry(N, H, W, C, K, R, S, stride, pad, &g)) return PPY_ERR_BAD_ARG;
    int c, s;
    pick_config(g, K, &c, &s);
    if (cfg_out) *cfg_out = c;
    if (splitk_out) *splitk_out = s;
    return PPY_OK;
}

static int resolve(const Geometry &g, int K, int cfg, int splitk, int *c, int *s) {
    if (cfg >= kNumCfgs) return PPY_ERR_BAD_ARG;
    int hc, hs;
    pick_config(g, K, &hc, &hs);
    *c = cfg < 0 ? hc : cfg;
    *s = splitk <= 0 ? (cfg < 0 ? hs : 1) : splitk;
    if (*s > g.chunks) *s = g.chunks;
    // normalise so that no split is empty
    const int per = ceil_div(g.chunks, *s);
    *s = ceil_div(g.chunks, per);
    return PPY_OK;
}

extern "C" size_t ppy_conv2d_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride,
                                             int pad, int cfg, int splitk) {
    Geometry g;
    if (!conv_geometry(N, H, W, C, K, R, S, stride, pad, &g)) return 0;
    int c, s;
    if (resolve(g, K, cfg, splitk, &c, &s) != PPY_OK) return 0;
    return s > 1 ? (size_t)s * g.M * K * sizeof(float) : 0;
}

extern "C" int ppy_conv2d_bn_act_f32(const float *x, int x_ld, const float *w_krsc, const float *scale,
                                     const float *shift, const float *residual, int res_ld,
                                     const float *posbias, float *y, int y_ld, int N, int H, int W, int C,
                                     int K, int R, int S, int stride, int pad, int act, int upsample2x,
                                     int cfg, int splitk, void *ws, size_t ws_bytes, void *stream) {
    PPY_CHECK_ARG(x && w_krsc && scale && shift && y);
    Geometry g;
    if (!conv_geometry(N, H, W, C, K, R, S, stride, pad, &g)) return PPY_ERR_BAD_ARG;
    PPY_CHECK_ARG(x_ld >= C && x_ld % 4 == 0 && y_ld >= K);
    PPY_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_krsc & 15) == 0);
    PPY_CHECK_ARG(!residual || res_ld >= K);
    PPY_CHECK_ARG(act == PPY_ACT_NONE || act == PPY_ACT_RELU || act == PPY_ACT_LEAKY);
    int c, s;
    int rc = resolve(g, K, cfg, splitk, &c, &s);
    if (rc != PPY_OK) return rc;
    if (s > 1) {
        const size_t need = (size_t)s * g.M * K * sizeof(float);
        if (!ws || ws_bytes < need) return PPY_ERR_WORKSPACE;
    }
    ConvArgs p;
    p.x = x; p.w = w_krsc; p.scale = scale; p.shift = shift; p.res = residual; p.posb = posbias;
    p.y = y; p.part = (float *)ws;
    p.x_ld = x_ld; p.res_ld = res_ld; p.y_ld = y_ld;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = g.Ho; p.Wo = g.Wo; p.K = K; p.R = R; p.S = S;
    p.stride = stride; p.pad = pad; p.act = act; p.ups = upsample2x ? 1 : 0;
    p.M = g.M; p.Kred = g.Kred; p.cchunks = C / BK; p.chunks_total = g.chunks;
    p.chunks_per_split = ceil_div(g.chunks, s);
    hipStream_t st = (hipStream_t)stream;
    switch (c) {
        case 0: return launch_cfg<128, 128, 64, 64>(p, s, st);
        case 1: return launch_cfg<128, 64, 64, 32>(p, s, st);
        case 2: return launch_cfg<64, 128, 32, 64>(p, s, st);
        case 3: return launch_cfg<64, 64, 32, 32>(p, s, st);
        case 4: return launch_cfg<256, 32, 64, 32>(p, s, st);
        case 5: return launch_cfg<128, 32, 32, 32>(p, s, st);
        case 6: return launch_cfg<32, 128, 32, 32>(p, s, st);
    }
    return PPY_ERR_BAD_ARG;
}
