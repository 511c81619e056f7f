// Pieces shared by the convolution kernels (conv_igemm.hip: exact-fp32 MFMA; conv_x3.hip: 3-way
// bf16 operand split on the bf16 MFMA): argument block, epilogues, split-K combine, geometry.
#pragma once
#include "common.h"

#ifndef PPY_X3_BBLOCK
#define PPY_X3_BBLOCK 1   // f16x2 weight planes in [chunk][K][32] order (0 = [K][Kred], for A/B rebuilds); conv_x3.hip, conv_bwd.hip
#endif

struct ConvArgs {
    const float *x, *w, *scale, *shift, *res, *posb;
    const unsigned short *w3;    // weights as 3 bf16 planes [3][K][R][S][C] (conv_x3.hip), or NULL
    const unsigned short *wf16;  // weights * per-channel power of two as 2 fp16 planes [2][K][R][S][C], or NULL
    const float *scale_f16;      // `scale` with the inverse weight scale folded in (f16x2 kernels)
    const float *posb_f16;       // posb times the per-channel weight scale (f16x2 kernels), or NULL
    const float *amax_in;        // tracked per-image max|x| of the input tensor (N * AMAX_SLOTS slots), or NULL
    const float *amax_in2 = nullptr;   // round 6: a SECOND block of tracked maxima covering part of the input's channels (the folded
                                 // shortcut's wide buffer [conv2 output | pooled block input]: the pooled part keeps the slots of the
                                 // tensor it was pooled from); the launch scales by the larger of the two.  conv_amax_in() below.
    float *amax_out;             // where this launch records per-image max|y| (N * AMAX_SLOTS slots), or NULL
    float *y, *part;
    int x_ld, res_ld, y_ld;
    int N, H, W, C, Ho, Wo, K, R, S, stride, pad, act, ups;
    int M, Kred, cchunks, chunks_total, chunks_per_split;
    int nstages;                 // conv_x3 kernels: LDS stages (2..4 chunks resident; 2 in every other kernel)
    int panel_n = 0;             // tile order: column panels of this many N-tiles (ppy_panel_n below); 0 = one panel (tile_m-major)
    unsigned long long *trace;   // debug: per-workgroup timeline (ppy_debug_set_trace), NULL in production
    // training forward (ppy_conv2d_train_fwd_f32): per-channel BatchNorm statistics of y from the epilogue, as (n, mean, M2)
    // triples [slice][K][3] with one slice per wave row-tile (tile_bn_stats below); the launcher reports the slice count to
    // *bn_slices_host (a HOST pointer: never dereferenced on the device)
    float *bn_part = nullptr;
    int *bn_slices_host = nullptr;
    int bn_capacity = 0;         // slices bn_part has room for (a launcher that needs more returns PPY_ERR_WORKSPACE)
    // round 4, training forward of FROZEN layers on the streaming 1x1 kernel (conv_stream.hip): a first launch with bn_nostore
    // writes the statistics only, a second one -- the batch statistics are known then -- applies the BatchNorm to its own
    // accumulators, y = act((v - mean) * (invstd * gamma) + beta [+ res]) with v the convolution's value, the arithmetic of
    // bn_apply_kernel (train.hip): the raw tensor is never written or read
    int bn_nostore = 0;
    const float *bn_mean = nullptr, *bn_invstd = nullptr, *bn_gamma = nullptr, *bn_beta = nullptr;
    // "global pre-split" (round 3; f16x2 kernels): a producer whose output has ONE consumer, a convolution on
    // these kernels, stores it as the consumer's MFMA operand -- per 32-channel group of a pixel 32 fp16 first terms then 32 fp16
    // second terms of y * s (the same 128 bytes the fp32 values would take) -- with s = a power of two per image from a STATIC
    // bound of |y|: ysplit_mul * (an upper bound of the input's tracked maximum) + ysplit_add.  The consumer then reads finished
    // operands: no scale / split VALU work per (tap, wave) in its main loop.
    float *yscale = nullptr;        // producer: [N] per-image scales of the split output it writes (NULL: plain fp32 output)
    float ysplit_mul = 0.f, ysplit_add = 0.f;
    const float *xscale = nullptr;  // consumer: [N] per-image scales of its pre-split input (NULL: fp32 input, tracked maximum)
};

// scale of a pre-split tensor from a bound of its magnitude: the power of two that puts `bound` into [2^13, 2^14), as the
// activation scales of the f16x2 kernels (conv_x3.hip), clamped to [2^-24, 2^40]
static __device__ __forceinline__ float split_scale_of(float bound) {
    const int e = (int)((__float_as_uint(bound) >> 23) & 0xffu);
    int f = 267 - e;
    f = f < 103 ? 103 : (f > 167 ? 167 : f);
    return __uint_as_float((unsigned)f << 23);
}
static __device__ __forceinline__ float pow2_above(float mx) {       // the power of two in (mx, 2 mx]  (mx >= 0; 2^-126 for 0)
    const unsigned e = (__float_as_uint(mx) >> 23) & 0xffu;
    return __uint_as_float((e >= 253u ? 254u : e + 1u) << 23);
}
static __device__ __forceinline__ float pow2_inverse(float s) {      // 1 / s for s = 2^k, exactly
    return __uint_as_float((254u - ((__float_as_uint(s) >> 23) & 0xffu)) << 23);
}

#if defined(__HIPCC__)
// per-image max|x| of a launch's input: the maximum over its one or two tracked blocks (ConvArgs::amax_in, amax_in2)
static __device__ __forceinline__ float conv_amax_in(const ConvArgs &p, int n) {
    float mx = amax_read(p.amax_in, n);
    if (p.amax_in2) mx = fmaxf(mx, amax_read(p.amax_in2, n));
    return mx;
}
#endif

struct Geometry {
    int Ho, Wo, M, Kred, chunks;
};

// conv_x3.hip: configurations of the split-bf16 kernel (ids local to that file)
int ppy_x3_num_configs();
int ppy_x3_f16_base();        // first local id of the f16x2 scheme
int ppy_x3_dispatch(const ConvArgs &p, int local_cfg, int splits, hipStream_t stream);
// conv_stream.hip: persistent streaming kernel for 1x1 convolutions with C = 64 (f16x2 operands), optional 2x2 average output
// Tile order (round 5).  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2; every XCD takes one CONTIGUOUS range of
// the tile order.  With the tile_m-major order (one panel) an XCD owns a few rows of tiles and ALL columns: the activation rows
// cross the fabric once, the weights eight times -- right for the large-image layers, wrong for the 19x19 / 38x38 stages whose
// weights outweigh their activations (3x3 512 -> 1024 at 19x19: 160 MB fetched for 25 MB of operands, profiles/r05_pmc_layers.txt).
// With gn column panels an XCD owns tiles_m / (8 / gn) rows of ONE panel: fabric reads ~ gn x A + (8 / gn) x W.
// ppy_panel_n picks gn in {1, 2, 4, 8} for the smaller sum (host); ppy_tile_of maps blockIdx.x (device, all conv tile kernels).
static inline int ppy_panel_n(const ConvArgs &p, int BM, int BN, int splits) {
    static const int mode = getenv("PPY_TILE_PANEL") ? atoi(getenv("PPY_TILE_PANEL")) : 1;      // 0 = off, 1 = by the estimate, 2/4/8 = that many panels
    if (mode == 0 || splits > 1) return 0;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.K + BN - 1) / BN;
    if (tiles_m * tiles_n < 16) return 0;
    const double A = (double)p.N * p.H * p.W * p.C * 4.0, W = (double)p.K * p.Kred * 4.0;
    int best = 1;
    double best_cost = A + 8.0 * W;
    for (int gn = 2; gn <= 8; gn *= 2) {
        if (gn > tiles_n || 8 / gn > tiles_m) continue;
        if (mode > 1 && gn != mode) continue;
        const double cost = gn * A + (8.0 / gn) * W;
        if (cost < 0.9 * best_cost || (mode > 1 && gn == mode)) { best = gn; best_cost = cost; }
    }
    return best == 1 ? 0 : (tiles_n + best - 1) / best;
}
#if defined(__HIPCC__)
__device__ __forceinline__ void ppy_tile_of(const ConvArgs &p, int tiles_n, int &tile_m, int &tile_n) {
    const int nb = (int)gridDim.x, q = nb >> 3, r = nb & 7;
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    const int tile_id = xcd * q + min(xcd, r) + idx;
    const int pn = p.panel_n;
    if (pn <= 0 || pn >= tiles_n) {
        tile_m = tile_id / tiles_n;
        tile_n = tile_id - tile_m * tiles_n;
        return;
    }
    const int tiles_m = nb / tiles_n, per_panel = tiles_m * pn;
    const int panel = tile_id / per_panel, within = tile_id - panel * per_panel;
    const int pw = min(pn, tiles_n - panel * pn);              // (the last panel may be narrower)
    tile_m = within / pw;
    tile_n = panel * pn + (within - tile_m * pw);
}
#endif

int ppy_stream_num_configs();
int ppy_stream_dispatch(const ConvArgs &p, int local_cfg, float *pool, int pool_ld, hipStream_t stream);
// conv_patch.hip: 3x3 / stride 1 / pad 1 with C = 32 (the stem layers), input patch staged once per output tile (f16x2 operands)
int ppy_patch_num_configs();
int ppy_patch_dispatch(const ConvArgs &p, int local_cfg, hipStream_t stream);
int ppy_patch_maxpool_dispatch(const ConvArgs &p, int Hp, int Wp, hipStream_t stream);      // + MaxPool2d(3, 2, 1) from the epilogue; p.y = the pooled tensor
// conv_ws.hip: the f16x2 tiles with specialised waves (four deliver operands, four multiply)
int ppy_ws_num_configs();
int ppy_ws_dispatch(const ConvArgs &p, int local_cfg, int splits, hipStream_t stream);
// conv_small.hip (round 6): wave-private 32 x 32 / 32 x 64 output tiles for small outputs (batch 1, narrow layers); `splits` = k-parts
// INSIDE the workgroup (a power of two <= its waves, anything else is rounded down): no workspace, no combine launch
int ppy_small_num_configs();
int ppy_small_dispatch(const ConvArgs &p, int local_cfg, int splits, hipStream_t stream);

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 36;

// ---- exact operand splits for the 16-bit MFMA (conv_x3.hip header: bf16x3 = 3 bf16 terms, f16x2 = 2 fp16 terms) ----
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float floatx2;
typedef __attribute__((ext_vector_type(4))) unsigned uintx4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {    // RNE, a -> low half
    const floatx2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// halves of a packed pair back to fp32 (scalar bit casts: a vector bit_cast of the packed word was mis-combined
// across the pairs of a fragment by hipcc -O3 -- every pair subtracted the FIRST pair's value)
__device__ __forceinline__ float f16_lo(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p & 0xffffu)); }
__device__ __forceinline__ float f16_hi(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p >> 16)); }

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // RNE, a -> low half
    const floatx2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ float epilogue_store(const ConvArgs &p, int m, int col, float v,
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
    return v;
}

// Epilogue shared by both kernels.  Must be entered after a workgroup barrier that follows the
// last MFMA read of the operand tiles (the vector path reuses the LDS as a wave-private
// transpose patch).
template <int TM, int TN, int WM, int WN, bool SPLIT, bool VEC>
__device__ __forceinline__ void tile_epilogue(const ConvArgs &p, floatx16 (&acc)[TM][TN], float *smem, int m0,
                                              int n0, int wm, int wn, int lane, int wave, int split,
                                              const float (*rowscale)[4] = nullptr, const float (*rowsplit)[4] = nullptr,
                                              bool split_out = false) {
    // split_out (vector path, no split-K, no upsampling): the output is stored PRE-SPLIT for its one consumer (ConvArgs::yscale);
    // rowsplit = scale of the image of the rows (lane>>3) + 8t of tile i.  (A flag beside an always-valid array: a pointer that is
    // null at run time put the array into scratch memory.)
    // rowscale (vector path only): factor for the rows (lane>>3) + 8t of tile i that this lane finishes -- the f16x2
    // kernels undo their per-image activation scale here, after the transposition, instead of on the accumulators
    const int hw = p.Ho * p.Wo;
    if constexpr (VEC) {
        // ---- vector epilogue: each 32x32 accumulator tile goes through a wave-private LDS
        // patch so that a lane owns 4 consecutive channels of a pixel: 16-byte residual loads
        // and stores, 8 lanes per 128-byte row segment (the scalar form -- 4 bytes per lane --
        // measured ~2.2 TB/s on the output-heavy 1x1 layers vs 4.3 TB/s for float4 kernels).
        // The main loop ended with a barrier, so nobody reads the operand tiles any more.
        float *sE = smem + wave * (32 * LDS_LD);
        const int erow = lane >> 3, ec4 = (lane & 7) * 4;
        // max|y| per image of what this wave stores (p.amax_out), branch-free inside the store loop (an atomic behind a
        // branch there cost +70 % on the 1x1 expand layers: it fences the batched stores): rows before / from `bnd`, the
        // first row of the next image, go to two running maxima that are merged once at the end.
        const int mw0 = min(m0 + wm * WM, p.M - 1), mw1 = min(m0 + wm * WM + WM - 1, p.M - 1);
        const int n_lo = mw0 / hw, n_hi = mw1 / hw;
        const int bnd = (n_lo + 1) * hw;
        float amx = 0.0f, amx_hi = 0.0f;
        // All residual loads of the wave's sub-tile are issued up front (4*TM*TN x 16 B per lane): the 1x1
        // "expand" layers are bound by this read and the store below, and four loads in flight per wave
        // (one 32x32 tile at a time) left HBM at ~2.6 TB/s on them.
        floatx4 rv[TM][TN][4];
        if (!SPLIT && p.res) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int col = n0 + wn * WN + j * 32 + ec4;
                        const int m = m0 + wm * WM + i * 32 + erow + 8 * t;
                        rv[i][j][t] = floatx4{0.f, 0.f, 0.f, 0.f};
                        if (col < p.K && m < p.M)
                            rv[i][j][t] = *reinterpret_cast<const floatx4 *>(p.res + (long long)m * p.res_ld + col);
                    }
        } else if (!SPLIT && p.posb) {
            // round 6: the CoordConv position bias (a layer without shortcut: the head's convolutions) takes the same registers and is
            // requested up front as well -- loaded inside the store loop it stood between every tile's transposition and its stores, one
            // L2 round trip per 32 x 32 tile and wave (3x3 512 -> 1024 at 19x19: 96 us with the bias map, 82 us without)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int col = n0 + wn * WN + j * 32 + ec4;
                        const int m = m0 + wm * WM + i * 32 + erow + 8 * t;
                        rv[i][j][t] = floatx4{0.f, 0.f, 0.f, 0.f};
                        if (col < p.K && m < p.M)
                            rv[i][j][t] = *reinterpret_cast<const floatx4 *>(p.posb + (long long)(m % hw) * p.K + col);
                    }
        }
        // Round 6: NOTHING is loaded inside the store loop of a column tile.  The stores sit in divergent branches (row / column tails), so
        // the compiler cannot count them, and every first use of a loaded register behind one became `s_waitcnt vmcnt(0)` -- which on gfx950
        // also waits for the wave's outstanding STORES: one store in flight per wave, the whole epilogue a chain of write round trips (ISA
        // of the 192x256 tile: a vmcnt(0) in front of every 32-row group's arithmetic).  Every loaded register therefore passes through an
        // empty asm in front of the stores: the shortcut / bias rows once, the per-channel scale / shift of ALL column tiles with them where
        // the registers allow it (wave tiles of up to three 32 x 32 blocks), else at the head of their column tile (one wait per column tile).
        constexpr bool HOIST_SC = TM * TN < 4;
        floatx4 scv[HOIST_SC ? TN : 1], shv[HOIST_SC ? TN : 1];
        if constexpr (HOIST_SC) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * WN + j * 32 + ec4;
                scv[j] = floatx4{1.f, 1.f, 1.f, 1.f};
                shv[j] = floatx4{0.f, 0.f, 0.f, 0.f};
                if (!SPLIT && col < p.K) {
                    scv[j] = *reinterpret_cast<const floatx4 *>(p.scale + col);
                    shv[j] = *reinterpret_cast<const floatx4 *>(p.shift + col);
                }
            }
            if (!SPLIT) {
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(scv[j]), "+v"(shv[j]));
            }
        }
        if (!SPLIT && (p.res || p.posb)) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(rv[i][j][t]));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + ec4;
            const bool colok = col < p.K;
            floatx4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if constexpr (HOIST_SC) {
                sc = scv[j];
                sh = shv[j];
            } else if (!SPLIT) {
                if (colok) {
                    sc = *reinterpret_cast<const floatx4 *>(p.scale + col);
                    sh = *reinterpret_cast<const floatx4 *>(p.shift + col);
                }
                asm volatile("" : "+v"(sc), "+v"(sh));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mbase = m0 + wm * WM + i * 32 + erow;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    sE[row * LDS_LD + (lane & 31)] = acc[i][j][e];
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = mbase + 8 * t;
                    floatx4 v = *reinterpret_cast<const floatx4 *>(sE + (erow + 8 * t) * LDS_LD + ec4);
                    if (rowscale) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] *= rowscale[i][t];
                    }
                    if (colok && m < p.M) {
                        if (SPLIT) {
                            *reinterpret_cast<floatx4 *>(p.part + ((long long)split * p.M + m) * p.K + col) = v;
                        } else {
                            if (p.posb) {
                                floatx4 pb = rv[i][j][t];          // (requested in front of the loop; a layer with BOTH a bias map and a shortcut: here)
                                if (p.res) {                      // (its wait stays inside this branch: the empty asm is the first use)
                                    pb = *reinterpret_cast<const floatx4 *>(p.posb + (long long)(m % hw) * p.K + col);
                                    asm volatile("" : "+v"(pb));
                                }
#pragma unroll
                                for (int u = 0; u < 4; ++u) v[u] += pb[u];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                float o = fmaf(v[u], sc[u], sh[u]);
                                if (p.res) o += rv[i][j][t][u];
                                v[u] = ppy_apply_act(o, p.act);
                            }
                            {
                                const float rmx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                                amx = fmaxf(amx, m < bnd ? rmx : 0.0f);
                                amx_hi = fmaxf(amx_hi, m < bnd ? 0.0f : rmx);
                            }
                            if (rowsplit != nullptr && split_out) {
                                // two fp16 terms of v * s (RNE; the residual fma(v, s, -first) is exact), first terms of the
                                // pixel's 32-channel group in its bytes 0..63, second terms in bytes 64..127
                                typedef __attribute__((ext_vector_type(2))) unsigned uintx2_;
                                const float s = rowsplit[i][t];
                                const unsigned h0 = cvt_pk_f16(v[0] * s, v[1] * s), h1 = cvt_pk_f16(v[2] * s, v[3] * s);
                                const unsigned l0 = cvt_pk_f16(fmaf(v[0], s, -f16_lo(h0)), fmaf(v[1], s, -f16_hi(h0)));
                                const unsigned l1 = cvt_pk_f16(fmaf(v[2], s, -f16_lo(h1)), fmaf(v[3], s, -f16_hi(h1)));
                                char *o = reinterpret_cast<char *>(p.y + (long long)m * p.y_ld) + (col >> 5) * 128 + (col & 31) * 2;
                                *reinterpret_cast<uintx2_ *>(o) = uintx2_{h0, h1};
                                *reinterpret_cast<uintx2_ *>(o + 64) = uintx2_{l0, l1};
                            } else if (!p.ups) {
                                *reinterpret_cast<floatx4 *>(p.y + (long long)m * p.y_ld + col) = v;
                            } else {
                                const int n = m / hw, rem = m - n * hw;
                                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                                const long long W2 = 2LL * p.Wo;
                                float *o = p.y + (((long long)n * 2 * p.Ho + 2 * ho) * W2 + 2 * wo) * p.y_ld + col;
                                *reinterpret_cast<floatx4 *>(o) = v;
                                *reinterpret_cast<floatx4 *>(o + p.y_ld) = v;
                                *reinterpret_cast<floatx4 *>(o + W2 * p.y_ld) = v;
                                *reinterpret_cast<floatx4 *>(o + (W2 + 1) * p.y_ld) = v;
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (!SPLIT && p.amax_out) amax_track2(amx, amx_hi, n_lo, n_hi, p.amax_out, blockIdx.x * 8 + wave);
        return;
    }
    float amx = 0.0f, amx_hi = 0.0f;
    const int mw0 = min(m0 + wm * WM, p.M - 1), mw1 = min(m0 + wm * WM + WM - 1, p.M - 1);
    const int n_lo = mw0 / hw, n_hi = mw1 / hw;
    const int bnd = (n_lo + 1) * hw;
    // ---- scalar epilogue: lane l holds channel (l&31) of 16 pixels per 32x32 tile ----
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
                    {
                        const float a = fabsf(epilogue_store(p, m, col, acc[i][j][e], sc, sh));
                        amx = fmaxf(amx, m < bnd ? a : 0.0f);
                        amx_hi = fmaxf(amx_hi, m < bnd ? 0.0f : a);
                    }
                }
            }
        }
    }
    if (!SPLIT && p.amax_out) amax_track2(amx, amx_hi, n_lo, n_hi, p.amax_out, blockIdx.x * 8 + wave);
}

// Per-channel (n, mean, M2) of the values tile_epilogue is about to store for this wave's WM x WN sub-tile -- the first pass of a
// training-mode BatchNorm (reference model/custom_layers.py:243-253, torch.nn.BatchNorm2d on batch statistics) without reading
// y back: in the accumulator layout a lane holds 16 rows of ONE column per 32x32 tile, so a column's sum over the sub-tile is a
// sum over registers and one exchange with lane ^ 32.  Two passes over the registers (mean, then centred squares): as accurate
// as the two-pass slice kernel of train.hip; csrc/train.hip merges the slices with Chan's formula.  inv_sa: the f16x2 kernels'
// inverse activation scale of row (lane & 31) of tile i (the accumulators are still scaled).  Plain conv + bias only
// (no shortcut, no position bias, no activation, no upsampling): v = fma(acc * inv, scale, shift) as the epilogue computes it.
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void tile_bn_stats(const ConvArgs &p, const floatx16 (&acc)[TM][TN], const float (&inv_sa)[TM], int m0, int n0,
                                              int wm, int wn, int lane, int slice) {
    const int row0 = m0 + wm * WM;
    const int nvalid = min(max(p.M - row0, 0), WM);
    float inv[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) inv[i][e] = __shfl(inv_sa[i], (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5));
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WN + j * 32 + (lane & 31);
        const bool colok = col < p.K;
        const float sc = colok ? p.scale[col] : 1.f, sh = colok ? p.shift[col] : 0.f;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const float v = fmaf(acc[i][j][e] * inv[i][e], sc, sh);
                sum += r < nvalid ? v : 0.f;
            }
        sum += __shfl_xor(sum, 32);
        const float mean = nvalid > 0 ? sum / (float)nvalid : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const float d = fmaf(acc[i][j][e] * inv[i][e], sc, sh) - mean;
                m2 += r < nvalid ? d * d : 0.f;
            }
        m2 += __shfl_xor(m2, 32);
        if (colok && lane < 32) {
            float *o = p.bn_part + ((long long)slice * p.K + col) * 3;
            o[0] = (float)nvalid;
            o[1] = mean;
            o[2] = m2;
        }
    }
}

// 8 consecutive-k floats of one row -> three bf16x8 MFMA operands (exact 3-term split)
__device__ __forceinline__ void split8(const floatx4 lo, const floatx4 hi, bf16x8 (&out)[3]) {
    uintx4 p0, p1, p2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = q < 2 ? lo[2 * q] : hi[2 * q - 4], b = q < 2 ? lo[2 * q + 1] : hi[2 * q - 3];
        const unsigned P0 = cvt_pk_bf16(a, b);
        const float ra = a - __uint_as_float(P0 << 16), rb = b - __uint_as_float(P0 & 0xffff0000u);
        const unsigned P1 = cvt_pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(P1 << 16), sb = rb - __uint_as_float(P1 & 0xffff0000u);
        p0[q] = P0;
        p1[q] = P1;
        p2[q] = cvt_pk_bf16(sa, sb);
    }
    out[0] = __builtin_bit_cast(bf16x8, p0);
    out[1] = __builtin_bit_cast(bf16x8, p1);
    out[2] = __builtin_bit_cast(bf16x8, p2);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Deterministic split-K combine (fixed z order) + the same epilogue.  VEC: one 16-byte column
// group per thread (every load independent, 16-byte accesses); otherwise one element per thread.
template <bool VEC>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ConvArgs p, int splits) {
    constexpr int W = VEC ? 4 : 1;
    const long long total = (long long)p.M * p.K;
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * W;
    float amx = 0.0f;
    if (i < total) {
    const int m = (int)(i / p.K), col = (int)(i - (long long)m * p.K);
    if constexpr (VEC) {
        // the partials are requested EIGHT at a time and added in the fixed order z = 0, 1, 2, ... (round 6: the plain loop waited for
        // every load before it requested the next one -- a chain of `splits` memory round trips, 6-8 us for a few hundred KB)
        floatx4 v = *reinterpret_cast<const floatx4 *>(p.part + i);
        for (int z0 = 1; z0 < splits; z0 += 8) {
            floatx4 o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (z0 + q < splits) o[q] = *reinterpret_cast<const floatx4 *>(p.part + (long long)(z0 + q) * total + i);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (z0 + q < splits) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] += o[q][u];
                }
        }
        const floatx4 sc = *reinterpret_cast<const floatx4 *>(p.scale + col);
        const floatx4 sh = *reinterpret_cast<const floatx4 *>(p.shift + col);
        if (p.posb) {
            const floatx4 pb = *reinterpret_cast<const floatx4 *>(p.posb + (long long)(m % (p.Ho * p.Wo)) * p.K + col);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] += pb[u];
        }
        floatx4 r = {0.f, 0.f, 0.f, 0.f};
        if (p.res) r = *reinterpret_cast<const floatx4 *>(p.res + (long long)m * p.res_ld + col);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = ppy_apply_act(fmaf(v[u], sc[u], sh[u]) + (p.res ? r[u] : 0.f), p.act);
            amx = fmaxf(amx, fabsf(v[u]));
        }
        if (!p.ups) {
            *reinterpret_cast<floatx4 *>(p.y + (long long)m * p.y_ld + col) = v;
        } else {
            const int hw = p.Ho * p.Wo, n = m / hw, rem = m - n * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            const long long W2 = 2LL * p.Wo;
            float *o = p.y + (((long long)n * 2 * p.Ho + 2 * ho) * W2 + 2 * wo) * p.y_ld + col;
            *reinterpret_cast<floatx4 *>(o) = v;
            *reinterpret_cast<floatx4 *>(o + p.y_ld) = v;
            *reinterpret_cast<floatx4 *>(o + W2 * p.y_ld) = v;
            *reinterpret_cast<floatx4 *>(o + (W2 + 1) * p.y_ld) = v;
        }
    } else {
        float v = p.part[i];
        for (int z0 = 1; z0 < splits; z0 += 8) {      // (eight requests in flight, added in the fixed order: see the vector form)
            float o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (z0 + q < splits) o[q] = p.part[(long long)(z0 + q) * total + i];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (z0 + q < splits) v += o[q];
        }
        amx = fabsf(epilogue_store(p, m, col, v, p.scale[col], p.shift[col]));
    }
    }
    if (p.amax_out) {
        const long long ic = i < total ? i : total - 1;
        amax_track(amx, (int)(ic / p.K) / (p.Ho * p.Wo), p.amax_out, blockIdx.x * 4 + (threadIdx.x >> 6));
    }
}

// launch helper shared by both conv kernels
inline void launch_splitk_reduce(const ConvArgs &p, int splits, bool vec, hipStream_t stream) {
    const long long total = (long long)p.M * p.K;
    const long long threads = vec ? total / 4 : total;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (vec)
        hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(grid), dim3(256), 0, stream, p, splits);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(grid), dim3(256), 0, stream, p, splits);
}

// 16-byte epilogue accesses need 4-channel granularity and 16-byte aligned rows everywhere.
bool vec_epilogue_ok(const ConvArgs &p) {
    auto al = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
    if (p.K % 4 != 0 || p.y_ld % 4 != 0 || !al(p.y) || !al(p.scale) || !al(p.shift)) return false;
    if (p.res && (p.res_ld % 4 != 0 || !al(p.res))) return false;
    if (p.posb && !al(p.posb)) return false;
    if (p.part && !al(p.part)) return false;
    return true;
}


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

}  // namespace
