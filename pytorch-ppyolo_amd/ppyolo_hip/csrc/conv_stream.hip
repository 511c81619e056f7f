// 1x1 convolutions with a short reduction (C = 64) as a STREAM: the HBM-bound "expand" layers of ResNet-vd stage 2
// (reference model/resnet_vd.py:55-91: conv3 of a BottleNeck, C64 -> K256 at 152x152, + shortcut + ReLU).
//
// Same operator, same f16x2 arithmetic and the same order of operations as conv_x3.hip (results are bit-identical to
// its tiles), organised for bytes instead of MFMA rate -- the layer moves 425 MB for 6 GFLOP:
//   * persistent waves, no workgroup barrier anywhere: a wave owns 64 output channels, keeps their weights (both fp16
//     planes, all of C) in 64 VGPRs for its whole life and walks over 32-pixel tiles with a grid stride; the four waves of a
//     workgroup take the four channel slices of the same pixels, so the input rows come from HBM once and from the L1 after;
//   * every tile's reads are requested one tile ahead, straight into registers: the activations in the MFMA fragment layout
//     (no LDS), the shortcut rows in the layout of the vector epilogue -- requests of tile t+1 are in flight while tile t
//     is multiplied, transposed (wave-private LDS patch) and stored, so reads and writes overlap inside every wave
//     instead of alternating between the two workgroups of a CU;
//   * POOL: the tile's 32 rows are 8 blocks of 2x2 pixels (row r = block r&7, position r>>3), which puts the four pixels of
//     a block into ONE lane after the transposition: the vd shortcut's AvgPool2d(2, 2) (reference model/resnet_vd.py:29-33)
//     of this output is written from the epilogue, (((a + b) + c) + d) * 0.25 as stem_pool.hip does, and the separate
//     pooling launch disappears from the plan.
#include "conv_shared.h"
#pragma clang fp contract(off)

namespace {

constexpr unsigned ST_OOB = 0x80000000u;      // beyond any tensor this kernel accepts (< 2 GB): loads give 0, stores are dropped

struct StreamArgs {
    ConvArgs c;
    float *pool;      // POOL: [N][H/2][W/2][pool_ld], else NULL
    int pool_ld;
    int tiles;        // 32-row tiles
};

// ALDS = false, TN = 2: the C = 64 form above (activations straight into fragment registers, waves independent, two workgroups
// per CU).  ALDS = true, TN = 1 (C = 128, e.g. the stage-3 expand layers C128 -> K512 at 76x76): a wave owns 32 channels (64
// VGPRs of weights again), a workgroup 128, and since many slices re-reading the activations through the L1 would saturate
// its address path, the tile's [32][C] activations are staged ONCE per workgroup through two alternating LDS buffers
// (coalesced loads one tile ahead -> ds_write -> ONE workgroup barrier per tile -> swizzled fragment reads); the K / 128
// workgroups of a pixel stream sit on one XCD (block ids 8 apart), so the tile comes from HBM once.  (Three workgroups per CU
// would need <= 168 VGPRs: 20-25 spilled, and scratch reloads queue behind the prefetches in the in-order vmcnt.)
// BNA: BatchNorm on batch statistics applied in the epilogue (ConvArgs::bn_mean ...; training forward of frozen layers).
// NOSTORE (with BNS): statistics only, nothing is stored.
template <int KS, int TN, bool RES, bool POOL, bool ALDS, bool BNS = false, bool BNA = false, bool NOSTORE = false>      // (BNS: BatchNorm statistics from the epilogue, conv_x3.hip)
__global__ void __launch_bounds__(256, 2) conv1x1_stream_kernel(const StreamArgs q) {
    static_assert(!NOSTORE || (BNS && !RES && !POOL && !BNA), "statistics-only launches");
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs &p = q.c;
    extern __shared__ __attribute__((aligned(16))) char smem_st[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *sE = reinterpret_cast<float *>(smem_st) + wave * (32 * LDS_LD);
    constexpr int NW = 4, SW = 32 * TN;                         // waves per workgroup, channels per wave
    constexpr int C = KS * 16, ROW_BYTES = C * 4;
    char *abuf = smem_st + NW * 32 * LDS_LD * 4;                // ALDS: two [32][C] fp32 tiles
    const int nsl = p.K / SW;                                   // channel slices
    const int gw = (int)blockIdx.x * NW + wave;
    const int groups = ALDS ? nsl / NW : 1;                     // ALDS: workgroups that share the pixels of a stream
    const int bi = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int slice = ALDS ? (bi % groups) * NW + wave : gw % nsl;
    const int stream = ALDS ? (bi / groups) * 8 + xcd : gw / nsl;
    const int nstreams = ALDS ? (int)gridDim.x / groups : ((int)gridDim.x * 4) / nsl;
    const int n0 = slice * SW;
    const int hw = p.H * p.W;
    // a tile counts 32 pixels, or (POOL) 8 blocks of 2x2 pixels
    const int unit_hw = POOL ? hw >> 2 : hw, units = POOL ? p.M >> 2 : p.M, per_tile = POOL ? 8 : 32;
    const int Wq = p.W >> 1;
    auto pixel_of = [&](int idx, int sub) -> int {
        if constexpr (!POOL) return idx;
        const int n = idx / unit_hw, rem = idx - n * unit_hw;
        const int ph = rem / Wq, pw = rem - ph * Wq;
        return n * hw + (2 * ph + (sub >> 1)) * p.W + 2 * pw + (sub & 1);
    };
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, ST_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)(RES ? p.res : p.x), 0, ST_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, ST_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(POOL ? q.pool : p.y), 0, ST_OOB, 0x00020000);

    // ---- this wave's weights: B fragments of v_mfma_f32_32x32x16_f16 (column lane&31, k = 16s + 8(lane>>5) + [0,8)) from the
    // [plane][chunk][K][32] planes of ppy_conv2d_split_weights_f16x2
    uintx4 wf[2][TN][KS];
    {
        const char *wb = reinterpret_cast<const char *>(p.wf16);
        const long long plane_bytes = (long long)p.K * p.C * 2;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int k = n0 + j * 32 + (lane & 31);
                    const long long o = pl * plane_bytes + (((long long)(s >> 1) * p.K + k) * 32 + (s & 1) * 16 + 8 * (lane >> 5)) * 2;
                    wf[pl][j][s] = *reinterpret_cast<const uintx4 *>(wb + o);
                }
    }
    const int erow = lane >> 3, ec4 = (lane & 7) * 4;
    floatx4 sc[TN], sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        sc[j] = *reinterpret_cast<const floatx4 *>(p.scale + n0 + j * 32 + ec4);
        sh[j] = *reinterpret_cast<const floatx4 *>(p.shift + n0 + j * 32 + ec4);
    }

    floatx4 bmu[BNA ? TN : 1], bisg[BNA ? TN : 1], bbe[BNA ? TN : 1];
    if constexpr (BNA) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = n0 + j * 32 + ec4;
            bmu[j] = *reinterpret_cast<const floatx4 *>(p.bn_mean + c);
            bbe[j] = *reinterpret_cast<const floatx4 *>(p.bn_beta + c);
            const floatx4 is = *reinterpret_cast<const floatx4 *>(p.bn_invstd + c), ga = *reinterpret_cast<const floatx4 *>(p.bn_gamma + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) bisg[j][k] = is[k] * ga[k];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bmu[j]), "+v"(bisg[j]), "+v"(bbe[j]));
    }

    // ---- requests of a tile: activations in the A-fragment layout, shortcut rows in the epilogue layout
    // ALDS: this wave's share of the tile, 16 bytes per lane, LPR lanes per pixel row
    constexpr int LPR = C / 4, RPI = 64 / LPR, NI = ALDS ? (32 / NW) / RPI : 1;
    uintx4 stg[NI];
    auto stage_row = [&](int i) { return wave * (32 / NW) + i * RPI + lane / LPR; };
    uintx4 raw[ALDS ? 1 : 2 * KS];
    auto request_a = [&](int t) {
        if constexpr (ALDS) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = stage_row(i);
                const int idx = POOL ? t * 8 + (row & 7) : t * 32 + row;
                const bool ok = t >= 0 && t < q.tiles && idx < units;
                const unsigned off = ok ? (unsigned)pixel_of(idx, row >> 3) * (unsigned)(p.x_ld * 4) + (unsigned)(lane % LPR) * 16u : ST_OOB;
                stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0);
            }
            return;
        }
        const int fr = lane & 31;
        const int idx = POOL ? t * 8 + (fr & 7) : t * 32 + fr;
        const bool ok = t >= 0 && t < q.tiles && idx < units;
        const unsigned off = ok ? (unsigned)pixel_of(idx, fr >> 3) * (unsigned)(p.x_ld * 4) + (unsigned)(lane >> 5) * 32u : ST_OOB;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            raw[2 * s] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(off + s * 64), 0, 0);
            raw[2 * s + 1] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(off + s * 64 + 16), 0, 0);
        }
    };
    // pixel of epilogue row erow + 8u of tile t (u = 0..3), as a row number of y / res (or -1)
    auto epi_pixels = [&](int t, int (&pix)[4], int &idx0) {
        if constexpr (POOL) {
            idx0 = t * 8 + erow;
            const bool ok = t >= 0 && t < q.tiles && idx0 < units;
            const int base = pixel_of(idx0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) pix[u] = ok ? base + (u >> 1) * p.W + (u & 1) : -1;
        } else {
            idx0 = t * 32 + erow;
#pragma unroll
            for (int u = 0; u < 4; ++u) pix[u] = (t >= 0 && t < q.tiles && idx0 + 8 * u < units) ? idx0 + 8 * u : -1;
        }
    };
    uintx4 rv[TN][4];
    auto request_res = [&](int j, const int (&pix)[4]) {
        if constexpr (RES) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned off = pix[u] >= 0 ? (unsigned)pix[u] * (unsigned)(p.res_ld * 4) + (unsigned)(n0 + j * 32 + ec4) * 4u : ST_OOB;
                rv[j][u] = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)off, 0, 0);
            }
        }
    };

    // ---- per-image activation scale (conv_x3.hip: the power of two that puts the tracked maximum into [2^13, 2^14)) of the
    // (at most two: the host checks unit_hw >= per_tile) images a tile touches; re-read only when the first image changes
    int sc_n = -1;
    float sa0 = 1.f, sa1 = 1.f, inv0 = 1.f, inv1 = 1.f;
    auto scale_of = [&](int n, float &s, float &inv) {
        const float mx = conv_amax_in(p, min(n, p.N - 1));
        const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
        int f = 267 - e;
        f = f < 103 ? 103 : (f > 167 ? 167 : f);
        s = __uint_as_float((unsigned)f << 23);
        inv = __uint_as_float((unsigned)(254 - f) << 23);
    };
    // running maximum of |y| for p.amax_out: `run_mx` belongs to image run_n
    int run_n = -1;
    float run_mx = 0.f;
    auto flush = [&](float mx, int n) {
        if (p.amax_out && n >= 0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (lane == 0) amax_store(mx, p.amax_out, n, gw);
        }
    };

    // ONE code path for requests and waits: the loop starts one stride before the wave's first tile with an iteration that
    // stores nothing (all rows invalid) and only requests -- with a separate prologue the compiler's counted waits are the
    // minimum over both paths into the loop head, and the steady state then drains the queue of stores at every tile.
#pragma unroll
    for (int i = 0; i < (ALDS ? 1 : 2 * KS); ++i) raw[i] = uintx4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NI; ++i) stg[i] = uintx4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) rv[j][u] = uintx4{0u, 0u, 0u, 0u};
    // (and nothing may be pending at the loop head on the entry edge either: a use of the weight / scale registers here
    // makes the compiler wait for their loads in front of the loop instead of inside it)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wf[pl][j][s]));
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(sc[j]), "+v"(sh[j]));
    int pix_n[4] = {-1, -1, -1, -1}, idx_n = 0;
    const float slope = p.act == PPY_ACT_RELU ? 0.f : (p.act == PPY_ACT_LEAKY ? 0.1f : 1.f);
    int par = 0;
    for (int t = stream - nstreams; t < q.tiles; t += nstreams) {
        if constexpr (ALDS) {
            // the tile requested one iteration ago goes to the buffer last read two iterations ago: one barrier per tile
            par ^= 1;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = stage_row(i);
                *reinterpret_cast<uintx4 *>(abuf + par * (32 * ROW_BYTES) + row * ROW_BYTES + (((lane % LPR) ^ (row & 15)) << 4)) = stg[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            request_a(t + nstreams);
        }
        int pix[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pix[u] = pix_n[u];
        const int idx0 = idx_n;
        const int first = max(t, 0) * per_tile;
        const int n_lo = __builtin_amdgcn_readfirstlane(first / unit_hw);
        const int bnd = (n_lo + 1) * unit_hw;                              // first index of the next image
        const bool two = min(first + per_tile, units) > bnd;
        if (n_lo != sc_n) {
            scale_of(n_lo, sa0, inv0);
            scale_of(n_lo + 1, sa1, inv1);
            sc_n = n_lo;
        }
        if (n_lo != run_n) {
            flush(run_mx, run_n);
            run_n = n_lo;
            run_mx = 0.f;
        }
        float hi_mx = 0.f;
        const int fidx = POOL ? first + (lane & 7) : first + (lane & 31);
        const float sa = fidx < bnd ? sa0 : sa1;

        floatx16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uintx4 a0, a1, r2[2];
            if constexpr (ALDS) {       // 16-byte slot c of row r sits at c ^ (r & 15): the 16 rows of a quarter-wave hit 16 bank groups
                const int fr = lane & 31, c0 = 4 * s + 2 * (lane >> 5);
                const char *ab = abuf + par * (32 * ROW_BYTES) + fr * ROW_BYTES;
                r2[0] = *reinterpret_cast<const uintx4 *>(ab + ((c0 ^ (fr & 15)) << 4));
                r2[1] = *reinterpret_cast<const uintx4 *>(ab + (((c0 + 1) ^ (fr & 15)) << 4));
            } else {
                r2[0] = raw[ALDS ? 0 : 2 * s];
                r2[1] = raw[ALDS ? 0 : 2 * s + 1];
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float xa = __uint_as_float(r2[q4 >> 1][(q4 & 1) * 2]);
                const float xb = __uint_as_float(r2[q4 >> 1][(q4 & 1) * 2 + 1]);
                const unsigned P0 = cvt_pk_f16(xa * sa, xb * sa);
                a0[q4] = P0;
                a1[q4] = cvt_pk_f16(fmaf(xa, sa, -f16_lo(P0)), fmaf(xb, sa, -f16_hi(P0)));
            }
            // the three leading products, smallest first, as conv_x3.hip orders them: a1*b0, a0*b1, a0*b0
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, wf[0][j][s]), acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, wf[1][j][s]), acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, wf[0][j][s]), acc[j], 0, 0, 0);
        }
        if constexpr (BNS) {       // training forward: the BatchNorm's first pass from here (conv_shared.h), one slice per 32-pixel tile
            if (t >= 0) {
                const float inv_f[1] = {fidx < bnd ? inv0 : inv1};
                tile_bn_stats<1, TN, 32, SW>(p, reinterpret_cast<const floatx16(&)[1][TN]>(acc), inv_f, t * 32, n0, 0, 0, lane, t);
            }
        }
        // the activations of the next tile of this wave
        if constexpr (!ALDS) request_a(t + nstreams);
        epi_pixels(t + nstreams, pix_n, idx_n);

        float rowscale[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rowscale[u] = (POOL ? idx0 : idx0 + 8 * u) < bnd ? inv0 : inv1;
#pragma unroll
        for (int j = 0; j < (NOSTORE ? 0 : TN); ++j) {
            const int col = n0 + j * 32 + ec4;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                sE[row * LDS_LD + (lane & 31)] = acc[j][e];
            }
            __builtin_amdgcn_wave_barrier();
            floatx4 v[4];
            const floatx4 scj = sc[j], shj = sh[j];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = *reinterpret_cast<const floatx4 *>(sE + (erow + 8 * u) * LDS_LD + ec4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float o = fmaf(v[u][c] * rowscale[u], scj[c], shj[c]);
                    if constexpr (BNA) o = fmaf(o - bmu[j][c], bisg[j][c], bbe[j][c]);      // (bn_apply_kernel's sub, mul, fma, add)
                    if (RES) o += __uint_as_float(rv[j][u][c]);
                    v[u][c] = o > 0.f ? o : o * slope + 0.0f;       // (+0: ReLU gives +0 for negative inputs, as max(o, 0) does)
                }
                const float rmx = pix[u] >= 0 ? fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))) : 0.f;
                const bool lo = (POOL ? idx0 : idx0 + 8 * u) < bnd;
                run_mx = fmaxf(run_mx, lo ? rmx : 0.f);
                hi_mx = fmaxf(hi_mx, lo ? 0.f : rmx);
                const unsigned off = pix[u] >= 0 ? (unsigned)pix[u] * (unsigned)(p.y_ld * 4) + (unsigned)col * 4u : ST_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v[u]), ry, (int)off, 0, 0);
            }
            if constexpr (POOL) {
                floatx4 r;
#pragma unroll
                for (int c = 0; c < 4; ++c) r[c] = (((v[0][c] + v[1][c]) + v[2][c]) + v[3][c]) * 0.25f;
                const unsigned off = pix[0] >= 0 ? (unsigned)idx0 * (unsigned)(q.pool_ld * 4) + (unsigned)col * 4u : ST_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, r), rp, (int)off, 0, 0);
            }
            request_res(j, pix_n);
            __builtin_amdgcn_wave_barrier();
        }
        if (two) {
            flush(run_mx, run_n);
            run_n = n_lo + 1;
            run_mx = hi_mx;
        }
    }
    flush(run_mx, run_n);
#endif
}

template <int KS, int TN, bool RES, bool POOL, bool ALDS>
int launch_stream_one(const StreamArgs &q, int grid, hipStream_t stream) {
    auto k = conv1x1_stream_kernel<KS, TN, RES, POOL, ALDS>;
    const size_t lds = (size_t)4 * 32 * LDS_LD * sizeof(float) + (ALDS ? 2 * 32 * KS * 16 * sizeof(float) : 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, stream, q);
    return ppy_launch_status();
}

template <int KS, int TN, bool ALDS>
int launch_stream(const StreamArgs &q, int grid, hipStream_t stream) {
    const size_t lds = (size_t)4 * 32 * LDS_LD * sizeof(float) + (ALDS ? 2 * 32 * KS * 16 * sizeof(float) : 0);
    if (q.c.bn_part) {       // (the dispatcher has checked: no pooled output, no shortcut)
        if (q.c.bn_nostore)
            hipLaunchKernelGGL((conv1x1_stream_kernel<KS, TN, false, false, ALDS, true, false, true>), dim3(grid), dim3(256), lds, stream, q);
        else
            hipLaunchKernelGGL((conv1x1_stream_kernel<KS, TN, false, false, ALDS, true>), dim3(grid), dim3(256), lds, stream, q);
        return ppy_launch_status();
    }
    if (q.c.bn_mean) {       // (the dispatcher has checked: no pooled output)
        if (q.c.res)
            hipLaunchKernelGGL((conv1x1_stream_kernel<KS, TN, true, false, ALDS, false, true>), dim3(grid), dim3(256), lds, stream, q);
        else
            hipLaunchKernelGGL((conv1x1_stream_kernel<KS, TN, false, false, ALDS, false, true>), dim3(grid), dim3(256), lds, stream, q);
        return ppy_launch_status();
    }
    if (q.pool) return q.c.res ? launch_stream_one<KS, TN, true, true, ALDS>(q, grid, stream) : launch_stream_one<KS, TN, false, true, ALDS>(q, grid, stream);
    return q.c.res ? launch_stream_one<KS, TN, true, false, ALDS>(q, grid, stream) : launch_stream_one<KS, TN, false, false, ALDS>(q, grid, stream);
}

}  // namespace

int ppy_stream_num_configs() { return 2; }

// C = 64 (K % 64 == 0): local 0 = 512 workgroups of four waves (two per CU), local 1 = 256.  C = 128 (K % 128 == 0, K / 128 a
// power of two): the same grids.  `pool` / `pool_ld`: optional 2x2 average of y.
int ppy_stream_dispatch(const ConvArgs &p, int local, float *pool, int pool_ld, hipStream_t stream) {
    if (local < 0 || local >= ppy_stream_num_configs()) return PPY_ERR_BAD_ARG;
    // BAD_ARG, not UNSUPPORTED: an explicit id that does not apply is the caller's error (no silent other kernel)
    if (p.R != 1 || p.S != 1 || p.stride != 1 || p.pad != 0 || p.ups || p.posb) return PPY_ERR_BAD_ARG;
    if (!((p.C == 64 && p.K % 64 == 0) || (p.C == 128 && p.K % 128 == 0))) return PPY_ERR_BAD_ARG;
    if (!p.wf16 || ((uintptr_t)p.wf16 & 15) != 0 || !p.scale_f16 || !p.amax_in) return PPY_ERR_BAD_ARG;
    if (!vec_epilogue_ok(p) || ((uintptr_t)p.x & 15) != 0 || p.x_ld % 4 != 0) return PPY_ERR_BAD_ARG;
    const int nsl = p.C == 64 ? p.K / 64 : p.K / 128;       // C = 64: slices of 64 channels per wave; C = 128: workgroups per pixel stream
    if (nsl > 16 || (nsl & (nsl - 1)) != 0) return PPY_ERR_BAD_ARG;           // the slices / groups divide the wave / workgroup count
    const int hw = p.H * p.W;
    if (pool) {
        if (p.H % 2 != 0 || p.W % 2 != 0 || hw / 4 < 8 || pool_ld < p.K || pool_ld % 4 != 0 || ((uintptr_t)pool & 15) != 0) return PPY_ERR_BAD_ARG;
    } else if (hw < 32) {
        return PPY_ERR_BAD_ARG;
    }
    const long long lim = 0x7FFFF000LL;
    if ((long long)p.M * p.x_ld * 4 >= lim || (long long)p.M * p.y_ld * 4 >= lim || (p.res && (long long)p.M * p.res_ld * 4 >= lim))
        return PPY_ERR_UNSUPPORTED;
    if (p.bn_nostore && !p.bn_part) return PPY_ERR_BAD_ARG;
    if (p.bn_mean) {         // BatchNorm applied in the epilogue: all four vectors, 16-byte aligned; not combined with the statistics pass
        if (!p.bn_invstd || !p.bn_gamma || !p.bn_beta || p.bn_part || pool) return PPY_ERR_BAD_ARG;
        if ((((uintptr_t)p.bn_mean | (uintptr_t)p.bn_invstd | (uintptr_t)p.bn_gamma | (uintptr_t)p.bn_beta) & 15) != 0) return PPY_ERR_BAD_ARG;
    }
    if (p.bn_part) {         // BatchNorm statistics from the epilogue: plain conv + bias, one slice per 32-pixel tile
        if (pool || p.res || p.act != PPY_ACT_NONE) return PPY_ERR_UNSUPPORTED;
        if (ceil_div(p.M, 32) > p.bn_capacity) return PPY_ERR_WORKSPACE;
        if (p.bn_slices_host) *p.bn_slices_host = ceil_div(p.M, 32);
    }
    StreamArgs q;
    q.c = p;
    q.c.scale = p.scale_f16;
    q.pool = pool;
    q.pool_ld = pool_ld;
    q.tiles = pool ? ceil_div(p.M / 4, 8) : ceil_div(p.M, 32);
    if (p.C == 128) return launch_stream<8, 1, true>(q, local == 0 ? 512 : 256, stream);
    return launch_stream<4, 2, false>(q, local == 0 ? 512 : 256, stream);
}
