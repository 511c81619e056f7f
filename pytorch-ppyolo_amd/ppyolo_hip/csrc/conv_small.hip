// f16x2 implicit-GEMM convolution for SMALL outputs (round 6): batch 1 -- the reference's demo loop, demo.py:121-160 -- and the narrow
// layers of any batch (conv_offset K = 27, the 19x19 head tails).
//
// At M = N * Ho * Wo = 361 .. 5776 rows the tiles of conv_x3.hip / conv_ws.hip put 6 .. 90 row tiles on 256 CUs: the tables there pick
// split-K 2 .. 16 to fill the chip, i.e. a second launch (the deterministic combine, 6-8 us) and partial sums through memory for layers
// that take 10-25 us in all.  Same operator, same f16x2 arithmetic, same products in the same order as those tiles -- but organised
// around the WAVE:
//   * a wave owns a 32 x 32 (or 32 x 64) output tile and one k-part of the reduction; nothing is shared through the LDS in the main
//     loop and there is no workgroup barrier in it: both operands go from the L2 straight into MFMA fragment registers (the
//     activations split in registers, or read as finished operands when the producer stored them pre-split), DEPTH k-steps requested
//     ahead (16 KB .. 24 KB in flight per wave);
//   * the waves of a workgroup (four or eight) are `ks` k-parts of 4 / ks or 8 / ks neighbouring column tiles of the same rows (the
//     rows come from the L1 after the first wave): a split-K INSIDE the launch -- the k-parts' accumulators meet in the LDS and are
//     added in the fixed order 0, 1, 2, ... by the wave of part 0, which then runs the shared epilogue (conv_shared.h).  No workspace, no
//     second launch; with ks in {1, 2, 4, 8} the result is BIT-IDENTICAL to the tile kernels' split-K of ks (same chunk ranges, same
//     order of additions: a power-of-two row scale commutes with the sum).
// 361 rows x 512 channels are 192 wave tiles; with ks = 4 that is 768 waves on 1024 SIMDs where the 64x128 tile had 24 workgroups x split 6-8.
#include "conv_shared.h"

namespace {

constexpr unsigned SM_OOB = 0x80000000u;      // = num_records of every resource here (tensors < 2 GB): loads give 0

template <int TN, bool GP, bool VEC>
__global__ void __launch_bounds__(512) conv_small_kernel(const ConvArgs p, const int ks) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int DEPTH = TN == 1 ? 8 : 6;      // k-steps (16 deep) in flight per wave: 4 / 6 sixteen-byte loads per lane each
    static_assert(DEPTH % 2 == 0, "a chunk is two k-steps");
    extern __shared__ __attribute__((aligned(16))) char smem_sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwave = (int)blockDim.x >> 6;
    const int tpw = nwave / ks;                                   // wave tiles per workgroup
    const int tslot = wave / ks, kpart = wave - tslot * ks;
    const int tiles_n = (p.K + 32 * TN - 1) / (32 * TN), tiles_m = (p.M + 31) >> 5;
    int blk;
    {   // XCD-contiguous order of the workgroups (conv_x3.hip): the column tiles of a row tile share an L2
        const int nb = (int)gridDim.x, q = nb >> 3, r = nb & 7;
        const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
        blk = xcd * q + min(xcd, r) + idx;
    }
    const int tile = blk * tpw + tslot;
    const bool live = tile < tiles_m * tiles_n;
    const int tile_m = live ? tile / tiles_n : 0, tile_n = live ? tile - tile_m * tiles_n : 0;
    const int m0 = tile_m * 32, n0 = tile_n * 32 * TN;
    const int kc_begin = kpart * p.chunks_per_split;
    const int kc_end = min(kc_begin + p.chunks_per_split, p.chunks_total);
    const int nsteps = live ? 2 * max(kc_end - kc_begin, 0) : 0;
    const int hw = p.Ho * p.Wo;
    const int frow = lane & 31, fkh = lane >> 5;

    // ---- this lane's row of the A fragments (row frow of the tile, k-half fkh): byte offset of its pixel's tap (0, 0), validity per tap
    const long long bias = (long long)(p.pad * p.W + p.pad) * p.x_ld * 4;      // keeps offsets >= 0
    unsigned a_off, a_ok = 0u;
    int n_img;
    {
        const int m = m0 + frow, mc = min(m, p.M - 1);
        n_img = mc / hw;
        const int rem = mc - n_img * hw;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_off = (unsigned)((((long long)n_img * p.H + hi0) * p.W + wi0) * p.x_ld * 4 + bias + fkh * (GP ? 16 : 32));
        unsigned colmask = 0;
        for (int s2 = 0; s2 < p.S; ++s2)
            if ((unsigned)(wi0 + s2) < (unsigned)p.W) colmask |= 1u << s2;
        for (int r = 0; r < p.R; ++r)
            if ((unsigned)(hi0 + r) < (unsigned)p.H) a_ok |= colmask << (r * p.S);
        if (m >= p.M || !live) a_ok = 0u;
    }
    // B fragments: column frow of column tile j, k-half fkh, of the [plane][chunk][K][32] planes (ppy_conv2d_split_weights_f16x2)
    unsigned b_off[2][TN];
    {
        const long long plane_bytes = (long long)p.K * p.Kred * 2;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int k = min(n0 + j * 32 + frow, p.K - 1);          // columns >= K are masked at store
                b_off[pl][j] = (unsigned)(pl * plane_bytes + (long long)k * 64 + fkh * 16);
            }
    }
    // per-image activation scale of this lane's row (conv_x3.hip)
    float sa, inv_sa, xmax_up = 1.0f;
    if constexpr (GP) {
        sa = p.xscale[n_img];
        inv_sa = pow2_inverse(sa);
        if (p.yscale) xmax_up = pow2_above(conv_amax_in(p, n_img));
    } else {
        const float mx = conv_amax_in(p, n_img);
        const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
        int f = 267 - e;
        f = f < 103 ? 103 : (f > 167 ? 167 : f);
        sa = __uint_as_float((unsigned)f << 23);
        inv_sa = __uint_as_float((unsigned)(254 - f) << 23);
        xmax_up = 16384.0f * inv_sa;
    }

    struct Stage {               // what one k-step needs
        uintx4 a[2];             // fp32 input: the 8 values of this lane (32 bytes); pre-split input: first terms, second terms
        uintx4 b[2][TN];
    };
    Stage ring[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
        ring[u].a[0] = ring[u].a[1] = uintx4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) ring[u].b[pl][j] = uintx4{0u, 0u, 0u, 0u};
    }
    // request cursor: chunk kc = (channel chunk cc, tap), cc outer, tap inner (conv_x3.hip's order)
    const int RS = p.R * p.S;
    int i_cc = kc_begin / RS, i_tap = kc_begin - i_cc * RS;
    int i_r = i_tap / p.S, i_s = i_tap - i_r * p.S;
    const char *xb = reinterpret_cast<const char *>(p.x) - bias;
    const char *wb = reinterpret_cast<const char *>(p.wf16);
    // the requests of k-step `g` of this wave's range into `st` (beyond the range: out-of-range offsets, no memory traffic); `half` = g & 1
    auto request = [&](Stage &st, int g, const int half) {
        const bool have = g < nsteps;
        const long long a_uni = ((long long)(i_r * p.W + i_s) * p.x_ld + i_cc * 32) * 4;
        const long long b_uni = ((long long)i_tap * (p.C / 32) + i_cc) * p.K * 64;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(xb + (have ? a_uni : 0)), 0, SM_OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(wb + (have ? b_uni : 0)), 0, SM_OOB, 0x00020000);
        const unsigned ao = (have && ((a_ok >> i_tap) & 1u)) ? a_off : SM_OOB;
        if constexpr (GP) {      // a pixel's 32-channel group: 32 fp16 first terms, then 32 fp16 second terms
            st.a[0] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(ao + half * 32), 0, 0);
            st.a[1] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(ao + 64 + half * 32), 0, 0);
        } else {
            st.a[0] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(ao + half * 64), 0, 0);
            st.a[1] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(ao + half * 64 + 16), 0, 0);
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                st.b[pl][j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)((have ? b_off[pl][j] : SM_OOB) + half * 32), 0, 0);
        if (half) {              // the chunk is requested: next tap / channel chunk
            ++i_tap;
            ++i_s;
            if (i_s == p.S) { i_s = 0; ++i_r; }
            if (i_tap == RS) { i_tap = 0; i_r = 0; i_s = 0; ++i_cc; }
        }
    };

    floatx16 acc[1][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;
    auto multiply = [&](const Stage &st) {
        uintx4 a0, a1;
        if constexpr (GP) {
            a0 = st.a[0];
            a1 = st.a[1];
        } else {                 // scale, two fp16 terms; the residual fma(x, s, -first) is exact (conv_x3.hip)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xa = __uint_as_float(q < 2 ? st.a[0][2 * q] : st.a[1][2 * q - 4]);
                const float xb2 = __uint_as_float(q < 2 ? st.a[0][2 * q + 1] : st.a[1][2 * q - 3]);
                const unsigned P0 = cvt_pk_f16(xa * sa, xb2 * sa);
                a0[q] = P0;
                a1[q] = cvt_pk_f16(fmaf(xa, sa, -f16_lo(P0)), fmaf(xb2, sa, -f16_hi(P0)));
            }
        }
        // the three leading products, smallest first, as the tiles order them: a1*b0, a0*b1, a0*b0
#pragma unroll
        for (int j = 0; j < TN; ++j)
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, st.b[0][j]), acc[0][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, st.b[1][j]), acc[0][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, st.b[0][j]), acc[0][j], 0, 0, 0);
    };
    // ONE code path for requests and waits (conv_stream.hip): the first pass over the ring multiplies nothing and only requests
    for (int g0 = -DEPTH; g0 < nsteps; g0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int g = g0 + u;
            if (g >= 0 && g < nsteps) multiply(ring[u]);
            request(ring[u], g + DEPTH, u & 1);
        }
    }

    // ---- the k-parts of a tile meet: [wave][column tile][register][lane] floats behind the transposition patches
    if (ks > 1) {
        float *xch = reinterpret_cast<float *>(smem_sm + nwave * (32 * LDS_LD * 4));
        if (kpart > 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) xch[((wave * TN + j) * 16 + e) * 64 + lane] = acc[0][j][e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kpart == 0) {
            for (int q = 1; q < ks; ++q) {
                if (q * p.chunks_per_split >= p.chunks_total) break;      // (an empty part: nothing to add)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[0][j][e] += xch[(((wave + q) * TN + j) * 16 + e) * 64 + lane];
            }
        }
    }
    if (!live || kpart != 0) return;

    float rowscale[1][4], rowsplit[1][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) rowsplit[0][t] = 0.f;
    bool split_out = false;
    if constexpr (VEC) {         // the output goes to one consumer as finished operands (conv_x3.hip, ConvArgs::yscale)
        split_out = p.yscale != nullptr;
        if (split_out) {
            const float ys = split_scale_of(fmaf(p.ysplit_mul, xmax_up, p.ysplit_add));
            const int mr = m0 + frow;
            if (n0 == 0 && lane < 32 && mr < p.M) p.yscale[mr / hw] = ys;
#pragma unroll
            for (int t = 0; t < 4; ++t) rowsplit[0][t] = __shfl(ys, (lane >> 3) + 8 * t);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) rowscale[0][t] = __shfl(inv_sa, (lane >> 3) + 8 * t);
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float inv = __shfl(inv_sa, (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5));
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[0][j][e] *= inv;
        }
    }
    tile_epilogue<1, TN, 32, 32 * TN, false, VEC>(p, acc, reinterpret_cast<float *>(smem_sm), m0, n0, 0, 0, lane, wave, 0,
                                                   VEC ? rowscale : nullptr, VEC ? rowsplit : nullptr, split_out);
#endif
}

template <int TN, bool GP, bool VEC>
int launch_small_one(const ConvArgs &p, int nwave, int ks, hipStream_t stream) {
    auto k = conv_small_kernel<TN, GP, VEC>;
    const size_t lds = (size_t)nwave * 32 * LDS_LD * sizeof(float) + (ks > 1 ? (size_t)nwave * TN * 16 * 64 * sizeof(float) : 0);
    static PpyLdsAttr attr;
    if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), 128 * 1024) != PPY_OK) return PPY_ERR_LAUNCH;
    const int tiles = ceil_div(p.M, 32) * ceil_div(p.K, 32 * TN), tpw = nwave / ks;
    hipLaunchKernelGGL(k, dim3(ceil_div(tiles, tpw)), dim3(64 * nwave), lds, stream, p, ks);
    return ppy_launch_status();
}

template <int TN>
int launch_small(ConvArgs p, int nwave, int splits, hipStream_t stream) {
    const long long xbytes = (long long)p.N * p.H * p.W * p.x_ld * 4 + (long long)(p.pad * p.W + p.pad) * p.x_ld * 4;
    const long long wbytes = (long long)p.K * p.Kred * 2 * 2;
    if (xbytes >= 0x7FFFF000LL || wbytes >= 0x7FFFF000LL || p.R * p.S > 32) return PPY_ERR_UNSUPPORTED;
    if (p.bn_part) return PPY_ERR_UNSUPPORTED;
    p.chunks_total = p.R * p.S * (p.C / 32);
    // `splits` k-parts INSIDE the workgroup: a power of two <= the waves of a workgroup (anything else is rounded down), none empty
    int ks = 1;
    while (2 * ks <= splits && 2 * ks <= nwave && 2 * ks <= p.chunks_total) ks *= 2;
    p.chunks_per_split = ceil_div(p.chunks_total, ks);
    const bool vec = vec_epilogue_ok(p);
    if ((p.xscale || p.yscale) && !vec) return PPY_ERR_BAD_ARG;
    if (p.yscale && p.ups) return PPY_ERR_BAD_ARG;
    if (p.yscale && (p.K % 32 != 0 || p.y_ld % 32 != 0)) return PPY_ERR_BAD_ARG;
    if (p.xscale) return vec ? launch_small_one<TN, true, true>(p, nwave, ks, stream) : PPY_ERR_BAD_ARG;
    return vec ? launch_small_one<TN, false, true>(p, nwave, ks, stream) : launch_small_one<TN, false, false>(p, nwave, ks, stream);
}

}  // namespace

// local ids: 0 = 32x32 wave tiles, four waves per workgroup; 1 = 32x64, four; 2 = 32x32, eight; 3 = 32x64, eight
int ppy_small_num_configs() { return 4; }

int ppy_small_dispatch(const ConvArgs &p, int c, int s, hipStream_t st) {
    if (!p.wf16 || ((uintptr_t)p.wf16 & 15) != 0 || !p.scale_f16 || !p.amax_in || (p.posb && !p.posb_f16)) return PPY_ERR_BAD_ARG;
    ConvArgs q = p;
    q.scale = p.scale_f16;
    q.posb = p.posb ? p.posb_f16 : nullptr;
    q.part = nullptr;
    switch (c) {
        case 0: return launch_small<1>(q, 4, s, st);
        case 1: return launch_small<2>(q, 4, s, st);
        case 2: return launch_small<1>(q, 8, s, st);
        case 3: return launch_small<2>(q, 8, s, st);
    }
    return PPY_ERR_BAD_ARG;
}
