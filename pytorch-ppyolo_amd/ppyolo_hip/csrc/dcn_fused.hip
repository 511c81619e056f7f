// DCNv2 with the bilinear gather fused into the contraction (reference model/custom_layers.py:551-677).
//
// y[m, k] = act(scale[k] * sum_{tap, c} sample(m, tap, c) * w[k, tap, c] + shift[k]),  m = (n, ho, wo)
// sample(m, tap, c) = sigmoid(mask) * bilinear(x, position(m, tap) + learned offset)[c]        (:565-662)
//
// There is no "columns" matrix: a workgroup owns a BM x BN output tile and walks the reduction in 32-channel chunks
// (tap-major, as the [K][3][3][C] weights are laid out).  Per chunk
//   * the 256 threads BUILD the [BM][32] A tile of sampled values: a thread owns one tile row and 8 or 16 consecutive
//     channels; sampling position, the 4 corner offsets, the 4 bilinear weights and the mask are computed once per
//     (row, tap) and reused for the C/32 chunks of that tap; the corner pixels are fetched with 16-byte buffer loads
//     (out-of-image corners are out-of-range offsets: the buffer unit returns zeros, no branches), blended op-for-op as
//     ppy_dcnv2_sample_f32 does (no fp contraction in this file), split into the MFMA operand format ONCE per workgroup
//     (conv_x3.hip splits per wave) and written to LDS in the fragment layout;
//   * the weights arrive through the LDS-DMA loader, as in conv_x3.hip;
//   * the MFMA loop is operand reads + MFMAs only.
// The corner loads and the weight DMA of chunk k+1 are issued before the MFMAs of chunk k and consumed after them
// (two LDS stages, one barrier per chunk).  Split-K over chunk ranges + the shared deterministic combine.
// Three math modes, as the dense convolutions: exact fp32 MFMA (MODE 0), bf16x3 (1), f16x2 (2; activations scaled per
// image by the tracked maximum of the INPUT tensor -- |sample| <= max|x| because the bilinear weights and the mask
// are <= 1).
//
// HBM-level traffic per layer: x, offsets, weights, y -- the 53 MB columns buffer (write + read) of the two-kernel
// form (csrc/dcn.hip, kept as ppy_dcnv2_sample_f32) is gone.
#include <math.h>

#include "conv_shared.h"
#pragma clang fp contract(off)

namespace {

struct DcnArgs {
    ConvArgs c;
    const float *om;     // [M][om_ld]: 18 (y, x)-interleaved offsets, 9 mask logits
    int om_ld;
};

constexpr unsigned DCN_OOB = 0x80000000u;      // beyond any tensor this kernel accepts (< 2 GB)

template <int BM, int BN, int MODE, bool SPLIT, bool VEC, int NW = 4>
__global__ void __launch_bounds__(NW * 64) dcn_fused_kernel(const DcnArgs q) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs &p = q.c;
    constexpr int WCOLS = NW / 2;                               // waves: 2 (rows) x NW / 2 (columns)
    constexpr int WM = BM / 2, WN = BN / WCOLS, TM = WM / 32, TN = WN / 32;
    constexpr int NP = MODE == 2 ? 2 : (MODE == 1 ? 3 : 1);      // operand pieces = planes in LDS
    constexpr int ROWB = MODE == 0 ? 128 : 64;                   // bytes of a 32-deep row
    constexpr int A_BYTES = NP * BM * ROWB, B_BYTES = NP * BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int RPI = 1024 / ROWB;                             // weight rows per DMA instruction
    constexpr int B_PASS = NP * BN / (RPI * NW);
    static_assert(NP * BN % (RPI * NW) == 0, "whole DMA instructions per wave");
    constexpr int TPR = NW * 64 / BM;                            // gather threads per tile row (2, 4 or 8)
    constexpr int NV = 8 / TPR;                                  // 16-byte loads per corner per thread (4, 2 or 1)
    typedef __attribute__((address_space(3))) void *lds_ptr;

    extern __shared__ __attribute__((aligned(16))) char smem_dcn[];
    char *smem = smem_dcn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WCOLS, wn = wave - wm * WCOLS;
    const int tiles_n = (p.K + BN - 1) / BN;
    // XCD-contiguous order over (split, tile) (round 5): workgroups go round-robin to the 8 XCDs in launch order (x fastest, then
    // the split in y), and every XCD takes one contiguous range of the split-major list -- about one split of ALL row tiles, so a
    // split's weight slice crosses the fabric once or twice instead of eight times (with blockIdx.x alone the 23 row tiles of a
    // 19x19 layer's split sat on all eight XCDs: 121 MB fetched for 16 MB of operands, profiles/r05_pmc_layers.txt)
    int tile_id, split;
    {
        const int tiles = (int)gridDim.x, total = tiles * (int)gridDim.y, qd = total >> 3, r = total & 7;
        const int lin = (int)blockIdx.y * tiles + (int)blockIdx.x;
        const int xcd = lin & 7, idx = lin >> 3;
        const int v = xcd * qd + min(xcd, r) + idx;
        split = v / tiles;
        tile_id = v - split * tiles;
    }
    const int tile_m = tile_id / tiles_n, tile_n = tile_id - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kc_begin = split * p.chunks_per_split;
    const int kc_end = min(kc_begin + p.chunks_per_split, p.chunks_total);
    const int nchunks = kc_end - kc_begin;
    const int cch = p.C / 32;
    const int hw = p.Ho * p.Wo;

    // ---- weights: per-lane DMA source offsets, fixed for the tile ----
    unsigned b_off[B_PASS];
    {
        const int drow = MODE == 0 ? lane >> 3 : lane >> 2, dslot = MODE == 0 ? lane & 7 : lane & 3;
        const long long plane_bytes = (long long)p.K * p.Kred * 2;
#pragma unroll
        for (int j = 0; j < B_PASS; ++j) {
            const int rb = (j * NW + wave) * RPI + drow;          // row of the [NP * BN] tile
            const int plane = rb / BN, nrow = rb - plane * BN;
            const int k = min(n0 + nrow, p.K - 1);                // rows >= K are masked at store
            if constexpr (MODE == 0) {
                b_off[j] = (unsigned)((long long)k * p.Kred * 4 + ((dslot ^ ((rb >> 1) & 7)) << 4));
            } else {
                const int scol = dslot ^ ((rb >> 2) & 3);
                b_off[j] = MODE == 2 ? (unsigned)(plane * plane_bytes + (long long)k * 64 + scol * 16)      // [chunk][K][32] planes
                                     : (unsigned)(plane * plane_bytes + (long long)k * p.Kred * 2 + scol * 16);
            }
        }
    }
    const char *wb = MODE == 0 ? reinterpret_cast<const char *>(p.w)
                               : reinterpret_cast<const char *>(MODE == 2 ? p.wf16 : p.w3);
    auto issue_b = [&](int stage, int kc, bool have) {
        const long long uni = !have ? 0 : MODE == 2 ? (long long)kc * p.K * 64 : (long long)kc * 32 * (MODE == 0 ? 4 : 2);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(wb + uni), 0, 0xFFFFFF00u, 0x00020000);
#pragma unroll
        for (int j = 0; j < B_PASS; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(smem + stage * STAGE + A_BYTES + (j * NW + wave) * 1024), 16,
                                                     have ? b_off[j] : 0xFFFFFFF0u, 0, 0, 0);
    };

    // ---- gather: this thread's tile row ----
    const int grow = tid / TPR, cseg = tid - grow * TPR;
    const int gm = m0 + grow;
    const int gmc = min(gm, p.M - 1);
    const int gn = gmc / hw, grem = gmc - gn * hw;
    const int gho = grem / p.Wo, gwo = grem - gho * p.Wo;
    const float *omrow = q.om + (long long)gmc * q.om_ld;
    const int Hp = p.H + 2 * p.pad + 1;
    const float base_y = (float)(gho * p.stride + p.pad), base_x = (float)(gwo * p.stride + p.pad);
    const float row0 = (float)gn * (float)Hp;
    const float ymax = (float)(p.H + 2 * p.pad) - 1.0f, xmax = (float)(p.W + 2 * p.pad) - 1.0f;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.x, 0, (unsigned)((long long)p.N * p.H * p.W * p.x_ld * 4), 0x00020000);
    float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f, mask = 0.f;
    unsigned coff[4] = {DCN_OOB, DCN_OOB, DCN_OOB, DCN_OOB};
    float n_oy, n_ox, n_ml;                      // offsets / mask logit of the NEXT tap, requested one tap ahead
    int cur_tap = -2;                            // (-2: nothing requested yet)
    auto tap_load = [&](int tap) {
        n_oy = omrow[2 * tap];
        n_ox = omrow[2 * tap + 1];
        n_ml = omrow[18 + tap];
    };
    auto tap_setup = [&](int tap) {              // arithmetic of ppy_dcnv2_sample_f32 (csrc/dcn.hip), op for op
        const int kh = tap / 3, kw = tap - kh * 3;
        mask = 1.0f / (1.0f + expf(-n_ml));
        float py = (base_y + (float)(kh - 1)) + n_oy;
        float px = (base_x + (float)(kw - 1)) + n_ox;
        py = fminf(fmaxf(py, 0.0f), ymax);
        px = fminf(fmaxf(px, 0.0f), xmax);
        py = py + row0;
        const float y1f = floorf(py), x1f = floorf(px);
        const float lh = py - y1f, lw = px - x1f;
        const float hh = 1.0f - lh, hwt = 1.0f - lw;
        w1 = hh * hwt;
        w2 = hh * lw;
        w3 = lh * hwt;
        w4 = lh * lw;
        const int y1 = (int)y1f - gn * Hp - p.pad, x1 = (int)x1f - p.pad;
        const bool y1ok = (unsigned)y1 < (unsigned)p.H, y2ok = (unsigned)(y1 + 1) < (unsigned)p.H;
        const bool x1ok = (unsigned)x1 < (unsigned)p.W, x2ok = (unsigned)(x1 + 1) < (unsigned)p.W;
        const bool live = gm < p.M;
        const int rowb = p.x_ld * 4;
        const int o11 = (((gn * p.H + y1) * p.W + x1) * rowb);          // (only used when the corner is inside the image)
        coff[0] = (live && y1ok && x1ok) ? (unsigned)o11 : DCN_OOB;
        coff[1] = (live && y1ok && x2ok) ? (unsigned)(o11 + rowb) : DCN_OOB;
        coff[2] = (live && y2ok && x1ok) ? (unsigned)(o11 + p.W * rowb) : DCN_OOB;
        coff[3] = (live && y2ok && x2ok) ? (unsigned)(o11 + (p.W + 1) * rowb) : DCN_OOB;
        if (tap + 1 < 9) tap_load(tap + 1);
    };
    // The corner loads of chunk k+1 are issued before the MFMAs of chunk k and blended after them.  (Measured alternative:
    // two register sets with the loads TWO chunks ahead.  It wins where one workgroup per CU is latency-bound -- 64x128
    // without split-K 146 -> 124 us on the R50 layers -- and loses at the operating point, 3 workgroups per CU with
    // split-K 4: 77 -> 90 us, the 60 extra registers cost the third workgroup and the kernel is then bound by the bytes of
    // the gather through the CU's vector memory path, 4 corner pixels per sample: tools/dcn_bench.py.)
    // The set carries the blend weights of its chunk's tap.
    struct GSet {
        uintx4 v[4][NV];
        float w1, w2, w3, w4, mask;
    };
    GSet g0;
    auto gather_issue = [&](GSet &g, int kc, bool have) {       // !have: past the end of this split -- every load out of range
        if (have) {
            const int tap = kc / cch;
            if (tap != cur_tap) {                // (uniform)
                if (cur_tap + 1 != tap) tap_load(tap);
                tap_setup(tap);
                cur_tap = tap;
            }
        }
        const int cc = kc % cch;
        const unsigned cb = (unsigned)((cc * 32 + cseg * NV * 4) * 4);
        g.w1 = w1; g.w2 = w2; g.w3 = w3; g.w4 = w4; g.mask = mask;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int v = 0; v < NV; ++v)
                g.v[c][v] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)((have ? coff[c] : DCN_OOB) + cb + v * 16), 0, 0);
    };

    // f16x2: per-image activation scale from the tracked maximum of x (as conv_x3.hip): for the gather row, and the
    // inverse for the tile rows this lane finishes in the epilogue
    float gsa = 1.0f, inv_sa[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) inv_sa[i] = 1.0f;
    auto scale_exp = [&](int n) {
        const float mx = amax_read(p.amax_in, n);
        const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
        int f = 267 - e;
        return f < 103 ? 103 : (f > 167 ? 167 : f);
    };

    auto gather_store = [&](const GSet &g, int stage) {
        char *a_base = smem + stage * STAGE;
        float r[NV * 4];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v1 = __uint_as_float(g.v[0][v][u]), v2 = __uint_as_float(g.v[1][v][u]);
                const float v3 = __uint_as_float(g.v[2][v][u]), v4 = __uint_as_float(g.v[3][v][u]);
                float t = g.w1 * v1 + g.w2 * v2;
                t = t + g.w3 * v3;
                t = t + g.w4 * v4;
                r[v * 4 + u] = t * g.mask;
            }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int pos = (cseg * NV + v) ^ ((grow >> 1) & 7);
                *reinterpret_cast<floatx4 *>(a_base + grow * 128 + pos * 16) = floatx4{r[4 * v], r[4 * v + 1], r[4 * v + 2], r[4 * v + 3]};
            }
        } else if constexpr (NV == 1) {      // eight gather threads per row: 4 channels = half a 16-byte slot per plane (f16x2 only)
            static_assert(NV != 1 || MODE == 2, "the 8-wave tiles are f16x2 tiles");
            const int pos = (cseg >> 1) ^ ((grow >> 2) & 3);
            char *dst = a_base + grow * 64 + pos * 16 + (cseg & 1) * 8;
            typedef __attribute__((ext_vector_type(2))) unsigned uintx2;
            uintx2 P0, P1;
#pragma unroll
            for (int qd = 0; qd < 2; ++qd) {
                const float xa = r[2 * qd], xb = r[2 * qd + 1];
                const unsigned h = cvt_pk_f16(xa * gsa, xb * gsa);
                P0[qd] = h;
                P1[qd] = cvt_pk_f16(fmaf(xa, gsa, -f16_lo(h)), fmaf(xb, gsa, -f16_hi(h)));
            }
            *reinterpret_cast<uintx2 *>(dst) = P0;
            *reinterpret_cast<uintx2 *>(dst + BM * 64) = P1;
        } else {
#pragma unroll
            for (int s8 = 0; s8 < NV / 2; ++s8) {       // one 16-byte slot = 8 channels per plane
                const int pos = (cseg * (NV / 2) + s8) ^ ((grow >> 2) & 3);
                char *dst = a_base + grow * 64 + pos * 16;
                const float *e = r + s8 * 8;
                if constexpr (MODE == 2) {
                    uintx4 P0, P1;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const float xa = e[2 * qd], xb = e[2 * qd + 1];
                        const unsigned h = cvt_pk_f16(xa * gsa, xb * gsa);
                        P0[qd] = h;
                        P1[qd] = cvt_pk_f16(fmaf(xa, gsa, -f16_lo(h)), fmaf(xb, gsa, -f16_hi(h)));
                    }
                    *reinterpret_cast<uintx4 *>(dst) = P0;
                    *reinterpret_cast<uintx4 *>(dst + BM * 64) = P1;
                } else {
                    bf16x8 t3[3];
                    split8(floatx4{e[0], e[1], e[2], e[3]}, floatx4{e[4], e[5], e[6], e[7]}, t3);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uintx4 *>(dst + pl * BM * 64) = __builtin_bit_cast(uintx4, t3[pl]);
                }
            }
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31, fkh = lane >> 5;
    auto compute = [&](int stage) {
        const char *a_base = smem + stage * STAGE, *b_base = a_base + A_BYTES;
        if constexpr (MODE == 0) {
            // exact fp32: v_mfma_f32_32x32x2_f32; lane-half h holds k = 16h + [0, 16) of its row, MFMA step (u, e) pairs
            // k = 4u + e of half 0 with k = 16 + 4u + e of half 1 (any pairing works as long as A and B agree)
            floatx4 a[TM][4], b[TN][4];
            const int sw = (frow >> 1) & 7;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i][u] = *reinterpret_cast<const floatx4 *>(a_base + (wm * WM + i * 32 + frow) * 128 + (((4 * fkh + u) ^ sw) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j][u] = *reinterpret_cast<const floatx4 *>(b_base + (wn * WN + j * 32 + frow) * 128 + (((4 * fkh + u) ^ sw) << 4));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][u][e], b[j][u][e], acc[i][j], 0, 0, 0);
        } else {
            // partial products (piece of A, piece of B), smallest first; f16x2 uses the last three with pieces {0, 1}
            constexpr int ta[6] = {2, 1, 0, 1, 0, 0}, tb[6] = {0, 1, 2, 0, 1, 0};
            constexpr int T0 = MODE == 2 ? 3 : 0;
            const int sw = (frow >> 2) & 3;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int fo = frow * 64 + (((2 * s + fkh) ^ sw) << 4);
                uintx4 a[TM][NP], b[NP][TN];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        a[i][pl] = *reinterpret_cast<const uintx4 *>(a_base + (pl * BM + wm * WM + i * 32) * 64 + fo);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        b[pl][j] = *reinterpret_cast<const uintx4 *>(b_base + (pl * BN + wn * WN + j * 32) * 64 + fo);
                }
#pragma unroll
                for (int t = T0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if constexpr (MODE == 2)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][ta[t]]),
                                                                                  __builtin_bit_cast(f16x8, b[tb[t]][j]), acc[i][j], 0, 0, 0);
                            else
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][ta[t]]),
                                                                                   __builtin_bit_cast(bf16x8, b[tb[t]][j]), acc[i][j], 0, 0, 0);
                        }
            }
        }
    };

    constexpr bool DEEP = NW == 8 && NV == 1;      // eight gather threads per row: the corner loads run TWO chunks ahead (two register sets of 16 VGPRs)
    if (nchunks > 0) {
        issue_b(0, kc_begin, true);
        gather_issue(g0, kc_begin, true);
        GSet g1;
        if constexpr (DEEP) gather_issue(g1, kc_begin + 1, 1 < nchunks);
        if constexpr (MODE == 2) {      // behind the first requests, so that this latency hides behind theirs
            const int f = scale_exp(gn);
            gsa = __uint_as_float((unsigned)f << 23);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mrow = min(m0 + wm * WM + i * 32 + frow, p.M - 1);
                inv_sa[i] = __uint_as_float((unsigned)(254 - scale_exp(mrow / hw)) << 23);
            }
        }
        gather_store(g0, 0);
        if constexpr (DEEP) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * NV) : "memory");      // (the corners of chunk 1 may still be in flight)
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (!DEEP) {
            // iteration k: request the weights and the corners of chunk k+1; MFMAs of chunk k; blend + split + store chunk k+1.
            // Requests past the end of the split are issued out of range (no traffic, no branches in the loop body).
            for (int k = 0; k < nchunks; ++k) {
                const int st = k & 1;
                issue_b(st ^ 1, kc_begin + k + 1, k + 1 < nchunks);
                gather_issue(g0, kc_begin + k + 1, k + 1 < nchunks);
                __builtin_amdgcn_sched_barrier(0);
                compute(st);
                gather_store(g0, st ^ 1);
                // every LDS read of chunk k has returned, chunk k+1 is complete in the other stage (here and, after the
                // barrier, in all waves)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        } else {
            // iteration k: weights of chunk k+1 (DMA), corners of chunk k+2 into the set that chunk k's blend freed; MFMAs of chunk k;
            // blend + split + store chunk k+1 from the OTHER set (requested one iteration ago).  The counted wait leaves only
            // the 4 * NV corner loads just issued in flight: the DMA of chunk k+1, issued before them, has landed.
            auto body = [&](int k, GSet &issue_set, GSet &store_set) {
                const int st = k & 1;
                issue_b(st ^ 1, kc_begin + k + 1, k + 1 < nchunks);
                gather_issue(issue_set, kc_begin + k + 2, k + 2 < nchunks);
                __builtin_amdgcn_sched_barrier(0);
                compute(st);
                gather_store(store_set, st ^ 1);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * NV) : "memory");
                __builtin_amdgcn_s_barrier();
            };
            for (int k = 0; k < nchunks; k += 2) {
                body(k, g0, g1);
                if (k + 1 < nchunks) body(k + 1, g1, g0);
            }
        }
    }
    float rowscale[TM][4];
    if constexpr (MODE == 2) {
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) rowscale[i][t] = __shfl(inv_sa[i], (lane >> 3) + 8 * t);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float inv = __shfl(inv_sa[i], (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5));
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j][e] *= inv;
                }
        }
    }
    tile_epilogue<TM, TN, WM, WN, SPLIT, VEC>(p, acc, reinterpret_cast<float *>(smem), m0, n0, wm, wn, lane, wave, split,
                                              (MODE == 2 && VEC) ? rowscale : nullptr);
#endif
}

struct DcnTile {
    int bm, bn;
};
constexpr DcnTile kTiles[] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}, {64, 256}, {128, 256}};      // 4 waves (2 x 2) each
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);
// Round 3: f16x2 tiles with EIGHT waves (2 x 4), ids 3 * kNumTiles + i.  What bounds the kernel is the bytes of the gather
// through the CU (4 corner pixels per sample: 4x the bytes of a dense A tile) plus the weights every M-tile re-reads; 64 x 512
// gathers every sample ONCE for all 512 output channels (0.09 B per MAC against 0.18 at 64 x 128), and with eight gather threads
// per row a thread holds 4 corners x 16 B per chunk.  144 KB of LDS: one workgroup per CU, split-K fills the chip.
constexpr DcnTile kTiles8[] = {{64, 512}, {64, 256}, {128, 512}};
constexpr int kNumTiles8 = sizeof(kTiles8) / sizeof(kTiles8[0]);

template <int BM, int BN, int MODE, bool SPLIT, bool VEC, int NW = 4>
int launch_one(const DcnArgs &q, int splits, size_t lds, int tiles, hipStream_t stream) {
    auto k = dcn_fused_kernel<BM, BN, MODE, SPLIT, VEC, NW>;
    static PpyLdsAttr attr;
    if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), 160 * 1024) != PPY_OK) return PPY_ERR_LAUNCH;
    hipLaunchKernelGGL(k, dim3(tiles, splits), dim3(NW * 64), lds, stream, q);
    return PPY_OK;
}

template <int BM, int BN, int MODE, int NW = 4>
int launch_tile(DcnArgs q, int splits, hipStream_t stream) {
    ConvArgs &p = q.c;
    constexpr int NP = MODE == 2 ? 2 : (MODE == 1 ? 3 : 1), ROWB = MODE == 0 ? 128 : 64;
    size_t lds = 2 * (size_t)(NP * (BM + BN) * ROWB);
    const size_t epi = (size_t)NW * 32 * LDS_LD * sizeof(float);
    if (lds < epi) lds = epi;
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.K, BN);
    const bool vec = vec_epilogue_ok(p);
    int rc;
    if (splits > 1) {
        rc = vec ? launch_one<BM, BN, MODE, true, true, NW>(q, splits, lds, tiles, stream)
                 : launch_one<BM, BN, MODE, true, false, NW>(q, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
        launch_splitk_reduce(p, splits, vec, stream);
    } else {
        rc = vec ? launch_one<BM, BN, MODE, false, true, NW>(q, splits, lds, tiles, stream)
                 : launch_one<BM, BN, MODE, false, false, NW>(q, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
    }
    return ppy_launch_status();
}

template <int MODE>
int launch_mode(const DcnArgs &q, int tile, int splits, hipStream_t stream) {
    switch (tile) {
        case 0: return launch_tile<128, 128, MODE>(q, splits, stream);
        case 1: return launch_tile<64, 128, MODE>(q, splits, stream);
        case 2: return launch_tile<128, 64, MODE>(q, splits, stream);
        case 3: return launch_tile<64, 64, MODE>(q, splits, stream);
        case 4: return launch_tile<64, 256, MODE>(q, splits, stream);      // the gather is shared by 256 output channels
        case 5: return launch_tile<128, 256, MODE>(q, splits, stream);
    }
    if constexpr (MODE == 2) {
        switch (tile - kNumTiles) {
            case 0: return launch_tile<64, 512, 2, 8>(q, splits, stream);
            case 1: return launch_tile<64, 256, 2, 8>(q, splits, stream);
            case 2: return launch_tile<128, 512, 2, 8>(q, splits, stream);      // all 160 KB of LDS; the weights stream half as often
        }
    }
    return PPY_ERR_BAD_ARG;
}

// cfg < 0: a tile that gives the 256 CUs at least two workgroups each where the layer allows it
void pick(int M, int K, int chunks, int *tile, int *splits) {
    const int t = K <= 64 ? (M >= 16384 ? 2 : 3) : (M >= 16384 ? 0 : 1);
    const int tiles = ceil_div(M, kTiles[t].bm) * ceil_div(K, kTiles[t].bn);
    int s = ceil_div(700, tiles);
    const int smax = chunks / 8 > 0 ? chunks / 8 : 1;
    s = s > smax ? smax : s;
    s = s > 9 ? 9 : s;
    *tile = t;
    *splits = s < 1 ? 1 : s;
}

bool dcn_geometry(int N, int H, int W, int C, int K, int stride, int pad, int *Ho, int *Wo) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || stride <= 0 || pad < 0 || C % 32 != 0) return false;
    *Ho = (H + 2 * pad - 2) / stride;        // reference :567-568
    *Wo = (W + 2 * pad - 2) / stride;
    if (*Ho <= 0 || *Wo <= 0) return false;
    return (long long)N * *Ho * *Wo <= 0x7fffffffLL / 4;
}

int resolve(int M, int K, int chunks, int cfg, int splitk, int *mode, int *tile, int *splits) {
    if (cfg >= 3 * kNumTiles + kNumTiles8) return PPY_ERR_BAD_ARG;
    int ht, hs;
    pick(M, K, chunks, &ht, &hs);
    const bool w8 = cfg >= 3 * kNumTiles;                       // the eight-wave f16x2 tiles
    *mode = cfg < 0 ? -1 : (w8 ? 2 : cfg / kNumTiles);
    *tile = cfg < 0 ? ht : (w8 ? kNumTiles + cfg - 3 * kNumTiles : cfg % kNumTiles);
    int s = splitk <= 0 ? (cfg < 0 ? hs : 1) : splitk;
    if (s > chunks) s = chunks;
    *splits = ceil_div(chunks, ceil_div(chunks, s));      // no empty split
    return PPY_OK;
}

}  // namespace

extern "C" int ppy_dcnv2_num_configs(void) { return 3 * kNumTiles + kNumTiles8; }

extern "C" size_t ppy_dcnv2_workspace_bytes(int N, int H, int W, int C, int K, int stride, int pad, int cfg, int splitk) {
    int Ho, Wo, mode, tile, s;
    if (!dcn_geometry(N, H, W, C, K, stride, pad, &Ho, &Wo)) return 0;
    const int M = N * Ho * Wo;
    if (resolve(M, K, 9 * (C / 32), cfg, splitk, &mode, &tile, &s) != PPY_OK) return 0;
    return s > 1 ? (size_t)s * M * K * sizeof(float) : 0;
}

extern "C" int ppy_dcnv2_f32(const float *x, int x_ld, const float *w_krsc, const void *w_x3, const void *w_f16x2,
                             const float *scale, const float *scale_f16x2, const float *shift, const float *offset_mask,
                             int om_ld, float *y, int y_ld, int N, int H, int W, int C, int K, int stride, int pad, int act,
                             int cfg, int splitk, const float *amax_in, float *amax_out, void *ws, size_t ws_bytes,
                             void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && w_krsc && scale && shift && offset_mask && y);
    int Ho, Wo;
    PPY_CHECK_ARG(dcn_geometry(N, H, W, C, K, stride, pad, &Ho, &Wo));
    PPY_CHECK_ARG(x_ld >= C && x_ld % 4 == 0 && y_ld >= K && om_ld >= 27);
    PPY_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_krsc & 15) == 0);
    PPY_CHECK_ARG(act == PPY_ACT_NONE || act == PPY_ACT_RELU || act == PPY_ACT_LEAKY);
    const int M = N * Ho * Wo, chunks = 9 * (C / 32);
    int mode, tile, s;
    int rc = resolve(M, K, chunks, cfg, splitk, &mode, &tile, &s);
    if (rc != PPY_OK) return rc;
    const bool can_f16 = w_f16x2 && scale_f16x2 && amax_in, can_x3 = w_x3 != nullptr;
    if (mode < 0) mode = can_f16 ? 2 : (can_x3 ? 1 : 0);      // the best scheme the caller brought operands for
    PPY_CHECK_ARG(mode != 2 || can_f16);
    PPY_CHECK_ARG(mode != 1 || can_x3);
    PPY_CHECK_ARG(mode == 0 || ((uintptr_t)(mode == 2 ? w_f16x2 : w_x3) & 15) == 0);
    // 32-bit offsets: corner loads (out-of-image corners are offsets >= 2 GB) and the weight DMA
    const long long xbytes = (long long)N * H * W * x_ld * 4;
    const long long wbytes = (long long)K * 9 * C * (mode == 0 ? 4 : (mode == 1 ? 6 : 4));
    if (xbytes >= 0x7FFFF000LL || wbytes >= 0xFFFFF000LL) return PPY_ERR_UNSUPPORTED;
    if (s > 1) {
        if (!ws || ws_bytes < (size_t)s * M * K * sizeof(float)) return PPY_ERR_WORKSPACE;
    }
    DcnArgs q;
    ConvArgs &p = q.c;
    p.x = x; p.w = w_krsc; p.w3 = (const unsigned short *)w_x3; p.wf16 = (const unsigned short *)w_f16x2;
    p.scale = mode == 2 ? scale_f16x2 : scale;      // (1 / weight scale folded in)
    p.scale_f16 = scale_f16x2; p.posb_f16 = nullptr; p.amax_in = amax_in; p.amax_out = amax_out;
    p.shift = shift; p.res = nullptr; p.posb = nullptr;
    p.y = y; p.part = (float *)ws;
    p.x_ld = x_ld; p.res_ld = 0; p.y_ld = y_ld;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.K = K; p.R = 3; p.S = 3;
    p.stride = stride; p.pad = pad; p.act = act; p.ups = 0;
    p.M = M; p.Kred = 9 * C; p.cchunks = C / 32; p.chunks_total = chunks;
    p.chunks_per_split = ceil_div(chunks, s);
    p.nstages = 2;
    p.trace = nullptr;
    q.om = offset_mask;
    q.om_ld = om_ld;
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
        case 0: return launch_mode<0>(q, tile, s, st);
        case 1: return launch_mode<1>(q, tile, s, st);
        default: return launch_mode<2>(q, tile, s, st);
    }
}
