// fp32 implicit-GEMM convolution on the 16-bit MFMA of gfx950: "f16x2" and "bf16x3" operand splits.
//
// Same operator as conv_igemm.hip (Conv2dUnit.forward of the reference, model/custom_layers.py:243-253,
// with the residual / CoordConv / upsample terms around it), same tensors (fp32 NHWC in, fp32 out),
// same epilogue.  Only the inner product is organised differently -- gfx950 has no TF32-class path and its
// exact-fp32 MFMA runs at 1/16 of the 16-bit rate, so an fp32 product is assembled from exactly-split pieces:
//
//   bf16x3 (F16 = false): every fp32 operand is split EXACTLY into three bf16 terms  a = a0 + a1 + a2  (8 + 8 + 8
//     significant bits, round-to-nearest at each level, the residuals are exact in fp32) and  a*b  is evaluated as
//     the six partial products of weight >= 2^-16   a2*b0 + a1*b1 + a0*b2 + a1*b0 + a0*b1 + a0*b0
//     on v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are exact, accumulation is fp32); the three dropped
//     products are <= 2^-24 |a*b| each.  No scaling, no range question.  MFMA ceiling 2516.8 / 6 = 419.5 TFLOP/s.
//   f16x2 (F16 = true): both operands are first scaled by a power of two into the fp16 range -- weights per output
//     channel at plan time (ppy_conv2d_split_weights_f16x2, 1/s folded into the epilogue scale), activations per
//     image at run time from the maximum their producers track (amax_track / ConvArgs::amax_in, 8 atomic-max
//     slots per image) -- and split into TWO fp16 terms  a*s = a0 + a1  (RNE; the residual fma(a, s, -a0) is exact); a*b is
//     the three leading products  a1*b0 + a0*b1 + a0*b0  on v_mfma_f32_32x32x16_f16.  Representation error <= 2^-22
//     per operand, unbiased, so it averages out over the reduction.  MFMA ceiling 2516.8 / 3 = 839 TFLOP/s.
//   Measured on MI355X against fp64 (tools/probes/f16x2_probe.hip, bf16x_probe.hip -> profiles/r01_*_numerics.txt):
//   rms error of sum|a*b| for K = 64..4608: f16x2 1.5-3.3e-8, bf16x3 1.9-3.6e-8, exact-fp32 MFMA 2.3-4.5e-8 (a
//   k-ordered fma chain rounds once per product; here 16 products are summed per accumulator rounding).
//   (Inf/NaN inputs give NaN where the fp32 kernel gives Inf.)
//
// Data movement (all through the LDS-DMA loader of conv_igemm.hip, counted vmcnt, one barrier per 32-deep chunk):
//   * activations stay fp32 in HBM and in LDS ([BM][32] floats, XOR-swizzled 16-byte slots); each wave reads the
//     rows of ITS sub-tile, scales/splits them in registers (f16x2: 3 VALU instructions per element, bf16x3: 4.5,
//     issued in the shadow of the MFMAs) and feeds the MFMA directly;
//   * weights are constant: split once per plan into 2 fp16 / 3 bf16 planes [NP][K][R][S][C] and DMA'd as NP
//     [BN][32] 16-bit tiles per chunk.
// Wave tiles are wide in N (64x128 ... 32x64) so that one split of an A fragment feeds many MFMAs.
#include "conv_shared.h"

#include <type_traits>

#ifndef PPY_X3_BBLOCK
#define PPY_X3_BBLOCK 1   // f16x2 weight planes in [chunk][K][32] order (0 = [K][Kred], for A/B rebuilds)
#endif
#ifndef PPY_X3_XCD
#define PPY_X3_XCD 1      // XCD-contiguous tile order (0 = plain blockIdx order, for A/B rebuilds: +1.3 % on the R50 step)
#endif
#ifndef PPY_X3_ABL
#define PPY_X3_ABL 0      // ablation switch for experiments (tools/experiments/x3_ablate.sh; results are garbage, only the timing means something):
                          // 1 = no DMA in the loop, 2 = no split, 3 = neither, 4 = no mid-chunk barrier either, 5 = MFMA only,
                          // 6 = everything, but every DMA piece out of range (no memory traffic), 7 = operand delivery only (DMA +
                          // barriers), 8 = 7 with the weight pieces only, 9 = 7 with the activation pieces only.  DESIGN.md 8 item 1.
#endif

namespace {

__global__ void __launch_bounds__(256) split_weights_kernel(const float *w, long long n, unsigned short *out) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= n) return;
    const float a = w[i], b = i + 1 < n ? w[i + 1] : 0.f;
    const unsigned P0 = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(P0 << 16), rb = b - __uint_as_float(P0 & 0xffff0000u);
    const unsigned P1 = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(P1 << 16), sb = rb - __uint_as_float(P1 & 0xffff0000u);
    const unsigned P2 = cvt_pk_bf16(sa, sb);
    out[i] = (unsigned short)P0;
    out[n + i] = (unsigned short)P1;
    out[2 * n + i] = (unsigned short)P2;
    if (i + 1 < n) {
        out[i + 1] = (unsigned short)(P0 >> 16);
        out[n + i + 1] = (unsigned short)(P1 >> 16);
        out[2 * n + i + 1] = (unsigned short)(P2 >> 16);
    }
}

// A tile: [BM][32] fp32, 128-byte rows, 16-byte slot c of row r stored at c ^ ((r>>1)&7)   (as conv_igemm.hip)
// B tile: [3 planes][BN][32] bf16, 64-byte rows, slot c of row r stored at c ^ ((r>>2)&3)
// F16 = false: "bf16x3" (3 bf16 pieces, 6 products).  F16 = true: "f16x2" -- 2 fp16 pieces of the operands pre-scaled
// by powers of two into the fp16 range (weights per output channel at plan time, activations by the producer-tracked
// maximum of the input tensor), 3 products on v_mfma_f32_32x32x16_f16; see the file header.
// SLAB = true ("slab reuse", 3x3 / stride 1 / pad 1 layers, f16x2 only): the three taps (r, 0..2) of an output tile of BM
// consecutive pixels read the BM + 2 consecutive input pixels  m0 - 1 + (r-1)*W ... , so ONE slab of BM + 8 rows per
// (channel chunk, r) is DMA'd instead of three BM-row tiles (A bytes / 2.8, all operand bytes -32 % on a 128x128 tile);
// the fragment of tap s is the slab read at row offset s, and a (pixel, tap) pair that falls into the padding is zeroed by
// selecting 0 instead of the activation scale in the operand split of that lane.  Two slabs alternate; the pieces of the
// next slab ride in the DMA groups of the first two chunks of a super-chunk (the third carries out-of-range dummies so
// that every group has the same G pieces and the counted vmcnt waits stay what they are).
// BNS = true (training forward, ppy_conv2d_train_fwd_f32): the epilogue also emits the BatchNorm statistics of what it stores
// (tile_bn_stats, conv_shared.h).  A separate instantiation: as a run-time branch it cost the inference kernels 12-20 VGPRs.
// GP = true ("global pre-split", f16x2 only): the input tensor was stored by its producer as finished operands -- per pixel and
// 32-channel group 32 fp16 first terms, then 32 fp16 second terms of x * s_image (ConvArgs::xscale; conv_shared.h's split
// store) -- so a chunk's A row is the same 128 bytes, DMA'd and swizzled as before, but the fragment reads ARE the MFMA
// operands: no scale / split VALU instruction in the main loop.
template <int BM, int BN, int WM, int WN, bool F16, bool SPLIT, bool VEC, bool SLAB = false, bool BNS = false, bool GP = false>
__global__ void __launch_bounds__(64 * (BM / WM) * (BN / WN)) conv_igemm_x3_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NW = (BM / WM) * (BN / WN);
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int NP = F16 ? 2 : 3;                    // pieces per operand = weight planes
    constexpr int B_ROWS = NP * BN;
    static_assert(BM % (8 * NW) == 0 && B_ROWS % (16 * NW) == 0, "whole DMA instructions per wave");
    static_assert(!SLAB || F16, "the slab variant zeroes padding taps through the f16x2 activation scale");
    static_assert(!GP || (F16 && !SLAB && !SPLIT && VEC), "pre-split input: f16x2 tiles, one split, vector epilogue");
    constexpr int SLAB_PIECES = BM / 8 + 1;                        // slab rows 0 .. BM+7 (BM + 2 are used)
    constexpr int SLAB_BYTES = SLAB_PIECES * 1024;
    constexpr int SP_W = (SLAB_PIECES + NW - 1) / NW;              // slab pieces per wave per super-chunk
    constexpr int AS = (SP_W + 1) / 2;                             // ... carried by each of two DMA groups
    constexpr int A_PASS = SLAB ? AS : BM / (8 * NW), B_PASS = B_ROWS / (16 * NW);
    constexpr int G = A_PASS + B_PASS;                 // DMA instructions per wave per chunk
    constexpr int A_BYTES = SLAB ? 0 : BM * 128, B_BYTES = B_ROWS * 64;      // (slab variant: stages hold the weights only)
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int STAGE_BASE = SLAB ? 2 * SLAB_BYTES + NW * 1024 : 0;        // two slabs + 1 KB per wave for the dummy pieces
    typedef __attribute__((address_space(3))) void *lds_ptr;

    extern __shared__ __attribute__((aligned(16))) char smem_x3[];
    char *smem = smem_x3;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: LDS-DMA destinations stay on the SALU
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.K + BN - 1) / BN;
#if PPY_X3_XCD
    // every XCD (own L2) takes one contiguous range of the tile order -- tile_m-major, or column panels where the weights outweigh
    // the activations (conv_shared.h, ppy_panel_n / ppy_tile_of)
    int tile_m, tile_n;
    ppy_tile_of(p, tiles_n, tile_m, tile_n);
#else
    const int tile_id = blockIdx.x;
    const int tile_m = tile_id / tiles_n, tile_n = tile_id - tile_m * tiles_n;
#endif
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int kc_begin = split * p.chunks_per_split;
    const int kc_end = min(kc_begin + p.chunks_per_split, p.chunks_total);
    const int nchunks = kc_end - kc_begin;
    unsigned long long t_start = 0, c_start = 0, c_loop = 0, c_epi = 0;
    if (p.trace) {
        t_start = __builtin_amdgcn_s_memrealtime();     // 100 MHz, chip-wide
        c_start = __builtin_amdgcn_s_memtime();         // shader clock
    }

    // ---- per-lane DMA source offsets (bytes), fixed for the whole tile ----
    const int hw = p.Ho * p.Wo;
    const unsigned OOB = 0xFFFFFFF0u;
    const long long bias = (long long)(p.pad * p.W + p.pad) * p.x_ld * 4;   // keeps offsets >= 0
    unsigned a_off[A_PASS], a_ok[A_PASS], b_off[B_PASS];
    if constexpr (!SLAB) {
        const int drow = lane >> 3, dslot = lane & 7;
        // (n, ho, wo) of the first row by division, of the following rows (8*NW further each) incrementally
        const int step_rows = 8 * NW;
        const int step_ho = step_rows / p.Wo, step_wo = step_rows - step_ho * p.Wo;
        int m_first = min(m0 + wave * 8 + drow, p.M - 1);
        int n = m_first / hw, rem = m_first - n * hw;
        int ho = rem / p.Wo, wo = rem - ho * p.Wo;
#pragma unroll
        for (int j = 0; j < A_PASS; ++j) {
            const int row = (j * NW + wave) * 8 + drow;
            const int scol = dslot ^ ((row >> 1) & 7);
            const int mr = m0 + row;
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            a_off[j] = (unsigned)((((long long)n * p.H + hi0) * p.W + wi0) * p.x_ld * 4 + bias + scol * 16);
            // taps inside the image: (valid rows r) x (valid columns s) as a bit mask over r*S + s
            unsigned colmask = 0, okb = 0;
            for (int s2 = 0; s2 < p.S; ++s2)
                if ((unsigned)(wi0 + s2) < (unsigned)p.W) colmask |= 1u << s2;
            for (int r = 0; r < p.R; ++r)
                if ((unsigned)(hi0 + r) < (unsigned)p.H) okb |= colmask << (r * p.S);
            a_ok[j] = mr < p.M ? okb : 0u;
            // advance to the row of the next pass (rows beyond M are masked above; their offsets are unused)
            wo += step_wo;
            ho += step_ho;
            if (wo >= p.Wo) { wo -= p.Wo; ++ho; }
            while (ho >= p.Ho) { ho -= p.Ho; ++n; }
        }
    }
    {
        const int drow = lane >> 2, dslot = lane & 3;
        const long long plane_bytes = (long long)p.K * p.Kred * 2;
#pragma unroll
        for (int j = 0; j < B_PASS; ++j) {
            const int rb = (j * NW + wave) * 16 + drow;        // row of the [3*BN] B tile
            const int plane = rb / BN, nrow = rb - plane * BN;
            const int scol = dslot ^ ((rb >> 2) & 3);
            const int k = min(n0 + nrow, p.K - 1);             // rows >= K are masked at store
            b_off[j] = (F16 && PPY_X3_BBLOCK) ? (unsigned)(plane * plane_bytes + (long long)k * 64 + scol * 16)      // [chunk][K][32] planes
                           : (unsigned)(plane * plane_bytes + (long long)k * p.Kred * 2 + scol * 16);
        }
    }

    const int RS = p.R * p.S;
    int l_cc = kc_begin / RS, l_tap = kc_begin - l_cc * RS;
    int l_r = l_tap / p.S, l_s = l_tap - l_r * p.S;
    const char *xb = reinterpret_cast<const char *>(p.x) - bias;
    const char *wb = reinterpret_cast<const char *>(F16 ? p.wf16 : p.w3);

    // One chunk = G DMA pieces per wave.  begin/piece/end are separate so that the steady-state loop can
    // drop one piece into an MFMA slot at a time (a piece costs ~60+ issue cycles: M0 write + buffer_load);
    // `have` = false turns every piece into an out-of-range access (no memory traffic, zeros into LDS), which
    // keeps the loop body branch-free after the last chunk has been requested.
    __amdgpu_buffer_rsrc_t cur_ra, cur_rb;
    unsigned cur_tapbit = 0, cur_lds = 0;
    bool cur_have = false;
    // slab variant: the group issued at mid-chunk k (k = -1 for the last group of the prologue when three stages are in
    // flight) carries the pieces of the slab of super-chunk  (k + L) / 3 + 1  at position  (k + L) % 3,  L = NS - 2
    const int Mpix = p.N * p.H * p.W;
    int sl_q = -2;                               // q = k + L of the group being issued; the prologue groups are k = -NS .. -1, i.e. q = -2 ..
    int sl_pos = 2, sl_pb = 0, sl_dst = 0;       // position in the super-chunk, first pixel of the slab, LDS offset of the slab
    bool sl_have = false;
    const int drow8 = lane >> 3, dslot8 = lane & 7;
    auto slab_begin = [&]() {                    // (called from issue_begin)
        sl_have = false;
        sl_pos = 2;
        if (sl_q >= 0) {
            const int sc = sl_q / 3 + 1;         // target super-chunk, local to this split
            sl_pos = sl_q - (sc - 1) * 3;
            if (sl_pos < 2 && sc * 3 < nchunks) {
                const int kc = kc_begin + sc * 3;
                const int cc = kc / 9, r = (kc - cc * 9) / 3;
                sl_pb = m0 - 1 + (r - 1) * p.W;
                sl_dst = (sc & 1) * SLAB_BYTES;
                cur_ra = __builtin_amdgcn_make_buffer_rsrc((void *)(reinterpret_cast<const char *>(p.x) + (long long)cc * 128), 0,
                                                           0xFFFFFF00u, 0x00020000);
                sl_have = true;
            }
        }
        ++sl_q;
    };
    auto slab_piece = [&](int d) {               // d in [0, AS)
        const int idx = (sl_pos * AS + d) * NW + wave;           // piece of the slab = 8 rows
        const bool real = sl_have && sl_pos < 2 && idx < SLAB_PIECES;
        const int t = idx * 8 + drow8;
        const int pix = sl_pb + t;
        const unsigned off = (real && (unsigned)pix < (unsigned)Mpix)
                                 ? (unsigned)pix * (unsigned)(p.x_ld * 4) + (unsigned)((dslot8 ^ ((t >> 1) & 7)) << 4)
                                 : OOB;
        const unsigned dst = real ? (unsigned)(sl_dst + idx * 1024) : (unsigned)(2 * SLAB_BYTES + wave * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(cur_ra, (lds_ptr)(smem + dst), 16, off, 0, 0, 0);
    };
    auto issue_begin = [&](int stage, bool have) {
        if (PPY_X3_ABL == 6) have = false;       // ablation: every piece out of range (issue + LDS zero-fill cost, no memory traffic)
        const long long a_uni = ((long long)(l_r * p.W + l_s) * p.x_ld + l_cc * 32) * 4;
        const long long b_uni = (F16 && PPY_X3_BBLOCK) ? ((long long)l_tap * (p.C / 32) + l_cc) * p.K * 64
                                    : ((long long)l_tap * p.C + l_cc * 32) * 2;
        if constexpr (SLAB) {
            cur_ra = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, 0xFFFFFF00u, 0x00020000);
            slab_begin();
        } else {
            cur_ra = __builtin_amdgcn_make_buffer_rsrc((void *)(xb + (have ? a_uni : 0)), 0, 0xFFFFFF00u, 0x00020000);
        }
        cur_rb = __builtin_amdgcn_make_buffer_rsrc((void *)(wb + (have ? b_uni : 0)), 0, 0xFFFFFF00u, 0x00020000);
        cur_tapbit = have ? (1u << l_tap) : 0u;
        cur_have = have;
        cur_lds = (unsigned)(STAGE_BASE + stage * STAGE + wave * 1024);
        ++l_tap;
        ++l_s;
        if (l_s == p.S) { l_s = 0; ++l_r; }
        if (l_tap == RS) { l_tap = 0; l_r = 0; l_s = 0; ++l_cc; }
    };
    auto issue_piece = [&](int d) {       // d in [0, G): compile-time after unrolling
        if (d < A_PASS) {
            if (PPY_X3_ABL == 8) return;                    // (ablation: weight pieces only)
            if constexpr (SLAB) {
                slab_piece(d);
                return;
            }
            const unsigned off = (a_ok[d] & cur_tapbit) ? a_off[d] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(cur_ra, (lds_ptr)(smem + cur_lds + d * NW * 1024), 16, off, 0, 0, 0);
        } else {
            if (PPY_X3_ABL == 9) return;                    // (ablation: activation pieces only)
            const int j = d - A_PASS;
            const unsigned off = cur_have ? b_off[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(cur_rb, (lds_ptr)(smem + cur_lds + A_BYTES + j * NW * 1024), 16, off, 0,
                                                     0, 0);
        }
    };
    auto issue = [&](int stage) {
        issue_begin(stage, true);
#pragma unroll
        for (int d = 0; d < G; ++d) issue_piece(d);
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment read offsets (bytes).  MFMA k-step s (16 deep), lane-half h: k = 16s + 8h + [0,8)
    //   A: two 16-byte slots 4s+2h, 4s+2h+1 of row (lane&31);  B: one slot 2s+h of row (lane&31) per plane
    const int frow = lane & 31, fkh = lane >> 5;
    const int a_sw = (frow >> 1) & 7, b_sw = (frow >> 2) & 3;
    int a_foff[2][2], b_foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if constexpr (GP) {      // first terms in the 16-byte slots 0..3 of the row, second terms in 4..7: k-step s, half h -> slot 2s+h (+4)
            a_foff[s][0] = frow * 128 + (((2 * s + fkh) ^ a_sw) << 4);
            a_foff[s][1] = frow * 128 + (((4 + 2 * s + fkh) ^ a_sw) << 4);
        } else {
            a_foff[s][0] = frow * 128 + (((4 * s + 2 * fkh) ^ a_sw) << 4);
            a_foff[s][1] = frow * 128 + (((4 * s + 2 * fkh + 1) ^ a_sw) << 4);
        }
        b_foff[s] = frow * 64 + (((2 * s + fkh) ^ b_sw) << 4);
    }

    // f16x2: activation scale PER IMAGE = the power of two that puts the image's maximum (tracked by its producers,
    // amax_track in common.h) into [2^13, 2^14); a lane keeps the scale of the image of each of its TM tile rows.
    // Both scales are exact powers of two.
    float sa[TM], inv_sa[TM], xmax_up[TM];      // (xmax_up: a power of two above the image's tracked max|x|)
#pragma unroll
    for (int i = 0; i < TM; ++i) sa[i] = inv_sa[i] = xmax_up[i] = 1.0f;
    // slab variant, "fetch context" of the chunk whose operands are being read: taps inside the image for each of this
    // lane's fragment rows (bit r*3 + s), the scale with the padding taps zeroed, the slab and the tap's row shift
    unsigned f_ok[TM];
    float f_sa[TM];
    int f_slab = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        f_ok[i] = 0;
        f_sa[i] = 0.0f;
    }
    auto fetch_ctx = [&](int kc) {               // kc: chunk index in the whole reduction (tap = kc % 9)
        const int tap = kc % 9, s_tap = tap % 3;
        f_slab = ((kc - kc_begin) / 3 & 1) * SLAB_BYTES;
        const int t0 = frow + s_tap, sw = (t0 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a_foff[ks][0] = t0 * 128 + (((4 * ks + 2 * fkh) ^ sw) << 4);
            a_foff[ks][1] = t0 * 128 + (((4 * ks + 2 * fkh + 1) ^ sw) << 4);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) f_sa[i] = ((f_ok[i] >> tap) & 1u) ? sa[i] : 0.0f;
    };

    // One "k-step" = 16 reduction elements = one MFMA depth; a 32-deep chunk is two k-steps.
    // The wave is software-pipelined over k-steps BY HAND: the instruction stream of a step is a
    // sequence of NM slots { one MFMA of step g ; at most one LDS read for step g+1 ; a few VALU
    // instructions of the split of step g+1's A fragments }, fenced with sched_barrier so that hipcc
    // keeps that order (left alone it emits read -> split -> MFMA phases and the matrix pipe idles
    // during the split; sched_group_barrier pipelines were not honoured in a region this large).
    // Split of one (a, b) pair = 5 dependent stages (cvt | shift,and,sub,sub | cvt | ... | cvt); the
    // stages of the 4*TM pairs of a step are issued stage-major so neighbours are independent.
    struct Frag {        // operands of one k-step
        uintx4 a[TM][NP];        // A: NP 8-element terms (bf16x8 / f16x8 bits) per 32-row tile
        uintx4 b[NP][TN];        // B: plane x 32-column tile
    };
    constexpr int NPROD = F16 ? 3 : 6;        // partial products per multiply-add
    constexpr int NST = F16 ? 3 : 5;          // dependent split stages per pair
    constexpr int NM = NPROD * TM * TN;       // MFMAs per k-step
    constexpr int NRA = 2 * TM, NRB = NP * TN; // LDS reads per k-step
    constexpr int NSL = GP ? 0 : NST * 4 * TM;         // split stages per k-step (none when the input arrives pre-split)
    // slot plan: LDS reads first (RPS per slot; the raw A rows lead), split stages from slot LEAD on (PER per slot),
    // DMA pieces spread over the step (DPS per hosting slot)
    constexpr int NR = NRA + NRB;
    constexpr int RPS = (NR + NM - 1) / NM;
    constexpr int LEAD0 = (NRA + RPS - 1) / RPS + 1;
    constexpr int LEAD = LEAD0 < NM ? LEAD0 : NM - 1;
    constexpr int PER = (NSL + (NM - LEAD) - 1) / (NM - LEAD);
    constexpr int DPS = (G + NM - 1) / NM;
    constexpr int ND = (G + DPS - 1) / DPS;            // slots that host DMA pieces
    constexpr int DSTRIDE = NM / ND;
    static_assert(NM > LEAD && ND * DSTRIDE <= NM && DSTRIDE >= 1, "slots");
    auto step = [&](const Frag &cur, Frag &nxt, int stage, auto s_tag, auto dma_tag) {
        constexpr int s = decltype(s_tag)::value;
        constexpr bool DMA = decltype(dma_tag)::value;   // this step also carries the DMA pieces of a chunk      // k-step (0/1) of the chunk the NEXT operands come from
        // partial products (piece of A, piece of B), smallest first; f16x2 uses the last three with pieces {0,1}
        constexpr int ta[6] = {2, 1, 0, 1, 0, 0}, tb[6] = {0, 1, 2, 0, 1, 0};
        constexpr int T0 = 6 - NPROD;
        const char *a_ptr = SLAB ? smem + f_slab + wm * WM * 128 : smem + stage * STAGE + wm * WM * 128;
        const char *b_ptr = smem + STAGE_BASE + stage * STAGE + A_BYTES + wn * WN * 64;
        floatx4 raw[TM][2];
        float ra[TM][4], rb[TM][4];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (PPY_X3_ABL < 7) {   // MFMA m, term-major: consecutive MFMAs use different accumulators
                const int t = T0 + m / (TM * TN), i = (m / TN) % TM, j = m % TN;
                if constexpr (F16)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur.a[i][ta[t]]),
                                                                      __builtin_bit_cast(f16x8, cur.b[tb[t]][j]), acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.a[i][ta[t]]),
                                                                       __builtin_bit_cast(bf16x8, cur.b[tb[t]][j]), acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < RPS; ++u) {
                const int r = m * RPS + u;
                if ((PPY_X3_ABL >= 5 && PPY_X3_ABL != 6) || r >= NR) {
                } else if (r < NRA) {
                    if constexpr (GP)       // (tile r>>1, term r&1): the finished operand
                        nxt.a[r >> 1][r & 1] = *reinterpret_cast<const uintx4 *>(a_ptr + (r >> 1) * 32 * 128 + a_foff[s][r & 1]);
                    else
                        raw[r >> 1][r & 1] = *reinterpret_cast<const floatx4 *>(a_ptr + (r >> 1) * 32 * 128 + a_foff[s][r & 1]);
                } else {
                    const int pl = (r - NRA) / TN, j = (r - NRA) % TN;
                    nxt.b[pl][j] = *reinterpret_cast<const uintx4 *>(b_ptr + (pl * BN + j * 32) * 64 + b_foff[s]);
                }
            }
            if (DMA && PPY_X3_ABL != 1 && (PPY_X3_ABL < 3 || PPY_X3_ABL >= 6) && m % DSTRIDE == DSTRIDE - 1 && m / DSTRIDE < ND) {
#pragma unroll
                for (int u = 0; u < DPS; ++u)
                    if ((m / DSTRIDE) * DPS + u < G) issue_piece((m / DSTRIDE) * DPS + u);
            }
            if (m >= LEAD) {
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int sl = (m - LEAD) * PER + u;
                    if (sl < NSL && PPY_X3_ABL != 2 && (PPY_X3_ABL < 3 || PPY_X3_ABL == 6)) {
                        const int st = sl / (4 * TM), pr = sl % (4 * TM), i = pr / 4, q = pr % 4;
                        const float xa = raw[i][q >> 1][(q & 1) * 2], xb = raw[i][q >> 1][(q & 1) * 2 + 1];
                        if constexpr (F16) {
                            const float sc_i = SLAB ? f_sa[i] : sa[i];
                            if (st == 0) {
                                nxt.a[i][0][q] = cvt_pk_f16(xa * sc_i, xb * sc_i);
                            } else if (st == 1) {     // residual of the SCALED value: fma(x, sa, -a0) is exact
                                const unsigned P = nxt.a[i][0][q];
                                ra[i][q] = fmaf(xa, sc_i, -f16_lo(P));
                                rb[i][q] = fmaf(xb, sc_i, -f16_hi(P));
                            } else {
                                nxt.a[i][1][q] = cvt_pk_f16(ra[i][q], rb[i][q]);
                            }
                        } else if (st == 0) {
                            nxt.a[i][0][q] = cvt_pk_bf16(xa, xb);
                        } else if (st == 1) {
                            const unsigned P = nxt.a[i][0][q];
                            ra[i][q] = xa - __uint_as_float(P << 16);
                            rb[i][q] = xb - __uint_as_float(P & 0xffff0000u);
                        } else if (st == 2) {
                            nxt.a[i][1][q] = cvt_pk_bf16(ra[i][q], rb[i][q]);
                        } else if (st == 3) {
                            const unsigned P = nxt.a[i][1][q];
                            ra[i][q] -= __uint_as_float(P << 16);
                            rb[i][q] -= __uint_as_float(P & 0xffff0000u);
                        } else {
                            nxt.a[i][2][q] = cvt_pk_bf16(ra[i][q], rb[i][q]);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // pin the split results here: without a use in this block the optimiser sinks the VALU work of the
        // loop's last step across the back edge, in front of the next iteration's first MFMA
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) asm volatile("" : "+v"(nxt.a[i][pl]));
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;

    if (p.trace) c_loop = __builtin_amdgcn_s_memtime();
    if (nchunks > 0) {
        Frag f0, f1;
        // NS = p.nstages chunks are kept resident / in flight (2..4).  All NS stage slots are requested here, slots past
        // the end of the reduction as out-of-range dummies, so that the number of outstanding DMA pieces is the same
        // (NS-1)*G at every mid-chunk wait below.
        const int NS = p.nstages;
        if constexpr (SLAB) {      // the first slab, in front of (= older than) the NS weight groups
            const int cc = kc_begin / 9, r = (kc_begin - cc * 9) / 3;
            cur_ra = __builtin_amdgcn_make_buffer_rsrc((void *)(reinterpret_cast<const char *>(p.x) + (long long)cc * 128), 0,
                                                       0xFFFFFF00u, 0x00020000);
            sl_have = true;
            sl_pb = m0 - 1 + (r - 1) * p.W;
            sl_dst = 0;
#pragma unroll
            for (int pos = 0; pos < 2; ++pos) {
                sl_pos = pos;
#pragma unroll
                for (int d = 0; d < AS; ++d) slab_piece(d);
            }
            sl_q = -2;                 // = (k + L) of the first prologue group: k = -NS, L = NS - 2
        }
        for (int sidx = 0; sidx < NS; ++sidx) {
            issue_begin(sidx, sidx < nchunks);
#pragma unroll
            for (int d = 0; d < G; ++d) issue_piece(d);
        }
        if constexpr (F16) {       // after the first DMA requests, so that this latency (a division, 8 loads) hides behind theirs
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mrow = min(m0 + wm * WM + i * 32 + (lane & 31), p.M - 1);
                if constexpr (GP) {        // the scale its producer chose for this image
                    sa[i] = p.xscale[mrow / hw];
                    inv_sa[i] = pow2_inverse(sa[i]);
                    // (a link in a CHAIN of pre-split tensors bounds its own output from the input's TRACKED maximum, not from the
                    // input's bound: static bounds multiplied along a chain lose 2^6 per link and underflow the fp16 range)
                    if (p.yscale) xmax_up[i] = pow2_above(conv_amax_in(p, mrow / hw));
                    continue;
                }
                const float mx = conv_amax_in(p, mrow / hw);
                const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);      // biased exponent: mx in [2^(e-127), 2^(e-126))
                int f = 267 - e;                                                // biased exponent of 2^(13-(e-127))
                f = f < 103 ? 103 : (f > 167 ? 167 : f);                       // scale in [2^-24, 2^40]: all-zero / absurd tensors stay finite
                sa[i] = __uint_as_float((unsigned)f << 23);
                inv_sa[i] = __uint_as_float((unsigned)(254 - f) << 23);
                xmax_up[i] = 16384.0f * inv_sa[i];             // max|x| < 2^14 / sa
                if constexpr (SLAB) {          // taps of this fragment row that lie inside the image
                    const int mr = m0 + wm * WM + i * 32 + (lane & 31);
                    const int n = mrow / hw, rem = mrow - n * hw;
                    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    unsigned colmask = 0, okb = 0;
                    for (int s2 = 0; s2 < 3; ++s2)
                        if ((unsigned)(wo - 1 + s2) < (unsigned)p.W) colmask |= 1u << s2;
                    for (int r = 0; r < 3; ++r)
                        if ((unsigned)(ho - 1 + r) < (unsigned)p.H) okb |= colmask << (r * 3);
                    f_ok[i] = mr < p.M ? okb : 0u;
                }
            }
            if constexpr (SLAB) fetch_ctx(kc_begin);
        }
        // chunk 0 has landed once at most the NS-1 later requests are outstanding
        if (NS == 2) {
            wait_vmcnt<G>();
        } else if (NS == 3) {
            wait_vmcnt<2 * G>();
        } else {
            wait_vmcnt<3 * G>();
        }
        __builtin_amdgcn_s_barrier();
        {   // operands of (chunk 0, k-step 0): not overlapped with anything
            const char *a_ptr = smem + wm * WM * 128;          // (slab variant: slab 0 starts at 0 as well)
            const char *b_ptr = smem + STAGE_BASE + A_BYTES + wn * WN * 64;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (GP) {
                    f0.a[i][0] = *reinterpret_cast<const uintx4 *>(a_ptr + i * 32 * 128 + a_foff[0][0]);
                    f0.a[i][1] = *reinterpret_cast<const uintx4 *>(a_ptr + i * 32 * 128 + a_foff[0][1]);
                    continue;
                }
                const floatx4 lo = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * 128 + a_foff[0][0]);
                const floatx4 hi = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * 128 + a_foff[0][1]);
                if constexpr (F16) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float xa = q < 2 ? lo[2 * q] : hi[2 * q - 4], xb = q < 2 ? lo[2 * q + 1] : hi[2 * q - 3];
                        const float sc_i = SLAB ? f_sa[i] : sa[i];
                        const unsigned P0 = cvt_pk_f16(xa * sc_i, xb * sc_i);
                        f0.a[i][0][q] = P0;
                        f0.a[i][1][q] = cvt_pk_f16(fmaf(xa, sc_i, -f16_lo(P0)), fmaf(xb, sc_i, -f16_hi(P0)));
                    }
                } else {
                    bf16x8 t3[3];
                    split8(lo, hi, t3);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) f0.a[i][pl] = __builtin_bit_cast(uintx4, t3[pl]);
                }
            }
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    f0.b[pl][j] = *reinterpret_cast<const uintx4 *>(b_ptr + (pl * BN + j * 32) * 64 + b_foff[0]);
        }
        int st = 0;
        for (int k = 0; k < nchunks; ++k) {
            // k-step 0 of chunk k  ||  fetch + split k-step 1
            step(f0, f1, st, S1(), std::false_type());
            // mid-chunk: every LDS read of chunk k has returned (here and, after the barrier, in all waves)
            // and chunk k+1 -- issued one chunk ago -- has landed: refill this stage with chunk k+2
            if (NS == 2) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            } else if (NS == 3) {               // chunk k+2 may still be in flight
                wait_vmcnt<G>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {                            // chunks k+2, k+3
                wait_vmcnt<2 * G>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#if PPY_X3_ABL < 4
            __builtin_amdgcn_s_barrier();
#endif
            issue_begin(st, k + NS < nchunks);
            // k-step 1 of chunk k  ||  fetch + split k-step 0 of chunk k+1 (garbage after the last chunk, unused)
            st = st + 1 == NS ? 0 : st + 1;
            if constexpr (SLAB) fetch_ctx(kc_begin + k + 1);
            step(f1, f0, st, S0(), std::true_type());
        }
        // the epilogue reuses the LDS: all reads returned, and the (out-of-range, zero-filling) DMA pieces the
        // branch-free tail of the loop still issued have landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (p.trace) c_epi = __builtin_amdgcn_s_memtime();
    // back to the unscaled sum (the weight scale is folded into p.scale by the caller): the inverse scale of tile row r
    // sits in lane r.  Vector epilogue: applied after the transposition to the 4 rows per tile a lane finishes;
    // scalar epilogue (K % 4 != 0, rare): applied to the accumulators, element e = row (e&3) + 8(e>>2) + 4(lane>>5).
    if constexpr (BNS) tile_bn_stats<TM, TN, WM, WN>(p, acc, inv_sa, m0, n0, wm, wn, lane, tile_m * (BM / WM) + wm);
    float rowscale[TM][4], rowsplit[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) rowsplit[i][t] = 0.f;
    bool split_out = false;
    if constexpr (F16 && VEC && !SPLIT) {
        // this launch's output goes to ONE consumer as finished operands (ConvArgs::yscale): per-image scale from the static bound
        // |y| <= ysplit_mul * max|x| + ysplit_add, with max|x| < 2^14 / sa (the tracked maximum rounded up to a power of two)
        split_out = p.yscale != nullptr;
        if (split_out) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float ys = split_scale_of(fmaf(p.ysplit_mul, xmax_up[i], p.ysplit_add));
                const int mr = m0 + wm * WM + i * 32 + (lane & 31);
                if (n0 == 0 && wn == 0 && lane < 32 && mr < p.M) p.yscale[mr / hw] = ys;      // (every writer of an image writes the same value)
#pragma unroll
                for (int t = 0; t < 4; ++t) rowsplit[i][t] = __shfl(ys, (lane >> 3) + 8 * t);
            }
        }
    }
    if constexpr (F16) {
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
                                              (F16 && VEC) ? rowscale : nullptr, (VEC && !SPLIT) ? rowsplit : nullptr, split_out);
    if (p.trace && tid == 0) {     // debug timeline (ppy_debug_set_trace): wall-clock span + shader-clock phases
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned long long c_end = __builtin_amdgcn_s_memtime();
        const long long b = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        p.trace[b * 4 + 0] = t_start;
        p.trace[b * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        p.trace[b * 4 + 2] = hwid | ((unsigned long long)xcc << 32);
        // 3 x 21-bit phase lengths in units of 16 shader cycles: setup | main loop | epilogue
        const unsigned long long a = (c_loop - c_start) >> 4, m = (c_epi - c_loop) >> 4, e = (c_end - c_epi) >> 4;
        p.trace[b * 4 + 3] = (a & 0x1fffff) | ((m & 0x1fffff) << 21) | ((e & 0x1fffff) << 42);
    }
#endif
}

struct X3Cfg {
    int bm, bn, wm, wn;
};
constexpr X3Cfg kX3[] = {     // the same nine tiles for both schemes (LDS sizes for bf16x3 / f16x2, two stages)
    {256, 128, 64, 128},   // 0  4 waves (4x1), 112 / 96 KB
    {128, 128, 64, 64},    // 1  4 waves (2x2), 80 / 64 KB
    {128, 128, 32, 128},   // 2  4 waves (4x1), 80 / 64 KB
    {256, 64, 64, 64},     // 3  4 waves (4x1), 88 / 80 KB
    {128, 64, 32, 64},     // 4  4 waves (4x1), 56 / 48 KB
    {256, 128, 64, 64},    // 5  8 waves (4x2), 112 / 96 KB
    {128, 256, 64, 128},   // 6  4 waves (2x2), 128 / 96 KB
    {64, 128, 32, 64},     // 7  4 waves (2x2), 64 / 48 KB
    {64, 64, 32, 32},      // 8  4 waves (2x2), 40 / 32 KB: four workgroups per CU for the store-bound 1x1 expand layers
};
constexpr int kNumX3 = sizeof(kX3) / sizeof(kX3[0]);

template <int BM, int BN, int WM, int WN, bool F16, bool SPLIT, bool VEC, bool SLAB = false, bool BNS = false, bool GP = false>
int launch_x3_one(const ConvArgs &p, int splits, size_t lds, int tiles, hipStream_t stream) {
    if constexpr (F16 && !SPLIT && !BNS && !GP) {
        if (p.bn_part) return launch_x3_one<BM, BN, WM, WN, F16, SPLIT, VEC, SLAB, true>(p, splits, lds, tiles, stream);
    }
    if constexpr (F16 && !SPLIT && VEC && !SLAB && !BNS && !GP) {
        if (p.xscale) return launch_x3_one<BM, BN, WM, WN, F16, SPLIT, VEC, SLAB, false, true>(p, splits, lds, tiles, stream);
    }
    auto k = conv_igemm_x3_kernel<BM, BN, WM, WN, F16, SPLIT, VEC, SLAB, BNS, GP>;
    static PpyLdsAttr attr;      // (the stage count is a launch parameter: allow the whole LDS)
    if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), 160 * 1024) != PPY_OK) return PPY_ERR_LAUNCH;
    hipLaunchKernelGGL(k, dim3(tiles, splits), dim3(64 * (BM / WM) * (BN / WN)), lds, stream, p);
    return PPY_OK;
}

template <int BM, int BN, int WM, int WN, bool F16, bool SLAB = false>
int launch_x3(ConvArgs p, int splits, hipStream_t stream, int nstages = 2) {
    // 32-bit per-lane DMA offsets
    constexpr int NP = F16 ? 2 : 3;
    const long long xbytes = (long long)p.N * p.H * p.W * p.x_ld * 4 + (long long)(p.pad * p.W + p.pad) * p.x_ld * 4;
    const long long wbytes = (long long)p.K * p.Kred * 2 * NP;
    if (xbytes >= 0xFFFFF000LL || wbytes >= 0xFFFFF000LL || p.R * p.S > 32) return PPY_ERR_UNSUPPORTED;
    constexpr int NW = (BM / WM) * (BN / WN);
    // as many LDS stages as asked for, as far as 160 KB of LDS and the 6-bit vmcnt allow (ids of deeper variants of a big
    // tile then alias the deepest one that fits)
    constexpr int SLAB_PIECES = BM / 8 + 1, AS = ((SLAB_PIECES + NW - 1) / NW + 1) / 2;
    constexpr int FIXED = SLAB ? 2 * SLAB_PIECES * 1024 + NW * 1024 : 0;        // two slabs + the dummy pieces' landing strip
    constexpr int STAGE_BYTES = (SLAB ? 0 : BM * 128) + NP * BN * 64, PIECES = (SLAB ? AS : BM / (8 * NW)) + NP * BN / (16 * NW);
    if (SLAB) {
        // slab reuse: 3x3 / stride 1 / pad 1 ("same") layers only; two slabs alternate, which covers 2 or 3 stages
        // (BAD_ARG, not UNSUPPORTED: the caller of an explicit slab id gets an error instead of the fp32 fallback kernel)
        if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || p.Ho != p.H || p.Wo != p.W) return PPY_ERR_BAD_ARG;
        if (nstages > 3) nstages = 3;
    }
    while (nstages > 2 && (FIXED + nstages * STAGE_BYTES > 160 * 1024 || (nstages - 1) * PIECES + (SLAB ? 2 * AS : 0) > 63)) --nstages;
    if (FIXED + nstages * STAGE_BYTES > 160 * 1024) return PPY_ERR_UNSUPPORTED;
    size_t lds = (size_t)FIXED + (size_t)nstages * STAGE_BYTES;
    p.nstages = nstages;
    const size_t epi = (size_t)NW * 32 * LDS_LD * sizeof(float);
    if (lds < epi) lds = epi;
    p.chunks_total = p.R * p.S * (p.C / 32);
    p.chunks_per_split = ceil_div(p.chunks_total, splits);
    if (SLAB) p.chunks_per_split = ceil_div(p.chunks_per_split, 3) * 3;      // a split starts on a (channel chunk, r) boundary
    splits = ceil_div(p.chunks_total, p.chunks_per_split);
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.K, BN);
    p.panel_n = ppy_panel_n(p, BM, BN, splits);
    const bool vec = vec_epilogue_ok(p);
    if (p.bn_part) {         // BatchNorm statistics from the epilogue: f16x2, one split, plain conv + bias
        if (!F16 || splits > 1 || p.res || p.posb || p.ups || p.act != PPY_ACT_NONE) return PPY_ERR_UNSUPPORTED;
        if (ceil_div(p.M, BM) * (BM / WM) > p.bn_capacity) return PPY_ERR_WORKSPACE;
        if (p.bn_slices_host) *p.bn_slices_host = ceil_div(p.M, BM) * (BM / WM);
    }
    // pre-split tensors (ConvArgs::xscale / yscale): f16x2 tiles with one split and the vector epilogue only -- an explicit request
    // that cannot be honoured is an error (BAD_ARG), never a silently different interpretation of the bytes
    if ((p.xscale || p.yscale) && (!F16 || SLAB || splits > 1 || !vec || p.bn_part)) return PPY_ERR_BAD_ARG;
    if (p.yscale && p.ups) return PPY_ERR_BAD_ARG;          // (a pre-split INPUT with an upsampled fp32 store is fine)
    if (p.yscale && (p.K % 32 != 0 || p.y_ld % 32 != 0)) return PPY_ERR_BAD_ARG;
    // Round 6: the SCALAR epilogue (K % 4 != 0 or unaligned rows: rare, never a layer of the plan) exists for wave tiles of up to four
    // 32x32 accumulator tiles only.  With six or eight the fully unrolled scalar store loop spills (416-576 bytes of scratch, no
    // AGPRs), and hipcc (ROCm 7.2) then mixed up two of its row predicates: 192x256 / N3 C96 K258 6x25 left acc[1][1][10] of the last
    // tile_m unstored, deterministically (tools/experiments/r06_case11.py).  Such a launch is refused; ppy_conv2d_bn_act_f32 then takes
    // the exact-fp32 fall-back tile, as for any configuration that does not support a shape.
    constexpr bool BIG_WAVE_TILE = (WM / 32) * (WN / 32) >= 6;
    if (BIG_WAVE_TILE && !vec) return PPY_ERR_UNSUPPORTED;
    int rc;
    if (splits > 1) {
        if constexpr (BIG_WAVE_TILE)
            rc = launch_x3_one<BM, BN, WM, WN, F16, true, true, SLAB>(p, splits, lds, tiles, stream);
        else
            rc = vec ? launch_x3_one<BM, BN, WM, WN, F16, true, true, SLAB>(p, splits, lds, tiles, stream)
                     : launch_x3_one<BM, BN, WM, WN, F16, true, false, SLAB>(p, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
        launch_splitk_reduce(p, splits, vec, stream);
    } else {
        if constexpr (BIG_WAVE_TILE)
            rc = launch_x3_one<BM, BN, WM, WN, F16, false, true, SLAB>(p, splits, lds, tiles, stream);
        else
            rc = vec ? launch_x3_one<BM, BN, WM, WN, F16, false, true, SLAB>(p, splits, lds, tiles, stream)
                     : launch_x3_one<BM, BN, WM, WN, F16, false, false, SLAB>(p, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
    }
    return ppy_launch_status();
}

int dispatch_slab(const ConvArgs &p, int c, int s, hipStream_t st, int ns) {
    switch (c) {
        case 0: return launch_x3<256, 128, 64, 128, true, true>(p, s, st, ns);
        case 1: return launch_x3<128, 128, 64, 64, true, true>(p, s, st, ns);
        case 2: return launch_x3<128, 128, 32, 128, true, true>(p, s, st, ns);
        case 3: return launch_x3<256, 64, 64, 64, true, true>(p, s, st, ns);
        case 4: return launch_x3<128, 64, 32, 64, true, true>(p, s, st, ns);
        case 5: return launch_x3<256, 128, 64, 64, true, true>(p, s, st, ns);
        case 6: return launch_x3<128, 256, 64, 128, true, true>(p, s, st, ns);
        case 7: return launch_x3<64, 128, 32, 64, true, true>(p, s, st, ns);
        case 8: return launch_x3<64, 64, 32, 32, true, true>(p, s, st, ns);
    }
    return PPY_ERR_BAD_ARG;
}

// f16x2 tiles with 96 / 192 rows (three 32-row MFMA tiles per wave): the grids of the layers are small against 256 CUs, and a
// tile height of 1.5x fills the last round of workgroups where 128 / 256 rows leave 30-40 % of the slots empty
// (e.g. M = 46208: 361 x 2 tiles of 128x128 on 512 slots = 1.41 rounds; 241 x 2 tiles of 192x128 = 0.94).
int dispatch_extra(const ConvArgs &p, int c, int s, hipStream_t st, int ns) {
    switch (c) {
        case 0: return launch_x3<192, 128, 96, 64, true>(p, s, st, ns);     // 4 waves (2x2), 40 KB per stage
        case 1: return launch_x3<192, 256, 96, 64, true>(p, s, st, ns);     // 8 waves (2x4), 56 KB per stage
        case 2: return launch_x3<96, 256, 96, 64, true>(p, s, st, ns);      // 4 waves (1x4), 44 KB per stage
    }
    return PPY_ERR_BAD_ARG;
}
constexpr int kNumExtra = 3;

template <bool F16>
int dispatch_scheme(const ConvArgs &p, int c, int s, hipStream_t st, int ns = 2) {
    switch (c) {
        case 0: return launch_x3<256, 128, 64, 128, F16>(p, s, st, ns);
        case 1: return launch_x3<128, 128, 64, 64, F16>(p, s, st, ns);
        case 2: return launch_x3<128, 128, 32, 128, F16>(p, s, st, ns);
        case 3: return launch_x3<256, 64, 64, 64, F16>(p, s, st, ns);
        case 4: return launch_x3<128, 64, 32, 64, F16>(p, s, st, ns);
        case 5: return launch_x3<256, 128, 64, 64, F16>(p, s, st, ns);
        case 6: return launch_x3<128, 256, 64, 128, F16>(p, s, st, ns);
        case 7: return launch_x3<64, 128, 32, 64, F16>(p, s, st, ns);
        case 8: return launch_x3<64, 64, 32, 32, F16>(p, s, st, ns);
    }
    return PPY_ERR_BAD_ARG;
}

// One workgroup per output channel: s_w = the power of two that puts max|w[k,:]| into [2^13, 2^14); planes of w*s_w as two
// fp16 terms (RNE each), and the epilogue scale with 1/s_w folded in (exact).
__global__ void __launch_bounds__(256) split_weights_f16_kernel(const float *w, int K, long long kred, const float *scale,
                                                                unsigned short *out, float *scale_out) {
    __shared__ float smax[4];
    const int k = blockIdx.x, tid = threadIdx.x;
    const float *row = w + (long long)k * kred;
    float mx = 0.f;
    for (long long i = tid; i < kred; i += 256) mx = fmaxf(mx, fabsf(row[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) smax[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    int f = 267 - e;
    f = f < 103 ? 103 : (f > 167 ? 167 : f);      // [2^-24, 2^40], as for the activations
    const float sw = __uint_as_float((unsigned)f << 23), inv = __uint_as_float((unsigned)(254 - f) << 23);
    const long long n = (long long)K * kred;
    for (long long i = tid; i < kred; i += 256) {
        const float v = row[i] * sw;
        const _Float16 h0 = (_Float16)v;
        const _Float16 h1 = (_Float16)(v - (float)h0);
        // plane layout [chunk][K][32]: the 16 rows x 64 B that one DMA instruction of the conv kernel fetches are one
        // contiguous KB (8 whole cache lines instead of 16 half lines of rows that are kred*2 bytes apart)
        const long long o = PPY_X3_BBLOCK ? ((i >> 5) * K + k) * 32 + (i & 31) : (long long)k * kred + i;
        out[o] = __builtin_bit_cast(unsigned short, h0);
        out[n + o] = __builtin_bit_cast(unsigned short, h1);
    }
    if (tid == 0) scale_out[k] = scale[k] * inv;
}

}  // namespace

// local ids: [0, 9) bf16x3, [9, 18) f16x2 (two LDS stages), [18, 27) f16x2 with three stages, [27, 36) with four,
// [36, 45) f16x2 with slab reuse (3x3 / stride 1 / pad 1 layers) and two stages, [45, 54) the same with three,
// [54, 57) the 96 / 192-row f16x2 tiles with two stages, [57, 60) with three, [60, 63) with four
int ppy_x3_num_configs() { return 6 * kNumX3 + 3 * kNumExtra; }
int ppy_x3_f16_base() { return kNumX3; }

int ppy_x3_dispatch(const ConvArgs &p, int c, int s, hipStream_t st) {
    if (c < kNumX3) {
        if (!p.w3 || ((uintptr_t)p.w3 & 15) != 0 || p.xscale || p.yscale) return PPY_ERR_BAD_ARG;      // (pre-split tensors: f16x2 only)
        return dispatch_scheme<false>(p, c, s, st);
    }
    // f16x2 needs the split weights + folded scale, the tracked maximum of the input, and a CoordConv bias map
    // pre-multiplied by the per-channel weight scale
    if (!p.wf16 || ((uintptr_t)p.wf16 & 15) != 0 || !p.scale_f16 || !p.amax_in || (p.posb && !p.posb_f16)) return PPY_ERR_BAD_ARG;
    ConvArgs q = p;
    q.scale = p.scale_f16;
    q.posb = p.posb ? p.posb_f16 : nullptr;
    const int local = c - kNumX3;
    if (local >= 5 * kNumX3) return dispatch_extra(q, (local - 5 * kNumX3) % kNumExtra, s, st, 2 + (local - 5 * kNumX3) / kNumExtra);
    if (local >= 3 * kNumX3) return dispatch_slab(q, local % kNumX3, s, st, 2 + (local / kNumX3 - 3));
    return dispatch_scheme<true>(q, local % kNumX3, s, st, 2 + local / kNumX3);
}

extern "C" int ppy_conv2d_split_weights_f16x2(const float *w_krsc, int K, long long kred, const float *scale,
                                              void *out_planes, float *scale_out, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(w_krsc && scale && out_planes && scale_out && K > 0 && kred > 0);
    hipLaunchKernelGGL(split_weights_f16_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, w_krsc, K, kred, scale,
                       (unsigned short *)out_planes, scale_out);
    return ppy_launch_status();
}

extern "C" int ppy_conv2d_split_weights_bf16x3(const float *w, long long n, void *out, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(w && out && n > 0);
    const long long pairs = (n + 1) / 2;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       n, (unsigned short *)out);
    return ppy_launch_status();
}
