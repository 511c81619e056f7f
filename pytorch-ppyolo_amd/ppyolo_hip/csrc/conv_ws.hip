// f16x2 implicit-GEMM convolution with SPECIALISED waves: four waves only deliver operands, four only multiply.
//
// Same operator, same operand layouts in the LDS, same products in the same order and the same epilogue as the f16x2 tiles of
// conv_x3.hip (results are bit-identical).  What changes is who issues what.  In conv_x3.hip every wave drops its share of a
// chunk's LDS-DMA pieces into its own MFMA stream; measured there: operand delivery alone takes 699
// cycles per 32-deep chunk, the MFMAs 768, and together they take 1399 -- the times ADD, because a wave that issues a DMA piece
// into a backed-up memory pipeline stalls in order, its MFMAs included; and ONE extra producer wave could not issue fast enough
// (a piece every ~50 cycles).  Here a workgroup has eight waves: waves 4..7 issue ALL pieces (a quarter each) and wait on their
// own vmcnt, waves 0..3 (2 x 2 over the tile) never execute a vector-memory instruction inside the main loop -- LDS reads, the
// operand split and MFMAs only.  One workgroup barrier per chunk couples the two groups exactly as conv_x3.hip's does: at the
// barrier of chunk k the consumers have read all of chunk k and the producers have seen chunk k+1 land; behind it the
// producers refill chunk k's stage with chunk k+NS.
#include "conv_shared.h"

#include <type_traits>

namespace {

// PRE = true: the producer waves also SPLIT the activations.  A producer wave that has seen its own pieces of a chunk land
// (its counted vmcnt: no other wave involved) reads exactly those rows back from the landing area, scales and splits them into
// the two fp16 terms ONCE and writes them to two plane tiles of the stage; the consumers read finished A fragments (two
// 16-byte LDS reads per 32 rows and k-step, as before) and execute no VALU instruction per operand at all.  In the plain form
// every consumer wave splits the rows of its sub-tile itself -- twice per workgroup (the two waves of a row of the 2 x 2
// layout), in the issue slots of the waves that feed the MFMA pipe.  Same values, same products: bit-identical results.
// GP = true: the input arrives pre-split from its producer (conv_x3.hip, ConvArgs::xscale): same DMA, the consumers' fragment
// reads are the operands.
// KP = true (round 6, "k-parity"; 128-row tiles): EIGHT consumer waves as two groups of 2 x 2 over the SAME 128 x BN tile; group g
// multiplies k-step g (16 deep) of every 32-deep chunk.  A SIMD then holds two consumer waves that cover each other's LDS / barrier
// latencies -- what made the 256-row twelve-wave form the fastest on the big layers -- without a 256-row tile, whose grid is too
// coarse for the 19x19 / 38x38 maps (M = 2888 / 11552 rows at batch 8) and for batch 1.  Measured before (profiles/
// r06_conv_trace_phases.txt): one consumer wave per SIMD runs the 128x128 main loop at 34-46 % of the matrix pipe, two at 77 %.
// The groups' accumulators are added through the LDS when the reduction ends (group 0's + group 1's: one more fp32 rounding than
// the other tiles, like a split-K of two; deterministic), each group then finishes ONE 32-column half of every wave tile -- eight
// waves share the store-bound epilogue instead of four.
// CPS = 2 (k-parity tiles only): a stage holds TWO 32-deep chunks and the workgroup meets at ONE barrier per 64 -- group g multiplies
// k-step g of both.  The small-grid layers these tiles serve are bound by the per-barrier round trip (wait for the fragment reads, meet,
// read, multiply: ~1000 cycles around 200-400 cycles of MFMAs), not by the matrix pipe: twice the work per round trip.
template <int BM, int BN, int NS, bool SPLIT, bool VEC, bool BNS = false, bool PRE = false, bool GP = false, bool KP = false, int CPS = 1>      // (BNS: conv_x3.hip)
__global__ void __launch_bounds__((BM == 256 || KP) ? 768 : 512) conv_igemm_ws_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!KP || (BM <= 128 && !PRE && !BNS), "k-parity consumers: 64- / 128-row tiles, plain / pre-split input, no BatchNorm statistics");
    static_assert(CPS == 1 || (CPS == 2 && KP), "two chunks per stage: the k-parity tiles");
    constexpr int CM = BM == 256 ? 4 : 2, NC = CM * 2 * (KP ? 2 : 1);      // consumer waves: CM x 2 over the tile (256-row tiles: eight; KP: two such groups)
    constexpr int WM = BM / CM, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int NP = 2, B_ROWS = NP * BN, NWP = 4;                 // weight planes, producer waves
    static_assert(BM % (8 * NWP) == 0 && B_ROWS % (16 * NWP) == 0, "whole DMA instructions per producer wave");
    constexpr int A_PASS = BM / (8 * NWP), B_PASS = B_ROWS / (16 * NWP), G = CPS * (A_PASS + B_PASS);
    constexpr int A_BYTES = BM * 128, AP_BYTES = PRE ? 2 * BM * 64 : 0, B_BYTES = B_ROWS * 64;      // fp32 landing area, A planes, B planes
    constexpr int SUB = A_BYTES + AP_BYTES + B_BYTES, STAGE = CPS * SUB;                            // one 32-deep chunk; a stage
    static_assert((NS - 1) * G <= 63, "6-bit vmcnt");
    static_assert(!GP || (!PRE && !SPLIT && VEC && !BNS), "pre-split input: plain consumers, one split, vector epilogue");
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem_ws[];
    char *smem = smem_ws;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.K + BN - 1) / BN;
    int tile_m, tile_n;          // XCD-contiguous ranges of the tile order (conv_shared.h, ppy_tile_of)
    ppy_tile_of(p, tiles_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int kc_begin = split * p.chunks_per_split;
    const int kc_end = min(kc_begin + p.chunks_per_split, p.chunks_total);
    const int nchunks = kc_end - kc_begin;
    const int nsuper = (nchunks + CPS - 1) / CPS;      // stages to walk (CPS chunks each; the tail of an odd reduction is zero-filled)
    const int hw = p.Ho * p.Wo;

    if (wave >= NC) {
        // ================= producers: every LDS-DMA piece of the tile, a quarter per wave =================
        const int pw = wave - NC;
        const unsigned OOB = 0xFFFFFFF0u;
        const long long bias = (long long)(p.pad * p.W + p.pad) * p.x_ld * 4;   // keeps offsets >= 0
        unsigned a_off[A_PASS], a_ok[A_PASS], b_off[B_PASS];
        {
            const int drow = lane >> 3, dslot = lane & 7;
            const int step_rows = 8 * NWP;
            const int step_ho = step_rows / p.Wo, step_wo = step_rows - step_ho * p.Wo;
            int m_first = min(m0 + pw * 8 + drow, p.M - 1);
            int n = m_first / hw, rem = m_first - n * hw;
            int ho = rem / p.Wo, wo = rem - ho * p.Wo;
#pragma unroll
            for (int j = 0; j < A_PASS; ++j) {
                const int row = (j * NWP + pw) * 8 + drow;
                const int scol = dslot ^ ((row >> 1) & 7);
                const int mr = m0 + row;
                const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
                a_off[j] = (unsigned)((((long long)n * p.H + hi0) * p.W + wi0) * p.x_ld * 4 + bias + scol * 16);
                unsigned colmask = 0, okb = 0;
                for (int s2 = 0; s2 < p.S; ++s2)
                    if ((unsigned)(wi0 + s2) < (unsigned)p.W) colmask |= 1u << s2;
                for (int r = 0; r < p.R; ++r)
                    if ((unsigned)(hi0 + r) < (unsigned)p.H) okb |= colmask << (r * p.S);
                a_ok[j] = mr < p.M ? okb : 0u;
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
                const int rb = (j * NWP + pw) * 16 + drow;          // row of the [2*BN] B tile
                const int plane = rb / BN, nrow = rb - plane * BN;
                const int scol = dslot ^ ((rb >> 2) & 3);
                const int k = min(n0 + nrow, p.K - 1);              // rows >= K are masked at store
                b_off[j] = (unsigned)(plane * plane_bytes + (long long)k * 64 + scol * 16);      // [chunk][K][32] planes
            }
        }
        const int RS = p.R * p.S;
        int l_cc = kc_begin / RS, l_tap = kc_begin - l_cc * RS;
        int l_r = l_tap / p.S, l_s = l_tap - l_r * p.S;
        const char *xb = reinterpret_cast<const char *>(p.x) - bias;
        const char *wb = reinterpret_cast<const char *>(p.wf16);
        auto issue_sub = [&](int lds_base, bool have) {
            const long long a_uni = ((long long)(l_r * p.W + l_s) * p.x_ld + l_cc * 32) * 4;
            const long long b_uni = ((long long)l_tap * (p.C / 32) + l_cc) * p.K * 64;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(xb + (have ? a_uni : 0)), 0, 0xFFFFFF00u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(wb + (have ? b_uni : 0)), 0, 0xFFFFFF00u, 0x00020000);
            const unsigned tapbit = have ? (1u << l_tap) : 0u;
            const unsigned lds = (unsigned)(lds_base + pw * 1024);
#pragma unroll
            for (int d = 0; d < A_PASS; ++d) {
                const unsigned off = (a_ok[d] & tapbit) ? a_off[d] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(smem + lds + d * NWP * 1024), 16, off, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < B_PASS; ++j) {
                const unsigned off = have ? b_off[j] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(smem + lds + A_BYTES + AP_BYTES + j * NWP * 1024), 16, off, 0, 0, 0);
            }
            ++l_tap;
            ++l_s;
            if (l_s == p.S) { l_s = 0; ++l_r; }
            if (l_tap == RS) { l_tap = 0; l_r = 0; l_s = 0; ++l_cc; }
        };
        auto issue = [&](int stage, int sup) {          // the CPS chunks of stage index `sup` of this split (beyond the reduction: dummies)
#pragma unroll
            for (int u = 0; u < CPS; ++u) issue_sub(stage * STAGE + u * SUB, sup * CPS + u < nchunks);
        };
        // PRE: the scale of the image of each of this lane's piece rows, and the split of this wave's pieces of a stage
        float sa_p[A_PASS];
        if constexpr (PRE) {
#pragma unroll
            for (int j = 0; j < A_PASS; ++j) {
                const int mrow = min(m0 + (j * NWP + pw) * 8 + (lane >> 3), p.M - 1);
                const float mx = conv_amax_in(p, mrow / hw);
                const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
                int f = 267 - e;
                f = f < 103 ? 103 : (f > 167 ? 167 : f);
                sa_p[j] = __uint_as_float((unsigned)f << 23);
            }
        }
        auto split_stage = [&](int stage) {
            if constexpr (PRE) {
                typedef __attribute__((ext_vector_type(2))) unsigned uintx2;
                char *sb = smem + stage * STAGE;
#pragma unroll
                for (int d = 0; d < A_PASS; ++d) {
                    const int row = (d * NWP + pw) * 8 + (lane >> 3);
                    const int g4 = (lane & 7) ^ ((row >> 1) & 7);          // the 4-channel group this lane's 16 bytes hold (source-side swizzle)
                    const floatx4 v = *reinterpret_cast<const floatx4 *>(sb + pw * 1024 + d * NWP * 1024 + lane * 16);
                    const float sc_r = sa_p[d];
                    const unsigned h0 = cvt_pk_f16(v[0] * sc_r, v[1] * sc_r), h1 = cvt_pk_f16(v[2] * sc_r, v[3] * sc_r);
                    const unsigned l0 = cvt_pk_f16(fmaf(v[0], sc_r, -f16_lo(h0)), fmaf(v[1], sc_r, -f16_hi(h0)));
                    const unsigned l1 = cvt_pk_f16(fmaf(v[2], sc_r, -f16_lo(h1)), fmaf(v[3], sc_r, -f16_hi(h1)));
                    // plane row = 32 fp16 = 64 B, 16-byte slot c (channels 8c .. 8c+7) stored at c ^ ((row>>2)&3)
                    const int o = row * 64 + (((g4 >> 1) ^ ((row >> 2) & 3)) << 4) + (g4 & 1) * 8;
                    *reinterpret_cast<uintx2 *>(sb + A_BYTES + o) = uintx2{h0, h1};
                    *reinterpret_cast<uintx2 *>(sb + A_BYTES + BM * 64 + o) = uintx2{l0, l1};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        };
        if (nchunks > 0) {
            // NS stage slots are requested up front, slots past the end of the reduction as out-of-range dummies, so that the
            // number of outstanding pieces is the same at every wait below
#pragma unroll
            for (int sidx = 0; sidx < NS; ++sidx) issue(sidx, sidx);
            wait_vmcnt<(NS - 1) * G>();                  // chunk 0 has landed
            split_stage(0);
            __builtin_amdgcn_s_barrier();
            int st = 0;
            for (int k = 0; k < nsuper; ++k) {
                wait_vmcnt<(NS - 2) * G>();              // chunk k+1 has landed (this wave's pieces; the barrier makes it all of them)
                split_stage(st + 1 == NS ? 0 : st + 1);
                __builtin_amdgcn_s_barrier();            // ... and the consumers have read all of chunk k
                issue(st, k + NS);
                st = st + 1 == NS ? 0 : st + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ================= consumers: 2 x 2 waves over the tile; LDS reads, operand split, MFMAs =================
    const int grp = KP ? wave >> 2 : 0;              // KP: the k-step of every chunk this wave multiplies
    const int wq = KP ? wave & 3 : wave;
    const int wm = wq >> 1, wn = wq & 1;
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
        if constexpr (GP) {      // first terms in the 16-byte slots 0..3 of the row, second terms in 4..7 (conv_x3.hip)
            a_foff[s][0] = frow * 128 + (((2 * s + fkh) ^ a_sw) << 4);
            a_foff[s][1] = frow * 128 + (((4 + 2 * s + fkh) ^ a_sw) << 4);
        } else {
            a_foff[s][0] = frow * 128 + (((4 * s + 2 * fkh) ^ a_sw) << 4);
            a_foff[s][1] = frow * 128 + (((4 * s + 2 * fkh + 1) ^ a_sw) << 4);
        }
        b_foff[s] = frow * 64 + (((2 * s + fkh) ^ b_sw) << 4);
    }
    int ap_foff[2];                       // PRE: A planes, row (lane&31), slot 2s+h at (2s+h) ^ ((row>>2)&3)
#pragma unroll
    for (int s = 0; s < 2; ++s) ap_foff[s] = frow * 64 + (((2 * s + fkh) ^ b_sw) << 4);
    float sa[TM], inv_sa[TM], xmax_up[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {        // per-image activation scale (conv_x3.hip)
        xmax_up[i] = 1.0f;
        const int mrow = min(m0 + wm * WM + i * 32 + (lane & 31), p.M - 1);
        if constexpr (GP) {
            sa[i] = p.xscale[mrow / hw];
            inv_sa[i] = pow2_inverse(sa[i]);
            if (p.yscale) xmax_up[i] = pow2_above(conv_amax_in(p, mrow / hw));      // (conv_x3.hip: chains bound from the tracked maximum)
            continue;
        }
        const float mx = conv_amax_in(p, mrow / hw);
        const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
        int f = 267 - e;
        f = f < 103 ? 103 : (f > 167 ? 167 : f);
        sa[i] = __uint_as_float((unsigned)f << 23);
        inv_sa[i] = __uint_as_float((unsigned)(254 - f) << 23);
        xmax_up[i] = 16384.0f * inv_sa[i];
    }

    struct Frag {        // operands of one k-step
        uintx4 a[TM][NP];
        uintx4 b[NP][TN];
    };
    constexpr int NM = 3 * TM * TN;            // MFMAs per k-step
    constexpr int NRA = 2 * TM, NRB = NP * TN, NR = NRA + NRB;
    constexpr int NSL = (PRE || GP) ? 0 : 3 * 4 * TM;  // split stages per k-step (3 dependent stages x 4 pairs x TM)
    constexpr int RPS = (NR + NM - 1) / NM;
    constexpr int LEAD0 = (NRA + RPS - 1) / RPS + 1;
    constexpr int LEAD = LEAD0 < NM ? LEAD0 : NM - 1;
    constexpr int PER = (NSL + (NM - LEAD) - 1) / (NM - LEAD);
    // One k-step: NM slots { one MFMA of step g ; at most RPS LDS reads for step g+1 ; PER stages of the split of step g+1's A
    // fragments }, fenced so that hipcc keeps that order (conv_x3.hip's step() without the DMA pieces)
    auto step = [&](const Frag &cur, Frag &nxt, int stage, const int ao0, const int ao1, const int bo, const int apo) {
        // ao0 / ao1 / bo / apo: fragment offsets of the k-step (0/1) of the chunk the NEXT operands come from (a_foff[s][.], b_foff[s], ap_foff[s])
        constexpr int ta[3] = {1, 0, 0}, tb[3] = {0, 1, 0};      // a1*b0, a0*b1, a0*b0: smallest first
        const char *a_ptr = PRE ? smem + stage * STAGE + A_BYTES + wm * WM * 64 : smem + stage * STAGE + wm * WM * 128;
        const char *b_ptr = smem + stage * STAGE + A_BYTES + AP_BYTES + wn * WN * 64;
        floatx4 raw[TM][2];
        float ra[TM][4], rb[TM][4];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            {
                const int t = m / (TM * TN), i = (m / TN) % TM, j = m % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur.a[i][ta[t]]),
                                                                  __builtin_bit_cast(f16x8, cur.b[tb[t]][j]), acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < RPS; ++u) {
                const int r = m * RPS + u;
                if (r >= NR) {
                } else if (r < NRA) {
                    if constexpr (PRE)      // (tile r>>1, plane r&1)
                        nxt.a[r >> 1][r & 1] = *reinterpret_cast<const uintx4 *>(a_ptr + (r & 1) * BM * 64 + (r >> 1) * 32 * 64 + apo);
                    else if constexpr (GP)  // (tile r>>1, term r&1) of the pre-split rows
                        nxt.a[r >> 1][r & 1] = *reinterpret_cast<const uintx4 *>(a_ptr + (r >> 1) * 32 * 128 + ((r & 1) ? ao1 : ao0));
                    else
                        raw[r >> 1][r & 1] = *reinterpret_cast<const floatx4 *>(a_ptr + (r >> 1) * 32 * 128 + ((r & 1) ? ao1 : ao0));
                } else {
                    const int pl = (r - NRA) / TN, j = (r - NRA) % TN;
                    nxt.b[pl][j] = *reinterpret_cast<const uintx4 *>(b_ptr + (pl * BN + j * 32) * 64 + bo);
                }
            }
            if (m >= LEAD) {
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int sl = (m - LEAD) * PER + u;
                    if (sl < NSL) {
                        const int st = sl / (4 * TM), pr = sl % (4 * TM), i = pr / 4, q = pr % 4;
                        const float xa = raw[i][q >> 1][(q & 1) * 2], xb = raw[i][q >> 1][(q & 1) * 2 + 1];
                        if (st == 0) {
                            nxt.a[i][0][q] = cvt_pk_f16(xa * sa[i], xb * sa[i]);
                        } else if (st == 1) {     // residual of the SCALED value: fma(x, sa, -a0) is exact
                            const unsigned P = nxt.a[i][0][q];
                            ra[i][q] = fmaf(xa, sa[i], -f16_lo(P));
                            rb[i][q] = fmaf(xb, sa[i], -f16_hi(P));
                        } else {
                            nxt.a[i][1][q] = cvt_pk_f16(ra[i][q], rb[i][q]);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) asm volatile("" : "+v"(nxt.a[i][pl]));
    };
    // this wave's first k-step: 0, or (KP) the one of its group
    const int o_a0 = (KP && grp) ? a_foff[1][0] : a_foff[0][0], o_a1 = (KP && grp) ? a_foff[1][1] : a_foff[0][1];
    const int o_b = (KP && grp) ? b_foff[1] : b_foff[0], o_ap = (KP && grp) ? ap_foff[1] : ap_foff[0];

    if (nchunks > 0) {
        Frag f0, f1;
        __builtin_amdgcn_s_barrier();            // chunk 0 is in the LDS
        {   // operands of (chunk 0, first k-step): not overlapped with anything
            const char *a_ptr = smem + wm * WM * 128;
            const char *b_ptr = smem + A_BYTES + AP_BYTES + wn * WN * 64;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (PRE) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        f0.a[i][pl] = *reinterpret_cast<const uintx4 *>(smem + A_BYTES + pl * BM * 64 + (wm * WM + i * 32) * 64 + o_ap);
                    continue;
                }
                if constexpr (GP) {
                    f0.a[i][0] = *reinterpret_cast<const uintx4 *>(a_ptr + i * 32 * 128 + o_a0);
                    f0.a[i][1] = *reinterpret_cast<const uintx4 *>(a_ptr + i * 32 * 128 + o_a1);
                    continue;
                }
                const floatx4 lo = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * 128 + o_a0);
                const floatx4 hi = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * 128 + o_a1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xa = q < 2 ? lo[2 * q] : hi[2 * q - 4], xb = q < 2 ? lo[2 * q + 1] : hi[2 * q - 3];
                    const unsigned P0 = cvt_pk_f16(xa * sa[i], xb * sa[i]);
                    f0.a[i][0][q] = P0;
                    f0.a[i][1][q] = cvt_pk_f16(fmaf(xa, sa[i], -f16_lo(P0)), fmaf(xb, sa[i], -f16_hi(P0)));
                }
            }
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    f0.b[pl][j] = *reinterpret_cast<const uintx4 *>(b_ptr + (pl * BN + j * 32) * 64 + o_b);
        }
        int st = 0;
        if constexpr (KP && CPS == 2) {
            // a stage = two chunks: the loop of the one-group tiles with the two CHUNKS of a stage in the place of the two k-steps of a
            // chunk -- MFMAs of (stage k, chunk 0) beside the reads of (k, chunk 1), the barrier, MFMAs of (k, chunk 1) beside the reads of
            // (k + 1, chunk 0); this wave's k-step (its group's) in both
            for (int k = 0; k < nsuper; ++k) {
                step(f0, f1, st, o_a0 + SUB, o_a1 + SUB, o_b + SUB, o_ap);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of stage k has returned
                __builtin_amdgcn_s_barrier();
                st = st + 1 == NS ? 0 : st + 1;
                step(f1, f0, st, o_a0, o_a1, o_b, o_ap);
            }
        } else if constexpr (KP) {
            // one k-step per chunk and wave: behind the barrier of chunk k + 1 (it has landed; every wave has READ chunk k -- its stage
            // is refilled by the producers from here on, the operands are in registers) the MFMAs of chunk k run beside the
            // fragment reads of chunk k + 1.  Same barrier count as the producers' loop: one per chunk.
            auto sync = [&]() {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                st = st + 1 == NS ? 0 : st + 1;
            };
            int k = 0;
            for (; k + 2 <= nchunks; k += 2) {
                sync();
                step(f0, f1, st, o_a0, o_a1, o_b, o_ap);
                sync();
                step(f1, f0, st, o_a0, o_a1, o_b, o_ap);
            }
            if (k < nchunks) {
                sync();
                step(f0, f1, st, o_a0, o_a1, o_b, o_ap);         // (reads the dummy chunk behind the reduction: unused)
            }
        } else {
            for (int k = 0; k < nchunks; ++k) {
                step(f0, f1, st, a_foff[1][0], a_foff[1][1], b_foff[1], ap_foff[1]);      // k-step 0 of chunk k  ||  fetch + split k-step 1
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of chunk k has returned
                __builtin_amdgcn_s_barrier();
                st = st + 1 == NS ? 0 : st + 1;
                step(f1, f0, st, a_foff[0][0], a_foff[0][1], b_foff[0], ap_foff[0]);      // k-step 1 of chunk k  ||  fetch + split k-step 0 of chunk k+1
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if constexpr (BNS) tile_bn_stats<TM, TN, WM, WN>(p, acc, inv_sa, m0, n0, wm, wn, lane, tile_m * CM + wm);
    float rowscale[TM][4], rowsplit[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) rowsplit[i][t] = 0.f;
    bool split_out = false;
    if constexpr (VEC && !SPLIT) {        // the output goes to one consumer as finished operands (conv_x3.hip, ConvArgs::yscale)
        split_out = p.yscale != nullptr;
        if (split_out) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float ys = split_scale_of(fmaf(p.ysplit_mul, xmax_up[i], p.ysplit_add));
                const int mr = m0 + wm * WM + i * 32 + (lane & 31);
                if (n0 == 0 && wn == 0 && lane < 32 && mr < p.M) p.yscale[mr / hw] = ys;
#pragma unroll
                for (int t = 0; t < 4; ++t) rowsplit[i][t] = __shfl(ys, (lane >> 3) + 8 * t);
            }
        }
    }
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
    if constexpr (KP) {
        // the two groups' sums meet: every wave hands the 32-column half it does NOT finish to its partner (wave ^ 4: same rows and
        // columns, the other k-steps) through the LDS -- [wave][register][lane] floats behind the eight transposition patches -- and
        // finishes the other half: group 0's sum + group 1's, in that order on both sides (the addition commutes)
        static_assert(!KP || TN == 2, "one column half per group");
        constexpr int XCH_OFF = 40960, XCH_WAVE = TM * 16 * 64;          // (8 patches of 32 x LDS_LD floats = 36864 bytes)
        float *xch = reinterpret_cast<float *>(smem + XCH_OFF);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) xch[wave * XCH_WAVE + (i * 16 + e) * 64 + lane] = grp ? acc[i][0][e] : acc[i][1][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // (the producers have left: the barrier counts the eight consumers)
        floatx16 half[TM][1];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float theirs = xch[(wave ^ 4) * XCH_WAVE + (i * 16 + e) * 64 + lane];
                half[i][0][e] = grp ? theirs + acc[i][1][e] : acc[i][0][e] + theirs;
            }
        tile_epilogue<TM, 1, WM, 32, SPLIT, VEC>(p, half, reinterpret_cast<float *>(smem), m0, n0, wm, 2 * wn + grp, lane, wave, split,
                                                 VEC ? rowscale : nullptr, (VEC && !SPLIT) ? rowsplit : nullptr, split_out);
    } else {
        tile_epilogue<TM, TN, WM, WN, SPLIT, VEC>(p, acc, reinterpret_cast<float *>(smem), m0, n0, wm, wn, lane, wave, split,
                                                  VEC ? rowscale : nullptr, (VEC && !SPLIT) ? rowsplit : nullptr, split_out);
    }
#endif
}

template <int BM, int BN, int NS, bool PRE, bool SPLIT, bool VEC, bool BNS = false, bool GP = false, bool KP = false, int CPS = 1>
int launch_ws_one(const ConvArgs &p, int splits, size_t lds, int tiles, hipStream_t stream) {
    if constexpr (!SPLIT && !BNS && !GP && !KP) {
        if (p.bn_part) return launch_ws_one<BM, BN, NS, PRE, SPLIT, VEC, true>(p, splits, lds, tiles, stream);
    }
    if constexpr (!SPLIT && VEC && !BNS && !PRE && !GP) {
        if (p.xscale) return launch_ws_one<BM, BN, NS, PRE, SPLIT, VEC, false, true, KP, CPS>(p, splits, lds, tiles, stream);
    }
    auto k = conv_igemm_ws_kernel<BM, BN, NS, SPLIT, VEC, BNS, PRE, GP, KP, CPS>;
    static PpyLdsAttr attr;
    if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), 160 * 1024) != PPY_OK) return PPY_ERR_LAUNCH;
    hipLaunchKernelGGL(k, dim3(tiles, splits), dim3((BM == 256 || KP) ? 768 : 512), lds, stream, p);
    return PPY_OK;
}

template <int BM, int BN, int NS, bool PRE = false, bool KP = false, int CPS = 1>
int launch_ws(ConvArgs p, int splits, hipStream_t stream) {
    const long long xbytes = (long long)p.N * p.H * p.W * p.x_ld * 4 + (long long)(p.pad * p.W + p.pad) * p.x_ld * 4;
    const long long wbytes = (long long)p.K * p.Kred * 2 * 2;
    if (xbytes >= 0xFFFFF000LL || wbytes >= 0xFFFFF000LL || p.R * p.S > 32) return PPY_ERR_UNSUPPORTED;
    constexpr int STAGE_BYTES = CPS * (BM * 128 * (PRE ? 2 : 1) + 2 * BN * 64);
    static_assert(NS * STAGE_BYTES <= 160 * 1024, "LDS");
    size_t lds = (size_t)NS * STAGE_BYTES;
    const size_t epi = KP ? (size_t)40960 + 8 * (BM / 64) * 16 * 64 * sizeof(float)        // eight patches + the groups' exchange area
                          : (size_t)(BM == 256 ? 8 : 4) * 32 * LDS_LD * sizeof(float);
    if (lds < epi) lds = epi;
    if (KP && p.bn_part) return PPY_ERR_UNSUPPORTED;
    p.nstages = NS;
    p.chunks_total = p.R * p.S * (p.C / 32);
    p.chunks_per_split = ceil_div(p.chunks_total, splits);
    splits = ceil_div(p.chunks_total, p.chunks_per_split);
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.K, BN);
    p.panel_n = ppy_panel_n(p, BM, BN, splits);
    const bool vec = vec_epilogue_ok(p);
    if (p.bn_part) {         // BatchNorm statistics from the epilogue: one split, plain conv + bias
        if (splits > 1 || p.res || p.posb || p.ups || p.act != PPY_ACT_NONE) return PPY_ERR_UNSUPPORTED;
        if (ceil_div(p.M, BM) * (BM == 256 ? 4 : 2) > p.bn_capacity) return PPY_ERR_WORKSPACE;
        if (p.bn_slices_host) *p.bn_slices_host = ceil_div(p.M, BM) * (BM == 256 ? 4 : 2);
    }
    if ((p.xscale || p.yscale) && (splits > 1 || !vec || p.bn_part)) return PPY_ERR_BAD_ARG;      // (conv_x3.hip's rule)
    if (p.yscale && p.ups) return PPY_ERR_BAD_ARG;
    if (p.xscale && PRE) return PPY_ERR_BAD_ARG;
    if (p.yscale && (p.K % 32 != 0 || p.y_ld % 32 != 0)) return PPY_ERR_BAD_ARG;
    int rc;
    if (splits > 1) {
        rc = vec ? launch_ws_one<BM, BN, NS, PRE, true, true, false, false, KP, CPS>(p, splits, lds, tiles, stream)
                 : launch_ws_one<BM, BN, NS, PRE, true, false, false, false, KP, CPS>(p, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
        launch_splitk_reduce(p, splits, vec, stream);
    } else {
        rc = vec ? launch_ws_one<BM, BN, NS, PRE, false, true, false, false, KP, CPS>(p, splits, lds, tiles, stream)
                 : launch_ws_one<BM, BN, NS, PRE, false, false, false, false, KP, CPS>(p, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
    }
    return ppy_launch_status();
}

}  // namespace

// local ids: 0 = 128x128 tile with 3 stages, 1 = the same with 4, 2 = 64x128 with 4, 3 = 64x128 with 6; with the activations
// split by the producer waves (PRE): 4 = 128x128 with 3 stages, 5 = 64x128 with 4, 6 = 128x64 with 4; eight consumer waves (4 x 2)
// + four producers on a 256x128 tile: 7 = two stages, 8 = three; round 6, eight consumer waves as two k-parity groups (KP) on a
// 128x128 tile: 9 = three stages, 10 = four; on a 64x128 tile (32x64 wave tiles): 11 = four stages, 12 = six; with TWO chunks per stage
// (one barrier per 64-deep step): 13 = 128x128 with two stages, 14 / 15 = 64x128 with two / three
// (256x128 / 128x256 with 2 x 2 consumer waves: 128 accumulator + 128 shortcut-prefetch registers spill)
int ppy_ws_num_configs() { return 16; }

int ppy_ws_dispatch(const ConvArgs &p, int c, int s, hipStream_t st) {
    if (!p.wf16 || ((uintptr_t)p.wf16 & 15) != 0 || !p.scale_f16 || !p.amax_in || (p.posb && !p.posb_f16)) return PPY_ERR_BAD_ARG;
    ConvArgs q = p;
    q.scale = p.scale_f16;
    q.posb = p.posb ? p.posb_f16 : nullptr;
    switch (c) {
        case 0: return launch_ws<128, 128, 3>(q, s, st);
        case 1: return launch_ws<128, 128, 4>(q, s, st);
        case 2: return launch_ws<64, 128, 4>(q, s, st);
        case 3: return launch_ws<64, 128, 6>(q, s, st);
        case 4: return launch_ws<128, 128, 3, true>(q, s, st);
        case 5: return launch_ws<64, 128, 4, true>(q, s, st);
        case 6: return launch_ws<128, 64, 4, true>(q, s, st);
        case 7: return launch_ws<256, 128, 2>(q, s, st);
        case 8: return launch_ws<256, 128, 3>(q, s, st);
        case 9: return launch_ws<128, 128, 3, false, true>(q, s, st);
        case 10: return launch_ws<128, 128, 4, false, true>(q, s, st);
        case 11: return launch_ws<64, 128, 4, false, true>(q, s, st);
        case 12: return launch_ws<64, 128, 6, false, true>(q, s, st);
        case 13: return launch_ws<128, 128, 2, false, true, 2>(q, s, st);
        case 14: return launch_ws<64, 128, 2, false, true, 2>(q, s, st);
        case 15: return launch_ws<64, 128, 3, false, true, 2>(q, s, st);
    }
    return PPY_ERR_BAD_ARG;
}
