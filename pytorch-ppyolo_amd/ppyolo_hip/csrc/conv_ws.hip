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
template <int BM, int BN, int NS, bool SPLIT, bool VEC, bool BNS = false, bool PRE = false, bool GP = false>      // (BNS: conv_x3.hip)
__global__ void __launch_bounds__(BM == 256 ? 768 : 512) conv_igemm_ws_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CM = BM == 256 ? 4 : 2, NC = CM * 2;              // consumer waves: CM x 2 over the tile (256-row tiles: eight)
    constexpr int WM = BM / CM, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int NP = 2, B_ROWS = NP * BN, NWP = 4;                 // weight planes, producer waves
    static_assert(BM % (8 * NWP) == 0 && B_ROWS % (16 * NWP) == 0, "whole DMA instructions per producer wave");
    constexpr int A_PASS = BM / (8 * NWP), B_PASS = B_ROWS / (16 * NWP), G = A_PASS + B_PASS;
    constexpr int A_BYTES = BM * 128, AP_BYTES = PRE ? 2 * BM * 64 : 0, B_BYTES = B_ROWS * 64;      // fp32 landing area, A planes, B planes
    constexpr int STAGE = A_BYTES + AP_BYTES + B_BYTES;
    static_assert((NS - 1) * G <= 63, "6-bit vmcnt");
    static_assert(!GP || (!PRE && !SPLIT && VEC && !BNS), "pre-split input: plain consumers, one split, vector epilogue");
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem_ws[];
    char *smem = smem_ws;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.K + BN - 1) / BN;
    int tile_id;
    {   // XCD-contiguous tile order (conv_x3.hip)
        const int nb = (int)gridDim.x, q = nb >> 3, r = nb & 7;
        const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
        tile_id = xcd * q + min(xcd, r) + idx;
    }
    const int tile_m = tile_id / tiles_n, tile_n = tile_id - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int kc_begin = split * p.chunks_per_split;
    const int kc_end = min(kc_begin + p.chunks_per_split, p.chunks_total);
    const int nchunks = kc_end - kc_begin;
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
        auto issue = [&](int stage, bool have) {
            const long long a_uni = ((long long)(l_r * p.W + l_s) * p.x_ld + l_cc * 32) * 4;
            const long long b_uni = ((long long)l_tap * (p.C / 32) + l_cc) * p.K * 64;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(xb + (have ? a_uni : 0)), 0, 0xFFFFFF00u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(wb + (have ? b_uni : 0)), 0, 0xFFFFFF00u, 0x00020000);
            const unsigned tapbit = have ? (1u << l_tap) : 0u;
            const unsigned lds = (unsigned)(stage * STAGE + pw * 1024);
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
        // PRE: the scale of the image of each of this lane's piece rows, and the split of this wave's pieces of a stage
        float sa_p[A_PASS];
        if constexpr (PRE) {
#pragma unroll
            for (int j = 0; j < A_PASS; ++j) {
                const int mrow = min(m0 + (j * NWP + pw) * 8 + (lane >> 3), p.M - 1);
                const float mx = amax_read(p.amax_in, mrow / hw);
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
            for (int sidx = 0; sidx < NS; ++sidx) issue(sidx, sidx < nchunks);
            wait_vmcnt<(NS - 1) * G>();                  // chunk 0 has landed
            split_stage(0);
            __builtin_amdgcn_s_barrier();
            int st = 0;
            for (int k = 0; k < nchunks; ++k) {
                wait_vmcnt<(NS - 2) * G>();              // chunk k+1 has landed (this wave's pieces; the barrier makes it all of them)
                split_stage(st + 1 == NS ? 0 : st + 1);
                __builtin_amdgcn_s_barrier();            // ... and the consumers have read all of chunk k
                issue(st, k + NS < nchunks);
                st = st + 1 == NS ? 0 : st + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ================= consumers: 2 x 2 waves over the tile; LDS reads, operand split, MFMAs =================
    const int wm = wave >> 1, wn = wave & 1;
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
            if (p.yscale) xmax_up[i] = pow2_above(amax_read(p.amax_in, mrow / hw));      // (conv_x3.hip: chains bound from the tracked maximum)
            continue;
        }
        const float mx = amax_read(p.amax_in, mrow / hw);
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
    auto step = [&](const Frag &cur, Frag &nxt, int stage, auto s_tag) {
        constexpr int s = decltype(s_tag)::value;      // k-step (0/1) of the chunk the NEXT operands come from
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
                        nxt.a[r >> 1][r & 1] = *reinterpret_cast<const uintx4 *>(a_ptr + (r & 1) * BM * 64 + (r >> 1) * 32 * 64 + ap_foff[s]);
                    else if constexpr (GP)  // (tile r>>1, term r&1) of the pre-split rows
                        nxt.a[r >> 1][r & 1] = *reinterpret_cast<const uintx4 *>(a_ptr + (r >> 1) * 32 * 128 + a_foff[s][r & 1]);
                    else
                        raw[r >> 1][r & 1] = *reinterpret_cast<const floatx4 *>(a_ptr + (r >> 1) * 32 * 128 + a_foff[s][r & 1]);
                } else {
                    const int pl = (r - NRA) / TN, j = (r - NRA) % TN;
                    nxt.b[pl][j] = *reinterpret_cast<const uintx4 *>(b_ptr + (pl * BN + j * 32) * 64 + b_foff[s]);
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
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;

    if (nchunks > 0) {
        Frag f0, f1;
        __builtin_amdgcn_s_barrier();            // chunk 0 is in the LDS
        {   // operands of (chunk 0, k-step 0): not overlapped with anything
            const char *a_ptr = smem + wm * WM * 128;
            const char *b_ptr = smem + A_BYTES + AP_BYTES + wn * WN * 64;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (PRE) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        f0.a[i][pl] = *reinterpret_cast<const uintx4 *>(smem + A_BYTES + pl * BM * 64 + (wm * WM + i * 32) * 64 + ap_foff[0]);
                    continue;
                }
                if constexpr (GP) {
                    f0.a[i][0] = *reinterpret_cast<const uintx4 *>(a_ptr + i * 32 * 128 + a_foff[0][0]);
                    f0.a[i][1] = *reinterpret_cast<const uintx4 *>(a_ptr + i * 32 * 128 + a_foff[0][1]);
                    continue;
                }
                const floatx4 lo = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * 128 + a_foff[0][0]);
                const floatx4 hi = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * 128 + a_foff[0][1]);
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
                    f0.b[pl][j] = *reinterpret_cast<const uintx4 *>(b_ptr + (pl * BN + j * 32) * 64 + b_foff[0]);
        }
        int st = 0;
        for (int k = 0; k < nchunks; ++k) {
            step(f0, f1, st, S1());                              // k-step 0 of chunk k  ||  fetch + split k-step 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of chunk k has returned
            __builtin_amdgcn_s_barrier();
            st = st + 1 == NS ? 0 : st + 1;
            step(f1, f0, st, S0());                              // k-step 1 of chunk k  ||  fetch + split k-step 0 of chunk k+1
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
    tile_epilogue<TM, TN, WM, WN, SPLIT, VEC>(p, acc, reinterpret_cast<float *>(smem), m0, n0, wm, wn, lane, wave, split,
                                              VEC ? rowscale : nullptr, (VEC && !SPLIT) ? rowsplit : nullptr, split_out);
#endif
}

template <int BM, int BN, int NS, bool PRE, bool SPLIT, bool VEC, bool BNS = false, bool GP = false>
int launch_ws_one(const ConvArgs &p, int splits, size_t lds, int tiles, hipStream_t stream) {
    if constexpr (!SPLIT && !BNS && !GP) {
        if (p.bn_part) return launch_ws_one<BM, BN, NS, PRE, SPLIT, VEC, true>(p, splits, lds, tiles, stream);
    }
    if constexpr (!SPLIT && VEC && !BNS && !PRE && !GP) {
        if (p.xscale) return launch_ws_one<BM, BN, NS, PRE, SPLIT, VEC, false, true>(p, splits, lds, tiles, stream);
    }
    auto k = conv_igemm_ws_kernel<BM, BN, NS, SPLIT, VEC, BNS, PRE, GP>;
    static PpyLdsAttr attr;
    if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), 160 * 1024) != PPY_OK) return PPY_ERR_LAUNCH;
    hipLaunchKernelGGL(k, dim3(tiles, splits), dim3(BM == 256 ? 768 : 512), lds, stream, p);
    return PPY_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: PING-PONG tiles for the 1x1 "expand" layers (reference model/resnet_vd.py:55-91, conv3 + shortcut + ReLU of stages 3-5).
// On every other tile of this library a launch of such a layer is two chip-wide phases -- ~17 us in which all workgroups multiply
// (a reduction of only 4-16 chunks: HBM idles) and ~17 us in which they all read the shortcut and store (5.6 TB/s: the matrix
// pipe idles); four tile shapes end at the same 40.6 us (profiles/r04_expand_stagger.txt).  Here ONE persistent workgroup per
// CU keeps both phases in flight: twelve waves = four producers (all LDS-DMA pieces, as above) + TWO consumer groups of 2 x 2
// waves that take the workgroup's 128 x 128 tiles in turn -- while group A multiplies tile i, group B sends tile i-1 through its
// epilogue (own LDS patches, so the stages keep streaming), then they swap.  The producers run through the chunks of
// consecutive tiles without a gap, so a tile's first chunks land during the previous tile's last MFMAs.  Synchronisation is the
// workgroup barrier of the kernel above, one per chunk: s_barrier counts arrivals, so the group that is busy with an epilogue
// simply arrives NC times per period, between the four 32 x 32 pieces of its sub-tile.  Every wave of a workgroup executes
// 2 + tiles x NC barriers.  Same operand layouts, same products in the same order, same epilogue arithmetic as the f16x2 tiles:
// bit-identical results.
template <bool GP>
__global__ void __launch_bounds__(768) conv1x1_pp_kernel(const ConvArgs p, const int tiles_n, const int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, BN = 128, NS = 3, NP = 2, NWP = 4, TM = 2, TN = 2, WM = 64, WN = 64;
    constexpr int A_PASS = BM / (8 * NWP), B_ROWS = NP * BN, B_PASS = B_ROWS / (16 * NWP), G = A_PASS + B_PASS;
    constexpr int A_BYTES = BM * 128, B_BYTES = B_ROWS * 64, STAGE = A_BYTES + B_BYTES;
    static_assert((NS - 1) * G <= 63, "6-bit vmcnt");
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem_ws[];
    char *smem = smem_ws;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NC = p.chunks_total;                                       // 32-channel chunks of the reduction (R = S = 1)
    const int nt = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // this workgroup's tiles: b, b + grid, ...
    const int hw = p.Ho * p.Wo;

    if (wave >= 8) {
        // ================= producers: the chunks of all tiles, back to back through the NS stages =================
        const int pw = wave - 8;
        const unsigned OOB = 0xFFFFFFF0u;
        unsigned a_off[A_PASS], b_off[B_PASS];
        const long long plane_bytes = (long long)p.K * p.Kred * 2;
        auto setup_tile = [&](int q) {
            const int id = (int)blockIdx.x + q * (int)gridDim.x;
            const int tm = id / tiles_n, tn = id - tm * tiles_n;
#pragma unroll
            for (int d = 0; d < A_PASS; ++d) {
                const int row = (d * NWP + pw) * 8 + (lane >> 3);
                const int scol = (lane & 7) ^ ((row >> 1) & 7);
                const int mr = tm * BM + row;
                a_off[d] = mr < p.M ? (unsigned)((long long)mr * p.x_ld * 4 + scol * 16) : OOB;
            }
#pragma unroll
            for (int j = 0; j < B_PASS; ++j) {
                const int rb = (j * NWP + pw) * 16 + (lane >> 2);
                const int plane = rb / BN, nrow = rb - plane * BN;
                const int scol = (lane & 3) ^ ((rb >> 2) & 3);
                const int k = min(tn * BN + nrow, p.K - 1);              // rows >= K are masked at store
                b_off[j] = (unsigned)(plane * plane_bytes + (long long)k * 64 + scol * 16);
            }
        };
        const char *xb = reinterpret_cast<const char *>(p.x);
        const char *wb = reinterpret_cast<const char *>(p.wf16);
        int qi = 0, ki = 0;                                              // the next chunk to request: tile qi, chunk ki
        auto issue_next = [&](int stage) {
            const bool have = qi < nt;
            if (have && ki == 0) setup_tile(qi);
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(xb + (have ? (long long)ki * 128 : 0)), 0, 0xFFFFFF00u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(wb + (have ? (long long)ki * p.K * 64 : 0)), 0, 0xFFFFFF00u, 0x00020000);
            const unsigned lds = (unsigned)(stage * STAGE + pw * 1024);
#pragma unroll
            for (int d = 0; d < A_PASS; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(smem + lds + d * NWP * 1024), 16, have ? a_off[d] : OOB, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < B_PASS; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(smem + lds + A_BYTES + j * NWP * 1024), 16, have ? b_off[j] : OOB, 0, 0, 0);
            if (have && ++ki == NC) {
                ki = 0;
                ++qi;
            }
        };
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) issue_next(sidx);
        wait_vmcnt<(NS - 1) * G>();                      // chunk 0 has landed
        __builtin_amdgcn_s_barrier();
        int st = 0;
        const int J = nt * NC;
        for (int j = 0; j < J; ++j) {
            wait_vmcnt<(NS - 2) * G>();                  // chunk j+1 has landed (this wave's pieces; the barrier makes it all of them)
            __builtin_amdgcn_s_barrier();                // ... and its consumers have read all of chunk j
            issue_next(st);
            st = st + 1 == NS ? 0 : st + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        return;
    }

    // ================= consumers: group (wave >> 2) takes the tiles q with q & 1 == group =================
    const int grp = wave >> 2, cw = wave & 3, wm = cw >> 1, wn = cw & 1;
    float *sE = reinterpret_cast<float *>(smem + NS * STAGE) + wave * (32 * LDS_LD);
    const int frow = lane & 31, fkh = lane >> 5;
    const int a_sw = (frow >> 1) & 7, b_sw = (frow >> 2) & 3;
    int a_foff[2][2], b_foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if constexpr (GP) {
            a_foff[s][0] = frow * 128 + (((2 * s + fkh) ^ a_sw) << 4);
            a_foff[s][1] = frow * 128 + (((4 + 2 * s + fkh) ^ a_sw) << 4);
        } else {
            a_foff[s][0] = frow * 128 + (((4 * s + 2 * fkh) ^ a_sw) << 4);
            a_foff[s][1] = frow * 128 + (((4 * s + 2 * fkh + 1) ^ a_sw) << 4);
        }
        b_foff[s] = frow * 64 + (((2 * s + fkh) ^ b_sw) << 4);
    }
    struct Frag {
        uintx4 a[TM][NP];
        uintx4 b[NP][TN];
    };
    constexpr int NM = 3 * TM * TN, NRA = 2 * TM, NRB = NP * TN, NR = NRA + NRB;
    constexpr int NSL = GP ? 0 : 3 * 4 * TM;
    constexpr int RPS = (NR + NM - 1) / NM;
    constexpr int LEAD0 = (NRA + RPS - 1) / RPS + 1;
    constexpr int LEAD = LEAD0 < NM ? LEAD0 : NM - 1;
    constexpr int PER = (NSL + (NM - LEAD) - 1) / (NM - LEAD);
    floatx16 acc[TM][TN];
    float sa[TM], inv_sa[TM];
    // one k-step (conv_igemm_ws_kernel's step()): NM slots { one MFMA of step g ; LDS reads for step g+1 ; pieces of its split }
    auto step = [&](const Frag &cur, Frag &nxt, int stage, auto s_tag, const bool fetch) {
        constexpr int s = decltype(s_tag)::value;
        constexpr int ta[3] = {1, 0, 0}, tb[3] = {0, 1, 0};
        const char *a_ptr = smem + stage * STAGE + wm * WM * 128;
        const char *b_ptr = smem + stage * STAGE + A_BYTES + wn * WN * 64;
        floatx4 raw[TM][2];
        float ra[TM][4], rb[TM][4];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            {
                const int t = m / (TM * TN), i = (m / TN) % TM, j = m % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur.a[i][ta[t]]),
                                                                  __builtin_bit_cast(f16x8, cur.b[tb[t]][j]), acc[i][j], 0, 0, 0);
            }
            if (fetch) {
#pragma unroll
                for (int u = 0; u < RPS; ++u) {
                    const int r = m * RPS + u;
                    if (r >= NR) {
                    } else if (r < NRA) {
                        if constexpr (GP)
                            nxt.a[r >> 1][r & 1] = *reinterpret_cast<const uintx4 *>(a_ptr + (r >> 1) * 32 * 128 + a_foff[s][r & 1]);
                        else
                            raw[r >> 1][r & 1] = *reinterpret_cast<const floatx4 *>(a_ptr + (r >> 1) * 32 * 128 + a_foff[s][r & 1]);
                    } else {
                        const int pl = (r - NRA) / TN, j = (r - NRA) % TN;
                        nxt.b[pl][j] = *reinterpret_cast<const uintx4 *>(b_ptr + (pl * BN + j * 32) * 64 + b_foff[s]);
                    }
                }
                if (m >= LEAD) {
#pragma unroll
                    for (int u = 0; u < PER; ++u) {
                        const int sl = (m - LEAD) * PER + u;
                        if (sl < NSL) {
                            const int stg = sl / (4 * TM), pr = sl % (4 * TM), i = pr / 4, q4 = pr % 4;
                            const float xa = raw[i][q4 >> 1][(q4 & 1) * 2], xb = raw[i][q4 >> 1][(q4 & 1) * 2 + 1];
                            if (stg == 0) {
                                nxt.a[i][0][q4] = cvt_pk_f16(xa * sa[i], xb * sa[i]);
                            } else if (stg == 1) {
                                const unsigned P = nxt.a[i][0][q4];
                                ra[i][q4] = fmaf(xa, sa[i], -f16_lo(P));
                                rb[i][q4] = fmaf(xb, sa[i], -f16_hi(P));
                            } else {
                                nxt.a[i][1][q4] = cvt_pk_f16(ra[i][q4], rb[i][q4]);
                            }
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
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;

    // the epilogue of a finished tile (tile_epilogue's vector path: scale, shift, shortcut, activation, tracked maxima), in four
    // 32 x 32 pieces; `sync`: this group is the passive one of a period and arrives at the period's NC barriers on the way
    auto epilogue = [&](const int m0, const int n0, const float (&rowscale)[TM][4], const bool sync) {
        const int erow = lane >> 3, ec4 = (lane & 7) * 4;
        const int mw0 = min(m0 + wm * WM, p.M - 1), mw1 = min(m0 + wm * WM + WM - 1, p.M - 1);
        const int n_lo = mw0 / hw, n_hi = mw1 / hw;
        const int bnd = (n_lo + 1) * hw;
        float amx = 0.0f, amx_hi = 0.0f;
        floatx4 rv[TM][TN][4];
        if (p.res) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int col = n0 + wn * WN + j * 32 + ec4;
                        const int m = m0 + wm * WM + i * 32 + erow + 8 * t;
                        rv[i][j][t] = floatx4{0.f, 0.f, 0.f, 0.f};
                        if (col < p.K && m < p.M) rv[i][j][t] = *reinterpret_cast<const floatx4 *>(p.res + (long long)m * p.res_ld + col);
                    }
        }
        int piece = 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + ec4;
            const bool colok = col < p.K;
            floatx4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (colok) {
                sc = *reinterpret_cast<const floatx4 *>(p.scale + col);
                sh = *reinterpret_cast<const floatx4 *>(p.shift + col);
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
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] *= rowscale[i][t];
                    if (colok && m < p.M) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            float o = fmaf(v[u], sc[u], sh[u]);
                            if (p.res) o += rv[i][j][t][u];
                            v[u] = ppy_apply_act(o, p.act);
                        }
                        const float rmx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                        amx = fmaxf(amx, m < bnd ? rmx : 0.0f);
                        amx_hi = fmaxf(amx_hi, m < bnd ? 0.0f : rmx);
                        *reinterpret_cast<floatx4 *>(p.y + (long long)m * p.y_ld + col) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (sync) {          // this piece's share of the period's NC arrivals
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const int nb = NC * (piece + 1) / 4 - NC * piece / 4;
                    for (int b = 0; b < nb; ++b) __builtin_amdgcn_s_barrier();
                }
                ++piece;
            }
        }
        if (p.amax_out) amax_track2(amx, amx_hi, n_lo, n_hi, p.amax_out, blockIdx.x * 8 + wave);
    };

    bool pend = false;
    int pend_m0 = 0, pend_n0 = 0;
    float pend_scale[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) pend_scale[i][t] = 0.f;
    __builtin_amdgcn_s_barrier();            // chunk 0 is in the LDS
    int st = 0;
    for (int q = 0; q < nt; ++q) {
        if ((q & 1) != grp) {
            // ---- passive period: the tile finished one period ago goes out; NC arrivals either way ----
            if (pend) {
                epilogue(pend_m0, pend_n0, pend_scale, true);
                pend = false;
            } else {
                for (int b = 0; b < NC; ++b) __builtin_amdgcn_s_barrier();
            }
            st = (st + NC) % NS;
            continue;
        }
        const int id = (int)blockIdx.x + q * (int)gridDim.x;
        const int tm_ = id / tiles_n, tn_ = id - tm_ * tiles_n;
        const int m0 = tm_ * BM, n0 = tn_ * BN;
#pragma unroll
        for (int i = 0; i < TM; ++i) {        // per-image activation scale (conv_x3.hip)
            const int mrow = min(m0 + wm * WM + i * 32 + (lane & 31), p.M - 1);
            if constexpr (GP) {
                sa[i] = p.xscale[mrow / hw];
                inv_sa[i] = pow2_inverse(sa[i]);
            } else {
                const float mx = amax_read(p.amax_in, mrow / hw);
                const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
                int f = 267 - e;
                f = f < 103 ? 103 : (f > 167 ? 167 : f);
                sa[i] = __uint_as_float((unsigned)f << 23);
                inv_sa[i] = __uint_as_float((unsigned)(254 - f) << 23);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        Frag f0, f1;
        {   // operands of (chunk 0, k-step 0) of this tile: landed since the last barrier
            const char *a_ptr = smem + st * STAGE + wm * WM * 128;
            const char *b_ptr = smem + st * STAGE + A_BYTES + wn * WN * 64;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (GP) {
                    f0.a[i][0] = *reinterpret_cast<const uintx4 *>(a_ptr + i * 32 * 128 + a_foff[0][0]);
                    f0.a[i][1] = *reinterpret_cast<const uintx4 *>(a_ptr + i * 32 * 128 + a_foff[0][1]);
                } else {
                    const floatx4 lo = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * 128 + a_foff[0][0]);
                    const floatx4 hi = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * 128 + a_foff[0][1]);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const float xa = q4 < 2 ? lo[2 * q4] : hi[2 * q4 - 4], xb = q4 < 2 ? lo[2 * q4 + 1] : hi[2 * q4 - 3];
                        const unsigned P0 = cvt_pk_f16(xa * sa[i], xb * sa[i]);
                        f0.a[i][0][q4] = P0;
                        f0.a[i][1][q4] = cvt_pk_f16(fmaf(xa, sa[i], -f16_lo(P0)), fmaf(xb, sa[i], -f16_hi(P0)));
                    }
                }
            }
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int j = 0; j < TN; ++j) f0.b[pl][j] = *reinterpret_cast<const uintx4 *>(b_ptr + (pl * BN + j * 32) * 64 + b_foff[0]);
        }
        for (int k = 0; k < NC; ++k) {
            step(f0, f1, st, S1(), true);                        // k-step 0 of chunk k  ||  fetch + split k-step 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of chunk k has returned
            __builtin_amdgcn_s_barrier();
            st = st + 1 == NS ? 0 : st + 1;
            step(f1, f0, st, S0(), k + 1 < NC);                  // k-step 1 of chunk k  ||  k-step 0 of chunk k+1 (not across a tile: the next tile is the other group's)
        }
        pend = true;
        pend_m0 = m0;
        pend_n0 = n0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) pend_scale[i][t] = __shfl(inv_sa[i], (lane >> 3) + 8 * t);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (pend) epilogue(pend_m0, pend_n0, pend_scale, false);
#endif
}

int launch_pp(ConvArgs p, int splits, hipStream_t stream) {
    if (p.R != 1 || p.S != 1 || p.stride != 1 || p.pad != 0 || p.ups || p.posb || p.yscale || p.bn_part || splits > 1) return PPY_ERR_BAD_ARG;
    if (!vec_epilogue_ok(p) || ((uintptr_t)p.x & 15) != 0 || p.x_ld % 4 != 0) return PPY_ERR_BAD_ARG;
    const long long xbytes = (long long)p.M * p.x_ld * 4, wbytes = (long long)p.K * p.Kred * 2 * 2;
    if (xbytes >= 0xFFFFF000LL || wbytes >= 0xFFFFF000LL) return PPY_ERR_UNSUPPORTED;
    p.nstages = 3;
    p.chunks_total = p.C / 32;
    p.chunks_per_split = p.chunks_total;
    const int tiles_n = ceil_div(p.K, 128), ntiles = ceil_div(p.M, 128) * tiles_n;
    int n_cu = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n_cu = prop.multiProcessorCount;
    }
    const int grid = ntiles < n_cu ? ntiles : n_cu;              // one persistent workgroup (twelve waves, 132 KB of LDS) per CU
    const size_t lds = (size_t)3 * (128 * 128 + 256 * 64) + (size_t)8 * 32 * LDS_LD * sizeof(float);
    static PpyLdsAttr attr_gp, attr_plain;
    if (p.xscale) {
        if (ppy_lds_attr(attr_gp, reinterpret_cast<const void *>(conv1x1_pp_kernel<true>), (int)lds) != PPY_OK) return PPY_ERR_LAUNCH;
        hipLaunchKernelGGL(conv1x1_pp_kernel<true>, dim3(grid), dim3(768), lds, stream, p, tiles_n, ntiles);
    } else {
        if (ppy_lds_attr(attr_plain, reinterpret_cast<const void *>(conv1x1_pp_kernel<false>), (int)lds) != PPY_OK) return PPY_ERR_LAUNCH;
        hipLaunchKernelGGL(conv1x1_pp_kernel<false>, dim3(grid), dim3(768), lds, stream, p, tiles_n, ntiles);
    }
    return ppy_launch_status();
}

template <int BM, int BN, int NS, bool PRE = false>
int launch_ws(ConvArgs p, int splits, hipStream_t stream) {
    const long long xbytes = (long long)p.N * p.H * p.W * p.x_ld * 4 + (long long)(p.pad * p.W + p.pad) * p.x_ld * 4;
    const long long wbytes = (long long)p.K * p.Kred * 2 * 2;
    if (xbytes >= 0xFFFFF000LL || wbytes >= 0xFFFFF000LL || p.R * p.S > 32) return PPY_ERR_UNSUPPORTED;
    constexpr int STAGE_BYTES = BM * 128 * (PRE ? 2 : 1) + 2 * BN * 64;
    static_assert(NS * STAGE_BYTES <= 160 * 1024, "LDS");
    size_t lds = (size_t)NS * STAGE_BYTES;
    const size_t epi = (size_t)(BM == 256 ? 8 : 4) * 32 * LDS_LD * sizeof(float);
    if (lds < epi) lds = epi;
    p.nstages = NS;
    p.chunks_total = p.R * p.S * (p.C / 32);
    p.chunks_per_split = ceil_div(p.chunks_total, splits);
    splits = ceil_div(p.chunks_total, p.chunks_per_split);
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.K, BN);
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
        rc = vec ? launch_ws_one<BM, BN, NS, PRE, true, true>(p, splits, lds, tiles, stream)
                 : launch_ws_one<BM, BN, NS, PRE, true, false>(p, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
        launch_splitk_reduce(p, splits, vec, stream);
    } else {
        rc = vec ? launch_ws_one<BM, BN, NS, PRE, false, true>(p, splits, lds, tiles, stream)
                 : launch_ws_one<BM, BN, NS, PRE, false, false>(p, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
    }
    return ppy_launch_status();
}

}  // namespace

// local ids: 0 = 128x128 tile with 3 stages, 1 = the same with 4, 2 = 64x128 with 4, 3 = 64x128 with 6; with the activations
// split by the producer waves (PRE): 4 = 128x128 with 3 stages, 5 = 64x128 with 4, 6 = 128x64 with 4; eight consumer waves (4 x 2)
// + four producers on a 256x128 tile: 7 = two stages, 8 = three
// (256x128 / 128x256 with 2 x 2 consumer waves: 128 accumulator + 128 shortcut-prefetch registers spill)
int ppy_ws_num_configs() { return 10; }      // (9 = the ping-pong tiles for 1x1 layers, conv1x1_pp_kernel)

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
        case 9: return launch_pp(q, s, st);
    }
    return PPY_ERR_BAD_ARG;
}
