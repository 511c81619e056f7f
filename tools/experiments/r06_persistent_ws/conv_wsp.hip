// The specialised-wave f16x2 tiles of conv_ws.hip as a PERSISTENT workgroup (round 6): a workgroup walks over several output tiles and
// its four producer waves treat the chunks of all of them as ONE stream -- when the consumer waves reach a tile's epilogue the
// producers have already requested the first NS chunks of the next tile, and they land while the tile is transposed and stored.
//
// Why: the layers with short reductions (the HBM-bound 1x1 layers of stages 2-3: 4-8 chunks per tile) spend a workgroup's life in three
// phases that do not overlap -- the first operand round trip, a handful of chunks, the store phase -- and a CU holds one or two such
// workgroups; the tables pick small tiles there to get more workgroups per CU, at the price of re-reading the activations through the
// L2 once per column tile.  Same operand layouts in the LDS, same products in the same order, same shared epilogue as conv_ws.hip's
// tiles 0-3: results are bit-identical to them (and to conv_x3.hip's).  The transposition patches of the epilogue live BEHIND the stages
// here (the stages keep streaming during the epilogue).
#include "conv_shared.h"

namespace {

template <int BM, int BN, int NS, bool GP>
__global__ void __launch_bounds__(512) conv_wsp_kernel(const ConvArgs p, const int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NC = 4;                                            // consumer waves, 2 x 2 over the tile
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int NP = 2, B_ROWS = NP * BN, NWP = 4;                 // weight planes, producer waves
    static_assert(BM % (8 * NWP) == 0 && B_ROWS % (16 * NWP) == 0, "whole DMA instructions per producer wave");
    constexpr int A_PASS = BM / (8 * NWP), B_PASS = B_ROWS / (16 * NWP), G = A_PASS + B_PASS;
    constexpr int A_BYTES = BM * 128, B_BYTES = B_ROWS * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int E_OFF = NS * STAGE;                                // four transposition patches behind the stages
    static_assert((NS - 1) * G <= 63, "6-bit vmcnt");
    static_assert(E_OFF + NC * 32 * LDS_LD * 4 <= 160 * 1024, "LDS");
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem_wsp[];
    char *smem = smem_wsp;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.K + BN - 1) / BN;
    const int grid = (int)gridDim.x;                                 // (a multiple of 8, or == ntiles: the launcher)
    const int T = (ntiles - (int)blockIdx.x + grid - 1) / grid;      // tiles of this workgroup: blockIdx.x + j * grid
    const int nchunks = p.chunks_total;
    const int total = T * nchunks;
    const int hw = p.Ho * p.Wo;
    // tile j of this workgroup -> (tile_m, tile_n): conv_shared.h's XCD-contiguous order over ALL ntiles (vb & 7 = this workgroup's XCD)
    auto tile_of = [&](int j, int &tile_m, int &tile_n) {
        const int vb = (int)blockIdx.x + j * grid;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int xcd = vb & 7, idx = vb >> 3;
        const int tile_id = grid == ntiles || (grid & 7) == 0 ? xcd * q + min(xcd, r) + idx : vb;
        const int pn = p.panel_n;
        if (pn <= 0 || pn >= tiles_n) {
            tile_m = tile_id / tiles_n;
            tile_n = tile_id - tile_m * tiles_n;
            return;
        }
        const int tiles_m = ntiles / tiles_n, per_panel = tiles_m * pn;
        const int panel = tile_id / per_panel, within = tile_id - panel * per_panel;
        const int pw = min(pn, tiles_n - panel * pn);
        tile_m = within / pw;
        tile_n = panel * pn + (within - tile_m * pw);
    };

    if (wave >= NC) {
        // ================= producers: the chunks of all of this workgroup's tiles as one stream =================
        const int pw = wave - NC;
        const unsigned OOB = 0xFFFFFFF0u;
        const long long bias = (long long)(p.pad * p.W + p.pad) * p.x_ld * 4;   // keeps offsets >= 0
        unsigned a_off[A_PASS], a_ok[A_PASS], b_off[B_PASS];
        auto setup_tile = [&](int j) {
            int tile_m, tile_n;
            tile_of(j, tile_m, tile_n);
            const int m0 = tile_m * BM, n0 = tile_n * BN;
            {
                const int drow = lane >> 3, dslot = lane & 7;
                const int step_rows = 8 * NWP;
                const int step_ho = step_rows / p.Wo, step_wo = step_rows - step_ho * p.Wo;
                int m_first = min(m0 + pw * 8 + drow, p.M - 1);
                int n = m_first / hw, rem = m_first - n * hw;
                int ho = rem / p.Wo, wo = rem - ho * p.Wo;
#pragma unroll
                for (int d = 0; d < A_PASS; ++d) {
                    const int row = (d * NWP + pw) * 8 + drow;
                    const int scol = dslot ^ ((row >> 1) & 7);
                    const int mr = m0 + row;
                    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
                    a_off[d] = (unsigned)((((long long)n * p.H + hi0) * p.W + wi0) * p.x_ld * 4 + bias + scol * 16);
                    unsigned colmask = 0, okb = 0;
                    for (int s2 = 0; s2 < p.S; ++s2)
                        if ((unsigned)(wi0 + s2) < (unsigned)p.W) colmask |= 1u << s2;
                    for (int r = 0; r < p.R; ++r)
                        if ((unsigned)(hi0 + r) < (unsigned)p.H) okb |= colmask << (r * p.S);
                    a_ok[d] = mr < p.M ? okb : 0u;
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
                for (int d = 0; d < B_PASS; ++d) {
                    const int rb = (d * NWP + pw) * 16 + drow;          // row of the [2*BN] B tile
                    const int plane = rb / BN, nrow = rb - plane * BN;
                    const int scol = dslot ^ ((rb >> 2) & 3);
                    const int k = min(n0 + nrow, p.K - 1);              // rows >= K are masked at store
                    b_off[d] = (unsigned)(plane * plane_bytes + (long long)k * 64 + scol * 16);      // [chunk][K][32] planes
                }
            }
        };
        const int RS = p.R * p.S;
        int i_tile = 0, i_chunk = 0, l_cc = 0, l_tap = 0, l_r = 0, l_s = 0;      // request cursor (chunk = (cc, tap): cc outer, tap inner)
        const char *xb = reinterpret_cast<const char *>(p.x) - bias;
        const char *wb = reinterpret_cast<const char *>(p.wf16);
        setup_tile(0);
        auto issue = [&](int stage) {          // the next chunk of the stream into `stage` (past the last tile: out-of-range dummies)
            const bool have = i_tile < T;
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
            for (int d = 0; d < B_PASS; ++d) {
                const unsigned off = have ? b_off[d] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(smem + lds + A_BYTES + d * NWP * 1024), 16, off, 0, 0, 0);
            }
            if (!have) return;
            ++l_tap;
            ++l_s;
            if (l_s == p.S) { l_s = 0; ++l_r; }
            if (l_tap == RS) { l_tap = 0; l_r = 0; l_s = 0; ++l_cc; }
            if (++i_chunk == nchunks) {        // the next request opens the next tile
                i_chunk = 0;
                l_cc = l_tap = l_r = l_s = 0;
                if (++i_tile < T) setup_tile(i_tile);
            }
        };
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) issue(sidx);
        wait_vmcnt<(NS - 1) * G>();                  // the first chunk has landed
        __builtin_amdgcn_s_barrier();
        int st = 0;
        for (int k = 0; k < total; ++k) {
            wait_vmcnt<(NS - 2) * G>();              // chunk k + 1 has landed (this wave's pieces; the barrier makes it all of them)
            __builtin_amdgcn_s_barrier();            // ... and the consumers have read all of chunk k
            issue(st);
            st = st + 1 == NS ? 0 : st + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        return;
    }

    // ================= consumers: 2 x 2 waves over the tile; LDS reads, operand split, MFMAs; epilogue per tile =================
    const int wm = wave >> 1, wn = wave & 1;
    // fragment read offsets (bytes; conv_ws.hip).  MFMA k-step s (16 deep), lane-half h: k = 16s + 8h + [0,8)
    const int frow = lane & 31, fkh = lane >> 5;
    const int a_sw = (frow >> 1) & 7, b_sw = (frow >> 2) & 3;
    int a_foff[2][2], b_foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if constexpr (GP) {      // first terms in the 16-byte slots 0..3 of the row, second terms in 4..7
            a_foff[s][0] = frow * 128 + (((2 * s + fkh) ^ a_sw) << 4);
            a_foff[s][1] = frow * 128 + (((4 + 2 * s + fkh) ^ a_sw) << 4);
        } else {
            a_foff[s][0] = frow * 128 + (((4 * s + 2 * fkh) ^ a_sw) << 4);
            a_foff[s][1] = frow * 128 + (((4 * s + 2 * fkh + 1) ^ a_sw) << 4);
        }
        b_foff[s] = frow * 64 + (((2 * s + fkh) ^ b_sw) << 4);
    }
    struct Scales {
        float sa[TM], inv_sa[TM], xmax_up[TM];
        int m0, n0;
    };
    auto scales_of = [&](int j, Scales &o) {      // per-image activation scale of this lane's rows of tile j (conv_x3.hip)
        int tile_m, tile_n;
        tile_of(j, tile_m, tile_n);
        o.m0 = tile_m * BM;
        o.n0 = tile_n * BN;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            o.xmax_up[i] = 1.0f;
            const int mrow = min(o.m0 + wm * WM + i * 32 + (lane & 31), p.M - 1);
            if constexpr (GP) {
                o.sa[i] = p.xscale[mrow / hw];
                o.inv_sa[i] = pow2_inverse(o.sa[i]);
                if (p.yscale) o.xmax_up[i] = pow2_above(conv_amax_in(p, mrow / hw));
            } else {
                const float mx = conv_amax_in(p, mrow / hw);
                const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
                int f = 267 - e;
                f = f < 103 ? 103 : (f > 167 ? 167 : f);
                o.sa[i] = __uint_as_float((unsigned)f << 23);
                o.inv_sa[i] = __uint_as_float((unsigned)(254 - f) << 23);
                o.xmax_up[i] = 16384.0f * o.inv_sa[i];
            }
        }
    };

    struct Frag {        // operands of one k-step
        uintx4 a[TM][NP];
        uintx4 b[NP][TN];
    };
    floatx16 acc[TM][TN];
    constexpr int NM = 3 * TM * TN;            // MFMAs per k-step
    constexpr int NRA = 2 * TM, NRB = NP * TN, NR = NRA + NRB;
    constexpr int NSL = GP ? 0 : 3 * 4 * TM;   // split stages per k-step (3 dependent stages x 4 pairs x TM)
    constexpr int RPS = (NR + NM - 1) / NM;
    constexpr int LEAD0 = (NRA + RPS - 1) / RPS + 1;
    constexpr int LEAD = LEAD0 < NM ? LEAD0 : NM - 1;
    constexpr int PER = (NSL + (NM - LEAD) - 1) / (NM - LEAD);
    // One k-step (conv_ws.hip's step()): NM slots { one MFMA of `cur` ; at most RPS LDS reads for `nxt` ; PER stages of the split of nxt's A
    // fragments with the scales `sa_n` of the tile nxt belongs to }
    auto step = [&](const Frag &cur, Frag &nxt, int stage, const int ao0, const int ao1, const int bo, const float (&sa_n)[TM]) {
        constexpr int ta[3] = {1, 0, 0}, tb[3] = {0, 1, 0};      // a1*b0, a0*b1, a0*b0: smallest first
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
#pragma unroll
            for (int u = 0; u < RPS; ++u) {
                const int r = m * RPS + u;
                if (r >= NR) {
                } else if (r < NRA) {
                    if constexpr (GP)
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
                        const int stg = sl / (4 * TM), pr = sl % (4 * TM), i = pr / 4, q = pr % 4;
                        const float xa = raw[i][q >> 1][(q & 1) * 2], xb2 = raw[i][q >> 1][(q & 1) * 2 + 1];
                        if (stg == 0) {
                            nxt.a[i][0][q] = cvt_pk_f16(xa * sa_n[i], xb2 * sa_n[i]);
                        } else if (stg == 1) {     // residual of the SCALED value: fma(x, sa, -a0) is exact
                            const unsigned P = nxt.a[i][0][q];
                            ra[i][q] = fmaf(xa, sa_n[i], -f16_lo(P));
                            rb[i][q] = fmaf(xb2, sa_n[i], -f16_hi(P));
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

    Scales cur, nxt;
    scales_of(0, cur);
    Frag f0, f1;
    __builtin_amdgcn_s_barrier();            // the first chunk is in the LDS
    {   // operands of (tile 0, chunk 0, first k-step): not overlapped with anything
        const char *a_ptr = smem + wm * WM * 128;
        const char *b_ptr = smem + A_BYTES + wn * WN * 64;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (GP) {
                f0.a[i][0] = *reinterpret_cast<const uintx4 *>(a_ptr + i * 32 * 128 + a_foff[0][0]);
                f0.a[i][1] = *reinterpret_cast<const uintx4 *>(a_ptr + i * 32 * 128 + a_foff[0][1]);
                continue;
            }
            const floatx4 lo = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * 128 + a_foff[0][0]);
            const floatx4 hi = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * 128 + a_foff[0][1]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xa = q < 2 ? lo[2 * q] : hi[2 * q - 4], xb2 = q < 2 ? lo[2 * q + 1] : hi[2 * q - 3];
                const unsigned P0 = cvt_pk_f16(xa * cur.sa[i], xb2 * cur.sa[i]);
                f0.a[i][0][q] = P0;
                f0.a[i][1][q] = cvt_pk_f16(fmaf(xa, cur.sa[i], -f16_lo(P0)), fmaf(xb2, cur.sa[i], -f16_hi(P0)));
            }
        }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) f0.b[pl][j] = *reinterpret_cast<const uintx4 *>(b_ptr + (pl * BN + j * 32) * 64 + b_foff[0]);
    }
    int st = 0;
    for (int j = 0; j < T; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;
        nxt = cur;
        if (j + 1 < T) scales_of(j + 1, nxt);      // (its loads return during this tile's chunks)
        for (int k = 0; k < nchunks; ++k) {
            step(f0, f1, st, a_foff[1][0], a_foff[1][1], b_foff[1], cur.sa);          // k-step 0 of chunk k  ||  fetch + split k-step 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of chunk k has returned
            __builtin_amdgcn_s_barrier();
            st = st + 1 == NS ? 0 : st + 1;
            // k-step 1 of chunk k  ||  fetch + split k-step 0 of the NEXT chunk of the stream: the next tile's first one behind this tile's last
            if (k + 1 < nchunks)
                step(f1, f0, st, a_foff[0][0], a_foff[0][1], b_foff[0], cur.sa);
            else
                step(f1, f0, st, a_foff[0][0], a_foff[0][1], b_foff[0], nxt.sa);
        }
        // ---- epilogue of tile j: the producers' requests for the next tile are in flight meanwhile; its patches sit behind the stages
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float rowscale[TM][4], rowsplit[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) rowsplit[i][t] = 0.f;
        const bool split_out = p.yscale != nullptr;
        if (split_out) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float ys = split_scale_of(fmaf(p.ysplit_mul, cur.xmax_up[i], p.ysplit_add));
                const int mr = cur.m0 + wm * WM + i * 32 + (lane & 31);
                if (cur.n0 == 0 && wn == 0 && lane < 32 && mr < p.M) p.yscale[mr / hw] = ys;
#pragma unroll
                for (int t = 0; t < 4; ++t) rowsplit[i][t] = __shfl(ys, (lane >> 3) + 8 * t);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) rowscale[i][t] = __shfl(cur.inv_sa[i], (lane >> 3) + 8 * t);
        tile_epilogue<TM, TN, WM, WN, false, true>(p, acc, reinterpret_cast<float *>(smem + E_OFF), cur.m0, cur.n0, wm, wn, lane, wave, 0, rowscale,
                                                   rowsplit, split_out);
        cur = nxt;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
}

template <int BM, int BN, int NS>
int launch_wsp(ConvArgs p, int splits, hipStream_t stream) {
    const long long xbytes = (long long)p.N * p.H * p.W * p.x_ld * 4 + (long long)(p.pad * p.W + p.pad) * p.x_ld * 4;
    const long long wbytes = (long long)p.K * p.Kred * 2 * 2;
    if (xbytes >= 0xFFFFF000LL || wbytes >= 0xFFFFF000LL || p.R * p.S > 32) return PPY_ERR_UNSUPPORTED;
    if (splits > 1) return PPY_ERR_BAD_ARG;                       // (one split: the point is one launch without gaps)
    if (p.bn_part) return PPY_ERR_UNSUPPORTED;
    if (!vec_epilogue_ok(p)) return PPY_ERR_UNSUPPORTED;          // (K % 4 != 0: the caller's fall-back tile)
    if (p.yscale && p.ups) return PPY_ERR_BAD_ARG;
    if (p.yscale && (p.K % 32 != 0 || p.y_ld % 32 != 0)) return PPY_ERR_BAD_ARG;
    p.nstages = NS;
    p.chunks_total = p.R * p.S * (p.C / 32);
    p.chunks_per_split = p.chunks_total;
    const int ntiles = ceil_div(p.M, BM) * ceil_div(p.K, BN);
    p.panel_n = ppy_panel_n(p, BM, BN, 1);
    static const int n_cu = [] {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        return cus & ~7;
    }();
    const int grid = ntiles <= n_cu ? ntiles : n_cu;              // one workgroup per CU (114 KB of LDS), each a multiple-of-8 stride apart
    constexpr size_t lds = (size_t)NS * (BM * 128 + 2 * BN * 64) + 4 * 32 * LDS_LD * sizeof(float);
    if (p.xscale) {
        auto k = conv_wsp_kernel<BM, BN, NS, true>;
        static PpyLdsAttr attr;
        if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), 160 * 1024) != PPY_OK) return PPY_ERR_LAUNCH;
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, stream, p, ntiles);
    } else {
        auto k = conv_wsp_kernel<BM, BN, NS, false>;
        static PpyLdsAttr attr;
        if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), 160 * 1024) != PPY_OK) return PPY_ERR_LAUNCH;
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, stream, p, ntiles);
    }
    return ppy_launch_status();
}

}  // namespace

// local ids: 0 = 128x128 tile with three stages, 1 = 64x128 with four
int ppy_wsp_num_configs() { return 2; }

int ppy_wsp_dispatch(const ConvArgs &p, int c, int s, hipStream_t st) {
    if (!p.wf16 || ((uintptr_t)p.wf16 & 15) != 0 || !p.scale_f16 || !p.amax_in || (p.posb && !p.posb_f16)) return PPY_ERR_BAD_ARG;
    ConvArgs q = p;
    q.scale = p.scale_f16;
    q.posb = p.posb ? p.posb_f16 : nullptr;
    switch (c) {
        case 0: return launch_wsp<128, 128, 3>(q, s, st);
        case 1: return launch_wsp<64, 128, 4>(q, s, st);
    }
    return PPY_ERR_BAD_ARG;
}
