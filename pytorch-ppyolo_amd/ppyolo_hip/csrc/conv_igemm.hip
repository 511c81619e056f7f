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
#include "conv_shared.h"

namespace {

template <int BM, int BN, int WM, int WN, bool SPLIT, bool VEC>
__global__ void __launch_bounds__(64 * (BM / WM) * (BN / WN)) conv_igemm_kernel(const ConvArgs p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NW = (BM / WM) * (BN / WN);     // 4 waves (one per SIMD) or 8 (two per SIMD)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int RPP = NW * 8;                   // tile rows staged per pass (8 lanes x 16 B per row)
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile must be a multiple of the staging pass");
    constexpr int A_PER = BM / RPP, B_PER = BN / RPP;

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

    // ---- per-thread loader coordinates: row = (tid>>3) + RPP*j, 16-byte column tid&7 ----
    // Loads are UNCONDITIONAL (straight-line code, exact vmcnt bookkeeping): a padding tap or a
    // row beyond M reads the row's own centre pixel instead and is zeroed when it is written
    // to LDS.
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
    long long a_base[A_PER];
    int a_hi0[A_PER], a_wi0[A_PER];
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int mr = m0 + lrow + RPP * j;
        const int m = min(mr, p.M - 1);
        const int n = m / hw, rem = m - n * hw;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        a_hi0[j] = ho * p.stride - p.pad;
        a_wi0[j] = wo * p.stride - p.pad;
        a_base[j] = (((long long)n * p.H + a_hi0[j]) * p.W + a_wi0[j]) * p.x_ld + lc4;
        if (mr >= p.M) a_hi0[j] = -(1 << 20);   // every tap fails the bounds test -> zero rows
    }
    const float *b_row[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
        const int k = min(n0 + lrow + RPP * j, p.K - 1);   // clamp: rows >= K are masked at store
        b_row[j] = p.w + (long long)k * p.Kred + lc4;
    }

    // two staging register sets: loads run TWO chunks ahead of the MFMAs
    floatx4 ra0[A_PER], rb0[B_PER], ra1[A_PER], rb1[B_PER];
    unsigned ok0 = 0, ok1 = 0;
    const int RS = p.R * p.S;
    // chunk cursor of the loader (chunks are fetched strictly in order): channel chunk, tap (r, s)
    int l_cc = kc_begin / RS, l_tap = kc_begin - l_cc * RS;
    int l_r = l_tap / p.S, l_s = l_tap - l_r * p.S;
    const long long centre_off = (long long)(p.pad * p.W + p.pad) * p.x_ld;

    auto load_tiles = [&](floatx4 (&ra)[A_PER], floatx4 (&rb)[B_PER], unsigned &okm) {
        const int coff = l_cc * BK;
        const long long tap_off = (long long)(l_r * p.W + l_s) * p.x_ld + coff;
        const long long ctr_off = centre_off + coff;
        unsigned m = 0;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int hi = a_hi0[j] + l_r, wi = a_wi0[j] + l_s;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            m |= ok ? (1u << j) : 0u;
            ra[j] = *reinterpret_cast<const floatx4 *>(p.x + a_base[j] + (ok ? tap_off : ctr_off));
        }
        okm = m;
        const int woff = l_tap * p.C + coff;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) rb[j] = *reinterpret_cast<const floatx4 *>(b_row[j] + woff);
        // advance the cursor
        ++l_tap;
        ++l_s;
        if (l_s == p.S) { l_s = 0; ++l_r; }
        if (l_tap == RS) { l_tap = 0; l_r = 0; l_s = 0; ++l_cc; }
    };
    auto store_tiles = [&](int buf, const floatx4 (&ra)[A_PER], const floatx4 (&rb)[B_PER], unsigned okm) {
        const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < A_PER; ++j)
            *reinterpret_cast<floatx4 *>(sA + (buf * BM + lrow + RPP * j) * LDS_LD + lc4) =
                ((okm >> j) & 1u) ? ra[j] : zero;
#pragma unroll
        for (int j = 0; j < B_PER; ++j)
            *reinterpret_cast<floatx4 *>(sB + (buf * BN + lrow + RPP * j) * LDS_LD + lc4) = rb[j];
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
        load_tiles(ra0, rb0, ok0);
        if (kc_begin + 1 < kc_end) load_tiles(ra1, rb1, ok1);
        store_tiles(0, ra0, rb0, ok0);
        __syncthreads();
        // steady state, unrolled by two so the register sets are named statically:
        //   issue loads of chunk k+2 | MFMAs of chunk k (LDS buffer k&1) | store chunk k+1 -> LDS
        int left = kc_end - kc_begin;          // chunks not yet computed
        while (true) {
            if (left > 2) load_tiles(ra0, rb0, ok0);
            compute(0);
            if (left == 1) break;
            store_tiles(1, ra1, rb1, ok1);
            __syncthreads();
            --left;
            if (left > 2) load_tiles(ra1, rb1, ok1);
            compute(1);
            if (left == 1) break;
            store_tiles(0, ra0, rb0, ok0);
            __syncthreads();
            --left;
        }
        __syncthreads();
    }

    tile_epilogue<TM, TN, WM, WN, SPLIT, VEC>(p, acc, smem, m0, n0, wm, wn, lane, wave, split);
}

// ---------------------------------------------------------------------------------------
// LDS-DMA variant: operand tiles go HBM/L2 -> LDS directly (`buffer_load_dwordx4 ... lds`), no
// VGPR staging, no ds_write pass, and a real multi-stage pipeline with COUNTED vmcnt waits
// (hipcc's own bookkeeping of register-staged loads drained the queue at every chunk: it waited
// vmcnt(3) where vmcnt(7) was enough, which pinned the prefetch distance at one chunk).
//  * The DMA writes wave-uniform base + lane*16, i.e. 8 tile rows of 128 B per wave instruction,
//    so tiles are stored UNPADDED and bank conflicts are avoided with an XOR swizzle applied to
//    the per-lane SOURCE column (16-byte column c of row r lives in slot c ^ ((r>>1)&7)); the MFMA
//    fragment reads apply the same XOR.  16 rows x 16 B then cover all 64 banks once.
//  * Addressing is a buffer descriptor rebuilt per chunk on the scalar unit (base = tensor +
//    uniform tap/channel offset) plus a per-lane 32-bit row offset computed ONCE; padding taps and
//    rows beyond M use an out-of-range offset, for which the hardware writes zeros into LDS
//    (probed on MI355X: tools/probes/glds_probe.hip).  Per chunk and tile row the vector unit
//    only executes a bit test and a select.
//  * Pipeline (STAGES LDS buffers): wait for chunk k (counted vmcnt) -> barrier (also proves
//    everyone finished chunk k-1, whose buffer is then refilled with chunk k+STAGES-1) -> MFMAs.
//  * BKT = reduction depth of one pipeline stage (32 or 16 channels of one tap).  16 halves the LDS per
//    stage (128x128 tile: 16 KB) so that 3-4 workgroups fit on a CU; a DMA instruction then covers
//    16 rows x 64 B and the swizzle is c ^ ((r>>2)&3) over the 4 slots of a row.
template <int BM, int BN, int WM, int WN, int STAGES, int BKT, bool SPLIT, bool VEC>
__global__ void __launch_bounds__(64 * (BM / WM) * (BN / WN)) conv_igemm_glds_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // body uses device-only builtins (buffer descriptor, LDS DMA)
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NW = (BM / WM) * (BN / WN);
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(BKT == 32 || BKT == 16, "stage depth");
    constexpr int SLOTS = BKT / 4;                 // 16-byte slots per tile row
    constexpr int RPI = 64 / SLOTS;                // tile rows per wave DMA instruction (1 KB)
    static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "whole DMA instructions per wave");
    constexpr int A_PASS = BM / (RPI * NW), B_PASS = BN / (RPI * NW);
    constexpr int G = A_PASS + B_PASS;             // DMA instructions per wave per chunk
    constexpr int STAGE = (BM + BN) * BKT;         // floats per pipeline stage
    static_assert(STAGES == 2 || STAGES == 3, "2 or 3 LDS stages");
    typedef __attribute__((address_space(3))) void *lds_ptr;

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.K + BN - 1) / BN;
    // (an XCD-aware remap of blockIdx and s_setprio around the MFMA block were both measured on
    // MI355X and were neutral-to-negative here: -1..-3 %, so the plain order is kept)
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int kc_begin = split * p.chunks_per_split;
    const int kc_end = min(kc_begin + p.chunks_per_split, p.chunks_total);
    const int nchunks = kc_end - kc_begin;
    unsigned long long t_start = 0;
    if (p.trace) t_start = __builtin_amdgcn_s_memrealtime();     // 100 MHz, chip-wide

    // ---- per-lane DMA source offsets (bytes), fixed for the whole tile ----
    const int drow = lane / SLOTS, dslot = lane % SLOTS;
    auto swz = [](int row) { return BKT == 32 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
    const int hw = p.Ho * p.Wo;
    const unsigned OOB = 0xFFFFFFF0u;
    const long long bias = (long long)(p.pad * p.W + p.pad) * p.x_ld * 4;   // keeps offsets >= 0
    unsigned a_off[A_PASS], a_ok[A_PASS], b_off[B_PASS];
#pragma unroll
    for (int j = 0; j < A_PASS; ++j) {
        const int row = (j * NW + wave) * RPI + drow;       // row inside the tile
        const int scol = dslot ^ swz(row);                  // source 16-byte column for this slot
        const int mr = m0 + row;
        const int m = min(mr, p.M - 1);
        const int n = m / hw, rem = m - n * hw;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_off[j] = (unsigned)((((long long)n * p.H + hi0) * p.W + wi0) * p.x_ld * 4 + bias + scol * 16);
        unsigned okb = 0;
        if (mr < p.M) {
            for (int r = 0; r < p.R; ++r)
                for (int s2 = 0; s2 < p.S; ++s2)
                    if ((unsigned)(hi0 + r) < (unsigned)p.H && (unsigned)(wi0 + s2) < (unsigned)p.W)
                        okb |= 1u << (r * p.S + s2);
        }
        a_ok[j] = okb;
    }
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) {
        const int row = (j * NW + wave) * RPI + drow;
        const int scol = dslot ^ swz(row);
        const int k = min(n0 + row, p.K - 1);
        b_off[j] = (unsigned)((long long)k * p.Kred * 4 + scol * 16);
    }

    const int RS = p.R * p.S;
    int l_cc = kc_begin / RS, l_tap = kc_begin - l_cc * RS;
    int l_r = l_tap / p.S, l_s = l_tap - l_r * p.S;
    const char *xb = reinterpret_cast<const char *>(p.x) - bias;
    const char *wb = reinterpret_cast<const char *>(p.w);

    auto issue = [&](int stage) {
        const long long a_uni = ((long long)(l_r * p.W + l_s) * p.x_ld + l_cc * BKT) * 4;
        const long long b_uni = ((long long)l_tap * p.C + l_cc * BKT) * 4;
        __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(xb + a_uni), 0, 0xFFFFFF00u, 0x00020000);
        __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(wb + b_uni), 0, 0xFFFFFF00u, 0x00020000);
        float *sA = smem + stage * STAGE;
        float *sB = sA + BM * BKT;
#pragma unroll
        for (int j = 0; j < A_PASS; ++j) {
            const unsigned off = ((a_ok[j] >> l_tap) & 1u) ? a_off[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sA + (j * NW + wave) * RPI * BKT), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_PASS; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sB + (j * NW + wave) * RPI * BKT), 16, b_off[j], 0, 0, 0);
        ++l_tap;
        ++l_s;
        if (l_s == p.S) { l_s = 0; ++l_r; }
        if (l_tap == RS) { l_tap = 0; l_r = 0; l_s = 0; ++l_cc; }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment read offsets: row (lane&31), 16-byte slot ((2q + lane>>5) ^ swizzle(row))
    const int frow = lane & 31, fsw = swz(frow), fkh = lane >> 5;
    int foff[BKT / 8];
#pragma unroll
    for (int q = 0; q < BKT / 8; ++q) foff[q] = frow * BKT + (((2 * q + fkh) ^ fsw) << 2);

    auto compute = [&](int stage) {
        const float *a_ptr = smem + stage * STAGE + wm * WM * BKT;
        const float *b_ptr = smem + stage * STAGE + BM * BKT + wn * WN * BKT;
        // all fragment reads of the chunk are issued up front (distinct registers), so the MFMA
        // chain only waits on counted lgkmcnt instead of a read -> wait -> 4 MFMA lock-step
        floatx4 a[BKT / 8][TM], b[BKT / 8][TN];
#pragma unroll
        for (int q = 0; q < BKT / 8; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const floatx4 *>(a_ptr + i * 32 * BKT + foff[q]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[q][j] = *reinterpret_cast<const floatx4 *>(b_ptr + j * 32 * BKT + foff[q]);
        }
#pragma unroll
        for (int q = 0; q < BKT / 8; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i][t], b[q][j][t], acc[i][j], 0, 0, 0);
    };

    if (nchunks > 0) {
        // prologue: STAGES-1 chunks in flight
        issue(0);
        if (STAGES == 3 && nchunks > 1) issue(1);
        int stage = 0;
        for (int k = 0; k < nchunks; ++k) {
            // chunks issued so far: k .. min(k + STAGES - 2, nchunks - 1); wait for chunk k only
            if (STAGES == 3 && k + 1 < nchunks)
                wait_vmcnt<G>();
            else
                wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (k + STAGES - 1 < nchunks) {
                int st = stage + STAGES - 1;
                if (st >= STAGES) st -= STAGES;
                issue(st);
            }
            compute(stage);
            if (++stage == STAGES) stage = 0;
        }
        __builtin_amdgcn_s_barrier();
    }
    tile_epilogue<TM, TN, WM, WN, SPLIT, VEC>(p, acc, smem, m0, n0, wm, wn, lane, wave, split);
    if (p.trace && tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const long long b = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        p.trace[b * 4 + 0] = t_start;
        p.trace[b * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        p.trace[b * 4 + 2] = hwid | ((unsigned long long)xcc << 32);
        p.trace[b * 4 + 3] = 0;
    }
#endif
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
constexpr int kNumTiles = sizeof(kCfgs) / sizeof(kCfgs[0]);
// Configuration ids:
//   [0,7)   4-wave tiles, VGPR-staged loader                      (kCfgs above)
//   [7,14)  8-wave (512-thread) tiles, VGPR-staged loader: two waves per SIMD inside ONE
//           workgroup share the operand tiles -> half the L2->LDS traffic and barriers per MFMA
//           {128,128,32,64} {128,128,64,32} {128,64,32,32} {64,128,32,32} {256,64,64,32}
//           {256,128,64,64} {128,256,64,64}
//   [14,26) LDS-DMA loader (conv_igemm_glds_kernel), see kGlds below
struct GldsCfg {
    int bm, bn, wm, wn, stages;
};
constexpr GldsCfg kGlds[] = {
    {64, 64, 32, 32, 3},     // 14
    {64, 64, 32, 32, 2},     // 15
    {128, 64, 32, 32, 3},    // 16  (8 waves)
    {64, 128, 32, 32, 3},    // 17  (8 waves)
    {128, 128, 32, 64, 2},   // 18  (8 waves)
    {128, 128, 64, 32, 2},   // 19  (8 waves)
    {128, 128, 64, 64, 2},   // 20
    {128, 64, 64, 32, 3},    // 21
    {64, 128, 32, 64, 3},    // 22
    {128, 32, 32, 32, 3},    // 23
    {32, 128, 32, 32, 3},    // 24
    {256, 64, 64, 32, 2},    // 25  (8 waves)
    // stage depth 16 (BKT): [26,31)
    {128, 128, 64, 64, 2},   // 26  4 waves, 32 KB
    {128, 128, 64, 64, 3},   // 27  4 waves, 48 KB
    {128, 128, 64, 32, 3},   // 28  8 waves, 48 KB
    {128, 64, 64, 32, 3},    // 29  4 waves, 36 KB
    {64, 64, 32, 32, 3},     // 30  4 waves, 24 KB
};
constexpr int kNumGlds = sizeof(kGlds) / sizeof(kGlds[0]);
constexpr int kNumCfgs = 14 + kNumGlds;

template <int BM, int BN, int WM, int WN, bool SPLIT, bool VEC>
int launch_one(const ConvArgs &p, int splits, size_t lds, int tiles, hipStream_t stream) {
    auto k = conv_igemm_kernel<BM, BN, WM, WN, SPLIT, VEC>;
    static PpyLdsAttr attr;      // (the LDS size is a function of the template parameters)
    if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), (int)lds) != PPY_OK) return PPY_ERR_LAUNCH;
    hipLaunchKernelGGL(k, dim3(tiles, splits), dim3(64 * (BM / WM) * (BN / WN)), lds, stream, p);
    return PPY_OK;
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const ConvArgs &p, int splits, hipStream_t stream) {
    const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.K, BN);
    const bool vec = vec_epilogue_ok(p);
    int rc;
    if (splits > 1) {
        rc = vec ? launch_one<BM, BN, WM, WN, true, true>(p, splits, lds, tiles, stream)
                 : launch_one<BM, BN, WM, WN, true, false>(p, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
        launch_splitk_reduce(p, splits, vec, stream);
    } else {
        rc = vec ? launch_one<BM, BN, WM, WN, false, true>(p, splits, lds, tiles, stream)
                 : launch_one<BM, BN, WM, WN, false, false>(p, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
    }
    return ppy_launch_status();
}

template <int BM, int BN, int WM, int WN, int STAGES, int BKT, bool SPLIT, bool VEC>
int launch_glds_one(const ConvArgs &p, int splits, size_t lds, int tiles, hipStream_t stream) {
    auto k = conv_igemm_glds_kernel<BM, BN, WM, WN, STAGES, BKT, SPLIT, VEC>;
    static PpyLdsAttr attr;      // (the LDS size is a function of the template parameters)
    if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), (int)lds) != PPY_OK) return PPY_ERR_LAUNCH;
    hipLaunchKernelGGL(k, dim3(tiles, splits), dim3(64 * (BM / WM) * (BN / WN)), lds, stream, p);
    return PPY_OK;
}

template <int BM, int BN, int WM, int WN, int STAGES, int BKT = BK>
int launch_glds(ConvArgs p, int splits, hipStream_t stream) {
    // the per-lane DMA offsets are 32-bit: tensors must stay below 4 GB (- margin)
    const long long xbytes = (long long)p.N * p.H * p.W * p.x_ld * 4 + (long long)(p.pad * p.W + p.pad) * p.x_ld * 4;
    const long long wbytes = (long long)p.K * p.Kred * 4;
    if (xbytes >= 0xFFFFF000LL || wbytes >= 0xFFFFF000LL || p.R * p.S > 32) return PPY_ERR_UNSUPPORTED;
    constexpr int NW = (BM / WM) * (BN / WN);
    size_t lds = (size_t)STAGES * (BM + BN) * BKT * sizeof(float);
    const size_t epi = (size_t)NW * 32 * LDS_LD * sizeof(float);
    if (lds < epi) lds = epi;
    // chunk bookkeeping in units of this configuration's stage depth
    p.chunks_total = p.R * p.S * (p.C / BKT);
    p.chunks_per_split = ceil_div(p.chunks_total, splits);
    splits = ceil_div(p.chunks_total, p.chunks_per_split);
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.K, BN);
    const bool vec = vec_epilogue_ok(p);
    int rc;
    if (splits > 1) {
        rc = vec ? launch_glds_one<BM, BN, WM, WN, STAGES, BKT, true, true>(p, splits, lds, tiles, stream)
                 : launch_glds_one<BM, BN, WM, WN, STAGES, BKT, true, false>(p, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
        launch_splitk_reduce(p, splits, vec, stream);
    } else {
        rc = vec ? launch_glds_one<BM, BN, WM, WN, STAGES, BKT, false, true>(p, splits, lds, tiles, stream)
                 : launch_glds_one<BM, BN, WM, WN, STAGES, BKT, false, false>(p, splits, lds, tiles, stream);
        if (rc != PPY_OK) return rc;
    }
    return ppy_launch_status();
}


// Cost model (cycles on one CU) used when the caller does not force a configuration.
// A work item (tile x split) costs BM*BN/4 MFMA-cycles per chunk on a CU (4 SIMDs x 64
// FLOP/clk); items are dealt round-robin to 256 CUs.  Small tiles pay more LDS/L1 traffic per
// FLOP (eff), split-K pays a combine pass through HBM plus a kernel boundary.
void pick_config(const Geometry &g, int K, int *cfg_out, int *split_out) {
    static const double eff[kNumTiles] = {1.00, 0.95, 0.95, 0.88, 0.85, 0.80, 0.80};
    static const int split_opts[] = {1, 2, 3, 4, 6, 8, 9, 12, 16, 18};
    double best = 1e30;
    int bc = 3, bs = 1;
    for (int c = 0; c < kNumTiles; ++c) {
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
    // the LDS-DMA kernel of the same tile shape wins everywhere it was measured (the 32-channel stem
    // layers excepted); 128x128 is run with 8 waves
    static const int to_glds[kNumTiles] = {19, 16, 17, 15, 23, 23, 24};
    *cfg_out = to_glds[bc];
    *split_out = bs;
}

}  // namespace

// configuration ids [kNumCfgs, kNumCfgs + ppy_x3_num_configs()) select the split-bf16 kernels of conv_x3.hip, the ids after
// them the streaming kernel of conv_stream.hip (1x1, C = 64 / 128, f16x2 operands: ppy_conv2d_stream_first_config() + {0, 1}),
// then the patch kernel of conv_patch.hip (3x3 / stride 1, C = 32, K = 32 / 64, f16x2 operands), then the f16x2 tiles with
// specialised waves of conv_ws.hip
static int stream_first() { return kNumCfgs + ppy_x3_num_configs(); }
static int patch_first() { return stream_first() + ppy_stream_num_configs(); }
static int ws_first() { return patch_first() + ppy_patch_num_configs(); }
static int small_first() { return ws_first() + ppy_ws_num_configs(); }      // (round 6: csrc/conv_small.hip, behind every older id)
extern "C" int ppy_conv2d_num_configs(void) { return small_first() + ppy_small_num_configs(); }
extern "C" int ppy_conv2d_ws_first_config(void) { return ws_first(); }
extern "C" int ppy_conv2d_small_first_config(void) { return small_first(); }
extern "C" int ppy_conv2d_stream_first_config(void) { return stream_first(); }
extern "C" int ppy_conv2d_patch_first_config(void) { return patch_first(); }

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
    if (cfg >= ppy_conv2d_num_configs()) return PPY_ERR_BAD_ARG;
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
    if (c >= small_first()) return 0;      // (conv_small.hip splits the reduction inside the workgroup)
    return s > 1 ? (size_t)s * g.M * K * sizeof(float) : 0;
}

static int dispatch_cfg(const ConvArgs &p, int c, int s, hipStream_t st);
static unsigned long long *g_trace = nullptr;
// Debug hook (deliberately not in the public header): the LDS-DMA kernel writes {start, end} (100 MHz
// real-time counter), HW_ID and XCC_ID of every workgroup of the following launches to `buf`
// (4 x u64 per workgroup); tools/conv_trace.py turns that into a per-CU timeline.
extern "C" void ppy_debug_set_trace(unsigned long long *buf) { g_trace = buf; }

static int conv2d_impl(const float *x, int x_ld, const float *w_krsc, const void *w_x3,
                       const void *w_f16x2, const float *scale, const float *scale_f16x2,
                       const float *shift, const float *residual, int res_ld,
                       const float *posbias, const float *posbias_f16x2, float *y, int y_ld, int N, int H,
                       int W, int C, int K, int R, int S, int stride, int pad, int act, int upsample2x,
                       int cfg, int splitk, const float *amax_in, float *amax_out, void *ws,
                       size_t ws_bytes, void *stream, const float *x_split_scale, float *y_split_scale, float y_bound_mul,
                       float y_bound_add, const float *amax_in2 = nullptr) {
    ppy_drop_stale_error();
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
    // no measured choice for this shape: with split weights at hand the 128x64 bf16x3 tile (two workgroups per
    // CU) is the one that won most layers of the measured tables; narrow / shallow layers stay on the fp32 kernels
    if (cfg < 0 && w_x3 && K >= 48 && g.chunks >= 4) c = kNumCfgs + 4;
    if (cfg < 0 && w_f16x2 && scale_f16x2 && amax_in && (!posbias || posbias_f16x2) && K >= 48 && g.chunks >= 4)
        c = kNumCfgs + ppy_x3_f16_base() + 4;      // the same tile on the f16x2 kernel
    if (s > 1 && c < small_first()) {
        const size_t need = (size_t)s * g.M * K * sizeof(float);
        if (!ws || ws_bytes < need) return PPY_ERR_WORKSPACE;
    }
    ConvArgs p;
    p.x = x; p.w = w_krsc; p.w3 = (const unsigned short *)w_x3; p.wf16 = (const unsigned short *)w_f16x2;
    p.scale_f16 = scale_f16x2; p.posb_f16 = posbias_f16x2; p.amax_in = amax_in; p.amax_out = amax_out; p.scale = scale; p.shift = shift; p.res = residual; p.posb = posbias;
    p.amax_in2 = amax_in ? amax_in2 : nullptr;
    p.y = y; p.part = (float *)ws;
    p.x_ld = x_ld; p.res_ld = res_ld; p.y_ld = y_ld;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = g.Ho; p.Wo = g.Wo; p.K = K; p.R = R; p.S = S;
    p.stride = stride; p.pad = pad; p.act = act; p.ups = upsample2x ? 1 : 0;
    p.M = g.M; p.Kred = g.Kred; p.cchunks = C / BK; p.chunks_total = g.chunks;
    p.chunks_per_split = ceil_div(g.chunks, s);
    p.nstages = 2;
    p.trace = g_trace;
    p.xscale = x_split_scale; p.yscale = y_split_scale; p.ysplit_mul = y_bound_mul; p.ysplit_add = y_bound_add;
    hipStream_t st = (hipStream_t)stream;
    rc = dispatch_cfg(p, c, s, st);
    if (rc == PPY_ERR_UNSUPPORTED && c >= 14 && !x_split_scale && !y_split_scale) {
        // LDS-DMA kernel declined (tensor >= 4 GB: 32-bit DMA offsets): VGPR-staged kernel, 64x64 tiles
        rc = dispatch_cfg(p, 3, s, st);
    }
    return rc;
}

extern "C" int ppy_conv2d_bn_act_f32(const float *x, int x_ld, const float *w_krsc, const void *w_x3,
                                     const void *w_f16x2, const float *scale, const float *scale_f16x2,
                                     const float *shift, const float *residual, int res_ld,
                                     const float *posbias, const float *posbias_f16x2, float *y, int y_ld, int N, int H,
                                     int W, int C, int K, int R, int S, int stride, int pad, int act, int upsample2x,
                                     int cfg, int splitk, const float *amax_in, float *amax_out, void *ws,
                                     size_t ws_bytes, void *stream) {
    return conv2d_impl(x, x_ld, w_krsc, w_x3, w_f16x2, scale, scale_f16x2, shift, residual, res_ld, posbias, posbias_f16x2, y, y_ld, N, H,
                       W, C, K, R, S, stride, pad, act, upsample2x, cfg, splitk, amax_in, amax_out, ws, ws_bytes, stream, nullptr,
                       nullptr, 0.f, 0.f);
}

// The same launch with PRE-SPLIT tensors on one or both sides (f16x2 tile kernels, explicit cfg, one split):
// x_split_scale != NULL: x holds finished operands -- per pixel and 32-channel group 32 fp16 first terms then 32 fp16 second
// terms of x * x_split_scale[n] (what a producer launch with y_split_scale wrote);  y_split_scale != NULL: y is WRITTEN in that
// form for its one consumer, with y_split_scale[n] = the power of two that puts  y_bound_mul * max|x_n| + y_bound_add  (a
// static bound of |y|: y_bound_mul >= max_k |scale_k| * sum|w_k|, y_bound_add >= max_k |shift_k| (+ the position bias)) into
// [2^13, 2^14).  PPY_ERR_BAD_ARG when the chosen kernel cannot read / write such tensors -- never a silent reinterpretation.
extern "C" int ppy_conv2d_bn_act_split_f32(const float *x, int x_ld, const float *w_krsc, const void *w_x3,
                                           const void *w_f16x2, const float *scale, const float *scale_f16x2,
                                           const float *shift, const float *residual, int res_ld,
                                           const float *posbias, const float *posbias_f16x2, float *y, int y_ld, int N, int H,
                                           int W, int C, int K, int R, int S, int stride, int pad, int act, int upsample2x,
                                           int cfg, int splitk, const float *amax_in, float *amax_out, void *ws,
                                           size_t ws_bytes, void *stream, const float *x_split_scale, float *y_split_scale,
                                           float y_bound_mul, float y_bound_add, const float *amax_in2) {
    // (amax_in2 alone -- a plain fp32 launch whose input has two tracked blocks -- keeps the freedoms of ppy_conv2d_bn_act_f32)
    if (!x_split_scale && !y_split_scale)
        return conv2d_impl(x, x_ld, w_krsc, w_x3, w_f16x2, scale, scale_f16x2, shift, residual, res_ld, posbias, posbias_f16x2, y, y_ld, N,
                           H, W, C, K, R, S, stride, pad, act, upsample2x, cfg, splitk, amax_in, amax_out, ws, ws_bytes, stream, nullptr,
                           nullptr, 0.f, 0.f, amax_in2);
    PPY_CHECK_ARG(cfg >= 0 && (splitk <= 1 || cfg >= small_first()) && !(upsample2x && y_split_scale));      // (conv_small.hip: the k-parts never leave the launch)
    PPY_CHECK_ARG(!y_split_scale || (y_bound_mul >= 0.f && y_bound_add >= 0.f && K % 32 == 0 && y_ld % 32 == 0 && ((uintptr_t)y & 127) == 0));
    PPY_CHECK_ARG(!x_split_scale || (C % 32 == 0 && x_ld % 32 == 0 && ((uintptr_t)x & 127) == 0));
    return conv2d_impl(x, x_ld, w_krsc, w_x3, w_f16x2, scale, scale_f16x2, shift, residual, res_ld, posbias, posbias_f16x2, y, y_ld, N, H,
                       W, C, K, R, S, stride, pad, act, upsample2x, cfg, splitk, amax_in, amax_out, ws, ws_bytes, stream, x_split_scale,
                       y_split_scale, y_bound_mul, y_bound_add, amax_in2);
}

static int dispatch_cfg(const ConvArgs &p, int c, int s, hipStream_t st) {
    // (statistics from the epilogue exist in the f16x2 kernels only: conv_x3.hip's tiles, conv_stream.hip, conv_patch.hip, conv_ws.hip)
    if (p.bn_part && c < kNumCfgs + ppy_x3_f16_base()) return PPY_ERR_UNSUPPORTED;
    // pre-split tensors exist on the f16x2 tiles (conv_x3.hip, conv_ws.hip) only: anything else would misread the bytes
    if ((p.xscale || p.yscale) && (c < kNumCfgs + ppy_x3_f16_base() || (c >= stream_first() && c < ws_first()))) return PPY_ERR_BAD_ARG;
    if (c >= small_first()) return ppy_small_dispatch(p, c - small_first(), s, st);
    if (c >= ws_first()) return ppy_ws_dispatch(p, c - ws_first(), s, st);
    if (c >= patch_first()) return s == 1 ? ppy_patch_dispatch(p, c - patch_first(), st) : PPY_ERR_BAD_ARG;
    if (c >= stream_first()) return s == 1 ? ppy_stream_dispatch(p, c - stream_first(), nullptr, 0, st) : PPY_ERR_BAD_ARG;
    if (c >= kNumCfgs) return ppy_x3_dispatch(p, c - kNumCfgs, s, st);
    switch (c) {
        case 0: return launch_cfg<128, 128, 64, 64>(p, s, st);
        case 1: return launch_cfg<128, 64, 64, 32>(p, s, st);
        case 2: return launch_cfg<64, 128, 32, 64>(p, s, st);
        case 3: return launch_cfg<64, 64, 32, 32>(p, s, st);
        case 4: return launch_cfg<256, 32, 64, 32>(p, s, st);
        case 5: return launch_cfg<128, 32, 32, 32>(p, s, st);
        case 6: return launch_cfg<32, 128, 32, 32>(p, s, st);
        case 7: return launch_cfg<128, 128, 32, 64>(p, s, st);
        case 8: return launch_cfg<128, 128, 64, 32>(p, s, st);
        case 9: return launch_cfg<128, 64, 32, 32>(p, s, st);
        case 10: return launch_cfg<64, 128, 32, 32>(p, s, st);
        case 11: return launch_cfg<256, 64, 64, 32>(p, s, st);
        case 12: return launch_cfg<256, 128, 64, 64>(p, s, st);
        case 13: return launch_cfg<128, 256, 64, 64>(p, s, st);
        case 14: return launch_glds<64, 64, 32, 32, 3>(p, s, st);
        case 15: return launch_glds<64, 64, 32, 32, 2>(p, s, st);
        case 16: return launch_glds<128, 64, 32, 32, 3>(p, s, st);
        case 17: return launch_glds<64, 128, 32, 32, 3>(p, s, st);
        case 18: return launch_glds<128, 128, 32, 64, 2>(p, s, st);
        case 19: return launch_glds<128, 128, 64, 32, 2>(p, s, st);
        case 20: return launch_glds<128, 128, 64, 64, 2>(p, s, st);
        case 21: return launch_glds<128, 64, 64, 32, 3>(p, s, st);
        case 22: return launch_glds<64, 128, 32, 64, 3>(p, s, st);
        case 23: return launch_glds<128, 32, 32, 32, 3>(p, s, st);
        case 24: return launch_glds<32, 128, 32, 32, 3>(p, s, st);
        case 25: return launch_glds<256, 64, 64, 32, 2>(p, s, st);
        case 26: return launch_glds<128, 128, 64, 64, 2, 16>(p, s, st);
        case 27: return launch_glds<128, 128, 64, 64, 3, 16>(p, s, st);
        case 28: return launch_glds<128, 128, 64, 32, 3, 16>(p, s, st);
        case 29: return launch_glds<128, 64, 64, 32, 3, 16>(p, s, st);
        case 30: return launch_glds<64, 64, 32, 32, 3, 16>(p, s, st);
    }
    return PPY_ERR_BAD_ARG;
}

// The streaming 1x1 kernel with its second output: y as ppy_conv2d_bn_act_f32 gives it (f16x2 operands, C = 64, stride 1),
// and `pooled` = the 2x2 / stride 2 average of y (the AvgPool2d of the ResNet-vd shortcut, reference model/resnet_vd.py:29-33)
// written from the same epilogue.  pooled == NULL: y only.
extern "C" int ppy_conv1x1_expand_f32(const float *x, int x_ld, const void *w_f16x2, const float *scale_f16x2,
                                      const float *shift, const float *residual, int res_ld, float *y, int y_ld,
                                      float *pooled, int pooled_ld, int N, int H, int W, int C, int K, int act, int variant,
                                      const float *amax_in, float *amax_out, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && w_f16x2 && scale_f16x2 && shift && y && amax_in);
    Geometry g;
    if (!conv_geometry(N, H, W, C, K, 1, 1, 1, 0, &g)) return PPY_ERR_BAD_ARG;
    PPY_CHECK_ARG(x_ld >= C && y_ld >= K && (!residual || res_ld >= K));
    PPY_CHECK_ARG(act == PPY_ACT_NONE || act == PPY_ACT_RELU || act == PPY_ACT_LEAKY);
    ConvArgs p;
    p.x = x; p.w = nullptr; p.w3 = nullptr; p.wf16 = (const unsigned short *)w_f16x2;
    p.scale_f16 = scale_f16x2; p.posb_f16 = nullptr; p.amax_in = amax_in; p.amax_out = amax_out; p.scale = scale_f16x2; p.shift = shift;
    p.res = residual; p.posb = nullptr; p.y = y; p.part = nullptr;
    p.x_ld = x_ld; p.res_ld = res_ld; p.y_ld = y_ld;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = H; p.Wo = W; p.K = K; p.R = 1; p.S = 1;
    p.stride = 1; p.pad = 0; p.act = act; p.ups = 0;
    p.M = g.M; p.Kred = C; p.cchunks = C / BK; p.chunks_total = g.chunks; p.chunks_per_split = g.chunks;
    p.nstages = 2;
    p.trace = nullptr;
    return ppy_stream_dispatch(p, variant, pooled, pooled_ld, (hipStream_t)stream);
}


// Training forward of a FROZEN 1x1 Conv2dUnit on the streaming kernel without the raw tensor (round 4; reference
// model/custom_layers.py:243-253 with the BatchNorm2d in training mode): ppy_conv1x1_stats_f32 = the convolution's BatchNorm
// partials only (as ppy_conv2d_train_fwd_f32 writes them, nothing else is stored); ppy_conv1x1_bn_apply_f32 = the convolution
// again with y = act((conv + bias - mean) * (invstd * gamma) + beta [+ residual]) from its epilogue -- value for value what
// ppy_conv2d_train_fwd_f32 + ppy_bn_train_apply_f32 give.
static long long bn_slice_capacity(long long M);
extern "C" size_t ppy_conv2d_bn_partials_bytes(long long M, int K);
static int stream_args(ConvArgs &p, const float *x, int x_ld, const void *w_f16x2, const float *scale_f16x2, const float *bias, int N, int H,
                       int W, int C, int K, const float *amax_in) {
    Geometry g;
    if (!conv_geometry(N, H, W, C, K, 1, 1, 1, 0, &g)) return PPY_ERR_BAD_ARG;
    if (x_ld < C) return PPY_ERR_BAD_ARG;
    p.x = x; p.w = nullptr; p.w3 = nullptr; p.wf16 = (const unsigned short *)w_f16x2;
    p.scale_f16 = scale_f16x2; p.posb_f16 = nullptr; p.amax_in = amax_in; p.amax_out = nullptr; p.scale = scale_f16x2; p.shift = bias;
    p.res = nullptr; p.posb = nullptr; p.y = nullptr; p.part = nullptr;
    p.x_ld = x_ld; p.res_ld = 0; p.y_ld = K;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = H; p.Wo = W; p.K = K; p.R = 1; p.S = 1;
    p.stride = 1; p.pad = 0; p.act = PPY_ACT_NONE; p.ups = 0;
    p.M = g.M; p.Kred = C; p.cchunks = C / BK; p.chunks_total = g.chunks; p.chunks_per_split = g.chunks;
    p.nstages = 2;
    p.trace = nullptr;
    return PPY_OK;
}

extern "C" int ppy_conv1x1_stats_f32(const float *x, int x_ld, const void *w_f16x2, const float *scale_f16x2, const float *bias, int N, int H,
                                     int W, int C, int K, int variant, const float *amax_in, float *bn_partials, size_t bn_partials_bytes,
                                     int *bn_slices, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && w_f16x2 && scale_f16x2 && bias && amax_in && bn_partials && bn_slices);
    ConvArgs p;
    const int rc = stream_args(p, x, x_ld, w_f16x2, scale_f16x2, bias, N, H, W, C, K, amax_in);
    if (rc != PPY_OK) return rc;
    if (bn_partials_bytes < ppy_conv2d_bn_partials_bytes(p.M, K)) return PPY_ERR_WORKSPACE;
    p.y = const_cast<float *>(x);           // (never written: bn_nostore; the vector-epilogue alignment checks want a pointer)
    p.bn_part = bn_partials;
    p.bn_slices_host = bn_slices;
    p.bn_capacity = (int)bn_slice_capacity(p.M);
    p.bn_nostore = 1;
    *bn_slices = 0;
    return ppy_stream_dispatch(p, variant, nullptr, 0, (hipStream_t)stream);
}

extern "C" int ppy_conv1x1_bn_apply_f32(const float *x, int x_ld, const void *w_f16x2, const float *scale_f16x2, const float *bias,
                                        const float *mean, const float *invstd, const float *gamma, const float *beta, const float *residual,
                                        int res_ld, float *y, int y_ld, int N, int H, int W, int C, int K, int act, int variant,
                                        const float *amax_in, float *amax_out, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && w_f16x2 && scale_f16x2 && bias && mean && invstd && gamma && beta && y && amax_in);
    PPY_CHECK_ARG(act == PPY_ACT_NONE || act == PPY_ACT_RELU || act == PPY_ACT_LEAKY);
    ConvArgs p;
    const int rc = stream_args(p, x, x_ld, w_f16x2, scale_f16x2, bias, N, H, W, C, K, amax_in);
    if (rc != PPY_OK) return rc;
    PPY_CHECK_ARG(y_ld >= K && (!residual || res_ld >= K));
    p.y = y; p.y_ld = y_ld; p.res = residual; p.res_ld = res_ld; p.act = act; p.amax_out = amax_out;
    p.bn_mean = mean; p.bn_invstd = invstd; p.bn_gamma = gamma; p.bn_beta = beta;
    return ppy_stream_dispatch(p, variant, nullptr, 0, (hipStream_t)stream);
}

extern "C" int ppy_conv3x3_maxpool_f32(const float *x, int x_ld, const void *w_f16x2, const float *scale_f16x2, const float *shift,
                                       float *pooled, int pooled_ld, int N, int H, int W, int C, int K, int act, const float *amax_in,
                                       float *amax_out, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && w_f16x2 && scale_f16x2 && shift && pooled && amax_in);
    Geometry g;
    if (!conv_geometry(N, H, W, C, K, 3, 3, 1, 1, &g)) return PPY_ERR_BAD_ARG;
    PPY_CHECK_ARG(x_ld >= C && pooled_ld >= K);
    PPY_CHECK_ARG(act == PPY_ACT_NONE || act == PPY_ACT_RELU || act == PPY_ACT_LEAKY);
    ConvArgs p;
    p.x = x; p.w = nullptr; p.w3 = nullptr; p.wf16 = (const unsigned short *)w_f16x2;
    p.scale_f16 = scale_f16x2; p.posb_f16 = nullptr; p.amax_in = amax_in; p.amax_out = amax_out; p.scale = scale_f16x2; p.shift = shift;
    p.res = nullptr; p.posb = nullptr; p.y = pooled; p.part = nullptr;
    p.x_ld = x_ld; p.res_ld = 0; p.y_ld = pooled_ld;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = H; p.Wo = W; p.K = K; p.R = 3; p.S = 3;
    p.stride = 1; p.pad = 1; p.act = act; p.ups = 0;
    p.M = g.M; p.Kred = 9 * C; p.cchunks = C / BK; p.chunks_total = g.chunks; p.chunks_per_split = g.chunks;
    p.nstages = 2;
    p.trace = nullptr;
    return ppy_patch_maxpool_dispatch(p, (H - 1) / 2 + 1, (W - 1) / 2 + 1, (hipStream_t)stream);
}

// Training-mode forward of Conv2dUnit's convolution (reference model/custom_layers.py:243-253: conv (+ bias) in front of a
// BatchNorm2d on batch statistics): y = conv(x, w) + bias on an f16x2 kernel (cfg: any f16x2 id -- tiles, streaming 1x1, stem
// patch, specialised waves; one split), AND the first pass of the BatchNorm from the same epilogue -- (n, mean, M2) of every channel per wave row-tile in
// bn_partials [*bn_slices][K][3] (ppy_conv2d_bn_partials_bytes(M, K) bytes are always enough), to be merged by
// ppy_bn_train_stats_merge_f32.  The separate statistics pass over y (ppy_bn_train_stats_f32) re-reads the whole tensor.
// scale_f16x2: from ppy_conv2d_split_weights_f16x2 with scale = 1.  PPY_ERR_UNSUPPORTED for any other kernel family.
// slices of 32 rows, plus what row-tiles (<= 256 rows) and the stem patch kernel's 8 x 32-pixel tiles over-cover at the borders
static long long bn_slice_capacity(long long M) { return 4 * ((M + 31) / 32) + 64; }
extern "C" size_t ppy_conv2d_bn_partials_bytes(long long M, int K) {
    return (size_t)(bn_slice_capacity(M) + (bn_slice_capacity(M) + 63) / 64) * K * 3 * sizeof(float);      // + the first merge level's output
}

extern "C" int ppy_conv2d_train_fwd_f32(const float *x, int x_ld, const float *w_krsc, const void *w_f16x2, const float *scale_f16x2,
                                        const float *bias, float *y, int y_ld, int N, int H, int W, int C, int K, int R, int S, int stride,
                                        int pad, int cfg, const float *amax_in, float *bn_partials, size_t bn_partials_bytes,
                                        int *bn_slices, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x && w_krsc && w_f16x2 && scale_f16x2 && bias && y && amax_in && bn_partials && bn_slices && cfg >= 0);
    Geometry g;
    if (!conv_geometry(N, H, W, C, K, R, S, stride, pad, &g)) return PPY_ERR_BAD_ARG;
    PPY_CHECK_ARG(x_ld >= C && x_ld % 4 == 0 && y_ld >= K && ((uintptr_t)x & 15) == 0);
    PPY_CHECK_ARG(cfg < ppy_conv2d_num_configs());
    if (bn_partials_bytes < ppy_conv2d_bn_partials_bytes(g.M, K)) return PPY_ERR_WORKSPACE;
    ConvArgs p;
    p.x = x; p.w = w_krsc; p.w3 = nullptr; p.wf16 = (const unsigned short *)w_f16x2;
    p.scale_f16 = scale_f16x2; p.posb_f16 = nullptr; p.amax_in = amax_in; p.amax_out = nullptr; p.scale = scale_f16x2; p.shift = bias;
    p.res = nullptr; p.posb = nullptr; p.y = y; p.part = nullptr;
    p.x_ld = x_ld; p.res_ld = 0; p.y_ld = y_ld;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = g.Ho; p.Wo = g.Wo; p.K = K; p.R = R; p.S = S;
    p.stride = stride; p.pad = pad; p.act = PPY_ACT_NONE; p.ups = 0;
    p.M = g.M; p.Kred = g.Kred; p.cchunks = C / BK; p.chunks_total = g.chunks; p.chunks_per_split = g.chunks;
    p.nstages = 2;
    p.trace = nullptr;
    p.bn_part = bn_partials;
    p.bn_slices_host = bn_slices;
    p.bn_capacity = (int)bn_slice_capacity(g.M);
    *bn_slices = 0;
    return dispatch_cfg(p, cfg, 1, (hipStream_t)stream);
}
