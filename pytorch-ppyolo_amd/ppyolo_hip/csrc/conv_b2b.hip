// Two convolutions back to back in ONE launch (round 5): conv A = 3x3 / stride 1 / pad 1, CA -> KA, + BatchNorm + ReLU, feeding
// conv B = 1x1, KA -> KB, + BatchNorm + shortcut + ReLU -- conv2 -> conv3 of a ResNet-vd identity bottleneck (reference
// model/resnet_vd.py:81-87: relu(bn3(conv3(relu(bn2(conv2(t1))))) + x)).
//
// Why: conv3 of stage 2 (C64 -> K256 at 152x152) moves 425 MB for 6 GFLOP and is HBM-bound, conv2 in front of it is MFMA-bound, and
// as two launches each ends / starts with a chip-wide phase in which the other resource idles.  Here a workgroup
// owns 128 output pixels: it runs conv A's implicit GEMM over them (the f16x2 tiles' loader and operand layouts: LDS-DMA,
// XOR-swizzled tiles, pre-split "GP" input), keeps the 128 x KA result ON CHIP -- scaled, split into its two fp16 terms and written
// to LDS as the A operand of the second GEMM, never to HBM -- and multiplies it with conv B's weights in two column halves: while
// the first half's 128 x 128 outputs stream out (with the shortcut rows requested before the multiplication), the second half's
// weights land.  With two workgroups per CU one is in its MFMA phase while the other stores.  The intermediate tensor (47 MB at
// 152x152x64, written once and read once before) and one launch disappear.
// POOL: the tile's rows are 32 blocks of 2x2 pixels (row r of a wave = block r & 7, position r >> 3, as conv_stream.hip), which
// puts the four pixels of a block into ONE lane after the transposition: the vd shortcut's AvgPool2d(2, 2) of this output
// (reference model/resnet_vd.py:29-33) is written from the epilogue, (((a + b) + c) + d) * 0.25 as stem_pool.hip evaluates it.
//
// Arithmetic: both GEMMs are the f16x2 scheme of conv_x3.hip (three products a1*b0 + a0*b1 + a0*b0 per multiply-add on
// v_mfma_f32_32x32x16_f16, fp32 accumulate).  Conv A is the same sum in the same order as the stand-alone tile; the intermediate is
// scaled per image by a power of two from the STATIC bound |t| <= t_mul * max|x| + t_add (as a pre-split link, conv_shared.h), where
// the stand-alone conv3 scales by the tracked maximum: results agree to fp32 rounding, not bit for bit.
#include "conv_shared.h"

namespace {

struct B2bArgs {
    const float *x;                       // conv A input, PRE-SPLIT by its producer (conv_shared.h split store): [N, H, W, x_ld]
    const unsigned short *wA, *wB;        // f16x2 weight planes: A [2][9 * CA / 32][KA][32], B [2][KA / 32][KB][32]
    const float *scaleA, *shiftA, *scaleB, *shiftB;      // folded BatchNorm, the weight scales divided out (ppy_conv2d_split_weights_f16x2)
    const float *res;                     // shortcut [N, H, W, res_ld]
    float *y;
    float *pool;                          // POOL: [N, H/2, W/2, pool_ld]
    const float *xscale;                  // [N] per-image scale of the pre-split input
    const float *amax_in;                 // tracked per-image max|x| of the input tensor
    float *amax_out;                      // per-image max|y| slots, or NULL
    float t_mul, t_add;                   // static bound of the intermediate
    int x_ld, res_ld, y_ld, pool_ld, N, H, W, M;
    int skip;                             // timing experiments only (PPY_B2B_SKIP): 1 = no conv A loop, 2 = nothing behind it, 4 = no stores
};

template <int CA, int KA, int KB, bool POOL>
__global__ void __launch_bounds__(256, 2) conv_b2b_kernel(const B2bArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, NW = 4;
    // round 6: the output tile is 8 x 16 pixels of ONE image and conv A reads its input from a WINDOW of (8 + 2) x (16 + 2) pixels that is
    // brought into the LDS once -- the nine taps are shifted reads of it -- instead of one 128-row operand tile per tap: with K = 64 the
    // main loop was bound by its L2 -> LDS traffic (18 chunks x 24 KB per tile; 9 TB/s chip-wide); now 45 KB of window + 18 x 8 KB of weights
    constexpr int TY = 8, TX = 16, WX = TX + 2, WPIX = (TY + 2) * WX;      // (152 = 8 x 19: one dimension of a 128-pixel tile is ragged either way)
    constexpr int TNA = KA / 32, HB = KB / 2, TNH = HB / 32;         // conv B in two column halves of HB channels
    constexpr int CCH = CA / 32, NCH = 9 * CCH, KCH = KA / 32;       // chunks of conv A / of conv B's reduction
    constexpr int WIN_CC = WPIX * 128, WIN_BYTES = CCH * WIN_CC;      // window: per 32-channel group a pixel's 128 bytes (first terms | second terms)
    constexpr int WIN_UNITS = CCH * WPIX * 8, WIN_PASS = (WIN_UNITS + 64 * NW - 1) / (64 * NW);      // 16-byte pieces; DMA instructions per wave
    constexpr int B_PASS = (2 * KA) / (16 * NW), G = B_PASS;          // conv A's weights of a chunk: DMA instructions per wave
    constexpr int BST = 2 * KA * 64, BS_OFF = WIN_PASS * NW * 1024;   // a stage of them; three stages behind the window
    constexpr int A2_BYTES = KCH * BM * 128;                         // the intermediate as GEMM B's A operand
    constexpr int W_OFF = A2_BYTES, W_BYTES = KCH * 2 * HB * 64;     // one column half of conv B's planes: [chunk][plane][HB][32]
    constexpr int E_OFF = W_OFF + W_BYTES;                           // transposition patches: 32 x 32 floats per wave, XOR-swizzled
    constexpr int B2_PASS = W_BYTES / (1024 * NW);
    static_assert(TY * TX == BM && TY == 2 * NW && TX == 16, "a wave owns two tile rows as eight 2x2 blocks");
    static_assert(KA % 32 == 0 && KB % 64 == 0 && CA % 32 == 0 && (2 * KA) % (16 * NW) == 0 && W_BYTES % (1024 * NW) == 0 && HB % 16 == 0, "shapes");
    static_assert(BS_OFF >= WIN_BYTES && BS_OFF + 3 * BST <= E_OFF + NW * 4096, "window + weight stages fit the allocation");
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem_b2b[];
    char *smem = smem_b2b;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_id;
    {   // XCD-contiguous tile order (conv_x3.hip)
        const int nb = (int)gridDim.x, q = nb >> 3, r = nb & 7;
        const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
        tile_id = xcd * q + min(xcd, r) + idx;
    }
    const int hw = p.H * p.W;
    // tile -> (image, tile row, tile column); tile row R (0..127) -> pixel: wave R >> 5 owns the tile's rows 2w, 2w + 1 as eight 2x2 blocks,
    // row r = R & 31 of the wave is block r & 7, position r >> 3 inside it (the four rows r, r + 8, r + 16, r + 24 a lane finishes after the
    // transposition are ONE block: what the pooled output needs)
    const int tiles_x = (p.W + TX - 1) / TX, tiles_y = (p.H + TY - 1) / TY;
    const int t_n = tile_id / (tiles_x * tiles_y), t_rem = tile_id - t_n * (tiles_x * tiles_y);
    const int t_y = t_rem / tiles_x, t_x = t_rem - t_y * tiles_x;
    const int y0 = t_y * TY, x0 = t_x * TX;
    auto pixel_of = [&](int R) -> int {
        const int w = R >> 5, r = R & 31;
        const int y = y0 + 2 * w + (r >> 4), x = x0 + 2 * (r & 7) + ((r >> 3) & 1);
        return (y < p.H && x < p.W) ? t_n * hw + y * p.W + x : -1;
    };
    const unsigned OOB = 0xFFFFFFF0u;
    // ---- the window: piece q = ((pass * NW + wave) * 64 + lane) is 16-byte slot q & 7 of window pixel (q >> 3) % WPIX of channel group
    // (q >> 3) / WPIX, and lands at byte 16 q; the slot is swizzled on the SOURCE side (content slot c of pixel wp sits at c ^ ((wp >> 1) & 7):
    // the sixteen lanes of a fragment read -- consecutive window pixels -- hit sixteen different bank groups).  Pixels outside the image: zeros.
    {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, 0xFFFFFF00u, 0x00020000);
#pragma unroll
        for (int d = 0; d < WIN_PASS; ++d) {
            const int q = (d * NW + wave) * 64 + lane;
            const int pq = q >> 3, cc = pq / WPIX, wp = pq - cc * WPIX;
            const int wy = wp / WX, wx = wp - wy * WX;
            const int y = y0 - 1 + wy, x = x0 - 1 + wx;
            const bool ok = q < WIN_UNITS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            const unsigned off = ok ? (unsigned)(((t_n * p.H + y) * p.W + x) * (p.x_ld * 4) + cc * 128 + (((q & 7) ^ ((wp >> 1) & 7)) << 4)) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + (d * NW + wave) * 1024), 16, off, 0, 0, 0);
        }
    }
    unsigned b_off[B_PASS];
    {
        const int drow = lane >> 2, dslot = lane & 3;
        const long long plane_bytes = (long long)KA * (9 * CA) * 2;
#pragma unroll
        for (int j = 0; j < B_PASS; ++j) {
            const int rb = (j * NW + wave) * 16 + drow;
            const int plane = rb / KA, nrow = rb - plane * KA;
            const int scol = dslot ^ ((rb >> 2) & 3);
            b_off[j] = (unsigned)(plane * plane_bytes + (long long)nrow * 64 + scol * 16);
        }
    }
    const char *wb = reinterpret_cast<const char *>(p.wA);
    auto issue = [&](int stage, int kc) {           // conv A's weights of chunk kc = (channel chunk cc, tap): cc outer, tap inner
        const int cc = kc / 9, tap = kc - cc * 9;
        const long long b_uni = ((long long)tap * CCH + cc) * KA * 64;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(wb + b_uni), 0, 0xFFFFFF00u, 0x00020000);
#pragma unroll
        for (int j = 0; j < B_PASS; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(smem + BS_OFF + stage * BST + (j * NW + wave) * 1024), 16, b_off[j], 0, 0, 0);
    };

    // fragment reads.  A: this lane's MFMA row is the wave's row frow = its pixel (2 wave + (frow >> 4), 2 (frow & 7) + ((frow >> 3) & 1)) of the
    // tile = window pixel wp0 at tap (0, 0); tap (r, s) reads window pixel wp0 + r WX + s.  k-step ks, lane half h: first terms in content
    // slot 2 ks + h, second terms in 4 + 2 ks + h.  B (conv_x3.hip): row frow of the plane, slot 2 ks + h swizzled by (frow >> 2) & 3.
    const int frow = lane & 31, fkh = lane >> 5;
    const int wp0 = (2 * wave + (frow >> 4)) * WX + 2 * (frow & 7) + ((frow >> 3) & 1);
    const int a_sw = (frow >> 1) & 7, b_sw = (frow >> 2) & 3;
    int a_foff[2][2], b_foff[2];          // (a_foff: GEMM B's A operand, the intermediate, by tile row -- conv_x3.hip's GP form)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        a_foff[s][0] = frow * 128 + (((2 * s + fkh) ^ a_sw) << 4);
        a_foff[s][1] = frow * 128 + (((4 + 2 * s + fkh) ^ a_sw) << 4);
        b_foff[s] = frow * 64 + (((2 * s + fkh) ^ b_sw) << 4);
    }

    // per-image scales of this lane's tile row (row lane & 31 of the wave's 32)
    float inv_sa, s2, inv_s2;
    {
        const int n = min(t_n, p.N - 1);          // (a tile lies inside one image)
        inv_sa = pow2_inverse(p.xscale[n]);
        s2 = split_scale_of(fmaf(p.t_mul, pow2_above(amax_read(p.amax_in, n)), p.t_add));
        inv_s2 = pow2_inverse(s2);
    }
    issue(0, 0);
    issue(1, 1);
    const int erow = lane >> 3, ec4 = (lane & 7) * 4;
    int pix[4];                              // pixel of the rows erow + 8t this lane finishes
#pragma unroll
    for (int t = 0; t < 4; ++t) pix[t] = pixel_of(wave * 32 + erow + 8 * t);
    unsigned roff[4], yoff[4];               // 32-bit byte offsets (the entry point bounds the tensors): uniform base + offset addressing
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        roff[t] = ((unsigned)max(pix[t], 0) * (unsigned)p.res_ld + (unsigned)ec4) * 4u;
        yoff[t] = ((unsigned)max(pix[t], 0) * (unsigned)p.y_ld + (unsigned)ec4) * 4u;
    }
    const char *res_b = reinterpret_cast<const char *>(p.res);
    // round 6: y and the pooled tensor are stored through buffer resources -- a row beyond the tensor gets an out-of-range offset (the store
    // is dropped) instead of a branch around the store.  Straight-line code is what lets the compiler COUNT its memory operations: behind a
    // divergent branch every later wait became `s_waitcnt vmcnt(0)`, which on gfx950 also waits for the wave's own stores -- one store in
    // flight per wave, eight waves per CU: ~2 TB/s of writes for a launch that writes 189 MB.
    constexpr unsigned B2B_OOB = 0x80000000u;          // = num_records (the entry point keeps the tensors below 2 GB)
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, B2B_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(POOL ? p.pool : p.y), 0, B2B_OOB, 0x00020000);
    const bool no_stores = (p.skip & 4) != 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (pix[t] < 0 || no_stores) yoff[t] = B2B_OOB;
    floatx4 rv[TNH][4];
    auto load_res = [&](int col0, floatx4 (&dst)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)          // unconditional (rows beyond the tensor read pixel 0 and are never stored): the count below relies on it
            dst[t] = *reinterpret_cast<const floatx4 *>(res_b + (size_t)roff[t] + col0 * 4);
    };
    // the shortcut rows of the first column half are requested HERE, behind the first two chunks: they cross the main loop in flight
    // (HBM is nearly idle during it) instead of standing between the two multiplications and their stores
#pragma unroll
    for (int jj = 0; jj < TNH; ++jj) load_res(jj * 32, rv[jj]);
    // ================= conv A: 128 x KA over 9 * CA; three LDS stages, chunks requested two ahead, one barrier per chunk =================
    floatx16 acc[TNA];
#pragma unroll
    for (int j = 0; j < TNA; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    if (!(p.skip & 1))
    for (int k = 0; k < NCH; ++k) {
        if (k < 2 && (p.skip & 16))
            wait_vmcnt<0>();
        else if (k < 2)
            wait_vmcnt<G + 4 * TNH>();       // chunk k has landed; behind it, in order: (chunk 1,) the shortcut requests, chunk k + 1
        else if (k + 1 < NCH)
            wait_vmcnt<G>();                 // chunk k has landed (chunk k + 1 may be in flight)
        else
            wait_vmcnt<0>();
        // (the compiler moves the last fragment read's wait and its products BEHIND a bare s_barrier: without this wait another
        // wave's DMA into the stage just read overtakes the read -- seen as run-to-run differences of 3e-4)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();        // chunk k is visible, and every wave has finished chunk k - 1: its stage takes chunk k + 2
        if (k + 2 < NCH) issue((k + 2) % 3, k + 2);
        const int st = k % 3;
        const int cc = k / 9, tap = k - cc * 9, tr = tap / 3;
        const int wp = wp0 + tr * WX + (tap - tr * 3);          // this lane's window pixel under the tap
        const char *a_ptr = smem + cc * WIN_CC + wp * 128;
        const int wsw = (wp >> 1) & 7;
        const char *b_ptr = smem + BS_OFF + st * BST;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uintx4 a0 = *reinterpret_cast<const uintx4 *>(a_ptr + (((2 * s + fkh) ^ wsw) << 4));
            const uintx4 a1 = *reinterpret_cast<const uintx4 *>(a_ptr + (((4 + 2 * s + fkh) ^ wsw) << 4));
            uintx4 b[2][TNA];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int j = 0; j < TNA; ++j) b[pl][j] = *reinterpret_cast<const uintx4 *>(b_ptr + (pl * KA + j * 32) * 64 + b_foff[s]);
#pragma unroll
            for (int j = 0; j < TNA; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b[0][j]), acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TNA; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b[1][j]), acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TNA; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b[0][j]), acc[j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // every wave has read the last stage: the region becomes A2 / conv B's weights

    // ---- conv B's weights, one column half at a time: [chunk][plane][HB rows][64 B], rows swizzled like every B tile ----
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void *)p.wB, 0, 0xFFFFFF00u, 0x00020000);
    auto issue_w2 = [&](int half) {          // (offsets recomputed per call: eight registers less across the first multiplication)
        const int drow = lane >> 2, dslot = lane & 3;
        const unsigned plane_bytes = (unsigned)KB * KA * 2;
#pragma unroll
        for (int j = 0; j < B2_PASS; ++j) {
            const int R = (j * NW + wave) * 16 + drow;               // row of the [KCH * 2 * HB]-row buffer
            const int blk = R / HB, r = R - blk * HB;                 // blk = chunk * 2 + plane
            const int scol = dslot ^ ((R >> 2) & 3);
            const unsigned off = (unsigned)(blk & 1) * plane_bytes + (unsigned)((blk >> 1) * KB * 64 + (half * HB + r) * 64 + scol * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, (lds_ptr)(smem + W_OFF + (j * NW + wave) * 1024), 16, off, 0, 0, 0);
        }
    };
    if (p.skip & 1) wait_vmcnt<0>();
    if (p.skip & 2) return;
    issue_w2(0);             // (the stages are dead: the last loop iteration ended with a barrier, and its region holds no stage)

    // ================= the intermediate: BatchNorm + ReLU, scaled, split, into LDS as GEMM B's A operand =================
    float *sE = reinterpret_cast<float *>(smem + E_OFF) + wave * 1024;       // [32][32], column c of row r at c ^ ((r & 7) << 2)
    {
        float rs1[4], rs2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            rs1[t] = __shfl(inv_sa, (lane >> 3) + 8 * t);
            rs2[t] = __shfl(s2, (lane >> 3) + 8 * t);
        }
#pragma unroll
        for (int j = 0; j < TNA; ++j) {
            const int col = j * 32 + ec4;
            const floatx4 sc = *reinterpret_cast<const floatx4 *>(p.scaleA + col), sh = *reinterpret_cast<const floatx4 *>(p.shiftA + col);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                sE[row * 32 + ((lane & 31) ^ ((row & 7) << 2))] = acc[j][e];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rr = erow + 8 * t;
                const int r = wave * 32 + rr;                        // row of the tile
                floatx4 v = *reinterpret_cast<const floatx4 *>(sE + rr * 32 + (ec4 ^ ((rr & 7) << 2)));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float o = fmaf(v[u] * rs1[t], sc[u], sh[u]);
                    v[u] = o > 0.f ? o : 0.f;
                }
                typedef __attribute__((ext_vector_type(2))) unsigned uintx2_;
                const float s = rs2[t];
                const unsigned h0 = cvt_pk_f16(v[0] * s, v[1] * s), h1 = cvt_pk_f16(v[2] * s, v[3] * s);
                const unsigned l0 = cvt_pk_f16(fmaf(v[0], s, -f16_lo(h0)), fmaf(v[1], s, -f16_hi(h0)));
                const unsigned l1 = cvt_pk_f16(fmaf(v[2], s, -f16_lo(h1)), fmaf(v[3], s, -f16_hi(h1)));
                char *o2 = smem + j * (BM * 128) + r * 128 + (ec4 & 7) * 2;
                const int sw = (r >> 1) & 7;
                *reinterpret_cast<uintx2_ *>(o2 + (((ec4 >> 3) ^ sw) << 4)) = uintx2_{h0, h1};
                *reinterpret_cast<uintx2_ *>(o2 + (((4 + (ec4 >> 3)) ^ sw) << 4)) = uintx2_{l0, l1};
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ================= conv B in two column halves; epilogue: BatchNorm + shortcut + ReLU, 16-byte loads / stores =================
    float rs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) rs[t] = __shfl(inv_s2, (lane >> 3) + 8 * t);
    const int n_lo = t_n, n_hi = t_n, bnd = (t_n + 1) * hw;          // (the tile lies inside image t_n: `amx_hi` stays 0)
    float amx = 0.f, amx_hi = 0.f;
    floatx4 sc_next = *reinterpret_cast<const floatx4 *>(p.scaleB + ec4), sh_next = *reinterpret_cast<const floatx4 *>(p.shiftB + ec4);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // this half's weights have landed: the first half's are the only loads in flight; behind the second half's DMA come the 16
        // shortcut loads of epilogue 0 (and its stores), in order, so at most 8 outstanding means the DMA is done
        if (half == 0)
            wait_vmcnt<0>();
        else if (p.skip & 8)
            wait_vmcnt<0>();
        else
            wait_vmcnt<8>();
        __builtin_amdgcn_s_barrier();            // (first half: every wave's part of A2 is written too)
        floatx16 acc2[TNH];
#pragma unroll
        for (int j = 0; j < TNH; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const char *a_ptr = smem + c * (BM * 128) + wave * 32 * 128;
            const char *b_ptr = smem + W_OFF + c * (2 * HB * 64);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uintx4 a0 = *reinterpret_cast<const uintx4 *>(a_ptr + a_foff[s][0]);
                const uintx4 a1 = *reinterpret_cast<const uintx4 *>(a_ptr + a_foff[s][1]);
#pragma unroll
                for (int j = 0; j < TNH; ++j) {
                    const uintx4 b0 = *reinterpret_cast<const uintx4 *>(b_ptr + (j * 32) * 64 + b_foff[s]);
                    const uintx4 b1 = *reinterpret_cast<const uintx4 *>(b_ptr + (HB + j * 32) * 64 + b_foff[s]);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b0), acc2[j], 0, 0, 0);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b1), acc2[j], 0, 0, 0);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), acc2[j], 0, 0, 0);
                }
            }
        }
        if (half == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();        // every wave has read the first half's weights: the second half may land
            issue_w2(1);
        }
#pragma unroll
        for (int jj = 0; jj < TNH; ++jj) {
            const int col = half * HB + jj * 32 + ec4;
            // this column tile's scale / shift were requested one tile ago, IN FRONT of that tile's stores (a wait for a load also waits
            // for everything older than it, never for what was issued behind it); the next tile's are requested here
            const floatx4 sc = sc_next, sh = sh_next;
            if (jj + 1 < TNH || half == 0) {
                const int coln = (jj + 1 < TNH ? half * HB + (jj + 1) * 32 : HB) + ec4;
                sc_next = *reinterpret_cast<const floatx4 *>(p.scaleB + coln);
                sh_next = *reinterpret_cast<const floatx4 *>(p.shiftB + coln);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                sE[row * 32 + ((lane & 31) ^ ((row & 7) << 2))] = acc2[jj][e];
            }
            __builtin_amdgcn_wave_barrier();
            floatx4 v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rr = erow + 8 * t;
                v[t] = *reinterpret_cast<const floatx4 *>(sE + rr * 32 + (ec4 ^ ((rr & 7) << 2)));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float o = fmaf(v[t][u] * rs[t], sc[u], sh[u]) + rv[jj][t][u];
                    v[t][u] = o > 0.f ? o : 0.f;
                }
                const float rmx = fmaxf(fmaxf(fabsf(v[t][0]), fabsf(v[t][1])), fmaxf(fabsf(v[t][2]), fabsf(v[t][3])));
                const bool live = pix[t] >= 0 && !no_stores;
                amx = fmaxf(amx, (live && pix[t] < bnd) ? rmx : 0.0f);
                amx_hi = fmaxf(amx_hi, (live && pix[t] >= bnd) ? rmx : 0.0f);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v[t]), ry, (int)(yoff[t] + (unsigned)((half * HB + jj * 32) * 4)), 0, 0);
            }
            if constexpr (POOL) {
                floatx4 r;
#pragma unroll
                for (int u = 0; u < 4; ++u) r[u] = (((v[0][u] + v[1][u]) + v[2][u]) + v[3][u]) * 0.25f;
                const int blk = (t_n * (p.H >> 1) + (y0 >> 1) + wave) * (p.W >> 1) + (x0 >> 1) + erow;      // pooled pixel of the lane's 2x2 block
                const unsigned poff = pix[0] >= 0 ? (unsigned)blk * (unsigned)(p.pool_ld * 4) + (unsigned)col * 4u : B2B_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, r), rp, (int)poff, 0, 0);
            }
            if (half == 0) load_res(HB + jj * 32, rv[jj]);           // the second half's shortcut rows take this tile's place
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (p.amax_out) amax_track2(amx, amx_hi, n_lo, n_hi, p.amax_out, blockIdx.x * 8 + wave);
#endif
}

}  // namespace

// conv A: 3x3 / stride 1 / pad 1, 64 -> 64 channels, pre-split input; conv B: 1x1, 64 -> 256, + shortcut; ReLU behind both.
// pool / pool_ld: optional AvgPool2d(2, 2) of y ([N, H/2, W/2, pool_ld]; H, W even).
extern "C" int ppy_conv3x3_conv1x1_f32(const float *x_split, int x_ld, const float *xscale, const float *amax_in, const void *wA_f16x2,
                                       const float *scaleA_f16x2, const float *shiftA, const void *wB_f16x2, const float *scaleB_f16x2,
                                       const float *shiftB, const float *residual, int res_ld, float *y, int y_ld, float *pool, int pool_ld,
                                       int N, int H, int W, int CA, int KA, int KB, float t_mul, float t_add, float *amax_out, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(x_split && xscale && amax_in && wA_f16x2 && scaleA_f16x2 && shiftA && wB_f16x2 && scaleB_f16x2 && shiftB && residual && y);
    PPY_CHECK_ARG(N > 0 && H > 0 && W > 0 && t_mul >= 0.f && t_add >= 0.f);
    if (!(CA == 64 && KA == 64 && KB == 256)) return PPY_ERR_UNSUPPORTED;
    PPY_CHECK_ARG(x_ld >= CA && x_ld % 32 == 0 && res_ld >= KB && res_ld % 4 == 0 && y_ld >= KB && y_ld % 4 == 0);
    auto al = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
    PPY_CHECK_ARG(al(x_split) && al(wA_f16x2) && al(wB_f16x2) && al(scaleA_f16x2) && al(shiftA) && al(scaleB_f16x2) && al(shiftB) && al(residual) && al(y));
    if (pool) PPY_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && pool_ld >= KB && pool_ld % 4 == 0 && al(pool));
    const long long M = (long long)N * H * W;
    if (M > 0x7fffffffLL / 4 || (M + W + 1) * x_ld * 4 >= 0xFFFFF000LL || M * res_ld * 4 >= 0xFFFFF000LL || M * y_ld * 4 >= 0x7FFFF000LL ||
        (pool && (M / 4) * pool_ld * 4 >= 0x7FFFF000LL)) return PPY_ERR_UNSUPPORTED;      // (y / pool: buffer stores with a 2 GB range)
    B2bArgs a;
    a.x = x_split; a.wA = (const unsigned short *)wA_f16x2; a.wB = (const unsigned short *)wB_f16x2;
    a.scaleA = scaleA_f16x2; a.shiftA = shiftA; a.scaleB = scaleB_f16x2; a.shiftB = shiftB;
    a.res = residual; a.y = y; a.pool = pool; a.xscale = xscale; a.amax_in = amax_in; a.amax_out = amax_out;
    a.t_mul = t_mul; a.t_add = t_add;
    { static const int sk = getenv("PPY_B2B_SKIP") ? atoi(getenv("PPY_B2B_SKIP")) : 0; a.skip = sk; }
    a.x_ld = x_ld; a.res_ld = res_ld; a.y_ld = y_ld; a.pool_ld = pool_ld; a.N = N; a.H = H; a.W = W; a.M = (int)M;
    constexpr size_t lds = 2 * 128 * 128 + 2 * 2 * 128 * 64 + 4 * 4096;       // A2 + one column half of conv B's planes + patches = 80 KB: two workgroups per CU
    const unsigned grid = (unsigned)(N * ((H + 7) / 8) * ((W + 15) / 16));          // 8 x 16-pixel tiles, each inside one image
    if (pool) {
        auto k = conv_b2b_kernel<64, 64, 256, true>;
        static PpyLdsAttr attr;
        if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), (int)lds) != PPY_OK) return PPY_ERR_LAUNCH;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
    } else {
        auto k = conv_b2b_kernel<64, 64, 256, false>;
        static PpyLdsAttr attr;
        if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), (int)lds) != PPY_OK) return PPY_ERR_LAUNCH;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
    }
    return ppy_launch_status();
}
