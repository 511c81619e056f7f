// 3x3 / stride 1 / pad 1 convolutions with C = 32 input channels at large maps: the two stem layers of ResNet-vd behind the
// strided first one (reference model/resnet_vd.py:108-110, conv1_2 C32 -> K32 and conv1_3 C32 -> K64 at 304x304 for a 608 input).
//
// On the implicit-GEMM tiles of conv_x3.hip these layers run at 2.5x their byte / MFMA floor: one 32-deep chunk per tap means
// nine operand tiles per output tile with nothing to amortise them over (K = 32 fills half of the narrowest tile), and the slab
// variants do not help (profiles/r02_stream_1x1_ab.txt, part 4).  Same operator, same f16x2 arithmetic, same order of the
// products as those tiles (results are bit-identical), organised around the INPUT PATCH instead:
//   * a persistent workgroup of eight waves per CU walks over 8 x 32-pixel output tiles; the (8+2) x (32+2)-pixel input patch
//     of a tile is requested one tile ahead (six 16-byte loads per thread, out-of-image pixels = out-of-range offsets = zeros),
//     scaled and split into its two fp16 terms ONCE (a tile lies inside one image, so its scale is uniform) and written to
//     two LDS planes -- the nine taps read their A fragments from there as shifted windows: 1.33 input bytes per output pixel
//     and channel from the L2 instead of 9, and no split in the inner loop;
//   * the weights of all nine taps stay in the LDS (36 / 72 KB for K = 32 / 64) for the life of the workgroup;
//   * wave w owns output row w of the tile (32 consecutive pixels = one MFMA row tile, K / 32 column tiles): 54 MFMAs per
//     column tile with two 16-byte LDS reads per operand pair, then the vector epilogue of conv_shared.h's layout (rows of
//     4 KB / 8 KB contiguous output per wave).
//
// MPOOL (round 4): the last stem layer with the MaxPool2d(3, 2, 1) behind it (reference model/resnet_vd.py:103, 136) in ONE launch:
// the full-resolution tensor (189 MB at 8 x 304 x 304 x 64) is never written or read back.  A tile then is the window of 3 x 15
// POOLED pixels: conv rows 6 by - 1 .. 6 by + 5 (waves 0..6; wave 7 only stages) and columns 30 bx - 1 .. 30 bx + 30 -- neighbouring
// tiles recompute one row / two columns (1.24 x the MFMA work).  Every wave finishes its row as before (scale, shift, activation),
// pools it along x through its transposition patch, leaves the 15 x K result in the (now idle) operand planes, and after one
// barrier the workgroup takes the maximum over the three rows of each pooled row and stores 3 x 15 x K values.  Same products in the
// same order per pixel and max() is exact: bit-identical to the two launches.
#include "conv_shared.h"
#pragma clang fp contract(off)

#ifndef PPY_PATCH_ABL
#define PPY_PATCH_ABL 0   // ablation switch (tools/experiments/patch_ablate.sh; results are garbage, only the timing means something): 1 = no MFMA
                          // phase, 2 = no split / plane writes, 3 = no epilogue, 4 = no patch requests, 5 = MFMAs without LDS reads
#endif

namespace {

constexpr unsigned PT_OOB = 0x80000000u;      // beyond any tensor this kernel accepts (< 2 GB): loads give 0, stores are dropped
constexpr int PT_TH = 8, PT_TW = 32, PT_PW = PT_TW + 2, PT_NPIX = (PT_TH + 2) * PT_PW;      // tile, patch
constexpr int PT_PLANE = PT_NPIX * 64;                                                       // one fp16 plane of the patch: 64 B per pixel
constexpr int PT_UNITS = PT_NPIX * 8, PT_NST = (PT_UNITS + 511) / 512;                        // 16-byte staging loads, per thread

constexpr int PM_ROWS = 3, PM_COLS = 15;      // MPOOL: pooled pixels per tile

struct PatchArgs {
    ConvArgs c;
    int tiles_x, tiles_y, ntiles;
    int Hp, Wp;       // MPOOL: pooled map (c.y / c.y_ld then describe the POOLED tensor [N][Hp][Wp][y_ld])
};

template <int TN, bool BNS = false, bool MPOOL = false>      // (BNS: BatchNorm statistics from the epilogue, conv_x3.hip)
__global__ void __launch_bounds__(512, 1) conv3x3_patch_kernel(const PatchArgs q) {
    static_assert(!(BNS && MPOOL), "statistics of a tensor that is not stored");
    static_assert(PT_TH * PM_COLS * 32 * TN * 4 <= 2 * PT_PLANE, "the x-pooled rows fit the operand planes");
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvArgs &p = q.c;
    constexpr int K = 32 * TN, WBYTES = 9 * 2 * K * 64;
    extern __shared__ __attribute__((aligned(16))) char smem_pt[];
    char *pl_hi = smem_pt, *pl_lo = smem_pt + PT_PLANE, *wl = smem_pt + 2 * PT_PLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *sE = reinterpret_cast<float *>(smem_pt + 2 * PT_PLANE + WBYTES) + wave * (32 * LDS_LD);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, PT_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, PT_OOB, 0x00020000);

    // ---- weights -> LDS: row (tap, plane, k) = 32 fp16 = 64 B, 16-byte slot c stored at c ^ ((k>>2)&3) (conv_x3.hip's B tiles);
    // source planes [plane][chunk = tap][K][32] (ppy_conv2d_split_weights_f16x2)
    {
        const char *wb = reinterpret_cast<const char *>(p.wf16);
        const long long plane_bytes = (long long)K * 9 * 32 * 2;
        // (round 6: all of a thread's pieces are requested before the first is written -- as a plain loop hipcc waited for every load
        // before it issued the next one: nine memory round trips at the head of every workgroup)
        constexpr int WUNITS = 9 * 2 * K * 4, NWL = (WUNITS + 511) / 512;
        uintx4 wv[NWL];
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int u = min(tid + 512 * i, WUNITS - 1);
            const int slot = u & 3, row = u >> 2;
            const int k = row % K, tp = row / K, plane = tp & 1, tap = tp >> 1;
            wv[i] = *reinterpret_cast<const uintx4 *>(wb + plane * plane_bytes + ((long long)(tap * K + k) * 32) * 2 + slot * 16);
        }
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int u = tid + 512 * i;
            const int slot = u & 3, row = u >> 2;
            const int k = row % K;
            if (u < WUNITS) *reinterpret_cast<uintx4 *>(wl + row * 64 + ((slot ^ ((k >> 2) & 3)) << 4)) = wv[i];
        }
    }
    const int erow = lane >> 3, ec4 = (lane & 7) * 4;
    floatx4 sc[TN], sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        sc[j] = *reinterpret_cast<const floatx4 *>(p.scale + j * 32 + ec4);
        sh[j] = *reinterpret_cast<const floatx4 *>(p.shift + j * 32 + ec4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(sc[j]), "+v"(sh[j]));      // (nothing pending at the loop head: conv_stream.hip)

    // ---- staging: unit u = (patch pixel u >> 3, 16-byte group u & 7 of its 32 fp32 channels), units tid + 512 i
    int s_py[PT_NST], s_px[PT_NST];
#pragma unroll
    for (int i = 0; i < PT_NST; ++i) {
        const int pp = min((tid + 512 * i) >> 3, PT_NPIX - 1);
        s_py[i] = pp / PT_PW;
        s_px[i] = pp - s_py[i] * PT_PW;
    }
    uintx4 stg_a[PT_NST], stg_b[PT_NST];      // two sets: a patch is requested TWO tiles ahead (one tile period is ~2.5 us: not
                                              // enough for a round trip to HBM under load with one workgroup per CU)
#pragma unroll
    for (int i = 0; i < PT_NST; ++i) stg_a[i] = stg_b[i] = uintx4{0u, 0u, 0u, 0u};
    const int tiles_img = q.tiles_x * q.tiles_y;
    auto tile_of = [&](int t, int &n, int &y0, int &x0) {
        n = t / tiles_img;
        const int r = t - n * tiles_img, by = r / q.tiles_x;
        y0 = MPOOL ? 2 * PM_ROWS * by - 1 : by * PT_TH;
        x0 = MPOOL ? 2 * PM_COLS * (r - by * q.tiles_x) - 1 : (r - by * q.tiles_x) * PT_TW;
    };
    auto request = [&](int t, uintx4 (&stg)[PT_NST]) {
        int n, y0, x0;
        tile_of(max(t, 0), n, y0, x0);
        const bool live = t >= 0 && t < q.ntiles;
#pragma unroll
        for (int i = 0; i < PT_NST; ++i) {
            const int y = y0 - 1 + s_py[i], x = x0 - 1 + s_px[i];
            const bool ok = live && tid + 512 * i < PT_UNITS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            const unsigned off = ok ? (unsigned)((n * p.H + y) * p.W + x) * (unsigned)(p.x_ld * 4) + (unsigned)((tid + 512 * i) & 7) * 16u : PT_OOB;
            if (PPY_PATCH_ABL != 4) stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0);
        }
    };

    int sc_n = -1;
    float sa = 1.f, inv_sa = 1.f;
    int run_n = -1;
    float run_mx = 0.f;
    auto flush = [&](float mx, int n) {
        if (p.amax_out && n >= 0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (lane == 0) amax_store(mx, p.amax_out, n, (int)blockIdx.x * 8 + wave);
        }
    };
    const float slope = p.act == PPY_ACT_RELU ? 0.f : (p.act == PPY_ACT_LEAKY ? 0.1f : 1.f);
    __syncthreads();      // the weights are in place
    // ONE code path for requests and waits (conv_stream.hip): the loop starts two strides in front of the workgroup's first
    // tile with iterations that only request; the barriers are executed by every wave in every iteration
    const int stride = (int)gridDim.x;
    auto do_tile = [&](int t, uintx4 (&stg)[PT_NST]) {
        const bool live = t >= 0 && t < q.ntiles;
        int n, y0, x0;
        tile_of(live ? t : 0, n, y0, x0);
        if (live) {
        if (n != sc_n) {      // per-image activation scale (conv_x3.hip): the power of two that puts the tracked maximum into [2^13, 2^14)
            const float mx = conv_amax_in(p, n);
            const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
            int f = 267 - e;
            f = f < 103 ? 103 : (f > 167 ? 167 : f);
            sa = __uint_as_float((unsigned)f << 23);
            inv_sa = __uint_as_float((unsigned)(254 - f) << 23);
            sc_n = n;
        }
        if (n != run_n) {
            flush(run_mx, run_n);
            run_n = n;
            run_mx = 0.f;
        }
        // ---- the patch requested one iteration ago: scale, split into two fp16 terms, two LDS planes (64 B per pixel and plane,
        // 16-byte slot c = channels 8c .. 8c+7 stored at c ^ ((pixel>>2)&3): the 16 pixels of a quarter-wave hit 16 bank groups)
#pragma unroll
        for (int i = 0; i < PT_NST; ++i) {
            const int u = tid + 512 * i;
            if (u < PT_UNITS && PPY_PATCH_ABL != 2) {
                const int pp = u >> 3, g = u & 7;
                const float x0f = __uint_as_float(stg[i][0]), x1f = __uint_as_float(stg[i][1]);
                const float x2f = __uint_as_float(stg[i][2]), x3f = __uint_as_float(stg[i][3]);
                const unsigned h0 = cvt_pk_f16(x0f * sa, x1f * sa), h1 = cvt_pk_f16(x2f * sa, x3f * sa);
                const unsigned l0 = cvt_pk_f16(fmaf(x0f, sa, -f16_lo(h0)), fmaf(x1f, sa, -f16_hi(h0)));
                const unsigned l1 = cvt_pk_f16(fmaf(x2f, sa, -f16_lo(h1)), fmaf(x3f, sa, -f16_hi(h1)));
                const int o = pp * 64 + (((g >> 1) ^ ((pp >> 2) & 3)) << 4) + (g & 1) * 8;
                typedef __attribute__((ext_vector_type(2))) unsigned uintx2;
                *reinterpret_cast<uintx2 *>(pl_hi + o) = uintx2{h0, h1};
                *reinterpret_cast<uintx2 *>(pl_lo + o) = uintx2{l0, l1};
            }
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        request(t + 2 * stride, stg);

        // ---- nine taps x two 16-deep steps: A fragments = the planes at the tap's shift, B fragments from the resident weights
        floatx16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        const int tx = lane & 31, kh = lane >> 5;
        if (live && PPY_PATCH_ABL != 1 && (!MPOOL || wave < 2 * PM_ROWS + 1)) {
            // 18 steps (tap, 16-deep half), software-pipelined by hand: the LDS reads of step g+1 are issued in front of the
            // MFMAs of step g and the order is fenced -- left alone, hipcc puts every step's reads directly in front of its
            // MFMAs behind an lgkmcnt(0), and the (dependent: one accumulator per column tile) MFMAs wait out the LDS latency
            // 18 times per tile (measured: 65 us for C32 -> K32 at 304x304 either way it fetched its operands).
            struct Frag {
                uintx4 a0, a1, b0[TN], b1[TN];
            };
            Frag f[2];
            auto fetch = [&](Frag &fr, int g) {
                const int tap = g >> 1, s2 = g & 1;
                const int pp = (wave + tap / 3) * PT_PW + tx + tap % 3;
                const int ao = pp * 64 + (((2 * s2 + kh) ^ ((pp >> 2) & 3)) << 4);
                fr.a0 = *reinterpret_cast<const uintx4 *>(pl_hi + ao);
                fr.a1 = *reinterpret_cast<const uintx4 *>(pl_lo + ao);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int k = j * 32 + tx;
                    const int bo = (((2 * s2 + kh) ^ ((k >> 2) & 3)) << 4);
                    fr.b0[j] = *reinterpret_cast<const uintx4 *>(wl + ((tap * 2 + 0) * K + k) * 64 + bo);
                    fr.b1[j] = *reinterpret_cast<const uintx4 *>(wl + ((tap * 2 + 1) * K + k) * 64 + bo);
                }
            };
            fetch(f[0], 0);
#pragma unroll
            for (int g = 0; g < 18; ++g) {
                if (g + 1 < 18 && PPY_PATCH_ABL != 5) fetch(f[(g + 1) & 1], g + 1);
                __builtin_amdgcn_sched_barrier(0);
                const Frag &c = f[PPY_PATCH_ABL == 5 ? 0 : g & 1];
                // the three leading products, smallest first, as conv_x3.hip orders them: a1*b0, a0*b1, a0*b0
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, c.a1), __builtin_bit_cast(f16x8, c.b0[j]), acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, c.a0), __builtin_bit_cast(f16x8, c.b1[j]), acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, c.a0), __builtin_bit_cast(f16x8, c.b0[j]), acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // every wave is done with the planes before anyone overwrites them (next iteration); the epilogue is wave-private
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!live || PPY_PATCH_ABL == 3) return;

        const int y = y0 + wave;
        const bool row_ok = (unsigned)y < (unsigned)p.H;
        if constexpr (MPOOL) {
            float *xp = reinterpret_cast<float *>(smem_pt);            // [wave][pooled column][K]: the operand planes are idle now
            constexpr int K4 = K / 4;
            if (wave < 2 * PM_ROWS + 1) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        sE[row * LDS_LD + (lane & 31)] = acc[j][e];
                    }
                    __builtin_amdgcn_wave_barrier();
                    // finished values back into the patch; positions outside the image never win a maximum (MaxPool2d pads with -inf)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        floatx4 v = *reinterpret_cast<const floatx4 *>(sE + (erow + 8 * u) * LDS_LD + ec4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float o = fmaf(v[c] * inv_sa, sc[j][c], sh[j][c]);
                            v[c] = o > 0.f ? o : o * slope + 0.0f;
                        }
                        const bool ok = row_ok && (unsigned)(x0 + erow + 8 * u) < (unsigned)p.W;
                        const float rmx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                        run_mx = fmaxf(run_mx, ok ? rmx : 0.f);
                        if (!ok) v = floatx4{-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
                        *reinterpret_cast<floatx4 *>(sE + (erow + 8 * u) * LDS_LD + ec4) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int jx = erow + 8 * h;              // pooled column: tile columns 2 jx .. 2 jx + 2
                        if (jx < PM_COLS) {
                            const floatx4 a = *reinterpret_cast<const floatx4 *>(sE + (2 * jx) * LDS_LD + ec4);
                            const floatx4 b = *reinterpret_cast<const floatx4 *>(sE + (2 * jx + 1) * LDS_LD + ec4);
                            const floatx4 c = *reinterpret_cast<const floatx4 *>(sE + (2 * jx + 2) * LDS_LD + ec4);
                            floatx4 m;
#pragma unroll
                            for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(a[k], b[k]), c[k]);
                            *reinterpret_cast<floatx4 *>(xp + (wave * PM_COLS + jx) * K + j * 32 + ec4) = m;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int by = (y0 + 1) / (2 * PM_ROWS), bx = (x0 + 1) / (2 * PM_COLS);
#pragma unroll
            for (int it = 0; it < (PM_ROWS * PM_COLS * K4 + 511) / 512; ++it) {
                const int i = tid + 512 * it;
                const int r = i / (PM_COLS * K4), rem = i - r * (PM_COLS * K4);
                const int jx = rem / K4, c4 = rem - jx * K4;
                const int py = PM_ROWS * by + r, px = PM_COLS * bx + jx;
                const bool ok = i < PM_ROWS * PM_COLS * K4 && py < q.Hp && px < q.Wp;
                const int rr = ok ? r : 0, jj = ok ? jx : 0;
                const floatx4 a = *reinterpret_cast<const floatx4 *>(xp + ((2 * rr) * PM_COLS + jj) * K + c4 * 4);
                const floatx4 b = *reinterpret_cast<const floatx4 *>(xp + ((2 * rr + 1) * PM_COLS + jj) * K + c4 * 4);
                const floatx4 c = *reinterpret_cast<const floatx4 *>(xp + ((2 * rr + 2) * PM_COLS + jj) * K + c4 * 4);
                floatx4 m;
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(a[k], b[k]), c[k]);
                const unsigned off = ok ? (unsigned)((n * q.Hp + py) * q.Wp + px) * (unsigned)(p.y_ld * 4) + (unsigned)c4 * 16u : PT_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, m), ry, (int)off, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // nobody rewrites the planes (next tile) before everybody has read the pooled rows
            return;
        }
        if constexpr (BNS) {      // training forward: the BatchNorm's first pass from here (conv_shared.h), one slice per (tile, output row)
            const int nv = row_ok ? min(max(p.W - x0, 0), 32) : 0;
            const float inv_f[1] = {inv_sa};
            tile_bn_stats<1, TN, 32, 32 * TN>(p, reinterpret_cast<const floatx16(&)[1][TN]>(acc), inv_f, p.M - nv, 0, 0, 0, lane, t * PT_TH + wave);
        }
        const unsigned rowbase = (unsigned)((n * p.H + y) * p.W + x0) * (unsigned)(p.y_ld * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                sE[row * LDS_LD + (lane & 31)] = acc[j][e];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                floatx4 v = *reinterpret_cast<const floatx4 *>(sE + (erow + 8 * u) * LDS_LD + ec4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float o = fmaf(v[c] * inv_sa, sc[j][c], sh[j][c]);
                    v[c] = o > 0.f ? o : o * slope + 0.0f;
                }
                const bool ok = row_ok && x0 + erow + 8 * u < p.W;
                const float rmx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                run_mx = fmaxf(run_mx, ok ? rmx : 0.f);
                const unsigned off = ok ? rowbase + (unsigned)(erow + 8 * u) * (unsigned)(p.y_ld * 4) + (unsigned)(j * 32 + ec4) * 4u : PT_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), ry, (int)off, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    for (int t = (int)blockIdx.x - 2 * stride; t < q.ntiles; t += 2 * stride) {
        do_tile(t, stg_a);
        do_tile(t + stride, stg_b);
    }
    flush(run_mx, run_n);
#endif
}

template <int TN, bool BNS = false, bool MPOOL = false>
int launch_patch(const PatchArgs &q, hipStream_t stream) {
    if constexpr (!BNS && !MPOOL) {
        if (q.c.bn_part) return launch_patch<TN, true>(q, stream);
    }
    auto k = conv3x3_patch_kernel<TN, BNS, MPOOL>;
    const size_t lds = (size_t)2 * PT_PLANE + (size_t)9 * 2 * 32 * TN * 64 + (size_t)8 * 32 * LDS_LD * sizeof(float);
    static PpyLdsAttr attr;
    if (ppy_lds_attr(attr, reinterpret_cast<const void *>(k), (int)lds) != PPY_OK) return PPY_ERR_LAUNCH;
    const int grid = q.ntiles < 256 ? q.ntiles : 256;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, stream, q);
    return ppy_launch_status();
}

}  // namespace

int ppy_patch_num_configs() { return 1; }

int ppy_patch_dispatch(const ConvArgs &p, int local, hipStream_t stream) {
    if (local != 0) return PPY_ERR_BAD_ARG;
    // BAD_ARG, not UNSUPPORTED: an explicit id that does not apply is the caller's error (no silent other kernel)
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || p.C != 32 || (p.K != 32 && p.K != 64) || p.ups || p.posb || p.res)
        return PPY_ERR_BAD_ARG;
    if (!p.wf16 || ((uintptr_t)p.wf16 & 15) != 0 || !p.scale_f16 || !p.amax_in) return PPY_ERR_BAD_ARG;
    if (!vec_epilogue_ok(p) || ((uintptr_t)p.x & 15) != 0 || p.x_ld % 4 != 0) return PPY_ERR_BAD_ARG;
    const long long lim = 0x7FFFF000LL;
    if ((long long)p.M * p.x_ld * 4 >= lim || (long long)p.M * p.y_ld * 4 >= lim) return PPY_ERR_UNSUPPORTED;
    if (p.bn_part) {
        if (p.act != PPY_ACT_NONE) return PPY_ERR_UNSUPPORTED;
        if (p.N * ceil_div(p.W, PT_TW) * ceil_div(p.H, PT_TH) * PT_TH > p.bn_capacity) return PPY_ERR_WORKSPACE;
        if (p.bn_slices_host) *p.bn_slices_host = p.N * ceil_div(p.W, PT_TW) * ceil_div(p.H, PT_TH) * PT_TH;
    }
    PatchArgs q;
    q.c = p;
    q.c.scale = p.scale_f16;
    q.Hp = q.Wp = 0;
    q.tiles_x = ceil_div(p.W, PT_TW);
    q.tiles_y = ceil_div(p.H, PT_TH);
    q.ntiles = p.N * q.tiles_x * q.tiles_y;
    return p.K == 32 ? launch_patch<1>(q, stream) : launch_patch<2>(q, stream);
}

// conv3x3 (C = 32 -> K = 64, stride 1, pad 1) + affine + activation + MaxPool2d(3, 2, 1) in one launch (MPOOL above)
int ppy_patch_maxpool_dispatch(const ConvArgs &p, int Hp, int Wp, hipStream_t stream) {
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || p.C != 32 || p.K != 64 || p.ups || p.posb || p.res || p.bn_part || p.xscale || p.yscale)
        return PPY_ERR_BAD_ARG;
    if (!p.wf16 || ((uintptr_t)p.wf16 & 15) != 0 || !p.scale_f16 || !p.amax_in) return PPY_ERR_BAD_ARG;
    if (((uintptr_t)p.x & 15) != 0 || p.x_ld % 4 != 0 || ((uintptr_t)p.y & 15) != 0 || p.y_ld % 4 != 0) return PPY_ERR_BAD_ARG;
    if (Hp != (p.H - 1) / 2 + 1 || Wp != (p.W - 1) / 2 + 1) return PPY_ERR_BAD_ARG;
    const long long lim = 0x7FFFF000LL;
    if ((long long)p.M * p.x_ld * 4 >= lim || (long long)p.N * Hp * Wp * p.y_ld * 4 >= lim) return PPY_ERR_UNSUPPORTED;
    PatchArgs q;
    q.c = p;
    q.c.scale = p.scale_f16;
    q.Hp = Hp;
    q.Wp = Wp;
    q.tiles_x = ceil_div(Wp, PM_COLS);
    q.tiles_y = ceil_div(Hp, PM_ROWS);
    q.ntiles = p.N * q.tiles_x * q.tiles_y;
    return launch_patch<2, false, true>(q, stream);
}
