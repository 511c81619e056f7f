// Image pre-processing in front of the hot path, on the device: the reference's Decode.process_image
// (model/decode_np.py:125-140) = BGR->RGB, cv2.resize(..., fx=S/w, fy=S/h, INTER_CUBIC) on uint8
// (tools/transform.py:996-1003), /255, -mean, /std in numpy (transform.py:895-917), HWC->CHW (transform.py:1052-1054)
// -- one kernel, uint8 HWC in, float32 NCHW out (the layout the stem kernel reads).
//
// Resize arithmetic = OpenCV 4.x's 8-bit bicubic (imgproc/src/resize.cpp), integer for integer:
//   fx = (float)((dx + 0.5) * scale_x - 0.5), sx = floor(fx), fx -= sx        (scale_x = 1 / (S / w), in double)
//   interpolateCubic(fx) with A = -0.75 in float32, one rounding per operation (fp contraction off below)
//   weights = cvRound(coef * 2048) as int16; taps sx-1 .. sx+2 with indices clamped to the image (replicated border)
//   horizontal sums in int32, vertical sum in int32, FixedPtCast: (v + 2^21) >> 22, saturated to uint8.
// (OpenCV's vectorised vertical pass evaluates the last sum in float and may differ by one grey level on a few pixels
// in 10^4; cv2 is not available to this build, see oracle/preprocess_oracle.py.)
// The numpy normalisation is a function of (grey level, channel): the caller passes it as a 3 x 256 float table built
// with the reference's own numpy expression, so that part is bit-exact by construction.
// Roofline: HBM-bound by design (source bytes read once through L2, S*S*3 floats written once); at 608x608 the launch
// is ~10 us of latency.
#include "common.h"

namespace {

constexpr int PRE_MAX_IMAGES = 16;
typedef unsigned u32_unaligned __attribute__((aligned(1)));
struct PreImage {
    const unsigned char *src;
    int h, w, stride;
    double scale_x, scale_y;
};
struct PreArgs {
    PreImage img[PRE_MAX_IMAGES];
    const float *lut;       // [3][256]
    float *out;             // [n][3][S][S]
    int S, swap_rb;
};

#pragma clang fp contract(off)
__device__ __forceinline__ void cubic_weights(float x, int (&wgt)[4]) {
    const float A = -0.75f;
    const float t = x + 1.0f;
    float c[4];
    c[0] = ((A * t - 5.0f * A) * t + 8.0f * A) * t - 4.0f * A;
    c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    const float u = 1.0f - x;
    c[2] = ((A + 2.0f) * u - (A + 3.0f)) * u * u + 1.0f;
    c[3] = 1.0f - c[0] - c[1] - c[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float v = rintf(c[k] * 2048.0f);                     // cvRound: ties to even
        wgt[k] = (int)fminf(fmaxf(v, -32768.0f), 32767.0f);        // saturate_cast<short>
    }
}

__device__ __forceinline__ int first_tap(int d, double scale, float &frac) {
    const float f = (float)(((double)d + 0.5) * scale - 0.5);
    const float fl = floorf(f);
    frac = f - fl;
    return (int)fl;
}

__global__ __launch_bounds__(256) void preprocess_kernel(const PreArgs p) {
    const PreImage &im = p.img[blockIdx.z];
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= p.S || dy >= p.S) return;
    float fx, fy;
    const int sx = first_tap(dx, im.scale_x, fx), sy = first_tap(dy, im.scale_y, fy);
    int ax[4], ay[4];
    cubic_weights(fx, ax);
    cubic_weights(fy, ay);
    const unsigned char *row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) row[k] = im.src + (long long)min(max(sy - 1 + k, 0), im.h - 1) * im.stride;
    // the 4 x 3 bytes of a row's taps: three unaligned dword loads when all four taps are inside the row (one thread
    // then issues 12 loads instead of 48), byte loads with clamped columns at the left / right border
    unsigned px[4][3];      // [row][dword]: bytes 0..11 = taps 0..3 x (B, G, R)
    if (sx >= 1 && sx + 2 <= im.w - 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32_unaligned *q = reinterpret_cast<const u32_unaligned *>(row[k] + (sx - 1) * 3);
            px[k][0] = q[0];
            px[k][1] = q[1];
            px[k][2] = q[2];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned char b[12];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cj = min(max(sx - 1 + j, 0), im.w - 1) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) b[3 * j + c] = row[k][cj + c];
            }
#pragma unroll
            for (int d = 0; d < 3; ++d)
                px[k][d] = (unsigned)b[4 * d] | ((unsigned)b[4 * d + 1] << 8) | ((unsigned)b[4 * d + 2] << 16) |
                           ((unsigned)b[4 * d + 3] << 24);
        }
    }
    const long long plane = (long long)p.S * p.S;
    float *o = p.out + (long long)blockIdx.z * 3 * plane + (long long)dy * p.S + dx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int cs = p.swap_rb ? 2 - c : c;
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int hs = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int byte = 3 * j + cs;                       // 0..11 (cs is wave-uniform)
                hs += (int)((px[k][byte >> 2] >> (8 * (byte & 3))) & 0xffu) * ax[j];
            }
            acc += hs * ay[k];
        }
        int v = (acc + (1 << 21)) >> 22;
        v = min(max(v, 0), 255);
        o[c * plane] = p.lut[c * 256 + v];
    }
}

// Round 5: the same arithmetic, separable and tiled.  The pixel kernel above evaluates 48 integer taps, two double-precision
// coordinate maps and two sets of cubic weights PER OUTPUT PIXEL (issue-bound: 56 us for 8 images 480 x 640 -> 608 x 608, 0.095 of
// HBM).  Here a workgroup owns PRE_TW output columns x R output rows of one image (R per image, chosen by the host so that the
// source rows of a tile fit the LDS):
//   phase 1  every source row the tile touches is filtered HORIZONTALLY once for the tile's columns (int32 sums, exactly the
//            pixel kernel's `hs`) into LDS -- a thread owns one column: coordinate map and weights once, 3 unaligned dword loads
//            and 12 multiply-adds per row;
//   phase 2  a thread owns 4 consecutive columns of every 8th row: coordinate map and weights once per row, the four vertical taps
//            as 16-byte LDS reads, FixedPtCast, the normalisation table from LDS, one 16-byte store per channel.
// Integer arithmetic is exact and the float expressions are the pixel kernel's own functions, so the result is bit-identical
// (tests/test_preprocess.py compares both kernels with the numpy oracle).
constexpr int PRE_TW = 128;      // output columns of a tile; rows: template parameters (MAXROWS source rows in LDS, MAXR output rows, U rows of loads in flight)
struct PreTileArgs {
    PreArgs a;
    int rows_per_tile[PRE_MAX_IMAGES];
};

template <int PRE_MAXROWS, int PRE_U>
__global__ __launch_bounds__(256) void preprocess_tile_kernel(const PreTileArgs p) {
    __shared__ int s_h[PRE_MAXROWS][3][PRE_TW];
    __shared__ float s_lut[3 * 256];
    const PreImage &im = p.a.img[blockIdx.z];
    const int S = p.a.S, R = p.rows_per_tile[blockIdx.z];
    const int dy0 = blockIdx.y * R;
    if (dy0 >= S) return;
    const int dy1 = min(dy0 + R, S) - 1;                      // last output row of the tile
    const int dx0 = blockIdx.x * PRE_TW;
    {   // (the three table rows requested together)
        const float l0 = p.a.lut[threadIdx.x], l1 = p.a.lut[256 + threadIdx.x], l2 = p.a.lut[512 + threadIdx.x];
        s_lut[threadIdx.x] = l0;
        s_lut[256 + threadIdx.x] = l1;
        s_lut[512 + threadIdx.x] = l2;
    }
    float ftmp;
    const int r_lo = first_tap(dy0, im.scale_y, ftmp) - 1;
    const int nrows = first_tap(dy1, im.scale_y, ftmp) + 2 - r_lo + 1;      // <= PRE_MAXROWS (host: rows_for)
    {   // ---- phase 1: horizontal pass
        const int col = threadIdx.x & (PRE_TW - 1), lane = threadIdx.x >> 7;      // 128 columns x 2 row lanes
        const int dx = dx0 + col;
        if (dx < S) {
            float fx;
            const int sx = first_tap(dx, im.scale_x, fx);
            int ax[4];
            cubic_weights(fx, ax);
            const bool inside = sx >= 1 && sx + 2 <= im.w - 1;
            int cj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cj[j] = min(max(sx - 1 + j, 0), im.w - 1) * 3;
            // PRE_U rows per pass, all their loads in flight together (one row at a time left every thread in a chain of dependent
            // ~1 us loads: the phase was latency-bound).  Round 6: the four source pixels of a row (12 bytes at an arbitrary byte address) are
            // ONE aligned 16-byte buffer load and three v_alignbyte -- as three unaligned dwords hipcc emitted nine 1- / 2-byte loads per row, and
            // because they sat inside the `inside` branch it could not count them: every row's loads were waited for before the next row's
            // were issued (ISA of the round-5 kernel), not PRE_U rows in flight.  The resource starts at the image pointer rounded DOWN to a
            // dword; a thread at the image border requests a clamped (valid) address here and gathers its bytes one by one below.
            const unsigned mis = (unsigned)(reinterpret_cast<uintptr_t>(im.src) & 3);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(im.src - mis), 0, (mis + (unsigned)im.h * (unsigned)im.stride + 3u) & ~3u, 0x00020000);
            const unsigned bo = mis + (unsigned)min(max(sx - 1, 0), max(im.w - 4, 0)) * 3u;
            for (int rb = lane; rb < nrows; rb += 2 * PRE_U) {
                unsigned px[PRE_U][3];
                uintx4 dw[PRE_U];
#pragma unroll
                for (int u = 0; u < PRE_U; ++u) {
                    const int r = min(rb + 2 * u, nrows - 1);       // (a repeated last row: same bytes, result not stored)
                    const unsigned a = (unsigned)min(max(r_lo + r, 0), im.h - 1) * (unsigned)im.stride + bo;
                    dw[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(a & ~3u), 0, 0);
                }
#pragma unroll
                for (int u = 0; u < PRE_U; ++u) {
                    const int r = min(rb + 2 * u, nrows - 1);
                    const unsigned sh = ((unsigned)min(max(r_lo + r, 0), im.h - 1) * (unsigned)im.stride + bo) & 3u;
#pragma unroll
                    for (int d = 0; d < 3; ++d) px[u][d] = __builtin_amdgcn_alignbyte(dw[u][d + 1], dw[u][d], sh);
                }
                if (!inside) {          // the image's first / last columns: clamped taps, byte by byte
#pragma unroll
                    for (int u = 0; u < PRE_U; ++u) {
                        const int r = min(rb + 2 * u, nrows - 1);
                        const unsigned char *row = im.src + (long long)min(max(r_lo + r, 0), im.h - 1) * im.stride;
                        unsigned char b[12];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int c = 0; c < 3; ++c) b[3 * j + c] = row[cj[j] + c];
#pragma unroll
                        for (int d = 0; d < 3; ++d)
                            px[u][d] = (unsigned)b[4 * d] | ((unsigned)b[4 * d + 1] << 8) | ((unsigned)b[4 * d + 2] << 16) | ((unsigned)b[4 * d + 3] << 24);
                    }
                }
#pragma unroll
                for (int u = 0; u < PRE_U; ++u) {
                    const int r = rb + 2 * u;
                    if (r >= nrows) break;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {          // c = SOURCE channel (byte position in the pixel)
                        int hs = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int byte = 3 * j + c;
                            hs += (int)((px[u][byte >> 2] >> (8 * (byte & 3))) & 0xffu) * ax[j];
                        }
                        s_h[r][c][col] = hs;
                    }
                }
            }
        }
    }
    __syncthreads();
    {   // ---- phase 2: vertical pass, 4 columns per thread
        const int cg = threadIdx.x & 31, lane = threadIdx.x >> 5;                  // 32 column groups x 8 row lanes
        const int dx = dx0 + cg * 4;
        if (dx >= S) return;
        const long long plane = (long long)S * S;
        const bool vec = (S & 3) == 0;           // (then dx + 3 < S and the row start is 16-byte aligned)
        for (int dy = dy0 + lane; dy <= dy1; dy += 8) {
            float fy;
            const int sy = first_tap(dy, im.scale_y, fy);
            int ay[4];
            cubic_weights(fy, ay);
            const int r0 = sy - 1 - r_lo;
            float *o = p.a.out + (long long)blockIdx.z * 3 * plane + (long long)dy * S + dx;
#pragma unroll
            for (int c = 0; c < 3; ++c) {        // c = OUTPUT channel
                const int cs = p.a.swap_rb ? 2 - c : c;
                int acc[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int4 h = *reinterpret_cast<const int4 *>(&s_h[r0 + k][cs][cg * 4]);
                    acc[0] += h.x * ay[k];
                    acc[1] += h.y * ay[k];
                    acc[2] += h.z * ay[k];
                    acc[3] += h.w * ay[k];
                }
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int q = (acc[e] + (1 << 21)) >> 22;
                    q = min(max(q, 0), 255);
                    v[e] = s_lut[c * 256 + q];
                }
                if (vec) {
                    floatx4 vv = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<floatx4 *>(o + c * plane) = vv;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (dx + e < S) o[c * plane + e] = v[e];
                }
            }
        }
    }
}

// output rows per tile such that the source rows a tile touches fit PRE_MAXROWS: first_tap is monotone, and rows dy0 .. dy0 + R - 1
// touch at most ceil((R - 1) * scale) + 1 (float rounding of the map) + 4 (taps) source rows
static int rows_for(double scale_y, int maxr, int maxrows) {
    int R = maxr;
    while (R > 1 && (long long)__builtin_ceil((R - 1) * scale_y) + 6 > maxrows) --R;
    return R;
}

}  // namespace

extern "C" int ppy_preprocess_u8_f32(int n, const unsigned char *const *images, const int *h, const int *w,
                                     const int *row_stride, int swap_rb, int S, const float *lut, float *out,
                                     void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(n > 0 && images && h && w && row_stride && lut && out && S > 0);
    for (int base = 0; base < n; base += PRE_MAX_IMAGES) {
        PreArgs a;
        const int cnt = n - base < PRE_MAX_IMAGES ? n - base : PRE_MAX_IMAGES;
        for (int i = 0; i < cnt; ++i) {
            const int hi = h[base + i], wi = w[base + i];
            PPY_CHECK_ARG(images[base + i] && hi > 0 && wi > 0 && row_stride[base + i] >= 3 * wi);
            // cv::resize with dsize = None: inv_scale = fx as passed (S / w in double), dsize = cvRound(w * inv_scale)
            const double inv_x = (double)S / (double)wi, inv_y = (double)S / (double)hi;
            PPY_CHECK_ARG((long long)__builtin_rint(wi * inv_x) == S && (long long)__builtin_rint(hi * inv_y) == S);
            a.img[i].src = images[base + i];
            a.img[i].h = hi;
            a.img[i].w = wi;
            a.img[i].stride = row_stride[base + i];
            a.img[i].scale_x = 1.0 / inv_x;
            a.img[i].scale_y = 1.0 / inv_y;
        }
        a.lut = lut;
        a.out = out + (long long)base * 3 * S * S;
        a.S = S;
        a.swap_rb = swap_rb;
        const char *e = getenv("PPY_PRE_PIXEL");              // read per call: the round-1 kernel, for A/B runs and as the test partner
        if (e && e[0] == '1') {
            hipLaunchKernelGGL(preprocess_kernel, dim3(ceil_div(S, 64), ceil_div(S, 4), cnt), dim3(256), 0, (hipStream_t)stream, a);
            continue;
        }
        // tile variants (PPY_PRE_VARIANT, read per call; default 0): 0 = 16 output rows / 24 source rows in LDS (40 KB: four workgroups
        // per CU) with four rows of loads in flight; 1 = the same with eight; 2 = 32 / 40 rows (two workgroups per CU); 3 = 8 / 16 rows
        const char *ev = getenv("PPY_PRE_VARIANT");
        const int variant = ev ? atoi(ev) : 0;
        const int maxr = variant == 2 ? 32 : (variant == 3 ? 8 : 16), maxrows = variant == 2 ? 40 : (variant == 3 ? 16 : 24);
        PreTileArgs t;
        t.a = a;
        int rmin = maxr;
        for (int i = 0; i < cnt; ++i) {
            t.rows_per_tile[i] = rows_for(a.img[i].scale_y, maxr, maxrows);
            rmin = t.rows_per_tile[i] < rmin ? t.rows_per_tile[i] : rmin;
        }
        const dim3 grid(ceil_div(S, PRE_TW), ceil_div(S, rmin), cnt);
        if (variant == 1) hipLaunchKernelGGL((preprocess_tile_kernel<24, 8>), grid, dim3(256), 0, (hipStream_t)stream, t);
        else if (variant == 2) hipLaunchKernelGGL((preprocess_tile_kernel<40, 4>), grid, dim3(256), 0, (hipStream_t)stream, t);
        else if (variant == 3) hipLaunchKernelGGL((preprocess_tile_kernel<16, 4>), grid, dim3(256), 0, (hipStream_t)stream, t);
        else hipLaunchKernelGGL((preprocess_tile_kernel<24, 4>), grid, dim3(256), 0, (hipStream_t)stream, t);
    }
    return ppy_launch_status();
}
