// Victim-side bisection of the packed-fp32 hazard: the ARITHMETIC of the decode kernel (csrc/decode_nms.hip), piece
// by piece, as stand-alone victims beside the minimal aggressor of pk_hazard_min.hip (a dependent chain of v_mfma_f32_16x16x32_bf16,
// registers only).  Every victim computes from per-lane inputs in registers, folds the bits of its results into one word per lane,
// and is compared BIT FOR BIT with a solo run of itself.  Built twice: with packed fp32 ops (default) and without
// (-Xclang -target-feature -Xclang -packed-fp32-ops).
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/probes/bin/pk_hazard_decode tools/probes/pk_hazard_decode.hip
//   hipcc --offload-arch=gfx950 -O3 -w -Xclang -target-feature -Xclang -packed-fp32-ops -o tools/probes/bin/pk_hazard_decode_nopk tools/probes/pk_hazard_decode.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

// DK 0: powf pair (IoU-aware score: sigmoid(obj)^0.6 * sigmoid(ioup)^0.4, reference model/head.py:121-125)
//    1: de-sigmoid  -log(clamp(1 / clamp(p) - 1))                                   (:97-109)
//    2: box arithmetic only: centre +- half size, rescale, clip (no transcendental)  (head.py:33-77)
//    3: exp / sigmoid of the box logits feeding the box arithmetic
//    4: everything (one anchor of yolo_decode)
//    5: powf pair on TWO independent values per lane (what lets the compiler pack pow's double-float arithmetic)
//    6: 4 with the logits LOADED from global memory every iteration (waves sleep on vmcnt between the packed stretches)
//    7: 6 + the candidate append (atomicAdd on a counter, store behind it) of the real kernel
template <int DK>
__global__ void __launch_bounds__(256) victim(int iters, unsigned *out, const float *in, unsigned *count, float *cand) {
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    float a = -3.0f + 0.011f * (float)(gid & 511), b = 1.5f - 0.007f * (float)(gid & 255);
    unsigned h = 0;
    auto fold = [&](float v) { h = (h << 5 | h >> 27) ^ __float_as_uint(v); };
    for (int i = 0; i < iters; ++i) {
        a += 0.001f;
        b -= 0.0007f;
        if (DK >= 6) {          // 64 MB of logits, a different line every iteration
            const floatx4 q = *reinterpret_cast<const floatx4 *>(in + ((((size_t)gid * 131 + (size_t)i * 524287) * 4) & ((16u << 20) - 4)));
            a = q[0] + q[1];
            b = q[2] - q[3];
        }
        if (DK == 0 || DK == 4 || DK >= 6) fold(powf(sigm(a), 0.6f) * powf(sigm(b), 0.4f));
        if (DK == 5) {
            fold(powf(sigm(a), 0.6f) * powf(sigm(b), 0.4f));
            fold(powf(sigm(b + 0.3f), 0.6f) * powf(sigm(a - 0.2f), 0.4f));
        }
        if (DK == 1 || DK == 4 || DK >= 6) {
            float p = sigm(a) * 0.9f + 0.01f;
            p = fminf(fmaxf(p, 1e-7f), 1e7f);
            fold(-logf(fminf(fmaxf(1.0f / p - 1.0f, 1e-7f), 1e7f)));
        }
        float tx = a * 0.5f, ty = b, tw = a * 0.25f, th = b * 0.3f;
        if (DK == 3 || DK == 4 || DK >= 6) { tx = sigm(a); ty = sigm(b); tw = expf(a * 0.25f); th = expf(b * 0.3f); }
        if (DK >= 2 && DK != 5) {
            const float gx = (float)(gid & 31), gy = (float)((gid >> 5) & 31), stride = 32.0f;
            const float cx = (1.05f * tx + gx - 0.025f) * stride, cy = (1.05f * ty + gy - 0.025f) * stride;
            const float w = tw * 116.0f, hh = th * 90.0f;
            float x0 = cx - w * 0.5f, y0 = cy - hh * 0.5f, x1 = cx + w * 0.5f, y1 = cy + hh * 0.5f;
            x0 = x0 / 19.0f / stride * 640.0f; y0 = y0 / 19.0f / stride * 480.0f;
            x1 = x1 / 19.0f / stride * 640.0f; y1 = y1 / 19.0f / stride * 480.0f;
            x0 = x0 < 0.0f ? x0 * 0.0f : x0; y0 = y0 < 0.0f ? y0 * 0.0f : y0;
            x1 = x1 > 640.0f ? 640.0f : x1; y1 = y1 > 480.0f ? 480.0f : y1;
            fold(x0); fold(y0); fold(x1); fold(y1);
            if (DK == 7 && x1 - x0 > 100.0f) {      // (which lane gets which slot varies run to run: not folded)
                const unsigned k = atomicAdd(count, 1u) & 0xffff;
                cand[k * 4] = x0; cand[k * 4 + 1] = y0; cand[k * 4 + 2] = x1; cand[k * 4 + 3] = y1;
            }
        }
    }
    out[gid] = h;
}
static const char *kVictims[8] = {"powf(s(a),0.6) * powf(s(b),0.4)", "de-sigmoid (rcp, clamp, log)", "box arithmetic only", "exp / sigmoid + box arithmetic",
                                  "whole decode of one anchor", "two independent powf pairs per lane", "whole decode, logits loaded from memory",
                                  "whole decode, loaded logits + atomic append"};
static const float *g_in;
static unsigned *g_count;
static float *g_cand;
static void launch_victim(int dk, int blocks, int iters, unsigned *out, hipStream_t st) {
    switch (dk) {
        case 0: hipLaunchKernelGGL(victim<0>, dim3(blocks), dim3(256), 0, st, iters, out, g_in, g_count, g_cand); break;
        case 1: hipLaunchKernelGGL(victim<1>, dim3(blocks), dim3(256), 0, st, iters, out, g_in, g_count, g_cand); break;
        case 2: hipLaunchKernelGGL(victim<2>, dim3(blocks), dim3(256), 0, st, iters, out, g_in, g_count, g_cand); break;
        case 3: hipLaunchKernelGGL(victim<3>, dim3(blocks), dim3(256), 0, st, iters, out, g_in, g_count, g_cand); break;
        case 4: hipLaunchKernelGGL(victim<4>, dim3(blocks), dim3(256), 0, st, iters, out, g_in, g_count, g_cand); break;
        case 5: hipLaunchKernelGGL(victim<5>, dim3(blocks), dim3(256), 0, st, iters, out, g_in, g_count, g_cand); break;
        case 6: hipLaunchKernelGGL(victim<6>, dim3(blocks), dim3(256), 0, st, iters, out, g_in, g_count, g_cand); break;
        default: hipLaunchKernelGGL(victim<7>, dim3(blocks), dim3(256), 0, st, iters, out, g_in, g_count, g_cand); break;
    }
}

// the minimal aggressor: dependent chain of v_mfma_f32_16x16x32_bf16 on ONE accumulator, registers only
__global__ void __launch_bounds__(256) aggressor(int iters, float *sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    floatx4 acc = {0, 0, 0, 0};
    bf16x8 ab = {}, bb = {};
    for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)((float)lane * 1e-3f + e); bb[e] = (__bf16)1.0f; }
    for (int i = 0; i < iters; ++i)
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc, 0, 0, 0);
    if (acc[0] == 12345.678f) sink[threadIdx.x] = acc[0];
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 100;
    const int blocks = 2048, n = blocks * 256;
    unsigned *out, *ref;
    float *sink;
    hipMalloc(&out, (size_t)rounds * n * 4); hipMalloc(&ref, (size_t)n * 4); hipMalloc(&sink, 4096);
    {       // logits: a fixed pseudo-random pattern in [-4, 4)
        std::vector<float> hin((size_t)16 << 20);
        unsigned s = 12345u;
        for (auto &v : hin) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (8.0f / 16777216.0f) - 4.0f; }
        float *din;
        hipMalloc(&din, hin.size() * 4);
        hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
        g_in = din;
        hipMalloc(&g_count, 4); hipMemset(g_count, 0, 4);
        hipMalloc(&g_cand, 65536 * 16);
    }
    hipStream_t sv, sa;
    hipStreamCreate(&sv); hipStreamCreate(&sa);
    hipFuncSetAttribute(reinterpret_cast<const void *>(aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned> h((size_t)n), r0((size_t)n);
    for (int dk = 0; dk < 8; ++dk)
        for (int with = 0; with < 2; ++with) {
            launch_victim(dk, blocks, 400, ref, sv);          // solo run = the reference bits
            hipDeviceSynchronize();
            hipMemcpy(r0.data(), ref, (size_t)n * 4, hipMemcpyDeviceToHost);
            for (int r = 0; r < rounds; ++r) {                // queued back to back, no host sync: sustained co-residency
                if (with) hipLaunchKernelGGL(aggressor, dim3(512), dim3(256), 64 * 1024, sa, 3000, sink);
                launch_victim(dk, blocks, 400, out + (size_t)r * n, sv);
            }
            hipDeviceSynchronize();
            int bad_launches = 0;
            unsigned long long lanes = 0;
            for (int r = 0; r < rounds; ++r) {
                hipMemcpy(h.data(), out + (size_t)r * n, (size_t)n * 4, hipMemcpyDeviceToHost);
                unsigned long long d = 0;
                for (int i = 0; i < n; ++i) d += h[i] != r0[i];
                bad_launches += d != 0;
                lanes += d;
            }
            printf("victim %-40s | %-38s | differs from its solo run in %3d of %d launches (%llu lane results)\n", kVictims[dk],
                   with ? "beside the 16x16x32_bf16 chain, 2 WG/CU" : "alone (control)", bad_launches, rounds, lanes);
            fflush(stdout);
        }
    return 0;
}
