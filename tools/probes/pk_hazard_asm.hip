// Victim-side bisection of the packed-fp32 hazard, last step: the exact INSTRUCTION FORMS the decode kernel's box arithmetic compiles
// to (csrc/decode_nms.hip:161-164 with packed ops:  v_pk_mul_f32 .. 0.5 op_sel_hi:[1,0] ;  v_pk_add_f32 .. op_sel:[0,1] op_sel_hi:[1,0]
// neg_lo:[0,1] neg_hi:[0,1] ; v_div_scale_f32 on the result) as inline-asm victims beside the minimal aggressor (dependent chain of
// v_mfma_f32_16x16x32_bf16).  Each lane checks every packed result against the same arithmetic done with scalar VALU ops.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/probes/bin/pk_hazard_asm tools/probes/pk_hazard_asm.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// AK 0: v_pk_add_f32 d, a, b                                   (plain)
//    1: v_pk_add_f32 d, a, b neg_lo:[0,1] neg_hi:[0,1]          (a - b, both halves)
//    2: v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]       (crossed halves of b)
//    3: both: the decode kernel's exact form
//    4: 3 behind v_pk_mul_f32 b, c, 0.5 op_sel_hi:[1,0]         (half sizes from an inline constant)
//    5: 4 + v_div_scale_f32 on the low result right behind it   (VOP3 with an SGPR-pair destination reading the fresh result)
//    6: 5 with the operands arriving from a global load (s_waitcnt vmcnt(0) directly in front of the packed ops)
template <int AK>
__global__ void __launch_bounds__(256) victim(int iters, unsigned *errs, float *bad, const float *in, unsigned *quarters) {
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    const float l = (float)(threadIdx.x & 63);
    unsigned wrong = 0, which = 0;
    float w0 = 0, w1 = 0, e0 = 0, e1 = 0;
    for (int i = 0; i < iters; ++i) {
        floatx2 a = {l * 3.0f + (float)(i & 15), 100.0f + l + (float)(i & 7)}, c = {8.0f + l, 40.0f + 2.0f * l};
        if (false) {
            const floatx4 q = *reinterpret_cast<const floatx4 *>(in + ((((size_t)gid * 131 + (size_t)i * 524287) * 4) & ((16u << 20) - 4)));
            a = floatx2{q[0] * 64.0f + l, q[1] * 32.0f + 100.0f};
            c = floatx2{q[2] * 16.0f + 70.0f, q[3] * 8.0f + 40.0f};
        }
        floatx2 b = c, d;
        float x0, x1;
        if (AK == 4) {
            asm volatile("v_pk_mul_f32 %0, %1, 0.5 op_sel_hi:[1,0]" : "=v"(b) : "v"(c));
        }
        if (AK == 0) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); x0 = a[0] + b[0]; x1 = a[1] + b[1]; }
        if (AK == 1) { asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); x0 = a[0] - b[0]; x1 = a[1] - b[1]; }
        if (AK == 2) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); x0 = a[0] + b[1]; x1 = a[1] + b[0]; }
        if (AK == 3 || AK == 4) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
            float bh0 = AK == 4 ? c[0] * 0.5f : b[0], bh1 = AK == 4 ? c[1] * 0.5f : b[1];
            x0 = a[0] - bh1;
            x1 = a[1] - bh0;
        }
        if (AK == 5) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b)); x0 = a[0] + b[1]; x1 = a[1] + b[1]; }
        if (AK == 6) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); x0 = a[0] + b[0]; x1 = a[1] + b[0]; }
        if (AK == 7) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); x0 = a[1] + b[0]; x1 = a[0] + b[1]; }
        if (AK == 8) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); x0 = a[0] * b[1]; x1 = a[1] * b[0]; }
        if (AK == 9) { asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b)); x0 = fmaf(a[0], b[1], a[0]); x1 = fmaf(a[1], b[0], a[1]); }
        if (AK == 10) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b)); x0 = a[1]; x1 = b[0]; }
        if (d[0] != x0 || d[1] != x1) {
            if (!wrong) { w0 = d[0]; w1 = d[1]; e0 = x0; e1 = x1; }
            ++wrong;
            which |= (d[0] != x0 ? 1u : 0u) | (d[1] != x1 ? 2u : 0u);
        }
    }
    if (wrong) {
        const unsigned k = atomicAdd(errs, 1u);
        atomicOr(quarters, (1u << ((threadIdx.x & 63) >> 4)) | (which << 4));
        if (k < 4) { bad[8 * k] = w0; bad[8 * k + 1] = w1; bad[8 * k + 2] = e0; bad[8 * k + 3] = e1; bad[8 * k + 4] = (float)(threadIdx.x & 63); bad[8 * k + 5] = (float)wrong; }
    }
}
static const char *kVictims[11] = {"v_pk_add_f32 plain", "v_pk_add_f32 neg_lo/neg_hi", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 halves crossed)",
                                   "... crossed + neg_lo/neg_hi (the decode kernel's form)", "pk_mul by inline 0.5, then the crossed / negated pk_add",
                                   "v_pk_add_f32 op_sel:[0,1] (lo result <- src1.hi)", "v_pk_add_f32 op_sel_hi:[1,0] (hi result <- src1.lo)",
                                   "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] (src0 halves crossed)", "v_pk_mul_f32, src1 halves crossed", "v_pk_fma_f32, src1 halves crossed",
                                   "v_pk_mov_b32 op_sel:[1,0]"};
static const float *g_in;
static unsigned *g_quarters;
static void launch_victim(int ak, int blocks, int iters, unsigned *errs, float *bad, hipStream_t st) {
    switch (ak) {
#define VCASE(k) case k: hipLaunchKernelGGL(victim<k>, dim3(blocks), dim3(256), 0, st, iters, errs, bad, g_in, g_quarters); break;
        VCASE(0) VCASE(1) VCASE(2) VCASE(3) VCASE(4) VCASE(5) VCASE(6) VCASE(7) VCASE(8) VCASE(9) VCASE(10)
    }
}

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
static const char *kAggr[10] = {"idle (s_sleep)", "scalar fp32 VALU fma chain", "v_mfma_f32_32x32x2_f32, 4 accumulators", "v_mfma_f32_32x32x16_bf16, 4 accumulators",
                               "v_mfma_f32_32x32x16_f16, 4 accumulators", "v_mfma_f32_32x32x16_bf16, ONE accumulator (dependent)",
                               "v_mfma_f32_16x16x32_bf16, ONE accumulator (dependent)", "v_mfma_f32_16x16x32_bf16, 4 accumulators",
                               "v_mfma_f32_16x16x32_f16, ONE accumulator (dependent)", "v_mfma_f32_32x32x16_f16, ONE accumulator (dependent)"};
template <int KIND>
__global__ void __launch_bounds__(256) aggressor(int iters, float *sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    floatx16 acc[4];
    floatx4 a4[4];
    for (int j = 0; j < 4; ++j) { for (int e = 0; e < 16; ++e) acc[j][e] = 0.f; a4[j] = floatx4{0, 0, 0, 0}; }
    float x = (float)lane * 1e-3f;
    bf16x8 ab = {}, bb = {};
    f16x8 ah = {}, bh = {};
    for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)(x + e); bb[e] = (__bf16)1.0f; ah[e] = (_Float16)(x + e); bh[e] = (_Float16)1.0f; }
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) __builtin_amdgcn_s_sleep(32);
        if (KIND == 1) for (int j = 0; j < 64; ++j) x = fmaf(x, 1.0001f, 0.5f);
        if (KIND == 2) for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.0f, acc[j], 0, 0, 0);
        if (KIND == 3) for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[j], 0, 0, 0);
        if (KIND == 4) for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
        if (KIND == 5) for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[0], 0, 0, 0);
        if (KIND == 6) for (int j = 0; j < 4; ++j) a4[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, a4[0], 0, 0, 0);
        if (KIND == 7) for (int j = 0; j < 4; ++j) a4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, a4[j], 0, 0, 0);
        if (KIND == 8) for (int j = 0; j < 4; ++j) a4[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, a4[0], 0, 0, 0);
        if (KIND == 9) for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0], 0, 0, 0);
    }
    float t = x;
    for (int j = 0; j < 4; ++j) t += acc[j][0] + acc[j][15] + a4[j][0];
    if (t == 12345.678f) sink[threadIdx.x] = t;
}
template <int KIND>
static void launch_aggr(int wgs, size_t lds, int iters, float *sink, hipStream_t st) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(aggressor<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(aggressor<KIND>, dim3(wgs), dim3(256), lds, st, iters, sink);
}
typedef void (*AggrFn)(int, size_t, int, float *, hipStream_t);
static AggrFn kAggrFn[10] = {launch_aggr<0>, launch_aggr<1>, launch_aggr<2>, launch_aggr<3>, launch_aggr<4>, launch_aggr<5>, launch_aggr<6>, launch_aggr<7>,
                             launch_aggr<8>, launch_aggr<9>};

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 100;
    unsigned *errs;
    float *bad, *sink;
    hipMalloc(&errs, 4 * rounds); hipMalloc(&bad, 256); hipMalloc(&sink, 4096); hipMalloc(&g_quarters, 4);
    {
        std::vector<float> hin((size_t)16 << 20);
        unsigned s = 12345u;
        for (auto &v : hin) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.0f / 16777216.0f); }
        float *din;
        hipMalloc(&din, hin.size() * 4);
        hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
        g_in = din;
    }
    hipStream_t sv, sa;
    hipStreamCreate(&sv); hipStreamCreate(&sa);
    std::vector<unsigned> h(rounds);
    for (int ak = 0; ak < 11; ++ak)
        for (int with = 0; with < 2; ++with) {
            hipMemset(errs, 0, 4 * rounds);
            hipMemset(bad, 0, 256);
            hipMemset(g_quarters, 0, 4);
            hipDeviceSynchronize();
            for (int r = 0; r < rounds; ++r) {
                if (with) kAggrFn[6](512, 64 * 1024, 3000, sink, sa);
                launch_victim(ak, 2048, 600, errs + r, bad, sv);
            }
            hipDeviceSynchronize();
            hipMemcpy(h.data(), errs, 4 * rounds, hipMemcpyDeviceToHost);
            float first[8];
            hipMemcpy(first, bad, 32, hipMemcpyDeviceToHost);
            int bl = 0;
            unsigned long long lanes = 0;
            for (int r = 0; r < rounds; ++r) { bl += h[r] != 0; lanes += h[r]; }
            printf("victim %-62s | %-24s | wrong in %3d of %d launches (%llu lanes)", kVictims[ak], with ? "beside the 16x16x32 chain" : "alone (control)", bl, rounds, lanes);
            unsigned qm = 0;
            hipMemcpy(&qm, g_quarters, 4, hipMemcpyDeviceToHost);
            if (bl) printf("  lane quarters hit 0x%x, halves wrong %s%s;", qm & 15, (qm & 16) ? "lo" : "", (qm & 32) ? " hi" : "");
            if (bl) printf("  first: got (%g, %g) expected (%g, %g) lane %g, %g of 600 iterations", first[0], first[1], first[2], first[3], first[4], first[5]);
            printf("\n");
            fflush(stdout);
        }
    printf("\n== the one-instruction victim (v_pk_add_f32 with op_sel:[0,1]: low result <- src1.hi) beside every aggressor\n");
    for (int kind = 0; kind < 10; ++kind)
        for (int li = 0; li < 2; ++li) {
            hipMemset(errs, 0, 4 * rounds);
            hipMemset(g_quarters, 0, 4);
            hipDeviceSynchronize();
            for (int r = 0; r < rounds; ++r) {
                kAggrFn[kind](li == 0 ? 512 : 256, li == 0 ? 64 * 1024 : 96 * 1024, kind == 0 ? 1000 : 3000, sink, sa);
                launch_victim(5, 2048, 600, errs + r, bad, sv);
            }
            hipDeviceSynchronize();
            hipMemcpy(h.data(), errs, 4 * rounds, hipMemcpyDeviceToHost);
            int bl = 0;
            unsigned long long lanes = 0;
            for (int r = 0; r < rounds; ++r) { bl += h[r] != 0; lanes += h[r]; }
            unsigned qm = 0;
            hipMemcpy(&qm, g_quarters, 4, hipMemcpyDeviceToHost);
            printf("aggressor %-56s %d WG/CU | victim wrong in %3d of %d launches (%llu lanes; lane quarters 0x%x)\n", kAggr[kind], li == 0 ? 2 : 1, bl, rounds, lanes, qm & 15);
            fflush(stdout);
        }
    return 0;
}
