// Minimal stand-alone reproducer / bisection of the packed-fp32 hazard of DESIGN.md 4.6 (no torch, no libppyolo_hip):
// a VICTIM kernel that does nothing but v_pk_{add,mul,fma}_f32 on registers and checks itself, beside an AGGRESSOR kernel
// on a second stream that is ONE ingredient of conv_igemm_x3_kernel at a time.  Build (packed ops allowed, the default):
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/probes/bin/pk_hazard_min tools/probes/pk_hazard_min.hip
//   hipcc --offload-arch=gfx950 -O3 -w -shared -fPIC -o tools/probes/bin/libpk_hazard_min.so tools/probes/pk_hazard_min.hip
// Run on the GPU box:  tools/probes/bin/pk_hazard_min [rounds]   -> one line per aggressor: victim launches with errors.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// ---- victims: ITERS dependent packed ops per lane on exactly representable values; errs += lanes whose result is wrong.
// VK = 0: packed ops only; 1: + fp64-rate VALU (v_cvt_f64_f32, v_frexp_exp_i32_f64: what powf / logf expand to);
// 2: + transcendental unit (v_exp_f32, v_rcp_f32, v_log_f32); 3: + LDS round trip (ds_write_b64 / ds_read_b64) of the operands
template <int VK>
__global__ void __launch_bounds__(256) victim(int iters, unsigned *errs, float *bad) {
    __shared__ floatx2 sh[256];
    const float l = (float)(threadIdx.x & 63);
    floatx2 v = {l, l + 1.0f}, w = {1.0f, 1.0f};
    const floatx2 a = {1.0f, 2.0f}, one = {1.0f, 1.0f}, c = {3.0f, 5.0f};
    float side = 1.0f;
    for (int i = 0; i < iters; ++i) {
        v = v + a;                                            // v_pk_add_f32
        w = w * one;                                          // v_pk_mul_f32
        v = __builtin_elementwise_fma(w, c, v) - c;           // v_pk_fma_f32, v_pk_add_f32: net + 0
        if (VK == 1) {
            int ex;
            const double d = frexp((double)(v[1] + side), &ex);                 // fp64-rate ops between the packed ones
            side = (d > 2.0) ? 2.0f : 1.0f;                                      // (always 1)
            w[1] = side;
        }
        if (VK == 2 || VK == 7) {
            side = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(side + 1.0f) - 1.0f) + 0.0f);   // = 1 exactly
            w[0] = side > 0.5f ? 1.0f : 0.0f;
        }
        if (VK == 7) { int ex; const double d = frexp((double)(v[1] + side), &ex); w[1] = (d > 2.0) ? 2.0f : 1.0f; }
        if (VK == 3 || VK == 7) {
            sh[threadIdx.x] = v;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            v = sh[threadIdx.x];
        }
        if (VK == 4 || VK == 7) {                 // packed ops under a partial EXEC mask (divergent branch)
            if ((threadIdx.x + i) & 1) {
                v = v + c;
                asm volatile("" : "+v"(v));
                v = v - c;
            } else {
                v = v * one + a;
                asm volatile("" : "+v"(v));
                v = v - a;
            }
        }
        if (VK == 5 || VK == 7) {                 // operand-select forms: swapped halves, broadcast inline constant
            floatx2 t = {v[1], v[0]};             // op_sel:[1,..] op_sel_hi:[0,..]
            t = t + 1.0f;                         // inline constant, op_sel_hi:[1,0]
            asm volatile("" : "+v"(t));
            t = t - 1.0f;
            v = floatx2{t[1], t[0]};
        }
        if (VK == 6 || VK == 7) {                 // workgroup barrier + wide LDS read each iteration
            __syncthreads();
            const floatx4 q = *reinterpret_cast<const floatx4 *>(&sh[(threadIdx.x & ~1) ^ 2]);
            side = (q[0] != q[0]) ? 2.0f : side;
        }
        asm volatile("" : "+v"(v), "+v"(w), "+v"(side));
    }
    const float e0 = l + (float)iters, e1 = l + 1.0f + 2.0f * (float)iters;
    if (v[0] != e0 || v[1] != e1 || w[0] != 1.0f || w[1] != 1.0f) {
        const unsigned k = atomicAdd(errs, 1u);
        if (k < 8) { bad[4 * k] = v[0] - e0; bad[4 * k + 1] = v[1] - e1; bad[4 * k + 2] = w[0]; bad[4 * k + 3] = w[1]; }
    }
}
static const char *kVictims[8] = {"packed ops only", "packed + fp64-rate VALU (cvt_f64, frexp_f64)", "packed + transcendentals (log, exp2, rcp)",
                                  "packed + LDS round trip", "packed under partial EXEC (divergent branch)",
                                  "packed with op_sel swaps / inline constants", "packed + s_barrier + ds_read_b128", "all of the above"};
static void launch_victim(int vk, int blocks, int iters, unsigned *errs, float *bad, hipStream_t st) {
    switch (vk) {
        case 0: hipLaunchKernelGGL(victim<0>, dim3(blocks), dim3(256), 0, st, iters, errs, bad); break;
        case 1: hipLaunchKernelGGL(victim<1>, dim3(blocks), dim3(256), 0, st, iters, errs, bad); break;
        case 2: hipLaunchKernelGGL(victim<2>, dim3(blocks), dim3(256), 0, st, iters, errs, bad); break;
        case 3: hipLaunchKernelGGL(victim<3>, dim3(blocks), dim3(256), 0, st, iters, errs, bad); break;
        case 4: hipLaunchKernelGGL(victim<4>, dim3(blocks), dim3(256), 0, st, iters, errs, bad); break;
        case 5: hipLaunchKernelGGL(victim<5>, dim3(blocks), dim3(256), 0, st, iters, errs, bad); break;
        case 6: hipLaunchKernelGGL(victim<6>, dim3(blocks), dim3(256), 0, st, iters, errs, bad); break;
        default: hipLaunchKernelGGL(victim<7>, dim3(blocks), dim3(256), 0, st, iters, errs, bad); break;
    }
}

// ---- aggressors: one ingredient each.  256 threads, one workgroup per CU x `wgs`, `lds` bytes of dynamic LDS to
// set the residency the conv kernel has (1-2 workgroups per CU, room left for the victim's waves).
enum { IDLE, VALU_F32, MFMA_F32, MFMA_BF16, MFMA_F16, MFMA_BF16_CVT, MFMA_16x16, LDS_READ, LDS_DMA, MFMA_BF16_DMA, MFMA_32_DEP, MFMA_16_INDEP,
       MFMA_F32_DEP, MFMA_16_F16_DEP, NKINDS };
static const char *kNames[NKINDS] = {"idle spin (s_sleep)", "scalar fp32 VALU fma chain", "v_mfma_f32_32x32x2_f32 (exact fp32 MFMA)",
    "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16", "32x32x16_bf16 + v_cvt_pk_bf16_f32 splits", "v_mfma_f32_16x16x32_bf16, ONE accumulator (dependent chain)",
    "ds_read_b128 stream", "buffer_load_dwordx4 ... lds (LDS-DMA, M0 writes)", "32x32x16_bf16 + LDS-DMA",
    "32x32x16_bf16, ONE accumulator (dependent chain)", "16x16x32_bf16, four independent accumulators", "32x32x2_f32, ONE accumulator",
    "16x16x32_f16, ONE accumulator"};

template <int KIND>
__global__ void __launch_bounds__(256) aggressor(int iters, const float *src, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    floatx16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    floatx4 accs = {0, 0, 0, 0};
    float x = (float)lane * 1e-3f;
    bf16x8 ab = {}, bb = {};
    f16x8 ah = {}, bh = {};
    for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)(x + e); bb[e] = (__bf16)(1.0f); ah[e] = (_Float16)(x + e); bh[e] = (_Float16)1.0f; }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0xFFFFFF00u, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_ptr;
    for (int i = 0; i < iters; ++i) {
        if (KIND == IDLE) __builtin_amdgcn_s_sleep(32);
        if (KIND == VALU_F32) for (int j = 0; j < 64; ++j) x = fmaf(x, 1.0001f, 0.5f);
        if (KIND == MFMA_F32) for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.0f, acc[j], 0, 0, 0);
        if (KIND == MFMA_BF16 || KIND == MFMA_BF16_CVT || KIND == MFMA_BF16_DMA)
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[j], 0, 0, 0);
        if (KIND == MFMA_F16) for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
        if (KIND == MFMA_16x16) for (int j = 0; j < 4; ++j)
            accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, accs, 0, 0, 0);
        if (KIND == MFMA_32_DEP) for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[0], 0, 0, 0);
        if (KIND == MFMA_F32_DEP) for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.0f, acc[0], 0, 0, 0);
        if (KIND == MFMA_16_F16_DEP) for (int j = 0; j < 4; ++j) accs = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, accs, 0, 0, 0);
        if (KIND == MFMA_16_INDEP) {
            floatx4 t4[4];
            for (int j = 0; j < 4; ++j) { t4[j] = floatx4{acc[j][0], acc[j][1], acc[j][2], acc[j][3]}; t4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, t4[j], 0, 0, 0); }
            for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = t4[j][e];
        }
        if (KIND == MFMA_BF16_CVT) {          // the bf16x3 operand split: cvt_pk, shift / mask, subtract
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            for (int e = 0; e < 8; e += 2) {
                const floatx2 f = {x + e, x * 3.0f + e};
                const unsigned P = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
                x += (f[0] - __uint_as_float(P << 16)) + (f[1] - __uint_as_float(P & 0xffff0000u));
                ab[e] = (__bf16)x;
            }
        }
        if (KIND == LDS_READ) for (int j = 0; j < 8; ++j) {
            const floatx4 t = *reinterpret_cast<const floatx4 *>(smem + ((lane * 16 + j * 1024 + i * 64) & 0xfff0));
            x += t[0] + t[3];
        }
        if (KIND == LDS_DMA || KIND == MFMA_BF16_DMA) {
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + wave * 4096 + j * 1024), 16,
                                                         (unsigned)(((i * 4 + j) & 1023) * 1024 + lane * 16), 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    float s = x + accs[0];
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
    if (s == 12345.678f) sink[threadIdx.x] = s;           // keep everything alive
}

template <int KIND>
static void launch(int wgs, size_t lds, int iters, const float *src, float *sink, hipStream_t st) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(aggressor<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(aggressor<KIND>, dim3(wgs), dim3(256), lds, st, iters, src, sink);
}
typedef void (*Launcher)(int, size_t, int, const float *, float *, hipStream_t);
static Launcher kLaunch[NKINDS] = {launch<IDLE>, launch<VALU_F32>, launch<MFMA_F32>, launch<MFMA_BF16>, launch<MFMA_F16>,
    launch<MFMA_BF16_CVT>, launch<MFMA_16x16>, launch<LDS_READ>, launch<LDS_DMA>, launch<MFMA_BF16_DMA>, launch<MFMA_32_DEP>,
    launch<MFMA_16_INDEP>, launch<MFMA_F32_DEP>, launch<MFMA_16_F16_DEP>};

// the same kernels for tools/pk_hazard_bisect.py (ctypes): launch on the caller's streams beside the REAL kernels
extern "C" void pk_victim_launch(int blocks, int iters, unsigned *errs, float *bad, void *stream) {
    launch_victim(0, blocks, iters, errs, bad, (hipStream_t)stream);
}
extern "C" void pk_victim_launch_kind(int vk, int blocks, int iters, unsigned *errs, float *bad, void *stream) {
    launch_victim(vk, blocks, iters, errs, bad, (hipStream_t)stream);
}
extern "C" int pk_num_aggressors() { return NKINDS; }
extern "C" const char *pk_aggressor_name(int kind) { return kind >= 0 && kind < NKINDS ? kNames[kind] : "?"; }
extern "C" void pk_aggressor_launch(int kind, int wgs, long lds, int iters, const float *src, float *sink, void *stream) {
    if (kind >= 0 && kind < NKINDS) kLaunch[kind](wgs, (size_t)lds, iters, src, sink, (hipStream_t)stream);
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 100;
    unsigned *errs;
    float *bad, *src, *sink;
    hipMalloc(&errs, 4 * rounds); hipMalloc(&bad, 128); hipMalloc(&src, 2 << 20); hipMalloc(&sink, 4096);
    hipMemset(src, 0, 2 << 20);
    hipStream_t sv, sa;
    hipStreamCreate(&sv); hipStreamCreate(&sa);
    std::vector<unsigned> h(rounds);
    const size_t ldss[2] = {64 * 1024, 96 * 1024};            // 2 / 1 aggressor workgroups per CU (the conv kernel's residencies)
    for (int vk = 0; vk < 8; ++vk)
        for (int kind = 0; kind < NKINDS; ++kind)
            for (int li = 0; li < 2; ++li) {
                hipMemset(errs, 0, 4 * rounds);
                hipMemset(bad, 0, 128);
                hipDeviceSynchronize();
                for (int r = 0; r < rounds; ++r) {                // queued back to back, no host sync: sustained co-residency
                    kLaunch[kind](256 * (li == 0 ? 2 : 1), ldss[li], kind == IDLE ? 1000 : 3000, src, sink, sa);
                    launch_victim(vk, 2048, 1500, errs + r, bad, sv);
                }
                hipDeviceSynchronize();
                hipMemcpy(h.data(), errs, 4 * rounds, hipMemcpyDeviceToHost);
                float first[4];
                hipMemcpy(first, bad, 16, hipMemcpyDeviceToHost);
                int bad_launches = 0;
                unsigned long long lanes = 0;
                for (int r = 0; r < rounds; ++r) { bad_launches += h[r] != 0; lanes += h[r]; }
                if (bad_launches || kind == MFMA_16x16)
                    printf("victim %-44s | aggressor %-52s lds %3zu KB | wrong: %3d of %d launches (%llu lanes; dv0 %g dv1 %g w %g %g)\n",
                           kVictims[vk], kNames[kind], ldss[li] >> 10, bad_launches, rounds, lanes, first[0], first[1], first[2], first[3]);
                fflush(stdout);
            }
    return 0;
}
