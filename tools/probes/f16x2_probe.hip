// Numerics probe: fp32 GEMM emulated with a 2-way fp16 operand split (3 products on v_mfma_f32_32x32x16_f16, fp32
// accumulate), operands pre-scaled by powers of two into the fp16 range, vs the exact-fp32 MFMA and the 3-way bf16
// split (6 products), all against an fp64 host reference.
// build: hipcc --offload-arch=gfx950 -O2 -o f16x2_probe f16x2_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// A [32][K], B [K][32] -> C [32][32].  sa, sb: power-of-two operand scales.  mode 0: fp32 mfma; 1: f16x2 (a0 = RTZ, a1 = RNE);
// 2: f16x2 (both RNE); 3: f16x2 RNE with 4 products
__global__ void probe(const float* A, const float* B, float* C, int K, int mode, float sa, float sb) {
  int l = threadIdx.x, i = l & 31, h = l >> 5;
  f32x16 acc = {0};
  if (mode == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[(k + h) * 32 + i], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 16) {
      f16x8 a0, a1, b0, b1;
      for (int j = 0; j < 8; ++j) {
        int k = k0 + 8 * h + j;
        float x = A[i * K + k] * sa, y = B[k * 32 + i] * sb;
        _Float16 x0, y0;
        x0 = (_Float16)x; y0 = (_Float16)y;                    // RNE
        if (mode == 1) {                                         // RTZ high piece
          auto px = __builtin_amdgcn_cvt_pkrtz(x, y);
          x0 = px[0]; y0 = px[1];
        }
        a0[j] = x0; b0[j] = y0;
        a1[j] = (_Float16)(x - (float)x0);
        b1[j] = (_Float16)(y - (float)y0);
      }
      if (mode == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    }
  }
  const float inv = mode == 0 ? 1.f : 1.f / (sa * sb);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    C[row * 32 + i] = acc[r] * inv;
  }
}

static float pow2_scale(float amax, int target_exp) {   // 2^e with amax * 2^e in [2^(target-1), 2^target)
  int e; frexpf(amax, &e);                              // amax = m * 2^e, m in [0.5, 1)
  return ldexpf(1.f, target_exp - e);
}

int main() {
  printf("# 2-way fp16 split (a*s = a0 + a1, s a power of two placing the tensor maximum just below 2^14), 3 products,\n"
         "# fp32 accumulate, vs exact-fp32 MFMA; error relative to sum|a*b| against fp64.  32x32 outputs.\n"
         "# dist 0: uniform; dist 1: relu-like wide range |u|*exp(3u); dist 2: dist 1 with activations x1e-4 (tiny tensor);\n"
         "# dist 3: dist 1 x1e4 (huge tensor); dist 4: dist 1 with one 1e3 outlier per row (scale set by the outlier)\n");
  const int Ks[3] = {64, 2304, 4608};
  for (int dist = 0; dist < 5; ++dist)
    for (int ki = 0; ki < 3; ++ki) {
      int K = Ks[ki];
      std::vector<float> A(32 * K), B(K * 32);
      srand(7 + ki);
      auto rnd = [&]() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
      for (auto& v : A) v = dist ? fabsf(rnd()) * expf(3.f * rnd()) : rnd();
      if (dist == 2) for (auto& v : A) v *= 1e-4f;
      if (dist == 3) for (auto& v : A) v *= 1e4f;
      if (dist == 4) for (int i = 0; i < 32; ++i) A[i * K + (i * 7) % K] = 1e3f;
      for (auto& v : B) v = rnd() * 0.05f;
      float amaxA = 0, amaxB = 0;
      for (auto v : A) amaxA = fmaxf(amaxA, fabsf(v));
      for (auto v : B) amaxB = fmaxf(amaxB, fabsf(v));
      const float sa = pow2_scale(amaxA, 14), sb = pow2_scale(amaxB, 14);
      std::vector<double> ref(1024), mag(1024);
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double s = 0, m = 0;
          for (int k = 0; k < K; ++k) { double p = (double)A[i * K + k] * B[k * 32 + j]; s += p; m += fabs(p); }
          ref[i * 32 + j] = s; mag[i * 32 + j] = m;
        }
      float *dA, *dB, *dC;
      hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      const char* names[4] = {"fp32 mfma", "f16x2 rtz+rne, 3 prod", "f16x2 rne, 3 prod", "f16x2 rne, 4 prod"};
      for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode, sa, sb);
        std::vector<float> Cc(1024);
        hipMemcpy(Cc.data(), dC, 4096, hipMemcpyDeviceToHost);
        double mx = 0, ss = 0, bias = 0;
        for (int e = 0; e < 1024; ++e) { double d = (Cc[e] - ref[e]) / mag[e]; mx = fmax(mx, fabs(d)); ss += d * d; bias += d; }
        printf("dist %d K %5d  %-24s max|err|/sum|ab| %.3e   rms %.3e   mean %+.2e\n", dist, K, names[mode], mx, sqrt(ss / 1024), bias / 1024);
      }
      hipFree(dA); hipFree(dB); hipFree(dC);
    }
  return 0;
}
