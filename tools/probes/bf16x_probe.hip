// Numerics probe: fp32 GEMM emulated with 3-way bf16 operand splits on v_mfma_f32_32x32x16_bf16
// (fp32 accumulate) vs the exact-fp32 v_mfma_f32_32x32x2_f32, both against an fp64 host reference.
// build: hipcc --offload-arch=gfx950 -O2 -o bf16x_probe bf16x_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ inline float bf_round(float x, unsigned short& bits) {   // RNE to bf16, return value as fp32
  unsigned u = __float_as_uint(x);
  unsigned r = u + 0x7FFFu + ((u >> 16) & 1u);
  bits = (unsigned short)(r >> 16);
  return __uint_as_float(r & 0xFFFF0000u);
}
__device__ inline float bf_trunc(float x, unsigned short& bits) {
  unsigned u = __float_as_uint(x);
  bits = (unsigned short)(u >> 16);
  return __uint_as_float(u & 0xFFFF0000u);
}

// A [32][K] row-major, B [K][32] row-major, C [32][32].  mode: 0 fp32 mfma, else number of terms (3,6,8,9);
// trunc: split by truncation instead of RNE.
__global__ void probe(const float* A, const float* B, float* C, int K, int mode, int trunc) {
  int l = threadIdx.x, i = l & 31, h = l >> 5;
  f32x16 acc = {0};
  if (mode == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[(k + h) * 32 + i], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 16) {
      union U { bf16x8 v; unsigned short s[8]; };
      U a[3], b[3];
      for (int j = 0; j < 8; ++j) {
        int k = k0 + 8 * h + j;
        float x = A[i * K + k], y = B[k * 32 + i];
        for (int p = 0; p < 3; ++p) {
          unsigned short xb, yb;
          float xr = trunc ? bf_trunc(x, xb) : bf_round(x, xb);
          float yr = trunc ? bf_trunc(y, yb) : bf_round(y, yb);
          a[p].s[j] = xb; b[p].s[j] = yb;
          x -= xr; y -= yr;
        }
      }
      // smallest terms first
      static const int ta[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};
      static const int tb[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0};
      // order: (2,2) (2,1) (1,2) | (2,0) (1,1) (0,2) | (1,0) (0,1) | (0,0); "mode" terms = the LAST mode entries
      for (int t = 9 - mode; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta[t]].v, b[tb[t]].v, acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    C[row * 32 + i] = acc[r];
  }
}

int main() {
  printf("# 3-way bf16 split of both operands (a = a0+a1+a2 exactly); N products = the N largest of the 9 partial\n"
         "# products on v_mfma_f32_32x32x16_bf16, fp32 accumulate.  The conv kernels (conv_x3.hip) use 6 products, RNE split.\n"
         "# dist 0: uniform [-1,1) x 0.05*uniform; dist 1: |u|*exp(3u) (relu-like, wide range) x 0.05*uniform.  32x32 outputs.\n");
  const int Ks[3] = {256, 2304, 4608};
  for (int dist = 0; dist < 2; ++dist)
    for (int ki = 0; ki < 3; ++ki) {
      int K = Ks[ki];
      std::vector<float> A(32 * K), B(K * 32);
      srand(7 + ki);
      auto rnd = [&]() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
      for (auto& v : A) v = dist ? fabsf(rnd()) * expf(3.f * rnd()) : rnd();   // dist 1: relu-like positive, wide range
      for (auto& v : B) v = rnd() * 0.05f;
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
      const int modes[6] = {0, 3, 6, 8, 9, 6};
      for (int mi = 0; mi < 6; ++mi) {
        int trunc = mi == 5;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, modes[mi], trunc);
        std::vector<float> Cc(1024);
        hipMemcpy(Cc.data(), dC, 4096, hipMemcpyDeviceToHost);
        double mx = 0, ss = 0;
        for (int e = 0; e < 1024; ++e) { double d = (Cc[e] - ref[e]) / mag[e]; mx = fmax(mx, fabs(d)); ss += d * d; }
        printf("dist %d K %5d  %-20s max|err|/sum|ab| %.3e   rms %.3e\n", dist, K,
               modes[mi] == 0 ? "fp32 mfma" : (trunc ? "6 products (trunc)" : (modes[mi] == 3 ? "3 products" : modes[mi] == 6 ? "6 products" : modes[mi] == 8 ? "8 products" : "9 products")), mx, sqrt(ss / 1024));
      }
      hipFree(dA); hipFree(dB); hipFree(dC);
    }
  return 0;
}
