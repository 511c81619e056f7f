// Probe of buffer_load ... lds semantics on gfx950: OOB voffset -> zero fill?  soffset?  LDS placement?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int nbytes, int bad, int soff) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = -7.f;   // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    int wave = threadIdx.x >> 6;
    unsigned voff = ((int)threadIdx.x == bad) ? 0xFFFFFFF0u : threadIdx.x * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + wave * 256), 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    floatx4 v = *reinterpret_cast<floatx4*>(sm + threadIdx.x * 4);
    *reinterpret_cast<floatx4*>(y + threadIdx.x * 4) = v;
}
int main() {
    const int n = 4096;
    std::vector<float> h(n), o(1024);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *x, *y;
    hipMalloc(&x, n * 4); hipMalloc(&y, 1024 * 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int soff : {0, 64}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, x, y, n * 4, 70, soff);
        hipMemcpy(o.data(), y, 1024 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 256; ++t) for (int u = 0; u < 4; ++u) {
            float want = (t == 70) ? 0.f : (float)(t * 4 + u + soff / 4);
            if (o[t * 4 + u] != want) { if (bad < 8) printf("soff %d thread %d elem %d: got %g want %g\n", soff, t, u, o[t*4+u], want); ++bad; }
        }
        printf("soff=%d mismatches=%d  (thread70: %g %g %g %g)\n", soff, bad, o[280], o[281], o[282], o[283]);
    }
    // range check: num_records small -> lanes beyond read zero?
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, x, y, 2048, -1, 0);
    hipMemcpy(o.data(), y, 1024 * 4, hipMemcpyDeviceToHost);
    printf("num_records=2048B: elem[511]=%g elem[512]=%g elem[1023]=%g\n", o[511], o[512], o[1023]);
    return 0;
}
