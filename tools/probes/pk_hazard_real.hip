// The REAL decode kernel of this library (csrc/decode_nms.hip, compiled here WITH packed fp32 ops) as the victim, stand-alone (no
// torch, no libppyolo_hip), beside the minimal aggressor of pk_hazard_min.hip -- the starting point for ablating the kernel's own
// source (-DPK_ABL=...).  Compared bit for bit with a solo run.
//   hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -o tools/probes/bin/pk_hazard_real tools/probes/pk_hazard_real.hip
#include "../../pytorch-ppyolo_amd/ppyolo_hip/csrc/decode_nms.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" void ppy_note_hip_error(int) {}      // (capi.hip diagnostic hook)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_p;
typedef float floatx4_p __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) aggressor(int iters, float *sink) {
    extern __shared__ char smem_a[];
    const int lane = threadIdx.x & 63;
    floatx4_p acc = {0, 0, 0, 0};
    bf16x8_p ab = {}, bb = {};
    for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)((float)lane * 1e-3f + e); bb[e] = (__bf16)1.0f; }
    for (int i = 0; i < iters; ++i)
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc, 0, 0, 0);
    if (acc[0] == 12345.678f) sink[threadIdx.x] = acc[0];
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 100;
    const int N = 4, A = 3, C = 80, S[3] = {10, 20, 40}, ds[3] = {32, 16, 8}, ld = 288;      // R50vd head at 320 px, 258 channels
    const float anchors[3][6] = {{116, 90, 156, 198, 373, 326}, {30, 61, 62, 45, 59, 119}, {10, 13, 16, 30, 33, 23}};
    float *head[3];
    int off[3], M = 0;
    unsigned s = 777u;
    for (int l = 0; l < 3; ++l) {
        std::vector<float> h((size_t)N * S[l] * S[l] * ld);
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (8.0f / 16777216.0f) - 5.0f; }
        hipMalloc(&head[l], h.size() * 4);
        hipMemcpy(head[l], h.data(), h.size() * 4, hipMemcpyHostToDevice);
        off[l] = M;
        M += A * S[l] * S[l];
    }
    const int cap = 1 << 16;
    float *boxes, *ims, *sink;
    uint32_t *ck, *ci;
    int *cc;
    hipMalloc(&boxes, (size_t)N * M * 16); hipMalloc(&ims, N * 8); hipMalloc(&sink, 4096);
    hipMalloc(&ck, (size_t)N * cap * 4); hipMalloc(&ci, (size_t)N * cap * 4); hipMalloc(&cc, N * 4);
    std::vector<float> him(N * 2);
    for (int n = 0; n < N; ++n) { him[2 * n] = 480.f; him[2 * n + 1] = 640.f; }
    hipMemcpy(ims, him.data(), N * 8, hipMemcpyHostToDevice);
    hipStream_t sv, sa;
    hipStreamCreate(&sv); hipStreamCreate(&sa);
    hipFuncSetAttribute(reinterpret_cast<const void *>(aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const float *hp[3] = {head[0], head[1], head[2]};
    const int hld[3] = {ld, ld, ld};
    const float *ap[3] = {anchors[0], anchors[1], anchors[2]};
    auto decode = [&]() {
        hipMemsetAsync(cc, 0, N * 4, sv);
        return ppy_yolo_decode_levels_f32(3, hp, hld, S, ds, ap, off, N, A, C, 1.05, 1, 0.4, 1, ims, boxes, M, 0.01f, ck, ci, cc, cap, sv);
    };
    const size_t nb = (size_t)N * M * 4;
    std::vector<float> ref(nb), got(nb);
    int rc = decode();
    hipDeviceSynchronize();
    if (rc != 0) { printf("decode failed: %d\n", rc); return 1; }
    hipMemcpy(ref.data(), boxes, nb * 4, hipMemcpyDeviceToHost);
    for (int with = 0; with < 2; ++with) {
        int bad = 0, shown = 0;
        unsigned long long floats = 0;
        for (int r = 0; r < rounds; ++r) {
            if (with) for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(aggressor, dim3(512), dim3(256), 64 * 1024, sa, 1500, sink);
            hipMemsetAsync(boxes, 0, nb * 4, sv);
            decode();
            hipStreamSynchronize(sv);
            hipMemcpy(got.data(), boxes, nb * 4, hipMemcpyDeviceToHost);
            unsigned long long d = 0;
            for (size_t i = 0; i < nb; ++i) {
                const bool ne = memcmp(&got[i], &ref[i], 4) != 0;
                if (ne && shown < 12) {
                    ++shown;
                    const size_t bx = i / 4;
                    printf("   decode %d box %zu (image %zu, row %zu, lane-ish %zu) coord %zu: got %.6g  solo %.6g   | the box solo: %.6g %.6g %.6g %.6g  got: %.6g %.6g %.6g %.6g\n",
                           r, bx, bx / M, bx % M, (bx % M) % 64, i % 4, got[i], ref[i], ref[bx * 4], ref[bx * 4 + 1], ref[bx * 4 + 2], ref[bx * 4 + 3],
                           got[bx * 4], got[bx * 4 + 1], got[bx * 4 + 2], got[bx * 4 + 3]);
                }
                d += ne;
            }
            bad += d != 0;
            floats += d;
        }
        hipDeviceSynchronize();
        printf("real decode kernel (packed build) | %-40s | %3d of %d decodes differ from the solo run (%llu floats)\n",
               with ? "beside the 16x16x32_bf16 chain, 2 WG/CU" : "alone (control)", bad, rounds, floats);
    }
    return 0;
}
