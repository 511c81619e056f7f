// CU-masked streams on MI355X: which (XCD, SE, CU) does bit i of hipExtStreamCreateWithCUMask's mask enable, does a hipGraph launched
// on such a stream keep the mask, and what does a mask that empties whole XCDs do?  Stand-alone (no torch, no library):
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cu_mask_probe.hip -o tools/probes/bin/cu_mask_probe && timeout 120 tools/probes/bin/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that a grid larger than the enabled CUs really spreads over all of them
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hwid;
        out[2 * blockIdx.x + 1] = xcc;
    }
}

struct Hist {
    int per_xcc[8] = {0};
    int cus = 0;
    std::string text;
};

static Hist summarize(const std::vector<unsigned>& h, int blocks) {
    // (xcc, se, sh, cu) -> count
    static int seen[8][8][2][16];
    memset(seen, 0, sizeof(seen));
    Hist r;
    for (int b = 0; b < blocks; ++b) {
        unsigned hw = h[2 * b], x = h[2 * b + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        if (x < 8) { seen[x][se][sh][cu]++; r.per_xcc[x]++; }
    }
    char line[256];
    for (int x = 0; x < 8; ++x) {
        int n = 0;
        for (int se = 0; se < 8; ++se) for (int sh = 0; sh < 2; ++sh) for (int cu = 0; cu < 16; ++cu) n += seen[x][se][sh][cu] > 0;
        r.cus += n;
        snprintf(line, sizeof(line), " xcc%d:%dcu/%dwg", x, n, r.per_xcc[x]);
        r.text += line;
    }
    return r;
}

static Hist run_on(hipStream_t st, unsigned* d, int blocks, int spin, bool as_graph, hipStream_t capture_stream) {
    std::vector<unsigned> h(2 * blocks);
    CK(hipMemsetAsync(d, 0xff, 2 * blocks * sizeof(unsigned), st));
    CK(hipStreamSynchronize(st));
    if (!as_graph) {
        hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, st, d, spin);
        CK(hipGetLastError());
    } else {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(capture_stream, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, capture_stream, d, spin);
        hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, capture_stream, d, spin);
        CK(hipStreamEndCapture(capture_stream, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, 2 * blocks * sizeof(unsigned), hipMemcpyDeviceToHost));
    return summarize(h, blocks);
}

static void mask_from(std::vector<uint32_t>& m, int nbits, bool (*pred)(int)) {
    m.assign((nbits + 31) / 32, 0);
    for (int i = 0; i < nbits; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32);
}

static std::vector<unsigned> raw_on(hipStream_t st, unsigned* d, int blocks, int spin) {
    std::vector<unsigned> h(2 * blocks);
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, st, d, spin);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, 2 * blocks * sizeof(unsigned), hipMemcpyDeviceToHost));
    return h;
}

static unsigned cu_key(unsigned hw, unsigned xcc) { return ((xcc & 0xf) << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf); }

// stages: "map" (always safe: all CUs but one), "half" (every XCD keeps CUs under either bit order), "xcd" (whole XCDs emptied:
// run it last and under `timeout`), "pair" (two masked streams side by side)
int main(int argc, char** argv) {
    const char* stage = argc > 1 ? argv[1] : "map";
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, stage %s\n", p.name, ncu, stage);
    const int blocks = 4096, spin = 20000;
    unsigned* d;
    CK(hipMalloc(&d, 2 * blocks * sizeof(unsigned)));
    hipStream_t plain;
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    Hist h0 = run_on(plain, d, blocks, spin, false, plain);
    printf("%-34s cus %3d |%s\n", "no mask", h0.cus, h0.text.c_str());
    fflush(stdout);

    if (!strcmp(stage, "map")) {
        std::vector<unsigned> all = raw_on(plain, d, blocks, spin);
        std::vector<char> present(1 << 16, 0);
        for (int b = 0; b < blocks; ++b) present[cu_key(all[2 * b], all[2 * b + 1])] = 1;
        int bits[] = {0, 1, 2, 3, 7, 8, 9, 15, 16, 31, 32, 33, 64, 127, 128, 255};
        for (int bit : bits) {
            if (bit >= ncu) continue;
            std::vector<uint32_t> m((ncu + 31) / 32, 0);
            for (int i = 0; i < ncu; ++i) if (i != bit) m[i / 32] |= 1u << (i % 32);
            hipStream_t st;
            CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)m.size(), m.data()));
            std::vector<char> now(1 << 16, 0);
            // several launches: one pass may miss a CU by chance
            for (int rep = 0; rep < 3; ++rep) {
                std::vector<unsigned> h = raw_on(st, d, blocks, spin);
                for (int b = 0; b < blocks; ++b) now[cu_key(h[2 * b], h[2 * b + 1])] = 1;
            }
            printf("mask without bit %3d: missing", bit);
            for (int k = 0; k < (1 << 16); ++k)
                if (present[k] && !now[k]) printf(" (xcc %d se %d sh %d cu %d)", k >> 12, (k >> 8) & 7, (k >> 4) & 1, k & 0xf);
            printf("\n");
            fflush(stdout);
            CK(hipStreamDestroy(st));
        }
        printf("done\n");
        return 0;
    }

    struct Case { const char* stage; const char* name; bool (*pred)(int); };
    Case cases[] = {
        // every XCD keeps half of its CUs whether bit i is (xcd i % 8, cu i / 8) or (xcd i / 32, cu i % 32)
        {"half", "bits (i%16<8) == ((i/32)%2==0)", [](int i) { return (i % 16 < 8) == ((i / 32) % 2 == 0); }},
        {"half", "the complement", [](int i) { return (i % 16 < 8) != ((i / 32) % 2 == 0); }},
        {"half8", "bits 0..127", [](int i) { return i < 128; }},
        {"half8", "bits 128..255", [](int i) { return i >= 128; }},
        {"half8", "bits (i/8)%2==0", [](int i) { return (i / 8) % 2 == 0; }},
        {"xcd", "bits i%8<4", [](int i) { return i % 8 < 4; }},
        {"xcd", "bits i%8>=4", [](int i) { return i % 8 >= 4; }},
        {"xcd", "bits i%8<2", [](int i) { return i % 8 < 2; }},
        {"xcd", "bits i%8==0", [](int i) { return i % 8 == 0; }},
    };
    for (auto& c : cases) {
        if (strcmp(c.stage, stage)) continue;
        std::vector<uint32_t> m;
        mask_from(m, ncu, c.pred);
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)m.size(), m.data());
        if (e != hipSuccess) { printf("%-34s create failed: %s\n", c.name, hipGetErrorString(e)); continue; }
        printf("%-34s ...\n", c.name);
        fflush(stdout);
        Hist h = run_on(st, d, blocks, spin, false, plain);
        printf("%-34s cus %3d |%s\n", c.name, h.cus, h.text.c_str());
        fflush(stdout);
        Hist hg = run_on(st, d, blocks, spin, true, plain);
        printf("%-34s cus %3d |%s   <- graph captured on a plain stream, launched on the masked one\n", "", hg.cus, hg.text.c_str());
        Hist hg2 = run_on(st, d, blocks, spin, true, st);
        printf("%-34s cus %3d |%s   <- graph captured on the masked stream itself\n", "", hg2.cus, hg2.text.c_str());
        fflush(stdout);
        CK(hipStreamDestroy(st));
    }
    if (!strcmp(stage, "pair") || !strcmp(stage, "pairx")) {
        // two masked streams side by side: do their kernels really overlap in time?
        const bool x = !strcmp(stage, "pairx");
        std::vector<uint32_t> ma, mb;
        if (x) {
            mask_from(ma, ncu, [](int i) { return i % 8 < 4; });
            mask_from(mb, ncu, [](int i) { return i % 8 >= 4; });
        } else {
            mask_from(ma, ncu, [](int i) { return (i % 16 < 8) == ((i / 32) % 2 == 0); });
            mask_from(mb, ncu, [](int i) { return (i % 16 < 8) != ((i / 32) % 2 == 0); });
        }
        hipStream_t sa, sb;
        CK(hipExtStreamCreateWithCUMask(&sa, (uint32_t)ma.size(), ma.data()));
        CK(hipExtStreamCreateWithCUMask(&sb, (uint32_t)mb.size(), mb.data()));
        unsigned* d2;
        CK(hipMalloc(&d2, 2 * blocks * sizeof(unsigned)));
        hipEvent_t e0, e1, e2, e3;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sa));
            hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, sa, d, 200000);
            CK(hipEventRecord(e1, sa));
            CK(hipEventRecord(e2, sb));
            hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, sb, d2, 200000);
            CK(hipEventRecord(e3, sb));
            CK(hipDeviceSynchronize());
            float ta, tb, tab;
            CK(hipEventElapsedTime(&ta, e0, e1)); CK(hipEventElapsedTime(&tb, e2, e3)); CK(hipEventElapsedTime(&tab, e0, e3));
            printf("two half-chip streams: a %.3f ms, b %.3f ms, first start -> last end %.3f ms (serial would be %.3f)\n", ta, tb, tab, ta + tb);
        }
        // the same two launches on two UNMASKED streams
        hipStream_t pa, pb;
        CK(hipStreamCreateWithFlags(&pa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&pb, hipStreamNonBlocking));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, pa));
            hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, pa, d, 200000);
            CK(hipEventRecord(e1, pa));
            CK(hipEventRecord(e2, pb));
            hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, pb, d2, 200000);
            CK(hipEventRecord(e3, pb));
            CK(hipDeviceSynchronize());
            float ta, tb, tab;
            CK(hipEventElapsedTime(&ta, e0, e1)); CK(hipEventElapsedTime(&tb, e2, e3)); CK(hipEventElapsedTime(&tab, e0, e3));
            printf("two unmasked streams:  a %.3f ms, b %.3f ms, first start -> last end %.3f ms\n", ta, tb, tab);
        }
    }
    printf("done\n");
    return 0;
}
