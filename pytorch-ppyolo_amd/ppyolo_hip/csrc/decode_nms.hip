// IoU-aware YOLO box decode + candidate extraction + Matrix-NMS for gfx950.
//
// Replaces (reference) model/head.py:21-141 (get_iou_aware_score, yolo_box), the threshold /
// argsort / top-k of model/matrix_nms.py:102-151 and _matrix_nms (:51-97).  No MFMA here:
// wave ballots / shuffles, LDS histograms and an LDS bitonic network.
//
// The reference materialises [N, 22743, 80] scores (58 MB at R50-608 bs=8) and sorts every
// (box, class) pair above the threshold.  Here the head output is read ONCE: boxes are written
// (2.9 MB), pairs above the threshold are appended to a per-image candidate list, and the
// top nms_top_k are found by an LDS radix select instead of a full sort.
//
// Order semantics: the reference calls torch.argsort(descending=True), which is not stable on
// CPU; this build defines the total order (score desc, candidate index asc) -- identical to
// the reference whenever scores are pairwise distinct.  Matrix-NMS arithmetic (IoU, decay,
// NaN propagation of torch.max/min) is reproduced op for op, so this file is compiled without
// fp contraction.
#include <math.h>
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ uint32_t score_to_key(float s) {
    const uint32_t b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b ^ 0x80000000u);
}
__device__ __forceinline__ float key_to_score(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// Wave-cooperative append of the lanes with `pass` set to the candidate list of image n.
__device__ __forceinline__ void append_candidates(bool pass, float score, uint32_t idx, uint32_t *cand_key,
                                                  uint32_t *cand_idx, int *cand_count, int cand_cap, int lane) {
    const unsigned long long bal = __ballot(pass);
    if (bal == 0ull) return;
    const int cnt = __popcll(bal);
    int base = 0;
    if (lane == 0) base = atomicAdd(cand_count, cnt);
    base = __shfl(base, 0);
    if (pass) {
        const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (pos < cand_cap) {
            cand_key[pos] = score_to_key(score);
            cand_idx[pos] = idx;
        }
    }
}

struct DecodeArgs {
    const float *head;
    const float *im_size;
    float *boxes;
    float *scores_dense;
    uint32_t *cand_key, *cand_idx;
    int *cand_count;
    int head_ld, N, S, A, C, M_total, box_offset, cand_cap, iou_aware, clip;
    float anchors[16];
    float stride_f, sxy, sxy_bias, e_obj, e_iou, thr;
};

// One (cell, anchor): IoU-aware objectness (reference model/head.py:121-126 + _de_sigmoid :97-109), the candidate bound in the
// logit domain, and the box (head.py:40-46, :61-77).  Shared by both decode kernels: the arithmetic is written ONCE.
__device__ __forceinline__ void decode_pair(const DecodeArgs &p, float ioup_logit, float t0, float t1, float t2, float t3,
                                            float obj_logit, int w, int h, int a, float im_h, float im_w, float &conf_out,
                                            float &bound_out, floatx4 &bb_out) {
    const float Sf = (float)p.S;
    if (p.iou_aware) {
        const float ioup = sigmoidf_(ioup_logit);
        const float obj = sigmoidf_(obj_logit);
        float nw = powf(obj, p.e_obj) * powf(ioup, p.e_iou);
        nw = fminf(fmaxf(nw, 1e-7f), 1e7f);
        nw = 1.0f / nw - 1.0f;
        nw = fminf(fmaxf(nw, 1e-7f), 1e7f);
        obj_logit = -logf(nw);
    }
    const float conf = sigmoidf_(obj_logit);
    // candidate bound in the logit domain, with a safety margin (exact test follows)
    float bound;
    if (p.thr <= 0.0f) {
        bound = -INFINITY;
    } else if (!(conf > p.thr)) {
        bound = INFINITY;                      // conf*sigmoid(.) <= conf <= thr: nothing passes
    } else {
        const float tq = p.thr / conf;         // in (0, 1)
        bound = logf(tq / (1.0f - tq)) - 0.02f;
    }
    conf_out = conf;
    bound_out = bound;
    const float bx = (p.sxy * sigmoidf_(t0) + (float)w - p.sxy_bias) * p.stride_f;
    const float by = (p.sxy * sigmoidf_(t1) + (float)h - p.sxy_bias) * p.stride_f;
    const float bw = expf(t2) * p.anchors[2 * a], bh = expf(t3) * p.anchors[2 * a + 1];
    float x0 = (bx - bw / 2.0f) / Sf / p.stride_f * im_w;
    float y0 = (by - bh / 2.0f) / Sf / p.stride_f * im_h;
    float x1 = (bx + bw / 2.0f) / Sf / p.stride_f * im_w;
    float y1 = (by + bh / 2.0f) / Sf / p.stride_f * im_h;
    if (p.clip) {
        x0 = x0 < 0.0f ? x0 * 0.0f : x0;   // keeps the reference's -0.0
        y0 = y0 < 0.0f ? y0 * 0.0f : y0;
        x1 = x1 > im_w ? im_w : x1;
        y1 = y1 > im_h ? im_h : y1;
    }
    bb_out = floatx4{x0, y0, x1, y1};
}

// 64 grid cells per 256-thread workgroup, three phases through LDS:
//   1. the cells' channels (contiguous in NHWC) are staged with coalesced loads;
//   2. one THREAD per (cell, anchor) does the IoU-aware objectness, the box decode and a
//      logit-domain candidate bound: score = conf*sigmoid(cls) > thr  <=>  cls > logit(thr/conf),
//      so the 80 class channels of a box need no exp at all unless they can pass;
//   3. one WAVE per (cell, anchor) sweeps the class logits against that bound (minus a safety
//      margin); only the survivors get the exact fp32 score and the exact `score > thr` test, then
//      a wave ballot appends them to the image's candidate list.
constexpr int DEC_CELLS = 64;

__device__ __forceinline__ void yolo_decode_body(const DecodeArgs &p, const int bx, const int by) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int per = 5 + p.C;
    const int nch = p.A * per + (p.iou_aware ? p.A : 0);
    float *vals = dsm;                                  // [DEC_CELLS][nch]
    float *s_conf = dsm + DEC_CELLS * nch;              // [DEC_CELLS * A]
    float *s_bound = s_conf + DEC_CELLS * p.A;          // [DEC_CELLS * A]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cells_img = p.S * p.S;
    const int cell_in_img0 = bx * DEC_CELLS;
    const long long cell0 = (long long)by * cells_img + cell_in_img0;
    const int ncl = min(DEC_CELLS, cells_img - cell_in_img0);

    // ---- phase 1: stage ----
    if (p.head_ld == nch) {
        // contiguous block of cells: 16-byte loads from the aligned-down address, all issued before
        // the first LDS write (a dependent 4-byte load loop here was latency-bound)
        const float *src = p.head + cell0 * nch;
        const int pre = (int)(((uintptr_t)src & 15) >> 2);     // floats between the 16-B line start and src
        const float *base = src - pre;
        const int total = ncl * nch + pre;
        const int n4 = total >> 2;                              // whole 16-byte groups
        constexpr int U = 8;
        for (int i0 = 0; i0 < n4; i0 += 256 * U) {
            floatx4 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 256 + tid;
                if (i < n4) r[u] = *reinterpret_cast<const floatx4 *>(base + 4 * i);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 256 + tid;
                if (i < n4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = 4 * i + e - pre;
                        if (idx >= 0) vals[idx] = r[u][e];
                    }
                }
            }
        }
        for (int i = (n4 << 2) - pre + tid; i < ncl * nch; i += 256)
            if (i >= 0) vals[i] = src[i];
    } else if ((p.head_ld & 3) == 0 && (((uintptr_t)p.head) & 15) == 0) {
        // rows with a padded pixel stride (the plan rounds K up to 4: 258 channels in rows of 260): 16-byte loads per row,
        // ceil(nch / 4) groups each, all of a batch issued before the first LDS write
        const int gpr = (nch + 3) >> 2, n4 = ncl * gpr;
        constexpr int U = 8;
        for (int i0 = 0; i0 < n4; i0 += 256 * U) {
            floatx4 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 256 + tid;
                if (i < n4) {
                    const int c = i / gpr, g = i - c * gpr;
                    r[u] = *reinterpret_cast<const floatx4 *>(p.head + (cell0 + c) * p.head_ld + 4 * g);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 256 + tid;
                if (i < n4) {
                    const int c = i / gpr, g = i - c * gpr;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * g + e < nch) vals[c * nch + 4 * g + e] = r[u][e];
                }
            }
        }
    } else {
        for (int c = 0; c < ncl; ++c)
            for (int i = tid; i < nch; i += 256) vals[c * nch + i] = p.head[(cell0 + c) * p.head_ld + i];
    }
    __syncthreads();

    // ---- phase 2: one thread per (cell, anchor) ----
    const int off0 = p.iou_aware ? p.A : 0;
    for (int q = tid; q < ncl * p.A; q += 256) {
        const int c = q / p.A, a = q - c * p.A;
        const long long cell = cell0 + c;
        const int w = (int)(cell % p.S), h = (int)((cell / p.S) % p.S), n = (int)(cell / ((long long)p.S * p.S));
        const float *v = vals + c * nch;
        const float *t = v + off0 + a * per;
        float conf, bound;
        floatx4 bb;
        decode_pair(p, p.iou_aware ? v[a] : 0.0f, t[0], t[1], t[2], t[3], t[4], w, h, a, p.im_size[n * 2 + 0], p.im_size[n * 2 + 1],
                    conf, bound, bb);
        s_conf[q] = conf;
        s_bound[q] = bound;
        const int box = p.box_offset + (h * p.S + w) * p.A + a;
        *reinterpret_cast<floatx4 *>(p.boxes + ((long long)n * p.M_total + box) * 4) = bb;
    }
    __syncthreads();

    // ---- phase 3: the same thread sweeps its box's class logits against the bound ----
    // A logit above the bound is rare (it is exactly the candidate condition up to the safety
    // margin), so the sweep is a read + compare per class; survivors get the exact fp32 score and
    // the exact `score > thr` test and go to a workgroup-local list in LDS.  ONE global atomic
    // per workgroup then reserves the slots (a returning atomic per wave on the image's single
    // counter serialised at ~90 atomics/us and dominated an earlier version).  A workgroup never
    // spans two images (grid = blocks-per-image x N).
    constexpr int LCAP = 1536;
    __shared__ uint32_t l_key[LCAP], l_idx[LCAP];
    __shared__ int l_n, l_base;
    if (tid == 0) l_n = 0;
    __syncthreads();
    const int n = by;
    uint32_t *ckey = p.cand_key + (long long)n * p.cand_cap;
    uint32_t *cidx = p.cand_idx + (long long)n * p.cand_cap;
    for (int q = tid; q < ncl * p.A; q += 256) {
        const int c = q / p.A, a = q - c * p.A;
        const float bound = s_bound[q];
        if (bound == INFINITY && !p.scores_dense) continue;
        const int cellw = cell_in_img0 + c;
        const int w = cellw % p.S, h = cellw / p.S;
        const int box = p.box_offset + (h * p.S + w) * p.A + a;
        const float conf = s_conf[q];
        const float *cl = vals + c * nch + off0 + a * per + 5;
        for (int k = 0; k < p.C; ++k) {
            const float lg = cl[k];
            if (p.scores_dense || lg > bound) {
                const float sc = conf * sigmoidf_(lg);
                if (p.scores_dense) p.scores_dense[((long long)n * p.M_total + box) * p.C + k] = sc;
                if (sc > p.thr) {
                    const int pos = atomicAdd(&l_n, 1);
                    if (pos < LCAP) {
                        l_key[pos] = score_to_key(sc);
                        l_idx[pos] = (uint32_t)(box * p.C + k);
                    } else {                      // dense regime: list full, append straight to global
                        const int g = atomicAdd(p.cand_count + n, 1);
                        if (g < p.cand_cap) {
                            ckey[g] = score_to_key(sc);
                            cidx[g] = (uint32_t)(box * p.C + k);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    const int nl = min(l_n, LCAP);
    if (tid == 0 && nl > 0) l_base = atomicAdd(p.cand_count + n, nl);
    __syncthreads();
    for (int i = tid; i < nl; i += 256) {
        const int g = l_base + i;
        if (g < p.cand_cap) {
            ckey[g] = l_key[i];
            cidx[g] = l_idx[i];
        }
    }
}

__global__ void __launch_bounds__(256) yolo_decode_kernel(const DecodeArgs p) { yolo_decode_body(p, blockIdx.x, blockIdx.y); }

// All head levels in ONE launch (blockIdx.x walks level 0's cell blocks, then level 1's, ...): the 19x19 and
// 38x38 levels are 48 / 184 workgroups each and were latency-bound on their own (19 + 20 + 36 us as three launches).
constexpr int DEC_MAX_LEVELS = 4;
struct DecodeMulti {
    DecodeArgs lv[DEC_MAX_LEVELS];
    int nb[DEC_MAX_LEVELS];
    int nlevels;
};
__global__ void __launch_bounds__(256) yolo_decode_multi_kernel(const DecodeMulti m) {
    int bx = blockIdx.x, l = 0;
    while (l + 1 < m.nlevels && bx >= m.nb[l]) {
        bx -= m.nb[l];
        ++l;
    }
    yolo_decode_body(m.lv[l], bx, blockIdx.y);
}

// ---------------------------------------------------------------------------------------
// Streaming decode (round 3): the same result as yolo_decode_body, organised for HBM bandwidth instead of around workgroup
// barriers and an LDS copy of the data.  A WAVE owns groups of 16 consecutive cells of one image and level; FOUR lanes share a
// cell (lane = 4 * cell + r): load i of lane r is the 16-byte group 4 i + r of the cell's row, so a wave instruction reads 16
// rows x 64 contiguous bytes, all DS_NJ loads of a group are in flight together, and the data STAYS IN REGISTERS.  Lane r < A
// of a cell is the cell's pair lane for anchor r: it fetches the pair's 6 header values with scalar loads (same cache lines),
// runs decode_pair, writes the box and holds conf / bound; its three neighbours get them by a quad permute (DPP, no LDS).  The
// channel of element (i, e) of lane r is 16 i + 4 r + e -- with A, C and the IoU-aware flag as template parameters the anchor
// and class of almost every element fold at compile time, so the sweep is one compare per class logit.  Survivors of the
// logit-domain bound are rare: they go UNSCORED (logit, conf, index) to a wave-private LDS list; the exact fp32 score and the
// exact `score > thr` test run when the list is flushed, and the image's candidate list is appended to with ONE global atomic
// per flush (normally one per wave; every 768 entries in the dense regime).  No workgroup barrier anywhere.
constexpr int DS_GC = 16;                 // cells per group (4 lanes each)
constexpr int DS_LW = 768;                // wave-private list of unscored survivors (entries)
constexpr int DS_WAVES = 8;               // waves per workgroup (72 KB of lists: two workgroups per CU)
struct DecodeStream {
    DecodeArgs lv[4];
    int nb[4];                            // workgroups per image of each level
    int nlevels;
    int abl;                              // experiments (PPY_DECODE_ABL; results are garbage): 1 = no flush atomic, 2 = no pair phase, 4 = no sweep
};

template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {      // value of lane (lane & ~3) + Q
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), Q * 0x55, 0xf, 0xf, false));
}

static_assert(3 * DS_WAVES * DS_LW * 4 + 64 <= PPY_LDS_MAX / 2, "the wave-private lists of TWO workgroups must fit one CU's LDS");
template <int A, int C, bool IOU>
__global__ void __launch_bounds__(64 * DS_WAVES, 4) yolo_decode_stream_kernel(const DecodeStream m) {
    constexpr int PER = 5 + C, OFF0 = IOU ? A : 0, NCH = A * PER + OFF0, N4ROW = (NCH + 3) / 4, NJ = (N4ROW + 3) / 4;
    static_assert(A >= 1 && A <= 4, "one pair lane per anchor inside a quad");
    __shared__ float l_all[DS_WAVES][3 * DS_LW];      // per wave: logits, confidences, indices of the list -- or the dense regime's key cache
    int bx = blockIdx.x, l = 0;
    while (l + 1 < m.nlevels && bx >= m.nb[l]) {
        bx -= m.nb[l];
        ++l;
    }
    const DecodeArgs &p = m.lv[l];
    const int nbx = m.nb[l];
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cells_img = p.S * p.S;
    const int ngroups = (cells_img + DS_GC - 1) / DS_GC;
    const float im_h = p.im_size[n * 2 + 0], im_w = p.im_size[n * 2 + 1];
    uint32_t *ckey = p.cand_key + (long long)n * p.cand_cap;
    uint32_t *cidx = p.cand_idx + (long long)n * p.cand_cap;
    float *wlg = l_all[wave], *wcf = wlg + DS_LW;
    uint32_t *wix = reinterpret_cast<uint32_t *>(wlg + 2 * DS_LW);
    int l_n = 0;                           // wave-uniform
    const int c = lane >> 2, r = lane & 3;

    // The private list: (1) exact score and test, in place (logit slot <- key, index slot <- ~0 when the entry fails) -> number of
    // survivors (wave-uniform); (2) append them at `base` of the image's candidate list.
    auto score_list = [&]() -> int {
        int total = 0;
        for (int i0 = 0; i0 < l_n; i0 += 64) {
            const int i = i0 + lane;
            bool ok = false;
            if (i < l_n) {
                const float sc = wcf[i] * sigmoidf_(wlg[i]);
                ok = sc > p.thr;
                wlg[i] = __uint_as_float(score_to_key(sc));
                if (!ok) wix[i] = 0xffffffffu;
            }
            total += __popcll(__ballot(ok));
        }
        return total;
    };
    auto write_list = [&](int base) {
        for (int i0 = 0; i0 < l_n; i0 += 64) {
            const int i = i0 + lane;
            const uint32_t ix = i < l_n ? wix[i] : 0xffffffffu;
            const bool ok = ix != 0xffffffffu;
            const unsigned long long bal = __ballot(ok);
            if (ok) {
                const int g = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (g < p.cand_cap) {
                    ckey[g] = __float_as_uint(wlg[i]);
                    cidx[g] = ix;
                }
            }
            base += __popcll(bal);
        }
        l_n = 0;
    };
    // A returning device-scope atomic on an image's ONE counter costs ~75 ns of serialised time (measured: 3790 of them, one
    // per wave, took the launch from 19 to 53 us; tools/decode_bench.py), so the normal case reserves ONCE PER WORKGROUP, at the
    // end; only a list that fills up mid-way (the all-pass regime) is flushed by its wave alone.
    auto flush_now = [&]() {
        const int total = score_list();
        int base = 0;
        if (lane == 0 && total > 0 && !(m.abl & 1)) base = atomicAdd(p.cand_count + n, total);
        write_list(__shfl(base, 0));
    };

    // (the trip count is the same for all waves of a workgroup -- a wave without a group of its own runs the iteration empty --
    // because the dense regime below reserves list space per WORKGROUP, behind barriers)
    __shared__ int w_cnt[2][DS_WAVES], w_base[2];
    for (int g0 = bx * DS_WAVES; g0 < ngroups; g0 += nbx * DS_WAVES) {
        const int g = g0 + wave;
        const bool have = g < ngroups;
        if (l_n > DS_LW / 2) flush_now();          // (several groups per wave, all of them busy: keep half of the list for this one)
        const int l_start = l_n;
        bool dense = false;
        const int cell_base = g * DS_GC;
        const int ncl = have ? min(DS_GC, cells_img - cell_base) : 0;
        const bool cell_ok = c < ncl;
        const float *row = p.head + ((long long)n * cells_img + cell_base + c) * p.head_ld;
        // round 6: the row is requested through a buffer resource -- a lane without a cell (or beyond the row's last 16-byte group) gets an
        // out-of-range offset and reads zeros.  The predicated form (`v = 0; if (ok) v = load`) put a branch and a zero fill in front of
        // every load, and hipcc placed `s_waitcnt vmcnt(1)` / `vmcnt(0)` in front of the 13th and 14th: twelve loads in flight, then a
        // round trip, then the last five (ISA of the round-5 kernel) -- not the seventeen this loop was written for.
        floatx4 v[NJ];
        {
            constexpr unsigned DS_OOB = 0x80000000u;      // = num_records (a level's head output is far below 2 GB; checked by the entry point)
            const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void *)p.head, 0, DS_OOB, 0x00020000);
            const unsigned rbase = cell_ok ? (unsigned)(((long long)n * cells_img + cell_base + c) * p.head_ld * 4) + (unsigned)r * 16u : DS_OOB;
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                const unsigned off = (4 * i + r < N4ROW) ? rbase + (unsigned)i * 64u : DS_OOB;
                v[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rh, (int)off, 0, 0));
            }
        }
        // pair lanes (r < A): header values, decode_pair, box store.  Round 6: the six header values of anchor r (IoU logit, t0 .. t4) are
        // taken from the quad's registers -- channel ch of the row sits in element ch % 4 of v[ch / 16] of quad lane (ch / 4) % 4 -- instead
        // of being loaded again: the reload was a second memory round trip in every wave's chain (and, short of registers, hipcc waited
        // for ALL seventeen row loads before it could form the reload's address)
        auto header = [&](int ch_q0, int step_q) -> float {      // channel ch_q0 + q * step_q for pair lane q
            float val = 0.0f;
#pragma unroll
            for (int q = 0; q < A; ++q) {
                const int ch = ch_q0 + q * step_q;
                const float src = v[ch >> 4][ch & 3];
                float b;
                switch ((ch >> 2) & 3) {
                    case 0: b = quad_bcast<0>(src); break;
                    case 1: b = quad_bcast<1>(src); break;
                    case 2: b = quad_bcast<2>(src); break;
                    default: b = quad_bcast<3>(src); break;
                }
                val = r == q ? b : val;
            }
            return val;
        };
        const float h_iou = IOU ? header(0, 1) : 0.0f;
        const float h_t0 = header(OFF0 + 0, PER), h_t1 = header(OFF0 + 1, PER), h_t2 = header(OFF0 + 2, PER);
        const float h_t3 = header(OFF0 + 3, PER), h_t4 = header(OFF0 + 4, PER);
        float conf = 0.0f, bound = INFINITY;
        if (cell_ok && r < A && !(m.abl & 2)) {
            const float iou_logit = h_iou;
            const float t0 = h_t0, t1 = h_t1, t2 = h_t2, t3 = h_t3, t4 = h_t4;
            const int cell = cell_base + c;
            const int h = cell / p.S, w = cell - h * p.S;
            floatx4 bb;
            decode_pair(p, iou_logit, t0, t1, t2, t3, t4, w, h, r, im_h, im_w, conf, bound, bb);
            const int box = p.box_offset + cell * A + r;
            *reinterpret_cast<floatx4 *>(p.boxes + ((long long)n * p.M_total + box) * 4) = bb;
        }
        float bnd[4], cnf[4];
        bnd[0] = quad_bcast<0>(bound); cnf[0] = quad_bcast<0>(conf);
        bnd[1] = quad_bcast<1>(bound); cnf[1] = quad_bcast<1>(conf);
        bnd[2] = quad_bcast<2>(bound); cnf[2] = quad_bcast<2>(conf);
        bnd[3] = quad_bcast<3>(bound); cnf[3] = quad_bcast<3>(conf);
        // class sweep: channel of element (i, e) = 16 i + 4 r + e
        const int box0 = p.box_offset + (cell_base + c) * A;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            if (m.abl & 4) break;
            if (l_n + 256 > DS_LW) {                  // no room for the 4 x 64 elements of this load: this group ALONE has > DS_LW / 2 - 256
                dense = true;                         // survivors of the bound -- the dense regime: redo it below without the list
                l_n = l_start;
                break;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = 16 * i + 4 * r + e;
                if (16 * i + e >= NCH) continue;                              // (compile time: beyond the row for every r)
                const int t = ch - OFF0;
                int a = 0;
#pragma unroll
                for (int q = 1; q < A; ++q) a += (t >= q * PER) ? 1 : 0;
                const int k = t - a * PER - 5;
                const bool cls = ch < NCH && t >= 0 && k >= 0;
                float b = bnd[0], cf = cnf[0];
#pragma unroll
                for (int q = 1; q < A; ++q) {
                    b = a >= q ? bnd[q] : b;
                    cf = a >= q ? cnf[q] : cf;
                }
                const float lg = v[i][e];
                const bool pass = cls && lg > b;
                const unsigned long long bal = __ballot(pass);
                if (bal == 0ull) continue;                                    // (the usual case)
                if (pass) {
                    const int pos = l_n + __popcll(bal & ((1ull << lane) - 1ull));
                    wlg[pos] = lg;
                    wcf[pos] = cf;
                    wix[pos] = (uint32_t)((box0 + a) * C + k);
                }
                l_n += __popcll(bal);
            }
        }
        // Dense regime (worst case: every pair of every cell passes, 3840 candidates per group).  Filling and flushing the 768-entry
        // list cost one returning atomic on the image's counter per 768 candidates -- 2370 per image, ~75 ns of serialised time
        // each: 180 of the launch's 263 us (round 3); one per group (474) or per half group (948) still queued 36 / 71 us behind
        // one address.  Now a group that overflows the list leaves it: in two halves of eight cells, a compact loop re-reads the
        // rows (8 KB, L2-hot, the next cell's row in flight) 64 channels per step, scores every pair ONCE and parks the key (0 =
        // not a candidate) in the wave's list storage -- 8 x 258 words of its 2304 --, counts the half's survivors, the
        // WORKGROUP reserves for its eight waves with ONE atomic (118 per image), and a second pass over the parked keys stores
        // them, consecutive lanes to consecutive slots.  (From the registers of the sweep above the same thing is 272 unrolled
        // bodies whose per-element invariants spill.)  conf / bound of (cell, anchor) sit in lane 4 cell + anchor.
        if (__syncthreads_or(dense ? 1 : 0)) {
            if (dense && l_start > 0) {        // the pending entries of earlier groups own the storage: out with them first
                l_n = l_start;
                flush_now();
            }
            if (dense) l_n = 0;
            uint32_t *park = reinterpret_cast<uint32_t *>(wlg);
            constexpr int NCK = (NCH + 63) / 64, HALF = DS_GC / 2;
            static_assert(HALF * NCH <= 3 * DS_LW, "the parked keys of half a group fit the wave's list storage");
#pragma unroll 1
            for (int c0 = 0; c0 < DS_GC; c0 += HALF) {
                const int hb = (c0 / HALF) & 1;
                const int c1 = dense ? min(c0 + HALF, ncl) : c0;          // (a wave that is not dense: nothing to do, but it meets the barriers)
                const float *grow = p.head + ((long long)n * cells_img + cell_base) * p.head_ld;
                auto load_row = [&](int cc, float (&dst)[NCK]) {
                    const float *rw = grow + (long long)cc * p.head_ld;
#pragma unroll
                    for (int j = 0; j < NCK; ++j) dst[j] = (cc < c1 && 64 * j + lane < NCH) ? rw[64 * j + lane] : 0.0f;
                };
                float cur[NCK], nxt[NCK];
                load_row(c0, cur);
                int cnt = 0;
                for (int cc = c0; cc < c1; ++cc) {
                    load_row(cc + 1, nxt);
                    float cfa[A], bda[A];
#pragma unroll
                    for (int q = 0; q < A; ++q) {
                        cfa[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(conf), 4 * cc + q));
                        bda[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bound), 4 * cc + q));
                    }
#pragma unroll
                    for (int j = 0; j < NCK; ++j) {
                        const int ch = 64 * j + lane;
                        const int t = ch - OFF0;
                        int a = 0;
#pragma unroll
                        for (int q = 1; q < A; ++q) a += (t >= q * PER) ? 1 : 0;
                        const int k = t - a * PER - 5;
                        const bool cls = ch < NCH && t >= 0 && k >= 0;
                        float b = bda[0], cf = cfa[0];
#pragma unroll
                        for (int q = 1; q < A; ++q) {
                            b = a >= q ? bda[q] : b;
                            cf = a >= q ? cfa[q] : cf;
                        }
                        const bool pass = cls && cur[j] > b;
                        uint32_t key = 0u;
                        if (__ballot(pass) != 0ull) {
                            const float sc = cf * sigmoidf_(cur[j]);
                            const bool ok = pass && sc > p.thr;
                            key = ok ? score_to_key(sc) : 0u;          // (a score above a threshold >= 0 is positive: its key has the top bit set)
                            cnt += __popcll(__ballot(ok));
                        }
                        if (ch < NCH) park[(cc - c0) * NCH + ch] = key;
                    }
#pragma unroll
                    for (int j = 0; j < NCK; ++j) cur[j] = nxt[j];
                }
                if (lane == 0) w_cnt[hb][wave] = cnt;
                __syncthreads();
                if (tid == 0) {
                    int sum = 0;
#pragma unroll
                    for (int w = 0; w < DS_WAVES; ++w) sum += w_cnt[hb][w];
                    w_base[hb] = (sum > 0 && !(m.abl & 1)) ? atomicAdd(p.cand_count + n, sum) : 0;
                }
                __syncthreads();
                int base = w_base[hb];
                for (int w = 0; w < wave; ++w) base += w_cnt[hb][w];
                for (int cc = c0; cc < c1; ++cc) {
                    const int boxc = p.box_offset + (cell_base + cc) * A;
#pragma unroll
                    for (int j = 0; j < NCK; ++j) {
                        const int ch = 64 * j + lane;
                        const uint32_t key = ch < NCH ? park[(cc - c0) * NCH + ch] : 0u;
                        const unsigned long long bal = __ballot(key != 0u);
                        if (key != 0u) {
                            const int t = ch - OFF0;
                            int a = 0;
#pragma unroll
                            for (int q = 1; q < A; ++q) a += (t >= q * PER) ? 1 : 0;
                            const int gpos = base + __popcll(bal & ((1ull << lane) - 1ull));
                            if (gpos < p.cand_cap) {
                                ckey[gpos] = key;
                                cidx[gpos] = (uint32_t)((boxc + a) * C + (t - a * PER - 5));
                            }
                        }
                        base += __popcll(bal);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    {   // one reservation for the whole workgroup
        __shared__ int w_tot[DS_WAVES], wg_base;
        const int total = score_list();
        if (lane == 0) w_tot[wave] = total;
        __syncthreads();
        if (tid == 0) {
            int sum = 0;
#pragma unroll
            for (int w = 0; w < DS_WAVES; ++w) sum += w_tot[w];
            wg_base = (sum > 0 && !(m.abl & 1)) ? atomicAdd(p.cand_count + n, sum) : 0;
        }
        __syncthreads();
        int base = wg_base;
        for (int w = 0; w < wave; ++w) base += w_tot[w];
        write_list(base);
    }
}

// Candidate extraction from dense scores [N][M][C] (reference model/matrix_nms.py:110-117).
__global__ void __launch_bounds__(256) dense_candidates_kernel(const float *__restrict__ scores, int M, int C,
                                                               float thr, uint32_t *cand_key, uint32_t *cand_idx,
                                                               int *cand_count, int cand_cap) {
    const int n = blockIdx.y, lane = threadIdx.x & 63;
    const long long total = (long long)M * C;
    const long long span = (long long)gridDim.x * blockDim.x;
    const long long rounds = (total + span - 1) / span;
    for (long long r = 0; r < rounds; ++r) {
        const long long i = r * span + (long long)blockIdx.x * blockDim.x + threadIdx.x;
        float s = 0.f;
        bool pass = false;
        if (i < total) {
            s = scores[(long long)n * total + i];
            pass = s > thr;
        }
        append_candidates(pass, s, (uint32_t)i, cand_key + (long long)n * cand_cap,
                          cand_idx + (long long)n * cand_cap, cand_count + n, cand_cap, lane);
    }
}

// ---------------------------------------------------------------------------------------
// Matrix-NMS: one 1024-thread workgroup per image.
constexpr int NT = 1024;      // threads
constexpr int KMAX = 1024;    // max nms_top_k
constexpr int RBITS = 11;     // radix-select digit
constexpr int CCAP = 8192;    // candidates staged in LDS (64 KB of dynamic shared memory)

// inclusive suffix sum over the workgroup (threads >= tid)
__device__ __forceinline__ int block_suffix_sum(int v, int *scratch /*[16+1]*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_down(s, d);
        if (lane + d < 64) s += o;
    }
    if (lane == 0) scratch[wv] = s;
    __syncthreads();
    int add = 0;
    for (int k = wv + 1; k < NT / 64; ++k) add += scratch[k];
    __syncthreads();
    return s + add;
}

// Descending sort of n <= KMAX DISTINCT non-zero keys (zeros = padding, they all land behind the real keys)
// by counting: rank(e) = #{i : key[i] > key[e]}.  Every thread of a wave reads the same key[i] (LDS
// broadcast), P2/NT threads share an element, and the whole sort costs two barriers instead of the
// log^2 barrier-separated stages of a bitonic network (1024-thread barriers are what this kernel waits on).
__device__ __forceinline__ void rank_sort_desc(const unsigned long long *in, unsigned long long *out, int n, int P2,
                                               int *rank /*[KMAX]*/) {
    const int tid = threadIdx.x;
    const int parts = NT / P2;                 // threads per element (P2 = power of two >= n, >= 64)
    const int e = tid & (P2 - 1), part = tid / P2;
    if (tid < KMAX) rank[tid] = 0;
    if (tid < KMAX) out[tid] = 0ull;
    __syncthreads();
    if (e < n) {
        const unsigned long long mine = in[e];
        const int len = (n + parts - 1) / parts, lo = part * len, hi = min(lo + len, n);
        int r = 0, i = lo;
        for (; i + 8 <= hi; i += 8) {          // 8 independent LDS reads in flight (the loop is latency-bound otherwise)
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) r += v[u] > mine ? 1 : 0;
        }
        for (; i < hi; ++i) r += in[i] > mine ? 1 : 0;
        if (parts == 1) rank[e] = r; else atomicAdd(&rank[e], r);
    }
    __syncthreads();
    if (tid < n && in[tid] != 0ull) out[rank[tid]] = in[tid];
    __syncthreads();
}

// MSB-first radix select over `count` keys of `total_bits` bits (key_of(c), c < count): the largest T such that at least `need`
// keys are >= T, exact unless a digit's bin holds exactly the keys still needed (then the whole bin is taken: T = the bin's
// lower edge, same set).  All NT threads of the workgroup call it; hist = [1 << RBITS], scratch = [32], sel = [3] in LDS.
template <typename KeyFn>
__device__ __forceinline__ unsigned long long radix_select_threshold(KeyFn key_of, int count, int need, int total_bits,
                                                                       unsigned int *hist, int *scratch, int *sel) {
    const int tid = threadIdx.x;
    unsigned long long prefix = 0ull;
    int shift = total_bits;
    while (shift > 0) {
        const int bits = shift < RBITS ? shift : RBITS;
        const int hi_shift = shift;
        shift -= bits;
        for (int i = tid; i < (1 << RBITS); i += NT) hist[i] = 0u;
        __syncthreads();
        for (int c = tid; c < count; c += NT) {
            const unsigned long long k = key_of(c);
            if ((hi_shift >= 64 ? 0ull : (k >> hi_shift)) == prefix)
                atomicAdd(&hist[(unsigned)((k >> shift) & ((1ull << bits) - 1ull))], 1u);
        }
        __syncthreads();
        const int h0 = (int)hist[2 * tid], h1 = (int)hist[2 * tid + 1];
        const int incl = block_suffix_sum(h0 + h1, scratch);
        const int after = incl - (h0 + h1);   // candidates in bins above this thread's pair
        if (after < need && need <= after + h1) {
            sel[0] = 2 * tid + 1; sel[1] = after; sel[2] = h1;
        } else if (after + h1 < need && need <= after + h1 + h0) {
            sel[0] = 2 * tid; sel[1] = after + h1; sel[2] = h0;
        }
        __syncthreads();
        prefix = (prefix << bits) | (unsigned long long)sel[0];
        need -= sel[1];
        const bool whole_bin = (sel[2] == need);
        __syncthreads();
        if (whole_bin) break;
    }
    return prefix << shift;
}

__device__ __forceinline__ float nanmin_(float a, float b) { return (a != a || b != b) ? NAN : fminf(a, b); }

struct NmsArgs {
    const float *boxes;
    const uint32_t *cand_key, *cand_idx;
    const int *cand_count;
    float *out_dets;
    int *out_count, *out_keep;
    int M_total, C, cand_cap, top_k, keep_k, gaussian, idx_bits, N;
    float post_thr, sigma;
    char *ws;                  // per-image intermediates between the four kernels (NmsWs below)
};

// Matrix-NMS runs as FOUR kernels.  As one 1024-thread workgroup per image (the first version) it took
// 141 us per step, 71 % of it in the two pairwise phases: 125 k box pairs x ~45 VALU instructions incl. two
// IEEE divisions is simply VALU-bound on ONE CU, while 248 CUs idle.  Now:
//   A  nms_select_kernel  (1 workgroup / image)   top-k select + sort + box gather           -> ws
//   B  nms_colmax_kernel  (NMS_G workgroups / image)  compensate-IoU column maxima of a column slice -> ws.comp
//   C  nms_decay_kernel   (NMS_G workgroups / image)  decay column minima of a column slice          -> ws.decay
//   D  nms_finish_kernel  (1 workgroup / image)   rescore, post-threshold, second sort, keep_top_k
// Kernel boundaries are the synchronisation (no spin-waits, no cross-XCD coherence games); arithmetic and
// order of every value are unchanged, so the results stay bit-identical.
constexpr int NMS_G = 32;     // (round 3: 8 -> 32 workgroups per image: one column per wave at nms_top_k = 500; the two pairwise kernels 8.6 + 12.5 -> us)
struct NmsWs {                 // one per image
    float box[KMAX][4];
    float score[KMAX], comp[KMAX], decay[KMAX];
    int label[KMAX], flat[KMAX];
    int K, pad[15];
};

// Large candidate lists (round 4).  A list of more than CCAP entries does not fit the select kernel's LDS cache, and walking
// it from global memory in ONE workgroup per image took 2.9 ms per step in the all-pass regime (1.8 M candidates per image,
// six walks of dependent loads on 8 of 256 CUs).  Now two chip-wide steps cut such a list down to a few thousand entries first:
//   S  nms_sample_kernel  (1 workgroup / image)   NMS_SAMPLE keys from evenly spaced strata (a hashed position in each), the
//      r-th largest of them by the LDS radix select = a score-key threshold t with ~4 x nms_top_k entries expected above it
//   C  nms_collect_kernel (NMS_CG workgroups / image)   one streaming pass over the keys: every entry with key >= t (ALL ties
//      included) goes to a compact list; the exact number of such entries is counted on the way
// and the select kernel runs on the compact list iff it is certain to hold the whole top-k: count(key >= t) >= nms_top_k, no
// overflow.  Otherwise (probability ~1e-9 for random order; heavy ties) it walks the original list as before -- the result is
// the same either way: the top-k by (score desc, index asc) of a superset of the top-k.
constexpr int NMS_SAMPLE = CCAP;          // sampled keys (staged in the same LDS cache)
constexpr int NMS_CG = 64;                // workgroups per image of the collect pass
constexpr int NMS_CCAP2 = 32768;          // compact list capacity per image (entries)
constexpr int NMS_CL = 4096;              // survivors one collect workgroup can hold
struct NmsBig {                            // one per image, behind the NmsWs array
    uint32_t thr_key;
    int ccount;                            // entries with key >= thr_key (exact), -1: the list is small, nothing to do
    int bad;                               // a collect workgroup overflowed: the compact list is incomplete
    int pad[13];
};
__device__ __forceinline__ NmsBig *nms_big(const NmsArgs &p, int n) {
    return reinterpret_cast<NmsBig *>(p.ws + (size_t)p.N * sizeof(NmsWs)) + n;
}
__device__ __forceinline__ uint32_t *nms_compact(const NmsArgs &p, int n) {      // [NMS_CCAP2] keys, then [NMS_CCAP2] indices
    return reinterpret_cast<uint32_t *>(p.ws + (size_t)p.N * (sizeof(NmsWs) + sizeof(NmsBig))) + (size_t)n * 2 * NMS_CCAP2;
}

__global__ void __launch_bounds__(NT) nms_sample_kernel(const NmsArgs p) {
    __shared__ unsigned int hist[1 << RBITS];
    __shared__ int scratch[32];
    __shared__ int s_sel[3];
    extern __shared__ unsigned long long scache[];      // [NMS_SAMPLE] (dynamic)
    const int n = blockIdx.x, tid = threadIdx.x;
    const int count = min(p.cand_count[n], p.cand_cap);
    NmsBig *hdr = nms_big(p, n);
    if (count <= CCAP) {
        if (tid == 0) { hdr->ccount = -1; hdr->bad = 0; }
        return;
    }
    const uint32_t *ckey = p.cand_key + (long long)n * p.cand_cap;
    // stratum i = [i count / S, (i + 1) count / S): one key from a hashed position inside it (a fixed stride would resonate
    // with the 240 (anchor, class) pairs per cell of the decode's append order)
    for (int base = 0; base < NMS_SAMPLE; base += NT * 8) {
        uint32_t kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned i = (unsigned)(base + u * NT + tid);
            const long long lo = (long long)i * count / NMS_SAMPLE, hi = (long long)(i + 1) * count / NMS_SAMPLE;
            const unsigned h = (i * 2654435761u) ^ ((i * 2246822519u) >> 15);
            kk[u] = ckey[lo + (long long)(h % (unsigned)(hi - lo))];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) scache[base + u * NT + tid] = (unsigned long long)kk[u];
    }
    __syncthreads();
    // the r-th largest sample sits at rank ~ r count / S of the whole list: aim at 4 x top_k, never fewer than 16 samples
    long long r = (4ll * p.top_k * NMS_SAMPLE + count - 1) / count;
    r = r < 16 ? 16 : (r > NMS_SAMPLE / 2 ? NMS_SAMPLE / 2 : r);
    auto key_of = [&](int c) -> unsigned long long { return scache[c]; };
    const unsigned long long T = radix_select_threshold(key_of, NMS_SAMPLE, (int)r, 32, hist, scratch, s_sel);
    if (tid == 0) { hdr->thr_key = (uint32_t)T; hdr->ccount = 0; hdr->bad = 0; }
}

__global__ void __launch_bounds__(512) nms_collect_kernel(const NmsArgs p) {
    __shared__ int l_pos[NMS_CL];
    __shared__ int l_n, l_base;
    const int n = blockIdx.y, tid = threadIdx.x;
    const int count = min(p.cand_count[n], p.cand_cap);
    if (count <= CCAP) return;
    NmsBig *hdr = nms_big(p, n);
    const uint32_t t = hdr->thr_key;
    const uint32_t *ckey = p.cand_key + (long long)n * p.cand_cap;
    const uint32_t *cidx = p.cand_idx + (long long)n * p.cand_cap;
    if (tid == 0) l_n = 0;
    __syncthreads();
    auto take = [&](uint32_t k, int c) {
        if (k >= t) {
            const int q = atomicAdd(&l_n, 1);
            if (q < NMS_CL) l_pos[q] = c;
        }
    };
    // this workgroup's slice, in units of four keys (16-byte loads when the list is aligned; four of them in flight per thread)
    const int quads = (count + 3) >> 2;
    const int per = (quads + NMS_CG - 1) / NMS_CG;
    const int q0 = blockIdx.x * per, q1 = min(q0 + per, quads);
    if ((reinterpret_cast<uintptr_t>(ckey) & 15) == 0 && (p.cand_cap & 3) == 0) {      // (whole quads inside this image's list)
        constexpr int U = 4;
        for (int qb = q0; qb < q1; qb += 512 * U) {
            uintx4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = qb + u * 512 + tid;
                v[u] = uintx4{0u, 0u, 0u, 0u};
                if (q < q1) v[u] = *reinterpret_cast<const uintx4 *>(ckey + 4 * (long long)q);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = qb + u * 512 + tid;
                if (q < q1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * q + e < count) take(v[u][e], 4 * q + e);
                }
            }
        }
    } else {
        for (int c = 4 * q0 + tid; c < min(4 * q1, count); c += 512) take(ckey[c], c);
    }
    __syncthreads();
    const int total = l_n, nl = min(total, NMS_CL);
    if (tid == 0) {
        l_base = total > 0 ? atomicAdd(&hdr->ccount, total) : 0;
        if (total > NMS_CL) atomicExch(&hdr->bad, 1);
    }
    __syncthreads();
    uint32_t *k2 = nms_compact(p, n), *i2 = k2 + NMS_CCAP2;
    for (int i = tid; i < nl; i += 512) {
        const int g = l_base + i, c = l_pos[i];
        if (g < NMS_CCAP2) {
            k2[g] = ckey[c];
            i2[g] = cidx[c];
        }
    }
}

__global__ void __launch_bounds__(NT) nms_select_kernel(const NmsArgs p) {
    __shared__ unsigned long long skey[KMAX], skey2[KMAX];
    __shared__ unsigned int hist[1 << RBITS];
    int *srank = reinterpret_cast<int *>(hist);                 // the histogram is dead once the threshold is known
    __shared__ int scratch[32];
    __shared__ int s_sel[3], s_cnt;

    const int n = blockIdx.x, tid = threadIdx.x;
    const uint32_t *ckey = p.cand_key + (long long)n * p.cand_cap;
    const uint32_t *cidx = p.cand_idx + (long long)n * p.cand_cap;
    int count = min(p.cand_count[n], p.cand_cap);
    if (count > CCAP) {          // a large list: the compact one instead, iff it holds every entry >= its threshold and >= top_k of them
        const NmsBig *hdr = nms_big(p, n);
        const int cc = hdr->ccount;
        if (hdr->bad == 0 && cc >= p.top_k && cc <= NMS_CCAP2) {
            ckey = nms_compact(p, n);
            cidx = ckey + NMS_CCAP2;
            count = cc;
        }
    }
    const int ib = p.idx_bits;
    const unsigned long long idx_mask = (1ull << ib) - 1ull;
    float *dets = p.out_dets + (long long)n * p.keep_k * 6;
    int *keep = p.out_keep + (long long)n * p.keep_k;

    for (int i = tid; i < p.keep_k * 6; i += NT) dets[i] = -1.0f;
    for (int i = tid; i < p.keep_k; i += NT) keep[i] = -1;
    NmsWs *w = reinterpret_cast<NmsWs *>(p.ws) + n;
    if (count == 0) {
        if (tid == 0) { p.out_count[n] = 0; w->K = 0; }
        return;
    }
    // composite key: larger == earlier in (score desc, candidate index asc)
    auto comp_global = [&](int c) -> unsigned long long {
        return ((unsigned long long)ckey[c] << ib) | (idx_mask - (unsigned long long)cidx[c]);
    };
    // The select passes below walk the candidate list up to six times; from global memory every walk is a
    // chain of dependent ~2 us loads (an LDS atomic sits between consecutive iterations).  Lists of up to
    // CCAP candidates are therefore staged in LDS once, with all loads of a thread in flight together.
    extern __shared__ unsigned long long scache[];      // [CCAP] (dynamic)
    const bool cached = count <= CCAP;
    if (cached) {
        for (int base = 0; base < count; base += NT * 8) {
            uint32_t kk[8], ii[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = base + u * NT + tid;
                kk[u] = c < count ? ckey[c] : 0u;
                ii[u] = c < count ? cidx[c] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = base + u * NT + tid;
                if (c < count) scache[c] = ((unsigned long long)kk[u] << ib) | (idx_mask - (unsigned long long)ii[u]);
            }
        }
        __syncthreads();
    }
    auto comp_of = [&](int c) -> unsigned long long { return cached ? scache[c] : comp_global(c); };

    // ---- 1. top-k threshold by MSB-first radix select (reference :120-125) ----
    const int K = min(p.top_k, count);
    unsigned long long T = 0ull;
    if (count > p.top_k) T = radix_select_threshold(comp_of, count, K, 32 + ib, hist, scratch, s_sel);
    // ---- 2. collect the K survivors, sort them (score desc, index asc) ----
    if (tid == 0) s_cnt = 0;
    for (int i = tid; i < KMAX; i += NT) skey[i] = 0ull;
    __syncthreads();
    for (int c = tid; c < count; c += NT) {
        const unsigned long long k = comp_of(c);
        if (k >= T) {
            const int pos = atomicAdd(&s_cnt, 1);
            if (pos < KMAX) skey[pos] = k + 1ull;   // +1: real keys are > the 0 padding
        }
    }
    __syncthreads();
    int P = 64;
    while (P < K) P <<= 1;
    rank_sort_desc(skey, skey2, K, P, srank);
    if (tid < K) {
        const unsigned long long k = skey2[tid] - 1ull;
        const int flat = (int)(idx_mask - (k & idx_mask));
        const int box = flat / p.C;
        w->flat[tid] = flat;
        w->label[tid] = flat - box * p.C;
        w->score[tid] = key_to_score((uint32_t)(k >> ib));
        *reinterpret_cast<floatx4 *>(&w->box[tid][0]) =
            *reinterpret_cast<const floatx4 *>(p.boxes + ((long long)n * p.M_total + box) * 4);
    }
    if (tid == 0) w->K = K;
}


// D[i][j] = IoU(i,j) * same_class(i,j) for i < j.  One WAVE per column j, lanes stride over the rows i < j,
// wave butterfly reductions; NaN propagates through max / min exactly like torch.max / torch.min.
__device__ __forceinline__ float nms_dval(const float (*sbox)[4], const int *slabel, int i, int lj, float bj0, float bj1,
                                          float bj2, float bj3, float area_j) {
    const floatx4 a = *reinterpret_cast<const floatx4 *>(&sbox[i][0]);
    const float iw = fmaxf(fminf(a[2], bj2) - fmaxf(a[0], bj0), 0.0f);
    const float ih = fmaxf(fminf(a[3], bj3) - fmaxf(a[1], bj1), 0.0f);
    const float inter = iw * ih;
    const float area_i = (a[2] - a[0]) * (a[3] - a[1]);
    const float iou = inter / ((area_i + area_j) - inter);
    return iou * (slabel[i] == lj ? 1.0f : 0.0f);
}

// 3a. compensate IoU: column max over the whole column (zeros on/below the diagonal)   (reference _matrix_nms :51-97)
__global__ void __launch_bounds__(NT) nms_colmax_kernel(const NmsArgs p) {
    __shared__ __attribute__((aligned(16))) float sbox[KMAX][4];
    __shared__ int slabel[KMAX];
    NmsWs *w = reinterpret_cast<NmsWs *>(p.ws) + blockIdx.y;
    const int K = w->K, tid = threadIdx.x;
    if (K == 0) return;
    if (tid < K) {
        *reinterpret_cast<floatx4 *>(&sbox[tid][0]) = *reinterpret_cast<const floatx4 *>(&w->box[tid][0]);
        slabel[tid] = w->label[tid];
    }
    __syncthreads();
    const int lane = tid & 63, gw = blockIdx.x * (NT / 64) + (tid >> 6), TW = gridDim.x * (NT / 64);
    for (int j = gw; j < K; j += TW) {
        const float bj0 = sbox[j][0], bj1 = sbox[j][1], bj2 = sbox[j][2], bj3 = sbox[j][3];
        const float area_j = (bj2 - bj0) * (bj3 - bj1);
        const int lj = slabel[j];
        float mx = 0.0f;
        bool nan = false;
        for (int i = lane; i < j; i += 64) {
            const float d = nms_dval(sbox, slabel, i, lj, bj0, bj1, bj2, bj3, area_j);
            nan |= (d != d);
            mx = fmaxf(mx, d);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const bool anynan = __ballot(nan) != 0ull;
        if (lane == 0) w->comp[j] = anynan ? NAN : mx;
    }
}

// 3b. decay coefficient: column min of (1-D)/(1-comp[i]) [gaussian: exp(-s D^2)/exp(-s comp^2)] over i < j, joined with
// the rows i >= j (D == 0 there) through a suffix nan-min of f(i) = 1/(1-comp[i])
__global__ void __launch_bounds__(NT) nms_decay_kernel(const NmsArgs p) {
    __shared__ __attribute__((aligned(16))) float sbox[KMAX][4];
    __shared__ int slabel[KMAX];
    __shared__ float scomp[KMAX], ssuf[KMAX], swave[NT / 64];
    NmsWs *w = reinterpret_cast<NmsWs *>(p.ws) + blockIdx.y;
    const int K = w->K, tid = threadIdx.x;
    if (K == 0) return;
    float f = INFINITY;
    if (tid < K) {
        *reinterpret_cast<floatx4 *>(&sbox[tid][0]) = *reinterpret_cast<const floatx4 *>(&w->box[tid][0]);
        slabel[tid] = w->label[tid];
        const float c = w->comp[tid];
        scomp[tid] = c;
        f = p.gaussian ? (expf(-1.0f * p.sigma * (0.0f * 0.0f)) / expf(-1.0f * p.sigma * (c * c)))
                       : ((1.0f - 0.0f) / (1.0f - c));
    }
    {   // suffix nan-min scan: inside each wave by shuffles, then the totals of the waves behind it
        float v = f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_down(v, d);
            if ((tid & 63) + d < 64) v = nanmin_(v, o);
        }
        if ((tid & 63) == 0) swave[tid >> 6] = v;
        __syncthreads();
        for (int k = (tid >> 6) + 1; k < NT / 64; ++k) v = nanmin_(v, swave[k]);
        ssuf[tid] = v;
        __syncthreads();
    }
    const int lane = tid & 63, gw = blockIdx.x * (NT / 64) + (tid >> 6), TW = gridDim.x * (NT / 64);
    for (int j = gw; j < K; j += TW) {
        const float bj0 = sbox[j][0], bj1 = sbox[j][1], bj2 = sbox[j][2], bj3 = sbox[j][3];
        const float area_j = (bj2 - bj0) * (bj3 - bj1);
        const int lj = slabel[j];
        float mn = INFINITY;
        bool nan = false;
        for (int i = lane; i < j; i += 64) {
            const float d = nms_dval(sbox, slabel, i, lj, bj0, bj1, bj2, bj3, area_j);
            const float c = scomp[i];
            const float t = p.gaussian ? (expf(-1.0f * p.sigma * (d * d)) / expf(-1.0f * p.sigma * (c * c)))
                                       : ((1.0f - d) / (1.0f - c));
            nan |= (t != t);
            mn = fminf(mn, t);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
        const bool anynan = __ballot(nan) != 0ull;
        if (lane == 0) w->decay[j] = nanmin_(anynan ? NAN : mn, ssuf[j]);
    }
}

// 4. rescore, post-threshold (>=), second sort, keep_top_k (reference matrix_nms :128-145)
__global__ void __launch_bounds__(NT) nms_finish_kernel(const NmsArgs p) {
    __shared__ unsigned long long skey[KMAX], skey2[KMAX];
    __shared__ int srank[KMAX];
    __shared__ float sscore[KMAX];
    __shared__ int s_cnt;
    const int n = blockIdx.x, tid = threadIdx.x;
    NmsWs *w = reinterpret_cast<NmsWs *>(p.ws) + n;
    const int K = w->K;
    if (K == 0) return;                    // out_count / padding were written by nms_select_kernel
    float *dets = p.out_dets + (long long)n * p.keep_k * 6;
    int *keep = p.out_keep + (long long)n * p.keep_k;
    if (tid == 0) s_cnt = 0;
    skey[tid] = 0ull;
    __syncthreads();
    if (tid < K) {
        const float ns = w->score[tid] * w->decay[tid];
        sscore[tid] = ns;
        if (ns >= p.post_thr) {
            atomicAdd(&s_cnt, 1);
            skey[tid] = (((unsigned long long)score_to_key(ns) << 10) | (unsigned long long)(KMAX - 1 - tid)) + 1ull;
        }
    }
    __syncthreads();
    const int kept = s_cnt;
    int P = 64;
    while (P < K) P <<= 1;
    rank_sort_desc(skey, skey2, K, P, srank);
    const int nout = min(kept, p.keep_k);
    if (tid < nout) {
        const int src = KMAX - 1 - (int)((skey2[tid] - 1ull) & 1023ull);
        float *o = dets + tid * 6;
        o[0] = (float)w->label[src];
        o[1] = sscore[src];
        o[2] = w->box[src][0]; o[3] = w->box[src][1]; o[4] = w->box[src][2]; o[5] = w->box[src][3];
        keep[tid] = w->flat[src];
    }
    if (tid == 0) p.out_count[n] = nout;
}

}  // namespace

static int decode_pack(DecodeArgs &p, const float *head_out, int head_ld, int N, int S, int A, int num_classes,
                       const float *h_anchors_px, int downsample, double scale_x_y, int iou_aware,
                       double iou_aware_factor, int clip_bbox, const float *im_size, float *boxes, int M_total,
                       int box_offset, float score_threshold, uint32_t *cand_key, uint32_t *cand_idx, int *cand_count,
                       int cand_cap, float *scores_dense, size_t *lds_out) {
    PPY_CHECK_ARG(head_out && h_anchors_px && im_size && boxes && cand_key && cand_idx && cand_count);
    PPY_CHECK_ARG(N > 0 && S > 0 && A > 0 && A <= 8 && num_classes > 0 && downsample > 0 && cand_cap > 0);
    const int nch = A * (5 + num_classes) + (iou_aware ? A : 0);
    PPY_CHECK_ARG(nch <= 272 && head_ld >= nch);
    PPY_CHECK_ARG(box_offset >= 0 && box_offset + S * S * A <= M_total);
    PPY_CHECK_ARG((long long)M_total * num_classes < (1ll << 31));
    PPY_CHECK_ARG(((uintptr_t)boxes & 15) == 0);
    p.head = head_out; p.im_size = im_size; p.boxes = boxes; p.scores_dense = scores_dense;
    p.cand_key = cand_key; p.cand_idx = cand_idx; p.cand_count = cand_count;
    p.head_ld = head_ld; p.N = N; p.S = S; p.A = A; p.C = num_classes; p.M_total = M_total;
    p.box_offset = box_offset; p.cand_cap = cand_cap; p.iou_aware = iou_aware ? 1 : 0; p.clip = clip_bbox ? 1 : 0;
    for (int i = 0; i < 2 * A; ++i) p.anchors[i] = h_anchors_px[i];
    p.stride_f = (float)downsample;
    // fp32 constants exactly as Python/torch derive them from doubles (reference head.py:40, :125)
    p.sxy = (float)scale_x_y;
    p.sxy_bias = (float)((scale_x_y - 1.0) * 0.5);
    p.e_obj = (float)(1.0 - iou_aware_factor);
    p.e_iou = (float)iou_aware_factor;
    p.thr = score_threshold;
    *lds_out = ((size_t)DEC_CELLS * nch + 2 * DEC_CELLS * A) * sizeof(float);
    if (*lds_out > 96 * 1024) return PPY_ERR_UNSUPPORTED;
    return PPY_OK;
}

// Test / experiment hook (tests/test_gpu_ops.py compares the two decode kernels in one process; tools/decode_bench.py): staged /
// per_wave = -1 keeps the environment's choice, >= 0 overrides it; abl (results deliberately wrong: 1 = no flush atomic, 2 = no
// pair phase, 4 = no sweep) only acts in a -DPPY_DECODE_ABLATE build.
static std::atomic<int> g_dec_staged{-1}, g_dec_per_wave{-1}, g_dec_abl{0};
extern "C" void ppy_debug_decode_mode(int staged, int per_wave, int abl) {
    g_dec_staged.store(staged);
    g_dec_per_wave.store(per_wave);
    g_dec_abl.store(abl);
}

template <typename Kern>
static int decode_attr(PpyLdsAttr &st, Kern k) {
    return ppy_lds_attr(st, reinterpret_cast<const void *>(k), 96 * 1024);
}

extern "C" int ppy_yolo_decode_f32(const float *head_out, int head_ld, int N, int S, int A, int num_classes,
                                   const float *h_anchors_px, int downsample, double scale_x_y, int iou_aware,
                                   double iou_aware_factor, int clip_bbox, const float *im_size, float *boxes,
                                   int M_total, int box_offset, float score_threshold, uint32_t *cand_key,
                                   uint32_t *cand_idx, int *cand_count, int cand_cap, float *scores_dense,
                                   void *stream) {
    ppy_drop_stale_error();
    DecodeArgs p;
    size_t lds;
    int rc = decode_pack(p, head_out, head_ld, N, S, A, num_classes, h_anchors_px, downsample, scale_x_y, iou_aware,
                         iou_aware_factor, clip_bbox, im_size, boxes, M_total, box_offset, score_threshold, cand_key,
                         cand_idx, cand_count, cand_cap, scores_dense, &lds);
    if (rc != PPY_OK) return rc;
    static PpyLdsAttr attr;
    if (decode_attr(attr, yolo_decode_kernel) != PPY_OK) return PPY_ERR_LAUNCH;
    hipLaunchKernelGGL(yolo_decode_kernel, dim3((unsigned)((S * S + DEC_CELLS - 1) / DEC_CELLS), N), dim3(256), lds,
                       (hipStream_t)stream, p);
    return ppy_launch_status();
}

extern "C" int ppy_yolo_decode_levels_f32(int nlevels, const float *const *head_out, const int *head_ld, const int *S,
                                          const int *downsample, const float *const *h_anchors_px,
                                          const int *box_offset, int N, int A, int num_classes, double scale_x_y,
                                          int iou_aware, double iou_aware_factor, int clip_bbox, const float *im_size,
                                          float *boxes, int M_total, float score_threshold, uint32_t *cand_key,
                                          uint32_t *cand_idx, int *cand_count, int cand_cap, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(nlevels >= 1 && nlevels <= DEC_MAX_LEVELS && head_out && head_ld && S && downsample && h_anchors_px &&
                  box_offset);
    DecodeMulti m;
    m.nlevels = nlevels;
    size_t lds = 0;
    unsigned blocks = 0;
    for (int l = 0; l < nlevels; ++l) {
        size_t one;
        int rc = decode_pack(m.lv[l], head_out[l], head_ld[l], N, S[l], A, num_classes, h_anchors_px[l], downsample[l],
                             scale_x_y, iou_aware, iou_aware_factor, clip_bbox, im_size, boxes, M_total, box_offset[l],
                             score_threshold, cand_key, cand_idx, cand_count, cand_cap, nullptr, &one);
        if (rc != PPY_OK) return rc;
        lds = one > lds ? one : lds;
        m.nb[l] = (S[l] * S[l] + DEC_CELLS - 1) / DEC_CELLS;
        blocks += (unsigned)m.nb[l];
    }
    // the streaming kernel: compiled for the two head layouts of the configurations (3 anchors, 80 classes, with / without the
    // IoU-aware channels); rows 16-byte aligned with a pixel stride that is a multiple of 4 floats (what the plan produces:
    // 258 channels in rows of 260); anything else takes the staged kernel
    // experiment switches, read ONCE (not per launch): PPY_DECODE_STAGED = the staged kernel; PPY_DECODE_PER_WAVE = groups a wave takes
    // in turn.  The ablation switch PPY_DECODE_ABL (deliberately wrong results) exists in -DPPY_DECODE_ABLATE builds only.
    static const bool env_staged0 = getenv("PPY_DECODE_STAGED") != nullptr;
    static const int env_per_wave0 = getenv("PPY_DECODE_PER_WAVE") ? atoi(getenv("PPY_DECODE_PER_WAVE")) : 0;
    const int m_staged = g_dec_staged.load(), m_per_wave = g_dec_per_wave.load();       // ppy_debug_decode_mode overrides the environment
    const bool env_staged = m_staged >= 0 ? m_staged != 0 : env_staged0;
    const int env_per_wave = m_per_wave >= 0 ? m_per_wave : env_per_wave0;
#ifdef PPY_DECODE_ABLATE
    const int env_abl = g_dec_abl.load();
#else
    const int env_abl = 0;
#endif
    bool stream_ok = !env_staged && A == 3 && num_classes == 80;
    for (int l = 0; l < nlevels && stream_ok; ++l)
        stream_ok = (head_ld[l] & 3) == 0 && head_ld[l] >= (A * (5 + num_classes) + (iou_aware ? A : 0) + 3) / 4 * 4 &&
                    (((uintptr_t)head_out[l]) & 15) == 0 &&      // (whole 16-byte groups of a row are read, pad channels included)
                    (long long)N * S[l] * S[l] * head_ld[l] * 4 < 0x7FFFF000LL;      // (32-bit buffer offsets with an out-of-range sentinel)
    if (stream_ok) {
        DecodeStream ds;
        ds.nlevels = nlevels;
        long long groups_total = 0;
        for (int l = 0; l < nlevels; ++l) groups_total += (S[l] * S[l] + DS_GC - 1) / DS_GC;
        // ONE group per wave while the whole launch fits on the chip at once (R50vd-608, 8 images: 3790 groups = 948 workgroups on
        // 1024 slots of 4 waves): a wave is a latency chain load -> decode_pair -> sweep -> flush of ~8 us, and waves that take
        // several groups in turn run those chains back to back (36 us measured with ~2 groups per wave, ~2048 waves)
        static const int n_cu = [] {
            int dev = 0, cus = 256;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
            return cus;
        }();
        const long long wave_slots = (long long)n_cu * 2 * DS_WAVES;      // waves resident at once (two workgroups of eight per CU)
        long long per_wave = (groups_total * N + wave_slots - 1) / wave_slots;       // groups a wave takes in turn
        if (env_per_wave > 0) per_wave = env_per_wave;
        ds.abl = env_abl;
        unsigned nblocks = 0;
        for (int l = 0; l < nlevels; ++l) {
            const long long gl = (S[l] * S[l] + DS_GC - 1) / DS_GC;
            long long nb = (gl + DS_WAVES * per_wave - 1) / (DS_WAVES * per_wave);
            if (nb < 1) nb = 1;
            ds.lv[l] = m.lv[l];
            ds.nb[l] = (int)nb;
            nblocks += (unsigned)nb;
        }
        if (iou_aware)
            hipLaunchKernelGGL((yolo_decode_stream_kernel<3, 80, true>), dim3(nblocks, N), dim3(64 * DS_WAVES), 0, (hipStream_t)stream, ds);
        else
            hipLaunchKernelGGL((yolo_decode_stream_kernel<3, 80, false>), dim3(nblocks, N), dim3(64 * DS_WAVES), 0, (hipStream_t)stream, ds);
        return ppy_launch_status();
    }
    static PpyLdsAttr attr;
    if (decode_attr(attr, yolo_decode_multi_kernel) != PPY_OK) return PPY_ERR_LAUNCH;
    hipLaunchKernelGGL(yolo_decode_multi_kernel, dim3(blocks, N), dim3(256), lds, (hipStream_t)stream, m);
    return ppy_launch_status();
}

extern "C" int ppy_nms_candidates_f32(const float *scores, int N, int M, int C, float score_threshold,
                                      uint32_t *cand_key, uint32_t *cand_idx, int *cand_count, int cand_cap,
                                      void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(scores && cand_key && cand_idx && cand_count && N > 0 && M > 0 && C > 0 && cand_cap > 0);
    PPY_CHECK_ARG((long long)M * C < (1ll << 31));
    const long long total = (long long)M * C;
    long long gx = (total + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(dense_candidates_kernel, dim3((unsigned)gx, N), dim3(256), 0, (hipStream_t)stream, scores,
                       M, C, score_threshold, cand_key, cand_idx, cand_count, cand_cap);
    return ppy_launch_status();
}

extern "C" size_t ppy_matrix_nms_workspace_bytes(int N) {
    return N > 0 ? (size_t)N * (sizeof(NmsWs) + sizeof(NmsBig) + (size_t)2 * NMS_CCAP2 * sizeof(uint32_t)) : 0;
}

extern "C" int ppy_matrix_nms_f32(const float *boxes, int M_total, int num_classes, const uint32_t *cand_key,
                                  const uint32_t *cand_idx, const int *cand_count, int cand_cap, int N,
                                  float post_threshold, int nms_top_k, int keep_top_k, int use_gaussian,
                                  float gaussian_sigma, float *out_dets, int *out_count, int *out_keep_idx,
                                  void *ws, size_t ws_bytes, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(boxes && cand_key && cand_idx && cand_count && out_dets && out_count && out_keep_idx);
    PPY_CHECK_ARG(N > 0 && M_total > 0 && num_classes > 0 && cand_cap > 0);
    PPY_CHECK_ARG(((uintptr_t)boxes & 15) == 0);
    if (!ws || ((uintptr_t)ws & 15) != 0 || ws_bytes < ppy_matrix_nms_workspace_bytes(N)) return PPY_ERR_WORKSPACE;
    if (nms_top_k < 1 || nms_top_k > KMAX || keep_top_k < 1 || keep_top_k > nms_top_k) return PPY_ERR_UNSUPPORTED;
    const long long span = (long long)M_total * num_classes;
    PPY_CHECK_ARG(span < (1ll << 31));
    int ib = 1;
    while ((1ll << ib) < span) ++ib;
    NmsArgs p;
    p.boxes = boxes; p.cand_key = cand_key; p.cand_idx = cand_idx; p.cand_count = cand_count;
    p.out_dets = out_dets; p.out_count = out_count; p.out_keep = out_keep_idx;
    p.M_total = M_total; p.C = num_classes; p.cand_cap = cand_cap; p.top_k = nms_top_k; p.keep_k = keep_top_k;
    p.gaussian = use_gaussian ? 1 : 0; p.idx_bits = ib; p.post_thr = post_threshold; p.sigma = gaussian_sigma;
    p.ws = (char *)ws;
    p.N = N;
    static PpyLdsAttr attr, attr_s;
    if (ppy_lds_attr(attr, reinterpret_cast<const void *>(nms_select_kernel), CCAP * 8) != PPY_OK) return PPY_ERR_LAUNCH;
    if (ppy_lds_attr(attr_s, reinterpret_cast<const void *>(nms_sample_kernel), NMS_SAMPLE * 8) != PPY_OK) return PPY_ERR_LAUNCH;
    hipStream_t st = (hipStream_t)stream;
    // a list can only be large if the caller's capacity is: the two chip-wide steps are not even launched otherwise
    if (cand_cap > CCAP) {
        hipLaunchKernelGGL(nms_sample_kernel, dim3(N), dim3(NT), NMS_SAMPLE * 8, st, p);
        hipLaunchKernelGGL(nms_collect_kernel, dim3(NMS_CG, N), dim3(512), 0, st, p);
    }
    hipLaunchKernelGGL(nms_select_kernel, dim3(N), dim3(NT), CCAP * 8, st, p);
    hipLaunchKernelGGL(nms_colmax_kernel, dim3(NMS_G, N), dim3(NT), 0, st, p);
    hipLaunchKernelGGL(nms_decay_kernel, dim3(NMS_G, N), dim3(NT), 0, st, p);
    hipLaunchKernelGGL(nms_finish_kernel, dim3(N), dim3(NT), 0, st, p);
    return ppy_launch_status();
}
