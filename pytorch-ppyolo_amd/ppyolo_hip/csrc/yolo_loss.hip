// YOLOv3Loss forward + backward for one head level in ONE kernel (BASELINE config 5):
// the reference's YOLOv3Loss._get_fine_grained_loss (model/losses.py:121-253) with IouLoss / IouAwareLoss
// (model/iou_losses.py:39-246) and the ignore mask of _calc_obj_loss (losses.py:296-356), and -- instead of an autograd
// graph of ~150 ATen ops per level -- the analytic gradient d(sum of all loss terms) / d(head output), which is what
// train.py:441 `all_loss.backward()` delivers to the head's last convolutions.
//
// One thread per (image, anchor, grid cell): it reads its 5 + C (+1 IoU) logits from the NHWC head output, its target
// column from the reference's target layout [N, an, 6 + C, S, S] and the image's ground-truth boxes, and writes its slice
// of dout plus six loss contributions (reduced afterwards in a fixed order).  HBM-bound by design: every input is read
// once (the head outputs are a few MB per level).
// Round 3: the head output is channel-minor (a cell's 258 logits are one 1 KB row) and the targets are cell-minor, so a
// thread per cell read and wrote its row with addresses 1 KB apart -- 64 cache lines per wave instruction, 400 us for the
// 76x76 level.  Now a workgroup owns LOSS_CELLS consecutive cells of one image: their rows -- one contiguous piece of HBM --
// are copied into LDS (row pitch odd: a column access by consecutive cells is conflict-free), the pair threads work in LDS
// and overwrite every logit with its gradient in place, the rows go back as one contiguous stream; the six loss terms of
// the workgroup are reduced in LDS in a fixed order and a last one-workgroup kernel adds the workgroups' sums.  Faithful to the reference's arithmetic, including
//   * the IoU-aware term's reduction over grid x before the multiplication with tobj (iou_losses.py:241-242):
//     loss = sum_h (sum_w tobj[h,w]) * (sum_w' iou[h,w'] * -log(ioup[h,w'] + 1e-9)),
//   * `+ 1e-9` INSIDE the logarithms, `+ 1e-10` in the IoU union of the IoU losses and none in the ignore-mask IoU,
//   * |.| gradients with sign(0) = 0, min / max ties sharing the gradient, clamp(min=0) passing it at 0 (torch autograd).
#include "common.h"

namespace {

struct LossArgs {
    const float *out, *target, *gt;
    float *dout, *part;
    float *amax_dout;      // per-image max|dout| slots (operand scale of the f16x2 gradient kernels of the output convolution) or NULL
    int out_ld, dout_ld;
    int N, S, an, C, G, iou_aware;
    float aw[4], ah[4];
    float downsample, scale_x_y, ignore_thresh, w_iou, w_iou_aware, inv_n;
    int loss_square;          // IouLoss(loss_square=): 1 - iou^2 (both PP-YOLO configurations), or 1 - iou (reference iou_losses.py:66-70)
};

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }
// d min(a, b) / d a  and  d max(a, b) / d a  as torch's elementwise min / max backward: ties share
__device__ __forceinline__ float dmin_a(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float dmax_a(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }

constexpr int LOSS_CELLS = 64;          // cells per workgroup = lanes of a wave
constexpr int LOSS_MAXAN = 4;
// Waves of a workgroup: (anchor a, part q), q = 0: box, objectness, ignore mask and the first C / 10 classes; q = 1 .. 3: a third
// of the remaining classes each (the class loop -- exp, two logs, three divisions per class -- is most of the arithmetic, and
// one wave per pair left a CU with 6 waves: 124 us per level on average; 12 waves per workgroup, two workgroups per CU).
constexpr int LOSS_Q = 4;

__global__ void __launch_bounds__(LOSS_CELLS * LOSS_MAXAN * LOSS_Q) yolo_loss_kernel(const LossArgs p, int blocks_per_image, int pitch) {
    extern __shared__ float s_rows[];                       // [LOSS_CELLS][pitch] logits -> gradients, then [LOSS_MAXAN][LOSS_CELLS] row sums of tobj
    __shared__ float s_red[LOSS_MAXAN * LOSS_Q][6];
    const int nt = (int)blockDim.x;
    const int cells = p.S * p.S;
    const int n = blockIdx.x / blocks_per_image;
    const int cell0 = (blockIdx.x - n * blocks_per_image) * LOSS_CELLS;
    const int ncell = min(LOSS_CELLS, cells - cell0);
    const int nch = p.an * (5 + p.C) + (p.iou_aware ? p.an : 0);
    float *s_T = s_rows + LOSS_CELLS * pitch;
    // ---- the rows of this workgroup's cells: one contiguous piece of the head output (4-byte accesses, 256 B per wave)
    {
        const float *src = p.out + ((long long)n * cells + cell0) * p.out_ld;
        const int total = ncell * p.out_ld;
        int row = threadIdx.x / p.out_ld, col = threadIdx.x - row * p.out_ld;
        const int drow = nt / p.out_ld, dcol = nt - drow * p.out_ld;
        for (int i = threadIdx.x; i < total; i += nt) {
            if (col < nch) s_rows[row * pitch + col] = src[i];
            row += drow;
            col += dcol;
            if (col >= p.out_ld) { col -= p.out_ld; ++row; }
        }
    }
    // ---- sum over grid x of tobj for the (anchor, row) pairs this workgroup touches (the IoU-aware term's broadcast):
    // one thread per pair, ascending x like the reference's sum
    const int h_first = cell0 / p.S, h_last = (cell0 + ncell - 1) / p.S;
    if (p.iou_aware) {
        const int nrows = h_last - h_first + 1;
        for (int j = threadIdx.x; j < nrows * p.an; j += nt) {
            const int a = j / nrows, hh = h_first + (j - a * nrows);
            const float *trow = p.target + ((long long)(n * p.an + a) * (6 + p.C) + 5) * cells + hh * p.S;
            float T = 0.f;
            for (int q = 0; q < p.S; ++q) T += trow[q];
            s_T[a * LOSS_CELLS + (hh - h_first)] = T;
        }
    }
    __syncthreads();
    const int grp = threadIdx.x / LOSS_CELLS, lc = threadIdx.x - grp * LOSS_CELLS;
    const int qq = grp / p.an, a = grp - qq * p.an;          // (wave-uniform)
    const bool active = lc < ncell;
    float dmax = 0.f;
    float l6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
    const int cell = cell0 + lc;
    const int h = cell / p.S, w = cell - h * p.S;
    const float S = (float)p.S;
    float *o = s_rows + lc * pitch;
    float *d = o;                                          // gradients overwrite the logits in place (each logit has ONE reader: its pair)
    auto put = [&](int idx, float v) {
        d[idx] = v;
        dmax = fmaxf(dmax, fabsf(v));
    };
    const int base = (p.iou_aware ? p.an : 0) + a * (5 + p.C);
    const float *t = p.target + ((long long)(n * p.an + a) * (6 + p.C)) * cells + cell;
    const float tobj = t[5 * cells];
    if (qq == 0) {
    const float tx = t[0], ty = t[cells], tw = t[2 * cells], th = t[3 * cells], tscale = t[4 * cells];
    const float ts = tscale * tobj;
    const float x = o[base], y = o[base + 1], lw = o[base + 2], lh = o[base + 3], obj = o[base + 4];
    const float sx = sigm(x), sy = sigm(y);
    const float sxy = p.scale_x_y;
    float l_xy, g_x, g_y;
    float px, py;                                          // decoded centre offset inside the cell
    if (fabsf(sxy - 1.0f) < 1e-10f) {                      // plain YOLOv3: binary cross-entropy on sigmoid(x)
        px = sx; py = sy;
        l_xy = (tx * (0.f - logf(sx + 1e-9f)) + (1.f - tx) * (0.f - logf(1.f - sx + 1e-9f))) * ts +
               (ty * (0.f - logf(sy + 1e-9f)) + (1.f - ty) * (0.f - logf(1.f - sy + 1e-9f))) * ts;
        g_x = ts * (-tx / (sx + 1e-9f) + (1.f - tx) / (1.f - sx + 1e-9f)) * sx * (1.f - sx);
        g_y = ts * (-ty / (sy + 1e-9f) + (1.f - ty) / (1.f - sy + 1e-9f)) * sy * (1.f - sy);
    } else {                                               // Grid Sensitive: L1 on the decoded offset
        px = sxy * sx - 0.5f * (sxy - 1.0f);
        py = sxy * sy - 0.5f * (sxy - 1.0f);
        l_xy = fabsf(px - tx) * ts + fabsf(py - ty) * ts;
        g_x = sgn(px - tx) * ts * sxy * sx * (1.f - sx);
        g_y = sgn(py - ty) * ts * sxy * sy * (1.f - sy);
    }
    const float l_wh = fabsf(lw - tw) * ts + fabsf(lh - th) * ts;
    float g_w = sgn(lw - tw) * ts, g_h = sgn(lh - th) * ts;

    // ---- IoU of the decoded box with the target box (iou_losses.py:135-190 -> :74-96), all in units of the image side
    const float aw = p.aw[a], ah = p.ah[a], den = S * p.downsample;
    const float cx = (px + (float)w) / S, cy = (py + (float)h) / S;
    const float pw = (expf(lw) * aw) / den, ph = (expf(lh) * ah) / den;
    const float x1 = cx - 0.5f * pw, y1 = cy - 0.5f * ph, x2r = cx + 0.5f * pw, y2r = cy + 0.5f * ph;
    const float cxg = (tx + (float)w) / S, cyg = (ty + (float)h) / S;
    const float pwg = (expf(tw) * aw) / den, phg = (expf(th) * ah) / den;
    const float x1g = cxg - 0.5f * pwg, y1g = cyg - 0.5f * phg, x2g = cxg + 0.5f * pwg, y2g = cyg + 0.5f * phg;
    const float x2 = fmaxf(x1, x2r), y2 = fmaxf(y1, y2r);
    const float m_x2 = dmax_a(x2r, x1), m_y2 = dmax_a(y2r, y1);          // share of x2r in x2 = max(x1, x2r) (1 unless degenerate)
    const float iw_raw = fminf(x2, x2g) - fmaxf(x1, x1g), ih_raw = fminf(y2, y2g) - fmaxf(y1, y1g);
    const float iw = fmaxf(iw_raw, 0.f), ih = fmaxf(ih_raw, 0.f);
    const float inter = iw * ih;
    const float uni = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter + 1e-10f;
    const float k = inter / uni;
    const float l_iou = (p.loss_square ? 1.f - k * k : 1.f - k) * p.w_iou * ts;
    float dk = (p.loss_square ? -2.f * k : -1.f) * p.w_iou * ts;         // d(all losses) / d k, before the batch mean
    float l_ia = 0.f, g_ioup = 0.f;
    if (p.iou_aware) {
        const float T = s_T[a * LOSS_CELLS + (h - h_first)];              // sum over grid x of tobj: the reference's broadcast
        const float ip = sigm(o[a]);
        const float ce = 0.f - logf(ip + 1e-9f);
        l_ia = T * k * ce * p.w_iou_aware;
        dk += T * ce * p.w_iou_aware;
        g_ioup = T * k * p.w_iou_aware * (-1.f / (ip + 1e-9f)) * ip * (1.f - ip);
    }
    if (dk != 0.f) {
        const float cw = iw_raw >= 0.f ? 1.f : 0.f, chh = ih_raw >= 0.f ? 1.f : 0.f;          // clamp(min=0) backward
        // d iw / d x2, d iw / d x1 (through x2 = max(x1, x2r): x1 also reaches iw via x2 when degenerate)
        const float diw_dx2 = cw * dmin_a(x2, x2g), diw_dx1 = -cw * dmax_a(x1, x1g);
        const float dih_dy2 = chh * dmin_a(y2, y2g), dih_dy1 = -chh * dmax_a(y1, y1g);
        const float di_dx2 = ih * diw_dx2, di_dx1 = ih * diw_dx1, di_dy2 = iw * dih_dy2, di_dy1 = iw * dih_dy1;
        const float du_dx2 = (y2 - y1) - di_dx2, du_dx1 = -(y2 - y1) - di_dx1;
        const float du_dy2 = (x2 - x1) - di_dy2, du_dy1 = -(x2 - x1) - di_dy1;
        const float iu2 = 1.0f / (uni * uni);
        float dk_dx2 = (di_dx2 * uni - inter * du_dx2) * iu2, dk_dx1 = (di_dx1 * uni - inter * du_dx1) * iu2;
        float dk_dy2 = (di_dy2 * uni - inter * du_dy2) * iu2, dk_dy1 = (di_dy1 * uni - inter * du_dy1) * iu2;
        // x2 = max(x1, x2r): hand x2's gradient to x2r / x1 by their shares
        const float dk_dx2r = dk_dx2 * m_x2, dk_dy2r = dk_dy2 * m_y2;
        dk_dx1 += dk_dx2 * (1.f - m_x2);
        dk_dy1 += dk_dy2 * (1.f - m_y2);
        const float dk_dcx = dk_dx1 + dk_dx2r, dk_dpw = 0.5f * (dk_dx2r - dk_dx1);
        const float dk_dcy = dk_dy1 + dk_dy2r, dk_dph = 0.5f * (dk_dy2r - dk_dy1);
        const float dpx_dx = (fabsf(sxy - 1.0f) < 1e-10f ? 1.f : sxy) * sx * (1.f - sx);
        const float dpy_dy = (fabsf(sxy - 1.0f) < 1e-10f ? 1.f : sxy) * sy * (1.f - sy);
        g_x += dk * dk_dcx * dpx_dx / S;
        g_y += dk * dk_dcy * dpy_dy / S;
        g_w += dk * dk_dpw * pw;
        g_h += dk * dk_dph * ph;
    }

    // ---- objectness with the ignore mask (losses.py:296-356): boxes as paddle_yolo_box decodes them (losses.py:22-83)
    const float bx = (sxy * sx + (float)w - (sxy - 1.0f) * 0.5f) * p.downsample, by = (sxy * sy + (float)h - (sxy - 1.0f) * 0.5f) * p.downsample;
    const float bw = expf(lw) * aw, bh = expf(lh) * ah;
    const float q0 = (bx - bw / 2) / S / p.downsample, q1 = (by - bh / 2) / S / p.downsample;
    const float q2 = (bx + bw / 2) / S / p.downsample, q3 = (by + bh / 2) / S / p.downsample;
    const float area_a = (q2 - q0) * (q3 - q1);
    float best = -__builtin_huge_valf();
    const float *gt = p.gt + (long long)n * p.G * 4;
    for (int g = 0; g < p.G; ++g) {
        const float gx = gt[4 * g], gy = gt[4 * g + 1], gw = gt[4 * g + 2], gh = gt[4 * g + 3];
        const float g0 = gx - gw / 2.f, g1 = gy - gh / 2.f, g2 = gx + gw / 2.f, g3 = gy + gh / 2.f;
        const float ww = fmaxf(fminf(q2, g2) - fmaxf(q0, g0), 0.f), hh = fmaxf(fminf(q3, g3) - fmaxf(q1, g1), 0.f);
        const float it = ww * hh;
        const float iou = it / (area_a + (g2 - g0) * (g3 - g1) - it);
        best = (iou > best || iou != iou) ? iou : best;                     // torch.max propagates NaN
    }
    const float iou_mask = best <= p.ignore_thresh ? 1.f : 0.f;
    const float noobj = (1.0f - (tobj > 0.f ? 1.f : 0.f)) * iou_mask;
    const float so = sigm(obj);
    const float l_obj = tobj * (0.f - logf(so + 1e-9f)) + noobj * (0.f - logf(1.f - so + 1e-9f));
    const float g_obj = (-tobj / (so + 1e-9f) + noobj / (1.f - so + 1e-9f)) * so * (1.f - so);

    put(base, g_x * p.inv_n);
    put(base + 1, g_y * p.inv_n);
    put(base + 2, g_w * p.inv_n);
    put(base + 3, g_h * p.inv_n);
    put(base + 4, g_obj * p.inv_n);
    if (p.iou_aware) put(a, g_ioup * p.inv_n);
    l6[0] = l_xy; l6[1] = l_wh; l6[2] = l_obj; l6[4] = l_iou; l6[5] = l_ia;
    }

    // ---- classification: this wave's share of the classes
    const int n0 = p.C / 10, per = (p.C - n0 + LOSS_Q - 2) / (LOSS_Q - 1);
    const int c_lo = qq == 0 ? 0 : min(p.C, n0 + (qq - 1) * per), c_hi = qq == 0 ? n0 : min(p.C, n0 + qq * per);
    float l_cls = 0.f;
    for (int c = c_lo; c < c_hi; ++c) {
        const float sc = sigm(o[base + 5 + c]), tc = t[(long long)(6 + c) * cells];
        l_cls += tc * (0.f - logf(sc + 1e-9f)) + (1.f - tc) * (0.f - logf(1.f - sc + 1e-9f));
        put(base + 5 + c, tobj * (-tc / (sc + 1e-9f) + (1.f - tc) / (1.f - sc + 1e-9f)) * sc * (1.f - sc) * p.inv_n);
    }
    l6[3] = l_cls * tobj;
    }
    if (p.amax_dout) amax_track(dmax, n, p.amax_dout, blockIdx.x * 4 + (threadIdx.x >> 6));
    // ---- the six sums: a shuffle tree inside every wave, then the waves in index order (fixed order: run-to-run identical)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float v = l6[j];
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_xor(v, o2);
        if ((threadIdx.x & 63) == 0) s_red[grp][j] = v;
    }
    __syncthreads();
    // ---- the gradient rows back as one contiguous stream
    {
        float *dst = p.dout + ((long long)n * cells + cell0) * p.dout_ld;
        const int total = ncell * p.dout_ld;
        int row = threadIdx.x / p.dout_ld, col = threadIdx.x - row * p.dout_ld;
        const int drow = nt / p.dout_ld, dcol = nt - drow * p.dout_ld;
        for (int i = threadIdx.x; i < total; i += nt) {
            if (col < nch) dst[i] = s_rows[row * pitch + col];
            row += drow;
            col += dcol;
            if (col >= p.dout_ld) { col -= p.dout_ld; ++row; }
        }
    }
    if (threadIdx.x < 6) {
        float v = 0.f;
        for (int g2 = 0; g2 < p.an * LOSS_Q; ++g2) v += s_red[g2][threadIdx.x];
        p.part[(long long)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// loss[j] = inv_n * sum over the workgroups' sums, fixed order: one workgroup, strided partial sums, tree
__global__ void __launch_bounds__(256) loss_reduce_kernel(const float *part, long long cells, float inv_n, float *loss, int accumulate) {
    __shared__ float red[6][256];
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long i = threadIdx.x; i < cells; i += 256)
#pragma unroll
        for (int j = 0; j < 6; ++j) s[j] += part[i * 6 + j];
#pragma unroll
    for (int j = 0; j < 6; ++j) red[j][threadIdx.x] = s[j];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int j = 0; j < 6; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 6) loss[threadIdx.x] = (accumulate ? loss[threadIdx.x] : 0.f) + red[threadIdx.x][0] * inv_n;
}

}  // namespace

extern "C" size_t ppy_yolov3_loss_workspace_bytes(int N, int S, int an) { return (size_t)N * an * S * S * 6 * sizeof(float); }

extern "C" int ppy_yolov3_loss_f32(const float *head_out, int out_ld, const float *target, const float *gt_box, int num_gt,
                                   const float *h_anchors_px, int an, int num_classes, int N, int S, int downsample, double scale_x_y,
                                   double ignore_thresh, double iou_loss_weight, int iou_loss_square, int iou_aware,
                                   double iou_aware_loss_weight, float *dout, int dout_ld, float *loss6, int accumulate, float *amax_dout,
                                   void *ws, size_t ws_bytes, void *stream) {
    ppy_drop_stale_error();
    PPY_CHECK_ARG(head_out && target && gt_box && h_anchors_px && dout && loss6 && N > 0 && S > 0 && an > 0 && an <= 4 && num_classes > 0);
    const int nch = an * (5 + num_classes) + (iou_aware ? an : 0);
    PPY_CHECK_ARG(out_ld >= nch && dout_ld >= nch && num_gt >= 0 && downsample > 0);
    if (!ws || ws_bytes < ppy_yolov3_loss_workspace_bytes(N, S, an)) return PPY_ERR_WORKSPACE;
    LossArgs p;
    p.out = head_out; p.target = target; p.gt = gt_box; p.dout = dout; p.part = (float *)ws; p.amax_dout = amax_dout;
    p.out_ld = out_ld; p.dout_ld = dout_ld; p.N = N; p.S = S; p.an = an; p.C = num_classes; p.G = num_gt; p.iou_aware = iou_aware ? 1 : 0;
    for (int a = 0; a < an; ++a) {
        p.aw[a] = h_anchors_px[2 * a];
        p.ah[a] = h_anchors_px[2 * a + 1];
    }
    p.downsample = (float)downsample; p.scale_x_y = (float)scale_x_y; p.ignore_thresh = (float)ignore_thresh;
    p.loss_square = iou_loss_square ? 1 : 0;
    p.w_iou = (float)iou_loss_weight; p.w_iou_aware = (float)iou_aware_loss_weight; p.inv_n = 1.0f / (float)N;
    // (a row must fit the staging: C up to a few hundred classes; the pair threads are LOSS_CELLS x an <= 256)
    const int pitch = nch | 1;
    const int lds = (LOSS_CELLS * pitch + LOSS_MAXAN * LOSS_CELLS) * (int)sizeof(float);
    PPY_CHECK_ARG(lds <= 150 * 1024);
    static PpyLdsAttr attr;
    if (ppy_lds_attr(attr, (const void *)yolo_loss_kernel, lds) != PPY_OK) return PPY_ERR_LAUNCH;
    const int blocks_per_image = (S * S + LOSS_CELLS - 1) / LOSS_CELLS;
    const long long blocks = (long long)N * blocks_per_image;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(yolo_loss_kernel, dim3((unsigned)blocks), dim3(LOSS_CELLS * an * LOSS_Q), lds, st, p, blocks_per_image, pitch);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, (const float *)ws, blocks, p.inv_n, loss6, accumulate);
    return ppy_launch_status();
}
