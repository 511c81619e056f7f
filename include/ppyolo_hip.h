/*
 * ppyolo_hip.h -- C ABI of libppyolo_hip.so: MI355X (gfx950) kernels for the PP-YOLO
 * inference hot path of miemie2013/Pytorch-PPYOLO.
 *
 * The reference has NO FFI on this path (it is a torch.nn.Module tree); the entry points
 * below are what a binding for this path would bind, one per reference operator that the
 * path replaces.  Each declaration cites the reference code it stands in for
 * (file:line inside the reference repo).  The only real FFI in the reference is the
 * optional, CUDA-only `_ext.dcn_v2_forward` (external/DCNv2/src/dcn_v2.h:9-23); the DCN
 * entry points mirror its argument order.
 *
 * Conventions (all functions):
 *   - plain C types only; every pointer is a DEVICE pointer unless named `h_*`;
 *   - activations are fp32 NHWC: element (n,h,w,c) of a tensor with pixel stride `ld`
 *     (in floats, >= C, multiple of 4) lives at  base[((n*H + h)*W + w)*ld + c];
 *     a channel slice of a wider (concat) buffer is expressed as base+offset with the
 *     wide buffer's `ld`;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only
 *     enqueue work, never synchronise, never allocate, keep no state => thread-safe per
 *     stream and capturable into a hipGraph;
 *   - return value: 0 = PPY_OK, negative = error (ppy_error_string()); nothing throws.
 */
#ifndef PPYOLO_HIP_H_
#define PPYOLO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPY_OK 0
#define PPY_ERR_BAD_ARG (-1)      /* shape / alignment / NULL-pointer contract violated   */
#define PPY_ERR_UNSUPPORTED (-2)  /* valid request outside what the kernels implement     */
#define PPY_ERR_WORKSPACE (-3)    /* workspace missing or too small                       */
#define PPY_ERR_LAUNCH (-4)       /* hipLaunch / hipFuncSetAttribute failed               */

#define PPY_ACT_NONE 0
#define PPY_ACT_RELU 1
#define PPY_ACT_LEAKY 2 /* LeakyReLU(0.1), reference model/custom_layers.py:133 */

int ppy_version(void);
const char *ppy_error_string(int code);
/* Name of the HIP runtime error behind the last PPY_ERR_LAUNCH on the calling thread (diagnostics). */
const char *ppy_last_hip_error(void);
void ppy_note_hip_error(int hip_error);      /* internal: set by the launch paths */

/* ------------------------------------------------------------------------------------
 * Conv2dUnit.forward (reference model/custom_layers.py:243-253): conv(k in {1,3},
 * pad=(k-1)/2) -> eval BatchNorm as per-channel affine -> [+ residual] -> activation.
 * Also covers `x + shortcut; relu` of ConvBlock/IdentityBlock/BasicBlock
 * (model/resnet_vd.py:55-56, :85-86, :265-266) through `residual`, the nearest x2
 * upsample of the head routes (model/head.py:396-397) through `upsample2x`, and the two
 * CoordConv channels (model/custom_layers.py:267-271) through `posbias`.
 *
 * Implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32):
 *   y[n,ho,wo,k] = act( (sum_{r,s,c} x[n,ho*stride+r-pad,wo*stride+s-pad,c]*w[k,r,s,c]
 *                         + posbias[ho,wo,k]) * scale[k] + shift[k] + residual[n,ho,wo,k] )
 * x: NHWC (ld x_ld), C % 32 == 0.   w: [K][R][S][C] ("KRSC").   scale/shift: [K].
 * residual: NULL or NHWC [N,Ho,Wo,>=K] with ld res_ld.   posbias: NULL or [Ho*Wo][K].
 * y: NHWC ld y_ld; with upsample2x != 0 y is [N,2Ho,2Wo,*] and every result is
 * written to its 2x2 nearest-neighbour block.
 * cfg: tile configuration id, -1 = built-in heuristic; splitk: 0 = heuristic, >=1 forced.
 * ws: scratch for split-K partial sums, ppy_conv2d_workspace_bytes() bytes (may be
 * NULL when that returns 0).
 * w_x3: NULL, or the same weights split into three bf16 planes by
 * ppy_conv2d_split_weights_bf16x3 ([3][K][R][S][C] bf16, 6*K*R*S*C bytes).  With w_x3 the
 * "bf16x3" kernels become selectable (cfg ids >= 31; the heuristic cfg = -1 then prefers
 * them): each fp32 operand is split exactly into 3 bf16 terms and the product is evaluated as
 * the 6 leading partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- fp32 in,
 * fp32 out, error vs fp64 at the level of the exact-fp32 fma chain (csrc/conv_x3.hip,
 * profiles/r01_bf16x3_numerics.txt), 6/16 of the fp32-MFMA cost.  Without w_x3 those ids
 * return PPY_ERR_BAD_ARG.
 * w_f16x2 / scale_f16x2: NULL, or the outputs of ppy_conv2d_split_weights_f16x2: the weights times a per-output-
 * channel power of two as two fp16 planes (4*K*R*S*C bytes; opaque layout [2][R*S*C/32][K][32]: the 32-deep reduction chunk is
 * the outer index, so that the 16 weight rows one DMA instruction fetches are one contiguous KB) and `scale` with the inverse of that
 * power folded in.  They make the "f16x2" kernels selectable (cfg ids >= 40; 9 tiles, ids 40-48 with two LDS stages,
 * 49-57 the same tiles with three and 58-66 with four chunks resident / in flight, as far as 160 KB of LDS allow):
 * 2-term fp16 split of both operands,
 * 3 partial products on v_mfma_f32_32x32x16_f16, fp32 accumulation; the activations of an image are scaled on the
 * fly by the power of two that puts that image's maximum into [2^13, 2^14), read from `amax_in`.  Error vs fp64 at the level of
 * the exact-fp32 fma chain (tools/probes/f16x2_probe.hip, profiles/r01_f16x2_numerics.txt).  Needs amax_in, and with a
 * posbias also posbias_f16x2 = posbias[., k] * scale[k] / scale_f16x2[k] (the bias map in the scaled-weight domain),
 * else PPY_ERR_BAD_ARG.
 * amax_in / amax_out: NULL, or N * PPY_AMAX_FLOATS_PER_IMAGE floats each: the running PER-IMAGE max|.| of the input
 * tensor (an upper bound is enough) and the slots into which this launch merges the per-image max|y| of what it stores
 * (atomic max; the owner zeroes the slots before the first producer of a tensor runs).  Image n owns 8 slots, one
 * 64-byte line apart: floats [(n*8 + s)*16], s = 0..7; a consumer takes the maximum over the 8 slots of an image.
 * Per-image scales keep every image's result independent of the rest of the batch.
 */
#define PPY_AMAX_FLOATS_PER_IMAGE 128 /* 8 slots x 16 floats */
int ppy_conv2d_split_weights_bf16x3(const float *w_krsc, long long n_elems, void *out_planes,
                                    void *stream);
int ppy_conv2d_split_weights_f16x2(const float *w_krsc, int K, long long kred, const float *scale,
                                   void *out_planes, float *scale_out, void *stream);
int ppy_conv2d_bn_act_f32(const float *x, int x_ld, const float *w_krsc, const void *w_x3,
                          const void *w_f16x2, const float *scale, const float *scale_f16x2,
                          const float *shift, const float *residual, int res_ld,
                          const float *posbias, const float *posbias_f16x2, float *y, int y_ld, int N,
                          int H, int W, int C, int K, int R, int S, int stride, int pad, int act,
                          int upsample2x, int cfg, int splitk, const float *amax_in, float *amax_out,
                          void *ws, size_t ws_bytes, void *stream);
/* The same launch with PRE-SPLIT tensors on one or both sides ("global pre-split"; f16x2 tile kernels of
 * csrc/conv_x3.hip / conv_ws.hip with an explicit cfg, one split; a pre-split y cannot be combined with upsample2x).  Between a producer convolution and the ONE
 * convolution that consumes its output the tensor can travel as the consumer's finished MFMA operands instead of fp32: per pixel
 * and 32-channel group, 32 fp16 first terms followed by 32 fp16 second terms of value * s_image (RNE; the residual is exact) -- the
 * same 128 bytes, the same pixel stride.  The consumer's main loop then carries no scale / split VALU work.
 *   y_split_scale != NULL: y is WRITTEN in that form (K and y_ld multiples of 32, y 128-byte aligned) and y_split_scale[n]
 *     receives the power of two s_n that puts the STATIC bound  y_bound_mul * max|x_n| + y_bound_add  of |y_n| into [2^13, 2^14)
 *     (caller's duty: y_bound_mul >= max_k |scale_k| * sum_c |w_k,c|, y_bound_add >= max_k |shift_k| (+ |posbias * scale|);
 *     max|x_n| is taken from amax_in, rounded up to a power of two); amax_out still receives max|y|.
 *   x_split_scale != NULL: x holds such operands, written by a launch whose y_split_scale it is (C and x_ld multiples of 32).
 * PPY_ERR_BAD_ARG when the chosen cfg cannot read / write such tensors (never a silent reinterpretation of the bytes).
 *   amax_in2 (round 6; may be NULL): a SECOND block of tracked per-image maxima, for an input whose channels were written by
 *     producers that track into different blocks (the folded projection shortcut's wide buffer [conv2 output | pooled block
 *     input], model/resnet_vd.py:27-33: the pooled part keeps the slots of the tensor it was pooled from, so that tensor's other
 *     readers never see conv2's maximum); the f16x2 kernels scale an image by the larger of the two.  With both split pointers
 *     NULL the call is ppy_conv2d_bn_act_f32 plus this one pointer (cfg -1 and split-K allowed).
 * Reference operator: the same Conv2dUnit.forward (model/custom_layers.py:243-253). */
int ppy_conv2d_bn_act_split_f32(const float *x, int x_ld, const float *w_krsc, const void *w_x3,
                                const void *w_f16x2, const float *scale, const float *scale_f16x2,
                                const float *shift, const float *residual, int res_ld,
                                const float *posbias, const float *posbias_f16x2, float *y, int y_ld, int N,
                                int H, int W, int C, int K, int R, int S, int stride, int pad, int act,
                                int upsample2x, int cfg, int splitk, const float *amax_in, float *amax_out,
                                void *ws, size_t ws_bytes, void *stream, const float *x_split_scale,
                                float *y_split_scale, float y_bound_mul, float y_bound_add, const float *amax_in2);
size_t ppy_conv2d_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride,
                                  int pad, int cfg, int splitk);
int ppy_conv2d_num_configs(void);
/* Map of the cfg ids (all families compute the same operator; the f16x2 families are bit-identical to each other):
 *   0-30   exact-fp32 MFMA tiles (csrc/conv_igemm.hip)            31-39  bf16x3 tiles (csrc/conv_x3.hip; need w_x3)
 *   40-66  f16x2 tiles, 9 shapes x {2, 3, 4} LDS stages           67-84  the same with slab reuse (3x3 / stride 1 / pad 1 only)
 *   85-93  f16x2 tiles of 96 / 192 rows x {2, 3, 4} stages        then, from the functions below:
 *   stream_first + {0, 1}   streaming 1x1 kernel (C = 64 / 128)   patch_first   patch kernel of the 3x3 stem layers (C = 32)
 *   ws_first + {0..15}      f16x2 tiles with specialised waves: 128x128 (3 / 4 stages), 64x128 (4 / 6), the three PRE variants,
 *                           256x128 with eight consumer waves (2 / 3 stages); round 6: + 9 / + 10 = 128x128 (3 / 4 stages), + 11 / + 12 = 64x128 (4 / 6 stages), + 13 = 128x128 and + 14 / + 15 = 64x128 with two
 *                           32-deep chunks per stage (one barrier per 64), all with eight
 *                           consumer waves as two "k-parity" groups (group g multiplies k-step g of every 32-deep chunk; the two
 *                           sums are added at the end -- one more fp32 rounding, so these seven are bit-identical to each other, not
 *                           to the other f16x2 tiles)
 *   small_first + {0..3}    wave-private tiles for small outputs (ppy_conv2d_small_first_config below)
 * An explicit id on a geometry its kernel does not cover returns PPY_ERR_BAD_ARG (never a silent other kernel); the one exception:
 * the scalar epilogue (K % 4 != 0 or unaligned rows) does not exist for the wave tiles of six / eight 32x32 blocks (40, 46, 85-93
 * and their deeper variants) -- such a launch runs on the exact-fp32 fall-back tile. */
/* First cfg id of the streaming 1x1 kernel (csrc/conv_stream.hip; two ids) and of the patch kernel for the 3x3 stem layers
 * with C = 32, K = 32 / 64, stride 1 (csrc/conv_patch.hip; one id) -- both f16x2 only; an explicit id on a geometry the kernel
 * does not cover is PPY_ERR_BAD_ARG. */
int ppy_conv2d_stream_first_config(void);
int ppy_conv2d_patch_first_config(void);
/* First cfg id of the f16x2 tiles with specialised waves (csrc/conv_ws.hip: four waves deliver operands, four multiply; any
 * geometry the f16x2 tiles take, split-K included; bit-identical results). */
int ppy_conv2d_ws_first_config(void);
/* First cfg id of the wave-private tiles for SMALL outputs (round 6, csrc/conv_small.hip; f16x2 only; four ids: 32x32 / 32x64 output
 * tiles per wave, four / eight waves per workgroup) -- batch 1 (the reference's demo loop, demo.py:121-160) and narrow layers.  A wave
 * owns its tile and one k-part of the reduction, operands go from the L2 straight into MFMA fragment registers.  For these ids
 * `splitk` counts k-parts INSIDE the workgroup (rounded down to a power of two <= the workgroup's waves): the parts are added in the LDS
 * in the order 0, 1, 2, ..., no workspace is needed (ppy_conv2d_workspace_bytes = 0), no combine launch follows, pre-split tensors
 * (ppy_conv2d_bn_act_split_f32) are accepted on both sides with splitk > 1 as well; with splitk in {1, 2, 4, 8} results are
 * bit-identical to the f16x2 tiles' split-K of the same count. */
int ppy_conv2d_small_first_config(void);
/* Writes the tile configuration / split the heuristic would pick. */
int ppy_conv2d_pick(int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                    int *cfg_out, int *splitk_out);

/* The HBM-bound 1x1 "expand" convolutions (ResNet-vd BottleNeck conv3 + shortcut + ReLU, reference
 * model/resnet_vd.py:55-91; C64 -> K256 at 152x152 moves 425 MB for 6 GFLOP) as a persistent streaming kernel
 * (csrc/conv_stream.hip): 1x1, stride 1, C == 64 with K % 64 == 0 or C == 128 with K % 128 == 0 (K / 64 resp. K / 128 a power of
 * two <= 16), f16x2 operands (w_f16x2 / scale_f16x2 / amax_in as above),
 * results bit-identical to the f16x2 tiles of ppy_conv2d_bn_act_f32.  The same kernel is selectable there as the cfg ids
 * ppy_conv2d_stream_first_config() + variant (variant 0: two workgroups per CU, 1: one); this entry point adds the second
 * output: pooled (or NULL) = [N][H/2][W/2] rows of ld pooled_ld holding the 2x2 / stride-2 average of y -- the AvgPool2d(2, 2)
 * in front of the vd projection shortcut (reference model/resnet_vd.py:29-33) written from the same epilogue, evaluated
 * (((a + b) + c) + d) * 0.25 exactly as ppy_avgpool2x2_f32 does (H, W even).  PPY_ERR_BAD_ARG for any other geometry:
 * there is no silent fall-back to another kernel. */
int ppy_conv1x1_expand_f32(const float *x, int x_ld, const void *w_f16x2, const float *scale_f16x2, const float *shift,
                           const float *residual, int res_ld, float *y, int y_ld, float *pooled, int pooled_ld, int N, int H,
                           int W, int C, int K, int act, int variant, const float *amax_in, float *amax_out, void *stream);

/* Training forward of a FROZEN 1x1 Conv2dUnit (reference model/custom_layers.py:243-253 with its BatchNorm2d on batch statistics;
 * train.py:259-262 freezes the backbone) on the streaming kernel of ppy_conv1x1_expand_f32 WITHOUT the raw convolution output in memory:
 *   ppy_conv1x1_stats_f32: the first pass of the BatchNorm only -- (n, mean, M2) partials [*bn_slices][K][3] exactly as
 *     ppy_conv2d_train_fwd_f32 writes them (merge with ppy_bn_train_stats_merge_f32); nothing else is stored.
 *   ppy_conv1x1_bn_apply_f32: the convolution again, y = act((conv + bias - mean) * (invstd * gamma) + beta [+ residual]) from its
 *     epilogue, amax_out (or NULL) = tracked per-image max|y| -- value for value ppy_conv2d_train_fwd_f32 + ppy_bn_train_apply_f32.
 * Same geometry rules as ppy_conv1x1_expand_f32 (C = 64 / 128, f16x2 operands, scale_f16x2 from ppy_conv2d_split_weights_f16x2 with
 * scale = 1); mean / invstd / gamma / beta: [K], 16-byte aligned.  Pays the layer's MFMA work twice to save a write and a read of its
 * output: for the HBM-bound stage-2 layers (6 GFLOP for 0.4-0.8 GB) that is a third of the time. */
int ppy_conv1x1_stats_f32(const float *x, int x_ld, const void *w_f16x2, const float *scale_f16x2, const float *bias, int N, int H,
                          int W, int C, int K, int variant, const float *amax_in, float *bn_partials, size_t bn_partials_bytes,
                          int *bn_slices, void *stream);
int ppy_conv1x1_bn_apply_f32(const float *x, int x_ld, const void *w_f16x2, const float *scale_f16x2, const float *bias,
                             const float *mean, const float *invstd, const float *gamma, const float *beta, const float *residual,
                             int res_ld, float *y, int y_ld, int N, int H, int W, int C, int K, int act, int variant,
                             const float *amax_in, float *amax_out, void *stream);

/* The last stem convolution (reference model/resnet_vd.py:110 conv1_3: 3x3, stride 1, pad 1, C = 32 -> K = 64, BatchNorm affine, ReLU)
 * AND the MaxPool2d(kernel_size=3, stride=2, padding=1) that follows it (model/resnet_vd.py:103, 136) in one launch: only the pooled
 * tensor [N][(H-1)/2+1][(W-1)/2+1][pooled_ld] is written; bit-identical to ppy_conv2d_bn_act_f32 on the patch kernel followed by
 * ppy_maxpool3x3s2_f32 (same products in the same order per pixel, exact maxima; padding positions never win, as with -inf padding).
 * f16x2 operands (w_f16x2 / scale_f16x2 from ppy_conv2d_split_weights_f16x2, amax_in the tracked maxima of x); amax_out (or NULL)
 * records max|.| of the pooled tensor per image (= that of the unpooled one: every pixel lies in a window).
 * PPY_ERR_BAD_ARG for any other geometry (C != 32, K != 64): there is no silent fall-back. */
int ppy_conv3x3_maxpool_f32(const float *x, int x_ld, const void *w_f16x2, const float *scale_f16x2, const float *shift,
                            float *pooled, int pooled_ld, int N, int H, int W, int C, int K, int act, const float *amax_in,
                            float *amax_out, void *stream);

/* ------------------------------------------------------------------------------------
 * Backward of the convolution -- training step, SURVEY 8f rank 2 / BASELINE config 5: what torch autograd computes for
 * the F.conv2d inside Conv2dUnit.forward (reference model/custom_layers.py:243-253) when train.py:441 calls
 * all_loss.backward().  Same tensors and layouts as ppy_conv2d_bn_act_f32: x [N,H,W,C] (ld x_ld), w [K][R][S][C],
 * dy = d loss / d conv output [N,Ho,Wo,K] (ld dy_ld; the BN / activation backward is the caller's, upstream of dy).
 *   ppy_conv2d_dgrad_f32: dx[n,h,w,c] = sum_{k,r,s} dy[n,h+pad-r,w+pad-s,k] * w[k,r,s,c]  (written, not accumulated).
 *     stride 1 only (PPY_ERR_UNSUPPORTED otherwise: with the reference's freeze_at = 5 only the head trains and it has
 *     no strided convolution).  Runs the forward implicit-GEMM kernels on the flipped / transposed weights; K need not
 *     be a multiple of 32 (the 258-channel output convolutions are zero-padded in the workspace).  cfg / splitk: tile
 *     configuration of that forward kernel for the geometry (N, Ho, Wo, C' = K rounded up to 32, K' = C), as in
 *     ppy_conv2d_bn_act_f32 (-1 / 0 = heuristic).  amax_dy (or NULL): tracked per-image maxima of dy -> the f16x2 kernels
 *     (cfg then names an f16x2 tile), else bf16x3.
 *   ppy_conv2d_wgrad_f32: dw[k,r,s,c] = sum_{n,ho,wo} dy[n,ho,wo,k] * x[n,ho*stride+r-pad,wo*stride+s-pad,c]
 *     (written, not accumulated; any stride / C / K).  bf16x3 on the 16-bit MFMA (exact fp32 MFMA where alignment rules it
 *     out); with amax_x / amax_dy -- tracked per-image maxima of both operands, as ppy_bn_train_apply_f32 / _bwd_f32 and the
 *     convolution entry point record them -- the f16x2 scheme (one power-of-two scale per operand, 3 products instead of 6).
 *     Pixel slices are combined in a fixed order, so results are run-to-run identical.
 * ws: ppy_conv2d_{dgrad,wgrad}_workspace_bytes() bytes, 256-byte aligned.
 */
/* Training-mode forward of the convolution in front of a BatchNorm2d on batch statistics (reference
 * model/custom_layers.py:243-253): y = conv(x, w) + bias on an f16x2 tile (cfg >= 0: an f16x2 id of csrc/conv_x3.hip or
 * csrc/conv_ws.hip; one split) and, from the same epilogue, the first pass of the BatchNorm: per wave row-tile ("slice") and
 * channel the triple (n, mean, M2) of y in bn_partials[*bn_slices][K][3].  ppy_conv2d_bn_partials_bytes(M, K) bytes always hold
 * the triples AND the scratch ppy_bn_train_stats_merge_f32 needs behind them; that call turns them into mean / invstd / running
 * statistics exactly as ppy_bn_train_stats_f32 does from a pass over y (same Chan merge, slices in order) -- without reading y
 * again.  w_f16x2 / scale_f16x2: ppy_conv2d_split_weights_f16x2 with scale = 1.  PPY_ERR_UNSUPPORTED: cfg names another kernel
 * family (the caller then runs ppy_conv2d_bn_act_f32 + ppy_bn_train_stats_f32). */
size_t ppy_conv2d_bn_partials_bytes(long long M, int K);
int ppy_conv2d_train_fwd_f32(const float *x, int x_ld, const float *w_krsc, const void *w_f16x2, const float *scale_f16x2,
                             const float *bias, float *y, int y_ld, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                             int cfg, const float *amax_in, float *bn_partials, size_t bn_partials_bytes, int *bn_slices, void *stream);
int ppy_bn_train_stats_merge_f32(float *partials, size_t partials_bytes, int slices, int C, float eps, float momentum, float *mean,
                                 float *invstd, float *running_mean, float *running_var, void *stream);
int ppy_conv2d_dgrad_f32(const float *dy, int dy_ld, const float *w_krsc, float *dx, int dx_ld, int N, int H, int W,
                         int C, int K, int R, int S, int stride, int pad, int cfg, int splitk, const float *amax_dy, void *ws,
                         size_t ws_bytes, void *stream);
size_t ppy_conv2d_dgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int cfg,
                                        int splitk);
/* Round 3: the f16x2 operand forms of ALL trainable convolution weights in three launches per step instead of two or three
 * per layer (train.py:442 `optimizer.step()` changes every weight every iteration, so the forward planes and the flipped /
 * transposed planes of the data gradients are rebuilt every step: 47 launches of 7-15 us for the R50vd head).
 *   desc[i]: one weight tensor [K][R][S][C] (C % 32 == 0).  fwd_planes / fwd_scale: exactly what
 *   ppy_conv2d_split_weights_f16x2(w, K, R*S*C, ones, ...) writes.  dgrad_planes / dgrad_scale / dgrad_wt (all three or none):
 *   w'[c][r'][s'][k] = w[k][R-1-r'][S-1-s'][c], K padded with zeros to Kp = K rounded up to 32 -- the planes
 *   [2][R*S*Kp/32][C][32], per-row scales [C] and fp32 copy [C][R][S][Kp] that ppy_conv2d_dgrad_f32 builds in its workspace
 *   (bit-identical).  row0 / blk0: prefix sums over the descriptors of K and of (C / 32) * PPY_PREP_SPLIT.  The table lives in
 *   device memory.  colmax: scratch of blocks_total * 32 floats.
 *   ppy_conv2d_dgrad_prepared_f32 = ppy_conv2d_dgrad_f32 (stride 1, f16x2: amax_dy required) on such buffers; ones / zeros: [C]
 *   constants; workspace as ppy_conv2d_dgrad_workspace_bytes says (the padded copy of a dy with K % 32 != 0, split-K). */
#define PPY_PREP_SPLIT 8
typedef struct PpyWeightPrep {
    const float *w;
    void *fwd_planes;
    float *fwd_scale;
    void *dgrad_planes;
    float *dgrad_scale;
    float *dgrad_wt;
    int K, R, S, C;
    int row0, blk0;
} PpyWeightPrep;
int ppy_train_prepare_weights_f16x2(const PpyWeightPrep *desc_dev, int count, int rows_total, int blocks_total, float *colmax,
                                    size_t colmax_bytes, void *stream);
int ppy_conv2d_dgrad_prepared_f32(const float *dy, int dy_ld, const float *wt, const void *planes, const float *scale_f16x2,
                                  const float *ones, const float *zeros, float *dx, int dx_ld, int N, int H, int W, int C, int K, int R,
                                  int S, int pad, int cfg, int splitk, const float *amax_dy, void *ws, size_t ws_bytes, void *stream);
int ppy_conv2d_wgrad_f32(const float *x, int x_ld, const float *dy, int dy_ld, float *dw_krsc, int N, int H, int W,
                         int C, int K, int R, int S, int stride, int pad, const float *amax_x, const float *amax_dy, void *ws,
                         size_t ws_bytes, void *stream);
size_t ppy_conv2d_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad);

/* ------------------------------------------------------------------------------------
 * Training-step operators around the convolutions (SURVEY 8f rank 2 / BASELINE config 5; csrc/train.hip).  Tensors are
 * fp32 NHWC slices (pointer + pixel stride), P = N*H*W pixels.
 *  ppy_bn_train_stats_f32 / _apply_f32: torch.nn.BatchNorm2d in TRAINING mode (the reference trains with every BatchNorm
 *    on batch statistics: train.py never calls .eval(); model/custom_layers.py:122): per-channel mean and
 *    invstd = 1/sqrt(biased var + eps) of x, running statistics updated in place with `momentum` (unbiased variance), then
 *    y = act((x - mean) * invstd * gamma + beta [+ residual]); amax_out (or NULL): per-image max|y| slots as the convolution
 *    entry point tracks them (zeroed by the caller; pixels_per_image = H*W) -- the operand scale of a following f16x2 kernel.
 *  ppy_bn_train_bwd_f32 (amax_dx or NULL: per-image max|dx| slots, as amax_out above): given dy = d loss / d y: dz = dy * act'(y), dbeta = sum dz, dgamma = sum dz * xhat,
 *    dx = gamma * invstd * (dz - (dbeta + xhat * dgamma) / P).
 *  ppy_act_bwd_f32: dx = dy * act'(y) alone.
 *  ppy_upsample2x_bwd_f32: backward of the nearest x2 upsample of the head routes (model/head.py:396-397): dx = sum of the
 *    2x2 block of dy (accumulate != 0: added to dx).
 *  ppy_spp_bwd_f32: backward of SPP (model/custom_layers.py:281-290): dy is [N,H,W,4C] = (x, pool5, pool9, pool13); every
 *    window's gradient goes to its first maximum in (h, w) scan order, as torch.max_pool2d's does.
 *  ppy_dropblock_mask_f32 / _apply_f32: DropBlock in training mode (custom_layers.py:303-342): seeds = u < gamma (gamma from
 *    the feature HEIGHT, like the reference), mask = 1 - maxpool3x3(seeds), scale = numel / sum(mask); y = x * mask * scale
 *    (the same map serves the backward).  u comes from a counter-based generator keyed by `seed` -- the reference draws from
 *    torch's global generator, so masks are comparable in distribution only; parity tests inject the reference's mask.
 *  ppy_sgd_momentum_f32: torch.optim.SGD(momentum, weight_decay) as built by train.py:271-280:
 *    d = g + wd * p; v = first_step ? d : mu * v + d; p -= lr * v.
 * All reductions run in a fixed order (no float atomics): results are run-to-run identical.
 *  ppy_avgpool2x2_bwd_f32 / ppy_maxpool3x3s2_bwd_f32: backward of the vd shortcut's AvgPool2d(2, 2) (resnet_vd.py:29-33) and of
 *    the stem's MaxPool2d(3, 2, 1) (:103; the gradient of a window goes to its first maximum in scan order, as in torch) --
 *    training with freeze_at < 5.  dx [N,H,W,C] is written whole.
 *  ppy_zero_insert_f32: up[n, i*s, j*s, :] = dy[n, i, j, :] into a zeroed [N,H1,W1,C]: the data gradient of a stride-s
 *    convolution = ppy_conv2d_dgrad_f32 (stride 1) of the zero-inserted gradient with H1 = H + 2*pad - R + 1.
 */
size_t ppy_bn_train_workspace_bytes(int P, int C);
int ppy_avgpool2x2_bwd_f32(const float *dy, int dy_ld, float *dx, int dx_ld, int N, int H, int W, int C, void *stream);
int ppy_maxpool3x3s2_bwd_f32(const float *x, int x_ld, const float *dy, int dy_ld, float *dx, int dx_ld, int N, int H, int W, int C,
                             void *stream);
int ppy_zero_insert_f32(const float *dy, int dy_ld, float *up, int up_ld, int N, int Ho, int Wo, int C, int H1, int W1, int stride,
                        void *stream);
int ppy_bn_train_stats_f32(const float *x, int x_ld, int P, int C, float eps, float momentum, float *mean, float *invstd,
                           float *running_mean, float *running_var, void *ws, size_t ws_bytes, void *stream);
int ppy_bn_train_apply_f32(const float *x, int x_ld, const float *mean, const float *invstd, const float *gamma,
                           const float *beta, const float *residual, int res_ld, float *y, int y_ld, int P, int C, int act,
                           int pixels_per_image, float *amax_out, void *stream);
int ppy_bn_train_bwd_f32(const float *x, int x_ld, const float *y, int y_ld, const float *dy, int dy_ld, const float *mean,
                         const float *invstd, const float *gamma, float *dx, int dx_ld, float *dgamma, float *dbeta, int P, int C,
                         int act, int pixels_per_image, float *amax_dx, void *ws, size_t ws_bytes, void *stream);
int ppy_act_bwd_f32(const float *dy, int dy_ld, const float *y, int y_ld, float *dx, int dx_ld, long long P, int C, int act,
                    void *stream);
int ppy_upsample2x_bwd_f32(const float *dy, int dy_ld, float *dx, int dx_ld, int N, int H, int W, int C, int accumulate,
                           void *stream);
size_t ppy_spp_bwd_workspace_bytes(int N, int H, int W, int C);
int ppy_spp_bwd_f32(const float *x, int x_ld, const float *dy, int dy_ld, float *dx, int dx_ld, int N, int H, int W, int C,
                    void *ws, size_t ws_bytes, void *stream);
size_t ppy_dropblock_workspace_bytes(int N, int H, int W, int C);
int ppy_dropblock_mask_f32(float *mask, float *scale_out, int N, int H, int W, int C, int block_size, float keep_prob,
                           unsigned long long seed, void *ws, size_t ws_bytes, void *stream);
int ppy_dropblock_apply_f32(const float *x, int x_ld, const float *mask, const float *scale, float *y, int y_ld, long long P,
                            int C, void *stream);
int ppy_sgd_momentum_f32(float *param, const float *grad, float *velocity, long long n, float lr, float momentum,
                         float weight_decay, int first_step, void *stream);
/* ExponentialMovingAverage.update of the reference (model/EMA.py:29-44; train.py:443-444), on the device instead of a numpy
 * round trip per step: shadow = decay * shadow + (1 - decay) * param in float32 with separate roundings (numpy's arithmetic);
 * the caller computes decay = min(ema_decay, (1 + t) / (10 + t)) and passes float32(decay), float32(1 - decay). */
int ppy_ema_update_f32(float *shadow, const float *param, long long n, float decay, float one_minus_decay, void *stream);
/* Plumbing of the training graph: dst += src (second gradient of a tensor with two consumers); the nearest x2 upsample
 * forward (model/head.py:396-397; the inference path fuses it into the producing convolution's store, which BatchNorm on
 * batch statistics rules out); per-channel sum over the pixels = gradient of a convolution bias
 * (ws: ppy_bn_train_workspace_bytes(P, C)). */
int ppy_add_inplace_f32(float *dst, int dst_ld, const float *src, int src_ld, long long P, int C, void *stream);
int ppy_upsample2x_f32(const float *x, int x_ld, float *y, int y_ld, int N, int H, int W, int C, void *stream);
int ppy_channel_sum_f32(const float *dy, int dy_ld, int P, int C, float *out, void *ws, size_t ws_bytes, void *stream);

/* YOLOv3Loss of one head level, forward AND backward (csrc/yolo_loss.hip): the reference's
 * YOLOv3Loss._get_fine_grained_loss (model/losses.py:121-253) with IouLoss / IouAwareLoss (model/iou_losses.py:39-246) and
 * the ignore mask (losses.py:296-356), plus the analytic gradient of the SUM of all loss terms with respect to the head
 * output -- what `all_loss.backward()` (train.py:441) hands to the output convolutions.
 * head_out: NHWC [N,S,S,*] raw output of this level, channels [an IoU logits if iou_aware][an x (x, y, w, h, obj, C classes)].
 * target: the reference's Gt2YoloTarget tensor [N, an, 6 + C, S, S] (tx, ty, tw, th, tscale, tobj, classes).
 * gt_box: [N, num_gt, 4] normalised (cx, cy, w, h), zero rows = padding.  h_anchors_px: HOST array, an x (w, h) of this level.
 * dout: NHWC like head_out (channels beyond an*(5+C)(+an) are not written).  loss6: device floats
 * {loss_xy, loss_wh, loss_obj, loss_cls, loss_iou, loss_iou_aware}, each the batch mean as the reference logs it
 * (accumulate != 0: added to what is there, for the sum over levels).  ws: ppy_yolov3_loss_workspace_bytes().
 * iou_loss_square: IouLoss(loss_square=), 1 = (1 - iou^2) * weight (both configurations), 0 = (1 - iou) * weight (iou_losses.py:66-70);
 * IouLoss(ciou_term=True) is not implemented (the reference's own branch stops at `alpha.requires_grad = False`, :131, under autograd).
 * amax_dout (or NULL): zeroed per-image maximum slots that receive max|dout| -- the operand scale of the f16x2 gradient kernels
 * of the output convolution.
 */
size_t ppy_yolov3_loss_workspace_bytes(int N, int S, int an);
int ppy_yolov3_loss_f32(const float *head_out, int out_ld, const float *target, const float *gt_box, int num_gt,
                        const float *h_anchors_px, int an, int num_classes, int N, int S, int downsample, double scale_x_y,
                        double ignore_thresh, double iou_loss_weight, int iou_loss_square, int iou_aware,
                        double iou_aware_loss_weight, float *dout, int dout_ld, float *loss6, int accumulate, float *amax_dout,
                        void *ws, size_t ws_bytes, void *stream);

/* First backbone conv, `stage1_conv1_1` (reference model/resnet_vd.py:100, :133): 3x3
 * stride-2 conv C_in=3 -> K (K % 4 == 0, K <= 64) + BN affine + ReLU, reading the
 * caller's NCHW input directly and writing NHWC (fuses the layout change).
 * x: [N,3,H,W] NCHW contiguous.  w: [K][3][3][3] in the reference's own KCRS order. */
int ppy_stem_conv3x3s2_nchw_f32(const float *x_nchw, const float *w_kcrs, const float *scale,
                                const float *shift, float *y, int y_ld, int N, int H, int W,
                                int K, int act, float *amax_out /* NULL or N * PPY_AMAX_FLOATS_PER_IMAGE floats */, void *stream);

/* The same operator on the bf16 MFMA (round 4): the 27-deep reduction as two k-steps of v_mfma_f32_32x32x16_bf16 on
 * exactly-split operands (3 bf16 terms per fp32 value, 6 partial products, fp32 accumulate -- no scaling, so any input
 * range).  K == 32 only (PPY_ERR_UNSUPPORTED otherwise).  Agrees with the fma chain of the entry above to fp32
 * rounding, not bit for bit; PPYOLO_HIP_MATH=fp32 keeps the entry above. */
int ppy_stem_conv3x3s2_nchw_x3_f32(const float *x_nchw, const float *w_kcrs, const float *scale,
                                   const float *shift, float *y, int y_ld, int N, int H, int W,
                                   int K, int act, float *amax_out, void *stream);

/* Image pre-processing in front of the path, reference Decode.process_image
 * (model/decode_np.py:125-140): BGR->RGB (swap_rb), cv2.resize(fx = S/w, fy = S/h, INTER_CUBIC) on uint8
 * (tools/transform.py:996-1003; OpenCV's 11-bit fixed-point bicubic, A = -0.75, replicated border), then the numpy
 * normalisation (transform.py:895-917) given as lut[c][v] = value of grey level v in OUTPUT channel c, and HWC->CHW
 * (transform.py:1052-1054).  images[i]: device pointer to an h[i] x w[i] x 3 uint8 image with row_stride[i] bytes per
 * row; the arrays images / h / w / row_stride are HOST arrays of length n.  out: device [n][3][S][S] float32. */
int ppy_preprocess_u8_f32(int n, const unsigned char *const *images, const int *h, const int *w,
                          const int *row_stride, int swap_rb, int S, const float *lut /* device [3][256] */,
                          float *out, void *stream);

/* torch.nn.MaxPool2d(3, 2, 1) of the stem (reference model/resnet_vd.py:103, :136). */
int ppy_maxpool3x3s2_f32(const float *x, int x_ld, float *y, int y_ld, int N, int H, int W,
                         int C, void *stream);
/* torch.nn.AvgPool2d(2, 2, 0) of the vd shortcut (reference model/resnet_vd.py:30, :53). */
int ppy_avgpool2x2_f32(const float *x, int x_ld, float *y, int y_ld, int N, int H, int W, int C,
                       void *stream);
/* SPP (reference model/custom_layers.py:275-290): max-pool 5/9/13, stride 1, "same".
 * x is channel slice [0,C) of the concat buffer; the three pooled maps are written to
 * y5, y9, y13 (normally x+C, x+2C, x+3C of the same buffer), all with ld y_ld. */
int ppy_spp_f32(const float *x, int x_ld, float *y5, float *y9, float *y13, int y_ld, int N,
                int H, int W, int C, void *stream);

/* DCNv2 (reference model/custom_layers.py:551-677; arg order after
 * external/DCNv2/src/dcn_v2.h:9-23: input, weight, bias-like, offset, mask, kernel,
 * stride, pad).  `offset_mask`: NHWC [N,Ho,Wo,27] RAW output of conv_offset (18
 * (y,x)-interleaved offsets then 9 mask logits; sigmoid is applied here).
 * ppy_dcnv2_f32 is ONE kernel: every workgroup gathers the modulated bilinear samples of its
 * output tile into LDS, chunk by chunk, and contracts them on the MFMA with w [K][3][3][C],
 * then BN affine + activation -- there is no "columns" buffer; ws holds split-K partials only
 * (ppy_dcnv2_workspace_bytes; 0 without split-K).  cfg: -1 = heuristic, else
 * math scheme * (ppy_dcnv2_num_configs() / 3) + tile, schemes 0 exact fp32, 1 bf16x3 (needs
 * w_x3), 2 f16x2 (needs w_f16x2, scale_f16x2, amax_in) -- the operands of the convolution entry
 * point above.  ppy_dcnv2_sample_f32 is the stand-alone gather: columns
 * cols[N*Ho*Wo][9*C] in (tap, c) order, the same arithmetic op for op. */
int ppy_dcnv2_sample_f32(const float *x, int x_ld, const float *offset_mask, int om_ld,
                         float *cols, int N, int H, int W, int C, int Ho, int Wo, int stride,
                         int pad, void *stream);
int ppy_dcnv2_f32(const float *x, int x_ld, const float *w_krsc, const void *w_x3, const void *w_f16x2,
                  const float *scale, const float *scale_f16x2, const float *shift,
                  const float *offset_mask, int om_ld, float *y, int y_ld, int N, int H, int W, int C,
                  int K, int stride, int pad, int act, int cfg, int splitk, const float *amax_in,
                  float *amax_out, void *ws, size_t ws_bytes, void *stream);
size_t ppy_dcnv2_workspace_bytes(int N, int H, int W, int C, int K, int stride, int pad, int cfg,
                                 int splitk);
int ppy_dcnv2_num_configs(void);

/* Backward of the deformable convolution (training with freeze_at < 5; arg order after the reference's only real FFI,
 * dcn_v2_backward(input, weight, bias, offset, mask, grad_output, ...) -> [dX, dOffset, dMask, dW, dBias],
 * external/DCNv2/src/dcn_v2.h:41-55; semantics = torch autograd of the pure-PyTorch DCNv2.forward,
 * model/custom_layers.py:551-677).  dy [N,Ho,Wo,K]: gradient of the contraction output (upstream of BN / activation).
 * Written: dx [N,H,W,C] -- the SAMPLING path only, conv_offset's own data gradient is a plain convolution backward of
 * d_offset_mask and the caller's to add --, d_offset_mask [N,Ho,Wo,27] with respect to the RAW conv_offset output (18
 * offsets, 9 mask logits), dw [K][3][3][C].  dx is accumulated with float atomics (data-dependent scatter, as
 * external/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:197-262 does): not bit-reproducible run to run; d_offset_mask and dw are.
 * ws: ppy_dcnv2_backward_workspace_bytes() bytes, 256-byte aligned (the columns + the inner wgrad / dgrad workspaces). */
int ppy_dcnv2_backward_f32(const float *x, int x_ld, const float *w_krsc, const float *offset_mask, int om_ld,
                           const float *dy, int dy_ld, float *dx, int dx_ld, float *d_offset_mask, int dom_ld,
                           float *dw_krsc, int N, int H, int W, int C, int K, int stride, int pad, void *ws,
                           size_t ws_bytes, void *stream);
size_t ppy_dcnv2_backward_workspace_bytes(int N, int H, int W, int C, int K, int stride, int pad);

/* ------------------------------------------------------------------------------------
 * get_iou_aware_score + yolo_box for ONE head level (reference model/head.py:21-141),
 * fused with the `scores > score_threshold` candidate extraction of matrix_nms
 * (reference model/matrix_nms.py:110-117).
 * head_out: NHWC [N,S,S,A*(5+C) (+A if iou_aware)] with ld head_ld (channel layout as in
 * the reference: [A IoU logits] then A x [tx,ty,tw,th,obj,C classes]).
 * anchors_px: HOST pointer to A (w,h) pairs in pixels.  im_size: [N][2] = (h, w).
 * boxes: [N][M_total][4]; this level writes rows [box_offset, box_offset + S*S*A) in
 * (h, w, anchor) order.  Candidates with score > score_threshold are appended to
 * cand_key/cand_idx ([N][cand_cap]; key = order-preserving uint32 image of the fp32
 * score: bits ^ 0x80000000 for non-negative scores, ~bits for negative ones; idx =
 * box*C + class) with the running count in cand_count[N] (zero it before the first
 * level; it keeps counting past cand_cap, entries beyond cand_cap are dropped, so size
 * cand_cap = M_total*C to make that impossible).  scale_x_y / iou_aware_factor are
 * doubles because the reference derives fp32 constants from Python doubles
 * ((scale_x_y - 1.0) * 0.5, 1 - iou_aware_factor; model/head.py:40, :125).
 * scores_dense: optional [N][M_total][C] (reference layout) or NULL. */
int ppy_yolo_decode_f32(const float *head_out, int head_ld, int N, int S, int A, int num_classes,
                        const float *h_anchors_px, int downsample, double scale_x_y,
                        int iou_aware, double iou_aware_factor, int clip_bbox,
                        const float *im_size, float *boxes, int M_total, int box_offset,
                        float score_threshold, uint32_t *cand_key, uint32_t *cand_idx,
                        int *cand_count, int cand_cap, float *scores_dense, void *stream);

/* The same for ALL head levels of a model in one launch (per-level arrays of length nlevels <= 4; the anchor
 * count A is common to the levels).  No dense-score output.  Candidate order inside an image's list differs
 * from level-by-level calls (the list is an unordered set; matrix_nms orders by (score, index)). */
int ppy_yolo_decode_levels_f32(int nlevels, const float *const *head_out, const int *head_ld, const int *S,
                               const int *downsample, const float *const *h_anchors_px,
                               const int *box_offset, int N, int A, int num_classes, double scale_x_y,
                               int iou_aware, double iou_aware_factor, int clip_bbox, const float *im_size,
                               float *boxes, int M_total, float score_threshold, uint32_t *cand_key,
                               uint32_t *cand_idx, int *cand_count, int cand_cap, void *stream);

/* matrix_nms (reference model/matrix_nms.py:102-151) for a batch, from the candidate
 * lists produced by ppy_yolo_decode_f32 / ppy_nms_candidates_f32.
 * Order (this build's total order where the reference calls the unstable
 * torch.argsort): score descending, ties by ascending candidate index.
 * out_dets [N][keep_top_k][6] = (label, score, x0,y0,x1,y1), rows >= out_count[n] are
 * -1; out_count[n] == 0 <=> the reference returns its [[-1]*6] sentinel.
 * out_keep_idx [N][keep_top_k] = box*C + class of every kept row (-1 padding).
 * Limits: 1 <= nms_top_k <= 1024, keep_top_k <= nms_top_k.
 * ws: ppy_matrix_nms_workspace_bytes(N) bytes of scratch, 16-byte aligned (the sorted top-k boxes and the
 * per-column compensate / decay values travel between the kernels of one call through it; round 4: + 264 KB per image
 * for the compact list of the large-list route).
 * Lists of more than 8192 candidates (only possible when cand_cap > 8192; up to 1.8 M per image when every (box,
 * class) pair passes the threshold) are first cut down chip-wide -- a threshold from 8192 sampled keys, then one
 * streaming pass that moves every entry at or above it to a compact list and counts them -- and the top-k select
 * runs on the compact list iff that list is certain to contain the whole top-k; otherwise it walks the original
 * list.  The result is the same either way (csrc/decode_nms.hip: nms_sample_kernel, nms_collect_kernel). */
size_t ppy_matrix_nms_workspace_bytes(int N);
int ppy_matrix_nms_f32(const float *boxes, int M_total, int num_classes, const uint32_t *cand_key,
                       const uint32_t *cand_idx, const int *cand_count, int cand_cap, int N,
                       float post_threshold, int nms_top_k, int keep_top_k, int use_gaussian,
                       float gaussian_sigma, float *out_dets, int *out_count, int *out_keep_idx,
                       void *ws, size_t ws_bytes, void *stream);
/* Candidate extraction from dense scores [N][M][C] (the reference's own input form,
 * model/matrix_nms.py:110-117), for callers that hold yolo_box outputs. */
int ppy_nms_candidates_f32(const float *scores, int N, int M, int C, float score_threshold,
                           uint32_t *cand_key, uint32_t *cand_idx, int *cand_count, int cand_cap,
                           void *stream);

/* ------------------------------------------------------------------------------------
 * conv2 -> conv3 of an identity bottleneck in ONE launch (round 5; reference model/resnet_vd.py:81-87:
 * relu(bn3(conv3(relu(bn2(conv2(t))))) + x)): conv A = 3x3 / stride 1 / pad 1, CA -> KA, BatchNorm, ReLU; conv B = 1x1, KA -> KB,
 * BatchNorm, + residual, ReLU.  The KA-channel intermediate stays on chip (csrc/conv_b2b.hip).  x_split: conv A's input as its
 * producer stored it pre-split (ppy_conv2d_bn_act_split_f32), xscale its [N] per-image scales, amax_in the tracked max|.| of that
 * tensor.  wA / wB + scaleA / scaleB: outputs of ppy_conv2d_split_weights_f16x2 for the two weight tensors; shiftA / shiftB the
 * folded BatchNorm shifts.  t_mul, t_add: static bound |intermediate| <= t_mul * max|x| + t_add (per-image operand scale of the
 * second contraction).  pool (or NULL) / pool_ld: [N, H/2, W/2, pool_ld] receives AvgPool2d(2, 2) of y as well (the vd shortcut of the
 * next stage, reference model/resnet_vd.py:29-33; H, W even), evaluated (((a + b) + c) + d) * 0.25 as ppy_avgpool2x2_f32.
 * Supported: CA = KA = 64, KB = 256 (stage 2 of ResNet50-vd); else PPY_ERR_UNSUPPORTED.  f16x2 arithmetic as
 * ppy_conv2d_bn_act_f32; agrees with the two-launch form to fp32 rounding. */
int ppy_conv3x3_conv1x1_f32(const float *x_split, int x_ld, const float *xscale, const float *amax_in, const void *wA_f16x2,
                            const float *scaleA_f16x2, const float *shiftA, const void *wB_f16x2, const float *scaleB_f16x2,
                            const float *shiftB, const float *residual, int res_ld, float *y, int y_ld, float *pool, int pool_ld, int N,
                            int H, int W, int CA, int KA, int KB, float t_mul, float t_add, float *amax_out, void *stream);

/* ------------------------------------------------------------------------------------
 * Lanes.  The reference evaluates one batch at a time (model/ppyolo.py:19-22, called from model/decode_np.py:142-150); this
 * path keeps several batches on the device at once, each on its own stream ("lane", ppyolo_hip/runtime.py InFlight).
 * ppy_lane_stream_create makes such a stream: mask_words == 0 -> an ordinary non-blocking stream; else h_cu_mask (HOST memory,
 * mask_words x 32 bits) restricts every kernel launched on it -- directly or as nodes of a hipGraph launched on it -- to the
 * compute units whose bit is set (hipExtStreamCreateWithCUMask).  On MI355X bit i is CU i / 8 of XCD i % 8
 * (tools/probes/cu_mask_probe.hip), so `i % 8 < 4` gives a lane four whole XCDs with their L2s.  An all-zero mask is refused.
 * These two calls are the only ones of the library that create or destroy state; they synchronise nothing. */
int ppy_lane_stream_create(void **stream, const uint32_t *h_cu_mask, int mask_words);
int ppy_lane_stream_destroy(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PPYOLO_HIP_H_ */
