/* libdfine_hip.so - C ABI of the MI355X (gfx950) D-FINE hot-path kernels.
 *
 * Plain C entry points: raw DEVICE pointers + sizes + a hipStream_t (passed as void*), int
 * status return (0 = ok, <0 = DFINE_E_*), no exceptions, no hidden allocation: every buffer,
 * including scratch, is owned by the caller.  All launches are asynchronous on `stream`.
 * Tensors are dense row-major ("contiguous") in the shapes given below.
 *
 * dtype codes: DFINE_F32 = 0, DFINE_BF16 = 1 (bf16 tensors are read/written as bf16 and all
 * arithmetic is fp32).
 *
 * The reference is pure Python/ATen, so each entry point replaces a
 * Python function / ATen call sequence of /root/reference rather than an existing FFI symbol;
 * the file:line it replaces is cited on every declaration.  INTEGRATION.md shows the ctypes
 * binding (custom_d_fine_amd/hip.py) that a maintainer of the reference would add.
 */
#ifndef DFINE_HIP_H
#define DFINE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFINE_OK 0
#define DFINE_E_BADARG (-1)   /* unsupported shape / dtype / null pointer */
#define DFINE_E_LAUNCH (-2)   /* hipLaunch / runtime error (see dfine_last_error) */

#define DFINE_F32 0
#define DFINE_BF16 1

#define DFINE_MAX_LEVELS 8
#define DFINE_MAX_POINTS 32   /* sum of sampling points over levels */

/* Library / build info: returns the ABI version (bumped on any signature change). */
int dfine_abi_version(void);
/* Text of the last HIP runtime error seen by this library on the calling thread ("" if none). */
const char *dfine_last_error(void);

/* Stream `to` waits for everything enqueued on stream `from` so far (hipEventRecord + hipStreamWaitEvent on an internal event
 * ring): fork / join of the second stream that runs the weight-gradient launches of a backward pass next to the data-gradient
 * chain (the reference runs both on one stream through autograd: torch.Tensor.backward in src/dl/train.py:575). */
int dfine_stream_fork(void *from, void *to);
/* A non-blocking stream of `priority` (clamped to the device's range: -1 high, 0 normal, 1 low on gfx950; torch.cuda.Stream only
 * offers 0 / -1) on the current device, for the low-priority weight-gradient side stream; destroyed by dfine_stream_destroy. */
int dfine_stream_create(int priority, void **out);
int dfine_stream_destroy(void *stream);
/* hipMemcpyAsync(dst, src, bytes, HostToDevice, stream) from memory the caller keeps pinned, alive and unchanged: the pointer
 * tables of launches recorded inside a HIP-graph capture of the backward pass (custom_d_fine_amd/dl/engine.py; the reference
 * launches the same work eagerly from torch.Tensor.backward, src/dl/train.py:575).  The copy becomes a memcpy node that
 * re-reads `src` at every replay. */
int dfine_upload(void *dst, const void *src, int64_t bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A7  Multi-scale deformable attention gather.
 * Replaces deformable_attention_core_func_v2 (src/d_fine/arch/utils.py:191-264): per level
 * F.grid_sample(bilinear, zeros, align_corners=False) + concat + mul + sum.
 *
 *   value   [B, L, H, D]   dtype `dtype`; L = sum_l h_l*w_l, row of pixel (y,x) of level l =
 *                          start_l + y*w_l + x   (the encoder memory as it is, no permute)
 *   loc     [B, Lq, H, P, 2] f32, (x, y) in [0,1];   weight [B, Lq, H, P] f32
 *   out     [B, Lq, H*D]   dtype `dtype`
 *   level_hw[2*n_levels] = h0,w0,h1,w1,..  level_points[n_levels], P = sum level_points
 * D must be 16, 32 or 64 (D-FINE: 32; size n: 16).
 */
int dfine_msda_fwd(const void *value, const float *loc, const float *weight, void *out,
                   int dtype, int B, int L, int H, int D, int Lq, int n_levels,
                   const int *level_hw, const int *level_points, void *stream);

/* Backward of dfine_msda_fwd (reference: autograd through grid_sample_backward + mul/sum).
 *   grad_out [B, Lq, H*D] dtype `dtype`
 *   grad_value_f32 [B, L, H, D] f32, MUST be zero-filled by the caller (accumulated with
 *                  hardware f32 atomics); grad_loc [B,Lq,H,P,2] f32; grad_weight [B,Lq,H,P] f32
 */
int dfine_msda_bwd(const void *value, const float *loc, const float *weight,
                   const void *grad_out, float *grad_value_f32, float *grad_loc,
                   float *grad_weight, int dtype, int B, int L, int H, int D, int Lq,
                   int n_levels, const int *level_hw, const int *level_points, void *stream);

/* Fused variant used by the decoder: also replaces MSDeformableAttention.forward's
 * softmax over points and sampling-location arithmetic
 * (src/d_fine/arch/dfine_decoder.py:147,156-166):
 *     loc = ref_xy + offsets * (1 / points_of_level) * ref_wh * offset_scale
 *     w   = softmax_P(logits)
 *   ref [B, Lq, 4] f32 cxcywh;  offsets [B, Lq, H, P, 2] and logits [B, Lq, H, P] dtype `dtype`
 */
int dfine_msda_fused_fwd(const void *value, const float *ref, const void *offsets,
                         const void *logits, void *out, int dtype, int B, int L, int H, int D,
                         int Lq, int n_levels, const int *level_hw, const int *level_points,
                         float offset_scale, void *stream);

/* grad_offsets / grad_logits have dtype `dtype`; grad_value_f32 as in dfine_msda_bwd. */
int dfine_msda_fused_bwd(const void *value, const float *ref, const void *offsets,
                         const void *logits, const void *grad_out, float *grad_value_f32,
                         void *grad_offsets, void *grad_logits, int dtype, int B, int L, int H,
                         int D, int Lq, int n_levels, const int *level_hw,
                         const int *level_points, float offset_scale, void *stream);

/* The same backward with a choice of how d(value) is accumulated (grad_value_acc [B, L, H, D], zero-filled by the
 * caller, shared by the decoder layers of one step):
 *   acc_mode 0: f32 words, hardware f32 atomics (= dfine_msda_fused_bwd);
 *   acc_mode 2: f16 words, ONE packed f16 atomic per channel pair (the L2 atomic units retire one dword per clock and
 *               channel: half the dwords = half the time), contributions multiplied by a power-of-two scale that keeps the
 *               largest possible sum below 2^15;
 *   acc_mode 3: int32 fixed-point words, the channel pair (2j, 2j+1) packed as lo + hi * 2^32 in one int64 and added
 *               with ONE integer atomic - exact and commutative (the sum does not depend on the order the atomics retire
 *               in), 10 % faster than f32.
 * Modes 2 / 3: fx_state = 4 device words zero-filled with the accumulator (scale, capacity, scratch); the call sizes the
 * scale from max |grad_out| so that hit_bound (= calls sharing the accumulator x Lq: every query adds at most |grad_out|
 * to one word) x capacity stays below the limit, and rescales the accumulator in place if a later call brings larger
 * gradients.  Finish with dfine_cast_scaled_acc. */
int dfine_msda_fused_bwd_acc(const void *value, const float *ref, const void *offsets,
                             const void *logits, const void *grad_out, void *grad_value_acc,
                             void *grad_offsets, void *grad_logits, int dtype, int B, int L, int H,
                             int D, int Lq, int n_levels, const int *level_hw,
                             const int *level_points, float offset_scale, int acc_mode,
                             float *fx_state, float hit_bound, void *stream);

/* dst[i] (f32 or bf16, n elements, n even) = word i of an acc_mode 2 / 3 accumulator / scale(fx_state). */
int dfine_cast_scaled_acc(const void *src, void *dst, int acc_mode, int dtype, int64_t n,
                          const float *fx_state, void *stream);

/* f32 -> bf16 (round to nearest even) copy of n elements; used to hand the f32-accumulated
 * grad_value back in the model's compute dtype. */
int dfine_cast_f32_to_bf16(const float *src, void *dst, int64_t n, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A11 + A12  Hungarian matcher: per-image cost block + linear sum assignment on the device.
 * Replaces HungarianMatcher.forward (src/d_fine/matcher.py:110-257): sigmoid/focal class cost,
 * cdist(p=1), generalized_box_iou, nan_to_num, C.cpu() and scipy.optimize.
 * linear_sum_assignment (scipy 1.15.x rectangular_lsap, float64, its tie-breaking rules).
 * K prediction heads are matched against the same targets in one launch.
 *
 *   logits [K, B, Q, C] f32; boxes [K, B, Q, 4] f32 cxcywh
 *   tgt_labels [T] i64; tgt_boxes [T, 4] f32; tgt_offset [B+1] i32 (prefix sums of the
 *   per-image target counts, T = tgt_offset[B]); Tmax = max count
 *   extra_cost [K, B, Tmax, Q] f32 or NULL (added before the NaN clean-up; mask costs)
 *   cost_out   [K, B, Tmax, Q] f32 workspace ("target-major": one target's costs against all
 *              queries are contiguous); on return holds the cost blocks (rows t >= T_b undefined)
 *   lsap_ws    workspace of dfine_match_ws_bytes(K,B,Q,Tmax) bytes (may be unused)
 *   match_out  [K, T] i32: query assigned to each target, -1 if unassigned (T_b > Q)
 *   T_total = T = tgt_offset[B] (passed by value: tgt_offset lives on the device)
 */
int64_t dfine_match_ws_bytes(int K, int B, int Q, int Tmax);
int dfine_match(const float *logits, const float *boxes, const int64_t *tgt_labels,
                const float *tgt_boxes, const int *tgt_offset, const float *extra_cost,
                float *cost_out, void *lsap_ws, int *match_out, int K, int B, int Q, int C,
                int Tmax, int T_total, float w_class, float w_bbox, float w_giou, float alpha, float gamma,
                void *stream);

/* Assignment only, on caller-provided cost blocks (same layout as cost_out above). */
int dfine_lsap(const float *cost, const int *tgt_offset, void *lsap_ws, int *match_out, int K,
               int B, int Q, int Tmax, int T_total, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A1/A2  Depthwise k x k convolution (groups == channels), NCHW.
 * Replaces nn.Conv2d(groups=C) inside LightConvBNAct.conv2 / HG_Stage.downsample
 * (src/d_fine/arch/hgnetv2.py:96-105,295-303) and SCDown.cv2 (src/d_fine/arch/hybrid_encoder.py:
 * 96-103), forward and both gradients (ATen: MIOpen naive_conv / per-image im2col+GEMM).
 *   x [B, C, H, W] dtype; w [C, 1, K, K] f32 (the fp32 master weights); y [B, C, OH, OW] dtype,
 *   OH = (H + 2*pad - K)/stride + 1.  K <= 7.
 *   bwd: dx (dtype, may be NULL), dw_f32 [C,1,K,K] f32 ZERO-FILLED by the caller (may be NULL; K in
 *   {1,3,5,7}).
 */
int dfine_dwconv_fwd(const void *x, const float *w, void *y, int dtype, int B, int C, int H, int W,
                     int K, int stride, int pad, void *stream);
/* The depthwise unit of LightConvBNAct in eval mode (src/d_fine/arch/hgnetv2.py:83-112: depthwise conv -> BatchNorm -> ReLU [-> LAB]) as
 * ONE launch: a one-shot request consumed by the next dfine_dwconv_fwd of the calling thread,
 * y = lab[0] * act(scale[c] * conv + shift[c]) + lab[1] on the fp32 sums before the store (scale / shift from dfine_bn_fold;
 * act 0 none / 1 ReLU / 2 SiLU; lab 2 floats or NULL; scale == NULL withdraws it).  Served where dfine_dwconv_affine_supported() != 0. */
int dfine_dwconv_affine_once(const float *scale, const float *shift, const float *lab, int act);
int dfine_dwconv_affine_supported(int dtype, int H, int W, int K, int stride, int pad);
int dfine_dwconv_bwd(const void *x, const float *w, const void *dy, void *dx, float *dw_f32,
                     int dtype, int B, int C, int H, int W, int K, int stride, int pad, void *stream);
/* dx += the data gradient of the K = 3 / stride 2 / pad 1 layer (bf16, H even, W % 8 == 0, W <= 320; DFINE_E_BADARG otherwise):
 * dx holds the gradient of the map's other consumer (HG_Stage.downsample reads a stage output that also leaves the backbone,
 * src/d_fine/arch/hgnetv2.py:295-303,520-526). */
int dfine_dwconv_s2_dgrad_acc(const float *w, const void *dy, void *dx, int B, int C, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A1/A2  Fused BatchNorm2d (+ activation) (+ LearnableAffineBlock), NCHW.
 * Replaces the bn -> act -> lab chain of ConvBNAct.forward (src/d_fine/arch/hgnetv2.py:25-32,
 * 75-80) and norm -> act of ConvNormLayer(_fuse).forward (src/d_fine/arch/hybrid_encoder.py:
 * 40-45,92-93), train (batch statistics, running-stat update with `momentum`, unbiased running
 * variance) and eval (running statistics) modes.
 *   y = lab_scale * act(gamma * (x - mean) * invstd + beta) + lab_bias
 *   act: 0 none, 1 ReLU, 2 SiLU.  x, y [B, C, HW] dtype; gamma/beta/running_* [C] f32 (gamma/beta
 *   may be NULL = 1/0); lab_scale/lab_bias scalars on the device or NULL.
 *   save_mean/save_invstd/scale/shift [C] f32 outputs (scale/shift: folded per-channel affine, also
 *   needed by the backward); ws: dfine_bn_ws_floats(B,C,HW) floats.
 *   bwd: dgamma/dbeta [C] f32 (may be NULL); dlab [2] f32 ZERO-FILLED (d lab_scale, d lab_bias) or
 *   NULL; in eval mode pass the running mean and rsqrt(running_var + eps) as save_mean/save_invstd (the eval-mode forward
 *   writes exactly those into its save_mean / save_invstd outputs when they are given).
 */
int64_t dfine_bn_ws_floats(int B, int C, int HW);
int dfine_bn_act_fwd(const void *x, void *y, const float *gamma, const float *beta,
                     float *running_mean, float *running_var, const float *lab_scale,
                     const float *lab_bias, float *save_mean, float *save_invstd, float *scale,
                     float *shift, float *ws, int dtype, int B, int C, int HW, int act, int training,
                     float momentum, float eps, void *stream);
/* One-shot: the NEXT dfine_bn_act_fwd of the calling thread stores y = unit(x) + res (res bf16 [B, C, HW]; unit(x) rounded to bf16
 * first - the sum a separate add of the two maps gives).  The residual connection of HG_Block behind the aggregation's second
 * unit (src/d_fine/arch/hgnetv2.py:274-275).  bf16, HW % 8 == 0; a launch that cannot serve it returns DFINE_E_BADARG. */
int dfine_bn_residual_once(const void *res);
int dfine_bn_act_bwd(const void *x, const void *dy, void *dx, const float *save_mean,
                     const float *save_invstd, const float *scale, const float *shift,
                     const float *lab_scale, float *dgamma, float *dbeta, float *dlab, float *ws,
                     int dtype, int B, int C, int HW, int act, int training, void *stream);



/* RepVGG unit: y = act(BN_a(x1) + BN_b(x2)) [+ residual] in one op (training mode, bf16 NCHW, H*W % 8 == 0).
 * Replaces VGGBlock.forward's two ConvNormLayer BatchNorms + add + activation and CSPLayer's residual add
 * (src/d_fine/arch/hybrid_encoder.py:106-156, 209-239).  `saved` [8][C] f32 receives mean, invstd, scale, shift of BN_a
 * then of BN_b; running statistics are updated like nn.BatchNorm2d (momentum, unbiased variance).
 * act: DFINE_ACT_* as in dfine_bn_act_fwd.  ws: dfine_bn2_ws_floats() floats.  dfine_bn2_supported() != 0 tells whether
 * the shape is handled (otherwise compose the unit from dfine_bn_act_fwd calls). */
int dfine_bn2_supported(int B, int C, int HW);
int64_t dfine_bn2_ws_floats(int B, int C, int HW);
int dfine_bn2_act_fwd(const void *x1, const void *x2, const void *residual, void *y, const float *gamma1,
                      const float *beta1, float *running_mean1, float *running_var1, const float *gamma2,
                      const float *beta2, float *running_mean2, float *running_var2, float *saved, float *ws,
                      int B, int C, int HW, int act, float momentum1, float eps1, float momentum2, float eps2,
                      void *stream);
/* Backward: dx1, dx2 (bf16) and the four parameter gradients [C] f32 (any of them may be NULL); the gradient of
 * `residual` is dy itself. */
int dfine_bn2_act_bwd(const void *x1, const void *x2, const void *dy, void *dx1, void *dx2, const float *saved,
                      float *dgamma1, float *dbeta1, float *dgamma2, float *dbeta2, float *ws, int B, int C, int HW,
                      int act, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A13/A14  All set-criterion losses of ONE prediction head, values and gradients in one call.
 * Replaces DFINECriterion.loss_labels_vfl / loss_boxes / loss_local (+ unimodal_distribution_
 * focal_loss, bbox2distance, translate_gt, box_iou, generalized_box_iou):
 * src/d_fine/dfine_criterion.py:92-237,837-858; src/d_fine/arch/utils.py:12-51,267-354.
 *
 *   logits  [B,Q,C] dtype, strided view (element strides l_sb, l_sq; inner stride 1)
 *   boxes   [B,Q,4] f32 view (cxcywh);  corners [B,Q,4*(reg_max+1)] dtype view or NULL;
 *   ref [B,Q,4] f32 view (FDR reference boxes);  teacher_corners / teacher_logits dtype views or
 *   NULL (no distillation term);  reg_max must be 32 when corners are given.
 *   cls_plan / box_plan: i64 [3, M] = (image, query, row of the batch-concatenated targets) of the
 *   matching used for the classification term resp. the box / local terms (the "GO" union);
 *   tgt_labels [T] i64, tgt_boxes [T,4] f32;  wtable [reg_max+1] HOST array W(n).
 *   s_vfl, s_l1, s_giou, s_fgl: final scale of each term (loss weight / normaliser);
 *   ddf_c_pos / ddf_c_neg: coefficient of a matched / unmatched edge row
 *        = w_ddf * sqrt(n) / ((sqrt(n_pos)+sqrt(n_neg)) * rows)   (dfine_criterion.py:223-235).
 *   outputs: out[5] = {vfl, l1, giou, fgl, ddf} (scaled);  grad_logits [B,Q,C] dtype;
 *   grad_l1, grad_giou [B,Q,4] f32;  grad_corners_fgl, grad_corners_ddf [B,Q,4*33] dtype.
 *   scratch: iou_cls [M_cls], iou_box [M_box] f32; map_cls, map_box [B*Q] i32; wrow [B*Q] f32.
 *   The call zero-fills out, grad_l1, grad_giou, map_cls, map_box and grad_corners_fgl itself; when the caller lays them out
 *   back to back - [out (8 floats) | grad_l1 | grad_giou | map_cls | map_box | pad to 16 bytes | grad_corners_fgl] - that is
 *   ONE fill instead of six.
 */
int dfine_head_losses(
    const void *logits, int64_t l_sb, int64_t l_sq, const float *boxes, int64_t b_sb, int64_t b_sq,
    const void *corners, int64_t c_sb, int64_t c_sq, const float *ref, int64_t r_sb, int64_t r_sq,
    const void *teacher_corners, int64_t tc_sb, int64_t tc_sq, const void *teacher_logits,
    int64_t tl_sb, int64_t tl_sq, const int64_t *cls_plan, int M_cls, const int64_t *box_plan,
    int M_box, const int64_t *tgt_labels, const float *tgt_boxes, const float *wtable, int reg_max,
    float reg_scale, float alpha, float gamma, float temp, float s_vfl, float s_l1, float s_giou,
    float s_fgl, float ddf_c_pos, float ddf_c_neg, void *grad_logits, float *grad_l1,
    float *grad_giou, void *grad_corners_fgl, void *grad_corners_ddf, float *iou_cls, float *iou_box,
    int *map_cls, int *map_box, float *wrow, float *out, int dtype, int B, int Q, int C,
    void *stream);

/* One-shot: the NEXT dfine_head_losses / dfine_head_losses_dev call of the calling thread finds its packed output block
 * [out (8 floats) | grad_l1 | grad_giou | map_cls | map_box | pad to 16 B | grad_corners_fgl] already zero and skips its own fill:
 * the caller cleared the blocks of all heads of a step (11 for D-FINE-m: dfine_criterion.py:609-777 runs its losses per head)
 * with one fill of a common arena.  A call whose buffers are not laid out that way returns DFINE_E_BADARG. */
int dfine_head_losses_prezeroed_once(void);
/* Backward of dfine_head_losses (reference: autograd through loss_labels_vfl / loss_boxes / loss_local,
 * src/d_fine/dfine_criterion.py:92-237): scales the gradients the forward call left behind by the upstream gradient
 * g[5] (device, f32) of its `out` vector, in place, one launch:
 *   grad_logits[n_logits] (dtype) *= g[0];   grad_l1[n_box] = grad_l1 * g[1] + grad_giou * g[2];
 *   grad_corners_fgl[n_corners] (dtype) = grad_corners_fgl * g[3] + grad_corners_ddf * g[4]   (ddf NULL: the first term;
 *   n_corners 0: a head without the local losses). */
int dfine_head_grads_scale(const float *g, void *grad_logits, int64_t n_logits, float *grad_l1,
                           const float *grad_giou, int64_t n_box, void *grad_corners_fgl,
                           const void *grad_corners_ddf, int64_t n_corners, int dtype, void *stream);

/* The launch group of dfine_head_losses with its six scalar factors (s_vfl, s_l1, s_giou, s_fgl, ddf_c_pos, ddf_c_neg) read from
 * DEVICE memory (`scales`, one row of dfine_criterion_scales' table) and - when `box_count` is not NULL - the number of valid
 * entries of the box plan read from device memory as well (M_box is then the row stride / capacity of the [3, M_box] plan).
 * Lets the criterion run without the host round trip of the matching (src/d_fine/dfine_criterion.py:619-652). */
int dfine_head_losses_dev(
    const void *logits, int64_t l_sb, int64_t l_sq, const float *boxes, int64_t b_sb, int64_t b_sq,
    const void *corners, int64_t c_sb, int64_t c_sq, const float *ref, int64_t r_sb, int64_t r_sq,
    const void *teacher_corners, int64_t tc_sb, int64_t tc_sq, const void *teacher_logits,
    int64_t tl_sb, int64_t tl_sq, const int64_t *cls_plan, int M_cls, const int64_t *box_plan,
    int M_box, const int64_t *tgt_labels, const float *tgt_boxes, const float *wtable, int reg_max,
    float reg_scale, float alpha, float gamma, float temp, const float *scales, const int *box_count,
    void *grad_logits, float *grad_l1, float *grad_giou, void *grad_corners_fgl,
    void *grad_corners_ddf, float *iou_cls, float *iou_box, int *map_cls, int *map_box, float *wrow,
    float *out, int dtype, int B, int Q, int C, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A13  Index bookkeeping of the criterion on the device (csrc/plans.hip): replaces the host side of
 * DFINECriterion.forward between the matcher and the loss terms - the per-head (batch, query, target) index lists
 * (_get_src_permutation_idx, src/d_fine/dfine_criterion.py:558-562), the GO union of all matchings (_get_go_indices, :570-591:
 * torch.unique(dim=0, return_counts=True) + torch.argsort(counts, descending=True) on the CPU + first pair per query) and
 * num_boxes_go (:634-641) - so that the train step has no host <-> device synchronisation.
 *   cols        int32 [K, T]   query assigned to target row t by head k (dfine_match / dfine_lsap), every entry >= 0
 *   tgt_offset  int32 [B + 1]  first target row of every image
 *   head_plans  int64 [K, 3, T] out: (image, query, target row) of head k, in target order
 *   go_plan     int64 [3, cap] out: the GO union, image-major, within an image in the reference's order; cap >= K * T
 *   go_count    int32 [1] out: its length;  go_count_f float [1] out or NULL: the same as a float (operand of a collective)
 *   ws          int32 [dfine_criterion_plans_ws_ints(K, T, B)] scratch
 * dfine_criterion_plans_supported: K * tmax <= 4096, Q <= 4096, tmax <= Q (every target matched). */
int dfine_criterion_plans_supported(int K, int tmax, int Q);
int64_t dfine_criterion_plans_ws_ints(int K, int T, int B);
int dfine_criterion_plans(const int *cols, const int *tgt_offset, int K, int T, int B, int Q, int tmax, int64_t *head_plans,
                          int64_t *go_plan, int cap, int *go_count, float *go_count_f, int *ws, void *stream);
/* The scalar factors of R head-loss launches from the GO size (the Python arithmetic of src/d_fine/dfine_criterion.py:
 * 639-652 normalisers, :213-235 DDF balance, in double / the float32 clamp division of the reference):
 *   params double [R, 12] = (s_vfl, box normaliser is num_boxes_go (1) or params[2] (0), n_box, w_bbox, w_giou, w_fgl, w_ddf,
 *                            has teacher, is denoising head, 4 * B * Q, 4 * matched pairs of a denoising head, 8 / B)
 *   go_sum float [1] or NULL: GO size summed over the ranks (world > 1);  scales float [R, 6] out. */
int dfine_criterion_scales(const double *params, int R, const int *go_count, const float *go_sum, int world, float *scales,
                           void *stream);


/* ---------------------------------------------------------------------------------------------
 * A16  Optimizer step on flat fp32 buffers (parameters are views into them).
 * Replaces clip_grad_norm_ + AdamW.step + zero_grad + ModelEMA.update
 * (src/dl/train.py:52-73,512-535; parameter groups: src/d_fine/dfine.py:87-124).
 *   dfine_grad_sqnorm      out[0] = grad_scale^2 * sum(grad^2)   (all groups, then sqrt in the step kernel:
 *                          global L2 norm of the averaged gradient).  Deterministic (fixed grid, fixed
 *                          summation order): data-parallel ranks must agree on the clip coefficient bit
 *                          for bit.  out: dfine_grad_sqnorm_ws_floats() floats (result + scratch).
 *   dfine_adamw_ema_step   one parameter group: g' = grad * grad_scale * min(1, max_norm /
 *                          (sqrt(sqnorm) + 1e-6)) (max_norm <= 0 or sqnorm NULL: no clipping);
 *                          AdamW update (torch.optim.AdamW semantics, `step` = 1-based step count);
 *                          ema = ema * m + (1 - m) * param (ema may be NULL); grad := 0.
 *   dfine_ema_update       ema = ema * m + (1 - m) * src   (BatchNorm statistics buffers)
 */
int64_t dfine_grad_sqnorm_ws_floats(void);
int dfine_grad_sqnorm(const float *grad, int64_t n, float grad_scale, float *out, void *stream);
int dfine_adamw_ema_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, float *ema,
                         int64_t n, const float *sqnorm, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int step, float grad_scale, float max_norm,
                         float ema_momentum, void *stream);
int dfine_ema_update(float *ema, const float *src, int64_t n, float momentum, void *stream);
/* Gather of many fp32 tensors into one flat buffer (per-parameter gradients -> flat gradient buffer):
 * table = DEVICE array of n_entries records {const float *src; int64_t dst_offset; int64_t count}
 * (24 bytes each, one block per record; split large tensors into <= 64 K element records). */
int dfine_multi_copy_f32(const void *table, int n_entries, float *dst, void *stream);
/* Same records, dst[dst_offset + i] += src[i]: the gradient tensors a captured backward segment (HIP graph of backbone +
 * encoder, custom_d_fine_amd/dl/engine.py) leaves are ADDED to the flat gradient buffer inside the graph - what
 * AccumulateGrad does per parameter in src/dl/train.py:570-575 (`loss.backward()`), also across the micro-steps of a
 * gradient-accumulation window.  Records of one launch must not overlap in dst. */
int dfine_multi_add_f32(const void *table, int n_entries, float *dst, void *stream);
/* dst = srcs[0] + ... + srcs[n - 1] (HOST array of 2 <= n <= 8 device pointers, fp32, count % 4 == 0, 16-byte aligned; dst may be
 * one of the sources): the gradient sum autograd forms pairwise for a decoder token-stream tensor with several consumers
 * (src/d_fine/arch/dfine_decoder.py:214-255: every LayerNorm output feeds 2 - 3 sub-layers / the residual path / the heads). */
int dfine_sum_f32(const void *const *srcs, int n, float *dst, int64_t count, void *stream);
/* bf16 shadow copies of the fp32 master weights (what autocast's per-call casts of nn.Linear / nn.Conv2d weights
 * produce, the modules of src/d_fine/arch under torch.autocast in src/dl/train.py:524-531) refreshed once per optimizer step:
 * device table of {const float *src; bf16 *dst; int64 n} records. */
int dfine_multi_cast_bf16(const void *table, int n_entries, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A1/A2  Dense 1x1 / 3x3 stride-1 "same" convolution on the MFMA units (NCHW, bf16, fp32 accumulate).
 * Replaces nn.Conv2d in ConvBNAct / ConvNormLayer(_fuse) / VGGBlock (src/d_fine/arch/hgnetv2.py:
 * 35-80, src/d_fine/arch/hybrid_encoder.py:21-156) for the forward pass and the data gradient.
 *   dfine_conv_pack_weights: fp32 [Cout, Cin, KS, KS] -> bf16 [KS*KS][NP][KP] (NP = n rounded up to
 *     16, KP = k rounded up to 32, zero padded; dfine_conv_packed_elems gives the element count).
 *     dgrad = 0: (n, k) = (Cout, Cin).  dgrad = 1: (n, k) = (Cin, Cout), taps flipped - feeding these to
 *     dfine_conv_fwd_bf16 with Cin/Cout exchanged yields dX from dY.
 *   dfine_conv_fwd_bf16: y [B, Cout, H, W] = conv(x [B, Cin, H, W]); KS in {1, 3}; Cin even; for
 *     KS = 3: W even and <= 160.
 */
int64_t dfine_conv_packed_elems(int Cout, int Cin, int KS, int dgrad);
int dfine_conv_pack_weights(const float *w, void *w2, int Cout, int Cin, int KS, int dgrad, void *stream);
/* All layers of a step in one launch: device table int64 [n_entries][8] = {w, w2, Cout, Cin, KS, NP, KP, dgrad}
 * (NP = n rounded up to 16, KP = k rounded up to 32 with (n, k) = dgrad ? (Cin, Cout) : (Cout, Cin)). */
int dfine_conv_pack_weights_multi(const void *table, int n_entries, void *stream);
int dfine_conv_fwd_bf16(const void *x, const void *w2, void *y, int B, int Cin, int Cout, int H, int W,
                        int KS, void *stream);

/* y += conv1x1(x) in bf16 (stride 1; same packed weights): the second data gradient of a RepVGG unit
 * (conv3x3 and conv1x1 on the same input, src/d_fine/arch/hybrid_encoder.py:106-156) accumulates onto the first.
 * DFINE_E_BADARG for shapes outside the LDS-DMA 1x1 kernel (H*W % 8 != 0, Cin % 4 != 0). */
int dfine_conv1x1_accum_bf16(const void *x, const void *w2, void *y, int B, int Cin, int Cout, int HW,
                             void *stream);

/* y += conv(x) with the packed weights of dfine_conv_fwd_bf16 (1x1 or 3x3, stride 1): a data gradient added onto the one
 * already in y - the sum autograd forms for a map with two consumers (HG_Block: layer i output -> layer i + 1 and the
 * aggregation, src/d_fine/arch/hgnetv2.py:265-274) - in the convolution's epilogue instead of an element-wise add pass.
 * dfine_conv_epilogue_supported() != 0 says whether the shape is served (1x1 on the LDS-DMA kernel, 3x3 on the
 * wave-specialised kernel); DFINE_E_BADARG otherwise. */
int dfine_conv_epilogue_supported(int B, int Cin, int Cout, int H, int W, int KS);
/* conv -> eval-mode BatchNorm (or the bias of a deployed layer) -> activation [-> learnable affine] as ONE launch
 * (src/d_fine/arch/hgnetv2.py:35-80 ConvBNAct in eval mode, hybrid_encoder.py:21-79 ConvNormLayer_fuse after convert_to_deploy()):
 * a one-shot request consumed by the next dfine_conv_fwd_bf16 / dfine_conv1x1_seg_fwd_bf16 of the calling thread,
 *     y[n] = lab[0] * act(scale[n] * conv[n] + shift[n]) + lab[1]      (act 0 none / 1 ReLU / 2 SiLU; lab: 2 floats or NULL)
 * on the fp32 accumulators in the kernel's store phase.  scale == NULL withdraws a pending request.  Served on the shapes
 * dfine_conv_affine_supported() accepts; a forward launch that cannot returns DFINE_E_BADARG and drops the request.
 * dfine_bn_fold: scale / shift of an eval-mode nn.BatchNorm2d / FrozenBatchNorm2d (gamma, beta may be NULL). */
int dfine_conv_affine_once(const float *scale, const float *shift, const float *lab, int act);
int dfine_conv_affine_supported(int B, int Cin, int Cout, int H, int W, int KS);
int dfine_bn_fold(const float *gamma, const float *beta, const float *running_mean, const float *running_var, float eps, int C,
                  float *scale, float *shift, void *stream);
int dfine_conv_accum_bf16(const void *x, const void *w2, void *y, int B, int Cin, int Cout, int H, int W, int KS,
                          void *stream);
/* Weight gradient of the same convolution: dw [Cout, Cin, KS, KS] f32 (overwritten) from x [B,Cin,H,W]
 * and dy [B,Cout,H,W] (bf16); ws = dfine_conv_wgrad_ws_floats(...) floats of scratch (split-K
 * partial sums).  KS = 3: W % 8 == 0 and W <= 160.  KS = 1: (H*W) % 8 == 0. */
int64_t dfine_conv_wgrad_ws_floats(int B, int Cin, int Cout, int H, int W, int KS);
int dfine_conv_wgrad_bf16(const void *x, const void *dy, float *dw, float *ws, int B, int Cin, int Cout,
                          int H, int W, int KS, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A8  Fine-grained distribution refinement head: Integral + distance2bbox + LQE statistics.
 * Replaces Integral.forward, distance2bbox and the softmax/top-k/mean half of LQE.forward
 * (src/d_fine/arch/dfine_decoder.py:291-295,307-311; src/d_fine/arch/utils.py:119-142).
 *   corners [N, 4*(reg_max+1)] dtype (N = B*Lq rows, contiguous); ref [N, 4] f32 cxcywh (detached
 *   reference boxes); wtable [reg_max+1] HOST array W(n); reg_max = 32, K = 4.
 *   fwd: boxes [N, 4] f32 cxcywh; stat [N, 4*(K+1)] f32 (per edge: top-K bin probabilities descending,
 *        then their mean); top_idx [N*4*K] u8 (saved for backward).
 *   bwd: g_boxes [N,4] f32 or NULL, g_stat [N, 4*(K+1)] f32 or NULL -> g_corners [N, 4*(reg_max+1)] dtype.
 */
int dfine_fdr_fwd(const void *corners, const float *ref, const float *wtable, float reg_scale, float *boxes,
                  float *stat, uint8_t *top_idx, int dtype, int N, int reg_max, int K, void *stream);
int dfine_fdr_bwd(const void *corners, const float *ref, const float *wtable, float reg_scale,
                  const float *g_boxes, const float *g_stat, const uint8_t *top_idx, void *g_corners,
                  int dtype, int N, int reg_max, int K, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A3  Query selection: score = max over classes, top-K anchors per image (descending score; ties by
 * ascending index).  Replaces torch.topk(outputs_logits.max(-1).values, K) of
 * DFINETransformer._select_topk (src/d_fine/arch/dfine_decoder.py:875-910).
 *   logits [B, Q, C] dtype with element strides (sb, sq) and unit class stride; out_idx [B, K] i64;
 *   out_score [B, K] f32 or NULL.  Q <= 16384, K <= min(Q, 1024).
 */
int dfine_topk_anchors(const void *logits, int64_t sb, int64_t sq, int64_t *out_idx, float *out_score,
                       int dtype, int B, int Q, int C, int K, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A1 / A2  1x1 convolution over a channel-wise concatenation that is never materialised: the HG_Block
 * aggregation conv (torch.cat of the block's maps, src/d_fine/arch/hgnetv2.py:265-274), RepNCSPELAN4.cv4
 * and the FPN / PAN fusion inputs (src/d_fine/arch/hybrid_encoder.py:196-206,460-486).
 * x_parts / y_parts: HOST arrays of n device pointers, part k = [B, channels[k], H, W] bf16 whose images are
 * bstrides[k] * H * W elements apart (bstrides NULL: contiguous parts; larger: a channel slice of a wider
 * tensor), sum(channels) = Cin / Cout; the forward entry point with a segmented OUTPUT and repacked weights
 * (dfine_conv_pack_weights dgrad = 1, Cin / Cout exchanged) is the data gradient written straight into one
 * tensor per concatenated input.  (H * W) % 8 == 0, at most 8 parts.
 */
int dfine_conv1x1_seg_fwd_bf16(const void *const *x_parts, const int *x_channels, const int *x_bstrides, int n_x,
                               const void *w2, void *const *y_parts, const int *y_channels, const int *y_bstrides,
                               int n_y, int B, int Cin, int Cout, int H, int W, void *stream);
/* y_parts += the same convolution (shapes of the LDS-DMA kernel only: DFINE_E_BADARG otherwise): the data gradient of a unit
 * that reads a channel slice of a wider map, added onto that slice of the map's gradient (RepNCSPELAN4's split,
 * src/d_fine/arch/hybrid_encoder.py:196-206). */
int dfine_conv1x1_seg_accum_bf16(const void *const *x_parts, const int *x_channels, const int *x_bstrides, int n_x,
                                 const void *w2, void *const *y_parts, const int *y_channels, const int *y_bstrides,
                                 int n_y, int B, int Cin, int Cout, int H, int W, void *stream);
/* ... added onto the output parts whose bit is set in accum_parts, the other parts are overwritten (HG_Block's aggregation data
 * gradient onto the residual connection's gradient of the block input, src/d_fine/arch/hgnetv2.py:265-275). */
int dfine_conv1x1_seg_accum_parts_bf16(const void *const *x_parts, const int *x_channels, const int *x_bstrides, int n_x,
                                       const void *w2, void *const *y_parts, const int *y_channels, const int *y_bstrides,
                                       int n_y, unsigned accum_parts, int B, int Cin, int Cout, int H, int W, void *stream);
int dfine_conv1x1_seg_wgrad_bf16(const void *const *x_parts, const int *x_channels, const int *x_bstrides, int n_x,
                                 const void *dy, float *dw, float *ws, int B, int Cin, int Cout, int H, int W,
                                 void *stream);

/* ---------------------------------------------------------------------------------------------
 * A16 / A17  Deferred weight-gradient reduction.  dfine_conv_wgrad_bf16, dfine_conv1x1_seg_wgrad_bf16 and
 * dfine_linear_wgrad_bf16 called with dw == NULL leave their per-split partial sums in `ws`
 * ([splits][NP16][CP16][taps] f32, NP16 / CP16 = Cout / Cin rounded up to 16; the linear entry point also leaves
 * [splits][NP16] bias partials behind them); dfine_*_wgrad_splits report `splits`.  dfine_multi_wgrad_reduce sums
 * the partials of many layers in ONE launch and ACCUMULATES into their destinations - the slots of the flat
 * gradient buffer the fused optimizer / all-reduce work on (the per-parameter `.grad` tensors and the
 * per-layer reductions of autograd's AccumulateGrad, src/dl/train.py:512-535, disappear).
 *   table: device int64 [n_entries][8] = {partials ptr, dst ptr, splits, Cout, Cin, taps, NP16, CP16}
 *   (bias gradient: Cin = taps = CP16 = 1).  max_blocks: the largest dfine_multi_wgrad_reduce_blocks(splits,
 *   Cout * Cin * taps) over the rows (the launch is max_blocks x n_entries workgroups).
 */
int dfine_conv_wgrad_splits(int B, int Cin, int Cout, int H, int W, int KS);
int dfine_linear_wgrad_splits(int M, int N, int K);
/* Many linear weight gradients (partial sums only, as dfine_linear_wgrad_bf16 with dw == NULL) in one launch: the backward ops
 * of the token-stream linears (arch/dfine_decoder.py:33-46,214-271, arch/hybrid_encoder.py:243-290) register their (x, dY)
 * pairs, a flush runs them together.  dfine_linear_wgrad_group_row fills a host row of 8 int64 and returns its workgroup
 * count; table = the rows on the device, max_blocks = the largest count. */
int dfine_linear_wgrad_group_row(const void *x, const void *dy, float *ws, int M, int N, int K, int64_t *row);
int dfine_linear_wgrad_group(const void *table, int n_problems, int max_blocks, void *stream);
/* The same for the 1x1 convolution weight gradients with whole-tensor inputs (partial sums as dfine_conv_wgrad_bf16 with
 * dw == NULL, KS = 1, H * W % 8 == 0): arch/hgnetv2.py:35-80, arch/hybrid_encoder.py:21-156. */
/* Splits / workspace floats of ONE problem of the grouped launch (fewer splits than a stand-alone launch of the same shape:
 * the group fills the chip, and unused splits are partial sums that are neither written nor reduced). */
int dfine_conv_wgrad1_group_splits(int B, int Cin, int Cout, int HW);
int64_t dfine_conv_wgrad1_group_ws_floats(int B, int Cin, int Cout, int HW);
int dfine_conv_wgrad1_group_row(const void *x, const void *dy, float *ws, int B, int Cin, int Cout, int HW, int64_t *row);
int dfine_conv_wgrad1_group(const void *table, int n_problems, int max_blocks, void *stream);
int dfine_multi_wgrad_reduce_blocks(int splits, int64_t elems);   /* blocks one row of the table needs */
int dfine_multi_wgrad_reduce(const void *table, int n_entries, int max_blocks, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A2 / A3 / A5 / A6  Token-stream linear layers: y[M, N] = act(x[M, K] . w[N, K]^T + bias[N]), bf16
 * operands (row strides ldx / ldw / ldy elements, unit inner stride), fp32 accumulate, bias fp32 or NULL,
 * act: 0 none, 1 ReLU, 2 GELU (erf), 3 SiLU; out_f32 = 1 writes fp32.  Replaces F.linear + activation of
 * MLP / FFN / Gate / in- and out-projections / enc_output / score and box heads
 * (src/d_fine/arch/dfine_decoder.py:33-46,119-178,214-271,828-873; src/d_fine/arch/hybrid_encoder.py:243-290).
 * The data gradient dX = dY . W is the same entry point on a transposed bf16 copy of the weight
 * (dfine_multi_cast_bf16_t: table rows {src fp32 [rows, cols] ptr, dst bf16 [cols, rows] ptr, rows, cols},
 * all shadows of a model in one launch); the weight / bias gradients are dfine_linear_wgrad_bf16.
 * dfine_act_fwd_bf16 / dfine_act_bwd_bf16: y = act(z) and d_pre = dy * act'(ref) (ref = saved output for
 * ReLU, saved pre-activation for GELU / SiLU), n % 8 == 0 elements.  These two also take act 4 = clamp(z, -10, 10), the
 * clamp of the decoder's query position embedding (src/d_fine/arch/dfine_decoder.py:466; ref = saved input, the gradient
 * passes where -10 <= z <= 10).
 */
int dfine_linear_act_fwd(const void *x, const void *w, const float *bias, void *y, int M, int N, int K, int ldx,
                         int ldw, int ldy, int act, int out_f32, void *stream);
/* dX [M, N] bf16 = (dY [M, K] . W [N, K]^T) where aux[m][n] > 0, else 0 (aux bf16 [M, N], row stride ldy): the data gradient of the
 * layer BEHIND a Linear + ReLU with that ReLU's backward in the store epilogue - the reference runs threshold_backward as its own
 * pass between the layers of MLP / FFN (src/d_fine/arch/dfine_decoder.py:33-46,214-231).  Same values as the two-pass form (the
 * mask is applied before the one rounding to bf16). */
int dfine_linear_dgrad_relu(const void *x, const void *w, const void *aux, void *y, int M, int N, int K, int ldx, int ldw,
                            int ldy, void *stream);
int dfine_multi_cast_bf16_t(const void *table, int n_entries, void *stream);
int dfine_act_fwd_bf16(const void *z, void *y, int64_t n, int act, void *stream);
int dfine_act_bwd_bf16(const void *dy, const void *ref, void *out, int64_t n, int act, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A2 / A6  Multi-head self-attention core, head_dim 32: o = softmax(q k^T * scale + mask) v per (batch,
 * head).  q, k, v, o (and the gradients) are [B, L, H * 32] bf16 views with row strides ld* (elements,
 * multiples of 8); mask uint8 [L, L] (non-zero = blocked) or NULL; lse2, delta fp32 [B, H, L] (softmax
 * statistics in the exp2 domain / rowsum(dO * O), written by fwd / bwd).  Replaces the
 * scaled_dot_product_attention inside nn.MultiheadAttention (src/d_fine/arch/hybrid_encoder.py:256,277,
 * src/d_fine/arch/dfine_decoder.py:200,239).
 */
int dfine_attn_fwd(const void *q, const void *k, const void *v, void *o, float *lse2, const uint8_t *mask,
                   int B, int L, int H, int hd, int ldq, int ldk, int ldv, int ldo, float scale, void *stream);
/* mask_bits (optional, with mask): the transposed bit-packed mask made by dfine_attn_mask_bits - dfine_attn_mask_bits_words(L)
 * uint32 words, bit j of word [key][w] = mask[32 w + j][key] - which gives the dK / dV kernel the 32 queries of a chunk in one
 * load per key instead of 32 byte loads (98 -> see profiles: the byte mask doubled that kernel's time). */
/* Tile summaries of the mask (free / mixed / blocked per 16 queries x 64 keys and per 64 keys x 32 queries): with them
 * (dfine_attn_fwd_ms / dfine_attn_bwd_ms) blocked tiles are skipped and free tiles run without the mask loads - the decoder's
 * denoising mask (src/d_fine/arch/utils.py:442-455) is block-structured.  Results are bit-identical to the plain entry points. */
int64_t dfine_attn_mask_summary_bytes(int L);
int dfine_attn_mask_summary(const uint8_t *mask, int L, uint8_t *sum, void *stream);
int dfine_attn_fwd_ms(const void *q, const void *k, const void *v, void *o, float *lse2, const uint8_t *mask,
                      const uint8_t *mask_summary, int B, int L, int H, int hd, int ldq, int ldk, int ldv, int ldo,
                      float scale, void *stream);
int dfine_attn_bwd_ms(const void *q, const void *k, const void *v, const void *o, const void *dout, const float *lse2,
                      const uint8_t *mask, const uint32_t *mask_bits, const uint8_t *mask_summary, void *dq, void *dk,
                      void *dv, float *delta, int B, int L, int H, int hd, int ldq, int ldk, int ldv, int ldo, int lddo,
                      int lddq, int lddk, int lddv, float scale, void *stream);
int64_t dfine_attn_mask_bits_words(int L);
int dfine_attn_mask_bits(const uint8_t *mask, int L, uint32_t *bits, void *stream);
int dfine_attn_bwd(const void *q, const void *k, const void *v, const void *o, const void *dout,
                   const float *lse2, const uint8_t *mask, const uint32_t *mask_bits, void *dq, void *dk, void *dv,
                   float *delta, int B, int L, int H, int hd, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq,
                   int lddk, int lddv, float scale, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A1 / A2 in fp32 (BASELINE configs[1]): dense convolutions on the f32-input matrix cores (v_mfma_f32_16x16x4_f32, exact
 * fp32 arithmetic) - the layers of dfine_conv_fwd_bf16 plus the HGNetv2 stem (3x3 stride 2, 2x2 on the bottom/right padded
 * map: hgnetv2.py:115-166) when the model runs without autocast.  Kernel size 1..3, stride 1 / 2, padding (pt, pl) on the
 * top / left, zeros outside the input.
 */
int64_t dfine_conv_f32_packed_elems(int Cout, int Cin, int KS, int dgrad);
/* fp32 master [Cout, Cin, KS, KS] -> [KS*KS][NP][KP] fp32 (NP = rows rounded up to 64, KP = k rounded up to 16); dgrad = 1:
 * rows = input channels, k = output channels, taps flipped (the data gradient is the forward kernel on this packing). */
int dfine_conv_f32_pack_weights(const float *w, float *w2, int Cout, int Cin, int KS, int dgrad, void *stream);
int dfine_conv_f32_fwd(const float *x, const float *w2, float *y, int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo,
                       int KS, int S, int pt, int pl, void *stream);
/* part [splits][NP16][CP16][KS*KS] f32 partial sums of the weight gradient (splits = dfine_conv_f32_wgrad_splits). */
int dfine_conv_f32_wgrad_splits(int B, int Cin, int Cout, int Ho, int Wo, int KS);
int dfine_conv_f32_wgrad(const float *x, const float *dy, float *part, int B, int Cin, int Cout, int Hi, int Wi, int Ho,
                         int Wo, int KS, int S, int pt, int pl, void *stream);
/* out [planes, Ho, Wo] = in [planes, H, W] with zeros inserted between the pixels (out[2y, 2x] = in[y, x]): the operand of
 * the data gradient of a stride-2 convolution.  Ho >= 2 H - 1, Wo >= 2 W - 1. */
int dfine_upsample2_zero_f32(const float *in, float *out, int64_t planes, int H, int W, int Ho, int Wo, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A10 / A15  Segmentation head (BASELINE configs[4]): the element-wise / reduction kernels of MaskDecoder
 * (src/d_fine/arch/dfine_decoder.py:316-370), of the mask losses (src/d_fine/dfine_criterion.py:335-450,504-556) and of
 * the matcher's mask costs (src/d_fine/matcher.py:19-71,175-237).  dtype: activations f32 or bf16; parameters, statistics
 * and loss terms f32.  The dense contractions of the head run on the convolution entry points (dfine_conv_fwd_bf16,
 * dfine_conv1x1_bw_bf16 for the per-image mask-logit einsum).
 */
/* y = [relu](GroupNorm_G(x) * gamma + beta), nn.GroupNorm(G, C) semantics (biased variance, eps inside the root).
 * x, y [B, C, HW]; stat [B, G, 2] f32 out = (mean, rstd), kept for the backward; ws: dfine_groupnorm_ws_floats floats. */
int64_t dfine_groupnorm_ws_floats(int B, int C, int G);
int dfine_groupnorm_fwd(const void *x, void *y, const float *gamma, const float *beta, float *stat, float *ws,
                        int dtype, int B, int C, int HW, int G, float eps, int relu, void *stream);
/* dx [B, C, HW]; part [B, C, 2] f32 out = per-plane (sum dz, sum dz * xhat) with dz = dy [* relu'] - the caller sums
 * them over B into d(beta) / d(gamma); ws: 2 * B * G floats. */
int dfine_groupnorm_bwd(const void *x, const void *dy, void *dx, const float *gamma, const float *beta,
                        const float *stat, float *part, float *ws, int dtype, int B, int C, int HW, int G, int relu,
                        void *stream);
/* y [planes, Ho, Wo] = [base +] bilinear resize of x [planes, Hi, Wi], align_corners = False
 * (F.interpolate(mode="bilinear")); base (NULL, or a map of the output shape, may alias y): MaskDecoder's upsample-sum of
 * the lateral maps.  _bwd: the adjoint of the resize, as a gather. */
int dfine_bilinear_fwd(const void *x, const void *base, void *y, int dtype, int planes, int Hi, int Wi, int Ho, int Wo,
                       void *stream);
int dfine_bilinear_bwd(const void *dy, void *dx, int dtype, int planes, int Hi, int Wi, int Ho, int Wo, void *stream);
/* Cropped BCE + Dice of M matched masks read in place from pm [B, Q, H, W] through (plan_b, plan_q) [M] i64;
 * tgt [rows, H, W] f32 and boxes [rows, 4] f32 (x1, y1, x2, y2 in mask pixels: pixel (x, y) counts iff x1 <= x < x2,
 * y1 <= y < y2), row of match m = plan_t[m] (the batch-concatenated target index) or m when plan_t is NULL.
 * sums [M, 4] f32 out = (BCE-with-logits, p * t, p, t) summed inside the box, p = sigmoid(logit). */
int dfine_mask_loss_sums(const void *pm, const int64_t *plan_b, const int64_t *plan_q, const int64_t *plan_t,
                         const float *tgt, const float *boxes, float *sums, int dtype, int M, int Q, int H, int W,
                         void *stream);
/* grad [B, Q, H, W] dtype (zero-filled by the caller; the M matched planes are written): per pixel inside the box
 * coef[m][0] * (p - t) + (coef[m][1] * t + coef[m][2]) * p * (1 - p), 0 outside.  coef [M, 3] f32 from the caller. */
int dfine_mask_loss_grad(const void *pm, const int64_t *plan_b, const int64_t *plan_q, const int64_t *plan_t,
                         const float *tgt, const float *boxes, const float *coef, void *grad, int dtype, int M, int Q,
                         int H, int W, void *stream);
/* Pairwise mask-cost sums of the LAST Q of the Qall queries of every image against its targets: gt [sum T, HW] f32
 * (concatenated over the batch), toff [B + 1] i32.  out [B, Q, Tmax, 2] f32 = (sum_p sigmoid(x) g, sum_p (pos - neg)(x) g)
 * with the focal terms pos = alpha (1 - p)^gamma (-log(p + 1e-8)), neg = (1 - alpha) p^gamma (-log(1 - p + 1e-8));
 * qsum [B, Q, 2] f32 = (sum_p sigmoid(x), sum_p neg(x)).  Entries t >= T_b of out are not written. */
int dfine_mask_cost(const void *pm, const float *gt, const int *toff, float *out, float *qsum, int dtype, int B,
                    int Qall, int Q, int HW, int Tmax, float alpha, float gamma, void *stream);
/* y[b] = conv1x1(x[b], W_b), one weight set per image: w2 [B][NP][KP] bf16 in the dfine_conv_pack_weights(KS = 1) layout
 * per image.  einsum("bqc,bchw->bqhw") of DFINETransformer._mask_logits_from_h (dfine_decoder.py:925-932) with
 * Cout = queries, Cin = mask_dim, and its gradient w.r.t. the mask features (weights = embeddings transposed).
 * (H*W) % 8 == 0, Cin % 4 == 0. */
int dfine_conv1x1_bw_bf16(const void *x, const void *w2, void *y, int B, int Cin, int Cout, int HW, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A18  Detection post-processor: sigmoid -> top-K over the Q*C (query, class) scores of every image ->
 * label = idx % C, query = idx // C -> normalised cxcywh -> absolute xyxy (floor / ceil + clamp when
 * to_round).  Replaces DFINEPostProcessor.forward (src/dl/export.py:61-100, box arithmetic :35-59) and
 * the identical top-K block of Trainer.preds_postprocess (src/dl/train.py:262-277) and
 * Torch_model._preds_postprocess (src/infer/torch_model.py:197-214).
 *   logits [B, Q, C] dtype, boxes [B, Q, 4] f32 ->
 *   labels [B, K] i64, query_idx [B, K] i64, out_boxes [B, K, 4] f32, scores [B, K] f32 (descending;
 *   ties: larger logit, then lower flat index; NaN logits rank first, as in torch.topk).  K <= min(Q*C, 4096);
 *   Q*C <= 32768 and K <= 1024 keep the keys in registers, larger problems (365 classes) re-read them from L2 per pass.
 */
int dfine_postprocess(const void *logits, const float *boxes, int64_t *labels, int64_t *query_idx,
                      float *out_boxes, float *scores, int dtype, int B, int Q, int C, int K, int height,
                      int width, int to_round, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (f3) Device-side data path, geometric augmentation of training samples (reference: host OpenCV / numpy in
 * CustomDataset._load_mosaic src/dl/dataset.py:258-377, random_affine / get_mosaic_coordinate src/dl/utils.py:325-414).
 * Images are uint8 HWC on the device.
 */
/* canvas [Hc, Wc, 3] region [ly1, ly2) x [lx1, lx2) <- cv2.resize(src [Hs, Ws, 3], (rw, rh), INTER_LINEAR) cropped from
 * (sx1, sy1): one mosaic quadrant (dataset.py:275-290). */
int dfine_mosaic_place_u8(const uint8_t *src, uint8_t *canvas, int Hs, int Ws, int rh, int rw, int Hc, int Wc, int lx1,
                          int ly1, int lx2, int ly2, int sx1, int sy1, void *stream);
/* dst [Hd, Wd, 3] = cv2.warpAffine(src [Hs, Ws, 3], m (forward 2 x 3, HOST doubles), dsize, INTER_LINEAR, constant border)
 * (utils.py:339-341); OpenCV's fixed-point arithmetic restated, parity unpinned (cv2 absent from the build image). */
int dfine_warp_affine_u8(const uint8_t *src, uint8_t *dst, int Hs, int Ws, int Hd, int Wd, const double *m, int border,
                         void *stream);
/* boxes [N, 4] xyxy -> out [N, 4]: the four corners through m (2 x 3, HOST floats), min / max, clip to
 * [0, target_w] x [0, target_h]; keep [N] u8 = box_candidates(box1 = boxes * scale, box2 = out, area_thr) (utils.py:343-377,283-295). */
int dfine_affine_boxes(const float *boxes, float *out, uint8_t *keep, int N, const float *m, float scale, float target_w,
                       float target_h, float area_thr, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (f1) Inference pre-processing: uint8 [B, Hs, Ws, 3] BGR frames (device) -> [B, 3, Ho, Wo] dtype, RGB / 255.
 * The source is bilinearly resized to (rh, rw) with OpenCV's 8-bit INTER_LINEAR arithmetic, placed at
 * (top, left) and surrounded by pad_value (letterbox: 114); plain resize: rh = Ho, rw = Wo, top = left = 0.
 * Replaces Torch_model._preprocess / _prepare_inputs + letterbox (src/infer/torch_model.py:240-298,378-418:
 * cv2.resize + copyMakeBorder on the host, flip / transpose in numpy, .float().div_(255) on the device).
 */
int dfine_preprocess_u8(const uint8_t *src, void *dst, int dtype, int B, int Hs, int Ws, int Ho, int Wo, int rh,
                        int rw, int top, int left, int pad_value, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A5/A6  Weight gradient of a token-stream nn.Linear: dw [N, K] f32 = dy [M, N]^T x [M, K] (bf16,
 * row-major; M = B*Lq rows), split over the M reduction (the autograd formula of F.linear used by
 * MLP / FFN / Gate / attention projections, src/d_fine/arch/dfine_decoder.py:33-46,214-271).
 *   db [N] f32 or NULL: bias gradient (column sums of dy), produced by one extra MFMA per k-step.
 *   ws: dfine_linear_wgrad_ws_floats(M, N, K) floats.
 */
int64_t dfine_linear_wgrad_ws_floats(int M, int N, int K);
int dfine_linear_wgrad_bf16(const void *x, const void *dy, float *dw, float *db, float *ws, int M, int N,
                            int K, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A1  HGNetv2 stem (src/d_fine/arch/hgnetv2.py:115-166: stem1 3x3/s2, F.pad + stem2a / stem2b 2x2,
 * MaxPool2d(2, 1, ceil_mode) on the padded map, stem3 3x3/s2, stem4 1x1): direct small-channel
 * convolutions in NCHW bf16 with fp32 accumulation.  Reads outside the plane return 0, which is both the
 * symmetric conv padding and the F.pad(x, (0, 1, 0, 1)) in front of the 2x2 layers.
 *   dfine_stem_supported      1 when (Cin, Cout, KS, stride) is an instantiated forward configuration.
 *   dfine_stem_pack_weights   fp32 master weights [Cout,Cin,KS,KS] -> wp (same element count).
 *                             mode 0: forward [(ci,ky,kx)][co]; mode 1: stride-1 data gradient
 *                             [(co,ky,kx) flipped][ci]; mode 2: stride-2 data gradient [(co,ky,kx)][ci].
 *   dfine_stem_conv_bf16      y [B,Cout,Ho,Wo] = conv(x [B,Cin,H,W]); with mode-1 weights, Cin/Cout
 *                             exchanged and pad' = KS-1-pad it is the data gradient of a stride-1 layer.
 *   dfine_stem_dgrad_s2_bf16  data gradient of a 3x3 / stride 2 / pad 1 layer (dx [B,Cin,2Ho,2Wo]).
 *   dfine_stem_wgrad_bf16     dw [Cout,Cin,KS,KS] f32 (overwritten) on the MFMA units; Wo % 32 == 0,
 *                             Cout <= 32; ws: dfine_stem_wgrad_ws_floats(...) floats.
 *   dfine_stem_pool_fwd/_bwd  2x2 / stride 1 max-pool over the map padded by one zero row and column
 *                             (bottom / right); the backward pass recomputes the argmax (first maximum in
 *                             scan order, as ATen's max_pool2d) from x.  planes = B * C.
 */
int dfine_stem_supported(int Cin, int Cout, int KS, int stride);
int dfine_stem_pack_weights(const float *w, float *wp, int Cout, int Cin, int KS, int mode, void *stream);
int dfine_stem_conv_bf16(const void *x, const float *wp, void *y, int B, int Cin, int Cout, int H, int W,
                         int Ho, int Wo, int KS, int stride, int pad, void *stream);
int dfine_stem_dgrad_s2_bf16(const void *dy, const float *wq, void *dx, int B, int Cin, int Cout, int Ho,
                             int Wo, void *stream);
int64_t dfine_stem_wgrad_ws_floats(int B, int Cin, int Cout, int KS, int Ho, int Wo);
int dfine_stem_wgrad_bf16(const void *x, const void *dy, float *dw, float *ws, int B, int Cin, int Cout,
                          int H, int W, int Ho, int Wo, int KS, int stride, int pad, void *stream);
/* The three stem operators with the input given as TWO tensors xa [B, Ca, H, W], xb [B, Cin - Ca, H, W] that the reference
 * concatenates along the channels first (StemBlock.forward: torch.cat([pool(stem1), stem2b]) -> stem3, hgnetv2.py:158-165);
 * the data gradient comes back as two contiguous tensors. */
int dfine_stem_conv2_bf16(const void *xa, const void *xb, int Ca, const float *wp, void *y, int B, int Cin, int Cout,
                          int H, int W, int Ho, int Wo, int KS, int stride, int pad, void *stream);
int dfine_stem_dgrad_s2_2_bf16(const void *dy, const float *wq, void *dxa, void *dxb, int Ca, int B, int Cin, int Cout,
                               int Ho, int Wo, void *stream);
int dfine_stem_wgrad2_bf16(const void *xa, const void *xb, int Ca, const void *dy, float *dw, float *ws, int B, int Cin,
                           int Cout, int H, int W, int Ho, int Wo, int KS, int stride, int pad, void *stream);
int dfine_stem_pool_fwd(const void *x, void *y, int64_t planes, int H, int W, void *stream);
int dfine_stem_pool_bwd(const void *x, const void *dy, void *dx, int64_t planes, int H, int W, void *stream);
/* dx += the pool's gradient (W % 8 == 0): dx already holds the gradient of the map's other consumer (stem2a) */
int dfine_stem_pool_bwd_acc(const void *x, const void *dy, void *dx, int64_t planes, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A5/A6  Residual / gate + LayerNorm of the token streams, one pass each way
 * (src/d_fine/arch/dfine_decoder.py:238-255 norm1 / norm3 of TransformerDecoderLayer.forward, :258-271 Gate.forward;
 *  src/d_fine/arch/hybrid_encoder.py:243-280 TransformerEncoderLayer).
 *   mode 0: z = a + b (b may be NULL)   mode 1: z = clamp(a + b, -clampv, clampv)
 *   mode 2: z = sigmoid(gate[:, :D]) * a + sigmoid(gate[:, D:]) * b      (gate [rows, 2 D])
 *   y [rows, D] f32 = (z - mean) * rstd * weight + bias;  mean / rstd [rows] f32 are kept for the backward.
 *   y_bf16 [rows, D] (may be NULL): the same values rounded to bf16, for the GEMMs that consume the stream next.
 *   a / b / gate: DFINE_F32 or DFINE_BF16 each (x_dt arguments), unit inner stride, row stride D (2 D for gate).
 *   D % 64 == 0, D <= 1024.  backward: dy f32; da / db / dgate in the inputs' storage types (NULL = not needed);
 *   dweight / dbias f32 [D] are overwritten (deterministic two-stage column sums through ws:
 *   dfine_ln_fused_bwd_ws_floats(rows, D) floats).
 */
int dfine_ln_fused_fwd(int mode, const void *a, int a_dt, const void *b, int b_dt, const void *gate, int g_dt,
                       const float *weight, const float *bias, float eps, float clampv, float *y, void *y_bf16,
                       float *mean, float *rstd, int64_t rows, int D, void *stream);
int dfine_ln_fused_bwd(int mode, const void *a, int a_dt, const void *b, int b_dt, const void *gate, int g_dt,
                       const float *weight, const float *mean, const float *rstd, const float *dy, float clampv,
                       void *da, void *db, void *dgate, float *dweight, float *dbias, float *ws, int64_t rows,
                       int D, void *stream);
int64_t dfine_ln_fused_bwd_ws_floats(int64_t rows, int D);

/* ---------------------------------------------------------------------------------------------
 * A3  Encoder maps <-> decoder token memory.  Replaces the flatten(2).permute(0, 2, 1) + concat of
 * DFINETransformer._get_encoder_input (src/d_fine/arch/dfine_decoder.py:778-801) and its autograd backward.
 *   map [B, C, HW] bf16 (NCHW level), tokens [B, L, C] bf16; the level occupies token rows [row0, row0 + HW).
 *   to_tokens != 0: tokens <- map;  to_tokens == 0: map <- tokens (the gradient direction).  C % 8 == 0, HW % 8 == 0.
 */
int dfine_maps_tokens_bf16(const void *map, void *tokens, int B, int C, int HW, int L, int row0, int to_tokens,
                           void *stream);

/* Nearest-neighbour 2x upsampling of the FPN top-down path, `F.interpolate(feat_heigh, scale_factor=2.0, mode="nearest")`
 * (src/d_fine/arch/hybrid_encoder.py:472), and its backward: x [planes, H, W], y [planes, 2H, 2W], bf16, W % 4 == 0.
 * backward = 0: y := upsample(x); backward = 1: x := sum of the 2 x 2 blocks of y (the gradient with respect to x). */
int dfine_upsample2_nearest_bf16(void *x, void *y, int64_t planes, int H, int W, int backward, void *stream);

/* Weight gradient of a small embedding table (A4: denoising_class_embed, src/d_fine/arch/utils.py:357-467 looks it up for every
 * denoising query; ATen's embedding_dense_backward sorts the lookups first): dw [rows, D] f32 (overwritten) = sum of the rows of
 * g [n, D] f32 whose idx [n] (int32 / int64: idx_bits = 32 / 64) names that row; padding_idx (-1: none) contributes nothing.  Deterministic (lookup order). */
int dfine_embedding_bwd(const float *g, const void *idx, int idx_bits, float *dw, int64_t n, int rows, int D, int padding_idx,
                        void *stream);

/* ---------------------------------------------------------------------------------------------
 * A2 / A5 / A6, fp32 (BASELINE config #2)  Token-stream GEMMs on the f32-input matrix cores.  Replaces the rocBLAS calls
 * behind nn.Linear (arch/dfine_decoder.py:33-46,119-178,214-271,828-873, arch/hybrid_encoder.py:243-290) and the two batched
 * products of F.scaled_dot_product_attention (hybrid_encoder.py:256,277, dfine_decoder.py:200,239) for fp32 tensors.
 *   C[z][M, N] (row stride ldc, z stride sc) = act(alpha * A[b][M, K] . B[b][N, K]^T + bias[N]),  z = b * splits + split;
 *   operands K-contiguous with row strides lda / ldb and batch strides sa / sb (elements; 0 = shared).  splits > 1 cuts K
 *   into chunks of `chunk` (multiple of 4): every z writes its own partial product (weight gradients: reduction over token rows).
 *   act: 0 none, 1 relu, 2 gelu (erf), 3 silu.
 */
int dfine_gemm_f32_nt(const float *A, const float *B, const float *bias, float *C, int batch, int M, int N, int K, int lda, int ldb,
                      int ldc, int64_t sa, int64_t sb, int64_t sc, int splits, int chunk, float alpha, int act, void *stream);
/* B given K-major ([K, N], rows N-contiguous): C[z] = act(alpha * A[b] . B[b] + bias).  The fp32 1x1 convolution on NCHW maps
 * (arch/hgnetv2.py:35-80, arch/hybrid_encoder.py:21-156): y[b] = W x[b], dx[b] = W^T dy[b], N = H * W. */
int dfine_gemm_f32_nn(const float *A, const float *B, const float *bias, float *C, int batch, int M, int N, int K, int lda, int ldb,
                      int ldc, int64_t sa, int64_t sb, int64_t sc, float alpha, int act, void *stream);

/* Bias gradient of an fp32 linear (ref src/d_fine/arch/dfine_decoder.py:33-46, nn.Linear backward): part [splits][N] = per-split column
 * sums of d [M, N] - or of dm = d * (relu_y > 0), written too, when relu_y is given.  N % 4 == 0, N <= 1024;
 * splits = dfine_colsum_f32_splits(M). */
int dfine_colsum_f32_splits(int M);
int dfine_colsum_f32(const float *d, const float *relu_y, float *dm, float *part, int M, int N, void *stream);
/* General form: a_kmajor / b_kmajor != 0 - that operand is stored K-major ([K, M] / [K, N], rows contiguous in M / N): the
 * products with a transposed first factor (linear weight gradient dY^T x, P^T dO, dS^T Q) without transposed copies. */
int dfine_gemm_f32(int a_kmajor, int b_kmajor, const float *A, const float *B, const float *bias, float *C, int batch, int M, int N, int K,
                   int lda, int ldb, int ldc, int64_t sa, int64_t sb, int64_t sc, int splits, int chunk, float alpha, int act, void *stream);

/* A4  Contrastive-denoising query group (src/d_fine/arch/utils.py:357-467, get_contrastive_denoising_training_group): padded class
 * ids with label noise and noised boxes in logit space for the 2 * groups * gmax denoising queries of every image, in ONE launch
 * (the reference: a Python loop per image and per group + ~40 element-wise launches).  labels int64 [T] / boxes fp32 [T, 4]: the
 * batch's targets concatenated; offsets int32 [bs + 1] (device): first target of every image.  The four random tensors are the
 * reference's draws, made by the caller in its order: flip_rand fp32 [bs, total] (rand_like), rnd_cls int32 [bs, total]
 * (randint_like 0 .. C - 1), sign01 fp32 [bs, total, 4] (randint_like 0 .. 1), mag fp32 [bs, total, 4] (rand_like); total =
 * 2 * groups * gmax.  flip_below = label_noise_ratio * 0.5.  Out: cls int32 [bs, total] (num_classes in padded slots),
 * box_unact fp32 [bs, total, 4] = inverse_sigmoid of the noised cxcywh box.  Bit-identical to the reference's fp32 op sequence. */
int dfine_cdn_group(const int64_t *labels, const float *boxes, const int *offsets, const float *flip_rand, const int *rnd_cls,
                    const float *sign01, const float *mag, int *cls_out, float *box_unact, int bs, int gmax, int groups,
                    int num_classes, float flip_below, float box_noise_scale, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (f2)  Instance-mask IoU of the evaluation hand-off.  Replaces Validator._pairwise_mask_iou
 * (src/dl/validator.py:283-293: uint8 masks -> fp32 matmul -> areas -> inter / union) and the pycocotools RLE round trip
 * the reference stores validation masks through (src/dl/utils.py:1040-1160) with 1-bit-per-pixel device masks.
 *   dfine_mask_bits_words(HW): uint64 words per packed mask.
 *   dfine_mask_pack_bits: masks [N, HW]; dtype 0 = uint8 (bit = value != 0), 1 = f32, 2 = bf16 (bit = value > thresh, the
 *     reference's `m > conf_thresh`); bits [N, words] uint64.  The bit order inside a 256-pixel chunk is private to the
 *     library (the same permutation for every mask).
 *   dfine_mask_iou_bits: iou [Np, Ng] f32 = |p & g| / |p | g| (0 where the union is empty), bit-identical to the reference's
 *     fp32 route for H * W < 2^24.
 */
int64_t dfine_mask_bits_words(int64_t HW);
int dfine_mask_pack_bits(const void *masks, int dtype, float thresh, int N, int64_t HW, void *bits, void *stream);
int dfine_mask_iou_bits(const void *pred_bits, const void *gt_bits, int Np, int Ng, int64_t words, float *iou, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFINE_HIP_H */
