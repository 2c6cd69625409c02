/*
 * semidetr_hip.h -- C ABI of libsemidetr_hip.so: the MI355X (gfx950) native hot path of Semi-DETR.
 *
 * Drop-in boundary: plain pointers + sizes, no torch types.  All pointers are DEVICE pointers unless
 * the parameter is documented as host.  `stream` is a hipStream_t passed as void* (NULL = the null
 * stream).  Every call is asynchronous on `stream`, keeps no global state between calls (apart from a
 * thread-local last-error string and the per-(device, call site) forward-kernel choice documented at
 * semidetr_msda_set_forward_policy, which also lists the one-time allocation it makes) and is re-entrant.  Return value: 0 on success, otherwise a
 * SEMIDETR_E_* code (negative: argument/precondition error detected on the host; positive: the
 * hipError_t returned by the launch).  semidetr_last_error() gives the message for the calling thread.
 *
 * Each entry point names the reference interface it replaces (paths relative to the Semi-DETR tree).
 */
#ifndef SEMIDETR_HIP_H
#define SEMIDETR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEMIDETR_OK 0
#define SEMIDETR_E_BADARG (-1)      /* null pointer / non-positive size / unsupported combination   */
#define SEMIDETR_E_TOOLARGE (-2)    /* an index would overflow the 32-bit arithmetic used on device  */
#define SEMIDETR_E_NODEVICE (-3)    /* no HIP device available                                        */

#define SEMIDETR_ABI_VERSION 7

int semidetr_abi_version(void);
const char *semidetr_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale deformable attention (MSDA).
 *
 * Replaces  ms_deformable_im2col_cuda   detr_od/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:923-954
 *           ms_deformable_col2im_cuda   detr_od/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:956-1327
 * i.e. the launchers behind ms_deform_attn_cuda_forward/backward (ms_deform_attn_cuda.cu:20-153), which
 * the pybind module MultiScaleDeformableAttention exports (src/vision.cpp:13-16).
 *
 * Layouts (contiguous, row-major), exactly the reference's:
 *   value          (batch, spatial_size, num_heads, channels)
 *   spatial_shapes (num_levels, 2) int64  [(H_l, W_l)]          -- device memory, read by the kernel
 *   level_start    (num_levels,)   int64                         -- device memory, read by the kernel
 *   sampling_loc   (batch, num_query, num_heads, num_levels, num_point, 2)   (x, y) in [0,1] coords
 *   attn_weight    (batch, num_query, num_heads, num_levels, num_point)
 *   out / grad_out (batch, num_query, num_heads * channels)
 * Semantics: bilinear sampling with align_corners=False pixel mapping (h = y*H - 0.5), zero padding,
 * a sample contributes only if -1 < h < H and -1 < w < W.  forward writes every element of `out`.
 * backward zero-fills grad_value itself (on `stream`: hipMemsetAsync, or as a side job of its first kernel),
 * accumulates into it with fp atomics, and writes every element of grad_sampling_loc / grad_attn_weight.
 * The whole batch is one launch (the reference's im2col_step chunking does not change results).
 * f32: fast path for channels == 32, generic path otherwise.  f64: generic path (gradcheck parity).
 *
 * `flags` (f32 entry points): SEMIDETR_MSDA_QUERIES_ARE_PIXELS tells the library that this is encoder
 * self-attention -- num_query == spatial_size, query i IS pixel i of the pyramid, and spatial_shapes /
 * level_start tile [0, spatial_size) exactly (level_start[l+1] == level_start[l] + H_l*W_l, sum H_l*W_l ==
 * spatial_size), with every H_l, W_l <= 32766.  The level table lives in device memory, so the library cannot verify this without a
 * host synchronisation: the CALLER vouches for it (the Python / pybind layer checks it once per
 * spatial_shapes tensor).  With the flag the forward / gather kernels take 2-D pixel patches and
 * grad_value is produced by the region-owned scatter kernel; without it every query set takes the strip
 * kernels, which make no assumption (the reference op has no such coupling).  Results are identical
 * either way (up to fp32 summation order); the flag only selects faster kernels.  With the flag the backward clears
 * grad_value inside its first kernel; grad_value of the encoder path is produced by the region-owned scatter.
 * ------------------------------------------------------------------------------------------- */
#define SEMIDETR_MSDA_QUERIES_ARE_PIXELS 1
/* Bits 8..15 of `flags`: the call site's slot of the forward-kernel choice (semidetr_msda_set_forward_policy below), 0..255.
 * The reference builds twelve MSDeformAttn instances per model (transformer.py:609,760); give each its own slot and its
 * encoder forward is chosen from ITS launches' sample spread.  Slot 0 (no bits set) is shared by every caller that names none. */
#define SEMIDETR_MSDA_POLICY_SLOT(s) (((s) & 0xff) << 8)
/* The forward kernel is to be a function of the arguments alone: with the adaptive policy WHICH of the two encoder forward
 * kernels runs depends on when earlier launches' counts reached the host, and the two sum a row's samples in different orders
 * (results agree to ~2e-6 absolute, not bit for bit) -- two passes over identical inputs may then differ in the last bits,
 * which the reference's single kernel never does.  With this flag the patch kernel runs, and the backward's small-gradient half is
 * the patch gather whatever the slot's counts say (the compiled front end sets it while
 * torch.use_deterministic_algorithms(True) is in force).  grad_value is accumulated with fp32 atomics either way, exactly as in
 * the reference (ms_deform_im2col_cuda.cuh:87-159): the BACKWARD's summation order is run-dependent there and here. */
#define SEMIDETR_MSDA_FIXED_FORWARD 2
/* Backward only (ABI 7): WHICH kernel computes grad_sampling_loc / grad_attn_weight of an encoder backward, stated by the caller instead of
 * read from the slot's live counts.  The two gathers agree to fp32 rounding, not bit for bit, and in the reference these two gradients
 * are deterministic (no atomics: ms_deform_im2col_cuda.cuh:301-403) -- so a caller that wants the backward to follow what was known when the
 * matching FORWARD ran (the autograd functions do: they ask semidetr_msda_gather_choice right after the forward launch and hand the answer
 * back here) sets one of the two.  Neither bit: the slot's state at the time of the backward decides (ABI <= 6 behaviour).
 * SEMIDETR_MSDA_FIXED_FORWARD still wins (patch gather). */
#define SEMIDETR_MSDA_GATHER_WINDOW (1 << 16)
#define SEMIDETR_MSDA_GATHER_PATCH (1 << 17)
int semidetr_msda_forward_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                              const int64_t *level_start, const float *sampling_loc,
                              const float *attn_weight, int batch, int spatial_size, int num_heads,
                              int channels, int num_levels, int num_query, int num_point, int flags, float *out);
int semidetr_msda_forward_f64(void *stream, const double *value, const int64_t *spatial_shapes,
                              const int64_t *level_start, const double *sampling_loc,
                              const double *attn_weight, int batch, int spatial_size, int num_heads,
                              int channels, int num_levels, int num_query, int num_point, double *out);
int semidetr_msda_backward_f32(void *stream, const float *grad_out, const float *value,
                               const int64_t *spatial_shapes, const int64_t *level_start,
                               const float *sampling_loc, const float *attn_weight, int batch,
                               int spatial_size, int num_heads, int channels, int num_levels,
                               int num_query, int num_point, int flags, float *grad_value,
                               float *grad_sampling_loc, float *grad_attn_weight);
int semidetr_msda_backward_f64(void *stream, const double *grad_out, const double *value,
                               const int64_t *spatial_shapes, const int64_t *level_start,
                               const double *sampling_loc, const double *attn_weight, int batch,
                               int spatial_size, int num_heads, int channels, int num_levels,
                               int num_query, int num_point, double *grad_value,
                               double *grad_sampling_loc, double *grad_attn_weight);

/* ---------------------------------------------------------------------------------------------
 * MSDA with the MSDeformAttn prologue / epilogue fused in (fp32, channels == 32).
 *
 * Replaces  MSDeformAttn.forward lines between the Linear layers
 *           detr_od/models/utils/ops/modules/ms_deform_attn.py:99-111  (softmax over L*P, sampling_locations =
 *           reference_points + offsets / (W_l, H_l)   [ref_dim 2]   or
 *           reference_points[:2] + offsets / P * reference_points[2:] * 0.5   [ref_dim 4])
 *           together with MSDeformAttnFunction.apply (:121-123) and their autograd backward.
 * The kernels read the RAW outputs of the two Linear layers, so the (N,Lq,M,L,P,2) locations tensor, the
 * softmaxed weights and their gradients never exist in HBM.
 *   reference_points  (batch, num_query, num_levels, ref_dim)       ref_dim 2 or 4; 16-byte aligned, < 4 GB (read with
 *                     one bounded 16-byte buffer load per (query, level))
 *   sampling_offsets  (batch, num_query, num_heads, num_levels, num_point, 2)   raw Linear output
 *   attn_logits       (batch, num_query, num_heads, num_levels * num_point)     raw Linear output (pre-softmax)
 *   padding_mask      (batch, spatial_size) bytes, nonzero = padded pixel, or NULL: `value.masked_fill(mask[..., None], 0)`
 *                     (ms_deform_attn.py:95-96) folded in -- a corner on a padded pixel reads as zero and receives no
 *                     gradient, so `value` is passed UNMASKED and grad_value comes back with zero rows there.
 *   mask_extents      (batch, num_levels) int32 words vh | vw << 16 written by semidetr_msda_mask_extents for THIS padding_mask
 *                     and level table, or NULL.  DETR's masks mark the band below / right of each image inside the batch canvas
 *                     (transformer.py:1268-1288), so "pixel (y, x) of level l is padding iff y >= vh or x >= vw" describes them
 *                     exactly; with it the kernels test a corner with two compares instead of a dependent byte load (which
 *                     cost +15 % on every kernel).  A level whose mask is not of that form carries -1 and its corners read
 *                     their bytes, as all corners do when mask_extents is NULL: the results are the mask's either way.
 *   grad_sampling_offsets / grad_attn_logits: same shapes, every element written.
 * (The gradient w.r.t. reference_points, when a caller needs it, follows from grad_sampling_offsets on the
 *  host side; the reference detaches reference points between decoder layers, transformer.py:1033.)
 * ------------------------------------------------------------------------------------------- */
int semidetr_msda_fused_forward_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                                    const int64_t *level_start, const float *reference_points, int ref_dim,
                                    const float *sampling_offsets, const float *attn_logits,
                                    const unsigned char *padding_mask, const int *mask_extents, int batch,
                                    int spatial_size, int num_heads, int channels, int num_levels,
                                    int num_query, int num_point, int flags, float *out);
int semidetr_msda_fused_backward_f32(void *stream, const float *grad_out, const float *value,
                                     const int64_t *spatial_shapes, const int64_t *level_start,
                                     const float *reference_points, int ref_dim,
                                     const float *sampling_offsets, const float *attn_logits,
                                     const unsigned char *padding_mask, const int *mask_extents, int batch,
                                     int spatial_size, int num_heads, int channels, int num_levels,
                                     int num_query, int num_point, int flags, float *grad_value,
                                     float *grad_sampling_offsets, float *grad_attn_logits);
/* Summarise a padding mask for the two calls above: one small launch (batch x num_levels workgroups, reads the mask once).
 *   extents (batch, num_levels) int32: vh | vw << 16 when level l of image n is padded exactly on rows >= vh and columns >= vw
 *   (no padding: H_l | W_l << 16; everything padded: 0), -1 otherwise (and for levels of more than 32767 rows / columns).  Valid for as long as the mask bytes and the
 *   level table do not change; the reference builds one mask per batch and hands it to all twelve layers
 *   (transformer.py:1309,1380), so one call per batch serves 12 forward + 12 backward launches. */
int semidetr_msda_mask_extents(void *stream, const unsigned char *padding_mask, const int64_t *spatial_shapes,
                               const int64_t *level_start, int batch, int spatial_size, int num_levels, int *extents);

/* ---------------------------------------------------------------------------------------------
 * Which kernel runs the encoder self-attention FORWARD (SEMIDETR_MSDA_QUERIES_ARE_PIXELS, num_point == 4, four or five levels;
 * with or without a padding mask).  The reference has one kernel for everything (ms_deform_im2col_cuda.cuh:237-299); here two
 * produce the same results at different speeds depending on how far the learned offsets reach:
 *   patch kernel   -- 4 x 8 query patches, every corner row through the vector-memory path; insensitive to the offsets
 *   window kernel  -- regions of up to 25 x 16 pixels, the coarse levels' corner rows from LDS windows +- 5 px (five levels: +- 4 px) around
 *                     the region; ~35 % faster while most samples stay inside (sigma <= 2 px), level with the patch kernel when
 *                     ~85 % of them are more than 4 px away (sigma ~8 px at four images, ~6 px for one)
 * policy 0 (default, adaptive): both kernels count, in a few workgroups, the share of samples further than 4 px from their
 *   query's pixel centre; launch k's count reaches the host through mapped pinned memory when launch k + 1 OF THE SAME SLOT starts
 *   (no copy command, no synchronisation) and the NEXT dispatch of that slot moves between the kernels with hysteresis (to the
 *   window kernel below 78 %, back above 86 %; five levels 82 % / 90 %).  State is per (device, slot) -- see
 *   SEMIDETR_MSDA_POLICY_SLOT; launches inside a stream capture keep their slot's kernel of the moment and count nothing.  The
 *   first adaptive dispatch on a device allocates the counters (64 KB of device memory, 4 KB of pinned host memory; the two
 *   allocation calls may synchronise that device once).  Launches of ONE slot on two streams of a device at the same time share
 *   its counter pair: the counts may mix, the results do not depend on them.  The choice makes the forward's last bits depend on
 *   timing: SEMIDETR_MSDA_FIXED_FORWARD (or policy 1) for bitwise reproducible forwards.
 *   LIFETIME of that state: the counters (64 KB of device memory + 4 KB of mapped pinned host memory per device) and the per-slot host words are
 *   allocated at a device's first adaptive dispatch and live until the process ends -- launches in flight reference them, so no call
 *   releases them (semidetr_msda_set_forward_policy only changes how dispatches READ them; a device reset invalidates them like any allocation).
 * policy 1: always the patch kernel.   policy 2: the window kernel whenever it applies.   (Process-wide.)
 * The encoder BACKWARD (four levels) reads the same slot: its small-gradient half runs as the lane-per-sample region-window gather
 *   (msda_gw_d32; whole backward 572 / 635 / 796 us against 659 / 707 / 834 us at sigma 1 / 2 / 3 px, bs 4) while the slot's last
 *   count has fewer than 55 % of the samples further than 4 px away (policy 2: always; policy 1 or no count yet: the patch gather) --
 *   or as SEMIDETR_MSDA_GATHER_WINDOW / _PATCH in the backward's `flags` say (the caller's record of the choice at forward time).
 *   grad_value's summation order is run-dependent either way (fp32 atomics); the two gathers agree to fp32 rounding.
 * If the device does not grant the window kernel its LDS (~150 KB per workgroup) the patch kernel runs instead.
 * semidetr_msda_forward_policy_state[_slot]: for the calling thread's current device and slot (0 without _slot) -- the policy,
 *   the kernel the adaptive policy stands on (0 patch, 1 window), the last far-sample fraction received (-1: none yet), the
 *   number of counts received.
 * ------------------------------------------------------------------------------------------- */
int semidetr_msda_set_forward_policy(int policy);
int semidetr_msda_forward_policy_state(int *policy, int *mode, float *far_fraction, unsigned *updates);
int semidetr_msda_forward_policy_state_slot(int slot, int *policy, int *mode, float *far_fraction, unsigned *updates);
/* What an encoder backward of `slot` issued NOW would pick for its small-gradient half, as the flag to pass later:
 * SEMIDETR_MSDA_GATHER_WINDOW or SEMIDETR_MSDA_GATHER_PATCH (never 0, never an error: an unknown slot / device answers PATCH). */
int semidetr_msda_gather_choice(int slot);

/* Names of the device kernels the LAST semidetr_msda_* call of the calling thread launched ("+"-separated, as the
 * profiler prints their base names), so that a benchmark reports what actually ran instead of a hand-kept table. */
const char *semidetr_msda_last_kernels(void);

/* ---------------------------------------------------------------------------------------------
 * Hungarian matcher: cost matrix + linear sum assignment + assignment scatter, batched and
 * device-resident (no host round trip inside).
 *
 * Replaces  HungarianAssigner.assign   thirdparty/mmdetection/mmdet/core/bbox/assigners/hungarian_assigner.py:55-188
 *           FocalLossCost/BBoxL1Cost/IoUCost.__call__  .../match_costs/match_cost.py:33-50,83-99,169-185
 *           bbox_overlaps(giou|iou)    .../iou_calculators/iou2d_calculator.py:200-261
 *           scipy.optimize.linear_sum_assignment (call sites hungarian_assigner.py:136,
 *           detr_ssod/models/dino_detr_ssod.py:279) -- third-party scipy, pinned at 1.15.3
 *           the inline twin of the above in DinoDetrSSOD.unsup_loss  detr_ssod/models/dino_detr_ssod.py:248-293
 *
 * A batch holds B independent problems; problem b has Q predictions (same Q for all) and
 * G_b = gt_offsets[b+1] - gt_offsets[b] ground truths (ragged, G_b may be 0).
 *   bbox_pred  (B, Q, 4) fp32  cx,cy,w,h normalised
 *   cls_pred   (B, Q, C) fp32  logits
 *   gt_bboxes  (sumG, 4) fp32  x1,y1,x2,y2 pixels ; gt_labels (sumG,) int64
 *   gt_offsets (B+1,) int32 DEVICE ; img_wh (B, 2) fp32 DEVICE (img_w, img_h)
 *   cost       (Q * sumG) fp32: problem b's matrix is TRANSPOSED-CONTIGUOUS, i.e. element (q, g) lives at
 *              cost[Q*gt_offsets[b] + g*Q + q]  (a (G_b, Q) row-major block; the (Q, G_b) matrix the
 *              reference builds is its transpose view).
 * ------------------------------------------------------------------------------------------- */
typedef struct semidetr_cost_params {
    float cls_weight;   /* FocalLossCost.weight  (DINO config: 2.0)  */
    float alpha;        /* 0.25 */
    float gamma;        /* 2.0  */
    float eps;          /* 1e-12 */
    float reg_weight;   /* BBoxL1Cost.weight (5.0) */
    int   reg_xywh;     /* 1: box_format='xywh', 0: 'xyxy' */
    float iou_weight;   /* IoUCost.weight (2.0) */
    int   iou_giou;     /* 1: 'giou', 0: 'iou' */
    int   pred_xyxy;    /* 0: bbox_pred is cx,cy,w,h (HungarianAssigner.assign); 1: it already is x1,y1,x2,y2
                           (the stand-alone IoUCost.__call__ contract, match_cost.py:169-185) */
} semidetr_cost_params;

int semidetr_match_cost_f32(void *stream, const float *bbox_pred, const float *cls_pred,
                            const float *gt_bboxes, const int64_t *gt_labels,
                            const int32_t *gt_offsets, const float *img_wh, int num_problems,
                            int num_query, int num_classes, int total_gt,
                            const semidetr_cost_params *params /* host */, float *cost);

/* Solve all B assignment problems on the device (one wavefront per problem), bit-exact with
 * scipy.optimize.linear_sum_assignment on the same fp32 matrix (up-cast to fp64 as scipy does).
 *   cost as laid out above.
 *   match_row / match_col (sumK,) int64 with K_b = min(Q, G_b) pairs for problem b stored at
 *       pair_offsets[b] = sum_{b'<b} min(Q, G_b')  -- because Q is shared this is computed on device
 *       from gt_offsets; rows ascending (scipy's output order), cols = matched gt index in the problem.
 *       Either may be NULL.
 *   assigned_gt_inds / assigned_labels (B, Q) int64: hungarian_assigner.py:142-147 scatter
 *       (0 / -1 for unmatched; when G_b == 0: gt_inds = 0, labels = -1). Either may be NULL.
 *   status (B,) int32: 0 ok, 1 infeasible, 2 invalid numeric entries (NaN / -inf) -- scipy raises
 *       ValueError for those; the host wrapper turns them into the same exception.
 *   workspace: device scratch of semidetr_lsap_workspace_bytes(B, Q, maxG) bytes.
 */
int64_t semidetr_lsap_workspace_bytes(int num_problems, int num_query, int max_gt);
int semidetr_lsap_solve(void *stream, const float *cost, const int32_t *gt_offsets,
                        const int64_t *gt_labels, int num_problems, int num_query, int total_gt,
                        int max_gt, int64_t *match_row, int64_t *match_col,
                        int64_t *assigned_gt_inds, int64_t *assigned_labels, int32_t *status,
                        void *workspace);

/* Training targets of all problems from their assignment, one launch.
 * Replaces the tail of  _get_target_single   detr_od/models/dense_heads/dino_detr_ssod_head.py:1170-1205,
 *                                            detr_od/models/dense_heads/dino_detr_head.py:937-980
 *           with PseudoSampler               thirdparty/mmdetection/mmdet/core/bbox/samplers/pseudo_sampler.py:35-41
 *   assigned_gt_inds (B,Q) int64 from semidetr_lsap_solve; gt_bboxes / gt_labels / gt_offsets / img_wh as above.
 *   labels (B,Q) int64 = num_classes for background else the gt label; label_weights (B,Q) = 1;
 *   bbox_targets (B,Q,4) = cxcywh(gt / (w,h,w,h)) for positives else 0; bbox_weights (B,Q,4) = 1 / 0;
 *   num_pos (B,) int32 (zeroed inside the call). */
int semidetr_build_targets(void *stream, const int64_t *assigned_gt_inds, const float *gt_bboxes,
                           const int64_t *gt_labels, const int32_t *gt_offsets, const float *img_wh,
                           int num_problems, int num_query, int64_t num_classes, int64_t *labels,
                           float *label_weights, float *bbox_targets, float *bbox_weights, int32_t *num_pos);

/* ---------------------------------------------------------------------------------------------
 * Mean-teacher EMA, one launch for the whole parameter list.
 *
 * Replaces  MeanTeacher.momentum_update   detr_ssod/utils/hooks/mean_teacher.py:60-64
 *           (tgt.mul_(m).add_(src, alpha=1-m) per parameter => ~1000 launches per step).
 *   teacher_ptrs / student_ptrs (T,) device arrays of device pointers (float*), numels (T,) int64
 *   device array, block_starts (T+1,) int32 device array = exclusive prefix sum of
 *   ceil(numel / SEMIDETR_EMA_CHUNK) -- one workgroup processes one chunk.
 *   Arithmetic per element: t = rn(t * (float)m); t = fma(s, (float)(1-m), t)   (torch's rounding).
 * ------------------------------------------------------------------------------------------- */
#define SEMIDETR_EMA_CHUNK 8192
int semidetr_ema_multi_f32(void *stream, float *const *teacher_ptrs, const float *const *student_ptrs,
                           const int64_t *numels, const int32_t *block_starts, int num_tensors,
                           int total_blocks, double momentum);
/* Same arithmetic on one flat arena (parameters laid out contiguously in HBM). */
int semidetr_ema_flat_f32(void *stream, float *teacher, const float *student, int64_t numel,
                          double momentum);

/* ---------------------------------------------------------------------------------------------
 * Pseudo-label filter: per image mean+std score threshold, then drop empty boxes; one launch for
 * the batch.
 *
 * Replaces  the per-image loop in DinoDetrSSOD.extract_teacher_info  detr_ssod/models/dino_detr_ssod.py:918-939
 *   proposals (sumK, 5) fp32 x1,y1,x2,y2,score ; labels (sumK,) int64 ; prop_offsets (B+1,) int32 DEVICE:
 *   image b's proposals are rows [prop_offsets[b], prop_offsets[b+1]); with prop_counts (B,) int32 DEVICE
 *   (nullable) only the first prop_counts[b] of them exist -- the padded layout semidetr_pseudo_nms_f32 emits.
 *   out_boxes (sumK,4), out_labels (sumK,), out_scores (sumK,): image b's kept entries are written
 *   compacted, in the original order, starting at prop_offsets[b]; out_count (B,) int32 kept per image;
 *   out_thr (B,) fp32 the threshold used (NaN for <2 proposals, as torch.std gives).
 * ------------------------------------------------------------------------------------------- */
int semidetr_pseudo_label_filter_f32(void *stream, const float *proposals, const int64_t *labels,
                                     const int32_t *prop_offsets, const int32_t *prop_counts,
                                     int num_images, float *out_boxes,
                                     int64_t *out_labels, float *out_scores, int32_t *out_keep_idx,
                                     int32_t *out_count, float *out_thr);

/* ---------------------------------------------------------------------------------------------
 * Teacher test-time box decoding for pseudo labels, whole batch, no host
 * round trip: sigmoid, cxcywh -> clamped pixel xyxy, score threshold, class-aware greedy NMS, the
 * max_per_img best by score.
 *
 * Replaces  DINODETRSSODHead._get_bboxes_single(for_pseudo_label=True)
 *               detr_od/models/dense_heads/dino_detr_ssod_head.py:1364-1395 (called per image from :1320-1331)
 *           multiclass_nms  thirdparty/mmdetection/mmdet/core/post_processing/bbox_nms.py:8-95
 *           mmcv.ops.batched_nms / nms (mmcv-full 1.3.16, un-vendored: README.md:11,30)
 *   cls_logits (B,Q,C) fp32 raw logits of the last decoder layer; bbox_pred (B,Q,4) fp32 normalised cxcywh;
 *   img_hw (B,2) fp32 DEVICE = img_meta['img_shape'][:2] (height, width); Q <= 2048; 1 <= max_per_img <= 2048.
 *   workspace: semidetr_nms_workspace_bytes(B,Q,C) bytes of device memory, 16-byte aligned.
 *   out_dets (B,max_per_img,5) x1,y1,x2,y2,score; out_labels (B,max_per_img) int64; out_count (B,) int32:
 *   image b's detections are the first out_count[b] rows of its slice, by descending score (equal logits:
 *   ascending query*C + class; the reference's order of exact ties is unspecified).
 * ------------------------------------------------------------------------------------------- */
size_t semidetr_nms_workspace_bytes(int batch, int num_query, int num_classes);
int semidetr_pseudo_nms_f32(void *stream, const float *cls_logits, const float *bbox_pred, const float *img_hw,
                            int batch, int num_query, int num_classes, float score_thr, float iou_thr,
                            int max_per_img, void *workspace, size_t workspace_bytes, float *out_dets,
                            int64_t *out_labels, int32_t *out_count);

/* ---------------------------------------------------------------------------------------------
 * Weak -> strong augmentation warp of the pseudo boxes, one launch for the batch.
 *
 * Replaces  Transform2D.transform_bboxes  detr_ssod/models/utils/bbox_utils.py:167-192 (bbox2points :18-25,
 *           points2bbox :28-41), called through DinoDetrSSOD._transform_bbox  detr_ssod/models/dino_detr_ssod.py:804-807
 *   boxes: rows of box_stride (>= 4) floats, x1,y1,x2,y2 first; image b owns rows [box_offsets[b], +n_b) with
 *   n_b = box_counts[b] if box_counts else box_offsets[b+1] - box_offsets[b] (both int32 DEVICE arrays);
 *   max_boxes_per_image >= every n_b (grid sizing); matrices (B,3,3) row major fp32; out_hw (B,2) fp32 (h, w)
 *   of the target image; out_boxes rows of 4 floats at the same row indices.
 * ------------------------------------------------------------------------------------------- */
int semidetr_transform_bboxes_f32(void *stream, const float *boxes, int box_stride, const int32_t *box_offsets,
                                  const int32_t *box_counts, int num_images, int max_boxes_per_image,
                                  const float *matrices, const float *out_hw, float *out_boxes);

/* ---------------------------------------------------------------------------------------------
 * One-to-many (task-aligned) assigner of the warm-up stage + its training targets (SURVEY.md section 8(f)
 * row 4), all (layer, image) problems of a loss() call in one launch.
 *
 * Replaces  O2MAssigner.assign  detr_od/core/bbox/assigners/o2m_assigner.py:50-170 (teacher_assign=False, or
 *           teacher_assign=True with multiple_pos=False via candidate_topk = 1)
 *           the in_warm_up branch of DINODETRSSODHead._get_target_single
 *               detr_od/models/dense_heads/dino_detr_ssod_head.py:1108-1165
 * Batch layout as semidetr_match_cost_f32 (gt_offsets (B+1,) int32 DEVICE, img_wh (B,2) fp32 DEVICE = (w, h)),
 * but cls_prob (B,Q,C) holds PROBABILITIES (the call site passes cls_score.sigmoid(), head.py:1111).
 * max_gt_per_problem >= every G_b (<= 1024); Q <= 2048; candidate_topk <= Q unless total_gt == 0.
 * dynamic_k != 0 = the `teacher_assign and multiple_pos` option (o2m_assigner.py:125-133): per ground truth the first k_g of its
 * top-candidate_topk candidates are positive whatever their metric, k_g = max(1, int(sum of its candidate_topk largest IoUs)).
 * Outputs, all (B,Q[,4]):
 *   gt_inds int64 (0 background, g+1 positive), labels int64 (class of the gt or -1),
 *   max_overlaps fp32 (IoU with the assigned gt; -1e8 when unassigned; 0 for a problem without gts),
 *   assign_metrics fp32 (score^alpha * IoU^beta of the assigned pair, else 0),
 *   labels_full int64 (class or num_classes), bbox_targets fp32 (gt as normalised cx,cy,w,h, else 0),
 *   norm_metrics fp32 = metric / (max metric of the gt's positives + 10e-8) * max IoU of the gt's positives.
 * ------------------------------------------------------------------------------------------- */
int semidetr_o2m_assign_f32(void *stream, const float *bbox_pred, const float *cls_prob, const float *gt_bboxes,
                            const int64_t *gt_labels, const int32_t *gt_offsets, const float *img_wh,
                            int num_problems, int num_query, int num_classes, int total_gt,
                            int max_gt_per_problem, int candidate_topk, int dynamic_k, float alpha, float beta,
                            int64_t *gt_inds, int64_t *labels, float *max_overlaps, float *assign_metrics,
                            int64_t *labels_full, float *bbox_targets, float *norm_metrics);

/* ---------------------------------------------------------------------------------------------
 * Task-aligned focal loss of the warm-up stage, fused with the sigmoid in front of it; loss sum and (optionally)
 * its gradient w.r.t. the logits in one streaming pass, deterministic summation.
 *
 * Replaces  task_aigned_focal_loss / TaskAlignedFocalLoss  detr_od/models/losses/task_aligned_focal_loss.py:35-66,:166-200
 *           as called at  detr_od/models/dense_heads/dino_detr_ssod_head.py:693-694  (prob = cls_scores.sigmoid())
 *   logits (N,C) fp32 (input_is_prob != 0: they already are probabilities, the reference module's contract, and the
 *   gradient is d/d prob); labels (N,) int64 in [0, C] (C = background); metrics (N,) fp32 normalised alignment metrics;
 *   workspace: semidetr_tal_loss_workspace_bytes() bytes of device memory;
 *   loss_sum (1,) fp32 = sum_ic |s - p|^gamma * BCE(p, s)  (the caller divides by avg_factor / applies loss_weight);
 *   grad_logits (N,C) fp32 or NULL = d loss_sum / d logits.
 * ------------------------------------------------------------------------------------------- */
size_t semidetr_tal_loss_workspace_bytes(void);
int semidetr_tal_loss_f32(void *stream, const float *logits, const int64_t *labels, const float *metrics,
                          int64_t num_rows, int num_classes, float gamma, int input_is_prob, void *workspace,
                          float *loss_sum, float *grad_logits);

#ifdef __cplusplus
}
#endif
#endif /* SEMIDETR_HIP_H */
