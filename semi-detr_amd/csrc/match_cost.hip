// Batched Hungarian cost matrix for gfx950: focal classification cost + L1 box cost + (G)IoU cost,
// one launch for all (decoder layer x image) problems of a loss() call.
//
// Behavioural spec: thirdparty/mmdetection/mmdet/core/bbox/assigners/hungarian_assigner.py:115-129,
// .../match_costs/match_cost.py:33-50 (BBoxL1Cost), :83-99 (FocalLossCost), :169-185 (IoUCost),
// .../iou_calculators/iou2d_calculator.py:200-261 (bbox_overlaps), .../transforms.py:222-247.
// The reference builds the matrix with ~30 small elementwise launches per problem; here one thread owns
// one prediction, keeps its box in registers and walks the problem's ground truths.  The matrix is
// written TRANSPOSED ((G_b, Q) row-major, see semidetr_hip.h) so the stores are coalesced along Q and
// the assignment kernel -- which always solves the transposed problem when Q > G -- streams rows.
//
// fp32 arithmetic in the reference's evaluation order; contraction is off so the result is the same
// expression tree the CPU oracle evaluates (transcendentals differ by <= 1 ulp).
#include <hip/hip_runtime.h>

#include "common.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float pow_gamma(float x, float gamma)
{
    return gamma == 2.0f ? x * x : powf(x, gamma);   // torch.pow(x, 2) is x*x as well
}

__global__ __launch_bounds__(256) void match_cost_kernel(
    const float *__restrict__ bbox_pred, const float *__restrict__ cls_pred,
    const float *__restrict__ gt_bboxes, const int64_t *__restrict__ gt_labels,
    const int32_t *__restrict__ gt_offsets, const float *__restrict__ img_wh, int Q, int C,
    semidetr_cost_params prm, float *__restrict__ cost)
{
    const int b = blockIdx.y;
    const int g0 = gt_offsets[b], G = gt_offsets[b + 1] - g0;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (G <= 0 || q >= Q) return;
    const float img_w = img_wh[2 * b], img_h = img_wh[2 * b + 1];
    const float4 bp = *reinterpret_cast<const float4 *>(bbox_pred + ((int64_t)b * Q + q) * 4);
    float cx = bp.x, cy = bp.y, bw = bp.z, bh = bp.w;
    // bbox_cxcywh_to_xyxy then * factor
    float nx0 = cx - 0.5f * bw, ny0 = cy - 0.5f * bh, nx1 = cx + 0.5f * bw, ny1 = cy + 0.5f * bh;
    if (prm.pred_xyxy) {   // caller hands x1,y1,x2,y2 directly (IoUCost / BBoxL1Cost used on their own)
        nx0 = bp.x; ny0 = bp.y; nx1 = bp.z; ny1 = bp.w;
        cx = (nx0 + nx1) / 2; cy = (ny0 + ny1) / 2; bw = nx1 - nx0; bh = ny1 - ny0;
    }
    const float x0 = nx0 * img_w, y0 = ny0 * img_h, x1 = nx1 * img_w, y1 = ny1 * img_h;
    const float area1 = (x1 - x0) * (y1 - y0);
    const float *cls_row = cls_pred + ((int64_t)b * Q + q) * C;
    float *out = cost + (int64_t)Q * g0 + q;
    for (int j = 0; j < G; ++j) {
        const float4 gt = *reinterpret_cast<const float4 *>(gt_bboxes + (int64_t)(g0 + j) * 4);
        const int label = (int)gt_labels[g0 + j];
        // --- FocalLossCost
        // a label outside [0, C) is an index error in the reference (match_cost.py:96); here it must neither read out
        // of bounds nor pass silently: the cost becomes NaN, which the LSAP status word reports (ValueError)
        const float x = (unsigned)label < (unsigned)C ? cls_row[label] : __builtin_nanf("");
        const float p = 1.0f / (1.0f + expf(-x));
        const float neg = -logf(1.0f - p + prm.eps) * (1.0f - prm.alpha) * pow_gamma(p, prm.gamma);
        const float pos = -logf(p + prm.eps) * prm.alpha * pow_gamma(1.0f - p, prm.gamma);
        const float c_cls = (pos - neg) * prm.cls_weight;
        // --- BBoxL1Cost on normalised boxes
        const float n0 = gt.x / img_w, n1 = gt.y / img_h, n2 = gt.z / img_w, n3 = gt.w / img_h;
        float l1;
        if (prm.reg_xywh) {
            const float t0 = (n0 + n2) / 2, t1 = (n1 + n3) / 2, t2 = n2 - n0, t3 = n3 - n1;
            l1 = fabsf(cx - t0) + fabsf(cy - t1) + fabsf(bw - t2) + fabsf(bh - t3);
        } else {
            l1 = fabsf(nx0 - n0) + fabsf(ny0 - n1) + fabsf(nx1 - n2) + fabsf(ny1 - n3);
        }
        const float c_reg = l1 * prm.reg_weight;
        // --- IoUCost (pixels)
        const float area2 = (gt.z - gt.x) * (gt.w - gt.y);
        const float ow = fmaxf(fminf(x1, gt.z) - fmaxf(x0, gt.x), 0.0f);
        const float oh = fmaxf(fminf(y1, gt.w) - fmaxf(y0, gt.y), 0.0f);
        const float overlap = ow * oh;
        const float uni = fmaxf(area1 + area2 - overlap, 1e-6f);
        float iou = overlap / uni;
        if (prm.iou_giou) {
            const float ew = fmaxf(fmaxf(x1, gt.z) - fminf(x0, gt.x), 0.0f);
            const float eh = fmaxf(fmaxf(y1, gt.w) - fminf(y0, gt.y), 0.0f);
            const float earea = fmaxf(ew * eh, 1e-6f);
            iou = iou - (earea - uni) / earea;
        }
        const float c_iou = -iou * prm.iou_weight;
        out[(int64_t)j * Q] = c_cls + c_reg + c_iou;
    }
}

}  // namespace

extern "C" int semidetr_match_cost_f32(void *stream, const float *bbox_pred, const float *cls_pred,
                                       const float *gt_bboxes, const int64_t *gt_labels,
                                       const int32_t *gt_offsets, const float *img_wh, int num_problems,
                                       int num_query, int num_classes, int total_gt,
                                       const semidetr_cost_params *params, float *cost)
{
    SEMIDETR_REQUIRE(num_problems >= 0 && num_query >= 0 && num_classes > 0 && total_gt >= 0,
                     SEMIDETR_E_BADARG, "match_cost: bad sizes (B=%d Q=%d C=%d sumG=%d)", num_problems,
                     num_query, num_classes, total_gt);
    SEMIDETR_REQUIRE(params, SEMIDETR_E_BADARG, "match_cost: null params");
    if (num_problems == 0 || num_query == 0 || total_gt == 0) return SEMIDETR_OK;
    SEMIDETR_REQUIRE(bbox_pred && cls_pred && gt_bboxes && gt_labels && gt_offsets && img_wh && cost,
                     SEMIDETR_E_BADARG, "match_cost: null pointer argument");
    SEMIDETR_REQUIRE(num_problems <= 65535, SEMIDETR_E_TOOLARGE, "match_cost: more than 65535 problems");
    SEMIDETR_REQUIRE((int64_t)num_query * total_gt < INT32_MAX, SEMIDETR_E_TOOLARGE, "match_cost: matrix too large");
    SEMIDETR_REQUIRE((((uintptr_t)bbox_pred | (uintptr_t)gt_bboxes) & 15) == 0, SEMIDETR_E_BADARG,
                     "match_cost: bbox_pred / gt_bboxes must be 16-byte aligned");
    const dim3 grid((num_query + 255) / 256, num_problems);
    hipLaunchKernelGGL(match_cost_kernel, grid, dim3(256), 0, semidetr::as_stream(stream), bbox_pred,
                       cls_pred, gt_bboxes, gt_labels, gt_offsets, img_wh, num_query, num_classes, *params,
                       cost);
    return semidetr::launch_status("match_cost_kernel");
}
