// One-to-many (task-aligned) assignment of the Semi-DETR warm-up stage for gfx950: every (layer, image)
// problem of a loss() call in ONE launch, assignment and training targets together, nothing leaves the device.
//
// Behavioural spec:
//   O2MAssigner.assign        detr_od/core/bbox/assigners/o2m_assigner.py:50-170   (per image; Python loops over the
//                             ground truths at :131-139 and :143-144)
//   bbox_overlaps(mode='iou') thirdparty/mmdetection/mmdet/core/bbox/iou_calculators/iou2d_calculator.py:200-261
//   the in_warm_up branch of  DINODETRSSODHead._get_target_single
//                             detr_od/models/dense_heads/dino_detr_ssod_head.py:1108-1165 (per-gt Python loop :1152-1157)
//
// One 256-thread workgroup per problem.  The decoded prediction boxes live in LDS.  Each wavefront takes ground
// truths round robin: lane l holds the alignment metric  score^alpha * IoU^beta  of queries l, l+64, ... in
// registers, and the top-k candidates are peeled off by k rounds of {local argmax, 64-bit wave max}; a candidate
// with a positive metric bids for its query with an LDS 64-bit atomic max on {IoU bits, ~gt} -- the query goes to
// the ground truth it overlaps most, the first one on ties, exactly `overlaps_inf.max(dim=1)`.  A second pass writes
// the assignment and collects each ground truth's largest metric / IoU over its positives (integer atomic max on
// the bits of non-negative floats), a third one the normalised metrics and the box / label targets.
//
// Conventions where torch leaves the order open: among equal metrics the smaller query index is taken first;
// integral exponents are evaluated by repeated squaring (oracle/o2m_oracle.c does the same; within 2 ulp of pow).
#include <hip/hip_runtime.h>

#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr float kO2mInf = 100000000.0f;
constexpr int kMaxGt = 1024;

__device__ __forceinline__ float ipow_(float x, float e)
{
    if (e == 1.0f) return x;
    if (e == 2.0f) return x * x;
    if (e == 6.0f) { const float x2 = x * x, x4 = x2 * x2; return x4 * x2; }
    return powf(x, e);
}

__device__ __forceinline__ float iou_(const float4 p, const float4 g)
{
    const float area1 = (p.z - p.x) * (p.w - p.y), area2 = (g.z - g.x) * (g.w - g.y);
    const float ow = fmaxf(fminf(p.z, g.z) - fmaxf(p.x, g.x), 0.0f);
    const float oh = fmaxf(fminf(p.w, g.w) - fmaxf(p.y, g.y), 0.0f);
    const float overlap = ow * oh;
    const float uni = fmaxf(area1 + area2 - overlap, 1e-6f);
    return overlap / uni;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, s, 64), hi = __shfl_xor((unsigned)(v >> 32), s, 64);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

struct O2mOut {
    int64_t *gt_inds, *labels;
    float *max_overlaps, *assign_metrics;
    int64_t *labels_full;
    float *bbox_targets, *norm_metrics;
};

// NQL = queries per lane (registers): 16 -> Q <= 1024, 32 -> Q <= 2048
// DYN: the teacher's `multiple_pos` option (o2m_assigner.py:125-133): per ground truth the first k_g of its top-`topk`
// candidates (by metric) are positive, k_g = max(1, int(sum of its `topk` largest IoUs)); no `metric > 0` filter.
template <int NQL, bool DYN = false>
__global__ __launch_bounds__(256) void o2m_assign_kernel(
    const float *__restrict__ bbox_pred, const float *__restrict__ cls_prob, const float *__restrict__ gt_bboxes,
    const int64_t *__restrict__ gt_labels, const int32_t *__restrict__ gt_offsets, const float *__restrict__ img_wh,
    int Q, int C, int topk, float alpha, float beta, int64_t num_classes, O2mOut out)
{
    constexpr int QMAX = NQL * 64;
    __shared__ float4 pb[QMAX];                    // decoded prediction boxes (pixels)
    __shared__ unsigned long long best[QMAX];      // per query: {IoU bits, ~gt} of the winning bid, 0 = none
    __shared__ int gmax_m[kMaxGt], gmax_i[kMaxGt]; // per gt: bits of the largest metric / IoU among its positives
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g0 = gt_offsets[b], G = gt_offsets[b + 1] - g0;
    const float img_w = img_wh[2 * b], img_h = img_wh[2 * b + 1];
    const float *probs = cls_prob + (size_t)b * Q * C;
    const size_t ob = (size_t)b * Q;

    if (G <= 0) {                                  // o2m_assigner.py:94-102: everything background, overlaps 0
        for (int q = tid; q < Q; q += 256) {
            out.gt_inds[ob + q] = 0; out.labels[ob + q] = -1;
            out.max_overlaps[ob + q] = 0.f; out.assign_metrics[ob + q] = 0.f;
            out.labels_full[ob + q] = num_classes; out.norm_metrics[ob + q] = 0.f;
            *reinterpret_cast<float4 *>(out.bbox_targets + (ob + q) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    for (int q = tid; q < QMAX; q += 256) {
        best[q] = 0;
        if (q < Q) {
            const float4 bp = *reinterpret_cast<const float4 *>(bbox_pred + (ob + q) * 4);
            pb[q] = make_float4((bp.x - 0.5f * bp.z) * img_w, (bp.y - 0.5f * bp.w) * img_h, (bp.x + 0.5f * bp.z) * img_w,
                                (bp.y + 0.5f * bp.w) * img_h);
        }
    }
    for (int g = tid; g < G; g += 256) { gmax_m[g] = 0; gmax_i[g] = 0; }
    __syncthreads();

    // ---- top-k candidates of every ground truth bid for their queries
    for (int g = wv; g < G; g += 4) {
        const float4 gt = *reinterpret_cast<const float4 *>(gt_bboxes + (size_t)(g0 + g) * 4);
        const int label = (int)gt_labels[g0 + g];
        float met[NQL];
#pragma unroll
        for (int t = 0; t < NQL; ++t) {
            const int q = lane + 64 * t;
            met[t] = -1.f;                         // not a query / already taken
            // a label outside [0, C) (an index error in the reference, o2m_assigner.py:98) never reads out of bounds:
            // such a ground truth simply gets no candidates
            if (q < Q && (unsigned)label < (unsigned)C)
                met[t] = ipow_(probs[(size_t)q * C + label], alpha) * ipow_(iou_(pb[q], gt), beta);
        }
        int rounds = topk;
        if (DYN) {
            // k_g: the sum of the ground truth's `topk` largest IoUs (taken in descending order, as torch.topk returns
            // them), truncated, at least 1
            float iv[NQL];
#pragma unroll
            for (int t = 0; t < NQL; ++t) {
                const int q = lane + 64 * t;
                iv[t] = q < Q ? iou_(pb[q], gt) : -1.f;
            }
            float sum = 0.f;
            for (int r = 0; r < topk; ++r) {
                float bv = -1.f;
                int bt = 0;
#pragma unroll
                for (int t = 0; t < NQL; ++t)
                    if (iv[t] > bv) { bv = iv[t]; bt = t; }
                const unsigned long long mine = bv >= 0.f ? ((unsigned long long)(__float_as_uint(bv) + 1u) << 32) |
                                                                (unsigned)(0xFFFFFFFFu - (unsigned)(lane + 64 * bt)) : 0ull;
                const unsigned long long win = wave_max_u64(mine);
                if (win == 0) break;
                sum += __uint_as_float((unsigned)(win >> 32) - 1u);
                if (win == mine) {
#pragma unroll
                    for (int t = 0; t < NQL; ++t)
                        if (t == bt) iv[t] = -1.f;
                }
            }
            rounds = min(topk, max((int)sum, 1));
        }
        for (int r = 0; r < rounds; ++r) {
            float bv = -1.f;
            int bt = 0;
#pragma unroll
            for (int t = 0; t < NQL; ++t)
                if (met[t] > bv) { bv = met[t]; bt = t; }          // first maximum = smallest query of the lane
            const int bq = lane + 64 * bt;
            // DYN: a candidate with metric 0 is still a candidate -- the key's high word is (bits + 1), 0 = nothing left
            const unsigned long long mine = bv >= 0.f ? ((unsigned long long)(__float_as_uint(bv) + (DYN ? 1u : 0u)) << 32) |
                                                            (unsigned)(0xFFFFFFFFu - (unsigned)bq) : 0ull;
            const unsigned long long win = wave_max_u64(mine);
            if ((win >> 32) == 0) break;           // !DYN: best remaining metric is 0 (or nothing left): is_pos is false from here on
            if (win == mine) {
#pragma unroll
                for (int t = 0; t < NQL; ++t)
                    if (t == bt) met[t] = -1.f;
                const float iou = iou_(pb[bq], gt);
                atomicMax(&best[bq], ((unsigned long long)__float_as_uint(iou) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)g));
            }
        }
    }
    __syncthreads();

    // ---- assignment of every query; per-gt maxima over the positives
    for (int q = tid; q < Q; q += 256) {
        const unsigned long long k = best[q];
        int64_t gi = 0, lab = -1;
        float mo = -kO2mInf, am = 0.f;
        if (k) {
            const int g = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
            const int label = (int)gt_labels[g0 + g];
            const float4 gt = *reinterpret_cast<const float4 *>(gt_bboxes + (size_t)(g0 + g) * 4);
            mo = __uint_as_float((unsigned)(k >> 32));
            am = ipow_(probs[(size_t)q * C + label], alpha) * ipow_(iou_(pb[q], gt), beta);
            gi = g + 1;
            lab = label;
            atomicMax(&gmax_m[g], __float_as_int(am));
            atomicMax(&gmax_i[g], __float_as_int(mo));
        }
        out.gt_inds[ob + q] = gi; out.labels[ob + q] = lab;
        out.max_overlaps[ob + q] = mo; out.assign_metrics[ob + q] = am;
    }
    __syncthreads();

    // ---- training targets (head.py:1128-1160)
    for (int q = tid; q < Q; q += 256) {
        const unsigned long long k = best[q];
        int64_t lf = num_classes;
        float4 bt = make_float4(0.f, 0.f, 0.f, 0.f);
        float nm = 0.f;
        if (k) {
            const int g = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
            const float4 gt = *reinterpret_cast<const float4 *>(gt_bboxes + (size_t)(g0 + g) * 4);
            const float n0 = gt.x / img_w, n1 = gt.y / img_h, n2 = gt.z / img_w, n3 = gt.w / img_h;
            bt = make_float4((n0 + n2) / 2, (n1 + n3) / 2, n2 - n0, n3 - n1);
            lf = gt_labels[g0 + g];
            nm = out.assign_metrics[ob + q] / (__int_as_float(gmax_m[g]) + 10e-8f) * __int_as_float(gmax_i[g]);
        }
        out.labels_full[ob + q] = lf;
        out.norm_metrics[ob + q] = nm;
        *reinterpret_cast<float4 *>(out.bbox_targets + (ob + q) * 4) = bt;
    }
}

}  // namespace

extern "C" int semidetr_o2m_assign_f32(void *stream, const float *bbox_pred, const float *cls_prob,
                                       const float *gt_bboxes, const int64_t *gt_labels, const int32_t *gt_offsets,
                                       const float *img_wh, int num_problems, int num_query, int num_classes,
                                       int total_gt, int max_gt_per_problem, int candidate_topk, int dynamic_k,
                                       float alpha, float beta, int64_t *gt_inds, int64_t *labels, float *max_overlaps,
                                       float *assign_metrics, int64_t *labels_full, float *bbox_targets,
                                       float *norm_metrics)
{
    const int B = num_problems, Q = num_query, C = num_classes;
    SEMIDETR_REQUIRE(B >= 0 && Q >= 0 && C > 0 && total_gt >= 0 && max_gt_per_problem >= 0, SEMIDETR_E_BADARG,
                     "o2m_assign: bad sizes (B=%d Q=%d C=%d sumG=%d)", B, Q, C, total_gt);
    if (B == 0 || Q == 0) return SEMIDETR_OK;
    SEMIDETR_REQUIRE(bbox_pred && cls_prob && gt_offsets && img_wh && gt_inds && labels && max_overlaps && assign_metrics &&
                         labels_full && bbox_targets && norm_metrics && (total_gt == 0 || (gt_bboxes && gt_labels)),
                     SEMIDETR_E_BADARG, "o2m_assign: null pointer argument");
    SEMIDETR_REQUIRE(candidate_topk >= 1, SEMIDETR_E_BADARG, "o2m_assign: candidate_topk must be >= 1");
    // torch.topk raises for k > num_query (o2m_assigner.py:121)
    SEMIDETR_REQUIRE(total_gt == 0 || candidate_topk <= Q, SEMIDETR_E_BADARG, "selected index k out of range");
    SEMIDETR_REQUIRE(Q <= 2048, SEMIDETR_E_TOOLARGE, "o2m_assign: at most 2048 queries per problem (got %d)", Q);
    SEMIDETR_REQUIRE(max_gt_per_problem <= kMaxGt, SEMIDETR_E_TOOLARGE,
                     "o2m_assign: at most %d ground truths per problem (got %d)", kMaxGt, max_gt_per_problem);
    SEMIDETR_REQUIRE((((uintptr_t)bbox_pred | (uintptr_t)gt_bboxes | (uintptr_t)bbox_targets) & 15) == 0, SEMIDETR_E_BADARG,
                     "o2m_assign: bbox_pred / gt_bboxes / bbox_targets must be 16-byte aligned");
    const O2mOut out = {gt_inds, labels, max_overlaps, assign_metrics, labels_full, bbox_targets, norm_metrics};
    hipStream_t st = semidetr::as_stream(stream);
#define O2M_LAUNCH(NQL_, DYN_)                                                                                       \
    hipLaunchKernelGGL((o2m_assign_kernel<NQL_, DYN_>), dim3(B), dim3(256), 0, st, bbox_pred, cls_prob, gt_bboxes, gt_labels, \
                       gt_offsets, img_wh, Q, C, candidate_topk, alpha, beta, (int64_t)C, out)
    if (Q <= 1024) { if (dynamic_k) O2M_LAUNCH(16, true); else O2M_LAUNCH(16, false); }
    else { if (dynamic_k) O2M_LAUNCH(32, true); else O2M_LAUNCH(32, false); }
#undef O2M_LAUNCH
    return semidetr::launch_status("o2m_assign_kernel");
}
