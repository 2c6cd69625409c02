/* CPU ORACLE (test infrastructure only -- never imported by the product) of the one-to-many assigner used in
 * the warm-up stage of Semi-DETR and of the target construction around it.
 *
 * Follows
 *   O2MAssigner.assign          /root/reference/detr_od/core/bbox/assigners/o2m_assigner.py:50-170
 *   bbox_overlaps(mode='iou')   /root/reference/thirdparty/mmdetection/mmdet/core/bbox/iou_calculators/iou2d_calculator.py:200-261
 *   bbox_cxcywh_to_xyxy         /root/reference/thirdparty/mmdetection/mmdet/core/bbox/transforms.py:222-233
 *   the in_warm_up branch of DINODETRSSODHead._get_target_single
 *                               /root/reference/detr_od/models/dense_heads/dino_detr_ssod_head.py:1108-1165
 * Pinned by tests/golden/o2m.npz (the reference's own O2MAssigner imported by path, oracle/gen_golden.py).
 *
 * Conventions where torch leaves the result open: torch.topk's order among equal metrics -> smaller query
 * index first (only matters when a tie straddles the k-th place; zero metrics are filtered by `> 0` anyway);
 * x ** beta for integral beta is evaluated by repeated squaring (x2 = x*x, x4 = x2*x2, x6 = x4*x2; within
 * 2 ulp of torch.pow), other exponents with powf.  fp32 throughout, contraction off (Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define O2M_INF 100000000.0f

static float ipow_(float x, float e)
{
    if (e == 1.0f) return x;
    if (e == 2.0f) return x * x;
    if (e == 6.0f) { const float x2 = x * x, x4 = x2 * x2; return x4 * x2; }
    return powf(x, e);
}

static float iou_(const float *p, const float *g)      /* iou2d_calculator.py:232-260, mode 'iou', eps 1e-6 */
{
    const float area1 = (p[2] - p[0]) * (p[3] - p[1]), area2 = (g[2] - g[0]) * (g[3] - g[1]);
    const float ow = fmaxf(fminf(p[2], g[2]) - fmaxf(p[0], g[0]), 0.0f);
    const float oh = fmaxf(fminf(p[3], g[3]) - fmaxf(p[1], g[1]), 0.0f);
    const float overlap = ow * oh;
    const float uni = fmaxf(area1 + area2 - overlap, 1e-6f);
    return overlap / uni;
}

/* One image.  cls_prob (Q,C) probabilities (the call site passes cls_score.sigmoid(), head.py:1111).
 * gt_inds, labels (Q,) int64; max_overlaps, assign_metrics (Q,) fp32. */
void o2m_assign_oracle(const float *bbox_pred, const float *cls_prob, const float *gt_bboxes, const int64_t *gt_labels,
                       int Q, int C, int G, float img_w, float img_h, int topk, int dynamic_k, float alpha, float beta,
                       int64_t *gt_inds, int64_t *labels, float *max_overlaps, float *assign_metrics)
{
    for (int q = 0; q < Q; ++q) { gt_inds[q] = G == 0 ? 0 : -1; labels[q] = -1; max_overlaps[q] = 0.f; assign_metrics[q] = 0.f; }
    if (G == 0 || Q == 0) return;                                         /* o2m_assigner.py:94-102 */
    float *iou = (float *)malloc(sizeof(float) * (size_t)Q * G), *met = (float *)malloc(sizeof(float) * (size_t)Q * G);
    float *inf = (float *)malloc(sizeof(float) * (size_t)Q * G);
    unsigned char *taken = (unsigned char *)malloc((size_t)(Q > 0 ? Q : 1));
    for (int q = 0; q < Q; ++q) {
        const float *b = bbox_pred + 4 * q;
        const float p[4] = {(b[0] - 0.5f * b[2]) * img_w, (b[1] - 0.5f * b[3]) * img_h, (b[0] + 0.5f * b[2]) * img_w,
                            (b[1] + 0.5f * b[3]) * img_h};
        for (int g = 0; g < G; ++g) {
            iou[(size_t)q * G + g] = iou_(p, gt_bboxes + 4 * g);
            met[(size_t)q * G + g] = ipow_(cls_prob[(size_t)q * C + gt_labels[g]], alpha) * ipow_(iou[(size_t)q * G + g], beta);
            inf[(size_t)q * G + g] = -O2M_INF;
        }
    }
    const int k = topk < Q ? topk : Q;
    for (int g = 0; g < G; ++g) {                                         /* topk per gt */
        int kpos = k;
        if (dynamic_k) {                                                  /* o2m_assigner.py:125-133: k_g = clamp(int(sum of the */
            float sum = 0.f;                                              /* top-k IoUs), min=1), added in descending order      */
            for (int q = 0; q < Q; ++q) taken[q] = 0;
            for (int r = 0; r < k; ++r) {
                int best = -1;
                for (int q = 0; q < Q; ++q)
                    if (!taken[q] && (best < 0 || iou[(size_t)q * G + g] > iou[(size_t)best * G + g])) best = q;
                taken[best] = 1;
                sum += iou[(size_t)best * G + g];
            }
            kpos = (int)sum < 1 ? 1 : (int)sum;
            if (kpos > k) kpos = k;
        }
        for (int q = 0; q < Q; ++q) taken[q] = 0;
        for (int r = 0; r < kpos; ++r) {
            int best = -1;
            for (int q = 0; q < Q; ++q)
                if (!taken[q] && (best < 0 || met[(size_t)q * G + g] > met[(size_t)best * G + g])) best = q;
            taken[best] = 1;
            /* is_pos: metric > 0 (:135), or -- dynamic k -- simply being among the first k_g candidates (:128-133) */
            if (dynamic_k || met[(size_t)best * G + g] > 0.f) inf[(size_t)best * G + g] = iou[(size_t)best * G + g];
        }
    }
    for (int q = 0; q < Q; ++q) {                                         /* max over gts, first maximum wins */
        int arg = 0;
        for (int g = 1; g < G; ++g)
            if (inf[(size_t)q * G + g] > inf[(size_t)q * G + arg]) arg = g;
        max_overlaps[q] = inf[(size_t)q * G + arg];
        gt_inds[q] = 0;
        if (max_overlaps[q] != -O2M_INF) {
            gt_inds[q] = arg + 1;
            assign_metrics[q] = met[(size_t)q * G + arg];
            labels[q] = gt_labels[arg];
        }
    }
    free(iou); free(met); free(inf); free(taken);
}

/* The warm-up branch of _get_target_single after the assignment (head.py:1114-1165):
 * labels_full (Q,) = class of the assigned gt or num_classes; bbox_targets (Q,4) = gt as normalised cxcywh for
 * positives, 0 otherwise; norm_metrics (Q,) = metric / (max metric of the gt's positives + 10e-8) * max IoU of them. */
void o2m_targets_oracle(const int64_t *gt_inds, const float *max_overlaps, const float *assign_metrics,
                        const float *gt_bboxes, const int64_t *gt_labels, int Q, int G, float img_w, float img_h,
                        int64_t num_classes, int64_t *labels_full, float *bbox_targets, float *norm_metrics)
{
    float *mm = (float *)calloc((size_t)(G > 0 ? G : 1), sizeof(float)), *mi = (float *)calloc((size_t)(G > 0 ? G : 1), sizeof(float));
    unsigned char *has = (unsigned char *)calloc((size_t)(G > 0 ? G : 1), 1);
    for (int q = 0; q < Q; ++q)
        if (gt_inds[q] > 0) {
            const int g = (int)gt_inds[q] - 1;
            const float iou = max_overlaps[q] == -O2M_INF ? 0.f : max_overlaps[q];
            if (!has[g] || assign_metrics[q] > mm[g]) mm[g] = assign_metrics[q];
            if (!has[g] || iou > mi[g]) mi[g] = iou;
            has[g] = 1;
        }
    for (int q = 0; q < Q; ++q) {
        labels_full[q] = num_classes;
        norm_metrics[q] = 0.f;
        for (int k = 0; k < 4; ++k) bbox_targets[4 * q + k] = 0.f;
        if (gt_inds[q] > 0) {
            const int g = (int)gt_inds[q] - 1;
            const float *b = gt_bboxes + 4 * g;
            const float n0 = b[0] / img_w, n1 = b[1] / img_h, n2 = b[2] / img_w, n3 = b[3] / img_h;
            bbox_targets[4 * q + 0] = (n0 + n2) / 2; bbox_targets[4 * q + 1] = (n1 + n3) / 2;
            bbox_targets[4 * q + 2] = n2 - n0; bbox_targets[4 * q + 3] = n3 - n1;
            labels_full[q] = gt_labels[g];
            norm_metrics[q] = assign_metrics[q] / (mm[g] + 10e-8f) * mi[g];
        }
    }
    free(mm); free(mi); free(has);
}

/* task_aigned_focal_loss (task_aligned_focal_loss.py:35-66) on probabilities or logits: returns the loss SUM and
 * writes d sum / d input when grad != NULL (BCE backward as torch: (p - s) / max((1 - p) p, 1e-12)). */
double tal_loss_oracle(const float *scores, const int64_t *labels, const float *metrics, int64_t N, int C, float gamma,
                       int input_is_prob, float *grad)
{
    double sum = 0.0;
    for (int64_t i = 0; i < N; ++i)
        for (int c = 0; c < C; ++c) {
            const float x = scores[i * C + c];
            const float s = labels[i] == c ? metrics[i] : 0.f;
            const float p = input_is_prob ? x : 1.0f / (1.0f + expf(-x));
            const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.0f - p), -100.f);
            const float ce = -(s * lp + (1.0f - s) * l1p);
            const float d = s - p, ad = fabsf(d);
            const float mod = gamma == 2.0f ? ad * ad : powf(ad, gamma);
            sum += (double)(mod * ce);
            if (grad) {
                const float dmod = gamma == 2.0f ? -2.0f * d : (ad > 0.f ? -gamma * powf(ad, gamma - 1.0f) * (d > 0.f ? 1.f : -1.f) : 0.f);
                const float dce = (p - s) / fmaxf((1.0f - p) * p, 1e-12f);
                const float dp = dmod * ce + mod * dce;
                grad[i * C + c] = input_is_prob ? dp : dp * (p * (1.0f - p));
            }
        }
    return sum;
}
