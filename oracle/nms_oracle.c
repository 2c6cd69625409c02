/* CPU ORACLE (test infrastructure only -- never imported by the product) of the teacher's test-time box
 * decoding for pseudo labels and of the weak->strong box warp.
 *
 * Follows
 *   DINODETRSSODHead._get_bboxes_single(for_pseudo_label=True)
 *       /root/reference/detr_od/models/dense_heads/dino_detr_ssod_head.py:1364-1395
 *   multiclass_nms  /root/reference/thirdparty/mmdetection/mmdet/core/post_processing/bbox_nms.py:8-95
 *   bbox_cxcywh_to_xyxy  /root/reference/thirdparty/mmdetection/mmdet/core/bbox/transforms.py:222-233
 *   Transform2D.transform_bboxes / bbox2points / points2bbox
 *       /root/reference/detr_ssod/models/utils/bbox_utils.py:167-192, :18-41
 * and, for the part that lives in an UN-VENDORED third-party dependency (mmcv-full, pinned to 1.3.16 by the
 * reference's README.md:11,30; not installed here => PARITY UNPINNED for the NMS keep decisions), restates the
 * published algorithm of mmcv.ops.batched_nms / mmcv.ops.nms:
 *   batched_nms: boxes_for_nms = boxes + label * (boxes.max() + 1); with split_thr = -1 (the reference's cfg)
 *                one greedy NMS per class, kept entries of all classes sorted by score, descending;
 *   nms (offset 0): visit boxes by descending score, drop box j if an earlier KEPT box i has
 *                inter / (area_i + area_j - inter) > iou_threshold,
 *                inter = max(min(x2) - max(x1), 0) * max(min(y2) - max(y1), 0), area = (x2-x1) * (y2-y1).
 * Tie rule (the reference's sorts are unstable, i.e. its order of equal scores is unspecified): candidates are
 * ordered by LOGIT descending (the same order as any monotonic sigmoid), equal logits by ascending flat index
 * q * C + c.  All arithmetic fp32, one rounding per operation (the Makefile switches contraction off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

typedef struct {
    float logit;
    int32_t flat;   /* q * C + c */
} cand_t;

static int cand_cmp(const void *a, const void *b)
{
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->logit > y->logit) return -1;
    if (x->logit < y->logit) return 1;
    return x->flat < y->flat ? -1 : (x->flat > y->flat ? 1 : 0);
}

/* box of query q in pixels: cxcywh -> xyxy, scale by the image size, clamp (head.py:1381-1385) */
static void decode_box(const float *bp, float img_h, float img_w, float *o)
{
    const float cx = bp[0], cy = bp[1], w = bp[2], h = bp[3];
    float x1 = (cx - 0.5f * w) * img_w, y1 = (cy - 0.5f * h) * img_h;
    float x2 = (cx + 0.5f * w) * img_w, y2 = (cy + 0.5f * h) * img_h;
    o[0] = fminf(fmaxf(x1, 0.f), img_w);
    o[1] = fminf(fmaxf(y1, 0.f), img_h);
    o[2] = fminf(fmaxf(x2, 0.f), img_w);
    o[3] = fminf(fmaxf(y2, 0.f), img_h);
}

static int iou_gt(const float *a, const float *b, float thr)
{
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float inter = width * height;
    const float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
    return inter / (sa + sb - inter) > thr;
}

/* One image.  cls_logits (Q,C), bbox_pred (Q,4) normalised cxcywh.  Writes up to max_num rows of
 * dets (x1,y1,x2,y2,score) / labels; returns the number written. */
int pseudo_nms_oracle(const float *cls_logits, const float *bbox_pred, int Q, int C, float img_h, float img_w,
                      float score_thr, float iou_thr, int max_num, float *dets, int64_t *labels)
{
    float *boxes = (float *)malloc(sizeof(float) * 4 * (size_t)(Q > 0 ? Q : 1));
    cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)(Q > 0 ? Q : 1) * (size_t)(C > 0 ? C : 1));
    cand_t *kept = (cand_t *)malloc(sizeof(cand_t) * (size_t)(Q > 0 ? Q : 1) * (size_t)(C > 0 ? C : 1));
    cand_t *cls = (cand_t *)malloc(sizeof(cand_t) * (size_t)(Q > 0 ? Q : 1));
    float *ob = (float *)malloc(sizeof(float) * 4 * (size_t)(Q > 0 ? Q : 1));
    unsigned char *sup = (unsigned char *)malloc((size_t)(Q > 0 ? Q : 1));
    int ncand = 0, nkept = 0;
    float maxc = -INFINITY;
    for (int q = 0; q < Q; ++q) decode_box(bbox_pred + 4 * q, img_h, img_w, boxes + 4 * q);
    for (int q = 0; q < Q; ++q) {
        int any = 0;
        for (int c = 0; c < C; ++c)
            if (sigmoidf_(cls_logits[(size_t)q * C + c]) > score_thr) {      /* bbox_nms.py:54 */
                cand[ncand].logit = cls_logits[(size_t)q * C + c];
                cand[ncand].flat = q * C + c;
                ++ncand;
                any = 1;
            }
        if (any)
            for (int k = 0; k < 4; ++k) maxc = fmaxf(maxc, boxes[4 * q + k]);   /* boxes.max() over candidates */
    }
    for (int c = 0; c < C && ncand; ++c) {
        int n = 0;
        for (int i = 0; i < ncand; ++i)
            if (cand[i].flat % C == c) cls[n++] = cand[i];
        if (!n) continue;
        qsort(cls, (size_t)n, sizeof(cand_t), cand_cmp);
        const float off = (float)c * (maxc + 1.0f);                           /* batched_nms offsets */
        for (int i = 0; i < n; ++i) {
            const float *bq = boxes + 4 * (cls[i].flat / C);
            for (int k = 0; k < 4; ++k) ob[4 * i + k] = bq[k] + off;
            sup[i] = 0;
        }
        for (int i = 0; i < n; ++i) {
            if (sup[i]) continue;
            kept[nkept++] = cls[i];
            for (int j = i + 1; j < n; ++j)
                if (!sup[j] && iou_gt(ob + 4 * i, ob + 4 * j, iou_thr)) sup[j] = 1;
        }
    }
    qsort(kept, (size_t)nkept, sizeof(cand_t), cand_cmp);                      /* scores.sort(descending) */
    int nout = nkept;
    if (max_num > 0 && nout > max_num) nout = max_num;                        /* bbox_nms.py:88-90 */
    for (int i = 0; i < nout; ++i) {
        const int q = kept[i].flat / C, c = kept[i].flat % C;
        memcpy(dets + 5 * i, boxes + 4 * q, sizeof(float) * 4);
        dets[5 * i + 4] = sigmoidf_(kept[i].logit);
        labels[i] = c;
    }
    free(boxes); free(cand); free(kept); free(cls); free(ob); free(sup);
    return nout;
}

/* Transform2D.transform_bboxes for one image: boxes (K,4) xyxy, M (3,3) row major, out_shape (h, w). */
void transform_bboxes_oracle(const float *boxes, int K, const float *M, float out_h, float out_w, float *out)
{
    for (int i = 0; i < K; ++i) {
        const float *b = boxes + 4 * i;
        const float px[4] = {b[0], b[2], b[2], b[0]}, py[4] = {b[1], b[1], b[3], b[3]};   /* bbox2points */
        float minx = INFINITY, miny = INFINITY, maxx = -INFINITY, maxy = -INFINITY;
        for (int k = 0; k < 4; ++k) {
            const float x = M[0] * px[k] + M[1] * py[k] + M[2];
            const float y = M[3] * px[k] + M[4] * py[k] + M[5];
            const float z = M[6] * px[k] + M[7] * py[k] + M[8];
            const float u = x / z, v = y / z;
            minx = fminf(minx, u); maxx = fmaxf(maxx, u);
            miny = fminf(miny, v); maxy = fmaxf(maxy, v);
        }
        out[4 * i + 0] = fminf(fmaxf(minx, 0.f), out_w);
        out[4 * i + 1] = fminf(fmaxf(miny, 0.f), out_h);
        out[4 * i + 2] = fminf(fmaxf(maxx, 0.f), out_w);
        out[4 * i + 3] = fminf(fmaxf(maxy, 0.f), out_h);
    }
}
