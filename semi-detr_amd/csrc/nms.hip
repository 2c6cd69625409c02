// Teacher test-time box decoding for pseudo labels on gfx950, one call for the whole batch and no host round
// trip: sigmoid -> cxcywh to clamped pixel xyxy -> score threshold -> class-aware greedy NMS -> the
// max_per_img best by score, plus the weak->strong affine warp of the surviving boxes.
//
// Behavioural spec:
//   DINODETRSSODHead._get_bboxes_single(for_pseudo_label=True)
//       detr_od/models/dense_heads/dino_detr_ssod_head.py:1364-1395   (per image, Python)
//   multiclass_nms      thirdparty/mmdetection/mmdet/core/post_processing/bbox_nms.py:8-95
//   batched_nms / nms   mmcv-full 1.3.16 (un-vendored; algorithm restated in oracle/nms_oracle.c):
//       boxes_for_nms = boxes + label * (boxes.max() + 1); one greedy NMS per class (split_thr = -1), kept
//       entries of all classes sorted by score descending; IoU as in mmcv's devIoU with offset 0.
//   Transform2D.transform_bboxes   detr_ssod/models/utils/bbox_utils.py:167-192 (+ bbox2points/points2bbox :18-41)
//
// Order: candidates are ranked by LOGIT (any monotonic sigmoid gives the same ranking), equal logits by
// ascending flat index q * C + c -- a 64-bit key {orderable(logit), ~flat}; all keys are distinct, so the
// result is deterministic although the final list is assembled with atomics.
//
// Three launches (after one memset of the per-image scalars):
//   nms_prepare_kernel (grid Q/4 x B): one wavefront per query: its box, boxes.max() over the thresholded candidates
//   nms_class_kernel   (grid C x B):   one or four wavefronts per (class, image): bitonic sort of the class's candidates in
//                                      LDS, greedy suppression in sorted order, survivors appended to the image's list
//   nms_topk_kernel    (grid B):       radix select of the max_per_img largest keys (when more than 2048 survive;
//                                      bank-spread per-lane histograms, stops as soon as the rest fits), sort, emit
#include <hip/hip_runtime.h>

#include "common.h"

#pragma clang fp contract(off)      // every result is one rounded operation, like the reference's separate torch ops

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ unsigned orderable(float x)      // monotonic float -> unsigned; -0 and +0 compare equal
{
    const unsigned u = x == 0.f ? 0u : __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float4 decode_box(const float *bp, float img_h, float img_w)
{
    const float cx = bp[0], cy = bp[1], w = bp[2], h = bp[3];
    const float x1 = (cx - 0.5f * w) * img_w, y1 = (cy - 0.5f * h) * img_h;
    const float x2 = (cx + 0.5f * w) * img_w, y2 = (cy + 0.5f * h) * img_h;
    return make_float4(fminf(fmaxf(x1, 0.f), img_w), fminf(fmaxf(y1, 0.f), img_h), fminf(fmaxf(x2, 0.f), img_w),
                       fminf(fmaxf(y2, 0.f), img_h));
}

// thr >= 0 (checked by the launcher): disjoint boxes have inter == 0, i.e. a ratio of 0 or NaN -- never above thr,
// so the division is only evaluated for intersecting pairs (same decisions as the plain formula).
__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr)
{
    const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    const float width = right - left, height = bottom - top;
    if (!(width > 0.f && height > 0.f)) return false;
    const float inter = width * height;
    const float sa = (a.z - a.x) * (a.w - a.y), sb = (b.z - b.x) * (b.w - b.y);
    return inter / (sa + sb - inter) > thr;
}

struct Workspace {
    float4 *boxes;                 // (B, Q)
    float *maxc;                   // (B,)
    int *count;                    // (B,)
    unsigned long long *keys;      // (B, Q * C)
};

__host__ __device__ inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline size_t workspace_bytes(int B, int Q, int C)
{
    return align256(sizeof(float4) * (size_t)B * Q) + align256(sizeof(float) * (size_t)B) +
           align256(sizeof(int) * (size_t)B) + align256(sizeof(unsigned long long) * (size_t)B * Q * C);
}

inline Workspace carve(void *ws, int B, int Q, int C)
{
    char *p = static_cast<char *>(ws);
    Workspace w;
    w.boxes = reinterpret_cast<float4 *>(p); p += align256(sizeof(float4) * (size_t)B * Q);
    w.maxc = reinterpret_cast<float *>(p); p += align256(sizeof(float) * (size_t)B);
    w.count = reinterpret_cast<int *>(p); p += align256(sizeof(int) * (size_t)B);
    w.keys = reinterpret_cast<unsigned long long *>(p);
    (void)C;
    return w;
}

// ---- boxes of the image + boxes.max() over the queries that have at least one class above the threshold.
// One wavefront per query (lanes stride the classes), 16 queries per workgroup; boxes are clamped to [0, size],
// so the running maximum is an integer atomic max on the float's bits, one per workgroup (the slot is zeroed by
// the launcher together with count).
constexpr int kPrepQueries = 16;
__global__ __launch_bounds__(256) void nms_prepare_kernel(const float *__restrict__ logits,
                                                          const float *__restrict__ bbox_pred,
                                                          const float *__restrict__ img_hw, int Q, int C,
                                                          float score_thr, Workspace ws)
{
    __shared__ float red[4];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float img_h = img_hw[2 * b], img_w = img_hw[2 * b + 1];
    float mx = 0.f;
    for (int r = wv; r < kPrepQueries; r += 4) {
        const int q = blockIdx.x * kPrepQueries + r;
        if (q >= Q) break;
        const float4 box = decode_box(bbox_pred + ((size_t)b * Q + q) * 4, img_h, img_w);
        const float *lr = logits + ((size_t)b * Q + q) * C;
        bool any = false;
        for (int c = lane; c < C; c += 64) any |= sigmoidf_(lr[c]) > score_thr;
        if (lane == 0) ws.boxes[(size_t)b * Q + q] = box;
        if (__ballot(any) != 0ull) mx = fmaxf(mx, fmaxf(fmaxf(box.x, box.y), fmaxf(box.z, box.w)));
    }
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (mx > 0.f) atomicMax(reinterpret_cast<int *>(ws.maxc) + b, __float_as_int(mx));   // >= 0: int order == float order
    }
}

// descending bitonic sort of n2 (power of two) 64-bit keys in LDS by NT threads
template <int NT>
__device__ __forceinline__ void bitonic_desc(unsigned long long *keys, int n2)
{
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < n2 / 2; t += NT) {
                const int i = 2 * t - (t & (j - 1));          // index with bit j clear
                const int p = i + j;
                const unsigned long long a = keys[i], c = keys[p];
                const bool desc = (i & k) == 0;
                if (desc ? a < c : a > c) { keys[i] = c; keys[p] = a; }
            }
        }
    }
    __syncthreads();
}

// ---- one (class, image): sort the class's candidates, greedy NMS, append the survivors to the image's list.
// NW wavefronts per problem; the C x B problems spread over the CUs.  The greedy scan runs in chunks of 64 sorted
// candidates (lane = candidate, every wavefront holds the same chunk): a lane first checks its box against the boxes kept
// so far (uniform LDS reads; wavefront w takes the w-th share of the kept list), then builds the bit row of earlier chunk
// members it overlaps (wavefront w: members 64 w / NW ...); hits and rows are OR-ed through LDS, and the chunk is resolved
// in order with scalar bit operations (redundantly in every wavefront: no second exchange).  Exactly the sequential
// algorithm's decisions, but ~n/64 dependent steps instead of n.  With random-init logits every query is a candidate of
// every class (n = Q = 900, nearly all kept): 405 k IoU tests per problem -- measured per launch of 80 x 4 problems:
// one wavefront 157 us, two 113, four 73, eight 99, sixteen 182 (the sort's and the chunks' barriers grow with the workgroup).
template <int NP2, int NW>
__global__ __launch_bounds__(64 * NW) void nms_class_kernel(const float *__restrict__ logits, int Q, int C,
                                                            float score_thr, float iou_thr, Workspace ws)
{
    constexpr int NT = 64 * NW;
    __shared__ unsigned long long keys[NP2];
    __shared__ float4 ob[NP2];       // class-offset boxes in score order
    __shared__ float4 kb[NP2];       // ... of the candidates kept so far
    __shared__ unsigned long long s_hit[NW], s_row[NW][64];
    __shared__ int s_base, s_n;
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // candidates above the threshold, compacted to the front (ballot prefix inside a wavefront, one LDS atomic per
    // wavefront and round; any order -- the keys are distinct and sorted next), so that the sort only covers the next
    // power of two >= n instead of all NP2 slots
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int q0 = 0; q0 < Q; q0 += NT) {
        const int q = q0 + tid;
        unsigned long long key = 0;
        if (q < Q) {
            const float x = logits[((size_t)b * Q + q) * C + c];
            if (sigmoidf_(x) > score_thr)
                key = ((unsigned long long)orderable(x) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)q);
        }
        const unsigned long long mask = __ballot(key != 0);
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(&s_n, __popcll(mask));
        base = __shfl(base, 0, 64);
        if (key) keys[base + __popcll(mask & lt)] = key;
    }
    __syncthreads();
    const int n = s_n;
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    for (int i = n + tid; i < n2; i += NT) keys[i] = 0;
    bitonic_desc<NT>(keys, n2);
    if (n == 0) return;
    const float off = (float)c * (ws.maxc[b] + 1.0f);
    for (int i = tid; i < n; i += NT) {
        const int q = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        const float4 bx = ws.boxes[(size_t)b * Q + q];
        ob[i] = make_float4(bx.x + off, bx.y + off, bx.z + off, bx.w + off);
    }
    __syncthreads();
    int m = 0;                                   // kept so far (uniform)
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int j = i0 + lane;
        const bool valid = j < n;
        const float4 bj = ob[valid ? j : 0];
        // (both loops are chains of uniform LDS reads: unrolled so that eight reads are in flight, no short circuit)
        bool hit = false;
        const int share = ((m + NW - 1) / NW + 7) & ~7;
        int k = wv * share;
        const int kend = min(m, k + share);
        for (; k + 8 <= kend; k += 8) {
            float4 kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kk[u] = kb[k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) hit |= iou_gt(kk[u], bj, iou_thr);
        }
        for (; k < kend; ++k) hit |= iou_gt(kb[k], bj, iou_thr);
        unsigned long long row = 0;              // earlier members of this chunk that would suppress me
        const int cn = n - i0 < 64 ? n - i0 : 64;
        int i = wv * (64 / NW);
        const int iend = min(cn, i + 64 / NW);
        for (; i + 8 <= iend; i += 8) {
            float4 kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kk[u] = ob[i0 + i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i + u < lane && iou_gt(kk[u], bj, iou_thr)) row |= 1ull << (i + u);
        }
        for (; i < iend; ++i)
            if (i < lane && iou_gt(ob[i0 + i], bj, iou_thr)) row |= 1ull << i;
        unsigned long long hits = __ballot(hit);
        if (NW > 1) {
            if (lane == 0) s_hit[wv] = hits;
            s_row[wv][lane] = row;
            __syncthreads();
            hits = 0;
            row = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                hits |= s_hit[w];
                row |= s_row[w][lane];
            }
        }
        const bool alive = valid && !((hits >> lane) & 1ull);
        // resolve the chunk in score order: member i is kept iff alive and no KEPT earlier member suppresses it
        unsigned long long cand = __ballot(alive), kept = 0;
        const unsigned row_lo = (unsigned)row, row_hi = (unsigned)(row >> 32);
        while (cand) {
            const int ci = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            const unsigned long long ri = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)row_hi, ci) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)row_lo, ci);
            if ((ri & kept) == 0) kept |= 1ull << ci;
        }
        const bool keep = wv == 0 && ((kept >> lane) & 1ull);
        const int total = __popcll(kept);
        if (keep) kb[m + __popcll(kept & lt)] = bj;
        if (tid == 0) s_base = total ? atomicAdd(&ws.count[b], total) : 0;
        __syncthreads();                         // kb and s_base visible
        if (keep) {
            const unsigned q = 0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull);
            const unsigned flat = q * (unsigned)C + (unsigned)c;
            ws.keys[(size_t)b * Q * C + s_base + __popcll(kept & lt)] =
                (keys[j] & 0xFFFFFFFF00000000ull) | (unsigned)(0xFFFFFFFFu - flat);
        }
        m += total;
        __syncthreads();                         // s_base, s_hit, s_row are rewritten by the next chunk
    }
}

// ---- per image: the max_num largest keys, sorted; decode and emit
constexpr int kTopThreads = 1024, kTopCap = 2048;
constexpr int kHistCopies = 64, kHistStride = 257;      // one histogram per lane id; the odd stride spreads a digit's
                                                        // 64 copies over the 64 LDS banks (no same-address atomics)

__global__ __launch_bounds__(kTopThreads) void nms_topk_kernel(const float *__restrict__ logits, int Q, int C,
                                                               int max_num, Workspace ws, float *__restrict__ dets,
                                                               int64_t *__restrict__ labels, int32_t *__restrict__ out_count)
{
    __shared__ unsigned long long sel[kTopCap];
    __shared__ int hist[kHistCopies * kHistStride];
    __shared__ int bins[256];
    __shared__ int s_digit, s_remaining, s_matching, s_fill;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int K = ws.count[b];
    const unsigned long long *keys = ws.keys + (size_t)b * Q * C;
    const int nout = K < max_num ? K : max_num;
    for (int i = tid; i < kTopCap; i += kTopThreads) sel[i] = 0;
    if (tid == 0) { s_fill = 0; s_remaining = nout; s_matching = K; }
    __syncthreads();
    if (K <= kTopCap) {
        for (int i = tid; i < K; i += kTopThreads) sel[i] = keys[i];
    } else {
        // Radix select, most significant byte first.  `prefix`/`mask` describe the keys still tied with the
        // nout-th largest one; taken = nout - remaining keys are known to be above them.  As soon as everything
        // not yet excluded fits the sort buffer the passes stop.
        unsigned long long prefix = 0, mask = 0;
        for (int pass = 7; pass >= 0; --pass) {
            for (int i = tid; i < kHistCopies * kHistStride; i += kTopThreads) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < K; i += kTopThreads) {
                const unsigned long long k = keys[i];
                if ((k & mask) == prefix) atomicAdd(&hist[lane * kHistStride + (int)((k >> (8 * pass)) & 255)], 1);
            }
            __syncthreads();
            if (tid < 256) {
                int v = 0;
                for (int cp = 0; cp < kHistCopies; ++cp) v += hist[cp * kHistStride + tid];
                bins[tid] = v;
            }
            __syncthreads();
            if (tid == 0) {
                int rem = s_remaining, d = 255;
                for (; d > 0; --d) {
                    if (bins[d] >= rem) break;
                    rem -= bins[d];
                }
                s_digit = d;
                s_remaining = rem;
                s_matching = bins[d];
            }
            __syncthreads();
            prefix |= (unsigned long long)s_digit << (8 * pass);
            mask |= 0xFFull << (8 * pass);
            if ((nout - s_remaining) + s_matching <= kTopCap) break;        // uniform
        }
        // everything above the tied group plus the tied group itself (<= kTopCap keys; exactly nout after 8 passes)
        for (int i = tid; i < K; i += kTopThreads) {
            const unsigned long long k = keys[i];
            if ((k & mask) >= prefix) sel[atomicAdd(&s_fill, 1)] = k;
        }
    }
    bitonic_desc<kTopThreads>(sel, kTopCap);
    for (int r = tid; r < nout; r += kTopThreads) {
        const unsigned flat = 0xFFFFFFFFu - (unsigned)(sel[r] & 0xFFFFFFFFull);
        const int q = (int)(flat / (unsigned)C), c = (int)(flat % (unsigned)C);
        const float4 bx = ws.boxes[(size_t)b * Q + q];
        float *o = dets + ((size_t)b * max_num + r) * 5;
        o[0] = bx.x; o[1] = bx.y; o[2] = bx.z; o[3] = bx.w;
        o[4] = sigmoidf_(logits[((size_t)b * Q + q) * C + c]);
        labels[(size_t)b * max_num + r] = c;
    }
    if (tid == 0) out_count[b] = nout;
}

// ---- weak -> strong box warp: 4 corners through the 3x3 matrix, bounding box of the images, clamp
__global__ __launch_bounds__(256) void transform_bboxes_kernel(const float *__restrict__ boxes, int box_stride,
                                                               const int32_t *__restrict__ offs,
                                                               const int32_t *__restrict__ counts,
                                                               const float *__restrict__ mats,
                                                               const float *__restrict__ out_hw,
                                                               float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int p0 = offs[b], K = counts ? counts[b] : offs[b + 1] - p0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K) return;
    const float *M = mats + 9 * b;
    const float *bx = boxes + (size_t)(p0 + i) * box_stride;
    const float px[4] = {bx[0], bx[2], bx[2], bx[0]}, py[4] = {bx[1], bx[1], bx[3], bx[3]};
    float minx = __builtin_huge_valf(), miny = minx, maxx = -minx, maxy = -minx;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x = M[0] * px[k] + M[1] * py[k] + M[2];
        const float y = M[3] * px[k] + M[4] * py[k] + M[5];
        const float z = M[6] * px[k] + M[7] * py[k] + M[8];
        const float u = x / z, v = y / z;
        minx = fminf(minx, u); maxx = fmaxf(maxx, u);
        miny = fminf(miny, v); maxy = fmaxf(maxy, v);
    }
    const float oh = out_hw[2 * b], ow = out_hw[2 * b + 1];
    float *o = out + (size_t)(p0 + i) * 4;
    o[0] = fminf(fmaxf(minx, 0.f), ow);
    o[1] = fminf(fmaxf(miny, 0.f), oh);
    o[2] = fminf(fmaxf(maxx, 0.f), ow);
    o[3] = fminf(fmaxf(maxy, 0.f), oh);
}

}  // namespace

extern "C" size_t semidetr_nms_workspace_bytes(int batch, int num_query, int num_classes)
{
    if (batch <= 0 || num_query <= 0 || num_classes <= 0) return 0;
    return workspace_bytes(batch, num_query, num_classes);
}

extern "C" int semidetr_pseudo_nms_f32(void *stream, const float *cls_logits, const float *bbox_pred,
                                       const float *img_hw, int batch, int num_query, int num_classes,
                                       float score_thr, float iou_thr, int max_per_img, void *workspace,
                                       size_t workspace_bytes_, float *out_dets, int64_t *out_labels,
                                       int32_t *out_count)
{
    const int B = batch, Q = num_query, C = num_classes;
    SEMIDETR_REQUIRE(B >= 0 && Q >= 0 && C >= 0, SEMIDETR_E_BADARG, "pseudo_nms: negative size");
    if (B == 0) return SEMIDETR_OK;
    SEMIDETR_REQUIRE(out_count, SEMIDETR_E_BADARG, "pseudo_nms: null out_count");
    hipStream_t st = semidetr::as_stream(stream);
    if (Q == 0 || C == 0) {
        hipError_t e = hipMemsetAsync(out_count, 0, sizeof(int32_t) * B, st);
        if (e != hipSuccess) return semidetr::fail((int)e, "pseudo_nms memset: %s", hipGetErrorString(e));
        return SEMIDETR_OK;
    }
    SEMIDETR_REQUIRE(max_per_img >= 1 && max_per_img <= kTopCap, SEMIDETR_E_BADARG,
                     "pseudo_nms: max_per_img must be in [1, %d] (got %d)", kTopCap, max_per_img);
    SEMIDETR_REQUIRE(cls_logits && bbox_pred && img_hw && out_dets && out_labels && workspace, SEMIDETR_E_BADARG,
                     "pseudo_nms: null pointer argument");
    SEMIDETR_REQUIRE(iou_thr >= 0.f, SEMIDETR_E_BADARG, "pseudo_nms: iou_thr must be >= 0 (got %g)", (double)iou_thr);
    SEMIDETR_REQUIRE(Q <= 2048, SEMIDETR_E_TOOLARGE, "pseudo_nms: at most 2048 queries per image (got %d)", Q);
    SEMIDETR_REQUIRE((int64_t)Q * C < (int64_t)0xFFFFFFFF, SEMIDETR_E_TOOLARGE, "pseudo_nms: Q * C too large");
    SEMIDETR_REQUIRE(workspace_bytes_ >= workspace_bytes(B, Q, C), SEMIDETR_E_BADARG,
                     "pseudo_nms: workspace too small (%zu < %zu bytes)", workspace_bytes_, workspace_bytes(B, Q, C));
    SEMIDETR_REQUIRE(((uintptr_t)workspace & 15) == 0, SEMIDETR_E_BADARG, "pseudo_nms: workspace must be 16-byte aligned");
    const Workspace ws = carve(workspace, B, Q, C);
    {   // maxc and count are adjacent 256-byte-aligned slots: one memset zeroes both (0 bits == 0.0f)
        hipError_t e = hipMemsetAsync(ws.maxc, 0, align256(sizeof(float) * (size_t)B) + sizeof(int) * (size_t)B, st);
        if (e != hipSuccess) return semidetr::fail((int)e, "pseudo_nms memset: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(nms_prepare_kernel, dim3((Q + kPrepQueries - 1) / kPrepQueries, B), dim3(256), 0, st, cls_logits, bbox_pred, img_hw, Q, C,
                       score_thr, ws);
    if (int rc = semidetr::launch_status("nms_prepare_kernel")) return rc;
    // one wavefront per problem for short candidate lists, four for the DINO query counts (the kept-list scan is
    // quadratic in the list length)
    if (Q <= 256)
        hipLaunchKernelGGL((nms_class_kernel<256, 1>), dim3(C, B), dim3(64), 0, st, cls_logits, Q, C, score_thr, iou_thr, ws);
    else if (Q <= 1024)
        hipLaunchKernelGGL((nms_class_kernel<1024, 4>), dim3(C, B), dim3(256), 0, st, cls_logits, Q, C, score_thr, iou_thr, ws);
    else
        hipLaunchKernelGGL((nms_class_kernel<2048, 4>), dim3(C, B), dim3(256), 0, st, cls_logits, Q, C, score_thr, iou_thr, ws);
    if (int rc = semidetr::launch_status("nms_class_kernel")) return rc;
    hipLaunchKernelGGL(nms_topk_kernel, dim3(B), dim3(kTopThreads), 0, st, cls_logits, Q, C, max_per_img, ws, out_dets,
                       out_labels, out_count);
    return semidetr::launch_status("nms_topk_kernel");
}

extern "C" int semidetr_transform_bboxes_f32(void *stream, const float *boxes, int box_stride,
                                             const int32_t *box_offsets, const int32_t *box_counts,
                                             int num_images, int max_boxes_per_image, const float *matrices,
                                             const float *out_hw, float *out_boxes)
{
    SEMIDETR_REQUIRE(num_images >= 0 && max_boxes_per_image >= 0, SEMIDETR_E_BADARG, "transform_bboxes: negative size");
    if (num_images == 0 || max_boxes_per_image == 0) return SEMIDETR_OK;
    SEMIDETR_REQUIRE(boxes && box_offsets && matrices && out_hw && out_boxes, SEMIDETR_E_BADARG,
                     "transform_bboxes: null pointer argument");
    SEMIDETR_REQUIRE(box_stride >= 4, SEMIDETR_E_BADARG, "transform_bboxes: box_stride must be >= 4 (got %d)", box_stride);
    hipLaunchKernelGGL(transform_bboxes_kernel, dim3((max_boxes_per_image + 255) / 256, num_images), dim3(256), 0,
                       semidetr::as_stream(stream), boxes, box_stride, box_offsets, box_counts, matrices, out_hw,
                       out_boxes);
    return semidetr::launch_status("transform_bboxes_kernel");
}
