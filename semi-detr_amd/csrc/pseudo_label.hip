// Teacher pseudo-label filter for gfx950: per-image adaptive threshold (mean + unbiased std of the
// scores), then drop degenerate boxes, compacted in post-NMS order -- one launch for the batch.
//
// Behavioural spec: detr_ssod/models/dino_detr_ssod.py:918-939 (a Python loop over images with
// torch.mean / torch.std / nonzero().unique() -- several launches and host syncs per image).
// One 256-thread workgroup per image; statistics in fp64 (torch accumulates mean/std of fp32 in double),
// order-preserving compaction by a workgroup prefix sum.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double *red)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void pseudo_label_kernel(
    const float *__restrict__ prop, const int64_t *__restrict__ labels, const int32_t *__restrict__ offs,
    const int32_t *__restrict__ counts, float *__restrict__ out_boxes, int64_t *__restrict__ out_labels, float *__restrict__ out_scores,
    int32_t *__restrict__ out_keep, int32_t *__restrict__ out_count, float *__restrict__ out_thr)
{
    __shared__ double red[4];
    __shared__ int wave_cnt[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int p0 = offs[b], K = counts ? counts[b] : offs[b + 1] - p0;
    const float *pb = prop + (int64_t)p0 * 5;
    if (K <= 0) {
        if (tid == 0) { out_count[b] = 0; out_thr[b] = __builtin_nanf(""); }
        return;
    }
    double s = 0.0;
    for (int i = tid; i < K; i += 256) s += (double)pb[5 * i + 4];
    const double mean = block_sum(s, red) / K;
    double ss = 0.0;
    for (int i = tid; i < K; i += 256) { const double d = (double)pb[5 * i + 4] - mean; ss += d * d; }
    ss = block_sum(ss, red);
    const float stdv = K > 1 ? (float)sqrt(ss / (K - 1)) : __builtin_nanf("");
    const float thr = __fadd_rn((float)mean, stdv);

    int base = 0;
    for (int i0 = 0; i0 < K; i0 += 256) {
        const int i = i0 + tid;
        bool keep = false;
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        float sc = 0.f;
        if (i < K) {
            box = make_float4(pb[5 * i], pb[5 * i + 1], pb[5 * i + 2], pb[5 * i + 3]);
            sc = pb[5 * i + 4];
            keep = (sc >= thr) && (__fsub_rn(box.z, box.x) > 0.f) && (__fsub_rn(box.w, box.y) > 0.f);
        }
        const unsigned long long mask = __ballot(keep);
        const int lane = tid & 63, wv = tid >> 6;
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) wave_cnt[wv] = __popcll(mask);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wv; ++w) wbase += wave_cnt[w];
        const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (keep) {
            const int64_t o = (int64_t)p0 + base + wbase + before;
            out_boxes[4 * o] = box.x; out_boxes[4 * o + 1] = box.y;
            out_boxes[4 * o + 2] = box.z; out_boxes[4 * o + 3] = box.w;
            out_scores[o] = sc;
            if (out_labels) out_labels[o] = labels[p0 + i];
            if (out_keep) out_keep[o] = i;
        }
        base += total;
    }
    if (tid == 0) { out_count[b] = base; out_thr[b] = thr; }
}

}  // namespace

extern "C" int semidetr_pseudo_label_filter_f32(void *stream, const float *proposals, const int64_t *labels,
                                                const int32_t *prop_offsets, const int32_t *prop_counts,
                                                int num_images, float *out_boxes, int64_t *out_labels, float *out_scores,
                                                int32_t *out_keep_idx, int32_t *out_count, float *out_thr)
{
    SEMIDETR_REQUIRE(num_images >= 0, SEMIDETR_E_BADARG, "pseudo_label: negative num_images");
    if (num_images == 0) return SEMIDETR_OK;
    SEMIDETR_REQUIRE(prop_offsets && out_count && out_thr, SEMIDETR_E_BADARG, "pseudo_label: null pointer argument");
    SEMIDETR_REQUIRE(!out_labels || labels, SEMIDETR_E_BADARG, "pseudo_label: out_labels without labels");
    hipLaunchKernelGGL(pseudo_label_kernel, dim3(num_images), dim3(256), 0, semidetr::as_stream(stream),
                       proposals, labels, prop_offsets, prop_counts, out_boxes, out_labels, out_scores, out_keep_idx,
                       out_count, out_thr);
    return semidetr::launch_status("pseudo_label_kernel");
}
