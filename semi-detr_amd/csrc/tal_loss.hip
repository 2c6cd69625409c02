// Task-aligned focal loss of the warm-up stage for gfx950, fused with the sigmoid in front of it: one streaming
// pass over the (N, C) logits produces the loss sum and (optionally) d loss / d logits; a second single-workgroup
// launch adds the per-workgroup partial sums in a fixed order (deterministic result).
//
// Behavioural spec:
//   task_aigned_focal_loss / TaskAlignedFocalLoss.forward   detr_od/models/losses/task_aligned_focal_loss.py:35-66, :166-200
//   called as loss_cls1(cls_scores.sigmoid(), labels, norm_alignment_metrics, avg_factor=...)
//                                                           detr_od/models/dense_heads/dino_detr_ssod_head.py:693-694
//   F.binary_cross_entropy (log terms clamped at -100; backward (p - s) / max((1 - p) p, 1e-12))
// The reference runs ~10 elementwise kernels forward and as many backward over (N, C) tensors per decoder layer.
//   soft label s_ic = metric_i if c == label_i else 0 (label == C: background row, all zeros)
//   loss_ic = |s - p|^gamma * BCE(p, s),  p = sigmoid(x_ic)
// HBM-streaming: reads 4 B/element, writes 4 B/element when the gradient is wanted.
#include <hip/hip_runtime.h>

#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int kTalThreads = 256, kTalMaxBlocks = 1024;

__global__ __launch_bounds__(kTalThreads) void tal_loss_kernel(const float *__restrict__ logits,
                                                               const int64_t *__restrict__ labels,
                                                               const float *__restrict__ metrics, int64_t total, int C,
                                                               float gamma, int input_is_prob,
                                                               float *__restrict__ partial, float *__restrict__ grad)
{
    __shared__ float red[kTalThreads / 64];
    float acc = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * kTalThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kTalThreads) {
        const int64_t i = e / C;
        const int c = (int)(e - i * C);
        const float s = labels[i] == c ? metrics[i] : 0.f;
        const float x = logits[e];
        const float p = input_is_prob ? x : 1.0f / (1.0f + expf(-x));
        const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.0f - p), -100.f);
        const float ce = -(s * lp + (1.0f - s) * l1p);
        const float d = s - p, ad = fabsf(d);
        const float mod = gamma == 2.0f ? ad * ad : powf(ad, gamma);
        acc += mod * ce;
        if (grad) {
            // d/dp: pow(|d|, gamma) -> -gamma |d|^(gamma-1) sign(d);   BCE -> (p - s) / max((1 - p) p, 1e-12)
            const float dmod = gamma == 2.0f ? -2.0f * d : (ad > 0.f ? -gamma * powf(ad, gamma - 1.0f) * (d > 0.f ? 1.f : -1.f) : 0.f);
            const float dce = (p - s) / fmaxf((1.0f - p) * p, 1e-12f);
            const float dp = dmod * ce + mod * dce;
            grad[e] = input_is_prob ? dp : dp * (p * (1.0f - p));
        }
    }
    for (int sft = 32; sft > 0; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(kTalThreads) void tal_reduce_kernel(const float *__restrict__ partial, int n,
                                                                 float *__restrict__ loss_sum)
{
    __shared__ double red[kTalThreads / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += kTalThreads) acc += (double)partial[i];
    for (int sft = 32; sft > 0; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss_sum = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

}  // namespace

extern "C" size_t semidetr_tal_loss_workspace_bytes(void) { return sizeof(float) * kTalMaxBlocks; }

extern "C" int semidetr_tal_loss_f32(void *stream, const float *logits, const int64_t *labels, const float *metrics,
                                     int64_t num_rows, int num_classes, float gamma, int input_is_prob, void *workspace,
                                     float *loss_sum, float *grad_logits)
{
    SEMIDETR_REQUIRE(num_rows >= 0 && num_classes > 0, SEMIDETR_E_BADARG, "tal_loss: bad sizes (N=%lld C=%d)",
                     (long long)num_rows, num_classes);
    SEMIDETR_REQUIRE(loss_sum && workspace, SEMIDETR_E_BADARG, "tal_loss: null pointer argument");
    hipStream_t st = semidetr::as_stream(stream);
    const int64_t total = num_rows * num_classes;
    if (total == 0) {
        hipError_t e = hipMemsetAsync(loss_sum, 0, sizeof(float), st);
        if (e != hipSuccess) return semidetr::fail((int)e, "tal_loss memset: %s", hipGetErrorString(e));
        return SEMIDETR_OK;
    }
    SEMIDETR_REQUIRE(logits && labels && metrics, SEMIDETR_E_BADARG, "tal_loss: null pointer argument");
    int blocks = (int)((total + kTalThreads * 4 - 1) / (kTalThreads * 4));
    if (blocks > kTalMaxBlocks) blocks = kTalMaxBlocks;
    float *partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL(tal_loss_kernel, dim3(blocks), dim3(kTalThreads), 0, st, logits, labels, metrics, total, num_classes,
                       gamma, input_is_prob, partial, grad_logits);
    if (int rc = semidetr::launch_status("tal_loss_kernel")) return rc;
    hipLaunchKernelGGL(tal_reduce_kernel, dim3(1), dim3(kTalThreads), 0, st, partial, blocks, loss_sum);
    return semidetr::launch_status("tal_reduce_kernel");
}
