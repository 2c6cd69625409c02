// Measurement aids exported through the C ABI (not part of the hot path): a plain float4 streaming copy, so that
// bench.py can quote roofline fractions against the best streaming rate THIS box shows instead of a library memcpy.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// every thread moves kPer consecutive-by-stride float4s; loads first, then stores (kPer independent requests in flight)
constexpr int kPer = 8;
template <bool NT>
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                          int64_t n4)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t base = (int64_t)blockIdx.x * 256 + threadIdx.x; base < n4; base += stride * kPer) {
        f32x4 v[kPer];
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            if (base + i * stride < n4) v[i] = NT ? __builtin_nontemporal_load(src + base + i * stride) : src[base + i * stride];
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            if (base + i * stride < n4) {
                if (NT) __builtin_nontemporal_store(v[i], dst + base + i * stride);
                else dst[base + i * stride] = v[i];
            }
    }
}

}  // namespace

extern "C" int semidetr_stream_copy_f32(void *stream, float *dst, const float *src, int64_t numel, int nontemporal)
{
    SEMIDETR_REQUIRE(dst && src && numel > 0 && numel % 4 == 0, SEMIDETR_E_BADARG,
                     "stream_copy: need non-null pointers and a positive multiple of 4 elements");
    SEMIDETR_REQUIRE((((uintptr_t)dst | (uintptr_t)src) & 15) == 0, SEMIDETR_E_BADARG, "stream_copy: 16-byte alignment");
    const int64_t n4 = numel / 4;
    const int64_t want = (n4 + 256 * kPer - 1) / (256 * kPer);
    const unsigned grid = (unsigned)(want < 256 * 32 ? (want > 0 ? want : 1) : 256 * 32);
    if (nontemporal)
        hipLaunchKernelGGL(stream_copy_kernel<true>, dim3(grid), dim3(256), 0, semidetr::as_stream(stream),
                           reinterpret_cast<const f32x4 *>(src), reinterpret_cast<f32x4 *>(dst), n4);
    else
        hipLaunchKernelGGL(stream_copy_kernel<false>, dim3(grid), dim3(256), 0, semidetr::as_stream(stream),
                           reinterpret_cast<const f32x4 *>(src), reinterpret_cast<f32x4 *>(dst), n4);
    return semidetr::launch_status("stream_copy_kernel");
}
