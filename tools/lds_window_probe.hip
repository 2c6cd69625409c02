// Probe (not product code): how much faster could the encoder forward's corner reads be if they came from an LDS window
// instead of the vector-memory path?  Two kernels doing the SAME synthetic work -- per workgroup 64 queries x 4 levels x 4
// points x 4 corner rows of 128 B, bilinear-weighted into one float4 per lane -- on a value map of the benchmark's size:
//   direct: 8 lanes x buffer-style 16-byte loads per corner straight from global memory (what msda_fwd_d32 does);
//   window: per level the workgroup first copies a WR-row window (rows 1 KB apart, as one head's rows are) into LDS, then
//           serves the corners with ds_read_b128 from pseudo-random rows of the window.
// Rows are picked by a hash, so the window kernel pays random-row LDS bank conflicts like the real thing would, and no
// records / misses / patch enumeration at all: an UPPER bound for an LDS-window forward.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_window_probe.hip -o /tmp/lds_window_probe && /tmp/lds_window_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kRowF4 = 8;             // 128-byte row = 8 float4
constexpr int kPitchF4 = 8 * 8;       // rows of one head are M * 128 B = 1 KB apart
constexpr int S = 22223, N = 4, M = 8;

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int WR>
__global__ __launch_bounds__(256) void k_direct(const float4 *__restrict__ value, float4 *__restrict__ out, int patches)
{
    const int wg = blockIdx.x, m = wg % M, patch = (wg / M) % patches, n = wg / M / patches;
    const int g = threadIdx.x >> 3, j = threadIdx.x & 7;
    const float4 *base = value + ((size_t)n * S * M + m) * kRowF4;
    float4 acc0 = make_float4(0, 0, 0, 0), acc1 = acc0;
    for (int half = 0; half < 2; ++half) {          // 2 x 32 queries, one per lane group
        float4 acc = make_float4(0, 0, 0, 0);
        const int q = half * 32 + g;
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const unsigned h = hash32((unsigned)(wg * 64 + q) * 16u + k);
            // window of level k / 4 starts somewhere depending on the patch; rows inside it
            const int wbase = (patch * 37 + (k >> 2) * 5000) % (S - WR - 40);
            const int r0 = wbase + (int)(h % (unsigned)(WR - 21)), r1 = r0 + 1, r2 = r0 + 20, r3 = r0 + 21;
            const float w = (float)(h >> 24) * (1.f / 256.f);
            const float4 v0 = base[(size_t)r0 * kPitchF4 + j], v1 = base[(size_t)r1 * kPitchF4 + j];
            const float4 v2 = base[(size_t)r2 * kPitchF4 + j], v3 = base[(size_t)r3 * kPitchF4 + j];
            acc.x += w * v0.x + (1 - w) * v1.x + w * v2.x + (1 - w) * v3.x;
            acc.y += w * v0.y + (1 - w) * v1.y + w * v2.y + (1 - w) * v3.y;
            acc.z += w * v0.z + (1 - w) * v1.z + w * v2.z + (1 - w) * v3.z;
            acc.w += w * v0.w + (1 - w) * v1.w + w * v2.w + (1 - w) * v3.w;
        }
        if (half == 0) acc0 = acc; else acc1 = acc;
    }
    out[((size_t)wg * 64 + g) * 8 + j] = acc0;
    out[((size_t)wg * 64 + 32 + g) * 8 + j] = acc1;
}

template <int WR>
__global__ __launch_bounds__(256) void k_window(const float4 *__restrict__ value, float4 *__restrict__ out, int patches)
{
    extern __shared__ float4 win[];                  // [WR][8]
    const int wg = blockIdx.x, m = wg % M, patch = (wg / M) % patches, n = wg / M / patches;
    const int g = threadIdx.x >> 3, j = threadIdx.x & 7;
    const float4 *base = value + ((size_t)n * S * M + m) * kRowF4;
    float4 acc0 = make_float4(0, 0, 0, 0), acc1 = acc0;
    for (int l = 0; l < 4; ++l) {
        const int wbase = (patch * 37 + l * 5000) % (S - WR - 40);
        __syncthreads();                             // previous level's reads done
        constexpr int kPer = (WR * 8 + 255) / 256;
        float4 t[kPer];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int e = threadIdx.x + i * 256;
            t[i] = e < WR * 8 ? base[(size_t)(wbase + (e >> 3)) * kPitchF4 + (e & 7)] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int e = threadIdx.x + i * 256;
            if (e < WR * 8) win[e] = t[i];
        }
        __syncthreads();
        for (int half = 0; half < 2; ++half) {
            float4 acc = half == 0 ? acc0 : acc1;
            const int q = half * 32 + g;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int k = l * 4 + p;
                const unsigned h = hash32((unsigned)(wg * 64 + q) * 16u + k);
                const int r0 = (int)(h % (unsigned)(WR - 21)), r1 = r0 + 1, r2 = r0 + 20, r3 = r0 + 21;
                const float w = (float)(h >> 24) * (1.f / 256.f);
                const float4 v0 = win[r0 * 8 + j], v1 = win[r1 * 8 + j], v2 = win[r2 * 8 + j], v3 = win[r3 * 8 + j];
                acc.x += w * v0.x + (1 - w) * v1.x + w * v2.x + (1 - w) * v3.x;
                acc.y += w * v0.y + (1 - w) * v1.y + w * v2.y + (1 - w) * v3.y;
                acc.z += w * v0.z + (1 - w) * v1.z + w * v2.z + (1 - w) * v3.z;
                acc.w += w * v0.w + (1 - w) * v1.w + w * v2.w + (1 - w) * v3.w;
            }
            if (half == 0) acc0 = acc; else acc1 = acc;
        }
    }
    out[((size_t)wg * 64 + g) * 8 + j] = acc0;
    out[((size_t)wg * 64 + 32 + g) * 8 + j] = acc1;
}

template <typename F>
float time_us(F launch, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

template <int WR>
void run(const float4 *value, float4 *out, int patches)
{
    const int grid = N * patches * M;
    const size_t lds = (size_t)WR * 128;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_window<WR>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    const float td = time_us([&] { hipLaunchKernelGGL(k_direct<WR>, dim3(grid), dim3(256), 0, 0, value, out, patches); }, 20);
    const float tw = time_us([&] { hipLaunchKernelGGL(k_window<WR>, dim3(grid), dim3(256), lds, 0, value, out, patches); }, 20);
    printf("window %4d rows (%5.1f KB LDS, %d workgroups/CU): direct %7.1f us   window %7.1f us   ratio %.2f\n", WR, lds / 1024.0,
           (int)(160 * 1024 / lds), td, tw, td / tw);
}

int main()
{
    const size_t nv = (size_t)N * S * M * kRowF4;
    const int patches = (S + 63) / 64;                        // 64 queries per workgroup
    float4 *value, *out;
    hipMalloc(&value, nv * sizeof(float4));
    hipMalloc(&out, (size_t)N * patches * M * 64 * 8 * sizeof(float4));
    hipMemset(value, 0, nv * sizeof(float4));
    printf("bs %d, %d patches of 64 queries per (image, head): %d workgroups; corner bytes per launch %.2f GB\n", N, patches,
           N * patches * M, (double)N * patches * M * 64 * 16 * 4 * 128 / 1e9);
    run<196>(value, out, patches);
    run<256>(value, out, patches);
    run<324>(value, out, patches);
    run<400>(value, out, patches);
    run<576>(value, out, patches);
    hipFree(value); hipFree(out);
    return 0;
}
