// Probe: what streaming rate does this MI355X box reach?  float4 copy / read-only / write-only over 1 GiB, a few launch shapes.
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/bin/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PER, bool NT>
__global__ __launch_bounds__(256) void copy_k(const f4 *__restrict__ s, f4 *__restrict__ d, long n)
{
    const long stride = (long)gridDim.x * 256;
    for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < n; b += stride * PER) {
        f4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) if (b + i * stride < n) v[i] = NT ? __builtin_nontemporal_load(s + b + i * stride) : s[b + i * stride];
#pragma unroll
        for (int i = 0; i < PER; ++i) if (b + i * stride < n) { if (NT) __builtin_nontemporal_store(v[i], d + b + i * stride); else d[b + i * stride] = v[i]; }
    }
}
// contiguous chunk per workgroup (each workgroup streams its own region)
template <int PER>
__global__ __launch_bounds__(256) void copy_chunk_k(const f4 *__restrict__ s, f4 *__restrict__ d, long n)
{
    const long per_wg = (n + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per_wg, hi = lo + per_wg < n ? lo + per_wg : n;
    for (long b = lo + threadIdx.x; b < hi; b += 256 * PER) {
        f4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) if (b + i * 256 < hi) v[i] = s[b + i * 256];
#pragma unroll
        for (int i = 0; i < PER; ++i) if (b + i * 256 < hi) d[b + i * 256] = v[i];
    }
}
template <int PER>
__global__ __launch_bounds__(256) void read_k(const f4 *__restrict__ s, float *__restrict__ out, long n)
{
    const long stride = (long)gridDim.x * 256;
    f4 acc = {0, 0, 0, 0};
    for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < n; b += stride * PER) {
#pragma unroll
        for (int i = 0; i < PER; ++i) if (b + i * stride < n) acc += s[b + i * stride];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void write_k(f4 *__restrict__ d, long n)
{
    const long stride = (long)gridDim.x * 256;
    const f4 z = {1, 2, 3, 4};
    for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < n; b += stride) d[b] = z;
}
template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { hipEventRecord(e0); for (int i = 0; i < 10; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best / 10;
}
int main()
{
    const long bytes = 1L << 30, n = bytes / 16;
    f4 *a, *b; float *o; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 64);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    for (int g : {2048, 4096, 8192, 16384, 65536}) {
        printf("grid %6d  copy per8 %.0f  per8nt %.0f  per4 %.0f  per16 %.0f  chunk8 %.0f   read8 %.0f  write %.0f  GB/s\n", g,
               2.0 * bytes / timeit([&] { copy_k<8, false><<<g, 256>>>(a, b, n); }) / 1e6,
               2.0 * bytes / timeit([&] { copy_k<8, true><<<g, 256>>>(a, b, n); }) / 1e6,
               2.0 * bytes / timeit([&] { copy_k<4, false><<<g, 256>>>(a, b, n); }) / 1e6,
               2.0 * bytes / timeit([&] { copy_k<16, false><<<g, 256>>>(a, b, n); }) / 1e6,
               2.0 * bytes / timeit([&] { copy_chunk_k<8><<<g, 256>>>(a, b, n); }) / 1e6,
               1.0 * bytes / timeit([&] { read_k<8><<<g, 256>>>(a, o, n); }) / 1e6,
               1.0 * bytes / timeit([&] { write_k<<<g, 256>>>(b, n); }) / 1e6);
    }
    printf("hipMemcpyDtoD %.0f GB/s\n", 2.0 * bytes / timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }) / 1e6);
    return 0;
}
