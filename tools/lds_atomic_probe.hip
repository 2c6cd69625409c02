// Probe: LDS fp32 atomic-add (ds_add_f32) throughput on gfx950, 32 lanes = one 128-byte row, rows picked
// pseudo-randomly inside a 256-row (32 KB) window; compares with plain LDS read-modify-write (non-atomic,
// wrong under contention, timing only) and with ds_add_u32.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_probe.hip -o tools/bin/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, int rows)
{
    __shared__ float win[256 * 32];
    for (int i = threadIdx.x; i < 256 * 32; i += 512) win[i] = 0.f;
    __syncthreads();
    const int hw = threadIdx.x >> 5, c = threadIdx.x & 31;
    unsigned h = blockIdx.x * 977u + hw * 131u;
    for (int i = 0; i < iters; ++i) {
        h = hash(h + i);
        const int r = h % rows;
        if (MODE == 0) atomicAdd(&win[r * 32 + c], 1.0f);
        else if (MODE == 1) win[r * 32 + c] += 1.0f;
        else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned *>(&win[r * 32 + c]), 1u);
        else { float old = atomicAdd(&win[r * 32 + c], 1.0f); if (old == -1.f) out[0] = old; }
    }
    __syncthreads();
    if (threadIdx.x < 32) out[blockIdx.x * 32 + threadIdx.x] = win[threadIdx.x];
}

int main()
{
    float *out; hipMalloc(&out, 4096 * 32 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 3, iters = 4096;
    const char *names[] = {"ds_add_f32 (no return)", "plain LDS rmw", "ds_add_u32", "ds_add_rtn_f32"};
    for (int rows : {256, 16, 1}) for (int mode = 0; mode < 4; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 512>>>(out, iters, rows);
            if (mode == 1) k<1><<<blocks, 512>>>(out, iters, rows);
            if (mode == 2) k<2><<<blocks, 512>>>(out, iters, rows);
            if (mode == 3) k<3><<<blocks, 512>>>(out, iters, rows);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double rowops = (double)blocks * 16 * iters;
        printf("rows %3d %-24s %8.1f us  %7.2f G row-ops/s  (%.2f clk per half-wave row-op per CU at 2.4 GHz)\n", rows,
               names[mode], ms * 1e3, rowops / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / rowops);
    }
    return 0;
}
