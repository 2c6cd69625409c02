// Stand-alone probe: does it matter WHICH XCD reads WHICH 128-byte address class?
// Rows are 128 B; "class" c = (address / 128) % 8, i.e. the head index of an (N, S, 8, 32) fp32 value map.  Every
// workgroup reads pseudo-random rows of ONE class with the access shape of the MSDA forward (8 lanes x 16 B per
// row).  Workgroups land on XCD = HW XCC_ID (blockIdx % 8 in practice, read back from the hardware register);
// class = (xcd + shift) % 8.  Per (xcd, class) the probe reports the mean workgroup duration (s_memrealtime ticks)
// and, per shift, the kernel time -- if the classes were equally fast from every XCD, all shifts would tie.
//   region = number of rows per class that are touched (small: L2 resident; large: memory side)
// build: hipcc --offload-arch=gfx950 -O3 tools/xcd_class_probe.hip -o tools/bin/xcd_class_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void k(const float4 *buf, unsigned region_rows, int iters, int shift, int fixed_class,
                                         unsigned long long *dur, int *xcd_of, int *cls_of, float *sink)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;      // HW_REG_XCC_ID
    const int cls = fixed_class >= 0 ? fixed_class : (int)((xcc + shift) & 7);
    const unsigned g = (blockIdx.x * 256 + threadIdx.x) >> 3, j = threadIdx.x & 7;
    const unsigned long long t0 = __builtin_readcyclecounter();
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll 4
    for (int i = 0; i < iters; ++i) {
        const unsigned r = hash(g * 9781u + i * 7919u) % region_rows;           // pixel
        const float4 v = buf[((size_t)r * 8 + cls) * 8 + j];                     // row (r, cls), lane j's 16 bytes
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { dur[blockIdx.x] = t1 - t0; xcd_of[blockIdx.x] = (int)xcc; cls_of[blockIdx.x] = cls; }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = acc.x;
}

int main(int argc, char **argv)
{
    const unsigned region = argc > 1 ? atoi(argv[1]) : 22223 * 4;      // pixels: 4 images of the 800x1333 pyramid
    const int iters = argc > 2 ? atoi(argv[2]) : 256;
    const int blocks = 8 * 32 * 6 * 4;
    float4 *buf; float *sink; unsigned long long *dur; int *xcd_of, *cls_of;
    hipMalloc(&buf, (size_t)region * 8 * 128);
    hipMemset(buf, 0, (size_t)region * 8 * 128);
    hipMalloc(&sink, 4); hipMalloc(&dur, blocks * 8); hipMalloc(&xcd_of, blocks * 4); hipMalloc(&cls_of, blocks * 4);
    std::vector<unsigned long long> hd(blocks); std::vector<int> hx(blocks), hc(blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("region %u pixels (%.1f MB per class), %d rows per 8-lane group, %d workgroups\n", region, region * 128.0 / 1e6, iters, blocks);
    static double sum[8][8]; static long cnt[8][8];
    for (int mode = 0; mode < 2; ++mode) {            // 0: class = (xcd + shift) % 8 ; 1: every workgroup the same class
        for (int s = 0; s < 8; ++s) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                k<<<blocks, 256>>>(buf, region, iters, s, mode ? s : -1, dur, xcd_of, cls_of, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            hipMemcpy(hd.data(), dur, blocks * 8, hipMemcpyDeviceToHost);
            hipMemcpy(hx.data(), xcd_of, blocks * 4, hipMemcpyDeviceToHost);
            hipMemcpy(hc.data(), cls_of, blocks * 4, hipMemcpyDeviceToHost);
            double per_x[8] = {0}; long nx[8] = {0};
            for (int b = 0; b < blocks; ++b) {
                per_x[hx[b]] += (double)hd[b]; nx[hx[b]]++;
                if (mode == 0) { sum[hx[b]][hc[b]] += (double)hd[b]; cnt[hx[b]][hc[b]]++; }
            }
            const double gb = (double)blocks * 32 * iters * 128 / 1e9;
            printf("%s %d: kernel %.1f us (%.0f GB/s); mean workgroup ticks per XCD:", mode ? "all workgroups class" : "class = (xcd + shift) % 8, shift", s,
                   best * 1e3, gb / (best * 1e-3));
            for (int x = 0; x < 8; ++x) printf(" %.0f", nx[x] ? per_x[x] / nx[x] : 0.0);
            printf("\n");
        }
    }
    printf("mean workgroup ticks [xcd][class] (from the shifted runs):\n");
    for (int x = 0; x < 8; ++x) { for (int c = 0; c < 8; ++c) printf(" %8.0f", cnt[x][c] ? sum[x][c] / cnt[x][c] : 0.0); printf("\n"); }
    return 0;
}
