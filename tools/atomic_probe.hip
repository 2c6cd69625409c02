// Stand-alone probe: fp32 global atomic-add throughput on gfx950 for the access shapes the MSDA backward
// can generate.  Rows are 128 B (32 floats), chosen pseudo-randomly in a `rows`-row buffer.
//   mode 0: 8 lanes per row, lane j adds to dwords 4j..4j+3 with 4 instructions (stride-16B lanes)
//   mode 1: 32 lanes per row, lane c adds dword c (one instruction covers 2 full rows)
//   mode 2: 16 lanes per row, lane j adds dwords 2j,2j+1 (2 instructions, stride-8B lanes)
//   mode 4: like 1 but workgroup-scope atomics (executed in the XCD-local L2; only valid when one XCD owns the row)
//   mode 5: like 1 but with rows partitioned by XCD (row % 8 == HW XCC_ID), workgroup scope -- the safe form
//   mode 6: like 2 but only ONE of the wave's four 16-lane groups is active per instruction (the region scatter's flushes:
//           a stream reaches a row end while its three neighbours do not) -- same rows, 4x the wave-instructions
//   mode 3: like 0 but the 4 dwords are written lane-transposed: instr k, lane j -> dword 8k+j (contiguous 32B)
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_probe.hip -o /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void k(float *buf, unsigned rows, int per_thread_rows, unsigned locality)
{
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    constexpr int LPR = MODE == 0 || MODE == 3 ? 8 : (MODE == 1 || MODE == 4 || MODE == 5 ? 32 : 16);
    unsigned xcc = 0;
    if (MODE == 5) { xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7; }
    const unsigned grp = tid / LPR, j = tid % LPR;
    for (int i = 0; i < per_thread_rows; ++i) {
        unsigned r = hash(grp * 9781u + i * 7919u);
        if (locality) r = (grp * 3u + (r % locality));     // neighbouring groups hit neighbouring rows
        r %= rows;
        float *p = buf + (size_t)r * 32;
        const float v = 1.0f;
        if (MODE == 0) { unsafeAtomicAdd(p + 4 * j, v); unsafeAtomicAdd(p + 4 * j + 1, v); unsafeAtomicAdd(p + 4 * j + 2, v); unsafeAtomicAdd(p + 4 * j + 3, v); }
        else if (MODE == 1) { unsafeAtomicAdd(p + j, v); }
        else if (MODE == 4) { __hip_atomic_fetch_add(p + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        else if (MODE == 5) { float *pp = buf + (size_t)((r & ~7u) | xcc) % rows * 32; __hip_atomic_fetch_add(pp + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        else if (MODE == 2) { unsafeAtomicAdd(p + 2 * j, v); unsafeAtomicAdd(p + 2 * j + 1, v); }
        else if (MODE == 6) {
            for (int turn = 0; turn < 4; ++turn)
                if (((tid >> 4) & 3) == (unsigned)turn) { unsafeAtomicAdd(p + j, v); unsafeAtomicAdd(p + 16 + j, v); }
        }
        else { unsafeAtomicAdd(p + j, v); unsafeAtomicAdd(p + 8 + j, v); unsafeAtomicAdd(p + 16 + j, v); unsafeAtomicAdd(p + 24 + j, v); }
    }
}

int main(int argc, char **argv)
{
    const unsigned rows = argc > 1 ? atoi(argv[1]) : 711136;       // 4 images x 22223 x 8 heads
    const int total_row_updates = argc > 2 ? atoi(argv[2]) : 45 * 1000 * 1000;
    float *buf;
    hipMalloc(&buf, (size_t)rows * 128);
    hipMemset(buf, 0, (size_t)rows * 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (unsigned loc : {0u, 64u}) {
        for (int mode = 0; mode < 7; ++mode) {
            const int lpr = mode == 0 || mode == 3 ? 8 : (mode == 1 || mode == 4 || mode == 5 ? 32 : 16);
            const int per = 16;
            const long groups = total_row_updates / per;
            const long threads = groups * lpr;
            const int blocks = (int)((threads + 255) / 256);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, 256>>>(buf, rows, per, loc);
                if (mode == 1) k<1><<<blocks, 256>>>(buf, rows, per, loc);
                if (mode == 2) k<2><<<blocks, 256>>>(buf, rows, per, loc);
                if (mode == 3) k<3><<<blocks, 256>>>(buf, rows, per, loc);
                if (mode == 4) k<4><<<blocks, 256>>>(buf, rows, per, loc);
                if (mode == 5) k<5><<<blocks, 256>>>(buf, rows, per, loc);
                if (mode == 6) k<6><<<blocks, 256>>>(buf, rows, per, loc);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double upd = (double)groups * per;
            printf("locality %3u mode %d: %8.1f us  %7.2f G row-updates/s  %8.1f GB/s payload\n", loc, mode, ms * 1e3,
                   upd / ms / 1e6, upd * 128 / ms / 1e6);
        }
    }
    return 0;
}
