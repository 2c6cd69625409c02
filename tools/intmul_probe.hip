// Throughput of the integer multiplies the address arithmetic uses, against full-rate VALU (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/intmul_probe.hip -o tools/bin/intmul_probe && tools/bin/intmul_probe
// Every thread runs a dependent chain of 8 independent accumulators x ITER instructions of one kind; time per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned a, unsigned b)
{
    unsigned x[8];
    unsigned long long y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 8 + i + a; y[i] = x[i]; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (KIND == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (KIND == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (KIND == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
            if (KIND == 4) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y[i]) : "v"(x[i]), "v"(b) : "vcc");
            if (KIND == 5) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (KIND == 6) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(y[i]));
            if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
            if (KIND == 8) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(y[i]) : "v"(x[i]), "v"(b) : "vcc");
            if (KIND == 9) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (unsigned)y[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
void run(const char *name, unsigned *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;      // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    k<KIND><<<blocks, 256>>>(out, 1, 3);
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, 1, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * ITER * 8;      // wave-instructions
    printf("%-16s %8.3f ms  %7.2f G wave-instr/s  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", name, ms, winstr / ms * 1e-6,
           2.4e9 * 1024 / (winstr / (ms * 1e-3)));
}
int main()
{
    unsigned *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", out); run<1>("v_mul_lo_u32", out); run<2>("v_mul_u32_u24", out); run<3>("v_mad_u32_u24", out);
    run<4>("v_mad_u64_u32", out); run<8>("v_mad_i64_i32", out); run<5>("v_mul_hi_u32", out); run<6>("v_lshl_add_u64", out);
    run<7>("v_fma_f32", out); run<9>("v_add3_u32", out);
    return 0;
}
