/*
 * semidetr_hip_experiments.h -- entry points that exist ONLY in libsemidetr_hip_exp.so (the product sources built with
 * -DSEMIDETR_EXPERIMENTS=1).  TEST / TUNING / MEASUREMENT aids: they are not part of the drop-in boundary and nothing
 * under semi-detr_amd/ needs them.  The experiments library is a superset of libsemidetr_hip.so (same ABI version, same
 * behaviour while no variant is forced).
 */
#ifndef SEMIDETR_HIP_EXPERIMENTS_H
#define SEMIDETR_HIP_EXPERIMENTS_H

#include "semidetr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Forces a kernel variant of the f32 / channels == 32 MSDA fast path for every later call of the process (0 = the
 * product dispatch; codes in DESIGN.md 2.3b).  Process-wide (two relaxed atomics); tests and tools reset it to (0, 0). */
void semidetr_msda_set_variant(int fwd_variant, int bwd_variant);

/* Per-phase cycle counters of the instrumented kernel builds (variants 73 / 696 / 707 / 7007): host array of 16 values;
 * reset != 0 zeroes the device counters after reading. */
int semidetr_debug_counters(unsigned long long *out16, int reset);

/* float4 streaming pass over `numel` fp32 values (multiple of 4, 16-byte aligned); mode 0 = copy, 1 = copy with
 * nontemporal accesses, 2 = read only.  bench.py quotes `frac_hbm_measured` against the best copy rate of the box. */
int semidetr_stream_copy_f32(void *stream, float *dst, const float *src, int64_t numel, int mode);

#ifdef __cplusplus
}
#endif
#endif
