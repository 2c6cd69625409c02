// Multi-scale deformable attention for gfx950 (MI355X, CDNA4): forward + backward, C-ABI entry points.
//
// Behavioural spec = the reference CUDA op (detr_od/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:
// forward :237-299 with bilinear fetch :33-84; backward :301-403 with scatter :87-159), re-designed for
// 64-wide wavefronts and the measured behaviour of this chip's atomic units -- not a translation.
//
// This file: shared sample geometry, the generic path (any D, fp32 / fp64: one wavefront per (n, q, m) row,
// lanes stride the channels) and the host-side dispatch.  msda_fast.h: the fp32 / D == 32 kernels
// (the DINO-DETR shape):
//  * forward: a 256-thread workgroup owns 32 queries of ONE (image, head) -- an 8 x 4 pixel patch for encoder
//    self-attention, 32 consecutive queries otherwise.  Workgroups are numbered head-fastest, so with the
//    observed block -> XCD round-robin every XCD's private 4 MiB L2 only sees value rows of "its" heads
//    (one head of one image = S * 128 B = 2.8 MB at 800x1333).  Phase 1 turns the tile's sampling locations /
//    attention weights into per-sample records {4 corner offsets, 4 weights} in LDS (the reference recomputes
//    them in all 32 channel threads and re-reads the int64 level table per sample); phase 2 covers one
//    128-byte value row with 8 lanes x float4, so every global_load_dwordx4 moves 8 complete cache lines.
//  * backward, any query set: 32 lanes per value row (lane = channel) so each global_atomic_add_f32
//    wave-instruction updates two COMPLETE lines (the L2 atomic unit charges per 64-byte half-line touched:
//    tools/atomic_probe.hip); channel sums for grad_attn / grad_loc by DPP inside the half-wave; small
//    gradients staged in LDS and written back coalesced.
//  * backward, encoder self-attention: a streaming gather kernel for the small gradients + an owner-computes
//    scatter kernel that buckets (sample, corner) pairs by target row in LDS with integer atomics only
//    (ds_add_f32 is lane-serial on gfx950: tools/lds_atomic_probe.hip) and issues one atomic per row run.
//  * every fast-path kernel is a template over an IO policy: the reference op contract (locations +
//    softmaxed weights) or the fused MSDeformAttn prologue (reference points + raw offsets + raw logits).
//
// Coordinates are computed WITHOUT fma contraction so cell selection (floor) is bit-identical to the
// CPU oracle; everything downstream may contract.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"
#if SEMIDETR_EXPERIMENTS
#include "semidetr_hip_experiments.h"
#endif

#ifndef SEMIDETR_SCATTER_SW      // 1 (tuning builds): the cell-sorted scatter (msda_sw.h, measured and rejected) instead of the region scatter
#define SEMIDETR_SCATTER_SW 0
#endif
#ifndef SEMIDETR_SW_NT
#define SEMIDETR_SW_NT 512
#define SEMIDETR_SW_Q 176
#define SEMIDETR_SW_RTH 8
#define SEMIDETR_SW_RTW 16
#define SEMIDETR_SW_WH 24
#define SEMIDETR_SW_WW 32
#define SEMIDETR_SW_WPE 4
#endif
namespace {

constexpr int kMaxLevels = 32;
// A/B knobs of tools/archive/r04_ab_multi.sh (same-box comparison of several builds inside the bench step).  Round 4, rotated inputs:
// encoder gather with 8 instead of 16 corner loads in flight at five waves per SIMD: backward 701 -> 695 us at bs 4 (with
// replayed inputs round 2 had measured it level); region scatter with 176 queries per pass at six waves per SIMD (three
// workgroups per CU): with 13 VGPRs spilled 702 vs 701 us at bs 4, 205 vs 196 us at bs 1; once the thread-derived constants were
// kept out of the kernel-long registers (msda_region.h: 102 -> 84 VGPRs at four waves, 2 spilled dwords at six) 700 -> 680 us at
// bs 4, 197 -> 194.5 us at bs 1, fused prologue 733 -> 712 us -- adopted.
#ifndef SEMIDETR_GATHER_WPE      // encoder gather: waves per SIMD the register budget is set for, corner loads in flight / 4
#define SEMIDETR_GATHER_WPE 5
#define SEMIDETR_GATHER_KB 2
#endif
#ifndef SEMIDETR_GATHER5_WPE     // ... the five-level (20-sample) instantiation
#define SEMIDETR_GATHER5_WPE 4      // (5 / 2 as for four levels: COCO-Full encoder backward 9.77 -> 9.69 ms per step, but the fused-prologue
#define SEMIDETR_GATHER5_KB 4       //  instantiation spills 4 registers there -- left as it is)
#endif
#ifndef SEMIDETR_SCATTER_Q       // region scatter: queries per pass (LDS) and waves per SIMD
#define SEMIDETR_SCATTER_Q 256     // (an 8 x 24 region of a halving pyramid has 192 + 48 + 12 + 3 queries: one pass; more take several)
#define SEMIDETR_SCATTER_WPE 6
#endif
#ifndef SEMIDETR_SCATTER_WU
#define SEMIDETR_SCATTER_WU 8      // entries per stream and trip of the row walk (+ 1000: paired corners, msda_region.h)
#endif
#ifndef SEMIDETR_SCATTER_NT
#define SEMIDETR_SCATTER_NT 512    // (704 threads = one sample per thread, no half-empty second sample slot, two workgroups per CU:
#endif                             //  encoder backward 726 against 678 us at bs 4 -- the third workgroup is worth more)
#ifndef SEMIDETR_SCATTER_TIGHT
#define SEMIDETR_SCATTER_TIGHT 1
#endif
#ifndef SEMIDETR_SCATTER_RTH     // region scatter: region (pixels of the finest level) and window per sampling level
// Round 5, after the flush path lost its 64-bit arithmetic (the kernel then ran at the atomic unit's rate, compute 328 of 398 us): 8 x 24
// regions, 24 x 40 windows, 256 queries per pass, 74.5 KB of LDS = TWO workgroups per CU flush 10 % fewer rows -- 400 -> 365 us at bs 4,
// 118 -> 114 us at bs 1 (tools/r05_ab_kern.sh; 8 x 16 / 176 queries / three per CU was the optimum while compute was the co-limit).
// Others measured: 12 x 16 390, 10 x 16 385, 8 x 20 387, 8 x 22 383, 8 x 24 with 272 queries (three samples per thread) 375, 576 / 640
// threads 414 / 406, 16 x 16 at one workgroup per CU 457-514.
#define SEMIDETR_SCATTER_RTH 8
#define SEMIDETR_SCATTER_RTW 24
#define SEMIDETR_SCATTER_WH 24
#define SEMIDETR_SCATTER_WW 40
#endif
#ifndef SEMIDETR_GW_NT           // msda_gw_d32 (lane-per-sample gather of the encoder backward): threads, region, margins of level 0 / the coarse levels
#define SEMIDETR_GW_NT 1024      // (round 5, second half: 768 -> 1024 once nothing spilled there)
#define SEMIDETR_GW_RTH 15      // (round 6: 15, not 16 -- the grad_out rows of a round need 8 KB of LDS beside the windows; a 100-row level is 7 x 14.3 rows either way)
#define SEMIDETR_GW_RTW 16
#define SEMIDETR_GW_H0 4
#define SEMIDETR_GW_HC 4
#endif
#ifndef SEMIDETR_GW_RTH5
#define SEMIDETR_GW_RTH5 13      // ... five levels: the windows of all five at margin 4 fit beside the query list with regions of up to 13 x 16 pixels
#define SEMIDETR_GW_NT5 1024     //     (a 100-row level = 8 x 12.5 rows; 14 rows -- the same tiling -- would need 165 KB)
#define SEMIDETR_GW_NT5_RAW 1024 //     (with the grad_out rows in LDS the fused prologue fits 1024 threads too: 117 / 128 VGPRs, no spill)
#endif
#ifndef SEMIDETR_GW_FB
#define SEMIDETR_GW_FB 4         // msda_gw_d32: far samples whose loads are in flight together
#endif
#ifndef SEMIDETR_GW_DBG
#define SEMIDETR_GW_DBG 0        // timing aids of msda_gw_d32 (results wrong), tuning builds only
#endif
#ifndef SEMIDETR_RW_NT
#define SEMIDETR_RW_NT 768       // msda_rw_d32: threads per workgroup.  Its windows take most of the LDS, so a CU holds ONE workgroup and
                                 // the workgroup's size is the CU's occupancy: 12 instead of 8 waves 192.9 -> 176.7 us inside the step
                                 // (1024 threads: only with margin 5 and 25 spilled registers so far, 262 us)
#endif
#ifndef SEMIDETR_RW_TAIL
#define SEMIDETR_RW_TAIL 1       // msda_rw_d32 forward: tail split (helper workgroups for the last, partly filled wave of workgroups)
#endif
#ifndef SEMIDETR_RW_DBG
#define SEMIDETR_RW_DBG 0        // timing aids of the four-level msda_rw_d32 (results wrong), tuning builds only: see msda_rw.h
#endif
#ifndef SEMIDETR_RW_TUNE
#define SEMIDETR_RW_TUNE 98320   // (round 5: + 6400 level constants from an LDS table, + 12800 region query list in LDS, + 25600 compact records for
                                 //  out-of-window samples, + 400 one such sample per octet and trip -- the geometry of a round 284 -> ~130 VALU
                                 //  instructions, no publish step: probe (tools/r05_ab_fwd.sh, medians of 5) 169.1 -> 164.4 us at sigma 1 px,
                                 //  179.4 -> 172.7 at 2 px, 202.3 -> 190.1 at 3 px; + 51200 window offsets as two 32-bit byte offsets: another -1 %)
                                 // 1920 = msda_rw_d32: 10 x compute-loop samples between scheduling barriers (two: with four samples' LDS reads
                                 // in flight round 4's first version spilled; one: 2 us slower) + 100: one level-0 sample's corner loads in
                                 // flight instead of two (-33 VGPRs) + 200: level constants re-selected where they are used and staging
                                 // coordinates rebuilt per region instead of living in registers (256 -> 160 VGPRs: what lets 768 threads run)
                                 // + 1600: the fused prologue's location arithmetic at the start of the round that uses the loaded data, not
                                 // where the loads are issued (a round early, waiting for them): fused-prologue forward 189.8 -> 188.0 us
#endif
#ifndef SEMIDETR_RW_SB_LOCATTN
#define SEMIDETR_RW_SB_LOCATTN 0      // added to SEMIDETR_RW_TUNE for the reference contract: 10 x (samples between barriers - 2).  Round 5: 20 (four
                                      // samples: -1.3 ... -2 %); round 6, with the branch-free round loop (msda_rw.h SEMIDETR_BRFREE; 129 VGPRs at any spacing):
                                      // two / three / four samples 162.5-163.8 / 168.5-171.0 / 164.3-165.4 us (tools/r05_ab_kern.sh, same box) -> two, like RawIO
#endif
#ifndef SEMIDETR_RW_TUNE_MASK
#define SEMIDETR_RW_TUNE_MASK 98320   // the instantiation with the padding mask (166 VGPRs; with the table but without the compact records it spills)
#endif
#ifndef SEMIDETR_RW_RTH
#define SEMIDETR_RW_RTH 25       // msda_rw_d32, four levels: LARGEST region height (x 16 columns; the grid is tiled evenly, msda_rw.h) and coarse-level
                                 // margin.  24 x 16 regions hold 510 queries
#endif
#ifndef SEMIDETR_RW_HC
#define SEMIDETR_RW_HC 5         // (5.3 rounds of 96: better balanced than 16 x 16's 3.5) and stage 1.8 instead of 2.9 window rows per query; margin 5
                                 // is the widest that fits then.  In the step 167.2 -> 162.2 us (16 x 16 / margin 6 -> 24 x 16 / margin 5; 32 x 16 /
                                 // margin 4: 165.6); by spread (probe): -6 % at 1 px, -1..-4 % at 2 px, level at 2.5 - 4 px, +3 % at 5 px
#endif
#ifndef SEMIDETR_RW_RTH5
#define SEMIDETR_RW_RTH5 24      // ... five levels.  With the evenly tiled grid (a 100-row level = 4 x 25 or 5 x 20 rows instead of 4 x 24 + 4), in
#endif                           // the step: four levels 25 / 24 rows 162.2 / 166.7 us (fused prologue 172.7 / 179.1; uneven 24: 162.0 / 176.5),
                                 // five levels 254 / 249 us (uneven 24: 257) -- what decides is how evenly a region's queries fill its rounds
#ifndef SEMIDETR_RW_NT5
#define SEMIDETR_RW_NT5 960      // ... the five-level instantiation: margin 4 is what fits either way; 24 x 16 regions like the four-level one, and the
                                 //     largest workgroup that fits beside their windows: 15 waves (16 x 16 / 1024 threads: 202 / 209 / 243 us at sigma 1 /
                                 //     2 / 3 px, 24 x 16 / 960: 195 / 204 / 241, / 896: 190 / 204 / 242)
#define SEMIDETR_RW_TUNE5 47910  // (round 5: as four levels -- split loads, level table, query list, compact records, one out-of-window sample per trip;
                                 //  the wide window offsets do not fit beside 120 octets' records.  The fused prologue keeps round 4's configuration (with table +
                                 //  query list it spills at 128 registers): SEMIDETR_RW_TUNE5_RAW / _MASK)
#ifndef SEMIDETR_RW_TUNE5_RAW
#define SEMIDETR_RW_TUNE5_RAW 47910      // (round 6: with no branch around the round loop's loads -- msda_rw.h SEMIDETR_BRFREE -- the fused prologue's five-level
#define SEMIDETR_RW_TUNE5_MASK 47910     //  instantiations fit the table / list / compact-record configuration too: 112 / 113 VGPRs, no spill; rounds 4-5: 1110)
#endif
                                 // 1110 =     (16 waves per CU, 128 VGPRs): ONE sample between scheduling barriers (three passes of samples per lane) and
                                 //     + 800: everything derived from the thread index rebuilt per round / region.  768 threads: 223 / 232 / 273 us
                                 //     at sigma 1 / 2 / 3 px, 1024: 208 / 225 / 255.  (Four levels at 1024 threads would have to give up margin 6
                                 //     for 5: 206 against 205 us -- no gain.)
#endif

// ---------------------------------------------------------------------------------------------
// sample geometry (ms_deform_im2col_cuda.cuh:285-288 pixel mapping, :56-78 corner validity)
// ---------------------------------------------------------------------------------------------
// Individually rounded multiply / subtract.  HIP's __fmul_rn / __fsub_rn are plain operators that hipcc's default
// -ffp-contract=fast fuses into one fma(y, H, -0.5); the pixel coordinate (hence floor(), the bilinear cell)
// must round exactly like the oracle's two C operations, so contraction is switched off for these four.
template <typename T>
__device__ __forceinline__ T mul_rn(T a, T b)
{
#pragma clang fp contract(off)
    return a * b;
}
template <typename T>
__device__ __forceinline__ T sub_rn(T a, T b)
{
#pragma clang fp contract(off)
    return a - b;
}

// off[i] = element offset of corner i relative to (value + n*S*M*D + m*D), or -1 when the corner is
// outside the level (zero padding).  Returns false when the whole sample is skipped.
template <typename T>
__device__ __forceinline__ bool sample_setup(T x, T y, int H, int W, int start, int rowstride,
                                             int (&off)[4], T &lw, T &lh)
{
    const T h = sub_rn(mul_rn(y, (T)H), (T)0.5);
    const T w = sub_rn(mul_rn(x, (T)W), (T)0.5);
    off[0] = off[1] = off[2] = off[3] = -1;
    lw = lh = 0;
    if (!(h > (T)-1 && w > (T)-1 && h < (T)H && w < (T)W)) return false;
    const int h0 = (int)floor(h), w0 = (int)floor(w);
    lh = sub_rn(h, (T)h0);
    lw = sub_rn(w, (T)w0);
    const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lef = w0 >= 0, rig = w0 + 1 <= W - 1;
    const int base = (start + h0 * W + w0) * rowstride;
    if (top && lef) off[0] = base;
    if (top && rig) off[1] = base + rowstride;
    if (bot && lef) off[2] = base + W * rowstride;
    if (bot && rig) off[3] = base + (W + 1) * rowstride;
    return true;
}

// Same geometry for the fast-path kernels that LOAD the corners, in the form the buffer instructions want:
// off[i] = BYTE offset of corner i inside the image's value slice (head / channel offset not included), or
// kOob for a corner outside the level / a skipped sample.  The kernels read the value map through a raw buffer
// resource whose size is exactly one image slice: the hardware bounds check of buffer_load returns 0 for kOob,
// which IS the op's zero padding -- no per-corner exec-mask branches, no zero-initialised destination
// registers in the hot loop (they were 1/3 of its VALU instructions), and values elsewhere in memory can never
// leak into a sample (NaN-safe exactly like the reference).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFF000u;       // + any in-row lane offset (< 4096) stays out of range, no wrap

__device__ __forceinline__ bool sample_setup_oob(float x, float y, int H, int W, int start, unsigned row_bytes,
                                                 unsigned (&off)[4], float &lw, float &lh)
{
    const float h = sub_rn(mul_rn(y, (float)H), 0.5f);
    const float w = sub_rn(mul_rn(x, (float)W), 0.5f);
    off[0] = off[1] = off[2] = off[3] = kOob;
    lw = lh = 0;
    if (!(h > -1.f && w > -1.f && h < (float)H && w < (float)W)) return false;
    const int h0 = (int)floorf(h), w0 = (int)floorf(w);
    lh = sub_rn(h, (float)h0);
    lw = sub_rn(w, (float)w0);
    const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lef = w0 >= 0, rig = w0 + 1 <= W - 1;
    const unsigned base = (unsigned)(start + h0 * W + w0) * row_bytes;     // may wrap for h0/w0 == -1: unused then
    if (top && lef) off[0] = base;
    if (top && rig) off[1] = base + row_bytes;
    if (bot && lef) off[2] = base + (unsigned)W * row_bytes;
    if (bot && rig) off[3] = base + (unsigned)(W + 1) * row_bytes;
    return true;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const float *image_base, unsigned image_bytes)
{
    // The descriptor is wave-uniform (it depends on blockIdx only) but the compiler cannot prove it and would
    // wrap every buffer op in a waterfall loop; readfirstlane of its inputs makes the uniformity explicit.
    const unsigned long long b = reinterpret_cast<unsigned long long>(image_base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane(image_bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, nb,
                                             0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float2 buf_ld2(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}
__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

__device__ __forceinline__ void fp_atomic_add(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void fp_atomic_add(double *p, double v) { unsafeAtomicAdd(p, v); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// generic path: one wavefront per (n, q, m) row
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_generic(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const T *__restrict__ loc, const T *__restrict__ attn, int N, int S, int M, int D, int L, int Lq,
    int P, T *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= (int64_t)N * Lq * M) return;
    const int m = (int)(row % M);
    const int n = (int)(row / ((int64_t)M * Lq));
    const T *vb = value + ((int64_t)n * S * M + m) * D;
    const T *lrow = loc + row * L * P * 2;
    const T *arow = attn + row * L * P;
    const int rs = M * D;
    for (int c0 = 0; c0 < D; c0 += 64) {
        const int c = c0 + lane;
        const bool act = c < D;
        T acc = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)starts[l];
            for (int p = 0; p < P; ++p) {
                int off[4];
                T lw, lh;
                const T x = lrow[(l * P + p) * 2], y = lrow[(l * P + p) * 2 + 1];
                if (!sample_setup(x, y, H, W, st, rs, off, lw, lh)) continue;
                const T a = arow[l * P + p];
                const T hh = 1 - lh, hw = 1 - lw;
                if (act) {
                    const T v1 = off[0] >= 0 ? vb[off[0] + c] : (T)0;
                    const T v2 = off[1] >= 0 ? vb[off[1] + c] : (T)0;
                    const T v3 = off[2] >= 0 ? vb[off[2] + c] : (T)0;
                    const T v4 = off[3] >= 0 ? vb[off[3] + c] : (T)0;
                    acc += a * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
                }
            }
        }
        if (act) out[row * D + c] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void msda_bwd_generic(
    const T *__restrict__ gout, const T *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const T *__restrict__ loc, const T *__restrict__ attn, int N,
    int S, int M, int D, int L, int Lq, int P, T *__restrict__ gvalue, T *__restrict__ gloc,
    T *__restrict__ gattn)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= (int64_t)N * Lq * M) return;
    const int m = (int)(row % M);
    const int n = (int)(row / ((int64_t)M * Lq));
    const int64_t vo = ((int64_t)n * S * M + m) * D;
    const T *vb = value + vo;
    T *gvb = gvalue + vo;
    const T *lrow = loc + row * L * P * 2;
    const T *arow = attn + row * L * P;
    const T *grow = gout + row * D;
    const int rs = M * D;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)starts[l];
        for (int p = 0; p < P; ++p) {
            int off[4];
            T lw, lh;
            const T x = lrow[(l * P + p) * 2], y = lrow[(l * P + p) * 2 + 1];
            const bool inside = sample_setup(x, y, H, W, st, rs, off, lw, lh);
            T s_attn = 0, s_x = 0, s_y = 0;
            if (inside) {
                const T a = arow[l * P + p];
                const T hh = 1 - lh, hw = 1 - lw;
                const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                for (int c = lane; c < D; c += 64) {
                    const T g = grow[c], ga = g * a;
                    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                    if (off[0] >= 0) { v1 = vb[off[0] + c]; fp_atomic_add(gvb + off[0] + c, w1 * ga); }
                    if (off[1] >= 0) { v2 = vb[off[1] + c]; fp_atomic_add(gvb + off[1] + c, w2 * ga); }
                    if (off[2] >= 0) { v3 = vb[off[2] + c]; fp_atomic_add(gvb + off[2] + c, w3 * ga); }
                    if (off[3] >= 0) { v4 = vb[off[3] + c]; fp_atomic_add(gvb + off[3] + c, w4 * ga); }
                    s_attn += g * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                    s_x += ga * (hh * (v2 - v1) + lh * (v4 - v3));
                    s_y += ga * (hw * (v3 - v1) + lw * (v4 - v2));
                }
                s_attn = wave_sum(s_attn);
                s_x = wave_sum(s_x);
                s_y = wave_sum(s_y);
            }
            if (lane == 0) {
                const int64_t k = row * L * P + l * P + p;
                gattn[k] = s_attn;
                gloc[2 * k] = (T)W * s_x;
                gloc[2 * k + 1] = (T)H * s_y;
            }
        }
    }
}

#include "msda_fast.h"   // IO policies + the D == 32 fp32 kernels
#if SEMIDETR_EXPERIMENTS
#include "msda_fast_experiments.h"   // windowed scatter, resident-level forward, 512-thread merged / cooperative-fill backward
#endif
#include "msda_region.h" // region-owned windowed scatter for encoder self-attention
#include "msda_rw.h"     // region-window forward (product since round 4) / gather (experiments) for encoder self-attention
#include "msda_gw.h"     // lane-per-sample region-window gather for the encoder backward (round 5)
#if SEMIDETR_SCATTER_SW
#include "msda_sw.h"     // cell-sorted region scatter (round 5): measured and rejected, tuning builds only (-DSEMIDETR_SCATTER_SW=1, tools/ab_build.sh)
#endif
#if SEMIDETR_EXPERIMENTS    // negative results kept as evidence: only in libsemidetr_hip_exp.so
#include "msda_dest.h"   // destination-owned grad_value kernel for encoder self-attention
#include "msda_lw.h"     // LDS-window forward for encoder self-attention
#include "msda_own.h"    // owner-computes grad_value for arbitrary query sets
#endif

thread_local const char *g_last_kernels = "";

int check_common(const void *value, const void *shapes, const void *starts, const void *loc,
                 const void *attn, int N, int S, int M, int D, int L, int Lq, int P, size_t elem)
{
    SEMIDETR_REQUIRE(value && shapes && starts && loc && attn, SEMIDETR_E_BADARG, "msda: null pointer argument");
    SEMIDETR_REQUIRE(N > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0, SEMIDETR_E_BADARG,
                     "msda: sizes must be positive (N=%d S=%d M=%d D=%d L=%d Lq=%d P=%d)", N, S, M, D, L, Lq, P);
    // device index arithmetic inside one image is 32-bit (as in the reference, .cuh:255-269)
    SEMIDETR_REQUIRE((int64_t)(S + 1) * M * D < INT32_MAX, SEMIDETR_E_TOOLARGE,
                     "msda: spatial_size*num_heads*channels = %lld exceeds 32-bit indexing", (long long)S * M * D);
    SEMIDETR_REQUIRE((int64_t)N * Lq * M < INT32_MAX / 4, SEMIDETR_E_TOOLARGE, "msda: too many (n,q,m) rows");
    (void)elem;
    return SEMIDETR_OK;
}

template <typename T>
int forward_impl(void *stream, const T *value, const int64_t *shapes, const int64_t *starts, const T *loc,
                 const T *attn, int N, int S, int M, int D, int L, int Lq, int P, T *out)
{
    if (int rc = check_common(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P, sizeof(T))) return rc;
    SEMIDETR_REQUIRE(out, SEMIDETR_E_BADARG, "msda_forward: null output");
    const int64_t rows = (int64_t)N * Lq * M;
    const int blocks = (int)((rows + 3) / 4);
    hipLaunchKernelGGL(msda_fwd_generic<T>, dim3(blocks), dim3(256), 0, semidetr::as_stream(stream), value,
                       shapes, starts, loc, attn, N, S, M, D, L, Lq, P, out);
    g_last_kernels = "msda_fwd_generic";
    return semidetr::launch_status("msda_fwd_generic");
}

template <typename T>
int backward_impl(void *stream, const T *gout, const T *value, const int64_t *shapes, const int64_t *starts,
                  const T *loc, const T *attn, int N, int S, int M, int D, int L, int Lq, int P, T *gvalue,
                  T *gloc, T *gattn)
{
    if (int rc = check_common(value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P, sizeof(T))) return rc;
    SEMIDETR_REQUIRE(gout && gvalue && gloc && gattn, SEMIDETR_E_BADARG, "msda_backward: null pointer argument");
    hipError_t e = hipMemsetAsync(gvalue, 0, sizeof(T) * (size_t)N * S * M * D, semidetr::as_stream(stream));
    if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward memset: %s", hipGetErrorString(e));
    const int64_t rows = (int64_t)N * Lq * M;
    const int blocks = (int)((rows + 3) / 4);
    hipLaunchKernelGGL(msda_bwd_generic<T>, dim3(blocks), dim3(256), 0, semidetr::as_stream(stream), gout,
                       value, shapes, starts, loc, attn, N, S, M, D, L, Lq, P, gvalue, gloc, gattn);
    g_last_kernels = "fillBufferAligned+msda_bwd_generic";
    return semidetr::launch_status("msda_bwd_generic");
}

// fast-path applicability: channels == 32, 16-byte aligned pointers, LDS budget
bool fast_ok(const void *a, const void *b, const void *c, int D, int L, int P)
{
    const uintptr_t al = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
    return D == kD && L <= kMaxLevels && (al & 15) == 0 && (int64_t)L * P <= 256;
}

// kOob + (head offset + lane offset) must stay out of range without wrapping: the in-row byte offset of a lane is
// < num_heads * 128, so num_heads <= 32 keeps it below the 4096-byte guard band of kOob.
bool heads_ok(int M) { return (int64_t)M * kD * 4 <= 4096; }

// the fast-path kernels address one image's value slice with 32-bit BYTE offsets (buffer instructions)
bool slice_ok(int S, int M) { return (int64_t)S * M * kD * 4 < (int64_t)0xFFFFF000u; }

int pick_split(int forced, int N, int Lq, int M)
{
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    // enough 32-row workgroups to fill 256 CUs several times over -> no split; otherwise spread the
    // samples of a row over more lanes so small problems (decoder, 300-900 queries) still fill the chip.
    const int64_t wg32 = (int64_t)N * M * ((Lq + 31) / 32);
    if (wg32 >= 2048) return 1;
    if (wg32 >= 1024) return 2;
    return 4;
}


// dynamic LDS above 64 KB has to be allowed per kernel AND per device (function attributes are per device: ADVICE r02)
template <typename K>
int allow_big_lds(K kern, size_t bytes, const char *what)
{
    if (bytes <= 64 * 1024) return SEMIDETR_OK;
    // what was granted, per (kernel, device) of this thread: kernels of one signature share the pointer TYPE, so the
    // bookkeeping is keyed on the pointer VALUE
    struct Granted { const void *kern; int dev; size_t bytes; };
    static thread_local Granted table[32];
    static thread_local int used = 0;
    const void *kp = reinterpret_cast<const void *>(kern);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return semidetr::fail((int)e, "%s: hipGetDevice: %s", what, hipGetErrorString(e));
    Granted *g = nullptr;
    for (int i = 0; i < used; ++i)
        if (table[i].kern == kp && table[i].dev == dev) g = &table[i];
    if (g && g->bytes >= bytes) return SEMIDETR_OK;
    e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return semidetr::fail((int)e, "%s: hipFuncSetAttribute(%zu bytes of LDS): %s", what, bytes, hipGetErrorString(e));
    if (!g && used < 32) g = &table[used++];
    if (g) *g = Granted{kp, dev, bytes};           // a full table only costs a repeated hipFuncSetAttribute
    return SEMIDETR_OK;
}

// ---- which kernel runs the encoder self-attention forward: the patch kernel (msda_fwd_d32<1,4,408>) or the region-window
//      kernel (msda_rw_d32, LDS windows +- 5 px around a region, five levels +- 4 px).  The second is ~30 % faster while the learned
//      offsets keep most samples within a few pixels of their queries (sigma <= 2 px: 162 against 234 us at bs 4 inside the step) and slower
//      once ~70 % of them are further than 4 px away (sigma ~5.5 px; profiles/r04_region_window_dispatch.txt), so the choice
//      follows the DATA: both kernels count how far their samples reach (FwdStats, msda_fast.h), the count of launch k is
//      handed to the host by the first thread of launch k + 1 through mapped pinned memory, and the dispatcher -- whenever it
//      next gets here, it never waits -- moves between the kernels with hysteresis.  Per (device, call-site slot); a mutex serialises
//      the few host words.  semidetr_msda_set_forward_policy pins the choice (tests, A/B timing).
// Thresholds re-measured with the 768 / 1024-thread window kernels (tools/archive/r04_rw_cross2.sh, rotated inputs, bs 4): four levels 233 /
// 243 / 251 / 261 / 283 us against the patch kernel's 270-273 us at sigma 3.5 / 4 / 4.5 / 5 / 6 px -- level at ~5.5 px = a far share of
// ~0.71; five levels level at ~5 px = ~0.67.  One image alone (616 regions x heads for 256 CUs) used to need its own, lower bound;
// the larger workgroups removed the difference (57 / 63 / 65.5 / 68.5 against 68 / 70.5 / 71.5 / 72 us at 2 / 3 / 3.5 / 4 px).
// Round 5 (the window kernel's geometry halved, tools/r05_crossover.sh): four levels, bs 4: 193 / 213 / 232 / 246 / 255 / 260 us against the patch
// kernel's 263-270 at sigma 3 / 4 / 5 / 6 / 7 / 8 px -- level beyond 8 px (far share 0.85); one image: 56 / 63 / 68 / 70.5 / 74.5 against 70-71 --
// level at ~6 px (0.76).  The band moves from 0.60 / 0.70 to 0.72 / 0.80.
// Round 6 (tools/r06_spread.sh, profiles/r06_sample_patterns.txt; Gaussian spreads AND the reference's own initial offset star,
// ops/modules/ms_deform_attn.py:62-70): four levels, bs 4 -- window 197 / 215 / 229 / 246 us against the patch kernel's 250 / 253 / 253 / 249 at sigma
// 4 / 5 / 6 / 8 px (far share 0.53 / 0.67 / 0.76 / 0.85): level at ~0.85; star 140 against 222, star x 2 (far share 0.57) 200 against 220.  Five levels:
// window 233 / 252 / 261 / 274 against 299 / 301 / 301 / 298 -- ahead at every spread measured (round 5's band 0.56 / 0.66 predates the level table and
// the compact records in the five-level kernel and sent sigma >= 5 px to the patch kernel: 301 instead of 252 us).
constexpr float kFarToWindow = 0.78f;      // patch -> window when fewer than this share of the samples are far ...
constexpr float kFarToPatch = 0.86f;       // ... window -> patch above this one
constexpr float kFarToWindow5 = 0.82f, kFarToPatch5 = 0.90f;      // five levels (COCO-Full pyramid)
// Per CALL SITE (round 5): the reference builds twelve MSDeformAttn instances per model (transformer.py:609,760) whose learned offsets
// reach differently far, so the state is kept per (device, slot): the caller names the slot in bits 8..15 of `flags`
// (SEMIDETR_MSDA_POLICY_SLOT; the Python module gives every instance its own), slot 0 is the state every caller that names none shares.
// A slot's counts come from its own launches only and are judged by the thresholds of its own pyramid.
constexpr int kPolicySlots = 256;
struct FwdSlot {
    unsigned launches = 0, seen_seq = 0, updates = 0;
    int mode = 0;                          // 0 = patch kernel, 1 = region-window kernel
    float last_frac = -1.f;
};
struct FwdAdapt {
    unsigned *dev_cnt = nullptr;           // device words per slot: {far, total, kind, -} x launch parity, [8] = publications so far
    unsigned *pub_host = nullptr;          // mapped pinned memory per slot: {sequence, far, total, kind}
    unsigned *pub_dev = nullptr;           // the same memory as the device sees it
    bool failed = false;                   // allocation failed once: stay with the patch kernel, silently
    FwdSlot slot[kPolicySlots];
};
constexpr int kMaxDevices = 64;
std::mutex g_adapt_mu;
FwdAdapt *g_adapt[kMaxDevices];            // allocated at a device's first adaptive dispatch
std::atomic<int> g_fwd_policy{0};          // 0 adaptive, 1 always the patch kernel, 2 the window kernel whenever it applies

// the launch's FwdStats and the kernel to use; called with the stream the launch goes to
// compute units of the current device (cached per device index; 0 if the runtime will not say: the callers then do without)
int device_cus()
{
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        int v = 0;
        c = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? v : -1;
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c > 0 ? c : 0;
}

int fwd_adapt_next(hipStream_t st, bool allow_window, int slot_id, int levels, FwdStats &fs, bool &use_window)
{
    fs = FwdStats{nullptr, 0u};
    const int policy = g_fwd_policy.load(std::memory_order_relaxed);
    use_window = allow_window && policy == 2;
    if (policy != 0 || !allow_window) return SEMIDETR_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return SEMIDETR_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lock(g_adapt_mu);
    if (!g_adapt[dev]) g_adapt[dev] = new (std::nothrow) FwdAdapt();
    if (!g_adapt[dev]) return SEMIDETR_OK;
    FwdAdapt &a = *g_adapt[dev];
    if (!a.dev_cnt && !a.failed && !capturing) {       // first use on this device (never inside a stream capture): two small
        // allocations and a SYNCHRONOUS clear (ONCE per process and device; the allocations may synchronise anyway).  The counter block
        // is shared by every slot and every stream of the device: a clear merely queued on the first launch's stream could run after
        // another stream's first counting launch had started adding.
        void *h = nullptr, *d = nullptr, *c = nullptr;
        // (the device block of a slot: counters zero, words [10..11] = where the slot's record lives in the mapped host memory -- FwdStats)
        std::vector<unsigned> init((size_t)kPolicySlots * 16, 0u);
        if (hipHostMalloc(&h, kPolicySlots * 16, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess &&
            hipMalloc(&c, kPolicySlots * 64) == hipSuccess) {
            for (int sid = 0; sid < kPolicySlots; ++sid) {
                const unsigned *pp = static_cast<unsigned *>(d) + 4 * sid;
                std::memcpy(&init[(size_t)sid * 16 + 10], &pp, sizeof(pp));
            }
        }
        if (h && d && c && hipMemcpy(c, init.data(), init.size() * sizeof(unsigned), hipMemcpyHostToDevice) == hipSuccess) {
            std::fill_n(static_cast<unsigned *>(h), kPolicySlots * 4, 0u);
            a.pub_host = static_cast<unsigned *>(h);
            a.pub_dev = static_cast<unsigned *>(d);
            a.dev_cnt = static_cast<unsigned *>(c);
        } else {
            (void)hipGetLastError();
            a.failed = true;
        }
    }
    FwdSlot &sl = a.slot[slot_id & (kPolicySlots - 1)];
    if (a.dev_cnt) {
        const int sid = slot_id & (kPolicySlots - 1);
        volatile unsigned *pub = a.pub_host + 4 * sid;
        const unsigned seq = pub[0];
        if (seq != sl.seen_seq) {                      // a finished launch's counts have arrived since the last look
            const unsigned far = pub[1], total = pub[2];
            if (pub[0] != seq || far > total) {        // caught between two records (the device writes 16 bytes at once,
                                                       // so this is paranoia): look again at the next dispatch
            } else if (total) {
                sl.seen_seq = seq;
                sl.last_frac = (float)far / (float)total;
                ++sl.updates;
                if (sl.mode == 0 && sl.last_frac < (levels == 5 ? kFarToWindow5 : kFarToWindow)) sl.mode = 1;
                else if (sl.mode == 1 && sl.last_frac > (levels == 5 ? kFarToPatch5 : kFarToPatch)) sl.mode = 0;
            }
        }
        if (!capturing) {                              // a captured launch keeps the kernel of the moment and counts nothing
            const unsigned par = sl.launches++ & 1u;
            unsigned *base = a.dev_cnt + 16 * sid;
            fs = FwdStats{base, par};
        }
    }
    use_window = sl.mode == 1;
    return SEMIDETR_OK;
}

// What the slot's forward launches counted about its samples, for the BACKWARD's gather (same data, one forward earlier): true = few enough
// of them are far from their queries for the lane-per-sample window gather (msda_gw.h) to pay.  It crosses the patch gather earlier
// than the window forward crosses the patch forward (tools/r05_gw_sigma.sh, bs 4, whole backward: 572 / 635 / 796 / 976 / 1157 us
// against 659 / 707 / 834 / 986 / 1111 us at sigma 1 / 2 / 3 / 4 / 5 px; bs 1 level at 3 px): below a far share of ~0.45
// (sigma ~3.5 px).  Reads the state, counts nothing; no count received yet = the patch gather.
// Round 6, with a round's grad_out rows in LDS (same script): whole backward, four levels, window gather 644 / 794 / 933 / 1091 us against the patch
// gather's 715 / 830 / 933 / 1062 at sigma 3 / 4 / 5 / 6 px (far share 0.33 / 0.53 / 0.67 / 0.76); the reference's star 490 against 577, but the star
// twice as far (far share 0.57: half the samples 6 and 8 px out, all of them on the cooperative path) 620 against 574 -- the patch gather profits from
// the star's regularity.  Five levels: 761 / 910 / 1072 against 856 / 965 / 1088 at sigma 3 / 4 / 5 px, star x 2 794 against 770.  0.45 -> 0.55.
constexpr float kFarToWindowGather = 0.55f;
bool slot_samples_are_near(int slot_id)
{
    const int policy = g_fwd_policy.load(std::memory_order_relaxed);
    if (policy != 0) return policy == 2;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return false;
    std::lock_guard<std::mutex> lock(g_adapt_mu);
    if (!g_adapt[dev]) return false;
    const FwdSlot &sl = g_adapt[dev]->slot[slot_id & (kPolicySlots - 1)];
    return sl.updates > 0 && sl.last_frac >= 0.f && sl.last_frac < kFarToWindowGather;
}

// ---- product dispatch of the fast path (fp32, channels == 32), shared by the reference contract (LocAttnIO) and the
//      fused prologue (RawIO).  Apart from the forward-kernel choice above, what runs is a function of the arguments only.
template <typename IO>
int launch_fast_forward(hipStream_t st, const float *value, const int64_t *spatial_shapes,
                        const int64_t *level_start, const IO &io, int N, int S, int M, int L, int Lq, int P,
                        int flags, float *out)
{
    SEMIDETR_REQUIRE(!(flags & SEMIDETR_MSDA_QUERIES_ARE_PIXELS) || Lq == S, SEMIDETR_E_BADARG,
                     "msda_forward: SEMIDETR_MSDA_QUERIES_ARE_PIXELS needs num_query == spatial_size");
    // (a patch's 32 x (L * P + 1) records have to fit 64 KB: pyramids of more than 15 levels x 4 points take the strips)
    const bool pixels = (flags & SEMIDETR_MSDA_QUERIES_ARE_PIXELS) != 0 && (size_t)32 * (L * P + 1) * 32 <= 63 * 1024;
#define LAUNCH_FWD(SP, PT, TILES, LDS)                                                                          \
    hipLaunchKernelGGL((msda_fwd_d32<SP, 4, PT, IO>), dim3((unsigned)((int64_t)N * (TILES) * M)), dim3(256), \
                       (LDS), st, value, spatial_shapes, level_start, io, S, M, L, Lq, P, (TILES), out)
    if (pixels) {
        // encoder self-attention.  The region-window kernel is built for num_point == 4 and four or five levels.  A padding mask
        // (fused prologue; the reference ALWAYS passes one: transformer.py:1309,1460 -> ops/modules/ms_deform_attn.py:95-96) has its
        // own instantiation: padded rows are staged as zeros, level 0 drops padded corners from its records (msda_rw.h).
        // SEMIDETR_MSDA_FIXED_FORWARD: the caller wants the kernel to be a function of the arguments alone (bitwise reproducible
        // forward: the two kernels sum in different orders) -- the patch kernel, whatever the policy says.
        // (an image's sampling data and output below 2^32 bytes: the window kernel indexes them with 32 bits inside the image)
        const bool window_ok = P == kPT && (L == 4 || L == 5) && !(flags & SEMIDETR_MSDA_FIXED_FORWARD) &&
                               (uint64_t)S * M * L * P * 8 < (1ull << 32) && (uint64_t)S * M * kD * 4 < (1ull << 32);
        FwdStats fs;
        bool use_window = false;
        if (int rc = fwd_adapt_next(st, window_ok, (flags >> 8) & 0xff, L, fs, use_window)) return rc;
        if (use_window) {
            bool fell_back = false;      // a device / runtime that does not grant the window kernel its ~150 KB of LDS gets the patch kernel, not an error
            // level 0 through global loads, windows of the coarse levels with the widest margin that fits beside the
            // octet records: one workgroup per CU either way, and the workgroup as large as its registers allow (SEMIDETR_RW_NT).
            //   four levels: 24 x 16 regions, margin FIVE, 113 KB of windows + 34.5 KB of records (16 x 16 regions at margin 4 / 5 / 6 and
            //                sigma 2 px: 239 / 231 / 219-229 us, at 3 px: 290 / 265 / 252 us with 512 threads; SEMIDETR_RW_RTH above)
            //   five levels: 24 x 16 regions, margin FOUR (102 + 53 KB; margin 5 fits only a 640-thread workgroup), 960 threads; patch
            //                kernel 314 / 292 / 299 us at sigma 1 / 2 / 3 px, this one 195 / 204 / 241
            // kern_tail: the same configuration with the tail split compiled in (TUNE + 102400; null: this configuration has none)
            auto launch_window = [&](auto kern, decltype(kern) kern_tail, size_t wlds, int threads, int region_px) -> int {
                // grid sizing hint: the finest level of a DETR pyramid holds ~3/4 of the pixels; a workgroup takes regions slot,
                // slot + bound, ... so any bound >= 1 is correct (the level table lives in device memory)
                const int wbound = ((S * 3 / 4 + region_px - 1) / region_px) * 9 / 8 + 2 * L;
                // + one helper workgroup per CU for the tail split (msda_rw.h: the units of the last, partly filled wave of workgroups are
                // cut into parts; one workgroup per CU is what the kernel's LDS allows).  SEMIDETR_RW_TAIL=0 builds do without.
                // Only launches of fewer than ~three waves of workgroups get them (the kernel's own rule, from the real region count): on a
                // bs-4 launch 256 idle helpers, each waiting for a whole CU to read the level table and leave, cost 7 us of 154.
                const int cus = (SEMIDETR_RW_TAIL && kern_tail != nullptr) ? device_cus() : 0;
                const int tail = (int64_t)N * M * ((S * 3 / 4 + region_px - 1) / region_px) < (int64_t)3 * cus ? cus : 0;
                if (tail > 0) kern = kern_tail;
                if (int rc = allow_big_lds(kern, wlds, "msda_forward")) {      // refused for this instantiation: the patch kernel below
                    fell_back = true;
                    return rc;
                }
                SEMIDETR_REQUIRE((int64_t)N * wbound * M + tail < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_forward: grid too large");
                hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)N * wbound * M + tail)), dim3(threads), wlds, st, (const float *)nullptr,
                                   value, spatial_shapes, level_start, io, S, M, wbound, out, (float4 *)nullptr, (int64_t)0, fs, tail);
                g_last_kernels = "msda_rw_d32";
                return semidetr::launch_status("msda_rw_d32<forward>");
            };
            auto pick_window = [&]() -> int {
                constexpr int kTune5 = std::is_same<IO, RawIO>::value ? SEMIDETR_RW_TUNE5_RAW : SEMIDETR_RW_TUNE5;
                // (the reference contract's main instantiation has the registers for THREE (round 5: four) samples between scheduling barriers since the
                //  compact records -- 168 VGPRs, no spill: -1.3 ... -2 % in the probe; the fused prologue's and the tail-split ones spill there)
                constexpr int kTune4 = std::is_same<IO, RawIO>::value ? SEMIDETR_RW_TUNE : SEMIDETR_RW_TUNE + SEMIDETR_RW_SB_LOCATTN;
                constexpr size_t wlds4 = rw_lds_bytes<SEMIDETR_RW_NT, SEMIDETR_RW_RTH, 16, -1, SEMIDETR_RW_HC, 4, SEMIDETR_RW_TUNE>(), wlds5 = rw_lds_bytes<SEMIDETR_RW_NT5, SEMIDETR_RW_RTH5, 16, -1, 4, 5, kTune5>();
                static_assert(wlds4 <= 160 * 1024 && wlds5 <= 160 * 1024, "region-window configuration does not fit the LDS");
                if constexpr (std::is_same<IO, RawIO>::value) {
                    if (io.has_mask()) {
                        if (L == 4) {
                            constexpr size_t wlds4m = rw_lds_bytes<SEMIDETR_RW_NT, SEMIDETR_RW_RTH, 16, -1, SEMIDETR_RW_HC, 4, SEMIDETR_RW_TUNE_MASK>();
                            static_assert(wlds4m <= 160 * 1024, "region-window configuration does not fit the LDS");
                            return launch_window(&msda_rw_d32<IO, SEMIDETR_RW_NT, SEMIDETR_RW_RTH, 16, -1, SEMIDETR_RW_HC, 4, false, SEMIDETR_RW_DBG, SEMIDETR_RW_TUNE_MASK, true>,
                                                 &msda_rw_d32<IO, SEMIDETR_RW_NT, SEMIDETR_RW_RTH, 16, -1, SEMIDETR_RW_HC, 4, false, SEMIDETR_RW_DBG, SEMIDETR_RW_TUNE_MASK + 102400, true>,
                                                 wlds4m, SEMIDETR_RW_NT, SEMIDETR_RW_RTH * 16);
                        }
                        constexpr size_t wlds5m = rw_lds_bytes<SEMIDETR_RW_NT5, SEMIDETR_RW_RTH5, 16, -1, 4, 5, SEMIDETR_RW_TUNE5_MASK>();
                        return launch_window(&msda_rw_d32<IO, SEMIDETR_RW_NT5, SEMIDETR_RW_RTH5, 16, -1, 4, 5, false, 0, SEMIDETR_RW_TUNE5_MASK, true>, nullptr, wlds5m,
                                             SEMIDETR_RW_NT5, SEMIDETR_RW_RTH5 * 16);
                    }
                }
                if (L == 4)
                    return launch_window(&msda_rw_d32<IO, SEMIDETR_RW_NT, SEMIDETR_RW_RTH, 16, -1, SEMIDETR_RW_HC, 4, false, SEMIDETR_RW_DBG, kTune4>,
                                         &msda_rw_d32<IO, SEMIDETR_RW_NT, SEMIDETR_RW_RTH, 16, -1, SEMIDETR_RW_HC, 4, false, SEMIDETR_RW_DBG, SEMIDETR_RW_TUNE + 102400>, wlds4,
                                         SEMIDETR_RW_NT, SEMIDETR_RW_RTH * 16);
                return launch_window(&msda_rw_d32<IO, SEMIDETR_RW_NT5, SEMIDETR_RW_RTH5, 16, -1, 4, 5, false, 0, kTune5>, nullptr, wlds5, SEMIDETR_RW_NT5,
                                     SEMIDETR_RW_RTH5 * 16);
            };
            const int wrc = pick_window();
            if (!fell_back) return wrc;
            (void)hipGetLastError();
            fs = FwdStats{nullptr, 0u};      // (this launch's counter parity was the window kernel's: count nothing)
        }
        // 4 x 8 query patches.  Grid sizing hint: about the number of 32-pixel patches of a usual pyramid (ragged edges
        // included); a workgroup takes patches slot, slot + hint, ... so any hint >= 1 is correct
        const int bound = (S + 31) / 32 * 5 / 4 + 4 * L;
        SEMIDETR_REQUIRE((int64_t)N * bound * M < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_forward: grid too large");
        hipLaunchKernelGGL((msda_fwd_d32<1, 4, 408, IO>), dim3((unsigned)((int64_t)N * bound * M)), dim3(256),
                           (size_t)32 * (L * P + 1) * 32, st, value, spatial_shapes, level_start, io, S, M, L, Lq, P, bound, out, fs);
        g_last_kernels = "msda_fwd_d32<1, 4, 408";
        return semidetr::launch_status("msda_fwd_d32<patch>");
    }
    int split = pick_split(0, N, Lq, M);
    // the records of a workgroup's rows have to fit the 64 KB every kernel may use without asking (L * P up to 256 is on this
    // path: 32 rows x 257 records x 32 bytes would be 263 KB): fewer rows per workgroup for the very wide cases
    while (split < 4 && (size_t)(32 / split) * (L * P + 1) * 32 > 64 * 1024) split *= 2;
    const int rpb = 32 / split;
    const int tiles = (Lq + rpb - 1) / rpb;
    SEMIDETR_REQUIRE((int64_t)N * tiles * M < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_forward: grid too large");
    const size_t lds = (size_t)rpb * (L * P + 1) * 32;
    if (int rc = allow_big_lds(&msda_fwd_d32<4, 4, 0, IO>, lds, "msda_forward")) return rc;      // only L * P > 255 at 8 rows
    if (split == 1) LAUNCH_FWD(1, 0, tiles, lds);
    else if (split == 2) LAUNCH_FWD(2, 0, tiles, lds);
    else LAUNCH_FWD(4, 0, tiles, lds);
#undef LAUNCH_FWD
    g_last_kernels = split == 1 ? "msda_fwd_d32<1, 4, 0" : (split == 2 ? "msda_fwd_d32<2, 4, 0" : "msda_fwd_d32<4, 4, 0");
    return semidetr::launch_status("msda_fwd_d32");
}

template <typename IO>
int launch_fast_backward(hipStream_t st, const float *grad_out, const float *value, const int64_t *spatial_shapes,
                         const int64_t *level_start, const IO &io, int N, int S, int M, int L, int Lq, int P,
                         int flags, float *grad_value)
{
    SEMIDETR_REQUIRE(!(flags & SEMIDETR_MSDA_QUERIES_ARE_PIXELS) || Lq == S, SEMIDETR_E_BADARG,
                     "msda_backward: SEMIDETR_MSDA_QUERIES_ARE_PIXELS needs num_query == spatial_size");
    const bool pixels = (flags & SEMIDETR_MSDA_QUERIES_ARE_PIXELS) != 0 && (size_t)32 * (L * P + 1) * 32 <= 63 * 1024;
    const size_t fill = sizeof(float) * (size_t)N * S * M * kD;
    // (S * M * 128 < 2^32: the region scatter addresses grad_value rows by 32-bit byte offsets from the image's first row)
    // (... and its sampling data inside the image's view with 32-bit indices: S * M * L * P * 8 < 2^32)
    // (M * 128 <= 0xffff: the region scatter's flush multiplies a 16-bit pixel key by the row pitch into 32 bits -- implied by heads_ok,
    //  stated here because the kernel's inline assembly depends on it)
    if (pixels && P == kPT && S < (1 << 23) && (uint64_t)S * M * kD * 4 < (1ull << 32) && (uint64_t)S * M * L * P * 8 < (1ull << 32) &&
        (int64_t)M * kD * 4 <= 0xffff) {
        // ---- encoder self-attention: patch gather (the two small gradients; it clears grad_value as a side job, the
        //      scatter that accumulates into it is the NEXT launch) + region-owned scatter (msda_region.h)
#ifndef SEMIDETR_SEPARATE_FILL
#define SEMIDETR_SEPARATE_FILL 0      // tuning builds: 1 = hipMemsetAsync before the gather instead of the gather's side job
#endif
        const bool fill_in_gather = !SEMIDETR_SEPARATE_FILL && (L * P == 16 || L * P == 20) && (reinterpret_cast<uintptr_t>(grad_value) & 15) == 0;
        if (!fill_in_gather) {
            const hipError_t e = hipMemsetAsync(grad_value, 0, fill, st);
            if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward memset: %s", hipGetErrorString(e));
        }
        const size_t glds = (size_t)32 * (L * P + 1) * 32 + 2 * kMaxLevels * sizeof(float);
        const int gbound = (S + 31) / 32 * 5 / 4 + 4 * L;      // patch grid hint, see launch_fast_forward
        const int gt = (Lq + 31) / 32;
        SEMIDETR_REQUIRE((int64_t)N * std::max(gbound, gt) * M < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
        float4 *zero = fill_in_gather ? reinterpret_cast<float4 *>(grad_value) : nullptr;
        bool window_gather = false;
        // (SEMIDETR_MSDA_FIXED_FORWARD: which gather runs must not depend on earlier launches either -- the patch gather)
        // (Lq * M * L * P * 8 bytes per image < 2^32: msda_gw_d32 indexes the sampling data inside an image with 32 bits)
        // (SEMIDETR_MSDA_GATHER_WINDOW / _PATCH: the caller's record of what the slot said when the matching forward ran; neither: its state now)
        const bool near = (flags & SEMIDETR_MSDA_GATHER_WINDOW) ? true : ((flags & SEMIDETR_MSDA_GATHER_PATCH) ? false : slot_samples_are_near((flags >> 8) & 0xff));
        if ((L == 4 || L == 5) && P == kPT && (fill_in_gather || SEMIDETR_SEPARATE_FILL) && !(flags & SEMIDETR_MSDA_FIXED_FORWARD) && near &&
            (uint64_t)Lq * M * L * P * 8 < (1ull << 32)) {
            // lane-per-sample gather on region windows (msda_gw.h): 16 x 16 regions, margin 4 on every level, one 1024-thread workgroup per CU
            // (round 5: 1024 threads = 16 waves per CU for the reference contract and the fused prologue without a mask -- 124 / 128 VGPRs
            //  once the region grid's division reciprocals and the float copies of the level sizes are rebuilt per region: 232 -> 217 us
            //  in the probe; the masked instantiation too since the thread index is rebuilt from the wave number where it is needed)
            // Five levels (round 6, the COCO-Full pyramid): 20 lanes per (query, head) row, three rows per wave; the windows of five levels
            // at margin 4 fit the LDS with regions of up to 13 x 16 pixels (SEMIDETR_GW_RTH5).
            auto launch_gw = [&](auto kern, auto nt_c, size_t wl, int region_px) -> bool {
                constexpr int kGwNT = decltype(nt_c)::value;
                if (allow_big_lds(kern, wl, "msda_backward") != SEMIDETR_OK) {      // refused: the patch gather below
                    (void)hipGetLastError();
                    return false;
                }
                const int wbound = ((S * 3 / 4 + region_px - 1) / region_px) * 9 / 8 + 2 * L;
                if ((int64_t)N * wbound * M >= INT32_MAX) return false;
                hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)N * wbound * M)), dim3(kGwNT), wl, st, grad_out, value, spatial_shapes,
                                   level_start, io, S, M, wbound, zero, (int64_t)(fill / 16));
                return true;
            };
            constexpr int kNT5 = std::is_same<IO, RawIO>::value ? SEMIDETR_GW_NT5_RAW : SEMIDETR_GW_NT5;
            constexpr size_t wl4 = gw_lds_bytes<SEMIDETR_GW_NT, SEMIDETR_GW_RTH, SEMIDETR_GW_RTW, SEMIDETR_GW_H0, SEMIDETR_GW_HC, 4>();
            constexpr size_t wl5 = gw_lds_bytes<kNT5, SEMIDETR_GW_RTH5, SEMIDETR_GW_RTW, SEMIDETR_GW_H0, SEMIDETR_GW_HC, 5>();
            constexpr int px4 = SEMIDETR_GW_RTH * SEMIDETR_GW_RTW, px5 = SEMIDETR_GW_RTH5 * SEMIDETR_GW_RTW;
            const std::integral_constant<int, SEMIDETR_GW_NT> nt4;
            const std::integral_constant<int, kNT5> nt5;
            if constexpr (std::is_same<IO, RawIO>::value) {
                if (io.has_mask())
                    window_gather = L == 4
                        ? launch_gw(&msda_gw_d32<IO, SEMIDETR_GW_NT, SEMIDETR_GW_RTH, SEMIDETR_GW_RTW, SEMIDETR_GW_H0, SEMIDETR_GW_HC, 4, true, SEMIDETR_GW_DBG>, nt4, wl4, px4)
                        : launch_gw(&msda_gw_d32<IO, kNT5, SEMIDETR_GW_RTH5, SEMIDETR_GW_RTW, SEMIDETR_GW_H0, SEMIDETR_GW_HC, 5, true, SEMIDETR_GW_DBG>, nt5, wl5, px5);
            }
            if (!window_gather && !io.has_mask())
                window_gather = L == 4
                    ? launch_gw(&msda_gw_d32<IO, SEMIDETR_GW_NT, SEMIDETR_GW_RTH, SEMIDETR_GW_RTW, SEMIDETR_GW_H0, SEMIDETR_GW_HC, 4, false, SEMIDETR_GW_DBG>, nt4, wl4, px4)
                    : launch_gw(&msda_gw_d32<IO, kNT5, SEMIDETR_GW_RTH5, SEMIDETR_GW_RTW, SEMIDETR_GW_H0, SEMIDETR_GW_HC, 5, false, SEMIDETR_GW_DBG>, nt5, wl5, px5);
        }
        if (window_gather) {
        } else if (L * P == 16)             // DINO: sample loop unrolled, results in registers
            hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408, SEMIDETR_GATHER_WPE, SEMIDETR_GATHER_KB>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound, zero, (int64_t)(fill / 16));
        else if (L * P == 20)        // five levels (COCO-Full recipe)
        {
            // (the fused prologue's instantiation holds its softmax / location arithmetic besides: with 16 corner loads in flight it
            //  spills at four waves per SIMD since it also carries the padding mask's summary -- 8 in flight: 98 registers)
            constexpr int kb5 = std::is_same<IO, RawIO>::value ? 2 : SEMIDETR_GATHER5_KB;
            hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 20, 408, SEMIDETR_GATHER5_WPE, kb5>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound, zero, (int64_t)(fill / 16));
        }
        else
            hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 0>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt);
        if (int rc = semidetr::launch_status(window_gather ? "msda_gw_d32" : "msda_bwd_gather_d32")) return rc;
#if SEMIDETR_SCATTER_SW
        {
            auto kern = &msda_sw_d32<IO, SEMIDETR_SW_NT, SEMIDETR_SW_Q, SEMIDETR_SW_RTH, SEMIDETR_SW_RTW, SEMIDETR_SW_WH, SEMIDETR_SW_WW, SEMIDETR_SW_WPE>;
            constexpr size_t rlds = sw_lds_bytes<SEMIDETR_SW_NT, SEMIDETR_SW_Q, SEMIDETR_SW_WH, SEMIDETR_SW_WW>();
            if (int rc = allow_big_lds(kern, rlds, "msda_backward")) return rc;
            const int rbound = (S * 3 / 4 + SEMIDETR_SW_RTH * SEMIDETR_SW_RTW - 1) / (SEMIDETR_SW_RTH * SEMIDETR_SW_RTW) * 9 / 8 + 4 * L;
            const int64_t rgrid = (int64_t)N * rbound * M;
            SEMIDETR_REQUIRE(rgrid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
            hipLaunchKernelGGL(kern, dim3((unsigned)rgrid), dim3(SEMIDETR_SW_NT), rlds, st, grad_out, spatial_shapes, level_start, io, S, M, L,
                               rbound, grad_value);
            g_last_kernels = window_gather ? "msda_gw_d32+msda_sw_d32" : "msda_bwd_gather_d32+msda_sw_d32";
            return semidetr::launch_status("msda_sw_d32");
        }
#endif
        // (two workgroups per CU are four waves per SIMD: the fused prologue's instantiation takes that register budget -- at six it
        //  spills two registers; the reference contract's measured 1.5 % faster with the tighter one)
        constexpr int kScatterWpe = std::is_same<IO, RawIO>::value ? 4 : SEMIDETR_SCATTER_WPE;
        auto kern = &msda_bwd_scatter_d32_reg<IO, SEMIDETR_SCATTER_NT, SEMIDETR_SCATTER_Q, SEMIDETR_SCATTER_RTH, SEMIDETR_SCATTER_RTW,
                                              SEMIDETR_SCATTER_WH, SEMIDETR_SCATTER_WW, 0, kScatterWpe, SEMIDETR_SCATTER_WU>;
        const size_t rlds = reg_lds_bytes<SEMIDETR_SCATTER_NT, SEMIDETR_SCATTER_Q, SEMIDETR_SCATTER_WH, SEMIDETR_SCATTER_WW>();
        if (int rc = allow_big_lds(kern, rlds, "msda_backward")) return rc;
        constexpr int kRegPix = SEMIDETR_SCATTER_RTH * SEMIDETR_SCATTER_RTW;
        // grid sizing hint as for the window kernels: the finest level of a DETR pyramid holds ~3/4 of the pixels (any bound >= 1 is correct:
        // a workgroup takes regions slot, slot + bound, ...).  The old hint (S / region * 5 / 4 + 4 L = 161 for the 91 regions of the 800 x
        // 1333 pyramid) launched 2240 workgroups per bs-4 launch that found no region -- each holds one of a CU's two slots until it has read
        // the level table.
        const int rbound = SEMIDETR_SCATTER_TIGHT ? ((S * 3 / 4 + kRegPix - 1) / kRegPix) * 9 / 8 + 2 * L : (S + kRegPix - 1) / kRegPix * 5 / 4 + 4 * L;
        const int64_t rgrid = (int64_t)N * rbound * M;
        SEMIDETR_REQUIRE(rgrid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
        hipLaunchKernelGGL(kern, dim3((unsigned)rgrid), dim3(SEMIDETR_SCATTER_NT), rlds, st, grad_out, spatial_shapes, level_start, io, S, M, L,
                           rbound, grad_value);
        g_last_kernels = window_gather ? "msda_gw_d32+msda_bwd_scatter_d32_reg"
                         : (fill_in_gather ? "msda_bwd_gather_d32+msda_bwd_scatter_d32_reg"
                                           : "fillBufferAligned+msda_bwd_gather_d32+msda_bwd_scatter_d32_reg");
        return semidetr::launch_status("msda_bwd_scatter_d32_reg");
    }
    // ---- any query set
    hipError_t e = hipMemsetAsync(grad_value, 0, fill, st);
    if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward memset: %s", hipGetErrorString(e));
    // (four gather blocks of 2 * 32 * (L * P + 1) records share a merged workgroup's LDS: beyond L * P = 36 the launch takes
    //  the strips kernel below)
    const size_t merged_gather_lds = (size_t)(kLvlThreadsWide / 256) * ((size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4) * 16;
    // (S < 2^20: lvl_scatter_body packs a level-local row index into 20 bits of its entry word -- ADVICE r03; larger maps take
    //  the strips kernel below)
    if ((int64_t)N * Lq >= 512 && P <= 8 && merged_gather_lds <= 159 * 1024 && S < (1 << 20)) {
        // level-aggregated scatter workgroups + gather workgroups side by side in ONE launch (msda_bwd_lvl_merged_wide).
        // bucketed levels: as many queries per workgroup as its LDS takes (fewest flushed rows); levels too large to bucket:
        // <= 192 queries per workgroup (parallelism), and enough workgroups for small launches
        const int qcap = P <= 4 ? kLvlQWide : kLvlQWide / 2;             // the entry list is chunk * P * 4 * 8 bytes
        const int want = (128 + N * L * M - 1) / (N * L * M);            // small launches: >= ~128 scatter workgroups
        const int fine = std::min(want, (Lq + 63) / 64);
        // ONE chunk for the bucketed levels whenever the queries fit (Lq <= 384, the two-stage Deformable-DETR / BASELINE shape):
        // the workgroup then owns the level's rows of its (image, head) and STORES their sums instead of adding them atomically
        // (lvl_scatter_body, "exclusive") -- micro-benchmark backward 36.3 -> 30.7 us, bs 4 / Lq 300 63 -> 53 us
        const int chunks_b = Lq <= qcap ? 1 : std::max((Lq + qcap - 1) / qcap, fine), chunk_q_b = (Lq + chunks_b - 1) / chunks_b;
#ifndef SEMIDETR_LVL_CHUNKQ
#define SEMIDETR_LVL_CHUNKQ 224      // queries per scatter workgroup of the levels too large to bucket.  Round 5 (tools/r05_ab_step.sh): the merged
                                     // launch takes one 1024-thread workgroup per CU, so what counts is how its workgroups fill waves of 256 --
                                     // Lq 1100 at bs 4: 192 -> 6 chunks -> 768 + 280 = 1048 workgroups = 4.09 waves; 224 / 256 -> 5 chunks -> 920 =
                                     // 3.6: decoder backward 0.901 -> 0.864 ms per step (bs 4), 0.306 -> 0.255 (bs 1: 262 -> 230 workgroups, one
                                     // wave); 208 (6 chunks) 0.902 / 0.306, 320 (4 chunks) 0.932 / 0.269
#endif
        const int chunks = std::max(std::max(chunks_b, fine), (Lq + SEMIDETR_LVL_CHUNKQ - 1) / SEMIDETR_LVL_CHUNKQ);
        const int chunk_q = (Lq + chunks - 1) / chunks;
        const int gt = (Lq + 31) / 32;                                   // gather: 32 query rows per 256-thread block
        const int64_t gblocks = (int64_t)N * gt * M, sblocks = (int64_t)N * chunks * L * M;
        const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
        constexpr int kParts = kLvlThreadsWide / 256;                    // gather blocks per workgroup
        const size_t slds = std::max((size_t)kLvlQWide * kD * 4 + ((size_t)chunk_q_b * P * 4 + 8) * 8 + (size_t)2 * kLvlRows * 4,
                                     kParts * half_f4 * 16);
        const int64_t grid = sblocks + (gblocks + kParts - 1) / kParts;
        SEMIDETR_REQUIRE(grid < INT32_MAX && slds <= 159 * 1024, SEMIDETR_E_TOOLARGE, "msda_backward: merged launch too large");
#define LAUNCH_MERGED(KLP_)                                                                                          \
        do {                                                                                                             \
            auto kern = &msda_bwd_lvl_merged_wide<IO, KLP_>;                                                             \
            if (int rc = allow_big_lds(kern, slds, "msda_backward")) return rc;                                          \
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kLvlThreadsWide), slds, st, grad_out, value, spatial_shapes, \
                               level_start, io, S, M, L, Lq, P, chunks, chunk_q, chunks_b, chunk_q_b, (int)sblocks, gt, (int)gblocks,      \
                               grad_value);                                                                              \
        } while (0)
        if (L * P == 16) LAUNCH_MERGED(16);
        else LAUNCH_MERGED(0);
#undef LAUNCH_MERGED
        g_last_kernels = "fillBufferAligned+msda_bwd_lvl_merged_wide";
        return semidetr::launch_status("msda_bwd_lvl_merged_wide");
    }
    // small launches: one fused kernel after the fill; 32 query rows per workgroup, 8 when that would not fill 256 CUs
    // (and 8 when 32 rows of records would not fit 64 KB of LDS: L * P > 62)
    const int rpb = ((int64_t)N * M * ((Lq + 31) / 32) >= 1024 && (size_t)32 * (L * P + 1) * 32 + 2 * kMaxLevels * sizeof(float) <= 64 * 1024) ? 32 : 8;
    const int tiles = (Lq + rpb - 1) / rpb;
    const int64_t grid = (int64_t)N * tiles * M;
    SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
    const size_t lds = (size_t)rpb * (L * P + 1) * 32 + 2 * kMaxLevels * sizeof(float);
    if (int rc = allow_big_lds(&msda_bwd_d32<8, IO>, lds, "msda_backward")) return rc;      // only L * P > 254 at 8 rows
    if (rpb == 32)
        hipLaunchKernelGGL((msda_bwd_d32<32, IO>), dim3((unsigned)grid), dim3(256), lds, st, grad_out, value, spatial_shapes,
                           level_start, io, S, M, L, Lq, P, tiles, grad_value);
    else
        hipLaunchKernelGGL((msda_bwd_d32<8, IO>), dim3((unsigned)grid), dim3(256), lds, st, grad_out, value, spatial_shapes,
                           level_start, io, S, M, L, Lq, P, tiles, grad_value);
    g_last_kernels = rpb == 32 ? "fillBufferAligned+msda_bwd_d32<32" : "fillBufferAligned+msda_bwd_d32<8";
    return semidetr::launch_status("msda_bwd_d32");
}

#if SEMIDETR_EXPERIMENTS
#include "msda_experiments.h"
#endif

}  // namespace

extern "C" const char *semidetr_msda_last_kernels(void) { return g_last_kernels; }

extern "C" int semidetr_msda_set_forward_policy(int policy)
{
    SEMIDETR_REQUIRE(policy >= 0 && policy <= 2, SEMIDETR_E_BADARG, "msda_set_forward_policy: 0 adaptive, 1 patch kernel, 2 window kernel");
    g_fwd_policy.store(policy, std::memory_order_relaxed);
    return SEMIDETR_OK;
}

extern "C" int semidetr_msda_forward_policy_state_slot(int slot, int *policy, int *mode, float *far_fraction, unsigned *updates)
{
    SEMIDETR_REQUIRE(slot >= 0 && slot < kPolicySlots, SEMIDETR_E_BADARG, "msda_forward_policy_state: slot must be 0..%d", kPolicySlots - 1);
    int dev = 0;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return semidetr::fail((int)e, "msda_forward_policy_state: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(g_adapt_mu);
    const FwdAdapt *a = g_adapt[dev >= 0 && dev < kMaxDevices ? dev : 0];
    const FwdSlot none;
    const FwdSlot &sl = a ? a->slot[slot] : none;
    if (policy) *policy = g_fwd_policy.load(std::memory_order_relaxed);
    if (mode) *mode = sl.mode;
    if (far_fraction) *far_fraction = sl.last_frac;
    if (updates) *updates = sl.updates;
    return SEMIDETR_OK;
}

extern "C" int semidetr_msda_gather_choice(int slot)
{
    return (slot >= 0 && slot < kPolicySlots && slot_samples_are_near(slot)) ? SEMIDETR_MSDA_GATHER_WINDOW : SEMIDETR_MSDA_GATHER_PATCH;
}

extern "C" int semidetr_msda_forward_policy_state(int *policy, int *mode, float *far_fraction, unsigned *updates)
{
    return semidetr_msda_forward_policy_state_slot(0, policy, mode, far_fraction, updates);
}

#if SEMIDETR_EXPERIMENTS
// tuning aid: per-phase cycle counters of the instrumented kernels; reset = 1 zeroes them
extern "C" int semidetr_debug_counters(unsigned long long *out16, int reset)
{
    hipError_t e = hipSuccess;
    if (out16) e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dest_dbg), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) {
        const unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_dest_dbg), z, sizeof(z));
    }
    return e == hipSuccess ? SEMIDETR_OK : semidetr::fail((int)e, "debug_counters: %s", hipGetErrorString(e));
}

extern "C" void semidetr_msda_set_variant(int fwd_variant, int bwd_variant)
{
    g_fwd_variant_a.store(fwd_variant, std::memory_order_relaxed);
    g_bwd_variant_a.store(bwd_variant, std::memory_order_relaxed);
}
#define SEMIDETR_FWD_VARIANT (g_fwd_variant_a.load(std::memory_order_relaxed))
#define SEMIDETR_BWD_VARIANT (g_bwd_variant_a.load(std::memory_order_relaxed))
#else
#define SEMIDETR_FWD_VARIANT 0
#define SEMIDETR_BWD_VARIANT 0
#endif

// fast path of the f32 entry points: the product dispatch, or (experiments library, a variant forced) the tuning dispatch
template <typename IO>
static int dispatch_fast_forward(hipStream_t st, const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                 const IO &io, int N, int S, int M, int L, int Lq, int P, int flags, float *out)
{
#if SEMIDETR_EXPERIMENTS
    if (SEMIDETR_FWD_VARIANT != 0) return exp_launch_fast_forward(st, value, spatial_shapes, level_start, io, N, S, M, L, Lq, P, flags, out);
#endif
    return launch_fast_forward(st, value, spatial_shapes, level_start, io, N, S, M, L, Lq, P, flags, out);
}
template <typename IO>
static int dispatch_fast_backward(hipStream_t st, const float *grad_out, const float *value, const int64_t *spatial_shapes,
                                  const int64_t *level_start, const IO &io, int N, int S, int M, int L, int Lq, int P, int flags,
                                  float *grad_value)
{
#if SEMIDETR_EXPERIMENTS
    if (SEMIDETR_BWD_VARIANT != 0)
        return exp_launch_fast_backward(st, grad_out, value, spatial_shapes, level_start, io, N, S, M, L, Lq, P, flags, grad_value);
#endif
    return launch_fast_backward(st, grad_out, value, spatial_shapes, level_start, io, N, S, M, L, Lq, P, flags, grad_value);
}

extern "C" int semidetr_msda_forward_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                                         const int64_t *level_start, const float *sampling_loc,
                                         const float *attn_weight, int batch, int spatial_size,
                                         int num_heads, int channels, int num_levels, int num_query,
                                         int num_point, int flags, float *out)
{
    const int N = batch, S = spatial_size, M = num_heads, D = channels, L = num_levels, Lq = num_query,
              P = num_point;
    if (SEMIDETR_FWD_VARIANT == 99 || !fast_ok(value, sampling_loc, out, D, L, P) || !slice_ok(S, M) || !heads_ok(M))
        return forward_impl<float>(stream, value, spatial_shapes, level_start, sampling_loc, attn_weight, N,
                                   S, M, D, L, Lq, P, out);
    if (int rc = check_common(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L,
                              Lq, P, 4))
        return rc;
    SEMIDETR_REQUIRE(out, SEMIDETR_E_BADARG, "msda_forward: null output");
    const LocAttnIO io = {sampling_loc, attn_weight, nullptr, nullptr};
    return dispatch_fast_forward(semidetr::as_stream(stream), value, spatial_shapes, level_start, io, N, S, M, L, Lq,
                                 P, flags, out);
}

extern "C" int semidetr_msda_backward_f32(void *stream, const float *grad_out, const float *value,
                                          const int64_t *spatial_shapes, const int64_t *level_start,
                                          const float *sampling_loc, const float *attn_weight, int batch,
                                          int spatial_size, int num_heads, int channels, int num_levels,
                                          int num_query, int num_point, int flags, float *grad_value,
                                          float *grad_sampling_loc, float *grad_attn_weight)
{
    const int N = batch, S = spatial_size, M = num_heads, D = channels, L = num_levels, Lq = num_query,
              P = num_point;
    if (SEMIDETR_BWD_VARIANT == 99 || !fast_ok(value, sampling_loc, grad_out, D, L, P) || !slice_ok(S, M) || !heads_ok(M) ||
        !fast_ok(grad_value, grad_sampling_loc, grad_attn_weight, D, L, P))
        return backward_impl<float>(stream, grad_out, value, spatial_shapes, level_start, sampling_loc,
                                    attn_weight, N, S, M, D, L, Lq, P, grad_value, grad_sampling_loc,
                                    grad_attn_weight);
    if (int rc = check_common(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L,
                              Lq, P, 4))
        return rc;
    SEMIDETR_REQUIRE(grad_out && grad_value && grad_sampling_loc && grad_attn_weight, SEMIDETR_E_BADARG,
                     "msda_backward: null pointer argument");
    const LocAttnIO io = {sampling_loc, attn_weight, grad_sampling_loc, grad_attn_weight};
    return dispatch_fast_backward(semidetr::as_stream(stream), grad_out, value, spatial_shapes, level_start, io, N, S,
                                  M, L, Lq, P, flags, grad_value);
}

// ---- fused MSDeformAttn prologue / epilogue (fp32, channels == 32) --------------------------------------
static int check_fused(const void *value, const void *shapes, const void *starts, const void *ref, int ref_dim,
                       const void *off, const void *logit, int N, int S, int M, int D, int L, int Lq, int P)
{
    if (int rc = check_common(value, shapes, starts, off, logit, N, S, M, D, L, Lq, P, 4)) return rc;
    SEMIDETR_REQUIRE(ref, SEMIDETR_E_BADARG, "msda_fused: null reference_points");
    SEMIDETR_REQUIRE(ref_dim == 2 || ref_dim == 4, SEMIDETR_E_BADARG,
                     "Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
    SEMIDETR_REQUIRE(D == kD && L <= kMaxLevels && (int64_t)L * P <= 256 && slice_ok(S, M) && heads_ok(M), SEMIDETR_E_BADARG,
                     "msda_fused: only channels == 32 (got %d), <= %d levels, L*P <= 256, <= 32 heads, image slice < 4 GB", D, kMaxLevels);
    SEMIDETR_REQUIRE((((uintptr_t)value | (uintptr_t)off | (uintptr_t)ref) & 15) == 0, SEMIDETR_E_BADARG,
                     "msda_fused: value / sampling_offsets / reference_points must be 16-byte aligned");
    SEMIDETR_REQUIRE((int64_t)N * Lq * L * ref_dim * 4 < (int64_t)0xFFFFFFF0u, SEMIDETR_E_TOOLARGE,
                     "msda_fused: reference_points must be smaller than 4 GB");
    return SEMIDETR_OK;
}

extern "C" int semidetr_msda_fused_forward_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                                               const int64_t *level_start, const float *reference_points,
                                               int ref_dim, const float *sampling_offsets,
                                               const float *attn_logits, const unsigned char *padding_mask,
                                               const int *mask_extents, int batch,
                                               int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                               int num_point, int flags, float *out)
{
    if (int rc = check_fused(value, spatial_shapes, level_start, reference_points, ref_dim, sampling_offsets,
                             attn_logits, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point))
        return rc;
    SEMIDETR_REQUIRE(out && ((uintptr_t)out & 15) == 0, SEMIDETR_E_BADARG, "msda_fused_forward: bad output pointer");
    const unsigned ref_bytes = (unsigned)((int64_t)batch * num_query * num_levels * ref_dim * 4);
    const RawIO io = {reference_points, sampling_offsets, attn_logits, nullptr, nullptr, ref_dim, num_heads,
                      num_levels, padding_mask, spatial_size, ref_bytes, padding_mask ? mask_extents : nullptr};
    SEMIDETR_REQUIRE(!padding_mask || SEMIDETR_FWD_VARIANT == 0, SEMIDETR_E_BADARG,
                     "msda_fused_forward: the experimental kernel variants do not take a padding mask");
    return dispatch_fast_forward(semidetr::as_stream(stream), value, spatial_shapes, level_start, io, batch,
                                 spatial_size, num_heads, num_levels, num_query, num_point, flags, out);
}

extern "C" int semidetr_msda_fused_backward_f32(void *stream, const float *grad_out, const float *value,
                                                const int64_t *spatial_shapes, const int64_t *level_start,
                                                const float *reference_points, int ref_dim,
                                                const float *sampling_offsets, const float *attn_logits,
                                                const unsigned char *padding_mask, const int *mask_extents, int batch,
                                                int spatial_size, int num_heads, int channels, int num_levels,
                                                int num_query, int num_point, int flags, float *grad_value,
                                                float *grad_sampling_offsets, float *grad_attn_logits)
{
    if (int rc = check_fused(value, spatial_shapes, level_start, reference_points, ref_dim, sampling_offsets,
                             attn_logits, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point))
        return rc;
    SEMIDETR_REQUIRE(grad_out && grad_value && grad_sampling_offsets && grad_attn_logits, SEMIDETR_E_BADARG,
                     "msda_fused_backward: null pointer argument");
    SEMIDETR_REQUIRE((((uintptr_t)grad_out | (uintptr_t)grad_value | (uintptr_t)grad_sampling_offsets) & 15) == 0,
                     SEMIDETR_E_BADARG, "msda_fused_backward: pointers must be 16-byte aligned");
    const unsigned ref_bytes = (unsigned)((int64_t)batch * num_query * num_levels * ref_dim * 4);
    const RawIO io = {reference_points, sampling_offsets, attn_logits, grad_sampling_offsets, grad_attn_logits,
                      ref_dim, num_heads, num_levels, padding_mask, spatial_size, ref_bytes, padding_mask ? mask_extents : nullptr};
    SEMIDETR_REQUIRE(!padding_mask || SEMIDETR_BWD_VARIANT == 0, SEMIDETR_E_BADARG,
                     "msda_fused_backward: the experimental kernel variants do not take a padding mask");
    return dispatch_fast_backward(semidetr::as_stream(stream), grad_out, value, spatial_shapes, level_start, io, batch,
                                  spatial_size, num_heads, num_levels, num_query, num_point, flags, grad_value);
}

// ---- padding mask -> one word per (image, level): vh | vw << 16 or -1 (MaskExt, msda_fast.h).  One workgroup per (image, level): the first
//      padded pixel of row 0 / column 0 gives the candidate (vw, vh); every pixel is then checked against "padding iff y >= vh or
//      x >= vw" -- exactly what F.interpolate of an image-sized padding band produces (dense_heads/dino_detr_head.py:305-318).
// Round 6: 1024 threads, 16 mask bytes per thread and trip (one division per trip instead of one per pixel; the level's 16.7 k bytes of the
// 100 x 167 level are ONE trip): 24 -> ~4 us per launch.  It runs once per mask tensor -- in training: per batch, i.e. a few times per step.
__global__ __launch_bounds__(1024) void msda_mask_extents_kernel(const unsigned char *__restrict__ mask, const int64_t *__restrict__ shapes,
                                                                 const int64_t *__restrict__ starts, int S, int L, int *__restrict__ ext)
{
    const int n = (int)blockIdx.x / L, l = (int)blockIdx.x % L;
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)starts[l];
    const unsigned char *mk = mask + (int64_t)n * S + st;
    __shared__ int s_vh, s_vw, s_bad;
    if (threadIdx.x == 0) { s_vh = H; s_vw = W; s_bad = 0; }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 1024)
        if (mk[x]) atomicMin(&s_vw, x);
    for (int y = threadIdx.x; y < H; y += 1024)
        if (mk[(int64_t)y * W]) atomicMin(&s_vh, y);
    __syncthreads();
    int vh = s_vh, vw = s_vw;
    if (vh == 0 || vw == 0) vh = vw = 0;      // pixel (0, 0) is padding: only "everything is" has the form
    bool bad = false;
    const int HW = H * W;
    // 16 consecutive pixels per thread: aligned 16-byte loads where the level's first byte allows it (it need not be aligned: st is any sum of H*W)
    const int head = (int)((16 - ((uintptr_t)mk & 15)) & 15);       // bytes before the first aligned 16
    for (int i = threadIdx.x; i < min(head, HW); i += 1024) {
        const int y = i / W, x = i - y * W;
        bad = bad || ((mk[i] != 0) != (y >= vh || x >= vw));
    }
    for (int i0 = head + 16 * (int)threadIdx.x; i0 < HW; i0 += 16 * 1024) {
        int y = i0 / W, x = i0 - y * W;
        if (i0 + 16 <= HW) {
            const uint4 q = *reinterpret_cast<const uint4 *>(mk + i0);
            const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const bool m = ((w4[b >> 2] >> (8 * (b & 3))) & 0xffu) != 0;
                bad = bad || (m != (y >= vh || x >= vw));
                if (++x == W) { x = 0; ++y; }
            }
        } else {
            for (int i = i0; i < HW; ++i) {
                bad = bad || ((mk[i] != 0) != (y >= vh || x >= vw));
                if (++x == W) { x = 0; ++y; }
            }
        }
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (threadIdx.x == 0) ext[blockIdx.x] = (s_bad || H > 0x7fff || W > 0x7fff) ? -1 : (vh | (vw << 16));
}

extern "C" int semidetr_msda_mask_extents(void *stream, const unsigned char *padding_mask, const int64_t *spatial_shapes,
                                          const int64_t *level_start, int batch, int spatial_size, int num_levels, int *extents)
{
    SEMIDETR_REQUIRE(padding_mask && spatial_shapes && level_start && extents, SEMIDETR_E_BADARG, "msda_mask_extents: null pointer argument");
    SEMIDETR_REQUIRE(batch > 0 && spatial_size > 0 && num_levels > 0 && num_levels <= kMaxLevels && (int64_t)batch * num_levels < INT32_MAX,
                     SEMIDETR_E_BADARG, "msda_mask_extents: sizes must be positive (batch=%d spatial_size=%d num_levels=%d)", batch,
                     spatial_size, num_levels);
    hipLaunchKernelGGL(msda_mask_extents_kernel, dim3((unsigned)(batch * num_levels)), dim3(1024), 0, semidetr::as_stream(stream),
                       padding_mask, spatial_shapes, level_start, spatial_size, num_levels, extents);
    return semidetr::launch_status("msda_mask_extents");
}

extern "C" int semidetr_msda_forward_f64(void *stream, const double *value, const int64_t *spatial_shapes,
                                         const int64_t *level_start, const double *sampling_loc,
                                         const double *attn_weight, int batch, int spatial_size,
                                         int num_heads, int channels, int num_levels, int num_query,
                                         int num_point, double *out)
{
    return forward_impl<double>(stream, value, spatial_shapes, level_start, sampling_loc, attn_weight, batch,
                                spatial_size, num_heads, channels, num_levels, num_query, num_point, out);
}

extern "C" int semidetr_msda_backward_f64(void *stream, const double *grad_out, const double *value,
                                          const int64_t *spatial_shapes, const int64_t *level_start,
                                          const double *sampling_loc, const double *attn_weight, int batch,
                                          int spatial_size, int num_heads, int channels, int num_levels,
                                          int num_query, int num_point, double *grad_value,
                                          double *grad_sampling_loc, double *grad_attn_weight)
{
    return backward_impl<double>(stream, grad_out, value, spatial_shapes, level_start, sampling_loc,
                                 attn_weight, batch, spatial_size, num_heads, channels, num_levels, num_query,
                                 num_point, grad_value, grad_sampling_loc, grad_attn_weight);
}
