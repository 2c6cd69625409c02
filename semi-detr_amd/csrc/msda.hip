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
#include <cstdlib>

#include "common.h"

namespace {

constexpr int kMaxLevels = 32;

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
#include "msda_dest.h"   // destination-owned grad_value kernel for encoder self-attention
#include "msda_region.h" // region-owned windowed scatter for encoder self-attention
#include "msda_lw.h"     // LDS-window forward for encoder self-attention
#include "msda_rw.h"     // region-window forward / gather for encoder self-attention

int g_fwd_variant = 0, g_bwd_variant = 0;

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


// ---- region-window kernels (msda_rw.h): one launcher for the forward and the gather -------------------------------
// dynamic LDS above 64 KB has to be allowed per kernel AND per device
template <typename K>
int allow_big_lds(K kern, const char *what)
{
    static thread_local int done_dev = -1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess && dev != done_dev) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) done_dev = dev;
    }
    if (e != hipSuccess) return semidetr::fail((int)e, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    return SEMIDETR_OK;
}

template <typename IO, int NT, int RTH, int RTW, int H0, int HC, int KL, bool GATHER, int DBG = 0, int TUNE = 42>
int launch_rw(hipStream_t st, const float *grad_out, const float *value, const int64_t *spatial_shapes,
              const int64_t *level_start, const IO &io, int N, int S, int M, float *out, float4 *zero, int64_t zero_n4)
{
    auto kern = &msda_rw_d32<IO, NT, RTH, RTW, H0, HC, KL, GATHER, DBG, TUNE>;
    if (int rc = allow_big_lds(kern, "msda region-window kernel")) return rc;
    // grid sizing hint: the finest level of a DETR pyramid holds ~3/4 of the pixels; a workgroup takes regions slot,
    // slot + bound, ... so any bound >= 1 is correct (the level table lives in device memory)
    const int rpx = RTH * RTW;
    const int bound = ((S * 3 / 4 + rpx - 1) / rpx) * 9 / 8 + 2 * KL;
    const int64_t grid = (int64_t)N * bound * M;
    SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda: grid too large");
    constexpr size_t lds = rw_lds_bytes<NT, RTH, RTW, H0, HC, KL>();
    static_assert(lds <= 160 * 1024, "region-window configuration does not fit the LDS");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, st, grad_out, value, spatial_shapes, level_start, io, S, M,
                       bound, out, zero, zero_n4);
    return semidetr::launch_status(GATHER ? "msda_rw_d32<gather>" : "msda_rw_d32<forward>");
}

// variant code -> configuration {threads, region, margins}; 0 = the default configuration
template <typename IO, int KL, bool GATHER>
int launch_rw_cfg(int cfg, hipStream_t st, const float *grad_out, const float *value, const int64_t *spatial_shapes,
                  const int64_t *level_start, const IO &io, int N, int S, int M, float *out, float4 *zero, int64_t zero_n4)
{
#define RW(NT_, RH_, RW_, H0_, HC_) RWT(NT_, RH_, RW_, H0_, HC_, 0, 42)
#define RWT(NT_, RH_, RW_, H0_, HC_, DBG_, TUNE_) \
    launch_rw<IO, NT_, RH_, RW_, H0_, HC_, KL, GATHER, DBG_, TUNE_>(st, grad_out, value, spatial_shapes, level_start, io, N, S, M, out, zero, zero_n4)
    if constexpr (KL == 4) {
        switch (cfg) {
        case 0: return RWT(512, 8, 16, 4, 5, 0, 40);
        case 1: if constexpr (!GATHER) return RWT(256, 16, 16, -1, 4, 0, 40); else break;
        case 2: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 4, 0, 40); else break;
        case 3: if constexpr (!GATHER) return RWT(256, 8, 16, -1, 3, 0, 40); else break;
        case 4: if constexpr (!GATHER) return RWT(256, 8, 8, -1, 4, 0, 40); else break;
        case 5: if constexpr (!GATHER) return RWT(256, 8, 16, -1, 4, 0, 40); else break;      // level 0 through global loads
        case 6: if constexpr (!GATHER) return RWT(512, 8, 16, -1, 5, 0, 40); else break;
        case 7: if constexpr (!GATHER) return RWT(256, 8, 16, -1, 4, 1, 40); else return RWT(512, 8, 16, 4, 5, 1, 40);
        case 8: if constexpr (!GATHER) return RWT(256, 8, 16, -1, 4, 2, 40); else return RWT(512, 8, 16, 4, 5, 2, 40);
        case 9: if constexpr (!GATHER) return RWT(256, 8, 16, -1, 4, 3, 40); else return RWT(512, 8, 16, 4, 5, 3, 40);
        default: break;
        }
    }
    if constexpr (KL == 4) return RWT(512, 8, 16, 4, 5, 0, 40);
    else return RWT(512, 8, 16, 4, 4, 0, 40);      // five levels: the margin-5 windows do not fit 160 KB
#undef RWT
#undef RW
}

// ---- fast-path launchers, shared by the reference contract (LocAttnIO) and the fused prologue (RawIO) ----
template <typename IO>
int launch_fast_forward(hipStream_t st, const float *value, const int64_t *spatial_shapes,
                        const int64_t *level_start, const IO &io, int N, int S, int M, int L, int Lq, int P,
                        int flags, float *out)
{
    const bool pixels = (flags & SEMIDETR_MSDA_QUERIES_ARE_PIXELS) != 0;
    SEMIDETR_REQUIRE(!pixels || Lq == S, SEMIDETR_E_BADARG,
                     "msda_forward: SEMIDETR_MSDA_QUERIES_ARE_PIXELS needs num_query == spatial_size");
    const int split = pick_split(g_fwd_variant % 10, N, Lq, M);
    const int rpb = 32 / split;
    const int tiles = (Lq + rpb - 1) / rpb;
    const int64_t grid = (int64_t)N * tiles * M;
    SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_forward: grid too large");
    const size_t lds = (size_t)rpb * (L * P + 1) * 32;
#define LAUNCH_FWD(SP, UN, PT, TILES)                                                                       \
    hipLaunchKernelGGL((msda_fwd_d32<SP, UN, PT, IO>), dim3((unsigned)((int64_t)N * (TILES) * M)), dim3(256), \
                       lds, st, value, spatial_shapes, level_start, io, S, M, L, Lq, P, (TILES), out)
    if constexpr (!IO::kSoftmax) {
        if (g_fwd_variant == 600 && pixels && L * P == 16 && P == kPT) {
            // LDS-window forward (msda_lw.h): 8 x 8 query patches, three workgroups per CU
            const int bound = (S + 63) / 64 * 5 / 4 + 4 * L;
            SEMIDETR_REQUIRE((int64_t)N * bound * M < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_forward: grid too large");
            hipLaunchKernelGGL((msda_fwd_d32_lw<IO>), dim3((unsigned)(N * bound * M)), dim3(256), lw_lds_bytes(), st, value,
                               spatial_shapes, level_start, io, S, M, L, bound, out);
            g_last_kernels = "msda_fwd_d32_lw";
            return semidetr::launch_status("msda_fwd_d32_lw");
        }
    }
    if (g_fwd_variant >= 700 && g_fwd_variant <= 709) {
        SEMIDETR_REQUIRE(pixels && P == kPT && (L == 4 || L == 5), SEMIDETR_E_BADARG,
                         "msda_forward: the region-window kernel needs SEMIDETR_MSDA_QUERIES_ARE_PIXELS, num_point == 4, 4 or 5 levels");
        g_last_kernels = "msda_rw_d32<forward>";
        if (L == 4)
            return launch_rw_cfg<IO, 4, false>(g_fwd_variant - 700, st, nullptr, value, spatial_shapes, level_start, io, N, S, M, out, nullptr, 0);
        return launch_rw_cfg<IO, 5, false>(0, st, nullptr, value, spatial_shapes, level_start, io, N, S, M, out, nullptr, 0);
    }
    if (g_fwd_variant >= 500 && g_fwd_variant <= 505) {
        SEMIDETR_REQUIRE(pixels, SEMIDETR_E_BADARG, "msda_forward: the resident-level kernel needs SEMIDETR_MSDA_QUERIES_ARE_PIXELS");
        // 500: 8 patches per workgroup, coarse levels resident; 501: same schedule, nothing resident (control);
        // 502 / 503: 4 patches per workgroup resident / control; 504 / 505: 2 patches
        const int grp = g_fwd_variant <= 501 ? 8 : (g_fwd_variant <= 503 ? 4 : 2);
        const int res_max = (g_fwd_variant & 1) ? 0 : kResRows;
        const int G = ((S + 31) / 32 * 5 / 4 + 4 * L + grp - 1) / grp;      // grid sizing hint as for the patch kernel
        const size_t rlds = (size_t)(kResRows + 1) * 128 + (size_t)2 * 32 * (L * P + 1) * 32;
        SEMIDETR_REQUIRE(rlds <= 160 * 1024, SEMIDETR_E_BADARG, "msda_forward: too many samples per query for the resident-level kernel");
#define LAUNCH_RES(GRP)                                                                                              \
        do {                                                                                                             \
            static bool lds_ok = false; /* dynamic LDS above 64 KB has to be allowed once per kernel */                \
            if (!lds_ok) {                                                                                               \
                const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_d32_res<IO, GRP>),   \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      \
                if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_forward: hipFuncSetAttribute: %s", hipGetErrorString(ae)); \
                lds_ok = true;                                                                                           \
            }                                                                                                            \
            hipLaunchKernelGGL((msda_fwd_d32_res<IO, GRP>), dim3((unsigned)(N * M * G)), dim3(512), rlds, st, value,     \
                               spatial_shapes, level_start, io, S, M, L, P, G, res_max, out);                          \
        } while (0)
        if (grp == 8) LAUNCH_RES(8);
        else if (grp == 4) LAUNCH_RES(4);
        else LAUNCH_RES(2);
#undef LAUNCH_RES
        g_last_kernels = "msda_fwd_d32_res";
        return semidetr::launch_status("msda_fwd_d32_res");
    }
    if ((pixels && g_fwd_variant == 0) || g_fwd_variant == 408 || g_fwd_variant == 804 || g_fwd_variant == 216) {
        SEMIDETR_REQUIRE(pixels, SEMIDETR_E_BADARG, "msda_forward: patch tiling needs SEMIDETR_MSDA_QUERIES_ARE_PIXELS");
        // grid sizing hint: about the number of 32-pixel patches of a usual pyramid (ragged edges included)
        const int bound = (S + 31) / 32 * 5 / 4 + 4 * L;
        SEMIDETR_REQUIRE((int64_t)N * bound * M < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_forward: grid too large");
        const size_t lds = (size_t)32 * (L * P + 1) * 32;
        // measured at the 800x1333 encoder shape, bs 4, with the head rotation of tile_of_block: 4x8 246 us, 8x4 252,
        // 2x16 253 (before the rotation: strips 299, 4x8 284, 8x4 281, 2x16 286)
        if (g_fwd_variant == 408) LAUNCH_FWD(1, 4, 408, bound);
        else if (g_fwd_variant == 216) LAUNCH_FWD(1, 4, 216, bound);
        else if (g_fwd_variant == 804) LAUNCH_FWD(1, 4, 804, bound);
        else LAUNCH_FWD(1, 4, 408, bound);
        g_last_kernels = "msda_fwd_d32<1, 4, 408";
        return semidetr::launch_status("msda_fwd_d32<patch>");
    }
    const int unroll = g_fwd_variant >= 10 && g_fwd_variant < 100 ? g_fwd_variant / 10 : 4;
    if (split == 1) { if (unroll == 2) LAUNCH_FWD(1, 2, 0, tiles); else if (unroll == 1) LAUNCH_FWD(1, 1, 0, tiles); else LAUNCH_FWD(1, 4, 0, tiles); }
    else if (split == 2) LAUNCH_FWD(2, 4, 0, tiles);
    else LAUNCH_FWD(4, 4, 0, tiles);
#undef LAUNCH_FWD
    g_last_kernels = split == 1 ? "msda_fwd_d32<1, 4, 0" : (split == 2 ? "msda_fwd_d32<2, 4, 0" : "msda_fwd_d32<4, 4, 0");
    return semidetr::launch_status("msda_fwd_d32");
}

// ---- strips backward (any query set).  Experiment (variants 808 / 832): SPLIT in two launches so that the zero fill of
// grad_value overlaps the half that does not need it: a side stream (one per host thread, joined back before the call returns to the caller's
// stream order) runs the fill while the caller's stream runs msda_bwd_gather_d32 (grad_sampling_loc / grad_attn_weight:
// reads value corners, never touches grad_value); the scatter-only instantiation of msda_bwd_d32 follows once both are
// done.  At the BASELINE micro-benchmark shape the 45.5 MB fill is 7-8 us of a 43 us backward.  Works under stream
// capture (the side stream forks from and joins the capturing stream through events).
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int device = -1;
};
thread_local SideStream g_side;

int side_stream(SideStream **out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward: hipGetDevice: %s", hipGetErrorString(e));
    if (g_side.device != dev) {      // first call of this thread on this device (objects of another device are leaked: rare)
        e = hipStreamCreateWithFlags(&g_side.s, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g_side.fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g_side.join, hipEventDisableTiming);
        if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward: side stream: %s", hipGetErrorString(e));
        g_side.device = dev;
    }
    *out = &g_side;
    return SEMIDETR_OK;
}

// zero-fill `bytes` at `ptr` on the side stream, ordered after everything already queued on `st`
int fill_on_side(hipStream_t st, void *ptr, size_t bytes, SideStream **side)
{
    if (int rc = side_stream(side)) return rc;
    hipError_t e = hipEventRecord((*side)->fork, st);
    if (e == hipSuccess) e = hipStreamWaitEvent((*side)->s, (*side)->fork, 0);
    if (e == hipSuccess) e = hipMemsetAsync(ptr, 0, bytes, (*side)->s);
    if (e == hipSuccess) e = hipEventRecord((*side)->join, (*side)->s);
    if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward fill on the side stream: %s", hipGetErrorString(e));
    return SEMIDETR_OK;
}
int join_side(hipStream_t st, SideStream *side)
{
    const hipError_t e = hipStreamWaitEvent(st, side->join, 0);
    if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward join: %s", hipGetErrorString(e));
    return SEMIDETR_OK;
}

template <typename IO>
int launch_strips_backward(hipStream_t st, size_t fill, const float *grad_out, const float *value,
                           const int64_t *spatial_shapes, const int64_t *level_start, const IO &io, int N, int S, int M,
                           int L, int Lq, int P, int rpb, int tiles, unsigned grid, size_t lds, float *grad_value)
{
    // default: ONE kernel after the fill on the caller's stream.  The split below lost on MI355X (measured, r02): the
    // fork / join through events costs more than the 7-8 us of fill it hides -- micro-benchmark backward 42.8 -> 63.1 us,
    // decoder bs 4 242 -> 267 us, encoder bs 4 875 -> 893 us.  Kept selectable (808 / 832) as the evidence.
    // level-aggregated scatter + gather in ONE merged launch: default for launches with at least 512 (image, query) pairs
    // (measured: decoder bs 4 / Lq 1100 240 -> 163 us, bs 1 66 -> 54 us, BASELINE micro-benchmark shape 43.2 -> 39.7 us;
    // as two launches -- gather, then scatter -- the micro-benchmark shape gained nothing: 8 + 27 us against 35 us fused)
    if ((g_bwd_variant == 900 || (g_bwd_variant == 0 && (int64_t)N * Lq >= 512)) && P <= 8) {
        // fill, then ONE launch: level-aggregated scatter workgroups + gather workgroups side by side
        hipError_t e = hipMemsetAsync(grad_value, 0, fill, st);
        if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward memset: %s", hipGetErrorString(e));
        // chunks of <= kLvlQ queries; small launches are cut finer so that at least ~128 scatter workgroups exist.  Measured
        // at the micro-benchmark shape (N=2, Lq=300: 64 (image, head, level) triples): 1 chunk 42.1 us, 2 chunks 36.1 us,
        // 4 chunks 40.1 us, 8 chunks 38.7 us -- bigger chunks aggregate more, a single one leaves the chip idle.
        int chunks = (Lq + kLvlQ - 1) / kLvlQ;
        static const int target_wgs = getenv("SEMIDETR_LVL_WGS") ? atoi(getenv("SEMIDETR_LVL_WGS")) : 128;   // tuning aid
        const int want = (target_wgs + N * L * M - 1) / (N * L * M);
        chunks = std::max(chunks, std::min(want, (Lq + 63) / 64));
        const int chunk_q = (Lq + chunks - 1) / chunks;
        const int gt = (Lq + 31) / 32;                                   // gather: 32 query rows per 256-thread block
        const int64_t gblocks = (int64_t)N * gt * M, sblocks = (int64_t)N * chunks * L * M;
        const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
        const size_t slds = std::max((size_t)kLvlQ * kD * 4 + ((size_t)chunk_q * P * 4 + 8) * 8 + (size_t)2 * kLvlRows * 4,
                                     2 * half_f4 * 16);
        const int64_t grid = sblocks + (gblocks + 1) / 2;
        SEMIDETR_REQUIRE(grid < INT32_MAX && slds <= 159 * 1024, SEMIDETR_E_TOOLARGE, "msda_backward: merged launch too large");
#define LAUNCH_MERGED(KLP_)                                                                                          \
        do {                                                                                                             \
            static bool lds_ok = false; /* dynamic LDS above 64 KB has to be allowed once per kernel (static LDS too) */ \
            if (!lds_ok) {                                                                                               \
                const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_lvl_merged<IO, KLP_>), \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);      \
                if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_backward: hipFuncSetAttribute: %s", hipGetErrorString(ae)); \
                lds_ok = true;                                                                                           \
            }                                                                                                            \
            hipLaunchKernelGGL((msda_bwd_lvl_merged<IO, KLP_>), dim3((unsigned)grid), dim3(kLvlThreads), slds, st, grad_out, \
                               value, spatial_shapes, level_start, io, S, M, L, Lq, P, chunks, chunk_q, (int)sblocks, gt, \
                               (int)gblocks, grad_value);                                                              \
        } while (0)
        if (L * P == 16) LAUNCH_MERGED(16);
        else LAUNCH_MERGED(0);
#undef LAUNCH_MERGED
        g_last_kernels = "fillBufferAligned+msda_bwd_lvl_merged";
        return semidetr::launch_status("msda_bwd_lvl_merged");
    }
    if (g_bwd_variant == 902 && P <= 8 && (reinterpret_cast<uintptr_t>(grad_value) & 15) == 0) {
        // EXPERIMENT: cooperative zero fill inside the merged launch (msda_bwd_lvl_coop), no hipMemsetAsync.  The three
        // counters of a launch live in a library-owned device buffer (64 slots handed out round robin; the kernel leaves its
        // slot zeroed).  Not capturable on its first call (hipMalloc), not meant for concurrent replays of one captured graph.
        static unsigned *sync_buf = nullptr;
        static int sync_dev = -1, next_slot = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!sync_buf || dev != sync_dev) {
            unsigned *p = nullptr;
            hipError_t e = hipMalloc(&p, 64 * 4 * sizeof(unsigned));
            if (e == hipSuccess) e = hipMemset(p, 0, 64 * 4 * sizeof(unsigned));
            if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward: sync buffer: %s", hipGetErrorString(e));
            sync_buf = p;
            sync_dev = dev;
        }
        unsigned *slot = sync_buf + 4 * (next_slot++ & 63);
        int chunks = (Lq + kLvlQ - 1) / kLvlQ;
        const int want = (128 + N * L * M - 1) / (N * L * M);
        chunks = std::max(chunks, std::min(want, (Lq + 63) / 64));
        const int chunk_q = (Lq + chunks - 1) / chunks;
        const int gt = (Lq + 31) / 32;
        const int64_t gblocks = (int64_t)N * gt * M, sblocks = (int64_t)N * chunks * L * M;
        const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
        const size_t slds = std::max((size_t)kLvlQ * kD * 4 + ((size_t)chunk_q * P * 4 + 8) * 8 + (size_t)2 * kLvlRows * 4,
                                     2 * half_f4 * 16);
        const int64_t grid = sblocks + (gblocks + 1) / 2;
        SEMIDETR_REQUIRE(grid < INT32_MAX && slds <= 159 * 1024, SEMIDETR_E_TOOLARGE, "msda_backward: merged launch too large");
#define LAUNCH_COOP(KLP_)                                                                                            \
        do {                                                                                                             \
            static bool lds_ok = false;                                                                                  \
            if (!lds_ok) {                                                                                               \
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_lvl_coop<IO, KLP_>),                  \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);                      \
                lds_ok = true;                                                                                           \
            }                                                                                                            \
            hipLaunchKernelGGL((msda_bwd_lvl_coop<IO, KLP_>), dim3((unsigned)grid), dim3(kLvlThreads), slds, st, grad_out,   \
                               value, spatial_shapes, level_start, io, S, M, L, Lq, P, chunks, chunk_q, (int)sblocks, gt, \
                               (int)gblocks, grad_value, reinterpret_cast<float4 *>(grad_value), (int64_t)(fill / 16),  \
                               slot, 1 << 20);                                                                         \
        } while (0)
        if (L * P == 16) LAUNCH_COOP(16);
        else LAUNCH_COOP(0);
#undef LAUNCH_COOP
        g_last_kernels = "msda_bwd_lvl_coop";
        return semidetr::launch_status("msda_bwd_lvl_coop");
    }
    if (g_bwd_variant == 901 && P <= 8) {
        // experiment: NO memset -- the gather launch zero-fills grad_value as a side job, the level-aggregated scatter
        // follows as its own launch
        int chunks = (Lq + kLvlQ - 1) / kLvlQ;
        const int want = (128 + N * L * M - 1) / (N * L * M);
        chunks = std::max(chunks, std::min(want, (Lq + 63) / 64));
        const int chunk_q = (Lq + chunks - 1) / chunks;
        const int gt = (Lq + 31) / 32;
        const size_t glds = (size_t)32 * (L * P + 1) * 32 + 2 * kMaxLevels * sizeof(float);
        SEMIDETR_REQUIRE(fill % 16 == 0, SEMIDETR_E_BADARG, "msda_backward: grad_value size not a multiple of 16 bytes");
        if (L * P == 16)
            hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt,
                               reinterpret_cast<float4 *>(grad_value), (int64_t)(fill / 16));
        else
            hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 0>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt,
                               reinterpret_cast<float4 *>(grad_value), (int64_t)(fill / 16));
        if (int rc = semidetr::launch_status("msda_bwd_gather_d32")) return rc;
        const size_t slds = (size_t)kLvlQ * kD * 4 + ((size_t)chunk_q * P * 4 + 8) * 8 + (size_t)2 * kLvlRows * 4;
        static bool lds_ok = false;
        if (!lds_ok) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_scatter_d32_lvl<IO>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
            lds_ok = true;
        }
        hipLaunchKernelGGL((msda_bwd_scatter_d32_lvl<IO>), dim3((unsigned)((int64_t)N * chunks * L * M)), dim3(kLvlThreads), slds, st,
                           grad_out, spatial_shapes, level_start, io, S, M, L, Lq, P, chunks, chunk_q, grad_value);
        g_last_kernels = "msda_bwd_gather_d32+msda_bwd_scatter_d32_lvl";
        return semidetr::launch_status("msda_bwd_scatter_d32_lvl");
    }
    const bool fused = g_bwd_variant != 808 && g_bwd_variant != 832;
    if (fused) {
        hipError_t e = hipMemsetAsync(grad_value, 0, fill, st);
        if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward memset: %s", hipGetErrorString(e));
        if (rpb == 32)
            hipLaunchKernelGGL((msda_bwd_d32<32, IO>), dim3(grid), dim3(256), lds, st, grad_out, value, spatial_shapes,
                               level_start, io, S, M, L, Lq, P, tiles, grad_value);
        else
            hipLaunchKernelGGL((msda_bwd_d32<8, IO>), dim3(grid), dim3(256), lds, st, grad_out, value, spatial_shapes,
                               level_start, io, S, M, L, Lq, P, tiles, grad_value);
        g_last_kernels = rpb == 32 ? "fillBufferAligned+msda_bwd_d32<32" : "fillBufferAligned+msda_bwd_d32<8";
        return semidetr::launch_status("msda_bwd_d32");
    }
    SideStream *side = nullptr;
    if (int rc = fill_on_side(st, grad_value, fill, &side)) return rc;
    {   // the two small gradients (32 query rows per workgroup)
        const int gt = (Lq + 31) / 32;
        const size_t glds = (size_t)32 * (L * P + 1) * 32 + 2 * kMaxLevels * sizeof(float);
        if (L * P == 16)
            hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt);
        else
            hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 0>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt);
        const int grc = semidetr::launch_status("msda_bwd_gather_d32");
        if (int rc = join_side(st, side)) return rc;            // never leave the side stream un-joined
        if (grc) return grc;
    }
    if (rpb == 32)
        hipLaunchKernelGGL((msda_bwd_d32<32, IO, true>), dim3(grid), dim3(256), lds, st, grad_out, value, spatial_shapes,
                           level_start, io, S, M, L, Lq, P, tiles, grad_value);
    else
        hipLaunchKernelGGL((msda_bwd_d32<8, IO, true>), dim3(grid), dim3(256), lds, st, grad_out, value, spatial_shapes,
                           level_start, io, S, M, L, Lq, P, tiles, grad_value);
    g_last_kernels = rpb == 32 ? "fillBufferAligned+msda_bwd_gather_d32+msda_bwd_d32<32" : "fillBufferAligned+msda_bwd_gather_d32+msda_bwd_d32<8";
    return semidetr::launch_status("msda_bwd_d32<scatter>");
}

template <typename IO>
int launch_fast_backward(hipStream_t st, const float *grad_out, const float *value, const int64_t *spatial_shapes,
                         const int64_t *level_start, const IO &io, int N, int S, int M, int L, int Lq, int P,
                         int flags, float *grad_value)
{
    const bool pixels = (flags & SEMIDETR_MSDA_QUERIES_ARE_PIXELS) != 0;
    SEMIDETR_REQUIRE(!pixels || Lq == S, SEMIDETR_E_BADARG,
                     "msda_backward: SEMIDETR_MSDA_QUERIES_ARE_PIXELS needs num_query == spatial_size");
    const size_t fill = sizeof(float) * (size_t)N * S * M * kD;
    const bool win_ok = P == kPT;
    if ((pixels && S < (1 << 24) && win_ok && g_bwd_variant == 0) || ((g_bwd_variant >= 64 && g_bwd_variant <= 74) || (g_bwd_variant >= 690 && g_bwd_variant <= 699) || (g_bwd_variant >= 6900 && g_bwd_variant <= 7009))) {
        SEMIDETR_REQUIRE(pixels && S < (1 << 24), SEMIDETR_E_BADARG,
                         "msda_backward: the self-attention kernels need SEMIDETR_MSDA_QUERIES_ARE_PIXELS and spatial_size < 2^24");
        // the unrolled gather launch clears grad_value as a side job (no hipMemsetAsync): the scatter that accumulates into it
        // is the NEXT launch.  Measured (tools/r02_fillgather_try.sh): encoder bs 4 774 -> 766 us, bs 1 204 -> 201 us; 6991
        // forces the memset.  (The same idea for arbitrary query sets -- gather + fill, then the level scatter as a second
        // launch, variant 901 -- loses against the merged launch: micro-benchmark 36.2 -> 41-44 us, decoder bs 4 161 -> 169.)
        const bool rw_gather = g_bwd_variant >= 7000 && g_bwd_variant <= 7009 && P == kPT && (L == 4 || L == 5);
        const bool gather_kernel_runs = !(g_bwd_variant == 697 || g_bwd_variant == 68);
        const bool fill_in_gather = (L * P == 16 || rw_gather) && gather_kernel_runs && g_bwd_variant != 6991 && g_bwd_variant != 66 &&
                                    g_bwd_variant != 67 && g_bwd_variant != 6962 && g_bwd_variant != 6952 && g_bwd_variant != 6948 &&
                                    (reinterpret_cast<uintptr_t>(grad_value) & 15) == 0;
        if (!fill_in_gather) {
            hipError_t e = hipMemsetAsync(grad_value, 0, fill, st);
            if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward memset: %s", hipGetErrorString(e));
        }
        if (g_bwd_variant == 697 && L * P == 16 && P == kPT && S < (1 << 23)) {
            // experiment: region scatter + gather in ONE launch, roles dealt out in groups of eight workgroups
            const int rbound = (S + 127) / 128 * 5 / 4 + 4 * L;
            const int gbound = (S + 31) / 32 * 5 / 4 + 4 * L;
            const int64_t sblocks = (int64_t)N * rbound * M, gblocks = (int64_t)N * gbound * M, gwgs = (gblocks + 1) / 2;
            const int64_t sgroups = (sblocks + 7) / 8, ggroups = (gwgs + 7) / 8;
            const int period = (int)std::max<int64_t>(2, (sgroups + ggroups) / sgroups);
            const int64_t groups = std::max(sgroups + ggroups, (sgroups - 1) * period + 1);
            SEMIDETR_REQUIRE(groups * 8 < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
            const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
            const size_t mlds = std::max(reg_lds_bytes<512, 208, 24, 32>(), 2 * half_f4 * 16);
            auto kern = &msda_bwd_encreg_merged<IO, 16>;
            static bool lds_ok = false;
            if (!lds_ok) {
                const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
                if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_backward: hipFuncSetAttribute: %s", hipGetErrorString(ae));
                lds_ok = true;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)(groups * 8)), dim3(512), mlds, st, grad_out, value, spatial_shapes,
                               level_start, io, S, M, L, P, rbound, (int)sblocks, gbound, (int)gblocks, (int)sgroups, period,
                               grad_value);
            g_last_kernels = "fillBufferAligned+msda_bwd_encreg_merged";
            return semidetr::launch_status("msda_bwd_encreg_merged");
        }
        if (g_bwd_variant == 68 && L * P == 16 && P == kPT) {
            // Experiment (variant 68): ONE launch, windowed-scatter workgroups interleaved with pairs of gather blocks
            // (msda_bwd_enc_merged).  Measured at bs 4: 1771 us against 879 us for the two launches -- every workgroup of
            // the launch carries the scatter's 77 KB of LDS, so a CU holds two workgroups in any mix and the gather, which
            // needs 16-24 waves per CU to stream, is left with 8.  (The same trick WINS for arbitrary query sets, where
            // the scatter's LDS is small enough for the halves to share CUs at full occupancy: msda_bwd_lvl_merged.)
            const int tiles_bound = (S + 255) / 256 * 5 / 4 + 4 * L;
            const int gbound = (S + 31) / 32 * 5 / 4 + 4 * L;
            const int64_t sblocks = (int64_t)N * tiles_bound * M, gblocks = (int64_t)N * gbound * M;
            const int64_t gwgs = (gblocks + 1) / 2, total = sblocks + gwgs;
            const int period = (int)std::max<int64_t>(2, total / sblocks);         // every period-th workgroup scatters
            // all scatter indices 0, period, 2*period, ... must fall inside the grid
            const int64_t grid = std::max(total, (sblocks - 1) * period + 1);
            SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
            const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
            const size_t mlds = std::max(win_lds_bytes<16, 16, 32, 32>(), 2 * half_f4 * 16);
            static bool lds_ok = false;
            if (!lds_ok) {
                const hipError_t ae = hipFuncSetAttribute(
                    reinterpret_cast<const void *>(&msda_bwd_enc_merged<IO, 16, 16, 16, 32, 32>),
                    hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
                if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_backward: hipFuncSetAttribute: %s", hipGetErrorString(ae));
                lds_ok = true;
            }
            hipLaunchKernelGGL((msda_bwd_enc_merged<IO, 16, 16, 16, 32, 32>), dim3((unsigned)grid), dim3(kWinThreads), mlds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, P, tiles_bound, (int)sblocks, gbound,
                               (int)gblocks, period, grad_value);
            g_last_kernels = "fillBufferAligned+msda_bwd_enc_merged";
            return semidetr::launch_status("msda_bwd_enc_merged");
        }
        if (rw_gather) {   // gather half through the region windows (msda_rw.h)
            float4 *z = fill_in_gather ? reinterpret_cast<float4 *>(grad_value) : nullptr;
            const int rc = L == 4 ? launch_rw_cfg<IO, 4, true>(g_bwd_variant - 7000, st, grad_out, value, spatial_shapes, level_start,
                                                               io, N, S, M, nullptr, z, (int64_t)(fill / 16))
                                  : launch_rw_cfg<IO, 5, true>(0, st, grad_out, value, spatial_shapes, level_start, io, N, S, M,
                                                               nullptr, z, (int64_t)(fill / 16));
            if (rc) return rc;
        } else
        {   // gather half: the two small gradients, streams like the forward
            const int gt = (Lq + 31) / 32;
            const size_t glds = (size_t)32 * (L * P + 1) * 32 + 2 * kMaxLevels * sizeof(float);
            // DINO (L * P == 16): sample loop unrolled, results in registers, 8 x 4 query patches like the forward;
            // measured at the encoder shape, bs 4: generic strips 370 us, unrolled strips 346 us; 66 / 67 force them
            const int gbound = (S + 31) / 32 * 5 / 4 + 4 * L;      // patch grid hint, see launch_fast_forward
            if (L * P == 16 && g_bwd_variant == 6962)            // tuning: 8 loads in flight, 6 waves per SIMD
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408, 6, 2>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound);
            else if (L * P == 16 && g_bwd_variant == 6952)       // 8 loads in flight, 5 waves per SIMD
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408, 5, 2>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound);
            else if (L * P == 16 && g_bwd_variant == 6948)       // timing aid: nothing stored
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408, 4, 104>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound);
            else if (fill_in_gather)
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound,
                                   reinterpret_cast<float4 *>(grad_value), (int64_t)(fill / 16));
            else if (L * P == 16 && g_bwd_variant != 66 && g_bwd_variant != 67)
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound);
            else if (L * P == 16 && g_bwd_variant == 67)
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                                   grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt);
            else
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 0>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                                   grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt);
            if (int rc = semidetr::launch_status("msda_bwd_gather_d32")) return rc;
        }
        if ((g_bwd_variant == 0 || (g_bwd_variant >= 69 && g_bwd_variant <= 69) || (g_bwd_variant >= 690 && g_bwd_variant <= 699) || (g_bwd_variant >= 6900 && g_bwd_variant <= 7009)) && P == kPT && S < (1 << 23)) {
            // region-owned windowed scatter (msda_region.h): one workgroup per tile of the finest level, all query levels.
            // DEFAULT since round 2.  Measured at the 800x1333 encoder shape (backward incl. fill + gather): bs 4 886 us
            // (windowed kernel, variant 65) -> 867 us (16 x 16 regions, 1024 threads, 690) -> 823 us (8 x 16 regions, 512
            // threads, two workgroups per CU); bs 1 248 -> 226 -> 216 us; row atomics 590 MB -> 358 MB (16 x 16).
            const bool small = g_bwd_variant != 690;              // 8 x 16 regions, 512 threads, two workgroups per CU
            const int rpx = g_bwd_variant == 692 || g_bwd_variant == 693 ? 64 : (small ? 128 : 256);
            const int rbound = (S + rpx - 1) / rpx * 5 / 4 + 4 * L;
            const int64_t rgrid = (int64_t)N * rbound * M;
            SEMIDETR_REQUIRE(rgrid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
#define LAUNCH_REG(NT_, Q_, RH_, RW_, WH_, WW_) LAUNCH_REGW(NT_, Q_, RH_, RW_, WH_, WW_, 4)
#define LAUNCH_REGW(NT_, Q_, RH_, RW_, WH_, WW_, WPE_) LAUNCH_REGU(NT_, Q_, RH_, RW_, WH_, WW_, WPE_, 8)
#define LAUNCH_REGU(NT_, Q_, RH_, RW_, WH_, WW_, WPE_, WU_)                                                                   \
            do {                                                                                                         \
                static bool lds_ok = false;                                                                              \
                auto kern = &msda_bwd_scatter_d32_reg<IO, NT_, Q_, RH_, RW_, WH_, WW_, 0, WPE_, WU_>;                                  \
                if (!lds_ok) {                                                                                           \
                    const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                      \
                                                              hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);  \
                    if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_backward: hipFuncSetAttribute: %s", hipGetErrorString(ae)); \
                    lds_ok = true;                                                                                       \
                }                                                                                                        \
                const size_t rlds = reg_lds_bytes<NT_, Q_, WH_, WW_>();                                                  \
                hipLaunchKernelGGL(kern, dim3((unsigned)rgrid), dim3(NT_), rlds, st, grad_out, spatial_shapes, level_start, \
                                   io, S, M, L, rbound, grad_value);                                                    \
            } while (0)
            if (g_bwd_variant == 696) {                                             // instrumented build
                auto kern = &msda_bwd_scatter_d32_reg<IO, 512, 208, 8, 16, 24, 32, 1>;
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
                const size_t rlds = reg_lds_bytes<512, 208, 24, 32>();
                hipLaunchKernelGGL(kern, dim3((unsigned)rgrid), dim3(512), rlds, st, grad_out, spatial_shapes, level_start, io, S,
                                   M, L, rbound, grad_value);
            } else if (g_bwd_variant == 691) LAUNCH_REG(512, 208, 16, 8, 32, 24);   // tuning variants
            else if (g_bwd_variant == 692) LAUNCH_REG(256, 112, 8, 8, 24, 24);
            else if (g_bwd_variant == 693) LAUNCH_REG(512, 112, 8, 8, 24, 24);
            else if (g_bwd_variant == 694) LAUNCH_REG(512, 208, 8, 16, 32, 32);
            else if (g_bwd_variant == 695) LAUNCH_REG(768, 208, 8, 16, 24, 32);
            else if (g_bwd_variant == 698) LAUNCH_REGW(512, 176, 8, 16, 24, 32, 6);   // three workgroups per CU
            else if (g_bwd_variant == 699) LAUNCH_REGW(512, 176, 8, 16, 24, 32, 4);   // same LDS, register budget of two
            else if (g_bwd_variant == 6981) LAUNCH_REGU(512, 208, 8, 16, 24, 32, 4, 4);    // walk unrolled by 4
            else if (g_bwd_variant == 6982) LAUNCH_REGU(512, 208, 8, 16, 24, 32, 4, 16);   // ... by 16
            else if (small) LAUNCH_REG(512, 208, 8, 16, 24, 32);
            else LAUNCH_REG(1024, 384, 16, 16, 32, 32);
#undef LAUNCH_REG
#undef LAUNCH_REGW
#undef LAUNCH_REGU
            g_last_kernels = rw_gather ? "msda_rw_d32<gather>+msda_bwd_scatter_d32_reg" : (fill_in_gather ? "msda_bwd_gather_d32+msda_bwd_scatter_d32_reg" : "fillBufferAligned+msda_bwd_gather_d32+msda_bwd_scatter_d32_reg");
            return semidetr::launch_status("msda_bwd_scatter_d32_reg");
        }
        // grad_value: destination-owned tiles (msda_dest.h) unless a windowed variant is forced (64..67) or the pyramid
        // has more levels than the kernel's LDS tables hold
        if (L <= kDestMaxLevels && g_bwd_variant >= 70 && g_bwd_variant <= 74) {
            // grid sizing hint: about 2.5 units per 256 rows of a usual 4-level pyramid (coarse tiles are split);
            // workgroups take units slot, slot + bound, ... so any bound >= 1 is correct
            const int bound = (S / 256 + 1) * 3 + 64;
            const int64_t grid = (int64_t)N * bound * M;
            SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
            if (g_bwd_variant == 74)
                hipLaunchKernelGGL((msda_bwd_dest_d32<IO, 16, 16, 8, 3>), dim3((unsigned)grid), dim3(kDestThreads), 0, st,
                                   grad_out, spatial_shapes, level_start, io, S, M, L, P, bound, grad_value);
            else if (g_bwd_variant == 73)
                hipLaunchKernelGGL((msda_bwd_dest_d32<IO, 16, 16, 8, 2>), dim3((unsigned)grid), dim3(kDestThreads), 0, st,
                                   grad_out, spatial_shapes, level_start, io, S, M, L, P, bound, grad_value);
            else if (g_bwd_variant == 72)
                hipLaunchKernelGGL((msda_bwd_dest_d32<IO, 16, 16, 8, 1>), dim3((unsigned)grid), dim3(kDestThreads), 0, st,
                                   grad_out, spatial_shapes, level_start, io, S, M, L, P, bound, grad_value);
            else if (g_bwd_variant == 71)
                hipLaunchKernelGGL((msda_bwd_dest_d32<IO, 16, 16, 6>), dim3((unsigned)grid), dim3(kDestThreads), 0, st,
                                   grad_out, spatial_shapes, level_start, io, S, M, L, P, bound, grad_value);
            else
                hipLaunchKernelGGL((msda_bwd_dest_d32<IO, 16, 16, 8>), dim3((unsigned)grid), dim3(kDestThreads), 0, st,
                                   grad_out, spatial_shapes, level_start, io, S, M, L, P, bound, grad_value);
            g_last_kernels = fill_in_gather ? "msda_bwd_gather_d32+msda_bwd_dest_d32" : "fillBufferAligned+msda_bwd_gather_d32+msda_bwd_dest_d32";
            return semidetr::launch_status("msda_bwd_dest_d32");
        }
        // windowed (source-owned) kernel: patches are enumerated on the device (the level table lives in device
        // memory); a workgroup takes patches slot, slot + tiles_bound, ... so any bound >= 1 is correct.
        // measured at the 800x1333 encoder shape, bs 4: 8x16 patches 687 us / 784 MB of row atomics, 16x16 patches
        // 593 us / 604 MB (fewer halo rows per query); 64 forces the small patch
        SEMIDETR_REQUIRE(P == kPT, SEMIDETR_E_BADARG, "msda_backward: the windowed kernel needs num_point == 4");
        const bool big = g_bwd_variant != 64;
        const int patch = big ? 256 : 128;
        const int tiles_bound = (S + patch - 1) / patch * 5 / 4 + 4 * L;
        const int64_t grid = (int64_t)N * tiles_bound * M;
        SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
        const size_t wlds = big ? win_lds_bytes<16, 16, 32, 32>() : win_lds_bytes<8, 16, 24, 32>();
#define ALLOW_LDS(KERNEL)                                                                                            \
        do {                                                                                                             \
            static bool lds_ok = false;                                                                                  \
            if (!lds_ok) {                                                                                               \
                const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL),                       \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);      \
                if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_backward: hipFuncSetAttribute: %s", hipGetErrorString(ae)); \
                lds_ok = true;                                                                                           \
            }                                                                                                            \
        } while (0)
        if (big) {
            ALLOW_LDS((msda_bwd_scatter_d32_win<IO, 16, 16, 32, 32>));
            hipLaunchKernelGGL((msda_bwd_scatter_d32_win<IO, 16, 16, 32, 32>), dim3((unsigned)grid), dim3(kWinThreads),
                               wlds, st, grad_out, spatial_shapes, level_start, io, S, M, L, tiles_bound, grad_value);
        } else {
            ALLOW_LDS((msda_bwd_scatter_d32_win<IO, 8, 16, 24, 32>));
            hipLaunchKernelGGL((msda_bwd_scatter_d32_win<IO, 8, 16, 24, 32>), dim3((unsigned)grid), dim3(kWinThreads),
                               wlds, st, grad_out, spatial_shapes, level_start, io, S, M, L, tiles_bound, grad_value);
        }
        g_last_kernels = fill_in_gather ? "msda_bwd_gather_d32+msda_bwd_scatter_d32_win" : "fillBufferAligned+msda_bwd_gather_d32+msda_bwd_scatter_d32_win";
        return semidetr::launch_status("msda_bwd_scatter_d32_win");
    }
    // rows per workgroup: 32 normally, 8 when the problem is too small to fill 256 CUs with 32-row tiles
    int rpb = (int64_t)N * M * ((Lq + 31) / 32) >= 1024 ? 32 : 8;
    if (g_bwd_variant == 8 || g_bwd_variant == 32) rpb = g_bwd_variant % 100;
    if (g_bwd_variant == 808 || g_bwd_variant == 832) rpb = g_bwd_variant - 800;
    const int tiles = (Lq + rpb - 1) / rpb;
    const int64_t grid = (int64_t)N * tiles * M;
    SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
    const size_t lds = (size_t)rpb * (L * P + 1) * 32 + 2 * kMaxLevels * sizeof(float);
    return launch_strips_backward<IO>(st, fill, grad_out, value, spatial_shapes, level_start, io, N, S, M, L, Lq, P, rpb,
                                      tiles, (unsigned)grid, lds, grad_value);
}

}  // namespace

extern "C" const char *semidetr_msda_last_kernels(void) { return g_last_kernels; }

// tuning aid: per-phase cycle counters of the instrumented destination-owned kernel (variant 73); reset = 1 zeroes them
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
    g_fwd_variant = fwd_variant;
    g_bwd_variant = bwd_variant;
}

extern "C" int semidetr_msda_forward_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                                         const int64_t *level_start, const float *sampling_loc,
                                         const float *attn_weight, int batch, int spatial_size,
                                         int num_heads, int channels, int num_levels, int num_query,
                                         int num_point, int flags, float *out)
{
    const int N = batch, S = spatial_size, M = num_heads, D = channels, L = num_levels, Lq = num_query,
              P = num_point;
    if (g_fwd_variant == 99 || !fast_ok(value, sampling_loc, out, D, L, P) || !slice_ok(S, M) || !heads_ok(M))
        return forward_impl<float>(stream, value, spatial_shapes, level_start, sampling_loc, attn_weight, N,
                                   S, M, D, L, Lq, P, out);
    if (int rc = check_common(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L,
                              Lq, P, 4))
        return rc;
    SEMIDETR_REQUIRE(out, SEMIDETR_E_BADARG, "msda_forward: null output");
    const LocAttnIO io = {sampling_loc, attn_weight, nullptr, nullptr};
    return launch_fast_forward(semidetr::as_stream(stream), value, spatial_shapes, level_start, io, N, S, M, L, Lq,
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
    if (g_bwd_variant == 99 || !fast_ok(value, sampling_loc, grad_out, D, L, P) || !slice_ok(S, M) || !heads_ok(M) ||
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
    return launch_fast_backward(semidetr::as_stream(stream), grad_out, value, spatial_shapes, level_start, io, N, S,
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
    SEMIDETR_REQUIRE((((uintptr_t)value | (uintptr_t)off) & 15) == 0, SEMIDETR_E_BADARG,
                     "msda_fused: value / sampling_offsets must be 16-byte aligned");
    return SEMIDETR_OK;
}

extern "C" int semidetr_msda_fused_forward_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                                               const int64_t *level_start, const float *reference_points,
                                               int ref_dim, const float *sampling_offsets,
                                               const float *attn_logits, int batch, int spatial_size,
                                               int num_heads, int channels, int num_levels, int num_query,
                                               int num_point, int flags, float *out)
{
    if (int rc = check_fused(value, spatial_shapes, level_start, reference_points, ref_dim, sampling_offsets,
                             attn_logits, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point))
        return rc;
    SEMIDETR_REQUIRE(out && ((uintptr_t)out & 15) == 0, SEMIDETR_E_BADARG, "msda_fused_forward: bad output pointer");
    const RawIO io = {reference_points, sampling_offsets, attn_logits, nullptr, nullptr, ref_dim, num_heads,
                      num_levels};
    return launch_fast_forward(semidetr::as_stream(stream), value, spatial_shapes, level_start, io, batch,
                               spatial_size, num_heads, num_levels, num_query, num_point, flags, out);
}

extern "C" int semidetr_msda_fused_backward_f32(void *stream, const float *grad_out, const float *value,
                                                const int64_t *spatial_shapes, const int64_t *level_start,
                                                const float *reference_points, int ref_dim,
                                                const float *sampling_offsets, const float *attn_logits, int batch,
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
    const RawIO io = {reference_points, sampling_offsets, attn_logits, grad_sampling_offsets, grad_attn_logits,
                      ref_dim, num_heads, num_levels};
    return launch_fast_backward(semidetr::as_stream(stream), grad_out, value, spatial_shapes, level_start, io, batch,
                                spatial_size, num_heads, num_levels, num_query, num_point, flags, grad_value);
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
