// Device-resident rectangular linear-sum-assignment for gfx950, bit-exact with
// scipy.optimize.linear_sum_assignment (scipy 1.15.3; the solver the reference calls at
// thirdparty/mmdetection/mmdet/core/bbox/assigners/hungarian_assigner.py:136 and
// detr_ssod/models/dino_detr_ssod.py:279), followed by the assignment scatter of
// hungarian_assigner.py:142-147.
//
// Why on the device: the reference copies every cost matrix to the host (`cost.detach().cpu()`,
// hungarian_assigner.py:132 -- a blocking sync), solves it in scipy and copies indices back; a Semi-DETR
// step does that ~39 times.  Here all problems of a loss() call are solved in ONE launch, one 64-lane
// wavefront per problem, with every solver array in LDS; nothing leaves HBM.
//
// Algorithm (published: Crouse 2016, shortest augmenting paths with dual variables; restated in
// oracle/lsap_oracle.c).  The sequential inner scan over the not-yet-scanned columns becomes a
// wavefront-parallel scan + a lexicographic wave reduction that reproduces scipy's tie rule exactly:
// among the columns with the minimal reduced cost pick the LAST unassigned one in list order if there is
// any, otherwise the FIRST one in list order.  The list ("remaining") is kept in scipy's order
// (descending initialisation, swap-with-last removal) so ties resolve identically.  fp64 arithmetic,
// contraction off, same expression order => identical comparisons => identical indices.
#include <hip/hip_runtime.h>

#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr double kInf = __builtin_huge_val();

struct Best {
    double low;
    int first;     // first list position attaining `low`
    int last_un;   // last list position attaining `low` whose column is unassigned, or -1
};

__device__ __forceinline__ Best combine(const Best &a, const Best &b)
{
    if (b.low < a.low) return b;
    if (a.low < b.low) return a;
    Best r = a;
    r.first = min(a.first, b.first);
    r.last_un = max(a.last_un, b.last_un);
    return r;
}

// Wave-wide combine without LDS round trips: four DPP steps (quad_perm x2, row_half_mirror, row_mirror) leave every lane of
// a 16-lane row with its row's result, the four rows are merged through readlane.  (As six `__shfl_xor` rounds of four
// dwords this reduction was ds_bpermute-latency bound: ~1000 cycles of every Dijkstra step.)
template <int CTRL>
__device__ __forceinline__ Best dpp_best(const Best &x)
{
    const unsigned long long lb = __builtin_bit_cast(unsigned long long, x.low);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)lb, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(lb >> 32), CTRL, 0xF, 0xF, true);
    Best o;
    o.low = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    o.first = __builtin_amdgcn_update_dpp(0, x.first, CTRL, 0xF, 0xF, true);
    o.last_un = __builtin_amdgcn_update_dpp(0, x.last_un, CTRL, 0xF, 0xF, true);
    return o;
}

__device__ __forceinline__ Best lane_best(const Best &x, int l)
{
    const unsigned long long lb = __builtin_bit_cast(unsigned long long, x.low);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)lb, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(lb >> 32), l);
    Best o;
    o.low = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    o.first = __builtin_amdgcn_readlane(x.first, l);
    o.last_un = __builtin_amdgcn_readlane(x.last_un, l);
    return o;
}

__device__ __forceinline__ Best wave_best(Best x)
{
    x = combine(x, dpp_best<0xB1>(x));     // quad_perm [1,0,3,2]
    x = combine(x, dpp_best<0x4E>(x));     // quad_perm [2,3,0,1]
    x = combine(x, dpp_best<0x141>(x));    // row_half_mirror
    x = combine(x, dpp_best<0x140>(x));    // row_mirror: every lane holds its row's result
    const Best a = combine(lane_best(x, 0), lane_best(x, 16)), c = combine(lane_best(x, 32), lane_best(x, 48));
    return combine(a, c);
}

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline size_t lsap_bytes(int ncmax, int nrmax)
{
    return align16((size_t)ncmax * 8) * 2 + align16((size_t)nrmax * 8) + align16((size_t)ncmax * 4) * 3 +
           align16((size_t)nrmax * 4) + align16((size_t)(nrmax + 1) * 4);
}

// COST_LDS: the problem's cost block is copied into LDS (while it is checked for NaN / -inf) and the searches read
// it from there -- every Dijkstra step otherwise waits for a dependent batch of global loads.
template <bool USE_LDS, bool COST_LDS = false>
__global__ __launch_bounds__(64) void lsap_kernel(
    const float *__restrict__ cost, const int32_t *__restrict__ gt_offsets,
    const int64_t *__restrict__ gt_labels, int Q, int ncmax, int nrmax, int64_t *__restrict__ match_row,
    int64_t *__restrict__ match_col, int64_t *__restrict__ gt_inds, int64_t *__restrict__ labels,
    int32_t *__restrict__ status, char *__restrict__ ws, size_t ws_stride)
{
    extern __shared__ double smem_d[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int g0 = gt_offsets[b], G = gt_offsets[b + 1] - g0;

    // defaults of the scatter: every prediction background (hungarian_assigner.py:142 / :108-114)
    for (int q = lane; q < Q; q += 64) {
        if (gt_inds) gt_inds[(int64_t)b * Q + q] = 0;
        if (labels) labels[(int64_t)b * Q + q] = -1;
    }
    if (lane == 0) status[b] = 0;
    if (G <= 0 || Q <= 0) return;

    int po = 0;   // where this problem's (row, col) pairs start
    for (int bb = lane; bb < b; bb += 64) po += min(Q, gt_offsets[bb + 1] - gt_offsets[bb]);
    po = wave_sum_i(po);

    const bool tr = G < Q;                        // scipy solves the transposed problem when nc < nr
    const int nr = tr ? G : Q, nc = tr ? Q : G;
    const float *cb = cost + (int64_t)Q * g0;     // (G, Q) row-major block, element (q, g) at cb[g*Q + q]
    const int64_t si = tr ? Q : 1, sj = tr ? 1 : Q;   // solver (i, j) -> cb[i*si + j*sj]

    char *base = USE_LDS ? reinterpret_cast<char *>(smem_d) : ws + (size_t)b * ws_stride;
    double *v = reinterpret_cast<double *>(base);       base += align16((size_t)ncmax * 8);
    double *spc = reinterpret_cast<double *>(base);     base += align16((size_t)ncmax * 8);
    double *u = reinterpret_cast<double *>(base);       base += align16((size_t)nrmax * 8);
    int *path = reinterpret_cast<int *>(base);          base += align16((size_t)ncmax * 4);
    int *row4col = reinterpret_cast<int *>(base);       base += align16((size_t)ncmax * 4);
    int *remaining = reinterpret_cast<int *>(base);     base += align16((size_t)ncmax * 4);
    int *col4row = reinterpret_cast<int *>(base);       base += align16((size_t)nrmax * 4);
    int *scanned = reinterpret_cast<int *>(base);       base += align16((size_t)(nrmax + 1) * 4);
    float *cl = reinterpret_cast<float *>(base);        // COST_LDS only

    // scipy rejects NaN and -inf up front ("matrix contains invalid numeric entries")
    int bad = 0;
    {
        // one wavefront reads the whole block (900 x 15 floats = 54 KB for a DINO problem): 16-byte loads, eight in flight
        // per lane -- as a scalar loop this copy took longer than the solve (211 dependent-latency rounds)
        const int64_t total = (int64_t)nr * nc;
        auto check = [&](float c) { if (c != c || c == -__builtin_huge_valf()) bad = 1; };
        int64_t k = 0;
        if ((reinterpret_cast<uintptr_t>(cb) & 15) == 0) {
            const float4 *cb4 = reinterpret_cast<const float4 *>(cb);
            float4 *cl4 = reinterpret_cast<float4 *>(cl);
            const int64_t n4 = total / 4;
            for (int64_t k4 = lane; k4 < n4; k4 += 64 * 8) {
                float4 c[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) c[t] = k4 + 64 * t < n4 ? cb4[k4 + 64 * t] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    check(c[t].x); check(c[t].y); check(c[t].z); check(c[t].w);
                    if (COST_LDS && k4 + 64 * t < n4) cl4[k4 + 64 * t] = c[t];
                }
            }
            k = n4 * 4;
        }
        for (k += lane; k < total; k += 64) {
            const float c = cb[k];
            check(c);
            if (COST_LDS) cl[k] = c;
        }
    }
    if (__any(bad)) {
        if (lane == 0) status[b] = 2;
        return;
    }

    for (int j = lane; j < nc; j += 64) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    __syncthreads();

    for (int cur = 0; cur < nr; ++cur) {
        for (int j = lane; j < nc; j += 64) { spc[j] = kInf; remaining[j] = nc - j - 1; }
        __syncthreads();
        int num_rem = nc, ns = 0, sink = -1, i = cur;
        double min_val = 0.0;
        while (sink == -1) {
            const double ui = u[i];
            const float *crow = (COST_LDS ? cl : cb) + (int64_t)i * si;
            Best best = {kInf, 0x7fffffff, -1};
            // kScan list positions per lane and trip: the list-indirected reads (remaining -> cost, v, spc, row4col)
            // are independent across positions (distinct columns), so they are issued together instead of as dependent
            // chains -- a DINO problem (900 columns) is one trip of two LDS round trips; positions are still consumed in
            // increasing order (the tie rule needs that)
            constexpr int kScan = 16;
            for (int it0 = lane; it0 < num_rem; it0 += 64 * kScan) {
                int jj[kScan];
                double cc[kScan], vv[kScan], ss[kScan];
                int rr[kScan];
#pragma unroll
                for (int t = 0; t < kScan; ++t) jj[t] = it0 + 64 * t < num_rem ? remaining[it0 + 64 * t] : -1;
#pragma unroll
                for (int t = 0; t < kScan; ++t) {
                    const int j = jj[t] < 0 ? 0 : jj[t];
                    cc[t] = (double)crow[(int64_t)j * sj];
                    vv[t] = v[j];
                    ss[t] = spc[j];
                    rr[t] = row4col[j];
                }
#pragma unroll
                for (int t = 0; t < kScan; ++t) {
                    if (jj[t] < 0) break;
                    const int it = it0 + 64 * t, j = jj[t];
                    const double r = min_val + cc[t] - ui - vv[t];
                    double s = ss[t];
                    if (r < s) { path[j] = i; spc[j] = r; s = r; }
                    const bool un = rr[t] == -1;
                    if (s < best.low) { best.low = s; best.first = it; best.last_un = un ? it : -1; }
                    else if (s == best.low && un) best.last_un = it;
                }
            }
            best = wave_best(best);
            min_val = best.low;
            if (min_val == kInf) {                 // "cost matrix is infeasible"
                if (lane == 0) status[b] = 1;
                return;
            }
            const int idx = best.last_un >= 0 ? best.last_un : best.first;
            const int j = remaining[idx];
            const int r4c = row4col[j];
            __syncthreads();
            if (lane == 0) { scanned[ns] = j; remaining[idx] = remaining[num_rem - 1]; }
            ++ns; --num_rem;
            if (r4c == -1) sink = j; else i = r4c;
            __syncthreads();
        }
        // dual update: rows / columns touched by this search are exactly the scanned columns
        if (lane == 0) u[cur] += min_val;
        for (int t = lane; t < ns; t += 64) {
            const int j = scanned[t];
            const double d = min_val - spc[j];
            if (j != sink) u[row4col[j]] += d;
            v[j] -= d;
        }
        __syncthreads();
        if (lane == 0) {                           // augment along the alternating path
            int j = sink;
            for (;;) {
                const int ii = path[j];
                row4col[j] = ii;
                const int t = col4row[ii];
                col4row[ii] = j;
                j = t;
                if (ii == cur) break;
            }
        }
        __syncthreads();
    }

    // pairs in scipy's output order (rows ascending) + hungarian_assigner.py:142-147 scatter
    for (int i = lane; i < nr; i += 64) {
        const int c = col4row[i];
        int q, g, rank;
        if (tr) {
            q = c; g = i; rank = 0;
            for (int k = 0; k < nr; ++k) rank += col4row[k] < c;
        } else {
            q = i; g = c; rank = i;
        }
        if (match_row) match_row[po + rank] = q;
        if (match_col) match_col[po + rank] = g;
        if (gt_inds) gt_inds[(int64_t)b * Q + q] = g + 1;
        if (labels) labels[(int64_t)b * Q + q] = gt_labels[g0 + g];
    }
}

constexpr size_t kLdsLimit = 64 * 1024;
constexpr size_t kLdsBig = 150 * 1024;      // with the cost block; one problem per CU then, fine for a latency-bound solver

// Training targets from the assignment, all problems in one launch:
// dino_detr_ssod_head.py:1170-1205 / dino_detr_head.py:937-980 with PseudoSampler
// (thirdparty/mmdetection/mmdet/core/bbox/samplers/pseudo_sampler.py:35-41):
//   labels = num_classes (background) except labels[pos] = gt_labels[gt_inds[pos]-1]; label_weights = 1;
//   bbox_targets[pos] = cxcywh(gt_bboxes[gt_inds[pos]-1] / (w,h,w,h)); bbox_weights[pos] = 1; num_pos per problem.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void build_targets_kernel(
    const int64_t *__restrict__ gt_inds, const float *__restrict__ gt_bboxes, const int64_t *__restrict__ gt_labels,
    const int32_t *__restrict__ gt_offsets, const float *__restrict__ img_wh, int Q, int64_t num_classes,
    int64_t *__restrict__ labels, float *__restrict__ label_weights, float *__restrict__ bbox_targets,
    float *__restrict__ bbox_weights, int32_t *__restrict__ num_pos)
{
    const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    bool pos = false;
    if (q < Q) {
        const int64_t o = (int64_t)b * Q + q;
        const int64_t gi = gt_inds[o];
        pos = gi > 0;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t lab = num_classes;
        if (pos) {
            const int64_t g = gt_offsets[b] + gi - 1;
            const float4 bx = *reinterpret_cast<const float4 *>(gt_bboxes + g * 4);
            const float w = img_wh[2 * b], h = img_wh[2 * b + 1];
            const float x1 = bx.x / w, y1 = bx.y / h, x2 = bx.z / w, y2 = bx.w / h;
            t = make_float4((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1);
            lab = gt_labels[g];
        }
        labels[o] = lab;
        label_weights[o] = 1.f;
        *reinterpret_cast<float4 *>(bbox_targets + o * 4) = t;
        const float wgt = pos ? 1.f : 0.f;
        *reinterpret_cast<float4 *>(bbox_weights + o * 4) = make_float4(wgt, wgt, wgt, wgt);
    }
    const int n = __popcll(__ballot(pos));
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(&num_pos[b], n);
}

}  // namespace

extern "C" int64_t semidetr_lsap_workspace_bytes(int num_problems, int num_query, int max_gt)
{
    if (num_problems <= 0 || num_query <= 0 || max_gt <= 0) return 0;
    const int ncmax = num_query > max_gt ? num_query : max_gt;
    const int nrmax = num_query < max_gt ? num_query : max_gt;
    const size_t per = lsap_bytes(ncmax, nrmax);
    return per <= kLdsLimit ? 0 : (int64_t)(per * (size_t)num_problems);
}

extern "C" int semidetr_lsap_solve(void *stream, const float *cost, const int32_t *gt_offsets,
                                   const int64_t *gt_labels, int num_problems, int num_query, int total_gt,
                                   int max_gt, int64_t *match_row, int64_t *match_col,
                                   int64_t *assigned_gt_inds, int64_t *assigned_labels, int32_t *status,
                                   void *workspace)
{
    SEMIDETR_REQUIRE(num_problems >= 0 && num_query >= 0 && total_gt >= 0 && max_gt >= 0 && max_gt <= total_gt,
                     SEMIDETR_E_BADARG, "lsap: bad sizes (B=%d Q=%d sumG=%d maxG=%d)", num_problems, num_query,
                     total_gt, max_gt);
    if (num_problems == 0) return SEMIDETR_OK;
    SEMIDETR_REQUIRE(gt_offsets && status, SEMIDETR_E_BADARG, "lsap: null gt_offsets/status");
    SEMIDETR_REQUIRE(total_gt == 0 || num_query == 0 || cost, SEMIDETR_E_BADARG, "lsap: null cost");
    SEMIDETR_REQUIRE(!assigned_labels || total_gt == 0 || gt_labels, SEMIDETR_E_BADARG, "lsap: null gt_labels");
    const int ncmax = num_query > max_gt ? num_query : max_gt;
    const int nrmax = num_query < max_gt ? num_query : max_gt;
    const size_t per = lsap_bytes(ncmax > 0 ? ncmax : 1, nrmax > 0 ? nrmax : 1);
    hipStream_t st = semidetr::as_stream(stream);
    const size_t cost_bytes = (size_t)max_gt * (size_t)num_query * sizeof(float);
    if (per <= kLdsLimit && per + cost_bytes <= kLdsBig) {         // solver state AND the cost block in LDS
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lsap_kernel<true, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBig);
        if (e != hipSuccess) return semidetr::fail((int)e, "lsap: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL((lsap_kernel<true, true>), dim3(num_problems), dim3(64), per + cost_bytes, st, cost, gt_offsets,
                           gt_labels, num_query, ncmax, nrmax, match_row, match_col, assigned_gt_inds,
                           assigned_labels, status, (char *)nullptr, (size_t)0);
    } else if (per <= kLdsLimit) {
        hipLaunchKernelGGL(lsap_kernel<true>, dim3(num_problems), dim3(64), per, st, cost, gt_offsets,
                           gt_labels, num_query, ncmax, nrmax, match_row, match_col, assigned_gt_inds,
                           assigned_labels, status, (char *)nullptr, (size_t)0);
    } else {
        SEMIDETR_REQUIRE(workspace, SEMIDETR_E_BADARG, "lsap: workspace of %lld bytes required",
                         (long long)(per * (size_t)num_problems));
        hipLaunchKernelGGL(lsap_kernel<false>, dim3(num_problems), dim3(64), 0, st, cost, gt_offsets,
                           gt_labels, num_query, ncmax, nrmax, match_row, match_col, assigned_gt_inds,
                           assigned_labels, status, (char *)workspace, per);
    }
    return semidetr::launch_status("lsap_kernel");
}

extern "C" int semidetr_build_targets(void *stream, const int64_t *assigned_gt_inds, const float *gt_bboxes,
                                      const int64_t *gt_labels, const int32_t *gt_offsets, const float *img_wh,
                                      int num_problems, int num_query, int64_t num_classes, int64_t *labels,
                                      float *label_weights, float *bbox_targets, float *bbox_weights,
                                      int32_t *num_pos)
{
    SEMIDETR_REQUIRE(num_problems >= 0 && num_query >= 0, SEMIDETR_E_BADARG, "build_targets: negative sizes");
    if (num_problems == 0) return SEMIDETR_OK;
    SEMIDETR_REQUIRE(assigned_gt_inds && gt_offsets && img_wh && labels && label_weights && bbox_targets &&
                         bbox_weights && num_pos,
                     SEMIDETR_E_BADARG, "build_targets: null pointer argument");
    SEMIDETR_REQUIRE(num_problems <= 65535, SEMIDETR_E_TOOLARGE, "build_targets: more than 65535 problems");
    hipStream_t st = semidetr::as_stream(stream);
    hipError_t e = hipMemsetAsync(num_pos, 0, sizeof(int32_t) * (size_t)num_problems, st);
    if (e != hipSuccess) return semidetr::fail((int)e, "build_targets memset: %s", hipGetErrorString(e));
    if (num_query == 0) return SEMIDETR_OK;
    hipLaunchKernelGGL(build_targets_kernel, dim3((num_query + 255) / 256, num_problems), dim3(256), 0, st,
                       assigned_gt_inds, gt_bboxes, gt_labels, gt_offsets, img_wh, num_query, num_classes, labels,
                       label_weights, bbox_targets, bbox_weights, num_pos);
    return semidetr::launch_status("build_targets_kernel");
}
