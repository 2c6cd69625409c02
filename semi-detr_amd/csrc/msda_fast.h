// Fast-path device code of the multi-scale deformable attention for gfx950 (fp32, 32 channels per head):
// IO policies (reference op contract / fused MSDeformAttn prologue), forward, plain backward, and the
// gather + owner-computes scatter pair used for encoder self-attention.  Included by msda.hip only (inside
// its anonymous namespace); see msda.hip for the shared sample geometry and the host-side dispatch.
#pragma once

// ---------------------------------------------------------------------------------------------
// Where a fast-path kernel gets the (x, y, attention) of a sample from, and where the backward puts the two
// small gradients.  LocAttnIO is the reference op contract.  RawIO is the fused prologue/epilogue of
// MSDeformAttn.forward (detr_od/models/utils/ops/modules/ms_deform_attn.py:94-111): the kernels consume the
// reference points, the RAW sampling offsets and the RAW attention logits (the outputs of the two Linear
// layers) and do the softmax over L*P and the location arithmetic themselves, so the (N,Lq,M,L,P[,2])
// intermediates (and their backward) never touch HBM.
// ---------------------------------------------------------------------------------------------
// 1 / x as ONE v_rcp_f32 (1 ulp).  __frcp_rn is the correctly rounded reciprocal -- on gfx950 the full IEEE division
// sequence (v_div_scale / v_rcp / 4 fma / v_div_fmas / v_div_fixup), ten instructions where the fused prologue wants one.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Streaming (non-temporal) accesses for data that is read or written exactly ONCE per launch -- sampling locations, attention
// weights / logits, the op's outputs: the lines are not kept in the caches, so they stop displacing the value rows every
// workgroup re-reads.  Measured on the encoder forward at bs 4 (fused prologue): 234 -> 205 us from the output store alone.
// It is the `nt` bit that does it: `sc1` and `sc0 sc1` (write-through, the line dropped from L2) stores measure like plain ones
// (5.25 / 5.23 ms per step for the 24 launches), `nt` and `sc1 nt` 4.62 ms.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#ifndef SEMIDETR_NT_LOADS
#define SEMIDETR_NT_LOADS 0
#endif
__device__ __forceinline__ float ld_stream1(const float *p) { return SEMIDETR_NT_LOADS ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ float2 ld_stream2(const float *p)
{
    if (!SEMIDETR_NT_LOADS) return *reinterpret_cast<const float2 *>(p);
    const f32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x2_t *>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void st_stream4(float *p, const float4 v)
{
    f32x4_t t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4_t *>(p));
}
__device__ __forceinline__ void st_stream2(float *p, const float2 v)
{
    f32x2_t t;
    t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x2_t *>(p));
}
__device__ __forceinline__ void st_stream1(float *p, float v) { __builtin_nontemporal_store(v, p); }

// The padding mask of one (image, level) summarised (semidetr_msda_mask_extents, msda.hip): DETR's masks mark the band below /
// right of an image inside the batch canvas (dense_heads/dino_detr_head.py:305-318 -> F.interpolate per level), i.e. a pixel (y, x)
// is padding iff y >= vh or x >= vw.  With the two numbers a kernel tests a corner with two compares instead of a dependent
// byte load per corner (+15 % on every kernel, round 5); vh < 0: this level's mask is NOT of that form (or nobody summarised
// it) -- the corner's byte is read.  Either way the result is the mask's, bit for bit.
// One word per (image, level): vh | vw << 16, or -1 (levels beyond 32767 rows / columns are never summarised).
struct MaskExt {
    int ve;
    __device__ __forceinline__ bool summarised() const { return ve >= 0; }
    __device__ __forceinline__ int vh() const { return ve & 0xffff; }
    __device__ __forceinline__ int vw() const { return (int)((unsigned)ve >> 16); }
};

struct LocAttnIO {
    const float *loc, *attn;
    float *gloc, *gattn;
    static constexpr bool kSoftmax = false;      // attn already holds probabilities
    // The accessors take the row index (image, query, head) and the (image, query) index in the caller's integer type: int64_t over the
    // whole batch, or -- on the view of ONE image (pointers advanced, wave-uniform) -- unsigned indices inside the image, whose address
    // arithmetic is 32-bit (the 64-bit multiplies are ~40 half-rate VALU instructions per round of the lane-per-sample gather).
    __device__ __forceinline__ LocAttnIO image_view(int n, int Lq, int M_, int LP) const
    {
        const int64_t o = (int64_t)n * Lq * M_ * LP;
        return LocAttnIO{loc + o * 2, attn + o, gloc ? gloc + o * 2 : nullptr, gattn ? gattn + o : nullptr};
    }
    __device__ __forceinline__ bool masked(int n, int pixel) const { (void)n; (void)pixel; return false; }
    __host__ __device__ __forceinline__ bool has_mask() const { return false; }
    __device__ __forceinline__ MaskExt mask_ext(int n, int l) const { (void)n; (void)l; return MaskExt{-1}; }
    __device__ __forceinline__ void same_dims(int S_, int M_, int L_) const { (void)S_; (void)M_; (void)L_; }
    template <typename R>
    __device__ __forceinline__ void load_xy(R row, R nq, int LP, int k, int l, int P, int H, int W,
                                            float &x, float &y) const
    {
        (void)nq; (void)l; (void)P; (void)H; (void)W;
        const float2 xy = ld_stream2(loc + (row * LP + k) * 2);
        x = xy.x;
        y = xy.y;
    }
    // two-step form for callers that load a round ahead: what the loads return, and the arithmetic that turns it into (x, y)
    struct RawXY { float2 xy; };
    template <typename R>
    __device__ __forceinline__ RawXY load_xy_raw(R row, R nq, int LP, int k, int l) const
    {
        (void)nq; (void)l;
        return RawXY{ld_stream2(loc + (row * LP + k) * 2)};
    }
    __device__ __forceinline__ void finish_xy_raw(const RawXY &r, int P, int H, int W, float &x, float &y) const
    {
        (void)P; (void)H; (void)W;
        x = r.xy.x;
        y = r.xy.y;
    }
    // round 6, the window kernels' form (see RawIO): what the loads return; (x, y) from it and the level's reciprocals (unused here)
    struct RawXYc { float2 xy; };
    template <typename R>
    __device__ __forceinline__ RawXYc load_xy_c(R row, R nq, int LP, int k, int l) const
    {
        (void)nq; (void)l;
        return RawXYc{ld_stream2(loc + (row * LP + k) * 2)};
    }
    template <bool POINTS = false, typename R>
    __device__ __forceinline__ void finish_xy_c(const RawXYc &r, R nq, int l, int P, float invW, float invH, float &x, float &y) const
    {
        (void)nq; (void)l; (void)P; (void)invW; (void)invH;
        x = r.xy.x;
        y = r.xy.y;
    }
    // g_x, g_y = d/d (pixel x, pixel y) of the sample; the reference contract returns d/d (normalised location) = that x (W, H)
    template <bool POINTS = false, typename R>
    __device__ __forceinline__ void store_px(R row, R nq, int LP, int k, int l, int P, float Hf, float Wf, float g_a, float g_x,
                                             float g_y, float a, float dot) const
    {
        (void)nq; (void)l; (void)P; (void)a; (void)dot;
        st_stream1(gattn + row * LP + k, g_a);
        st_stream2(gloc + (row * LP + k) * 2, make_float2(g_x * Wf, g_y * Hf));
    }
    template <typename R>
    __device__ __forceinline__ float load_w(R row, int LP, int k) const { return ld_stream1(attn + row * LP + k); }
    // res = {d/d attn, d/d loc.x, d/d loc.y, attn} of sample k; row_res = the LP results of the same (n,q,m) row
    template <typename R>
    __device__ __forceinline__ void store(R row, R nq, int LP, int k, int l, int P, int H, int W,
                                          const float4 res, const float4 *row_res) const
    {
        (void)nq; (void)l; (void)P; (void)H; (void)W; (void)row_res;
        st_stream1(gattn + row * LP + k, res.x);
        st_stream2(gloc + (row * LP + k) * 2, make_float2(res.y, res.z));
    }
    // same, for callers that hold the row's sum_j a_j g_j already (unused here)
    template <typename R>
    __device__ __forceinline__ void store_with_dot(R row, R nq, int LP, int k, int l, int P, int H, int W,
                                                   const float4 res, float dot) const
    {
        (void)dot;
        store(row, nq, LP, k, l, P, H, W, res, nullptr);
    }
#if SEMIDETR_EXPERIMENTS
    // the same results piecewise, for the fused encoder backward experiment (msda_region.h), which produces a row's samples level by level:
    // gx / gy = d/d loc.x, d/d loc.y;  store_attn_partial is final here, finish_attn has nothing left to do
    __device__ __forceinline__ void store_xy(int64_t row, int64_t nq, int LP, int k, int l, int P, int H, int W, float gx,
                                             float gy) const
    {
        (void)nq; (void)l; (void)P; (void)H; (void)W;
        *reinterpret_cast<float2 *>(gloc + (row * LP + k) * 2) = make_float2(gx, gy);
    }
    __device__ __forceinline__ void store_attn_partial(int64_t row, int LP, int k, float g) const { gattn[row * LP + k] = g; }
    __device__ __forceinline__ void finish_attn(int64_t row, int LP, int k, float a, float dot) const
    {
        (void)row; (void)LP; (void)k; (void)a; (void)dot;
    }
#endif
};

__device__ __forceinline__ float rawio_group_sum(float x, int LP);      // = lp_group_sum, defined below (needs dpp_mov)

struct RawIO {
    const float *ref, *off, *logit;      // (N,Lq,L,ref_dim), (N,Lq,M,L,P,2), (N,Lq,M,L*P)
    float *goff, *glogit;
    int ref_dim, M, L;
    // input_padding_mask of MSDeformAttn.forward (ms_deform_attn.py:95-96: value.masked_fill(mask[..., None], 0)), folded in:
    // (N, S) bytes, nonzero = padding, or null.  A corner on a masked pixel is treated like a corner outside its level -- it
    // reads as zero and receives no gradient -- so the masked copy of `value` (one 91 MB read + write per encoder layer call
    // at bs 4) and the masking of grad_value never happen.
    const unsigned char *mask;
    int S;
    unsigned ref_bytes;                          // size of `ref` (N * Lq * L * ref_dim floats): bound of its buffer resource
    const int *mext;                             // (N, L) words per (image, level), see MaskExt; null: none (every corner reads its byte)
    static constexpr bool kSoftmax = true;       // load_w returns a raw logit; the kernel normalises the row
    // view of ONE image for 32-bit index arithmetic (see LocAttnIO::image_view); the mask and its summaries keep their image index
    __device__ __forceinline__ RawIO image_view(int n, int Lq, int M_, int LP) const
    {
        RawIO r = *this;
        const int64_t o = (int64_t)n * Lq * M_ * LP, ro = (int64_t)n * Lq * L * ref_dim;
        r.ref = ref + ro;
        r.off = off + o * 2;
        r.logit = logit + o;
        r.goff = goff ? goff + o * 2 : nullptr;
        r.glogit = glogit ? glogit + o : nullptr;
        r.ref_bytes = (unsigned)Lq * L * ref_dim * 4u;
        return r;
    }
    __device__ __forceinline__ bool masked(int n, int pixel) const { return mask[(int64_t)n * S + pixel] != 0; }
    __host__ __device__ __forceinline__ bool has_mask() const { return mask != nullptr; }
    // The kernels get S / M / L as arguments of their own: telling the compiler that the copies in here are the same numbers lets it
    // keep ONE scalar register per dimension (the fused-prologue kernels spill scalar registers as it is).
    __device__ __forceinline__ void same_dims(int S_, int M_, int L_) const
    {
        __builtin_assume(S == S_);
        __builtin_assume(M == M_);
        __builtin_assume(L == L_);
    }
    // (a 4-byte load whose address depends on (image, level) only: issue it beside the sample's own loads, not behind them)
    __device__ __forceinline__ MaskExt mask_ext(int n, int l) const
    {
        if (!mask || !mext) return MaskExt{-1};
        return MaskExt{mext[n * L + l]};
    }
    template <typename R>
    __device__ __forceinline__ void load_xy(R row, R nq, int LP, int k, int l, int P, int H, int W,
                                            float &x, float &y) const
    {
        // BRANCH-FREE on purpose.  With `if (ref_dim == 2) ... else ...` around the loads every call became its own
        // chain of basic blocks, and a kernel that calls load_xy for several samples "loads first" paid one global
        // round trip PER SAMPLE instead of one in total (instrumented forward: record phase 9.9 k cycles against 4.7 k
        // for the reference contract, the whole 10 % the fused prologue ran behind).  The reference point is one
        // 16-byte buffer load for both layouts -- for ref_dim 2 it also picks up the next (query, level)'s point, unused;
        // past the end of the tensor the buffer bound returns zeros -- and the two formulas differ by a uniform select.
        const float2 o = ld_stream2(off + (row * LP + k) * 2);
        const float4 r = buf_ld4(image_rsrc(ref, ref_bytes), (unsigned)((nq * L + l) * ref_dim) * 4u);
        finish_xy(o, r, P, H, W, x, y);
    }
    struct RawXY { float2 o; float4 r; };      // offset and reference point as loaded (see LocAttnIO::RawXY)
    template <typename R>
    __device__ __forceinline__ RawXY load_xy_raw(R row, R nq, int LP, int k, int l) const
    {
        return RawXY{ld_stream2(off + (row * LP + k) * 2), buf_ld4(image_rsrc(ref, ref_bytes), (unsigned)((nq * L + l) * ref_dim) * 4u)};
    }
    __device__ __forceinline__ void finish_xy_raw(const RawXY &r, int P, int H, int W, float &x, float &y) const
    {
        finish_xy(r.o, r.r, P, H, W, x, y);
    }
    __device__ __forceinline__ void finish_xy(const float2 o, const float4 r, int P, int H, int W, float &x, float &y) const
    {
        // x / W as x * rcp(W) (v_rcp_f32, 1 ulp): a true division is ~10 instructions per coordinate, and the location is
        // compared with the reference's to 1e-4, not bit for bit (the oracle-exact path is LocAttnIO)
        const bool box = ref_dim == 4;               // ms_deform_attn.py:106-108 (boxes) vs :102-105 (points)
        const float ip = 0.5f * fast_rcp((float)P);
        const float sx = box ? ip * r.z : fast_rcp((float)W), sy = box ? ip * r.w : fast_rcp((float)H);
        x = r.x + o.x * sx;
        y = r.y + o.y * sy;
    }
    // Round 6, the window kernels' form.  Their queries are pixels, whose reference points are 2-d in every DETR encoder
    // (transformer.py:675-691): the prefetched data is the offset and the 8-byte point -- not the 16-byte box, two registers per
    // sample held a round ahead -- and 1 / W, 1 / H come from the caller's per-level table (v_rcp_f32 of the float sizes, worked out
    // once per launch instead of by every lane in every round: a transcendental costs four plain instructions).  A box (ref_dim 4)
    // fetches its (w, h) where it is used: a dependent load, correct and slow, never taken by the reference's call sites.  Same
    // bits as finish_xy for both layouts.
    struct RawXYc { float2 o; float2 c; };
    template <typename R>
    __device__ __forceinline__ RawXYc load_xy_c(R row, R nq, int LP, int k, int l) const
    {
        return RawXYc{ld_stream2(off + (row * LP + k) * 2), buf_ld2(image_rsrc(ref, ref_bytes), (unsigned)((nq * L + l) * ref_dim) * 4u)};
    }
    // POINTS (round 6): the caller has established that ref_dim == 2 -- the box branch, a LOAD inside a rarely taken branch of a hot loop,
    // is compiled out (such a branch costs although it is never taken: every basic-block boundary around a load makes the waits for the
    // loads in flight conservative; msda_gw_d32 242 -> 232 us with its rare-case branches gone, the masked msda_rw_d32 190 -> 181)
    template <bool POINTS = false, typename R>
    __device__ __forceinline__ void finish_xy_c(const RawXYc &r, R nq, int l, int P, float invW, float invH, float &x, float &y) const
    {
        float sx = invW, sy = invH;
        if (!POINTS && ref_dim == 4) {      // (wave-uniform)
            const float2 wh = *reinterpret_cast<const float2 *>(ref + (nq * L + l) * ref_dim + 2);
            const float ip = 0.5f * fast_rcp((float)P);
            sx = ip * wh.x;
            sy = ip * wh.y;
        }
        x = r.c.x + r.o.x * sx;
        y = r.c.y + r.o.y * sy;
    }
    // g_x, g_y = d/d (pixel x, pixel y) of the sample.  Points: offsets are in pixels of the level (ms_deform_attn.py:102-105), so the
    // offset gradient IS the pixel gradient -- no x W / W round trip (store_with_dot multiplies by W and by v_rcp_f32(W))
    template <bool POINTS = false, typename R>
    __device__ __forceinline__ void store_px(R row, R nq, int LP, int k, int l, int P, float Hf, float Wf, float g_a, float g_x,
                                             float g_y, float a, float dot) const
    {
        st_stream1(glogit + row * LP + k, a * (g_a - dot));      // softmax backward: a_k * (g_k - sum_j a_j g_j)
        float2 g = make_float2(g_x, g_y);
        if (!POINTS && ref_dim != 2) {
            const float2 wh = *reinterpret_cast<const float2 *>(ref + (nq * L + l) * ref_dim + 2);
            const float ip = 0.5f * fast_rcp((float)P);
            g = make_float2(g_x * Wf * wh.x * ip, g_y * Hf * wh.y * ip);
        }
        st_stream2(goff + (row * LP + k) * 2, g);
    }
    template <typename R>
    __device__ __forceinline__ float load_w(R row, int LP, int k) const { return ld_stream1(logit + row * LP + k); }
    // called by the LP consecutive threads that own the row's samples (see row_softmax)
    template <typename R>
    __device__ __forceinline__ void store(R row, R nq, int LP, int k, int l, int P, int H, int W,
                                          const float4 res, const float4 *row_res) const
    {
        float dot = res.w * res.x;               // softmax backward: a_k * (g_k - sum_j a_j g_j)
        if ((LP & (LP - 1)) == 0 && LP <= 64) {
            dot = rawio_group_sum(dot, LP);
        } else {                                 // generic LP: every thread re-reads the row (rare)
            dot = 0.f;
            for (int j = 0; j < LP; ++j) dot += row_res[j].w * row_res[j].x;
        }
        store_with_dot(row, nq, LP, k, l, P, H, W, res, dot);
    }
    // dot = sum_j a_j g_j over the row (softmax backward), supplied by the caller
    template <typename R>
    __device__ __forceinline__ void store_with_dot(R row, R nq, int LP, int k, int l, int P, int H, int W,
                                                   const float4 res, float dot) const
    {
        st_stream1(glogit + row * LP + k, res.w * (res.x - dot));
        float2 g;
        if (ref_dim == 2) {
            g = make_float2(res.y * fast_rcp((float)W), res.z * fast_rcp((float)H));
        } else {
            const float2 wh = *reinterpret_cast<const float2 *>(ref + (nq * L + l) * ref_dim + 2);
            const float ip = 0.5f * fast_rcp((float)P);
            g = make_float2(res.y * wh.x * ip, res.z * wh.y * ip);
        }
        st_stream2(goff + (row * LP + k) * 2, g);
    }
#if SEMIDETR_EXPERIMENTS
    // piecewise form for the fused encoder backward experiment (see LocAttnIO): the offset gradient is final at once; d/d a_k is parked in
    // the logit gradient until the whole row's sum_j a_j g_j is known, then finish_attn applies the softmax backward in place
    __device__ __forceinline__ void store_xy(int64_t row, int64_t nq, int LP, int k, int l, int P, int H, int W, float gx,
                                             float gy) const
    {
        float2 g;
        if (ref_dim == 2) {
            g = make_float2(gx * fast_rcp((float)W), gy * fast_rcp((float)H));
        } else {
            const float2 wh = *reinterpret_cast<const float2 *>(ref + (nq * L + l) * ref_dim + 2);
            const float ip = 0.5f * fast_rcp((float)P);
            g = make_float2(gx * wh.x * ip, gy * wh.y * ip);
        }
        *reinterpret_cast<float2 *>(goff + (row * LP + k) * 2) = g;
    }
    __device__ __forceinline__ void store_attn_partial(int64_t row, int LP, int k, float g) const { glogit[row * LP + k] = g; }
    __device__ __forceinline__ void finish_attn(int64_t row, int LP, int k, float a, float dot) const
    {
        float *p = glogit + row * LP + k;
        *p = a * (*p - dot);
    }
#endif
};

// Padding mask for the kernels that LOAD / scatter through byte offsets (forward, strips backward, gather): corners whose
// pixel is masked become kOob.  (x, y) -> top-left pixel exactly as sample_setup_oob derives it.  `me` = io.mask_ext(n, level).
// BYTES = false (round 6): the caller has established that every level of the image is summarised -- the byte path is compiled out
template <typename IO, bool BYTES = true>
__device__ __forceinline__ void mask_corners_oob(const IO &io, const MaskExt me, int n, float x, float y, int H, int W, int st,
                                                 unsigned (&off)[4])
{
    if (!io.has_mask()) return;
    const int h0 = (int)floorf(sub_rn(mul_rn(y, (float)H), 0.5f)), w0 = (int)floorf(sub_rn(mul_rn(x, (float)W), 0.5f));
    if (me.summarised()) {      // the level's padding is the band below row vh / right of column vw
        const int vh = me.vh(), vw = me.vw();
        const bool y0 = h0 >= vh, y1 = h0 + 1 >= vh, x0 = w0 >= vw, x1 = w0 + 1 >= vw;
        off[0] = (y0 || x0) ? kOob : off[0];
        off[1] = (y0 || x1) ? kOob : off[1];
        off[2] = (y1 || x0) ? kOob : off[2];
        off[3] = (y1 || x1) ? kOob : off[3];
        return;
    }
    if constexpr (BYTES) {
        const int pix = st + h0 * W + w0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (off[c] != kOob && io.masked(n, pix + (c & 1) + (c >> 1) * W)) off[c] = kOob;
    }
}
// ... and for the scatter kernels, which hold level-local pixel indices (or -1) of the four corners of the cell at (h0, w0)
template <typename IO>
__device__ __forceinline__ void mask_corners_idx(const IO &io, const MaskExt me, int n, int h0, int w0, int W, int st_plus_base,
                                                 int (&off)[4])
{
    if (!io.has_mask()) return;
    if (me.summarised()) {
        const int vh = me.vh(), vw = me.vw();
        const bool y0 = h0 >= vh, y1 = h0 + 1 >= vh, x0 = w0 >= vw, x1 = w0 + 1 >= vw;
        off[0] = (y0 || x0) ? -1 : off[0];
        off[1] = (y0 || x1) ? -1 : off[1];
        off[2] = (y1 || x0) ? -1 : off[2];
        off[3] = (y1 || x1) ? -1 : off[3];
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (off[c] >= 0 && io.masked(n, st_plus_base + (c & 1) + (c >> 1) * W)) off[c] = -1;      // padded pixels receive no gradient
}

// ---------------------------------------------------------------------------------------------
// fast path: fp32, D == 32.  8 lanes x float4 per 128-byte value row.
// ---------------------------------------------------------------------------------------------
constexpr int kD = 32;

#if SEMIDETR_EXPERIMENTS
// debug / instrumentation counters of the experimental kernels (semidetr_debug_counters reads and resets them)
__device__ unsigned long long g_dest_dbg[16];
#define SEMIDETR_DBG_ADD(SLOT_, V_) atomicAdd(&g_dest_dbg[SLOT_], (unsigned long long)(V_))
#else
#define SEMIDETR_DBG_ADD(SLOT_, V_) ((void)0)
#endif

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// sum over the 8 lanes of a group: quad xor1, quad xor2, then mirror inside the half-row (i <-> 7-i).
__device__ __forceinline__ float group8_sum(float x)
{
    x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_mov<0x141>(x);  // row_half_mirror
    return x;
}

// Three 8-lane sums at once, as nine fused v_add_f32_dpp (dst = dpp(src) + src).  Written by hand because the compiler packs the
// adds of independent chains into v_pk_add_f32, which cannot take a DPP operand: every step then costs a v_mov_b32_dpp + an add
// + wait states (the unrolled gather: 18 DPP-related instructions and ~6 s_nop per sample).  Interleaving the three chains puts
// two independent instructions between a result and its next DPP read, which is exactly the hazard distance; the leading
// s_nop 1 covers the producers of a / b / c.
__device__ __forceinline__ void group8_sum3(float &a, float &b, float &c)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b), "+v"(c));
}

// Softmax over the L*P logits of one (n, q, m) row (ms_deform_attn.py:101), evaluated cooperatively by the LP
// consecutive threads that own the row's samples: one expf per sample, max / sum by xor-shuffles when LP is a
// power of two <= 64 (the DINO case LP = 16 is one DPP row); any other LP falls back to a per-thread loop.
__device__ __forceinline__ bool lp_shuffles(int LP) { return (LP & (LP - 1)) == 0 && LP <= 64; }
// sum / max over a lane-aligned group of LP threads (LP a power of two <= 64); every lane of the group gets the result
__device__ __forceinline__ float lp_group_sum(float x, int LP)
{
    if (LP >= 2) x += dpp_mov<0xB1>(x);      // quad_perm [1,0,3,2]
    if (LP >= 4) x += dpp_mov<0x4E>(x);      // quad_perm [2,3,0,1]
    if (LP >= 8) x += dpp_mov<0x141>(x);     // row_half_mirror
    if (LP >= 16) x += dpp_mov<0x140>(x);    // row_mirror
    if (LP >= 32) x += __shfl_xor(x, 16, 64);
    if (LP >= 64) x += __shfl_xor(x, 32, 64);
    return x;
}
__device__ __forceinline__ float rawio_group_sum(float x, int LP) { return lp_group_sum(x, LP); }
__device__ __forceinline__ float lp_group_max(float x, int LP)
{
    if (LP >= 2) x = fmaxf(x, dpp_mov<0xB1>(x));
    if (LP >= 4) x = fmaxf(x, dpp_mov<0x4E>(x));
    if (LP >= 8) x = fmaxf(x, dpp_mov<0x141>(x));
    if (LP >= 16) x = fmaxf(x, dpp_mov<0x140>(x));
    if (LP >= 32) x = fmaxf(x, __shfl_xor(x, 16, 64));
    if (LP >= 64) x = fmaxf(x, __shfl_xor(x, 32, 64));
    return x;
}

template <typename IO, typename R>
__device__ __forceinline__ float row_softmax(const IO &io, R row, int LP, int k, float raw)
{
    if (!IO::kSoftmax) return raw;
    if (lp_shuffles(LP)) {
        // reductions over the LP lane-aligned threads of the row: DPP inside a 16-lane row (quad xor 1, quad xor 2, half
        // mirror, row mirror -- VALU-rate), shuffles (LDS crossbar, ~100 clk each in a dependent chain) only beyond 16.
        // As eight __shfl_xor per softmax the fused prologue ran 14 % behind the reference-contract kernels.
        const float mx = lp_group_max(raw, LP);
        const float e = __expf(raw - mx);
        return e * fast_rcp(lp_group_sum(e, LP));
    }
    float mx = raw;
    for (int j = 0; j < LP; ++j) mx = fmaxf(mx, io.load_w(row, LP, j));
    float sum = 0.f;
    for (int j = 0; j < LP; ++j) sum += __expf(io.load_w(row, LP, j) - mx);
    (void)k;
    return __expf(raw - mx) * fast_rcp(sum);
}

// Two rows' softmax at once (the two samples a thread handles per trip of the record phase).  For L*P == 16 -- one DPP row per
// query row, the DINO case -- both chains sit in ONE basic block with compile-time step counts, so the scheduler interleaves
// them (a chain is 8 dependent DPP steps with two wait states each, an exp and a reciprocal).
template <typename IO>
__device__ __forceinline__ void row_softmax2(const IO &io, const int64_t (&rows)[2], int LP, const int (&kk)[2],
                                             const float (&raw)[2], float (&a)[2])
{
    if (IO::kSoftmax && LP == 16) {
        float m0 = raw[0], m1 = raw[1];
        m0 = fmaxf(m0, dpp_mov<0xB1>(m0));  m1 = fmaxf(m1, dpp_mov<0xB1>(m1));
        m0 = fmaxf(m0, dpp_mov<0x4E>(m0));  m1 = fmaxf(m1, dpp_mov<0x4E>(m1));
        m0 = fmaxf(m0, dpp_mov<0x141>(m0)); m1 = fmaxf(m1, dpp_mov<0x141>(m1));
        m0 = fmaxf(m0, dpp_mov<0x140>(m0)); m1 = fmaxf(m1, dpp_mov<0x140>(m1));
        const float e0 = __expf(raw[0] - m0), e1 = __expf(raw[1] - m1);
        float s0 = e0, s1 = e1;
        s0 += dpp_mov<0xB1>(s0);  s1 += dpp_mov<0xB1>(s1);
        s0 += dpp_mov<0x4E>(s0);  s1 += dpp_mov<0x4E>(s1);
        s0 += dpp_mov<0x141>(s0); s1 += dpp_mov<0x141>(s1);
        s0 += dpp_mov<0x140>(s0); s1 += dpp_mov<0x140>(s1);
        a[0] = e0 * fast_rcp(s0);
        a[1] = e1 * fast_rcp(s1);
        return;
    }
    a[0] = row_softmax(io, rows[0], LP, kk[0], raw[0]);
    a[1] = row_softmax(io, rows[1], LP, kk[1], raw[1]);
}

// The same softmax for an L*P that is not a power of two (five levels x four points = 20: the COCO-Full recipe), where the
// LP threads of a row do not line up with the wavefront's shuffle groups and the per-thread loop above costs LP loads + LP
// exponentials PER SAMPLE (2.3x on the whole forward).  Eight threads per query row instead: thread j takes logits j, j + 8,
// ..., the row maximum / sum are 3-step DPP reductions inside the octet, and the probabilities are parked in the .x of the
// row's (not yet written) records; the kernel's record loop picks them up after ONE extra barrier.
// rec: the float4 record array with LPP = LP + 1 entries per row; row_of(r) -> (n * Lq + q) * M + m or -1.
template <typename IO, typename RowOf>
__device__ __forceinline__ void softmax_rows_to_lds(const IO &io, int tid, int RPB, int LP, int LPP, float4 *rec, RowOf row_of,
                                                    int nthreads = 256)
{
    const int j = tid & 7;
    for (int r = tid >> 3; r < RPB; r += nthreads >> 3) {      // one trip for a 256-thread workgroup and 32 rows
    const int64_t row = row_of(r);
    float raw[8], mx = -__builtin_huge_valf();            // LP <= 64 (checked by the launcher for this path)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        raw[i] = -__builtin_huge_valf();
        if (row >= 0 && j + 8 * i < LP) raw[i] = io.load_w(row, LP, j + 8 * i);
        mx = fmaxf(mx, raw[i]);
    }
    mx = fmaxf(mx, dpp_mov<0xB1>(mx));
    mx = fmaxf(mx, dpp_mov<0x4E>(mx));
    mx = fmaxf(mx, dpp_mov<0x141>(mx));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        raw[i] = (row >= 0 && j + 8 * i < LP) ? __expf(raw[i] - mx) : 0.f;
        sum += raw[i];
    }
    sum = group8_sum(sum);
    const float inv = fast_rcp(sum);
    if (row >= 0)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (j + 8 * i < LP) rec[r * LPP + j + 8 * i].x = raw[i] * inv;
    }
}

// Workgroup -> (n, query tile, m).  Consecutive workgroups take consecutive heads, and the head of a given slot is
// rotated every kHeadRun tiles.  Why: rows of one head are M*128 bytes apart, so with M == 8 all addresses one head
// touches share bits [9:7]; workgroups go round robin to the 8 XCDs, so "blockIdx % 8 == head" pins each XCD to
// ONE such address class for the whole launch.  Measured at the encoder shape, bs 4 (rocprofv3 PMC,
// profiles/r01_msda_fwd_enc_mapping_pmc.txt): with the rotation the L1 traffic and L1 miss traffic are identical
// (91 M accesses, 15.4 M L1->L2 requests), L2 misses go UP 3.5 M -> 4.9 M, yet forward 286 -> 248 us, gather
// 341 -> 320 us, decoder forward 24.7 -> 20.8 us.  Run lengths 2..512 are equivalent in time (242-260 us; longer
// runs re-fetch fewer halo rows across XCDs: HBM reads 540 MB at 64, 418 MB at 256, 356 MB without rotation -- but
// the bench step was 1.5 % slower with 256 or a launch-size dependent length), 1 (adjacent tiles on different XCDs)
// gives 278, 2048 gives 268, >= 8192 is no rotation in practice.  Since a run of 512 tiles (one head per XCD at any
// instant, ~7 different heads per XCD over the launch) is as good as 64, what matters is that no XCD stays married
// to one address class: the classes are not equally fast from every XCD once the accesses leave the L2
// (tools/xcd_class_probe.hip: up to 1.5x between (XCD, class) pairs), every XCD has the same amount of work, and the
// launch ends with the slowest one.  Giving every XCD all 8 heads of a contiguous eighth of the
// tiles was as slow as no rotation (not understood); a head-major copy of the value map (N,M,S,D) brought nothing
// on top.  Speed only -- any bijection is correct.
constexpr int kHeadRun = 64;
struct Tile {
    int n, q0, m;
};
__device__ __forceinline__ Tile tile_of(int b, int M, int tiles_per_image, int rows_per_block);
__device__ __forceinline__ Tile tile_of_block(int M, int tiles_per_image, int rows_per_block)
{
    return tile_of((int)blockIdx.x, M, tiles_per_image, rows_per_block);
}
__device__ __forceinline__ Tile tile_of(int b, int M, int tiles_per_image, int rows_per_block)
{
    Tile t;
    const int r = b / M;
    t.m = (b % M + r / kHeadRun) % M;
    t.q0 = (r % tiles_per_image) * rows_per_block;
    t.n = r / tiles_per_image;
    return t;
}

// Encoder self-attention (num_query == spatial_size: query i IS pixel i of the multi-scale map): instead of
// 32 consecutive pixels (a 32 x 1 strip) a workgroup can take a PH x PW patch of one level -- its samples
// then land in a (PH + margin) x (PW + margin) neighbourhood on every level instead of a long thin one, which
// roughly halves the distinct value rows a workgroup pulls through its L1.  Patches are enumerated on the
// device because the level table lives in device memory; `tile` indexes them level by level.
struct Patch {
    int Hq, Wq, stq, y0, x0;     // level of the patch's queries and its top-left pixel; Hq == 0: no such patch
};
template <int PH, int PW>
__device__ __forceinline__ Patch find_patch(int tile, const int64_t *shapes, const int64_t *starts, int L)
{
    Patch p = {0, 0, 0, 0, 0};
    int acc = 0;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const int nx = (W + PW - 1) / PW, nt = ((H + PH - 1) / PH) * nx;
        if (tile < acc + nt) {
            p.Hq = H; p.Wq = W; p.stq = (int)starts[l];
            p.y0 = ((tile - acc) / nx) * PH; p.x0 = ((tile - acc) % nx) * PW;
            return p;
        }
        acc += nt;
    }
    return p;
}
template <int PW>
__device__ __forceinline__ int patch_query(const Patch &p, int r)      // r-th query of the patch or -1
{
    const int y = p.y0 + r / PW, x = p.x0 + r % PW;
    return (y < p.Hq && x < p.Wq) ? p.stq + y * p.Wq + x : -1;
}

// ---------------------------------------------------------------------------------------------
// How far the samples of an encoder self-attention launch reach -- the statistic the dispatcher picks the forward kernel by
// (msda.hip: launch_fast_forward).  The region-window kernel (msda_rw.h) serves the coarse levels' corners from LDS windows placed
// +- 5 px (five levels: +- 4 px) around a region: ~30 % faster than the patch kernel at sigma <= 2 px (162 against 234 us at bs 4 inside
// the step), level with it at sigma ~5.5 px = ~70 % of the samples further than 4 px away, slower beyond (the single source of
// these numbers: profiles/r04_region_window_dispatch.txt).  Both kernels therefore count, in a few sampled workgroups,
//     far   = samples on levels >= 1 more than kFarPx pixels (of the sampled level) from their query's own pixel centre
//     total = samples on levels >= 1
// into a device-side counter pair; the NEXT launch's first thread hands the finished pair to the host through mapped pinned
// memory (one plain 16-byte store: no PCIe atomics, no fence, no copy command, no synchronisation), where the dispatcher reads it
// when it gets there.  All pointers null: nothing is counted (fixed policy, stream capture, allocation failure).
// ---------------------------------------------------------------------------------------------
constexpr float kFarPx = 4.0f;
// ONE pointer and the launch's parity (round 6; it was four pointers: eight scalar registers held from the kernels' first instruction to their last, spilled at once by the
// fused-prologue instantiations): the call site's 16-word block in device memory --
//   words [0..3] / [4..7]  {far, total, kind, -} of even / odd launches: this launch accumulates into its parity's four, the other
//                          four are what the previous launch accumulated: published, then cleared
//   word  [8]              number of publications so far
//   words [10..11]         pointer to the slot's record in mapped host memory {sequence, far, total, kind} (written once by the host)
struct FwdStats {
    unsigned *blk;      // the slot's block (null: nothing is counted)
    unsigned parity;    // 0 / 1: which four words this launch accumulates into
    __device__ __forceinline__ bool on() const { return blk != nullptr; }
    __device__ __forceinline__ unsigned *base() const { return blk; }
    __device__ __forceinline__ unsigned *cur() const { return blk + 4 * parity; }
    __device__ __forceinline__ unsigned *prev() const { return blk + 4 * (1u - parity); }
};
__device__ __forceinline__ void fwd_stats_publish(const FwdStats &fs)
{
    unsigned *const prev = fs.prev(), *const seq = fs.base() + 8;
    const unsigned far = prev[0], total = prev[1], kind = prev[2];
    if (total) {
        // ONE 16-byte store, no fence: a system-scope release here writes back and invalidates this XCD's whole L2 (measured:
        // +10-15 us on a 270 us launch); the store reaches the host at the latest when this kernel ends, which is early enough
        // for a dispatcher that only ever looks at finished launches.  The host takes a record whose sequence number is new.
        const unsigned sq = seq[0] + 1u;
        seq[0] = sq;
        unsigned *const pub = *reinterpret_cast<unsigned *const *>(fs.base() + 10);
        *reinterpret_cast<uint4 *>(pub) = make_uint4(sq, far, total, kind);
    }
    prev[0] = 0u;
    prev[1] = 0u;
}
// one wave's counts -> the launch's counter pair
__device__ __forceinline__ void fwd_stats_add(const FwdStats &fs, unsigned far, unsigned total, unsigned kind)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        far += __shfl_xor(far, s, 64);
        total += __shfl_xor(total, s, 64);
    }
    // (lane number from the exec mask, not from threadIdx: the thread index would stay in a register from the kernel's first
    //  instruction to this one -- in the five-level region-window kernel that was one of three spilled registers)
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0 && total) {
        unsigned *const cur = fs.cur();
        atomicAdd(cur, far);
        atomicAdd(cur + 1, total);
        cur[2] = kind;
    }
}

// SPLIT = number of 8-lane groups that share one (q) row; each takes samples k = part, part+SPLIT, ...
template <int SPLIT, int UNROLL, int PATCH = 0, typename IO = LocAttnIO>
__global__ __launch_bounds__(256) void msda_fwd_d32(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const IO io, int S, int M, int L, int Lq, int P, int tiles_per_image, float *__restrict__ out,
    const FwdStats fs = FwdStats{nullptr, 0u})
{
    constexpr int RPB = 32 / SPLIT;   // query rows per workgroup
    extern __shared__ float4 smem[];
    const int LP = L * P, LPP = LP + 1;   // +1 record of padding: rows land on different LDS banks
    int4 *rec_off = reinterpret_cast<int4 *>(smem);
    float4 *rec_w = smem + RPB * LPP;

    Tile t = tile_of_block(M, tiles_per_image, RPB);
    const int rs = M * kD;
    constexpr int PH = PATCH / 100, PW = PATCH % 100;
    static_assert(PATCH == 0 || (PH * PW == RPB && SPLIT == 1), "a patch holds exactly the workgroup's rows");
    Patch pt = {0, 0, 0, 0, 0};
    bool sampled = false;                 // this workgroup counts how far its samples reach (FwdStats)
    if constexpr (PATCH != 0) {
        if (fs.on()) {
            sampled = (blockIdx.x & 255) == 1;
        }
    }
    // When L*P divides 256 a thread's samples s = tid, tid + 256, ... all have the same (level, point): its level's H, W,
    // start (three dependent int64 global loads per sample otherwise, in front of the coordinate arithmetic) are fetched once
    const bool fixed_k = (256 % LP) == 0;
    const int lf = ((int)threadIdx.x % LP) / P;
    const int Hf = fixed_k ? (int)shapes[2 * lf] : 0, Wf = fixed_k ? (int)shapes[2 * lf + 1] : 0, stf = fixed_k ? (int)starts[lf] : 0;
    const MaskExt mef = fixed_k ? io.mask_ext(t.n, lf) : MaskExt{-1};      // ... and the summary of the level's padding mask
    // PATCH: tiles_per_image is only a sizing hint for the grid -- a workgroup takes patches slot, slot + hint,
    // ... until the pyramid is exhausted, so any hint >= 1 is correct (the level table is device memory).
    for (int tile = t.q0 / RPB;; tile += tiles_per_image) {
#if SEMIDETR_EXPERIMENTS
    const unsigned long long tm0 = __builtin_readcyclecounter();
#endif
    if (PATCH) {
        pt = find_patch<PH ? PH : 1, PW ? PW : 1>(tile, shapes, starts, L);
        if (pt.Hq == 0) {
            // the launch's FIRST workgroup hands the previous launch's counts to the host when it is done: it started first,
            // so this is nowhere near the launch's tail (at its start the few dependent loads delayed its CU's whole queue)
            if (fs.on() && blockIdx.x == 0 && threadIdx.x == 0) fwd_stats_publish(fs);
            return;
        }
        __syncthreads();      // previous patch done with the LDS records
    }
#if SEMIDETR_EXPERIMENTS
    const unsigned long long tm1 = __builtin_readcyclecounter();
#endif
    auto query_of = [&](int r) { return PATCH ? patch_query<PW ? PW : 1>(pt, r) : (t.q0 + r < Lq ? t.q0 + r : -1); };

    // ---- phase 1: sample records -------------------------------------------------------------
    const bool sm_lds = IO::kSoftmax && !lp_shuffles(LP) && LP <= 64;      // see softmax_rows_to_lds
    if (sm_lds) {
        softmax_rows_to_lds(io, (int)threadIdx.x, RPB, LP, LPP, rec_w, [&](int r_) -> int64_t {
            const int q_ = query_of(r_);
            return q_ >= 0 ? ((int64_t)t.n * Lq + q_) * M + t.m : -1;
        });
        __syncthreads();
    }
    // Two samples per thread and trip, every global load of both issued before any arithmetic and without a branch around
    // them (an empty slot reads query 0 of the image): the loads' latencies overlap, and the two softmax reduction chains
    // of the fused prologue (8 dependent DPP steps + exp + rcp each) interleave instead of running back to back -- the
    // record phase is the serial section of a workgroup, everything it waits for is paid by the gather phase behind it.
    unsigned st_far = 0, st_total = 0;
    for (int s0 = threadIdx.x; s0 < RPB * LP; s0 += 512) {
        int rr[2], kk[2], qq[2];
        float x[2], y[2], raw[2];
        int64_t rows[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = s0 + 256 * u;
            rr[u] = min(s / LP, RPB - 1);
            kk[u] = s - (s / LP) * LP;
            qq[u] = s < RPB * LP ? query_of(rr[u]) : -1;
            const int l = kk[u] / P;
            const int H = fixed_k ? Hf : (int)shapes[2 * l], W = fixed_k ? Wf : (int)shapes[2 * l + 1];
            const int64_t nq = (int64_t)t.n * Lq + max(qq[u], 0);
            rows[u] = nq * M + t.m;
            io.load_xy(rows[u], nq, LP, kk[u], l, P, H, W, x[u], y[u]);
            raw[u] = sm_lds ? 0.f : io.load_w(rows[u], LP, kk[u]);
        }
        float a[2];
        if (sm_lds) {
            a[0] = rec_w[rr[0] * LPP + kk[0]].x;
            a[1] = rec_w[rr[1] * LPP + kk[1]].x;
        } else {
            row_softmax2(io, rows, LP, kk, raw, a);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (s0 + 256 * u >= RPB * LP) break;
            unsigned off[4] = {kOob, kOob, kOob, kOob};      // out of range -> the hardware returns zeros
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qq[u] >= 0) {
                const int l = kk[u] / P;
                const int H = fixed_k ? Hf : (int)shapes[2 * l], W = fixed_k ? Wf : (int)shapes[2 * l + 1];
                const int st = fixed_k ? stf : (int)starts[l];
                const MaskExt me = fixed_k ? mef : io.mask_ext(t.n, l);
                if (PATCH != 0 && sampled && l >= 1) {      // (workgroup-uniform: one workgroup in 256 pays for this)
                    constexpr int PW_ = PW ? PW : 1;
                    const float cx = ((float)(pt.x0 + rr[u] % PW_) + 0.5f) / (float)pt.Wq;
                    const float cy = ((float)(pt.y0 + rr[u] / PW_) + 0.5f) / (float)pt.Hq;
                    st_total += 1u;
                    st_far += (fabsf((x[u] - cx) * (float)W) > kFarPx || fabsf((y[u] - cy) * (float)H) > kFarPx) ? 1u : 0u;
                }
                float lw, lh;
                if (sample_setup_oob(x[u], y[u], H, W, st, (unsigned)rs * 4u, off, lw, lh)) {
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    w = make_float4(a[u] * (hh * hw), a[u] * (hh * lw), a[u] * (lh * hw), a[u] * (lh * lw));
                    mask_corners_oob(io, me, t.n, x[u], y[u], H, W, st, off);
                }
            }
            rec_off[rr[u] * LPP + kk[u]] = make_int4((int)off[0], (int)off[1], (int)off[2], (int)off[3]);
            rec_w[rr[u] * LPP + kk[u]] = w;
        }
    }
    if (PATCH != 0 && sampled) fwd_stats_add(fs, st_far, st_total, 1u);
    __syncthreads();
#if SEMIDETR_EXPERIMENTS
    const unsigned long long tm2 = __builtin_readcyclecounter();
#endif

    // ---- phase 2: gather + weighted sum ------------------------------------------------------
    const int g = threadIdx.x >> 3, j = threadIdx.x & 7;
    const int r = g / SPLIT, part = g % SPLIT;
    const int q = query_of(r);
    // value slice of image n as a raw buffer: out-of-range offsets (invalid corners) read as zero
    const __amdgpu_buffer_rsrc_t vr = image_rsrc(value + (int64_t)t.n * S * M * kD, (unsigned)S * M * kD * 4u);
    const unsigned lane_b = (unsigned)(t.m * kD + 4 * j) * 4u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int4 *ro = rec_off + r * LPP;
    const float4 *rw = rec_w + r * LPP;
#pragma unroll UNROLL
    for (int k = part; k < LP; k += SPLIT) {
        const int4 o = ro[k];
        const float4 w = rw[k];
        const float4 v1 = buf_ld4(vr, (unsigned)o.x + lane_b), v2 = buf_ld4(vr, (unsigned)o.y + lane_b);
        const float4 v3 = buf_ld4(vr, (unsigned)o.z + lane_b), v4 = buf_ld4(vr, (unsigned)o.w + lane_b);
        acc.x += w.x * v1.x + w.y * v2.x + w.z * v3.x + w.w * v4.x;
        acc.y += w.x * v1.y + w.y * v2.y + w.z * v3.y + w.w * v4.y;
        acc.z += w.x * v1.z + w.y * v2.z + w.z * v3.z + w.w * v4.z;
        acc.w += w.x * v1.w + w.y * v2.w + w.z * v3.w + w.w * v4.w;
    }
    if (SPLIT > 1) {
#pragma unroll
        for (int s = 8; s < 8 * SPLIT; s <<= 1) {
            acc.x += __shfl_xor(acc.x, s, 64);
            acc.y += __shfl_xor(acc.y, s, 64);
            acc.z += __shfl_xor(acc.z, s, 64);
            acc.w += __shfl_xor(acc.w, s, 64);
        }
    }
    if (part == 0 && q >= 0) {
        const int64_t row = ((int64_t)t.n * Lq + q) * M + t.m;
        st_stream4(out + row * kD + 4 * j, acc);
    }
#if SEMIDETR_EXPERIMENTS
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 5) {      // sampled (1 workgroup in 64: the counters are global atomics) --
        // wave 0's cycles per phase: 8 = patch search + previous patch's barrier, 9 = records, 10 = gather
        const unsigned long long tm3 = __builtin_readcyclecounter();
        SEMIDETR_DBG_ADD(8, tm1 - tm0);
        SEMIDETR_DBG_ADD(9, tm2 - tm1);
        SEMIDETR_DBG_ADD(10, tm3 - tm2);
        SEMIDETR_DBG_ADD(11, 1);
    }
#endif
    if (!PATCH) return;
    }
}

#if SEMIDETR_EXPERIMENTS
// ---------------------------------------------------------------------------------------------
// EXPERIMENT (forward variants 720-729): encoder self-attention forward, PRODUCER / CONSUMER form (queries are pixels, 4 x 8 patches as msda_fwd_d32<1,4,408>).
//
// Instrumented msda_fwd_d32 (wave-0 cycles per patch, bs 4): 950 waiting + 4600 in the record phase + 16 500 in the gather
// phase -- a fifth of a workgroup's life goes into a phase that issues no corner load, and most of THAT is the latency of the
// loads of the sampling locations / weights (HBM-cold, read once).  It cannot be prefetched by the waves that gather: vmcnt
// is an in-order counter, a load issued before the gather loop is waited for by the loop's first s_waitcnt.  So the record
// phase gets a wave of its own: a workgroup is 4 consumer waves + 1 producer wave and owns a SEQUENCE of patches (slot,
// slot + hint, ...; the grid is sized to the chip, ~4 workgroups per CU); while the consumers gather patch i from one half
// of the LDS records, the producer loads + computes the records of patch i + 1 into the other half; one barrier per patch.
// The consumers' loop is the one of msda_fwd_d32.
// ---------------------------------------------------------------------------------------------
constexpr int kWsThreads = 320;

template <typename IO>
__global__ __launch_bounds__(kWsThreads) void msda_fwd_d32_ws(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const IO io, int S, int M, int L, int Lq, int P, int hint, float *__restrict__ out)
{
    io.same_dims(S, M, L);
    constexpr int RPB = 32, PH = 4, PW = 8;
    extern __shared__ float4 smem[];
    const int LP = L * P, LPP = LP + 1;
    const int half_f4 = 2 * RPB * LPP;                 // one buffer: offsets + weights
    const Tile t = tile_of_block(M, hint, RPB);
    const int rs = M * kD;
    const int tid = threadIdx.x;
    const bool producer = tid >= 256;
    const int lane = tid & 63;
    // producer: sample s = lane + 64 u; for L*P dividing 64 its (level, point) is fixed
    const bool fixed_k = (64 % LP) == 0;
    const int lf = (lane % LP) / P;
    const int Hf = fixed_k ? (int)shapes[2 * lf] : 0, Wf = fixed_k ? (int)shapes[2 * lf + 1] : 0, stf = fixed_k ? (int)starts[lf] : 0;
    const MaskExt mef = fixed_k ? io.mask_ext(t.n, lf) : MaskExt{-1};      // ... and the summary of the level's padding mask
    const bool sm_lds = IO::kSoftmax && !lp_shuffles(LP) && LP <= 64;

    auto produce = [&](const Patch &pp, int buf) {
        int4 *rec_off = reinterpret_cast<int4 *>(smem + buf * half_f4);
        float4 *rec_w = smem + buf * half_f4 + RPB * LPP;
        auto query_of = [&](int r) { return patch_query<PW>(pp, r); };
        if (sm_lds) {      // L*P not a power of two: the row softmax by 8 threads per row, probabilities parked in rec_w.x
            softmax_rows_to_lds(io, lane, RPB, LP, LPP, rec_w, [&](int r_) -> int64_t {
                const int q_ = query_of(r_);
                return q_ >= 0 ? ((int64_t)t.n * Lq + q_) * M + t.m : -1;
            }, 64);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // one wave: its LDS writes are read back in order
            __builtin_amdgcn_wave_barrier();
        }
        for (int s0 = lane; s0 < RPB * LP; s0 += 128) {
            int rr[2], kk[2], qq[2];
            float x[2], y[2], raw[2];
            int64_t rows[2];
    #pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s = s0 + 64 * u;
                rr[u] = min(s / LP, RPB - 1);
                kk[u] = s - (s / LP) * LP;
                qq[u] = s < RPB * LP ? query_of(rr[u]) : -1;
                const int l = kk[u] / P;
                const int H = fixed_k ? Hf : (int)shapes[2 * l], W = fixed_k ? Wf : (int)shapes[2 * l + 1];
                const int64_t nq = (int64_t)t.n * Lq + max(qq[u], 0);
                rows[u] = nq * M + t.m;
                io.load_xy(rows[u], nq, LP, kk[u], l, P, H, W, x[u], y[u]);
                raw[u] = sm_lds ? 0.f : io.load_w(rows[u], LP, kk[u]);
            }
            float a[2];
            if (sm_lds) {
                a[0] = rec_w[rr[0] * LPP + kk[0]].x;
                a[1] = rec_w[rr[1] * LPP + kk[1]].x;
            } else {
                row_softmax2(io, rows, LP, kk, raw, a);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (s0 + 64 * u >= RPB * LP) break;
                unsigned off[4] = {kOob, kOob, kOob, kOob};
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (qq[u] >= 0) {
                    const int l = kk[u] / P;
                    const int H = fixed_k ? Hf : (int)shapes[2 * l], W = fixed_k ? Wf : (int)shapes[2 * l + 1];
                    const int st = fixed_k ? stf : (int)starts[l];
                    const MaskExt me = fixed_k ? mef : io.mask_ext(t.n, l);
                    float lw, lh;
                    if (sample_setup_oob(x[u], y[u], H, W, st, (unsigned)rs * 4u, off, lw, lh)) {
                        const float hh = 1.f - lh, hw = 1.f - lw;
                        w = make_float4(a[u] * (hh * hw), a[u] * (hh * lw), a[u] * (lh * hw), a[u] * (lh * lw));
                        mask_corners_oob(io, me, t.n, x[u], y[u], H, W, st, off);
                    }
                }
                rec_off[rr[u] * LPP + kk[u]] = make_int4((int)off[0], (int)off[1], (int)off[2], (int)off[3]);
                rec_w[rr[u] * LPP + kk[u]] = w;
            }
        }
    };

    int tile = t.q0 / RPB;
    Patch cur = find_patch<PH, PW>(tile, shapes, starts, L);
    if (cur.Hq == 0) return;
    if (producer) produce(cur, 0);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t vr = image_rsrc(value + (int64_t)t.n * S * M * kD, (unsigned)S * M * kD * 4u);
    for (int it = 0;; ++it) {
        const Patch nxt = find_patch<PH, PW>(tile + hint, shapes, starts, L);
        if (producer) {
            if (nxt.Hq != 0) produce(nxt, (it + 1) & 1);
        } else {
            const int r = tid >> 3, j = tid & 7;
            const int q = patch_query<PW>(cur, r);
            const unsigned lane_b = (unsigned)(t.m * kD + 4 * j) * 4u;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int4 *ro = reinterpret_cast<const int4 *>(smem + (it & 1) * half_f4) + r * LPP;
            const float4 *rw = smem + (it & 1) * half_f4 + RPB * LPP + r * LPP;
#pragma unroll 4
            for (int k = 0; k < LP; ++k) {
                const int4 o = ro[k];
                const float4 w = rw[k];
                const float4 v1 = buf_ld4(vr, (unsigned)o.x + lane_b), v2 = buf_ld4(vr, (unsigned)o.y + lane_b);
                const float4 v3 = buf_ld4(vr, (unsigned)o.z + lane_b), v4 = buf_ld4(vr, (unsigned)o.w + lane_b);
                acc.x += w.x * v1.x + w.y * v2.x + w.z * v3.x + w.w * v4.x;
                acc.y += w.x * v1.y + w.y * v2.y + w.z * v3.y + w.w * v4.y;
                acc.z += w.x * v1.z + w.y * v2.z + w.z * v3.z + w.w * v4.z;
                acc.w += w.x * v1.w + w.y * v2.w + w.z * v3.w + w.w * v4.w;
            }
            if (q >= 0) st_stream4(out + (((int64_t)t.n * Lq + q) * M + t.m) * kD + 4 * j, acc);
        }
        if (nxt.Hq == 0) return;
        __syncthreads();               // records of patch it + 1 complete, records of patch it free
        cur = nxt;
        tile += hint;
    }
}
#endif      // SEMIDETR_EXPERIMENTS

// Sum over the 32 lanes of each wavefront half (lane = channel).  After the five steps lanes 16..31 of
// each half hold the half's total; the writer is lane 16 / 48.
__device__ __forceinline__ float half32_sum(float x)
{
    x += dpp_mov<0xB1>(x);    // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);    // quad_perm [2,3,0,1]
    x += dpp_mov<0x141>(x);   // row_half_mirror
    x += dpp_mov<0x140>(x);   // row_mirror  -> every lane of a 16-lane row holds the row total
    // row_bcast:15 into rows 1 and 3 (row_mask 0xA): lane 15 of the previous row is added to every lane
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xA, 0xF, true));
    return x;
}

// Backward, fp32 / D == 32.  One value row (128 B) per 32 lanes, lane = channel: every global_atomic_add_f32
// wave-instruction updates two COMPLETE cache lines.  Measured on MI355X (tools/atomic_probe.hip): L2 fp32
// atomics cost ~one unit per 64-byte half-line touched (~20.8 G units/s chip-wide), so full-row updates move
// 4x more gradient per unit than the 8-lane x float4 layout the forward uses.
// RPB = query rows per 256-thread workgroup (8 half-waves, each walks RPB/8 rows).
// SCATTER_ONLY: grad_value only (no value reads, no channel reductions, no small gradients) -- the second half of the
// split backward, whose first half (msda_bwd_gather_d32) runs while grad_value is still being zero-filled.
template <int RPB, typename IO = LocAttnIO, bool SCATTER_ONLY = false>
__global__ __launch_bounds__(256) void msda_bwd_d32(
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int Lq, int P, int tiles_per_image,
    float *__restrict__ gvalue)
{
    io.same_dims(S, M, L);
    extern __shared__ float4 smem[];
    const int LP = L * P, LPP = LP + 1;
    int4 *rec_off = reinterpret_cast<int4 *>(smem);
    float4 *rec_p = smem + RPB * LPP;   // {lw, lh, a, level} ; overwritten with {g_attn, g_x, g_y, -}
    float *lev_w = reinterpret_cast<float *>(smem + 2 * RPB * LPP), *lev_h = lev_w + kMaxLevels;

    const Tile t = tile_of_block(M, tiles_per_image, RPB);
    const int rs = M * kD;
    if (threadIdx.x < L) {
        lev_h[threadIdx.x] = (float)shapes[2 * threadIdx.x];
        lev_w[threadIdx.x] = (float)shapes[2 * threadIdx.x + 1];
    }
    const bool sm_lds = IO::kSoftmax && !lp_shuffles(LP) && LP <= 64;      // see softmax_rows_to_lds
    if (sm_lds) {
        softmax_rows_to_lds(io, (int)threadIdx.x, RPB, LP, LPP, rec_p, [&](int r_) -> int64_t {
            return t.q0 + r_ < Lq ? ((int64_t)t.n * Lq + t.q0 + r_) * M + t.m : -1;
        });
        __syncthreads();
    }
    for (int s = threadIdx.x; s < RPB * LP; s += 256) {
        const int r = s / LP, k = s - r * LP;
        const int q = t.q0 + r;
        unsigned off[4] = {kOob, kOob, kOob, kOob};
        const int l = k / P;
        float4 pr = make_float4(0.f, 0.f, 0.f, __int_as_float(l));
        if (q < Lq) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)starts[l];
            const int64_t nq = (int64_t)t.n * Lq + q, row = nq * M + t.m;
            float x, y, lw, lh;
            io.load_xy(row, nq, LP, k, l, P, H, W, x, y);
            const MaskExt me = io.mask_ext(t.n, l);
            // kept for skipped samples too: the softmax backward of the fused epilogue needs every probability
            pr.z = sm_lds ? rec_p[r * LPP + k].x : row_softmax(io, row, LP, k, io.load_w(row, LP, k));
            if (sample_setup_oob(x, y, H, W, st, (unsigned)rs * 4u, off, lw, lh)) {
                pr.x = lw;
                pr.y = lh;
                mask_corners_oob(io, me, t.n, x, y, H, W, st, off);
            }
        }
        rec_off[r * LPP + k] = make_int4((int)off[0], (int)off[1], (int)off[2], (int)off[3]);
        rec_p[r * LPP + k] = pr;
    }
    __syncthreads();

    const int hw = threadIdx.x >> 5, c = threadIdx.x & 31;      // half-wave index, channel
    const __amdgpu_buffer_rsrc_t vr = image_rsrc(value + (int64_t)t.n * S * M * kD, (unsigned)S * M * kD * 4u);
    const unsigned lane_b = (unsigned)(t.m * kD + c) * 4u;
    float *gvb = gvalue + (int64_t)t.n * S * M * kD + t.m * kD + c;      // + corner byte offset / 4
    for (int r = hw; r < RPB; r += 8) {
        const int q = t.q0 + r;
        if (q >= Lq) break;
        const int64_t row = ((int64_t)t.n * Lq + q) * M + t.m;
        const float go = gout[row * kD + c];
        const int4 *ro = rec_off + r * LPP;
        float4 *rp = rec_p + r * LPP;
#pragma unroll 4
        for (int k = 0; k < LP; ++k) {
            const int4 o = ro[k];
            const float4 pr = rp[k];
            const float lw = pr.x, lh = pr.y, a = pr.z;
            const int l = __float_as_int(pr.w);
            const float hh = 1.f - lh, hwt = 1.f - lw;
            const float ga = go * a;
            if (SCATTER_ONLY) {
                if ((unsigned)o.x != kOob) fp_atomic_add(gvb + ((unsigned)o.x >> 2), hh * hwt * ga);
                if ((unsigned)o.y != kOob) fp_atomic_add(gvb + ((unsigned)o.y >> 2), hh * lw * ga);
                if ((unsigned)o.z != kOob) fp_atomic_add(gvb + ((unsigned)o.z >> 2), lh * hwt * ga);
                if ((unsigned)o.w != kOob) fp_atomic_add(gvb + ((unsigned)o.w >> 2), lh * lw * ga);
                continue;
            }
            // d_i = grad_out[c] * v_i[c]; corners outside the level read as zero (buffer bounds check)
            const float d1 = go * buf_ld1(vr, (unsigned)o.x + lane_b), d2 = go * buf_ld1(vr, (unsigned)o.y + lane_b);
            const float d3 = go * buf_ld1(vr, (unsigned)o.z + lane_b), d4 = go * buf_ld1(vr, (unsigned)o.w + lane_b);
            if ((unsigned)o.x != kOob) fp_atomic_add(gvb + ((unsigned)o.x >> 2), hh * hwt * ga);
            if ((unsigned)o.y != kOob) fp_atomic_add(gvb + ((unsigned)o.y >> 2), hh * lw * ga);
            if ((unsigned)o.z != kOob) fp_atomic_add(gvb + ((unsigned)o.z >> 2), lh * hwt * ga);
            if ((unsigned)o.w != kOob) fp_atomic_add(gvb + ((unsigned)o.w >> 2), lh * lw * ga);
            float pa = hh * hwt * d1 + hh * lw * d2 + lh * hwt * d3 + lh * lw * d4;
            float px = a * (hh * (d2 - d1) + lh * (d4 - d3));
            float py = a * (hwt * (d3 - d1) + lw * (d4 - d2));
            pa = half32_sum(pa);
            px = half32_sum(px);
            py = half32_sum(py);
            if (c == 16) rp[k] = make_float4(pa, lev_w[l] * px, lev_h[l] * py, a);
        }
    }
    if (SCATTER_ONLY) return;
    __syncthreads();

    // ---- coalesced write-back of grad_attn_weight / grad_sampling_loc --------------------------
    for (int s = threadIdx.x; s < RPB * LP; s += 256) {
        const int rr = s / LP, k = s - rr * LP;
        const int qq = t.q0 + rr;
        if (qq >= Lq) continue;
        const int64_t nq = (int64_t)t.n * Lq + qq, row = nq * M + t.m;
        const int l = k / P;
        io.store(row, nq, LP, k, l, P, (int)lev_h[l], (int)lev_w[l], rec_p[rr * LPP + k], rec_p + rr * LPP);
    }
}


// Gather half of the backward on its own (fp32, D == 32): grad_attn_weight and grad_sampling_loc for every
// sample, NO grad_value scatter.  Same tiling / LDS records / 8-lane x float4 loads as msda_fwd_d32<1>; the
// three channel sums per sample are 3 DPP steps inside the 8-lane group.  Streams like the forward (no
// atomics, no per-level barriers), used together with the owner-computes scatter kernel below.
// KLP = L * P at compile time (16: the DINO configuration) or 0.  With KLP the sample loop is fully unrolled, the
// three sums of sample k stay in the registers of lane k % 8 of the group (no LDS store inside the loop, so the
// compiler can keep 16 corner loads in flight like the forward does) and go to global memory straight from there.
// PATCH = PH * 100 + PW: the 32 queries of a workgroup are a patch of one level (num_query == spatial_size), as in
// the forward; 0: 32 consecutive queries.
// The body is a device function of (virtual block, thread in 0..255, LDS base, active) so that the merged backward
// launch (msda_bwd_lvl_merged) can run two of these per 512-thread workgroup next to its scatter workgroups; an
// inactive half (odd block count) walks through the same barriers without touching global memory.
// PAIR: two gather blocks share one 512-thread workgroup (merged launches): __syncthreads() is workgroup-wide, so both
// halves must take the same barriers -- a half that has run out of patches keeps walking, inactive, until the other is
// done too (decided by __syncthreads_or).
template <typename IO, int KLP, int PATCH, bool PAIR = false, int KB = 4>
__device__ __forceinline__ void gather_body(
    const int bid, const int tid, float4 *smem, const bool active, const float *__restrict__ gout,
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const IO io, int S, int M, int L, int Lq, int P, int tiles_per_image)
{
    io.same_dims(S, M, L);
    constexpr int RPB = 32;
    const int LP = L * P, LPP = LP + 1;
    int4 *rec_off = reinterpret_cast<int4 *>(smem);
    float4 *rec_p = smem + RPB * LPP;   // {lw, lh, a, level} ; overwritten with {g_attn, g_x, g_y, -}
    float *lev_w = reinterpret_cast<float *>(smem + 2 * RPB * LPP), *lev_h = lev_w + kMaxLevels;

    const Tile t = tile_of(bid, M, tiles_per_image, RPB);
    const int rs = M * kD;
    if (tid < L) {
        lev_h[tid] = (float)shapes[2 * tid];
        lev_w[tid] = (float)shapes[2 * tid + 1];
    }
    constexpr int PH = PATCH / 100, PW = PATCH % 100;
    static_assert(PATCH == 0 || PH * PW == RPB, "a patch holds exactly the workgroup's rows");
    Patch pt = {0, 0, 0, 0, 0};
    // level constants of this thread's sample slot, fetched once when L*P divides 256 (see msda_fwd_d32)
    const bool fixed_k = (256 % LP) == 0;
    const int lf = (tid % LP) / P;
    const int Hf = fixed_k ? (int)shapes[2 * lf] : 0, Wf = fixed_k ? (int)shapes[2 * lf + 1] : 0, stf = fixed_k ? (int)starts[lf] : 0;
    const MaskExt mef = fixed_k ? io.mask_ext((active ? t.n : 0), lf) : MaskExt{-1};      // ... and the summary of the level's padding mask
    // PATCH: tiles_per_image is a grid sizing hint; a workgroup takes patches slot, slot + hint, ... (see the forward)
    for (int tile = t.q0 / RPB;; tile += tiles_per_image) {
#if SEMIDETR_EXPERIMENTS
    const unsigned long long tg0 = __builtin_readcyclecounter();
#endif
    if (PATCH) {
        pt = find_patch<PH ? PH : 1, PW ? PW : 1>(tile, shapes, starts, L);
        if (PAIR) {
            if (!__syncthreads_or(active && pt.Hq != 0)) return;     // both halves are out of patches
        } else if (pt.Hq == 0) {
            return;
        }
        __syncthreads();      // previous patch done with the LDS records
    }
    const bool live = active && (!PATCH || pt.Hq != 0);
    auto query_of = [&](int rr_) { return !live ? -1 : (PATCH ? patch_query<PW ? PW : 1>(pt, rr_) : (t.q0 + rr_ < Lq ? t.q0 + rr_ : -1)); };
    // this lane's piece of the query's grad_out row, requested BEFORE the record phase: it is read once (HBM-cold), and
    // issued in front of the gather loop its latency sat, exposed, at the top of every patch's loop phase
    const int q = query_of(tid >> 3);
    float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q >= 0) go = *reinterpret_cast<const float4 *>(gout + (((int64_t)t.n * Lq + q) * M + t.m) * kD + 4 * (tid & 7));
    const bool sm_lds = IO::kSoftmax && !lp_shuffles(LP) && LP <= 64;      // see softmax_rows_to_lds
    if (sm_lds) {
        softmax_rows_to_lds(io, tid, RPB, LP, LPP, rec_p, [&](int r_) -> int64_t {
            const int q_ = query_of(r_);
            return q_ >= 0 ? ((int64_t)t.n * Lq + q_) * M + t.m : -1;
        });
        __syncthreads();
    }
    // two samples per thread and trip, all global loads first and unconditional (see msda_fwd_d32)
    for (int s0 = tid; s0 < RPB * LP; s0 += 512) {
        int rr[2], kk[2], qq[2];
        float x[2], y[2], raw[2];
        int64_t rows[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = s0 + 256 * u;
            rr[u] = min(s / LP, RPB - 1);
            kk[u] = s - (s / LP) * LP;
            qq[u] = s < RPB * LP ? query_of(rr[u]) : -1;
            const int l = kk[u] / P;
            const int H = fixed_k ? Hf : (int)shapes[2 * l], W = fixed_k ? Wf : (int)shapes[2 * l + 1];
            // an empty slot reads query 0 of the image; an INACTIVE half of a merged launch (block index past the last
            // tile: its image index is one past the batch) reads query 0 of image 0
            const int64_t nq = live ? (int64_t)t.n * Lq + max(qq[u], 0) : 0;
            rows[u] = nq * M + t.m;
            io.load_xy(rows[u], nq, LP, kk[u], l, P, H, W, x[u], y[u]);
            raw[u] = sm_lds ? 0.f : io.load_w(rows[u], LP, kk[u]);
        }
        float a[2];
        if (sm_lds) {
            a[0] = rec_p[rr[0] * LPP + kk[0]].x;
            a[1] = rec_p[rr[1] * LPP + kk[1]].x;
        } else {
            row_softmax2(io, rows, LP, kk, raw, a);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (s0 + 256 * u >= RPB * LP) break;
            unsigned off[4] = {kOob, kOob, kOob, kOob};
            const int l = kk[u] / P;
            float4 pr = make_float4(0.f, 0.f, 0.f, __int_as_float(l));
            if (qq[u] >= 0) {
                const int H = fixed_k ? Hf : (int)shapes[2 * l], W = fixed_k ? Wf : (int)shapes[2 * l + 1];
                const int st = fixed_k ? stf : (int)starts[l];
                const MaskExt me = fixed_k ? mef : io.mask_ext(t.n, l);
                float lw, lh;
                // kept for skipped samples too: the softmax backward of the fused epilogue needs every probability
                pr.z = a[u];
                if (sample_setup_oob(x[u], y[u], H, W, st, (unsigned)rs * 4u, off, lw, lh)) {
                    pr.x = lw;
                    pr.y = lh;
                    mask_corners_oob(io, me, t.n, x[u], y[u], H, W, st, off);
                }
            }
            rec_off[rr[u] * LPP + kk[u]] = make_int4((int)off[0], (int)off[1], (int)off[2], (int)off[3]);
            rec_p[rr[u] * LPP + kk[u]] = pr;
        }
    }
    __syncthreads();
#if SEMIDETR_EXPERIMENTS
    const unsigned long long tg1 = __builtin_readcyclecounter();
#endif

    const int r = tid >> 3, j = tid & 7;
    const __amdgpu_buffer_rsrc_t vr = image_rsrc(value + (int64_t)t.n * S * M * kD, (unsigned)S * M * kD * 4u);
    const unsigned lane_b = (unsigned)(t.m * kD + 4 * j) * 4u;
    const int4 *ro = rec_off + r * LPP;
    float4 *rp = rec_p + r * LPP;
    if constexpr (KLP > 0) {
        static_assert(KLP % 4 == 0 && KLP <= 32, "results are spread over the 8 lanes of a group");
        constexpr int NM = (KLP + 7) / 8;          // samples kept per lane (the last one only by lanes j < KLP % 8)
        constexpr int kB = KB % 100;               // samples per batch: 4 -> 16 corner loads in flight
        float4 mine[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) mine[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int P_ = KLP / L;                    // == P (checked by the launcher)
        if constexpr (kB == 16 || kB == 8) {
            constexpr int kAhead = kB / 4;             // samples in flight
            // ROLLING window: 16 corner loads in flight all the time.  The batched form below issues 16 loads, waits, and
            // computes four samples with nothing in flight behind them; here the four loads of sample s + 4 are issued as
            // soon as sample s's corners have been reduced to their dot products -- same 64 registers of load data.
            float4 buf[4 * kAhead];
#pragma unroll
            for (int s = 0; s < kAhead; ++s) {
                const int4 o = ro[s];
                buf[4 * s + 0] = buf_ld4(vr, (unsigned)o.x + lane_b);
                buf[4 * s + 1] = buf_ld4(vr, (unsigned)o.y + lane_b);
                buf[4 * s + 2] = buf_ld4(vr, (unsigned)o.z + lane_b);
                buf[4 * s + 3] = buf_ld4(vr, (unsigned)o.w + lane_b);
            }
#pragma unroll
            for (int k = 0; k < KLP; ++k) {
                const float4 pr = rp[k];
                const int4 on = ro[min(k + kAhead, KLP - 1)];
                const unsigned onc[4] = {(unsigned)on.x, (unsigned)on.y, (unsigned)on.z, (unsigned)on.w};
                float d[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 vv = buf[4 * (k % kAhead) + c];
                    d[c] = go.x * vv.x + go.y * vv.y + go.z * vv.z + go.w * vv.w;
                    if (k + kAhead < KLP) buf[4 * (k % kAhead) + c] = buf_ld4(vr, onc[c] + lane_b);
                }
                const float lw = pr.x, lh = pr.y, a = pr.z;
                const float hh = 1.f - lh, hw = 1.f - lw;
                float pa = hh * hw * d[0] + hh * lw * d[1] + lh * hw * d[2] + lh * lw * d[3];
                float px = a * (hh * (d[1] - d[0]) + lh * (d[3] - d[2]));
                float py = a * (hw * (d[2] - d[0]) + lw * (d[3] - d[1]));
                group8_sum3(pa, px, py);
                if ((k & 7) == j) {
                    asm volatile("");      // a real branch, see below
                    mine[k >> 3] = make_float4(pa, px, py, a);
                }
            }
        } else {
#pragma unroll
        for (int k0 = 0; k0 < KLP; k0 += kB) {
            int4 o[kB];
            float4 pr[kB], v[kB][4];
#pragma unroll
            for (int u = 0; u < kB; ++u) {
                o[u] = ro[k0 + u];
                pr[u] = rp[k0 + u];
            }
#pragma unroll
            for (int u = 0; u < kB; ++u) {
                v[u][0] = buf_ld4(vr, (unsigned)o[u].x + lane_b);
                v[u][1] = buf_ld4(vr, (unsigned)o[u].y + lane_b);
                v[u][2] = buf_ld4(vr, (unsigned)o[u].z + lane_b);
                v[u][3] = buf_ld4(vr, (unsigned)o[u].w + lane_b);
            }
#pragma unroll
            for (int u = 0; u < kB; ++u) {
                const int k = k0 + u;
                const float lw = pr[u].x, lh = pr[u].y, a = pr[u].z;
                const int l = __float_as_int(pr[u].w);
                const float hh = 1.f - lh, hw = 1.f - lw;
                auto dot4 = [&](const float4 &vv) { return go.x * vv.x + go.y * vv.y + go.z * vv.z + go.w * vv.w; };
                const float d1 = dot4(v[u][0]), d2 = dot4(v[u][1]), d3 = dot4(v[u][2]), d4 = dot4(v[u][3]);
                float pa = hh * hw * d1 + hh * lw * d2 + lh * hw * d3 + lh * lw * d4;
                float px = a * (hh * (d2 - d1) + lh * (d4 - d3));
                float py = a * (hw * (d3 - d1) + lw * (d4 - d2));
                group8_sum3(pa, px, py);
                (void)l;
                if ((k & 7) == j) {
                    // a real branch (the empty asm cannot be speculated): as selects, the scheduler hoists all 64 loads of
                    // the unrolled loop to the top (256 VGPRs, one wave per SIMD) or, capped at 128 VGPRs, spills
                    asm volatile("");
                    mine[k >> 3] = make_float4(pa, px, py, a);
                }
            }
        }
        }
        // the W / H factors of grad_sampling_loc once per kept sample, after the loop: an LDS read inside the per-sample
        // branch put an `s_waitcnt lgkmcnt(0)` on every sample of the unrolled loop
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int l = min((j + 8 * i) / P_, L - 1);
            mine[i].y *= lev_w[l];
            mine[i].z *= lev_h[l];
        }
#if SEMIDETR_EXPERIMENTS
        const unsigned long long tg2 = __builtin_readcyclecounter();
#endif
        float dot = 0.f;                           // fused epilogue: sum_k a_k g_k over the row
        if (IO::kSoftmax) {
#pragma unroll
            for (int i = 0; i < NM; ++i) dot += (j + 8 * i < KLP) ? mine[i].w * mine[i].x : 0.f;
            dot = group8_sum(dot);
        }
        // KB >= 100: timing aid, results are computed but (practically) never stored
        if (q >= 0 && (KB < 100 || mine[0].x == 1.2345e30f)) {
            const int64_t nq = (int64_t)t.n * Lq + q, row = nq * M + t.m;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int k = j + 8 * i, l = min(k / P_, L - 1);
                if (k < KLP) io.store_with_dot(row, nq, LP, k, l, P_, (int)lev_h[l], (int)lev_w[l], mine[i], dot);
            }
        }
#if SEMIDETR_EXPERIMENTS
        if (tid == 0 && (bid & 63) == 5) {      // sampled wave-0 cycles: 12 = records (+ wait), 13 = gather loop, 14 = stores
            const unsigned long long tg3 = __builtin_readcyclecounter();
            SEMIDETR_DBG_ADD(12, tg1 - tg0);
            SEMIDETR_DBG_ADD(13, tg2 - tg1);
            SEMIDETR_DBG_ADD(14, tg3 - tg2);
            SEMIDETR_DBG_ADD(15, 1);
        }
#endif
        if (!PATCH) return;
        continue;
    }
    // Batches of kGU samples, unrolled by hand: the result store into rec_p would otherwise keep the compiler
    // from hoisting the next samples' record reads / corner loads above it (4 * kGU loads in flight per lane; measured on MI355X at the encoder shape, bs 4: kGU 1 / 2 / 4 -> 370 / 389 / 375 us, so 1).
    constexpr int kGU = 1;
    for (int k0 = 0; k0 < LP; k0 += kGU) {
        int4 o[kGU];
        float4 pr[kGU], v[kGU][4];
#pragma unroll
        for (int u = 0; u < kGU; ++u) {
            const int k = min(k0 + u, LP - 1);
            o[u] = ro[k];
            pr[u] = rp[k];
        }
#pragma unroll
        for (int u = 0; u < kGU; ++u) {      // corners outside the level read as zero (buffer bounds check)
            v[u][0] = buf_ld4(vr, (unsigned)o[u].x + lane_b);
            v[u][1] = buf_ld4(vr, (unsigned)o[u].y + lane_b);
            v[u][2] = buf_ld4(vr, (unsigned)o[u].z + lane_b);
            v[u][3] = buf_ld4(vr, (unsigned)o[u].w + lane_b);
        }
#pragma unroll
        for (int u = 0; u < kGU; ++u) {
            if (k0 + u >= LP) break;
            const float lw = pr[u].x, lh = pr[u].y, a = pr[u].z;
            const int l = __float_as_int(pr[u].w);
            const float hh = 1.f - lh, hw = 1.f - lw;
            // d_i = <grad_out, v_i> over this lane's 4 channels
            auto dot4 = [&](const float4 &vv) { return go.x * vv.x + go.y * vv.y + go.z * vv.z + go.w * vv.w; };
            const float d1 = dot4(v[u][0]), d2 = dot4(v[u][1]), d3 = dot4(v[u][2]), d4 = dot4(v[u][3]);
            float pa = hh * hw * d1 + hh * lw * d2 + lh * hw * d3 + lh * lw * d4;
            float px = a * (hh * (d2 - d1) + lh * (d4 - d3));
            float py = a * (hw * (d3 - d1) + lw * (d4 - d2));
            pa = group8_sum(pa);
            px = group8_sum(px);
            py = group8_sum(py);
            if (j == 0) rp[k0 + u] = make_float4(pa, lev_w[l] * px, lev_h[l] * py, a);
        }
    }
    __syncthreads();
    for (int s = tid; s < RPB * LP; s += 256) {
        const int rr = s / LP, k = s - rr * LP;
        const int qq = query_of(rr);
        if (qq < 0) continue;
        const int64_t nq = (int64_t)t.n * Lq + qq, row = nq * M + t.m;
        const int l = k / P;
        io.store(row, nq, LP, k, l, P, (int)lev_h[l], (int)lev_w[l], rec_p[rr * LPP + k], rec_p + rr * LPP);
    }
    if (!PATCH) return;
    }
}

template <typename IO = LocAttnIO, int KLP = 0, int PATCH = 0, int WPE = 4, int KB = 4>
__global__ __launch_bounds__(256, WPE) void msda_bwd_gather_d32(
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int Lq, int P, int tiles_per_image,
    float4 *__restrict__ zero = nullptr, int64_t zero_n4 = 0)
{
    io.same_dims(S, M, L);
    extern __shared__ float4 smem[];
    // optional side job: zero-fill `zero_n4` float4s (grad_value, which the scatter launch that FOLLOWS accumulates into) --
    // every workgroup clears one contiguous slice with fire-and-forget stores before its gather work
    if (zero) {
        const int64_t per = (zero_n4 + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < zero_n4 ? lo + per : zero_n4;
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256)
            st_stream4(reinterpret_cast<float *>(zero + i), make_float4(0.f, 0.f, 0.f, 0.f));
    }
    gather_body<IO, KLP, PATCH, false, KB>((int)blockIdx.x, (int)threadIdx.x, smem, true, gout, value, shapes, starts, io, S, M, L, Lq, P,
                                tiles_per_image);
}

#if SEMIDETR_EXPERIMENTS
// ---------------------------------------------------------------------------------------------
// EXPERIMENT (backward variant 920): the gather for encoder self-attention with FOUR lanes per query (8 channels each)
// and 8 x 8 query patches.  Parity-green, 310 us against 305 us: HALF the vector instructions per query changed nothing.
//
// msda_bwd_gather_d32 (8 lanes per query, 4 channels each) is VALU-bound, not load-bound: ~1300 vector instructions per
// wavefront and 32-query patch at 4 cycles each = 5.2 k cycles per SIMD and patch against 4.1 k cycles of corner loads per CU
// (instrumented: gather loop 11.9 k of a patch's 17.9 k wave cycles, 67 % of the load path).  What does not shrink with the
// channels per lane is everything AROUND the dot products -- the linear combinations that turn the four corner dots into
// the three results, the cross-lane sums, the record reads, the address adds, the keep-my-sample select: ~40 of the ~60
// vector instructions per sample.  With 8 channels per lane one pass of that overhead serves 16 queries of a wavefront
// instead of 8 (3.6 vector instructions per query and sample instead of 7.9), the cross-lane sum is two quad_perm steps,
// and the same 256 threads take an 8 x 8 patch, whose samples share more value rows (fewer L1 misses per read).
// Loads in flight per lane as before (2 samples x 4 corners x 2 halves = 16 x 16 bytes).
// ---------------------------------------------------------------------------------------------
typedef float v2f_t __attribute__((ext_vector_type(2)));

// three sums over the 4 lanes of a quad at once: six fused v_add_f32_dpp, the chains interleaved (see group8_sum3)
__device__ __forceinline__ void quad_sum3(float &a, float &b, float &c)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b), "+v"(c));
}

template <typename IO, int KLP>
__global__ __launch_bounds__(256, 4) void msda_bwd_gather4_d32(
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int Lq, int P, int tiles_per_image,
    float4 *__restrict__ zero, int64_t zero_n4)
{
    io.same_dims(S, M, L);
    static_assert(KLP % 4 == 0 && KLP <= 32, "results are spread over the 4 lanes of a quad");
    constexpr int RPB = 64, PH = 8, PW = 8, LP = KLP, LPP = LP + 1, NM = KLP / 4;
    extern __shared__ float4 smem[];
    int4 *rec_off = reinterpret_cast<int4 *>(smem);
    float4 *rec_p = smem + RPB * LPP;   // {lw, lh, a, -}
    float *lev_w = reinterpret_cast<float *>(smem + 2 * RPB * LPP), *lev_h = lev_w + kMaxLevels;
    const int tid = threadIdx.x;
    if (zero) {      // side job: clear one slice of grad_value (the scatter launch that follows accumulates into it)
        const int64_t per = (zero_n4 + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < zero_n4 ? lo + per : zero_n4;
        for (int64_t i = lo + tid; i < hi; i += 256)
            st_stream4(reinterpret_cast<float *>(zero + i), make_float4(0.f, 0.f, 0.f, 0.f));
    }
    const Tile t = tile_of_block(M, tiles_per_image, RPB);
    const int rs = M * kD;
    if (tid < L) {
        lev_h[tid] = (float)shapes[2 * tid];
        lev_w[tid] = (float)shapes[2 * tid + 1];
    }
    // 256 % LP == 0 for LP = 16; for LP = 20 a thread's (level, point) changes from sample to sample
    const bool fixed_k = (256 % LP) == 0;
    const int lf = (tid % LP) / P;
    const int Hf = fixed_k ? (int)shapes[2 * lf] : 0, Wf = fixed_k ? (int)shapes[2 * lf + 1] : 0, stf = fixed_k ? (int)starts[lf] : 0;
    const MaskExt mef = fixed_k ? io.mask_ext(t.n, lf) : MaskExt{-1};      // ... and the summary of the level's padding mask
    const __amdgpu_buffer_rsrc_t vr = image_rsrc(value + (int64_t)t.n * S * M * kD, (unsigned)S * M * kD * 4u);
    for (int tile = t.q0 / RPB;; tile += tiles_per_image) {
        const Patch pt = find_patch<PH, PW>(tile, shapes, starts, L);
        if (pt.Hq == 0) return;
        __syncthreads();      // previous patch done with the LDS records (first trip: lev_w / lev_h written)
        auto query_of = [&](int rr_) { return patch_query<PW>(pt, rr_); };
        const bool sm_lds = IO::kSoftmax && !lp_shuffles(LP);      // see softmax_rows_to_lds
        if (sm_lds) {
            softmax_rows_to_lds(io, tid, RPB, LP, LPP, rec_p, [&](int r_) -> int64_t {
                const int q_ = query_of(r_);
                return q_ >= 0 ? ((int64_t)t.n * Lq + q_) * M + t.m : -1;
            });
            __syncthreads();
        }
        // ---- records: two samples per thread and trip, every global load first and unconditional (see msda_fwd_d32)
        for (int s0 = tid; s0 < RPB * LP; s0 += 512) {
            int rr[2], kk[2], qq[2];
            float x[2], y[2], raw[2];
            int64_t rows[2];
    #pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s = s0 + 256 * u;
                rr[u] = min(s / LP, RPB - 1);
                kk[u] = s - (s / LP) * LP;
                qq[u] = s < RPB * LP ? query_of(rr[u]) : -1;
                const int l = kk[u] / P;
                const int H = fixed_k ? Hf : (int)shapes[2 * l], W = fixed_k ? Wf : (int)shapes[2 * l + 1];
                const int64_t nq = (int64_t)t.n * Lq + max(qq[u], 0);
                rows[u] = nq * M + t.m;
                io.load_xy(rows[u], nq, LP, kk[u], l, P, H, W, x[u], y[u]);
                raw[u] = sm_lds ? 0.f : io.load_w(rows[u], LP, kk[u]);
            }
            float a[2];
            if (sm_lds) {
                a[0] = rec_p[rr[0] * LPP + kk[0]].x;
                a[1] = rec_p[rr[1] * LPP + kk[1]].x;
            } else {
                row_softmax2(io, rows, LP, kk, raw, a);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (s0 + 256 * u >= RPB * LP) break;
                unsigned off[4] = {kOob, kOob, kOob, kOob};
                float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
                if (qq[u] >= 0) {
                    const int l = kk[u] / P;
                    const int H = fixed_k ? Hf : (int)shapes[2 * l], W = fixed_k ? Wf : (int)shapes[2 * l + 1];
                    const int st = fixed_k ? stf : (int)starts[l];
                    const MaskExt me = fixed_k ? mef : io.mask_ext(t.n, l);
                    float lw, lh;
                    pr.z = a[u];      // kept for skipped samples too: the softmax backward needs every probability
                    if (sample_setup_oob(x[u], y[u], H, W, st, (unsigned)rs * 4u, off, lw, lh)) {
                        pr.x = lw;
                        pr.y = lh;
                        mask_corners_oob(io, me, t.n, x[u], y[u], H, W, st, off);
                    }
                }
                rec_off[rr[u] * LPP + kk[u]] = make_int4((int)off[0], (int)off[1], (int)off[2], (int)off[3]);
                rec_p[rr[u] * LPP + kk[u]] = pr;
            }
        }
        __syncthreads();

        // ---- gather: quad = query, lane j = channels 4j .. 4j+3 and 16+4j .. 16+4j+3
        const int r = tid >> 2, j = tid & 3;
        const int q = query_of(r);
        const unsigned lane_b = (unsigned)(t.m * kD + 4 * j) * 4u;
        v2f_t g0 = {0.f, 0.f}, g1 = g0, g2 = g0, g3 = g0;
        if (q >= 0) {
            const float *gp = gout + (((int64_t)t.n * Lq + q) * M + t.m) * kD + 4 * j;
            const float4 lo4 = *reinterpret_cast<const float4 *>(gp), hi4 = *reinterpret_cast<const float4 *>(gp + 16);
            g0 = v2f_t{lo4.x, lo4.y}; g1 = v2f_t{lo4.z, lo4.w}; g2 = v2f_t{hi4.x, hi4.y}; g3 = v2f_t{hi4.z, hi4.w};
        }
        const int4 *ro = rec_off + r * LPP;
        const float4 *rp = rec_p + r * LPP;
        float4 mine[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) mine[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int kB = 2;                      // samples per batch: 2 x 4 corners x 2 halves = 16 loads in flight
#pragma unroll
        for (int k0 = 0; k0 < KLP; k0 += kB) {
            int4 o[kB];
            float4 pr[kB], v[kB][4][2];
#pragma unroll
            for (int u = 0; u < kB; ++u) {
                o[u] = ro[k0 + u];
                pr[u] = rp[k0 + u];
            }
#pragma unroll
            for (int u = 0; u < kB; ++u) {
                const unsigned oc[4] = {(unsigned)o[u].x, (unsigned)o[u].y, (unsigned)o[u].z, (unsigned)o[u].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[u][c][0] = buf_ld4(vr, oc[c] + lane_b);
                    v[u][c][1] = buf_ld4(vr, oc[c] + lane_b + 64u);
                }
            }
#pragma unroll
            for (int u = 0; u < kB; ++u) {
                const int k = k0 + u;
                const float lw = pr[u].x, lh = pr[u].y, a = pr[u].z;
                const float hh = 1.f - lh, hw = 1.f - lw;
                // d_c = <grad_out, corner c> over this lane's 8 channels, as packed pairs
                float d[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 va = v[u][c][0], vb = v[u][c][1];
                    v2f_t tt = g0 * v2f_t{va.x, va.y};
                    tt += g1 * v2f_t{va.z, va.w};
                    tt += g2 * v2f_t{vb.x, vb.y};
                    tt += g3 * v2f_t{vb.z, vb.w};
                    d[c] = tt.x + tt.y;
                }
                const v2f_t d13 = {d[0], d[2]}, d24 = {d[1], d[3]};
                const v2f_t tw = hw * d13 + lw * d24;              // (hw d1 + lw d2, hw d3 + lw d4)
                const v2f_t ab = d24 - d13;                        // (d2 - d1, d4 - d3)
                float pa = hh * tw.x + lh * tw.y;
                float px = a * (hh * ab.x + lh * ab.y);
                float py = a * (hw * (d[2] - d[0]) + lw * (d[3] - d[1]));
                quad_sum3(pa, px, py);
                if ((k & 3) == j) {
                    asm volatile("");      // a real branch: as selects the scheduler hoists every load of the unrolled loop
                    mine[k >> 2] = make_float4(pa, px, py, a);
                }
            }
        }
        const int P_ = KLP / L;                    // == P (checked by the launcher)
#pragma unroll
        for (int i = 0; i < NM; ++i) {             // the W / H factors of grad_sampling_loc, once per kept sample
            const int l = min((j + 4 * i) / P_, L - 1);
            mine[i].y *= lev_w[l];
            mine[i].z *= lev_h[l];
        }
        float dot = 0.f;                           // fused epilogue: sum_k a_k g_k over the row
        if (IO::kSoftmax) {
#pragma unroll
            for (int i = 0; i < NM; ++i) dot += mine[i].w * mine[i].x;
            dot += dpp_mov<0xB1>(dot);
            dot += dpp_mov<0x4E>(dot);
        }
        if (q >= 0) {
            const int64_t nq = (int64_t)t.n * Lq + q, row = nq * M + t.m;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int k = j + 4 * i, l = min(k / P_, L - 1);
                io.store_with_dot(row, nq, LP, k, l, P_, (int)lev_h[l], (int)lev_w[l], mine[i], dot);
            }
        }
    }
}
#endif      // SEMIDETR_EXPERIMENTS

// (the windowed scatter msda_bwd_scatter_d32_win, the resident-level forward msda_fwd_d32_res, the 512-thread merged
//  backward and the cooperative-fill launch -- measured and rejected -- live in msda_fast_experiments.h, compiled only
//  into the experiments library)
constexpr int kWinThreads = 512;                        // 8 wavefronts = 32 streams of 16 lanes (region scatter, msda_region.h)
constexpr int kPT = 4;                                  // num_point (compile time)
constexpr int kScatterHeadRun = 16;                     // head rotation, see tile_of_block

// ---------------------------------------------------------------------------------------------
// grad_value for ARBITRARY query sets (decoder cross-attention, the BASELINE micro-benchmark shape), fp32, D == 32,
// with the coarse levels pre-aggregated in LDS.  Runs after msda_bwd_gather_d32 (which produces the two small gradients).
//
// The plain backward issues one full-row atomic per (sample, corner) and sits exactly on the atomic unit's ceiling
// (10.4 G rows/s: 307 k rows = 29.5 us at the micro-benchmark shape, 2.25 M rows = 216 us for the bs-4 decoder).  But
// the queries of one (image, head) throw Lq * P samples at EVERY level, and the coarse levels are small: 300 queries x 4
// points x 4 corners = 4800 contributions land on the 273 rows of a 13 x 21 level.  So a workgroup takes (image, head,
// level, chunk of <= 256 queries), stages the chunk's grad_out rows in LDS, and
//   * level with <= kLvlRows pixels: buckets the (sample, corner) pairs by target row with integer LDS atomics (count ->
//     scan -> fill, as the windowed kernel does, the "window" being the whole level), lets 32 streams of 16 lanes walk
//     equal shares of the row-sorted entries with the row sum in registers, and issues ONE full-line atomic per row run;
//   * larger level: the same walk over the unsorted entries, every entry flushed on its own (= the plain kernel).
// Rows flushed, micro-benchmark shape: 19.2 k -> ~9 k per (image, head).
// ---------------------------------------------------------------------------------------------
constexpr int kLvlThreads = 512;
constexpr int kLvlQ = 256;               // queries per workgroup
constexpr int kLvlRows = 4352;           // largest level that is bucketed (counters: 2 x 17 KB of LDS)

struct NoWait {
    __device__ __forceinline__ void operator()() const {}
};

// `before_atomics` runs (all threads) right before the first row atomic: the cooperative-fill launch waits there for the
// zero fill of grad_value (msda_bwd_lvl_coop)
// `chunks_b` / `chunk_q_b` (0 = as `chunks` / `chunk_q`): the chunking of the BUCKETED levels.  A bucketed level flushes fewer
// rows the more queries share a workgroup, a level too large to bucket aggregates nothing and only wants parallelism; the
// grid has `chunks` slots per (image, level, head), a bucketed level uses the first `chunks_b` of them.
template <typename IO, typename Wait = NoWait, int NT = kLvlThreads, int LQ = kLvlQ>
__device__ __forceinline__ void lvl_scatter_body(
    int b, float4 *smem, const float *__restrict__ gout, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int Lq, int P, int chunks, int chunk_q,
    float *__restrict__ gvalue, const Wait before_atomics = Wait(), int chunks_b = 0, int chunk_q_b = 0)
{
    io.same_dims(S, M, L);
    constexpr int kLvlQ = LQ;            // (shadows the file-scope default)
    constexpr int kStreams = NT / 16;
    // layout: gtile [kLvlQ * 32 floats] | entries [emax float2] | cnt [kLvlRows] | start [kLvlRows]
    float *gtile = reinterpret_cast<float *>(smem);
    float2 *entries = reinterpret_cast<float2 *>(gtile + kLvlQ * kD);
    const int emax = max(chunk_q, chunk_q_b) * P * 4;
    int *cnt = reinterpret_cast<int *>(entries + emax + 8);
    int *start = cnt + kLvlRows;
    __shared__ int wsum[NT / 64], total_s;

    const int LP = L * P, rs = M * kD;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hw = tid >> 5, c = tid & 31;
    // block -> (chunk, level, head, image): heads fastest (XCD = head), levels next so that the four levels of a chunk
    // (different amounts of work) are spread over the launch
    const int m = b % M; b /= M;
    const int l = b % L; b /= L;
    const int ch = b % chunks, n = b / chunks;
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)starts[l];
    const MaskExt me = io.mask_ext(n, l);      // (workgroup-uniform: one image, one level)
    const int R = H * W;
    const bool bucket = R <= kLvlRows;
    if (bucket && chunks_b > 0) {
        if (ch >= chunks_b) return;
        chunk_q = chunk_q_b;
    }
    const int q0 = ch * chunk_q, nq = min(chunk_q, Lq - q0);
    if (nq <= 0) return;

    {   // grad_out rows of the chunk -> LDS, channels (c, c+16) interleaved; all loads of a thread are issued before its
        // stores (a load -> store loop would expose the global latency once per row)
        constexpr int kPass = kLvlQ * 8 / NT;               // float4 pieces per thread
        float4 v[kPass];
        const float4 *src = reinterpret_cast<const float4 *>(gout + (((int64_t)n * Lq + q0) * M + m) * kD);
#pragma unroll
        for (int ps = 0; ps < kPass; ++ps) {
            const int r = (tid >> 3) + ps * (NT / 8);
            v[ps] = r < nq ? src[(int64_t)r * (rs / 4) + (tid & 7)] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int ps = 0; ps < kPass; ++ps) {
            const int r = (tid >> 3) + ps * (NT / 8), c0 = 4 * (tid & 7);
            if (r < nq) {
                float *dst = gtile + r * kD;
                dst[((c0 + 0) & 15) * 2 + ((c0 + 0) >> 4)] = v[ps].x;
                dst[((c0 + 1) & 15) * 2 + ((c0 + 1) >> 4)] = v[ps].y;
                dst[((c0 + 2) & 15) * 2 + ((c0 + 2) >> 4)] = v[ps].z;
                dst[((c0 + 3) & 15) * 2 + ((c0 + 3) >> 4)] = v[ps].w;
            }
        }
    }
    if (bucket)
        for (int k = tid; k < R; k += NT) cnt[k] = 0;
    if (tid == 0) total_s = 0;

    // ---- geometry of this thread's samples (query i, point p), sample index tid + sp * NT: every global load first,
    //      then the arithmetic (so the SPT samples' latencies overlap instead of adding up)
    constexpr int SPT = (kLvlQ * 8 + NT - 1) / NT;           // up to 8 points per query
    float cw[SPT][4];
    int crow[SPT][4], rank[SPT][4], qi[SPT];
    const int nsamp = nq * P;
    float sx[SPT], sy[SPT], sa[SPT];
#pragma unroll
    for (int sp = 0; sp < SPT; ++sp) {
        const int sidx = min(tid + sp * NT, nsamp - 1);      // clamped: the loads stay unconditional
        const int i = sidx / P, p = sidx - i * P;
        const int64_t nqi = (int64_t)n * Lq + q0 + i, row = nqi * M + m;
        const int k = l * P + p;
        io.load_xy(row, nqi, LP, k, l, P, H, W, sx[sp], sy[sp]);
        sa[sp] = io.load_w(row, LP, k);
        if (IO::kSoftmax) {
            // __expf and v_rcp_f32 exactly as row_softmax (forward, gather) and the region scatter evaluate the same row:
            // the kernels of one op must agree on the attention weights
            constexpr int kLv = 8;       // levels held in registers
            float mx, sum = 0.f;
            if (P == 4 && L <= kLv) {
                // the four points of a (query, head) row sit in one quad of threads (sidx = 4 i + p, quads never straddle the
                // clamp): every thread loads its point's logit on each LEVEL -- L loads and exponentials instead of L * P of
                // each per sample (the fused-prologue decoder backward ran 12.6 % behind the reference-contract one) -- and the
                // row's max / sum are two DPP steps inside the quad, as in the region scatter
                float lg[kLv];
#pragma unroll
                for (int l2 = 0; l2 < kLv; ++l2) lg[l2] = l2 < L ? io.load_w(row, LP, l2 * P + p) : -__builtin_huge_valf();
                mx = lg[0];
#pragma unroll
                for (int l2 = 1; l2 < kLv; ++l2) mx = fmaxf(mx, lg[l2]);
                mx = fmaxf(mx, dpp_mov<0xB1>(mx));
                mx = fmaxf(mx, dpp_mov<0x4E>(mx));
#pragma unroll
                for (int l2 = 0; l2 < kLv; ++l2) sum += l2 < L ? __expf(lg[l2] - mx) : 0.f;
                sum += dpp_mov<0xB1>(sum);
                sum += dpp_mov<0x4E>(sum);
            } else {
                mx = sa[sp];
                for (int j = 0; j < LP; ++j) mx = fmaxf(mx, io.load_w(row, LP, j));
                for (int j = 0; j < LP; ++j) sum += __expf(io.load_w(row, LP, j) - mx);
            }
            sa[sp] = __expf(sa[sp] - mx) * fast_rcp(sum);
        }
    }
#pragma unroll
    for (int sp = 0; sp < SPT; ++sp) {
        const int sidx = tid + sp * NT;
        qi[sp] = -1;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) { crow[sp][ci] = -1; cw[sp][ci] = 0.f; rank[sp][ci] = 0; }
        int off[4];
        float lw, lh;
        if (sidx >= nsamp || !sample_setup(sx[sp], sy[sp], H, W, 0, 1, off, lw, lh)) continue;   // off = level-local pixel or -1
        if (io.has_mask()) {      // padded pixels receive no gradient
            const int h0 = (int)floorf(sub_rn(mul_rn(sy[sp], (float)H), 0.5f)), w0 = (int)floorf(sub_rn(mul_rn(sx[sp], (float)W), 0.5f));
            mask_corners_idx(io, me, n, h0, w0, W, st + h0 * W + w0, off);
        }
        const float a = sa[sp];
        const float hh = 1.f - lh, hwt = 1.f - lw;
        cw[sp][0] = hh * hwt * a; cw[sp][1] = hh * lw * a; cw[sp][2] = lh * hwt * a; cw[sp][3] = lh * lw * a;
        qi[sp] = sidx / P;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) crow[sp][ci] = off[ci];
    }
    __syncthreads();                     // counters zeroed, grad_out staged
    // ---- position of every (sample, corner) entry
    if (bucket) {
#pragma unroll
        for (int sp = 0; sp < SPT; ++sp)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
                if (crow[sp][ci] >= 0) rank[sp][ci] = atomicAdd(&cnt[crow[sp][ci]], 1);
        __syncthreads();
        // exclusive scan of R counters: thread t owns counters [t * K, t * K + K)
        const int K = (R + NT - 1) / NT;
        int local = 0;
        for (int k = tid * K; k < min(R, tid * K + K); ++k) local += cnt[k];
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int base = 0;
        for (int w2 = 0; w2 < wv; ++w2) base += wsum[w2];
        int run = base + incl - local;
        for (int k = tid * K; k < min(R, tid * K + K); ++k) { start[k] = run; run += cnt[k]; }
        if (tid == NT - 1) total_s = run;
        __syncthreads();
#pragma unroll
        for (int sp = 0; sp < SPT; ++sp)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                const int r = crow[sp][ci];
                if (r >= 0)
                    entries[start[r] + rank[sp][ci]] = make_float2(
                        cw[sp][ci], __int_as_float((rank[sp][ci] == cnt[r] - 1 ? (1 << 30) : 0) | (r << 10) | qi[sp]));
            }
    } else {        // level too large to bucket: unsorted list, every entry is the last of its row
#pragma unroll
        for (int sp = 0; sp < SPT; ++sp) {
            int nv = 0;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) nv += crow[sp][ci] >= 0;
            // wave-aggregated append
            int incl = nv;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(incl, d, 64);
                if (lane >= d) incl += t;
            }
            int wbase = 0;
            if (lane == 63 && incl) wbase = atomicAdd(&total_s, incl);
            wbase = __shfl(wbase, 63, 64);
            int pos = wbase + incl - nv;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
                if (crow[sp][ci] >= 0)
                    entries[pos++] = make_float2(cw[sp][ci], __int_as_float((1 << 30) | (crow[sp][ci] << 10) | qi[sp]));
        }
    }
    __syncthreads();
    before_atomics();
    // ---- walk: 32 streams of 16 lanes (lane = channels l16 and l16 + 16), equal shares, one atomic pair per row run
    {
        const int sid = tid >> 4, l16 = tid & 15;
        const float2 *gt2 = reinterpret_cast<const float2 *>(gtile);
        float *gvs = gvalue + (((int64_t)n * S + st) * M + m) * kD + l16;
        const int total = total_s;
        int lo = (int)((int64_t)total * sid / kStreams), hi = (int)((int64_t)total * (sid + 1) / kStreams);
        // EXCLUSIVE level: this workgroup holds ALL queries of the (image, head) for a bucketed level (one chunk), so nobody
        // else contributes to the level's rows -- a finished row is STORED over the zero the fill left there, no atomic.
        // That needs every row summed by ONE stream: the shares are moved forward to the next row boundary.  (Several chunks
        // taken in turn by one workgroup with plain load-add-store was measured: decoder bs 4 / Lq 1100 239 us against 157,
        // bs 1 169 against 44 -- the chain gets three times as long and every flush waits for its load.)
        bool excl = bucket && chunks_b == 1;
        if (excl) {      // ... and nobody else's level may alias these rows (a level table whose levels overlap is legal input:
            // the atomic form adds both levels' contributions correctly, stores would lose one)
            for (int l2 = 0; l2 < L; ++l2) {
                const int s2 = (int)starts[l2], r2 = (int)shapes[2 * l2] * (int)shapes[2 * l2 + 1];
                if (l2 != l && s2 < st + R && st < s2 + r2) excl = false;
            }
        }
        if (excl) {
            auto aligned = [&](int x) {
                if (x <= 0) return 0;
                while (x < total && !(__float_as_int(entries[x - 1].y) & (1 << 30))) ++x;
                return min(x, total);
            };
            lo = aligned(lo);
            hi = aligned(hi);
        }
        int cur = -1;
        float2 accv = make_float2(0.f, 0.f);
        auto flush = [&](int rowi) {
            float *pr = gvs + (int64_t)rowi * rs;
            if (excl) {
                pr[0] = accv.x;
                pr[16] = accv.y;
            } else {
                fp_atomic_add(pr, accv.x);
                fp_atomic_add(pr + 16, accv.y);
            }
        };
        auto step = [&](const float2 &en, const float2 &gq) {
            const int pk = __float_as_int(en.y);
            accv.x += en.x * gq.x;
            accv.y += en.x * gq.y;
            cur = (pk >> 10) & 0xfffff;
            if (pk & (1 << 30)) {
                flush(cur);
                accv = make_float2(0.f, 0.f);
                cur = -1;
            }
        };
        int e = lo;
        if (!bucket) {       // every entry is a row of its own: nothing to accumulate, eight independent atomics pairs per trip
            for (; e < hi; e += 8) {
                float2 en[8], gq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) en[u] = entries[min(e + u, hi - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) gq[u] = gt2[(__float_as_int(en[u].y) & 1023) * 16 + l16];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (e + u >= hi) break;
                    float *pr = gvs + (int64_t)((__float_as_int(en[u].y) >> 10) & 0xfffff) * rs;
                    fp_atomic_add(pr, en[u].x * gq[u].x);
                    fp_atomic_add(pr + 16, en[u].x * gq[u].y);
                }
            }
            return;
        }
        for (; e + 8 <= hi; e += 8) {
            float2 en[8], gq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) en[u] = entries[e + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) gq[u] = gt2[(__float_as_int(en[u].y) & 1023) * 16 + l16];
#pragma unroll
            for (int u = 0; u < 8; ++u) step(en[u], gq[u]);
        }
        if (e < hi) {
            float2 en[8], gq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) en[u] = entries[min(e + u, hi - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) gq[u] = gt2[(__float_as_int(en[u].y) & 1023) * 16 + l16];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e + u < hi) step(en[u], gq[u]);
        }
        if (cur >= 0) flush(cur);
    }
}

// The same launch with 1024-thread workgroups (64 streams, four gather blocks per workgroup): the walk of the row-sorted
// entries is a dependent chain per stream (fma -> test -> flush), so what bounds a workgroup is the LENGTH of a stream's
// share, not the instruction count -- twice the streams on the same chunk halve it without flushing any more rows.
constexpr int kLvlThreadsWide = 1024;
constexpr int kLvlQWide = 384;           // queries per workgroup (48 KB of grad_out rows + 48 KB of entries at P = 4)
template <typename IO, int KLP>
__global__ __launch_bounds__(kLvlThreadsWide) void msda_bwd_lvl_merged_wide(
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int Lq, int P, int chunks, int chunk_q,
    int chunks_b, int chunk_q_b, int scatter_blocks, int gather_tiles, int gather_blocks, float *__restrict__ gvalue)
{
    io.same_dims(S, M, L);
    extern __shared__ float4 smem[];
    if ((int)blockIdx.x < scatter_blocks) {
        lvl_scatter_body<IO, NoWait, kLvlThreadsWide, kLvlQWide>((int)blockIdx.x, smem, gout, shapes, starts, io, S, M, L, Lq, P,
                                                                 chunks, chunk_q, gvalue, NoWait(), chunks_b, chunk_q_b);
        return;
    }
    const int part = (int)threadIdx.x >> 8;
    const int vb = (kLvlThreadsWide / 256) * ((int)blockIdx.x - scatter_blocks) + part;
    const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;       // LDS of one gather block, in float4
    gather_body<IO, KLP, 0>(vb, (int)threadIdx.x & 255, smem + part * half_f4, vb < gather_blocks, gout, value, shapes,
                            starts, io, S, M, L, Lq, P, gather_tiles);
}
