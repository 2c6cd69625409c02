// Gather half of the encoder self-attention BACKWARD with ONE LANE PER SAMPLE on region windows in LDS (fp32, D == 32,
// num_point == 4; four levels: L * P == 16 samples per (query, head) = one DPP row, four rows per wave; five levels (round 6, the
// COCO-Full pyramid): L * P == 20, three rows of 20 lanes per wave, lanes 60..63 idle): grad_sampling_loc / grad_attn_weight
// (ms_deform_im2col_cuda.cuh:87-159, :301-403), optionally clearing grad_value for the scatter launch that follows.
// Included by msda.hip after msda_rw.h (window geometry RwWin, rw_first).  Round 5.
//
// Why.  The patch gather (msda_bwd_gather_d32) pulls 4 x 128 B per sample through the vector-memory path (5.8 GB per bs-4 launch
// at 64 B/clk/CU: 282 us, TA-bound like the patch forward); the region-window gather of round 4 (msda_rw_d32<GATHER>) moved the
// rows into LDS but kept the forward's layout -- 8 lanes x float4 per row, so each of a sample's four dot products <grad_out,
// corner row> is spread over 8 lanes and costs a DPP reduction: ~53 lane-instructions per sample and lane, VALU-bound at 325-342 us.
// Here a lane owns a sample: it holds its query's 32 grad_out channels in registers, reads its four corner rows completely
// (4 x 8 ds_read_b128) and forms the four dots in-lane with packed FMAs -- no cross-lane step at all, ~3.5 wave-instructions per
// sample instead of ~7.  Sixteen consecutive lanes = the 16 samples of one (query, head) row = one DPP row, so the fused prologue's
// softmax and the softmax backward's row sum are 4-step DPP reductions.
//   * windows of ALL levels in LDS (a lane cannot fetch a row from global memory without 32 uncoalesced loads): regions of
//     RTH x RTW pixels of the finest level, margins H0 / HC (RwWin, msda_rw.h), staged exactly like the forward's; rows outside
//     a level (and padded rows, MASK) are zeros, which IS the op's zero padding -- the dots need no corner validity;
//   * bank conflicts: ds_read_b128 is served in four groups of 16 lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32)
//     (MI355X_MICROARCH.md, LDS table); a 128-byte row covers half of the 64 banks.  A lane's position g in its group (0..15) gives it
//     a chunk swizzle j = g & 7 -- it reads the eight 16-byte chunks of a row in the order c ^ j, and holds grad_out in the same
//     order -- and a parity class g >> 3: window widths are odd, so a sample's four rows are two of each parity, and class-0 lanes
//     read them in the order even, odd, odd, even, class-1 lanes odd, even, even, odd.  The 16 lanes of a group then hit 16 distinct
//     (parity, chunk) = all 64 banks at every step: conflict-free by construction;
//   * a sample whose footprint leaves its window is handled by the WHOLE wave afterwards (rare while the offsets stay near the
//     queries): lane -> (corner, 8-byte piece) loads the four rows coalesced, multiplies with the query's grad_out piece, a DPP row
//     sum per corner, the owner takes the four dots.  Any input is correct; locality only decides speed (the dispatcher takes the
//     patch gather when the slot's forward policy says the samples are far, msda.hip).
#pragma once

#ifndef SEMIDETR_GW_SB
#define SEMIDETR_GW_SB 1            // dot-loop steps (five ds_read_b128 each) between scheduling barriers
#endif
constexpr int kGwQList = 512;       // queries of a region kept as a list in LDS (a region of a halving pyramid has ~340; larger ones compute them)
// far samples a wave hands to its rows at a time (32-byte entries in LDS; four rows take one entry each per trip: five levels, whose windows
// leave less room, hand over one trip's worth)
constexpr int gw_far_cap(int KL) { return KL >= 5 ? 4 : 8; }
template <int NT, int RTH, int RTW, int H0, int HC, int KL>
constexpr size_t gw_lds_bytes()
{
    // windows + query list + far-sample entries + level table + (round 6) the grad_out rows of a round: 64 / (KL * 4) rows of 128 bytes per wave
    return (size_t)RwWin<RTH, RTW, H0, HC, KL>::total * 128 + (size_t)kGwQList * 4 + (size_t)(NT / 64) * gw_far_cap(KL) * 32 + 128 +
           (size_t)(NT / 64) * (64 / (KL * 4)) * 128;
}

// four 16-lane (DPP row) sums at once, as sixteen fused v_add_f32_dpp (see group8_sum3): every lane of a row gets its row's totals
__device__ __forceinline__ void row16_sum4(float &a, float &b, float &c, float &d)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// Sum / maximum over a 20-lane row of the five-level layout (rows start at lanes 0, 20, 40: quad-aligned, not DPP-row aligned): two DPP
// steps inside the quad (= the four points of one level), then the five quad totals of the row through ds_bpermute, added in quad
// order by every lane -- all lanes of a row get the same bits.  bp = 4 * (row's first lane + my position in my quad).
__device__ __forceinline__ float gw_row_sum(float x, int bp)
{
    x += dpp_mov<0xB1>(x);
    x += dpp_mov<0x4E>(x);
    const int xi = __float_as_int(x);
    const float t0 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp, xi)), t1 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp + 16, xi));
    const float t2 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp + 32, xi)), t3 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp + 48, xi));
    const float t4 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp + 64, xi));
    return ((t0 + t1) + (t2 + t3)) + t4;
}
__device__ __forceinline__ float gw_row_max(float x, int bp)
{
    x = fmaxf(x, dpp_mov<0xB1>(x));
    x = fmaxf(x, dpp_mov<0x4E>(x));
    const int xi = __float_as_int(x);
    const float t0 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp, xi)), t1 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp + 16, xi));
    const float t2 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp + 32, xi)), t3 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp + 48, xi));
    const float t4 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp + 64, xi));
    return fmaxf(fmaxf(fmaxf(t0, t1), fmaxf(t2, t3)), t4);
}

template <typename IO, int NT, int RTH, int RTW, int H0, int HC, int KL, bool MASK = false, int DBG = 0>
__global__ __launch_bounds__(NT, 1) void msda_gw_d32(
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int regions_bound, float4 *__restrict__ zero, int64_t zero_n4)
{
    io.same_dims(S, M, KL);
    using Wn = RwWin<RTH, RTW, H0, HC, KL>;
    // RPW (query, head) rows of LP lanes per wave: 4 x 16 (a row = one DPP row) or 3 x 20 (five levels; the row reductions of the
    // fused prologue then are two DPP steps inside the level's quad + five ds_bpermute over the row's quads, gw_row_sum / gw_row_max)
    constexpr int P = kPT, LP = KL * P, RPW = 64 / LP, QPR = (NT / 64) * RPW;      // QPR: queries per round
    static_assert(H0 >= 0 && (LP == 16 || LP == 20), "every level has a window; 16 or 20 samples per (query, head) row");
    static_assert(gw_lds_bytes<NT, RTH, RTW, H0, HC, KL>() <= 160 * 1024, "windows do not fit the LDS");
    constexpr unsigned kZ0 = (unsigned)Wn::zrow * 128u;
    constexpr int kGwFarCap = gw_far_cap(KL);

    extern __shared__ __attribute__((aligned(128))) float4 smem[];      // (the chunk swizzle XORs into row addresses)
    char *const lds = reinterpret_cast<char *>(smem);
    int *const qlist = reinterpret_cast<int *>(lds + Wn::total * 128);                                  // the region's queries, in slot order
    char *const farl = lds + Wn::total * 128 + kGwQList * 4 + (threadIdx.x >> 6) * (gw_far_cap(KL) * 32);    // my wave's far-sample entries
    // per level {(float)H, (float)W, 1 / H, 1 / W} (round 6): a lane reads its level's entry once per round instead of converting and
    // inverting the sizes there (the fused prologue: four v_rcp_f32 per sample -- location arithmetic and offset gradient)
    float4 *const gtab = reinterpret_cast<float4 *>(lds + Wn::total * 128 + kGwQList * 4 + (NT / 64) * (gw_far_cap(KL) * 32));
    // the grad_out rows of my wave's (query, head) rows of the current round (round 6): each lane loads ONE 8-byte piece of its query's row
    // a round ahead and parks it here; the dot loop reads the row's 16-byte chunks from LDS beside the corner rows -- instead of every
    // lane pulling its query's whole 128-byte row through the vector-memory path into 32 registers (1.46 GB per bs-4 launch for a 91 MB
    // tensor; SQ_WAIT_ANY was 52 % of the wave cycles).  Only my wave touches its rows: LDS operations of a wave execute in order, no barrier.
    constexpr unsigned kGstAt = (unsigned)(Wn::total * 128 + kGwQList * 4 + (NT / 64) * (gw_far_cap(KL) * 32) + 128);
    static_assert(kGstAt % 128 == 0, "the chunk swizzle XORs into the staged rows' addresses");

    // the thread index is rebuilt where it is needed from the wave number (a scalar register) and the lane number (two VALU): as a
    // kernel-long vector register it was what the masked instantiation spilled at 1024 threads (the region scatter's trick, msda_region.h)
    const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    auto fresh_tid = [&]() {
        unsigned ones = ~0u;
        asm volatile("" : "+s"(ones));
        return wave_s * 64 + (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
    };
    const int tid = fresh_tid(), lane = tid & 63;
    // my (query, head) row of the wave, my sample of the row, its level (LP == 20: lanes 60..63 own nothing -- they walk along as
    // sample 0 of a query-less row and store nothing)
    const int rowi = LP == 16 ? lane >> 4 : lane / LP;
    const int k = LP == 16 ? (lane & 15) : (rowi < RPW ? lane - rowi * LP : 0), lvl = k / P;
    // LP == 20: ds_bpermute address of "my position in quad 0 of my row" (the row's quads follow at + 16 bytes each)
    const int bp_row = (rowi < RPW ? rowi * LP + (lane & 3) : lane) * 4;
    (void)bp_row;
    const int Lq = S, rs = M * kD;
    const int b = (int)blockIdx.x;
    const int m = (b % M + (b / M) / kRwHeadRun) % M;
    const int slot0 = (b / M) % regions_bound, n = (b / M) / regions_bound;

    if (zero && DBG != 6) {      // side job: clear grad_value, which the scatter launch that FOLLOWS accumulates into  (DBG 6, timing aid: not cleared)
        const int64_t per = (zero_n4 + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < zero_n4 ? lo + per : zero_n4;
        for (int64_t i = lo + tid; i < hi; i += NT) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // per-level data in lanes 0 .. KL-1 of every wave (as msda_rw_d32): wave-uniform copies by readlane, per-lane ones by __shfl
    const int r_H = lane < KL ? (int)shapes[2 * lane] : 1, r_W = lane < KL ? (int)shapes[2 * lane + 1] : 1;
    const int r_st = lane < KL ? (int)starts[lane] : 0;
    int r_wh = 1, r_ww = 1;
#pragma unroll
    for (int l = 0; l < KL; ++l)
        if (lane == l) { r_wh = Wn::wh(l); r_ww = Wn::ww(l); }
    if (tid < KL) gtab[tid] = make_float4((float)r_H, (float)r_W, fast_rcp((float)r_H), fast_rcp((float)r_W));      // (read behind the staging barriers)
    int Hs[KL], Ws[KL], sts[KL];
#pragma unroll
    for (int l = 0; l < KL; ++l) {
        Hs[l] = __builtin_amdgcn_readlane(r_H, l);
        Ws[l] = __builtin_amdgcn_readlane(r_W, l);
        sts[l] = __builtin_amdgcn_readlane(r_st, l);
    }
    // my sample's level: sizes and window geometry
    const int myH = __shfl(r_H, lvl, 64), myW = __shfl(r_W, lvl, 64), myst = __shfl(r_st, lvl, 64);
    int my_wh1 = 0, my_ww = 1, my_row0 = 0;
#pragma unroll
    for (int l = 0; l < KL; ++l)
        if (lvl == l) { my_wh1 = Wn::wh(l) - 1; my_ww = Wn::ww(l); my_row0 = Wn::row0(l); }
    // padding mask: every level's summary (MaskExt) as a wave-uniform word
    int ves[KL];
#pragma unroll
    for (int l = 0; l < KL; ++l) ves[l] = -1;
    if constexpr (MASK) {
        const int r_ve = lane < KL ? io.mask_ext(n, lane).ve : -1;
#pragma unroll
        for (int l = 0; l < KL; ++l) ves[l] = __builtin_amdgcn_readlane(r_ve, l);
    }
    // position in my ds_read_b128 lane group -> chunk swizzle and parity class (see the header)
    const int l5 = lane & 31;
    const int g = l5 < 4 ? l5 : (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : (l5 < 28 ? l5 - 12 : l5 - 16)));
    const unsigned jj = (unsigned)(g & 7) << 4;           // byte offset of the chunk I read first
    const int cls = g >> 3;

    const int Hb = Hs[0], Wb = Ws[0];                     // the region grid lives on level 0
    const int nry = (Hb + RTH - 1) / RTH, nrx = (Wb + RTW - 1) / RTW, nregions = nry * nrx;
    const __amdgpu_buffer_rsrc_t vr = image_rsrc(value + (int64_t)n * S * M * kD, (unsigned)S * M * kD * 4u);
    const __amdgpu_buffer_rsrc_t gr = image_rsrc(gout + (int64_t)n * Lq * M * kD, (unsigned)Lq * M * kD * 4u);
    const unsigned row_bytes = (unsigned)rs * 4u, head_b = (unsigned)(m * kD) * 4u;

    const int Hb_k = Hb, Wb_k = Wb, nrx_k = nrx, nry_k = nry;
    // The RARE cases -- reference BOXES (ref_dim 4; never with pixel queries in DETR) and levels whose mask is not a summarised band (their
    // bytes decide) -- are loads inside rarely taken branches of the staging, the round loop and the store.  Never taken, they still cost 4 %
    // (242 -> 232 us, round 6: the waits for the loads in flight turn conservative around every such block).  So the region loop exists twice for
    // the fused prologue: kSimple (points + every level of my image summarised: workgroup-uniform, known before the first region) without
    // any of them, and the general one.
    auto run_regions = [&](auto simple_tag) {
    constexpr bool kSimple = decltype(simple_tag)::value;
    for (int reg = slot0; reg < nregions; reg += regions_bound) {
        // (the region grid's sizes through an empty asm: the reciprocals of the divisions by them are rebuilt per region instead of being
        //  held -- and at 1024 threads spilled -- for the whole kernel, as in msda_rw_d32)
        int Hb_r = Hb_k, Wb_r = Wb_k, nrx_r = nrx_k, nry_r = nry_k;
        asm volatile("" : "+s"(Hb_r), "+s"(Wb_r), "+s"(nrx_r), "+s"(nry_r));
        const int Hb = Hb_r, Wb = Wb_r, nrx = nrx_r, nry = nry_r;
        // balanced tiling of the region grid (msda_rw_d32): heights / widths differ by at most one, none exceeds RTH / RTW
        const int ry_b = reg / nrx, rx_b = reg - ry_b * nrx;
        const int y0b = (ry_b * Hb) / nry, y1b = ((ry_b + 1) * Hb) / nry;
        const int x0b = (rx_b * Wb) / nrx, x1b = ((rx_b + 1) * Wb) / nrx;
        // the region's queries: on every level an exact rectangle; window origins: where the region centre maps to, minus half the window
        int r_ylo = 0, r_xlo = 0, r_w = 1, r_cnt = 0;
        float r_invw = 1.f;
        const float pcy = 0.5f * (float)(y0b + y1b) / (float)Hb, pcx = 0.5f * (float)(x0b + x1b) / (float)Wb;
        int r_Hh = r_H, r_Wh = r_W;      // (through an empty asm: their float copies are per-region temporaries, not kernel-long registers)
        asm volatile("" : "+v"(r_Hh), "+v"(r_Wh));
        const int r_wy0 = (int)floorf(pcy * (float)r_Hh - 0.5f) - r_wh / 2 + 1;
        const int r_wx0 = (int)floorf(pcx * (float)r_Wh - 0.5f) - r_ww / 2 + 1;
        if (lane < KL) {
            const int ylo_ = rw_first(y0b, r_H, Hb), yhi_ = y1b >= Hb ? r_H : rw_first(y1b, r_H, Hb);
            const int xlo_ = rw_first(x0b, r_W, Wb), xhi_ = x1b >= Wb ? r_W : rw_first(x1b, r_W, Wb);
            r_ylo = ylo_;
            r_xlo = xlo_;
            r_w = max(xhi_ - xlo_, 1);
            r_cnt = max(yhi_ - ylo_, 0) * max(xhi_ - xlo_, 0);
            r_invw = 1.f / (float)r_w;
        }
        int cnt[KL], wy0[KL], wx0[KL], nq_total = 0;
#pragma unroll
        for (int l = 0; l < KL; ++l) {
            cnt[l] = __builtin_amdgcn_readlane(r_cnt, l);
            wy0[l] = __builtin_amdgcn_readlane(r_wy0, l);
            wx0[l] = __builtin_amdgcn_readlane(r_wx0, l);
            nq_total += cnt[l];
        }
        const int my_wy0 = __shfl(r_wy0, lvl, 64), my_wx0 = __shfl(r_wx0, lvl, 64);
        auto slot_query = [&](int s) -> int {      // s-th query of the region (levels in order) or -1
            const bool ok = s < nq_total;
            int lq = 0;
#pragma unroll
            for (int l = 0; l < KL - 1; ++l)
                if (lq == l && s >= cnt[l]) { s -= cnt[l]; lq = l + 1; }
            const int yl = __shfl(r_ylo, lq, 64), xl = __shfl(r_xlo, lq, 64), w_ = __shfl(r_w, lq, 64);
            const int st_ = __shfl(r_st, lq, 64), W_ = __shfl(r_W, lq, 64);
            const int dy = (int)(((float)s + 0.5f) * __shfl(r_invw, lq, 64));      // s / w_ (s < 2^15: exact)
            return ok ? st_ + (yl + dy) * W_ + xl + (s - dy * w_) : -1;
        };

        // ---- stage the windows through registers: all loads of a thread first, then its stores (as msda_rw_d32)
        {
            constexpr int RPS = NT / 8;
            constexpr int kMaxSteps = (Wn::zrow + RPS - 1) / RPS + KL;
            float4 sv[kMaxSteps];
            unsigned smk[MASK ? kMaxSteps : 1];
            const unsigned char *mask_n = nullptr;
            if constexpr (MASK) mask_n = io.mask + (int64_t)n * S;
            // (the thread index through an empty asm: the per-step window coordinates below depend on nothing else, and hoisted out of
            //  the region loop they are ~40 registers held for the whole kernel -- msda_rw_d32's "lean" staging)
            int tids = fresh_tid();
            asm volatile("" : "+v"(tids));
            const int ocs = tids >> 3, j8s = tids & 7;
            const unsigned lane_bs = head_b + (unsigned)j8s * 16u;
            int nst = 0;
#pragma unroll
            for (int l = 0; l < KL; ++l) {
                const int ww_ = Wn::ww(l), rows_ = Wn::rows(l);
                int r = ocs, wy = ocs / ww_, wx = ocs - wy * ww_;
#pragma unroll
                for (int s = 0; s < (rows_ + RPS - 1) / RPS; ++s) {
                    const int py = wy0[l] + wy, px = wx0[l] + wx;
                    // (MASK: a summarised level's padded rows -- y >= vh or x >= vw -- are not loaded: the out-of-range offset returns the zeros
                    //  value.masked_fill(mask, 0) asks for, as in msda_rw_d32)
                    const int Hv = (MASK && ves[l] >= 0) ? (ves[l] & 0xffff) : Hs[l], Wv = (MASK && ves[l] >= 0) ? (int)((unsigned)ves[l] >> 16) : Ws[l];
                    const bool ok = r < rows_ && (unsigned)py < (unsigned)Hv && (unsigned)px < (unsigned)Wv;
                    const unsigned goff = ok ? (unsigned)(sts[l] + py * Ws[l] + px) * row_bytes + lane_bs : kOob;
                    if constexpr (MASK) {
                        smk[nst] = 0u;
                        if (!kSimple && ves[l] < 0) smk[nst] = mask_n[ok ? sts[l] + py * Ws[l] + px : 0];
                    }
                    sv[nst++] = DBG == 2 ? make_float4(0.f, 0.f, 0.f, 0.f) : buf_ld4(vr, goff);
                    r += RPS;
                    wx += RPS % ww_;
                    wy += RPS / ww_;
                    if (wx >= ww_) { wx -= ww_; ++wy; }
                }
            }
            __syncthreads();               // every wave is done with the previous region's windows
            int ist = 0;
#pragma unroll
            for (int l = 0; l < KL; ++l) {
                const int ww_ = Wn::ww(l), rows_ = Wn::rows(l);
                int r = ocs, wy = ocs / ww_, wx = ocs - wy * ww_;
                (void)wy; (void)wx;
#pragma unroll
                for (int s = 0; s < (rows_ + RPS - 1) / RPS; ++s) {
                    if constexpr (MASK) {      // a level whose mask is not a summarised band: the bytes loaded beside the rows decide
                        if (!kSimple && ves[l] < 0 && smk[ist] != 0) sv[ist] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    if (r < rows_) *reinterpret_cast<float4 *>(lds + (Wn::row0(l) + r) * 128 + j8s * 16) = sv[ist];
                    ++ist;
                    r += RPS;
                }
            }
            if (ocs < Wn::zrows) {
                float z = 0.f;
                asm volatile("" : "+v"(z));      // (a zero the compiler cannot keep in four registers for the whole kernel)
                *reinterpret_cast<float4 *>(lds + kZ0 + ocs * 128 + j8s * 16) = make_float4(z, z, z, z);
            }
            // the region's query list (slot -> pixel): worked out once here instead of by every lane in every round (~40 instructions)
            // (whole waves at a time: slot_query shuffles from lanes 0 .. KL-1, which must be active)
            if (nq_total <= kGwQList)
                for (int s0_ = tids & ~63; s0_ < nq_total; s0_ += NT) {
                    const int s_ = s0_ + (tids & 63), q_ = slot_query(s_);
                    if (s_ < nq_total) qlist[s_] = q_;
                }
        }
        __syncthreads();                   // the windows are complete
        const bool listed = nq_total <= kGwQList;
        auto query_of_slot = [&](int s_) -> int { return listed ? (s_ < nq_total ? qlist[s_] : -1) : slot_query(s_); };

        const int nrounds = (nq_total + QPR - 1) / QPR;
        // A round's data is loaded ahead of its use (without that every wave waited a full memory round trip per round, with three
        // waves per SIMD to hide it): my sample's location / weight a round ahead; my query's grad_out row (32 registers, chunks in my
        // swizzle order) as soon as the previous round's dot loop is done with those registers -- behind it lie that round's far
        // samples, results and stores and this round's softmax and geometry.
        struct Pre {
            int q;
            typename IO::RawXYc rxy;
            float raw;
            float2 g;      // my 8-byte piece of my query's grad_out row (lane k of the row: piece k & 15)
        };
        // (the sampling data and the results are addressed inside MY image: 32-bit index arithmetic on the image's view of the tensors --
        //  the launcher checks Lq * M * L * P * 8 < 2^32 -- instead of 64-bit multiplies by every lane in every round)
        const IO ion = io.image_view(n, Lq, M, LP);
        auto fetch = [&](int round_, Pre &p) {
            if constexpr (LP == 16) {
                p.q = query_of_slot(round_ * QPR + (fresh_tid() >> 4));
            } else {
                const int ln_ = fresh_tid() & 63, rw_ = ln_ / LP;
                p.q = query_of_slot(rw_ < RPW ? round_ * QPR + wave_s * RPW + rw_ : 0x3fffffff);
            }
            const unsigned qs_ = p.q >= 0 ? (unsigned)p.q : 0u;       // (a lane without a query reads query 0 of the image and stores nothing)
            const unsigned row_ = qs_ * (unsigned)M + (unsigned)m;
            p.rxy = ion.load_xy_c(row_, qs_, LP, k, lvl);
            p.raw = ion.load_w(row_, LP, k);
            p.g = buf_ld2(gr, row_ * 128u + (unsigned)(k & 15) * 8u);      // (LP == 20: lanes 16..19 of a row fetch pieces 0..3 again)
        };
        Pre nxt;
        fetch(0, nxt);
        for (int round = 0; round < nrounds; ++round) {
            const Pre cur = nxt;
            const int q = cur.q;
            if (!__any(q >= 0)) break;     // queries are dealt out in order: a wave without one has none later either
            if (SEMIDETR_BRFREE || round + 1 < nrounds) fetch(round + 1, nxt);      // (a slot past the region's last query reads query 0 and stores nothing)
            const bool act = q >= 0;
            const int qs = act ? q : 0;
            const unsigned nq = (unsigned)qs, row = nq * (unsigned)M + (unsigned)m;      // (inside my image: `ion`)
            const float4 lt = gtab[lvl];               // my level: (float)H, (float)W, 1 / H, 1 / W
            const float Hf = lt.x, Wf = lt.y;
            float x, y;
            ion.template finish_xy_c<kSimple>(cur.rxy, nq, lvl, P, lt.w, lt.z, x, y);
            const float raw = cur.raw;
            float a;                                                // fused prologue: softmax over the LP lanes of my row
            if constexpr (LP == 16 || !IO::kSoftmax) {
                a = row_softmax(ion, row, LP, k, raw);              // (16: one DPP row)
            } else {
                const float mx = gw_row_max(raw, bp_row);
                const float e = __expf(raw - mx);                   // __expf / v_rcp_f32 as row_softmax
                a = e * fast_rcp(gw_row_sum(e, bp_row));
            }

            // ---- geometry (ms_deform_im2col_cuda.cuh:285-288 pixel mapping, :56-78 zero padding)
            const float h = sub_rn(mul_rn(y, Hf), 0.5f), w = sub_rn(mul_rn(x, Wf), 0.5f);
            const bool inside = act && h > -1.f && w > -1.f && h < Hf && w < Wf;
            const float h0f = floorf(h), w0f = floorf(w);
            // (a NaN location must not leak through 0 * NaN: fractions are zero unless the sample counts)
            const float lh = inside ? sub_rn(h, h0f) : 0.f, lw = inside ? sub_rn(w, w0f) : 0.f;
            const int wy = (int)h0f - my_wy0, wx = (int)w0f - my_wx0;
            const bool inwin = inside && (unsigned)wy < (unsigned)my_wh1 && (unsigned)wx < (unsigned)(my_ww - 1);
            const bool far = inside && !inwin;
            const int rl = my_row0 + wy * my_ww + wx;      // window row of the top-left corner
            const int sw = (rl ^ cls) & 1;                 // reading order: column sw first (parity of the row ^ my class)
            // (top, sw), (top, !sw), (bottom, sw), (bottom, !sw); a sample not served from its window reads the zero rows
            const unsigned pitch = (unsigned)my_ww * 128u;
            const unsigned a0 = (inwin ? (unsigned)(rl + sw) * 128u : kZ0 + (unsigned)cls * 128u);
            const unsigned a1 = (inwin ? (unsigned)(rl + 1 - sw) * 128u : kZ0 + (unsigned)(1 - cls) * 128u);
            const unsigned a2 = a0 + pitch, a3 = a1 + pitch;

            // ---- four dots <grad_out, corner row>, 16 bytes of every row per step
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f e0 = {0.f, 0.f}, e1 = {0.f, 0.f}, e2 = {0.f, 0.f}, e3 = {0.f, 0.f};
            unsigned jl = jj;
            asm volatile("" : "+v"(jl));
            // my piece of my query's grad_out row -> my wave's staging rows.  (The idle lanes 60..63 of the five-level layout write nothing
            // and read row 1: in their ds_read_b128 group sit lanes of rows 1 and 2, and row 1's other half of the chunks is free there.)
            const unsigned grow = kGstAt + (unsigned)(wave_s * RPW + (rowi < RPW ? rowi : 1)) * 128u;
            if (rowi < RPW) *reinterpret_cast<float2 *>(lds + grow + (unsigned)(k & 15) * 8u) = cur.g;
            // (rows are 128-byte aligned, and so is the window array: my first chunk of each row, as LDS addresses)
            const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)lds;
            const unsigned b0 = (lds0 + a0) ^ jl, b1 = (lds0 + a1) ^ jl, b2 = (lds0 + a2) ^ jl, b3 = (lds0 + a3) ^ jl;
            const unsigned bg = (lds0 + grow) ^ jl;      // (my query's staged row, chunks in my swizzle order like the corner rows)
            if (DBG != 3) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                // (32-bit LDS addresses formed by hand: through the generic `lds + offset` every read carried an add of the array's base)
                typedef const __attribute__((address_space(3))) char *lds_cptr;
                const float4 f0 = *reinterpret_cast<const float4 *>((const char *)(lds_cptr)(uintptr_t)(b0 ^ ((unsigned)c << 4)));
                const float4 f1 = *reinterpret_cast<const float4 *>((const char *)(lds_cptr)(uintptr_t)(b1 ^ ((unsigned)c << 4)));
                const float4 f2 = *reinterpret_cast<const float4 *>((const char *)(lds_cptr)(uintptr_t)(b2 ^ ((unsigned)c << 4)));
                const float4 f3 = *reinterpret_cast<const float4 *>((const char *)(lds_cptr)(uintptr_t)(b3 ^ ((unsigned)c << 4)));
                const float4 gc = *reinterpret_cast<const float4 *>((const char *)(lds_cptr)(uintptr_t)(bg ^ ((unsigned)c << 4)));
                const v2f gl = {gc.x, gc.y}, gh = {gc.z, gc.w};
                e0 = __builtin_elementwise_fma(gl, v2f{f0.x, f0.y}, e0);
                e1 = __builtin_elementwise_fma(gl, v2f{f1.x, f1.y}, e1);
                e2 = __builtin_elementwise_fma(gl, v2f{f2.x, f2.y}, e2);
                e3 = __builtin_elementwise_fma(gl, v2f{f3.x, f3.y}, e3);
                e0 = __builtin_elementwise_fma(gh, v2f{f0.z, f0.w}, e0);
                e1 = __builtin_elementwise_fma(gh, v2f{f1.z, f1.w}, e1);
                e2 = __builtin_elementwise_fma(gh, v2f{f2.z, f2.w}, e2);
                e3 = __builtin_elementwise_fma(gh, v2f{f3.z, f3.w}, e3);
                // two steps' reads in flight (the scheduler would hoist all 32: 128 registers); ONE at 1024 threads (128 registers in all)
                if ((c % SEMIDETR_GW_SB) == SEMIDETR_GW_SB - 1) __builtin_amdgcn_sched_barrier(0);
            }
            }
            asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));      // (the partial sums are final here: nothing of the loop sinks below)
            const float s0 = e0.x + e0.y, s1 = e1.x + e1.y, s2 = e2.x + e2.y, s3 = e3.x + e3.y;
            // reading order -> corner order: top-left, top-right, bottom-left, bottom-right
            float d1 = sw ? s1 : s0, d2 = sw ? s0 : s1, d3 = sw ? s3 : s2, d4 = sw ? s2 : s3;

            // ---- samples that left their windows: the whole wave on their rows, four samples per trip (their loads in flight together:
            //      one at a time, every sample cost a memory round trip -- 4.4 us per round, the whole kernel's time).  The owner
            //      lane works out the four corner offsets (kOob: outside the level / padded); lane -> (corner lane / 16, 8-byte piece lane % 16)
            unsigned foff[4] = {kOob, kOob, kOob, kOob};
            if (far) {
                float lw_, lh_;
                sample_setup_oob(x, y, myH, myW, myst, row_bytes, foff, lw_, lh_);
                if constexpr (MASK) {
                    int ve_my = ves[0];
#pragma unroll
                    for (int l = 1; l < KL; ++l) ve_my = lvl == l ? __builtin_amdgcn_readfirstlane(ves[l]) : ve_my;
                    mask_corners_oob<IO, !kSimple>(io, MaskExt{ve_my}, n, x, y, myH, myW, myst, foff);
                }
            }
            const unsigned long long fb = DBG == 4 ? 0ull : __ballot(far);
            if (fb) {      // (wave-uniform)
                // Every far sample becomes a 32-byte entry {4 corner offsets, query} in my wave's LDS list, at its rank among the wave's
                // far samples; then each of the wave's four 16-lane rows takes one entry per trip: lane -> 8-byte piece of the rows,
                // four products with the query's grad_out piece, four DPP row sums, the dots go back through the entry.  (One sample
                // per WAVE and trip, with readlane broadcasts, was ~35 instructions per sample: as much as the whole window path.)
                const int nfar = __builtin_popcountll(fb);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(fb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fb, 0u));
                const int rrow = lane >> 4;
                const unsigned piece = (unsigned)(lane & 15) * 8u;
                for (int base = 0; base < nfar; base += kGwFarCap) {
                    const bool mine = far && rank >= base && rank < base + kGwFarCap;
                    if (mine) {
                        *reinterpret_cast<uint4 *>(farl + (rank - base) * 32) = make_uint4(foff[0], foff[1], foff[2], foff[3]);
                        *reinterpret_cast<int *>(farl + (rank - base) * 32 + 16) = qs;
                    }
                    const int nhere = min(nfar - base, kGwFarCap);
                    for (int e0_ = 0; e0_ < nhere; e0_ += 4) {
                        const int e_ = e0_ + rrow;
                        const bool on = e_ < nhere;
                        const uint4 o = *reinterpret_cast<const uint4 *>(farl + (on ? e_ : 0) * 32);
                        const int q_s = *reinterpret_cast<const int *>(farl + (on ? e_ : 0) * 32 + 16);
                        const unsigned hb = head_b + piece;
                        const float2 v0 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(vr, on ? o.x + hb : kOob, 0, 0));
                        const float2 v1 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(vr, on ? o.y + hb : kOob, 0, 0));
                        const float2 v2 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(vr, on ? o.z + hb : kOob, 0, 0));
                        const float2 v3 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(vr, on ? o.w + hb : kOob, 0, 0));
                        const float2 gg = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(gr, (unsigned)(q_s * M + m) * 128u + piece, 0, 0));
                        float p0 = v0.x * gg.x + v0.y * gg.y, p1 = v1.x * gg.x + v1.y * gg.y;
                        float p2 = v2.x * gg.x + v2.y * gg.y, p3 = v3.x * gg.x + v3.y * gg.y;
                        row16_sum4(p0, p1, p2, p3);
                        if (on && (lane & 15) == 0) *reinterpret_cast<float4 *>(farl + e_ * 32) = make_float4(p0, p1, p2, p3);
                    }
                    if (mine) {
                        const float4 dd = *reinterpret_cast<const float4 *>(farl + (rank - base) * 32);
                        d1 = dd.x; d2 = dd.y; d3 = dd.z; d4 = dd.w;
                    }
                }
            }

            // ---- the three small gradients of my sample (.cuh:87-159 without the scatter)
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float t_ = hw * d1 + lw * d2, b_ = hw * d3 + lw * d4;
            const float g_a = inside ? hh * t_ + lh * b_ : 0.f;
            const float g_x = inside ? a * (hh * (d2 - d1) + lh * (d4 - d3)) : 0.f;
            const float g_y = inside ? a * (b_ - t_) : 0.f;
            float dot = 0.f;                       // fused epilogue: sum_k a_k g_k over the row (softmax backward)
            if (IO::kSoftmax) dot = LP == 16 ? lp_group_sum(a * g_a, 16) : gw_row_sum(a * g_a, bp_row);
            if (act && (DBG != 5 || g_a == 1.2345e30f)) ion.template store_px<kSimple>(row, nq, LP, k, lvl, P, Hf, Wf, g_a, g_x, g_y, a, dot);      // (DBG 5, timing aid: nothing stored)
        }
    }
    };
    if constexpr (IO::kSoftmax) {
        bool simple = io.ref_dim == 2;
#pragma unroll
        for (int l = 0; l < KL; ++l) simple = simple && (!MASK || ves[l] >= 0);
        if (simple) run_regions(std::true_type());
        else run_regions(std::false_type());
    } else {
        run_regions(std::false_type());      // (the reference contract has no such branches)
    }
}
