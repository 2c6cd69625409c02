// Encoder self-attention with the value rows served from REGION WINDOWS in LDS (fp32, D == 32, num_point == 4):
// forward (msda_rw_d32<..., false>) and the gather half of the backward (<..., true>: grad_sampling_loc /
// grad_attn_weight, optionally clearing grad_value for the scatter launch that follows).  Included by msda.hip after
// msda_fast.h / msda_region.h.  PRODUCT since round 4: the forward instantiations <768 threads, regions of up to 25 x 16 pixels, level 0
// through global loads, margin 5 on the coarse levels> for four levels and <960 threads, up to 24 x 16, margin 4> for five are what
// launch_fast_forward picks while most samples stay within a few pixels of their queries (FwdStats); the other
// configurations and the gather half are reachable from the experiments library only.  The TUNE flags (scheduling barriers, level-0
// samples in flight, which values are rebuilt per round / region instead of held in registers) are listed at msda.hip's
// SEMIDETR_RW_TUNE; tools/rw_regs.sh compiles one configuration in seconds and prints its VGPRs and spills.
//
// Why.  The patch kernels (msda_fwd_d32<1,4,408>, msda_bwd_gather_d32) pull every corner row through the vector-memory
// path: 4 x 22223 x 8 heads x 16 samples x 4 corners x 128 B = 5.8 GB per bs-4 launch at 64 B/clk/CU -- TA busy 80 %,
// 0.18 of the HBM roofline, and a load that never touches the cache costs almost as much as a real one
// (profiles/r02_fwd_enc_TA.txt).  LDS reads run at 256 B/clk/CU (ds_read_b128), four times that rate, but a window only
// pays when many corner reads share it.  Three earlier attempts staged one window per (query patch of ONE level,
// sampling level): ~3 reads per staged row, a barrier pair and a dependent load -> geometry -> record chain per level,
// coarse-level patches whose footprint on the fine levels never fits -- all slower than the plain kernel.
//
// Here the unit of work is a REGION of the image, as in the region-owned scatter (msda_region.h): a workgroup owns an
// RTH x RTW tile of the finest level and ONE head and takes the queries of ALL levels whose pixel centres lie in it
// (128 + 32 + 8 + 2 for a halving pyramid).  Their samples on level l all fall around the same spot, so one window per
// sampling level, staged ONCE per region for all levels together, serves every query of the region: ~10 corner reads per
// staged row, one barrier between staging and compute, no per-level phases.
//   * staging: every thread issues all its 16-byte pieces as buffer loads (rows outside a level arrive as zeros through
//     the bounds check, which IS the op's zero padding, so the compute loop needs no per-corner validity at all), then
//     stores them with ds_write_b128.  (buffer_load ... lds was tried first: the DMA path sustains ~13 B/clk/CU here,
//     84 us of a 330 us launch.)
//   * 8 lanes per query, 8 queries per wavefront: lane j does the geometry of samples j, j + 8 (all lanes busy), the
//     records {4 corner weights} + {two window offsets} go through a per-octet LDS scratch that only the owning wave
//     touches (no workgroup barrier); in the compute loop the octet reads the four corner rows with two address
//     registers and the window pitch of the sample's level as an immediate, 8 lanes x float4 per row;
//   * bank conflicts: ds_read_b128 is served in groups of 16 lanes = half-rows of 4 different octets (A.first, B.second,
//     C.second, D.first -- MI355X_MICROARCH.md, LDS table); a 128-byte row covers half of the 64 banks, so A / D (and
//     B / C) collide iff their rows have equal parity.  All window widths are odd (left / right and top / bottom
//     neighbours differ in parity), octets A, B read their sample's corners in the order even, odd, odd, even, octets
//     C, D in the order odd, even, even, odd: conflict-free by construction;
//   * a sample whose footprint leaves its window (or any sampling pattern the windows were not sized for) takes the
//     plain kernel's path at the end of the query's round: global corner loads through the same buffer resource.  Any
//     input is correct; locality only decides speed.  A round in which more than a third of the samples fall outside is
//     processed entirely that way (the windows were staged in vain, nothing worse).
// Queries are enumerated from the level table exactly like the region scatter (by(qy) = ((2 qy + 1) Hb) / (2 Hq) is
// monotone: every level contributes an exact rectangle, every query belongs to exactly one region), so the kernels need
// SEMIDETR_MSDA_QUERIES_ARE_PIXELS and num_levels == KL.
#pragma once

// Round 6: NO BRANCH AROUND LOADS IN THE ROUND LOOPS of the window kernels (msda_rw_d32 forward, msda_gw_d32).  The next round's sampling data
// is requested unconditionally (a lane / a last round without a query reads query 0 of the image and uses nothing of it) and the window loop,
// which issues the level-0 corner loads, is not wrapped in the rare "plain round" test.  A branch around a load is a basic-block boundary at
// which the waits for everything in flight turn conservative: forward 185 -> 179 us, and the compiler stops holding two versions of the
// round's state -- 168 -> 129 VGPRs (reference contract), 165 -> 141 (fused + mask).  0 = round 5's control flow (A/B: tools/ab_build.sh).
#ifndef SEMIDETR_BRFREE
#define SEMIDETR_BRFREE 1
#endif
constexpr int kRwHeadRun = 16;      // head rotation of the region kernels (see tile_of_block)

// Window geometry at compile time.  Level l of a halving pyramid sees the region as (RTH >> l) x (RTW >> l) pixels;
// H0 / HC = margin in pixels around it on level 0 / on the coarser levels (whose pixel centres are not aligned with
// the region's edges: one more row / column).  Widths are rounded up to odd (bank parity, see above).
template <int RTH, int RTW, int H0, int HC, int KL>
struct RwWin {
    __host__ __device__ static constexpr int ext(int r, int l) { return (r >> l) < 1 ? 1 : (r >> l); }
    __host__ __device__ static constexpr int wh(int l) { return ext(RTH, l) + (l ? 2 * HC + 1 : 2 * H0); }
    __host__ __device__ static constexpr int ww(int l) { return (ext(RTW, l) + (l ? 2 * HC + 1 : 2 * H0)) | 1; }
    static constexpr bool fine_global = H0 < 0;           // level 0 has no window: its samples are loaded from global memory
    __host__ __device__ static constexpr int rows(int l) { return (l == 0 && H0 < 0) ? 0 : wh(l) * ww(l); }
    __host__ __device__ static constexpr int row0(int l)      // first row of level l's window: even (parity bookkeeping)
    {
        int s = 0;
        for (int i = 0; i < l; ++i) s += (rows(i) + 1) & ~1;
        return s;
    }
    __host__ __device__ static constexpr int maxww()
    {
        int w = 0;
        for (int i = (H0 < 0 ? 1 : 0); i < KL; ++i) w = ww(i) > w ? ww(i) : w;
        return w;
    }
    static constexpr int zrow = row0(KL);                 // even.  Rows [zrow, zrow + zrows) are zeros: a sample that is not
    static constexpr int zrows = (maxww() + 2 + 1) & ~1;  // served from its window reads rows zrow (+1) and those + its pitch
    static constexpr int total = zrow + zrows;
};

template <int KL, bool FG = false, bool PRE = false, bool WIDE = false>
constexpr int rw_oct_bytes()      // per octet: {float4 record} and {two 16-bit window offsets} per windowed sample | two 32-byte slots (+ bank spread)
{
    // WIDE (TUNE + 51200, forward with level 0 through global loads): the two window offsets of a sample as two 32-bit BYTE offsets --
    // the loop adds the lane's base to each (2 VALU per sample instead of shift / mask / add / bit-field extract / shift-add)
    if (WIDE) {
        constexpr int raww = (KL - 1) * kPT * 24 + kPT * 32;
        return raww + (raww % 128 == 0 ? 16 : 0);
    }
    // level 0 without a window (FG): its samples have no window record; their {corner offsets, weights} records (32 bytes each) are
    // dead by the time the out-of-window loop needs its two slots, so the slots lie on top of them -- unless out-of-window samples
    // are pre-issued (PRE: TUNE % 10 > 0), whose slots are written while those records are still live
    constexpr int raw = FG ? (KL - 1) * kPT * 20 + kPT * 32 + (PRE ? 64 : 0) : KL * kPT * 20 + 64;
    return raw + (raw % 128 == 0 ? 16 : 0);      // octet pitch: A, B, C, D on distinct banks
}

// Padding mask (MASK instantiations, fused prologue only: ops/modules/ms_deform_attn.py:95-96 `value.masked_fill(mask[..., None], 0)`).
// The windows are staged with the padded rows as zeros; level 0, which has no window, drops padded corners from its records.  What
// is padding comes from the level's summary {vh, vw} (MaskExt, msda_fast.h: rows >= vh / columns >= vw -- every mask DETR builds) as two
// compares per row / corner; a level without one (vh < 0) reads the mask's bytes: beside the staging loads for the windows (no
// dependent load), from global memory for level-0 corners and out-of-window samples (correct for any mask; only the summarised
// form is fast).
constexpr int kRwQList = 1024;      // TUNE + 12800: slots of the region's query list in LDS (regions with more queries work them out per round)
template <int NT, int RTH, int RTW, int H0, int HC, int KL, int TUNE = 0>
constexpr size_t rw_lds_bytes()
{
    return (size_t)RwWin<RTH, RTW, H0, HC, KL>::total * 128 + (size_t)(NT / 8) * rw_oct_bytes<KL, (H0 < 0), (TUNE % 10 > 0), (((TUNE / 100) & 512) != 0)>() +
           (((TUNE / 100) & 64) ? (size_t)(KL + 1) * 48 : 0) + (((TUNE / 100) & 128) ? (size_t)kRwQList * 4 : 0);
}

// smallest q in [0, nq] with ((2 q + 1) * nb) / (2 * nq) >= bound  (the first pixel of a level with nq rows whose centre
// maps into row >= bound of the nb-row grid level), i.e. with (2 q + 1) * nb >= 2 * nq * bound.  One 32-bit division; levels so
// large that the product needs more bits (> 32767 pixels wide) bisect instead -- a 64-bit division would put its ~40 registers
// into the kernel's budget although it never runs.
__device__ __forceinline__ int rw_first(int bound, int nq, int nb)
{
    const unsigned long long a = 2ull * (unsigned)nq * (unsigned)bound;
    if (((a + (unsigned)nb) >> 32) == 0) {
        const unsigned cc = ((unsigned)a + (unsigned)nb - 1u) / (unsigned)nb;      // ceil(a / nb): 2 q + 1 >= cc
        return min(nq, (int)(cc >> 1));
    }
    int lo = 0, hi = nq;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if ((2ull * (unsigned)mid + 1ull) * (unsigned)nb >= a) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

// acc += w.x f1 + w.y f2 + w.z f3 + w.w f4 over four channels as eight v_pk_fma_f32 whose weight operand is one HALF of an aligned
// register pair (op_sel): the compiler finds the low half (w.x, w.z) and the high half of the first pair (w.y), but copies w.w into
// a fresh pair first -- two v_mov per sample, 24 per round of the window loop.  Same operation order per channel as the fmaf chain.
typedef float rw_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rw_fma4(float4 &acc, const float4 &w, const float4 &f1, const float4 &f2, const float4 &f3,
                                        const float4 &f4)
{
    rw_v2f a01 = {acc.x, acc.y}, a23 = {acc.z, acc.w};
    const rw_v2f wxy = {w.x, w.y}, wzw = {w.z, w.w};
#define RW_PK_LO(ACC, W, FX, FY) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(ACC) : "v"(W), "v"(rw_v2f{FX, FY}))
#define RW_PK_HI(ACC, W, FX, FY) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(ACC) : "v"(W), "v"(rw_v2f{FX, FY}))
    RW_PK_LO(a01, wxy, f1.x, f1.y);
    RW_PK_LO(a23, wxy, f1.z, f1.w);
    RW_PK_HI(a01, wxy, f2.x, f2.y);
    RW_PK_HI(a23, wxy, f2.z, f2.w);
    RW_PK_LO(a01, wzw, f3.x, f3.y);
    RW_PK_LO(a23, wzw, f3.z, f3.w);
    RW_PK_HI(a01, wzw, f4.x, f4.y);
    RW_PK_HI(a23, wzw, f4.z, f4.w);
#undef RW_PK_LO
#undef RW_PK_HI
    acc = make_float4(a01.x, a01.y, a23.x, a23.y);
}

// DBG (tuning builds only): 1 = per-phase cycle counts of wave 0 into g_dest_dbg, 2 = windows not staged (results
// wrong, timing aid), 3 = compute loop skipped (results wrong, timing aid), 4 = out-of-window samples dropped, 5 = level-0 samples
// (global loads) dropped, 7 = 3 + 5, 8 = 3 + 4 + 5 (what is left: staging, sampling-data loads, geometry, records, result stores)
// TUNE = 100 * flags + 10 * (compute-loop samples between scheduling barriers) + (out-of-window samples per octet whose loads are
//        issued ahead of the compute loop: with level 0 through global loads their slots need 64 more bytes per octet; measured level,
//        0 in the product).  Flags (every combination gives the same results):
//          1  one level-0 sample's corner loads in flight instead of two (-33 VGPRs)
//          2  "lean": the per-lane level constants are re-selected where they are used, the staging coordinates rebuilt per region
//          4  one instead of two out-of-window samples per trip of the fall-back loop
//          8  everything else derived from the thread index (octet / window addresses, per-level lane values, division
//             reciprocals of the region grid) rebuilt per round / region through an empty asm: what fits 1024 threads into 128 VGPRs
//         16  the prefetched sampling data stay as loaded; the fused prologue's location arithmetic runs in the consuming round
//         32  window addresses by v_mad_u32_u16, packed FMAs with explicit op_sel (measured level; experiments)
//         64  (round 5, with 2 and 16) the per-lane level constants of the geometry come from a 48-byte-per-level table in LDS, three
//             ds_read_b128 per pass, instead of one v_mov + v_cndmask pair per constant (gfx950 VALU instructions take ONE scalar
//             operand, so "select between two wave-uniform values" is two instructions: 100 of the ~300 the geometry of a round took)
//        256  (with 64; forward) COMPACT records for out-of-window samples: where the geometry finds a sample outside its window it
//             writes {lw, lh, attention, top-left pixel} into the sample's own weight record (its rows are the zero rows, so the LDS
//             loop adds 0 x finite), and the out-of-window loop builds the four corner offsets from that and the level table on all
//             eight lanes -- instead of the owner lane redoing the whole geometry from (x, y) inside a branch (publish), a slot write,
//             a wait and a slot read per sample.  The sampling data (x, y, attention of both passes) die before the LDS loop.
//        512  (forward, level 0 through global loads) a sample's two window offsets as two 32-bit byte offsets: 8 instead of 4 bytes
//             per record, 2 instead of 5 VALU instructions per sample of the LDS loop, no packing in the geometry
//       1024  the tail split of small launches (helper workgroups take part of the rounds of the last wave's units, see the kernel)
//        128  the region's query list (slot -> query) is worked out once per region into LDS; a round reads its slot instead of
//             redoing the level search, five shuffles and a division (~35 instructions and 7 ds_bpermute per round)
//        Product: 1920 = 16 + 2 + 1, two samples per barrier (four levels); 1110 = 8 + 2 + 1, one sample per barrier (five levels).
template <typename IO, int NT, int RTH, int RTW, int H0, int HC, int KL, bool GATHER, int DBG = 0, int TUNE = 42, bool MASK = false>
__global__ __launch_bounds__(NT, (NT >= 512 ? NT / 256 : 2)) void msda_rw_d32(      // 256-thread workgroups: two per CU
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int regions_bound, float *__restrict__ out,
    float4 *__restrict__ zero, int64_t zero_n4, const FwdStats fs = FwdStats{nullptr, 0u}, int tail_cus = 0)
{
    io.same_dims(S, M, KL);
    using Wn = RwWin<RTH, RTW, H0, HC, KL>;
    constexpr int P = kPT, KLP = KL * P, G = NT / 8, NPASS = (KLP + 7) / 8;
    constexpr bool FG = Wn::fine_global;                  // level 0 through global loads (forward only)
    constexpr bool kWide = ((TUNE / 100) & 512) != 0;
    static_assert(!kWide || (FG && !GATHER && TUNE % 10 == 0), "wide window offsets: the forward's product shape");
    constexpr int kOctBytes = rw_oct_bytes<KL, FG, (TUNE % 10 > 0), kWide>();
    constexpr int kRec0 = FG ? P : 0;                     // first sample with a window record
    constexpr int kOffAt = (KLP - kRec0) * 16, kFineAt = (KLP - kRec0) * (kWide ? 24 : 20), kSlotAt = (FG && TUNE % 10 > 0) ? kFineAt + P * 32 : kFineAt;
    static_assert(Wn::total * 128 <= (1 << 20), "window offsets are kept in 16 bits, in units of 16 bytes");
    constexpr unsigned kZ0 = (unsigned)Wn::zrow * 128u;
    static_assert(P == 4 && KLP <= 32, "lane j of an octet owns samples j, j + 8, ...");
    static_assert(kOctBytes % 16 == 0, "records are read with ds_read_b128");
    static_assert(!MASK || (FG && !GATHER), "the padding mask is built into the product configuration: forward, level 0 through global loads");

    extern __shared__ float4 smem[];
    char *const lds = reinterpret_cast<char *>(smem);
    char *const recs = lds + Wn::total * 128;
    constexpr bool kTab = ((TUNE / 100) & 64) != 0, kQList = ((TUNE / 100) & 128) != 0, kCompact = ((TUNE / 100) & 256) != 0;
    static_assert(!kCompact || (kTab && !GATHER && TUNE % 10 == 0), "compact out-of-window records: forward, level table, nothing pre-issued");
    // (round 6, SEMIDETR_BRFREE) the window loop -- with the level-0 corner loads issued inside it -- is not wrapped in the "plain round"
    // branch: a plain round (rare: more than a third of the samples outside their windows) points every record at the zero rows and runs it
    // for nothing, the common round has one basic block less around loads in flight
    constexpr bool kLoopAlways = SEMIDETR_BRFREE && kCompact;
    static_assert(!kTab || (((TUNE / 100) & 2) && ((TUNE / 100) & 16) && !GATHER && H0 < 0), "the level table serves the lean forward with split loads");
    // level table: per level {H, W, start, window row0 | window rows - 1, window columns - 1, window pitch, window origin y |
    //                          window origin x, (float)H, (float)W, -}; then the query list
    int4 *const ltab = reinterpret_cast<int4 *>(recs + (NT / 8) * kOctBytes);
    int *const qlist = reinterpret_cast<int *>(recs + (NT / 8) * kOctBytes + (kTab ? (KL + 1) * 48 : 0));

    const int tid = threadIdx.x, lane = tid & 63;
    const int oc = tid >> 3, j8 = tid & 7;
    const int cls = (tid >> 4) & 1;                       // octets C, D of each 32-lane half read odd rows first
    const int Lq = S, rs = M * kD;
    const int b = (int)blockIdx.x;
    unsigned long long tmark = DBG == 1 ? __builtin_readcyclecounter() : 0ull;
    (void)tmark;
    auto lap = [&](int slot_) {
        if (DBG == 1 && tid == 0) {
            const unsigned long long now = __builtin_readcyclecounter();
            SEMIDETR_DBG_ADD(slot_, now - tmark);
            tmark = now;
        }
    };
    unsigned st_far = 0, st_total = 0;

    if (GATHER && zero) {      // side job: clear grad_value, which the scatter launch that FOLLOWS accumulates into
        const int64_t per = (zero_n4 + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < zero_n4 ? lo + per : zero_n4;
        for (int64_t i = lo + tid; i < hi; i += NT) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // Per-level data lives in lanes 0 .. KL-1 of every wave (r_*): wave-uniform copies come from readlane with a constant
    // lane, per-lane selections from __shfl with the level as the source lane.  (Select chains over small arrays are
    // turned into dynamically indexed scratch accesses by the compiler -- and a scratch load waits for vmcnt.)
    const int r_H = lane < KL ? (int)shapes[2 * lane] : 1, r_W = lane < KL ? (int)shapes[2 * lane + 1] : 1;
    const int r_st = lane < KL ? (int)starts[lane] : 0;
    int r_wh = 1, r_ww = 1, r_row0 = 0;          // window geometry of level `lane` (compile-time values, lane-selected)
#pragma unroll
    for (int l = 0; l < KL; ++l)
        if (lane == l) { r_wh = Wn::wh(l); r_ww = Wn::ww(l); r_row0 = Wn::row0(l); }
    const int r_H_k = r_H, r_W_k = r_W, r_st_k = r_st, r_wh_k = r_wh, r_ww_k = r_ww, r_row0_k = r_row0;      // (the region loop has its own)
    int Hs[KL], Ws[KL], sts[KL];
#pragma unroll
    for (int l = 0; l < KL; ++l) {
        Hs[l] = __builtin_amdgcn_readlane(r_H, l);
        Ws[l] = __builtin_amdgcn_readlane(r_W, l);
        sts[l] = __builtin_amdgcn_readlane(r_st, l);
    }
    // the region grid lives on level 0 (the finest level of a DETR pyramid; any order is correct, the windows are sized
    // for this one)
    const int Hb = Hs[0], Wb = Ws[0];
    const int nry = (Hb + RTH - 1) / RTH, nrx = (Wb + RTW - 1) / RTW;

    // ---- which (image, head, region) is mine -- and the TAIL SPLIT (round 5).  One workgroup per CU, every (image, head, region) unit
    // about equally long: a launch of U units takes ceil(U / CUs) unit times -- 1408 units on 256 CUs = 6 for 5.5 (bs 4), 352 = 2 for
    // 1.4 (bs 1).  So the units of the last, partly filled wave (in dispatch order: the last U mod CUs ones) are cut into s =
    // CUs / (U mod CUs) parts of their ROUNDS (rounds are independent); part 0 stays with the unit's own workgroup, the others go to
    // `tail_cus` helper workgroups appended to the grid (dispatched last; the ones not needed leave at once).  Each part stages the
    // region's windows again (8 % of a unit), so the last wave takes ~0.6 instead of 1 unit time.  tail_cus = 0: no helpers.
    // Small launches only, see s_ below.
    // It is its own instantiation (TUNE + 102400), launched for small launches only: with the code merely PRESENT the four-image launch
    // measured 1.7 % slower inside the step (different schedule of the same loop), which is more than the one-image launch gains.
    constexpr bool kTail = !GATHER && ((TUNE / 100) & 1024) != 0;
    int b_unit = b, part = 0, nparts = 1;
    if (kTail && tail_cus > 0) {
        const int nreg_all = nry * nrx, base_grid = (int)gridDim.x - tail_cus;
        const int nimg = base_grid / (regions_bound * M);
        const int U = nimg * nreg_all * M, full = U / tail_cus * tail_cus, rem = U - full;
        // (a workgroup takes one region only while the grid's bound covers the region grid; otherwise no split)
        // ... and only launches of fewer than three waves are cut: measured (tools/r05_ab_fwd.sh) one image 56.8 -> 50.8 us, but four
        // images (5.5 waves, unit times spread by the narrow edge regions, the tail filled anyway) 175.3 -> 178.6 us
        const int s_ = (rem > 0 && regions_bound >= nreg_all && U < 3 * tail_cus) ? min(tail_cus / rem, 4) : 1;
        if (b >= base_grid) {
            const int hidx = b - base_grid;
            if (s_ <= 1 || hidx >= rem * (s_ - 1)) return;      // (the whole workgroup, before any barrier)
            const int u = full + hidx % rem;
            part = 1 + hidx / rem;
            nparts = s_;
            b_unit = ((u / (nreg_all * M)) * regions_bound + (u / M) % nreg_all) * M + u % M;
        } else {
            const int sl = (b / M) % regions_bound, ni = (b / M) / regions_bound;
            if (s_ > 1 && sl < nreg_all && (ni * nreg_all + sl) * M + b % M >= full) nparts = s_;
        }
    }
    const int m = (b_unit % M + (b_unit / M) / kRwHeadRun) % M;
    const int slot0 = (b_unit / M) % regions_bound, n = (b_unit / M) / regions_bound;
    // how far the samples reach, for the dispatcher's choice between this kernel and the patch kernel (FwdStats, msda_fast.h):
    // one workgroup in 128 counts (its first round: G = NT / 8 queries x 12 samples; ~20 workgroups of a bs-4 launch), the launch's first
    // workgroup publishes the previous launch's pair
    const bool sampled = !GATHER && fs.on() && (b & 127) == 1 && part == 0;

    const __amdgpu_buffer_rsrc_t vr = image_rsrc(value + (int64_t)n * S * M * kD, (unsigned)S * M * kD * 4u);
    const unsigned lane_b = (unsigned)(m * kD + 4 * j8) * 4u;
    const unsigned row_bytes = (unsigned)rs * 4u;
    // forward: the sampling data and `out` are addressed inside MY image (the launcher checks that an image's tensors stay below 2^32
    // bytes): 32-bit index arithmetic instead of 64-bit multiplies per lane and round
    constexpr bool kView = !GATHER && kTab;
    const IO iov = kView ? io.image_view(n, Lq, M, KLP) : io;
    float *const out_n = kView ? out + (int64_t)n * Lq * M * kD : out;
    // MASK: this image's (S,) padding bytes.  Rebuilt where it is used (the image index goes through an empty asm): as a kernel-long
    // value the pointer was the scalar register pair that pushed another one out to scratch.
    const unsigned char *mask_n = nullptr;
    if constexpr (MASK) mask_n = io.mask + (int64_t)n * S;
    auto mask_of_image = [&]() -> const unsigned char * {
        if constexpr (MASK) {
            int n_ = n;
            asm volatile("" : "+s"(n_));
            return io.mask + (int64_t)n_ * S;
        } else {
            return nullptr;
        }
    };
    // ... and each level's summary {vh, vw} (MaskExt, msda_fast.h: padding = rows >= vh or columns >= vw; vh < 0: not of that form,
    // the bytes decide) in lane `level` of every wave, like r_H: two compares per corner instead of a byte per corner
    // (wave-uniform words: as scalar values they cost spill LANES of a register the kernel holds anyway; one more vector register
    //  for all of the kernel made the five-level instantiation spill)
    int ves[KL];
#pragma unroll
    for (int l = 0; l < KL; ++l) ves[l] = -1;
    if constexpr (MASK) {
        const int r_ve = lane < KL ? io.mask_ext(n, lane).ve : -1;
#pragma unroll
        for (int l = 0; l < KL; ++l) ves[l] = __builtin_amdgcn_readlane(r_ve, l);
    }
    auto ext_vh = [](int ve) { return ve < 0 ? -1 : (ve & 0xffff); };
    auto ext_vw = [](int ve) { return ve < 0 ? -1 : (int)((unsigned)ve >> 16); };

    // the level of this lane's sample in pass p (sample k = j8 + 8 p, level k / 4) and its static constants
    int myl[NPASS], myH[NPASS], myW[NPASS], myst[NPASS], my_wh1[NPASS], my_ww1[NPASS], my_ww[NPASS], my_row0[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int l = min((j8 + 8 * p) / P, KL - 1);
        myl[p] = l;
        myH[p] = __shfl(r_H, l, 64);
        myW[p] = __shfl(r_W, l, 64);
        myst[p] = __shfl(r_st, l, 64);
        my_wh1[p] = __shfl(r_wh, l, 64) - 1;
        my_ww[p] = __shfl(r_ww, l, 64);
        my_ww1[p] = my_ww[p] - 1;
        my_row0[p] = __shfl(r_row0, l, 64);
    }

    // TUNE >= 200: no per-lane copies of the level constants (22 VGPRs that live through the whole kernel); a pass's two levels
    // are compile-time, so each constant is ONE select between two wave-uniform values, redone where it is needed (the lane
    // predicate goes through an empty asm so that the selects are not hoisted back out of the round loop)
    constexpr bool kLean = ((TUNE / 100) & 2) != 0;
    static_assert(!kLean || P == 4, "a pass covers two levels: lanes 0-3 / 4-7");
    struct LvlC { int l, H, W, st, wh1, ww1, ww, row0; };
    auto lvlc = [&](int p) -> LvlC {
        if constexpr (!kLean) {
            return LvlC{myl[p], myH[p], myW[p], myst[p], my_wh1[p], my_ww1[p], my_ww[p], my_row0[p]};
        } else {
            int jj = j8;
            asm volatile("" : "+v"(jj));
            const bool hi = jj >= P;
            const int la = min(2 * p, KL - 1), lb = min(2 * p + 1, KL - 1);
            // (readfirstlane keeps "select of two array loads" from becoming "load at a selected index" = scratch)
            auto u = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
            return LvlC{hi ? lb : la, hi ? u(Hs[lb]) : u(Hs[la]), hi ? u(Ws[lb]) : u(Ws[la]), hi ? u(sts[lb]) : u(sts[la]),
                        hi ? Wn::wh(lb) - 1 : Wn::wh(la) - 1, hi ? Wn::ww(lb) - 1 : Wn::ww(la) - 1,
                        hi ? Wn::ww(lb) : Wn::ww(la), hi ? Wn::row0(lb) : Wn::row0(la)};
        }
    };

    char *const orec = recs + oc * kOctBytes;             // this octet's records
    const char *const wbase = lds + j8 * 16;

    const int Hb_k = Hb, Wb_k = Wb, nrx_k = nrx, nregions = nry * nrx;
    // MASK: a level whose mask is not a summarised band (MaskExt < 0) is handled through its BYTES -- loads inside branches of the staging, the
    // level-0 records and the out-of-window loop.  Never taken for the masks DETR builds, those branches still cost the masked kernel 5 % (190
    // -> 181 us, round 6: every basic-block boundary around a load is a full s_waitcnt, which also drains the prefetched sampling data and the
    // level-0 corner loads in flight).  So the region loop exists twice: without any byte path when all levels of my image are summarised
    // (workgroup-uniform, known before the first region), with them otherwise.
    auto run_regions = [&](auto bytes_tag) {
    constexpr bool kBytes = MASK && decltype(bytes_tag)::value;
    for (int reg = slot0; reg < nregions; reg += regions_bound) {
        int Hb_r = Hb_k, Wb_r = Wb_k, nrx_r = nrx_k;      // TUNE + 800: ... and the reciprocals of the divisions by these
        if ((TUNE / 100) & 8) asm volatile("" : "+s"(Hb_r), "+s"(Wb_r), "+s"(nrx_r));
        const int Hb = Hb_r, Wb = Wb_r, nrx = nrx_r;
        // BALANCED tiling: the grid's ceil(H / RTH) x ceil(W / RTW) regions share the rows / columns evenly (heights differ by at most
        // one row, none exceeds RTH: what the windows are sized for) instead of leaving a sliver at the bottom and right edges -- a
        // 100-row level is 4 x 25 or 5 x 20 rows, not 4 x 24 + 4
        const int nry_b = (Hb + RTH - 1) / RTH, ry_b = reg / nrx, rx_b = reg - ry_b * nrx;
        const int y0b = (ry_b * Hb) / nry_b, y1b = ((ry_b + 1) * Hb) / nry_b;
        const int x0b = (rx_b * Wb) / nrx, x1b = ((rx_b + 1) * Wb) / nrx;
        // TUNE + 800: the per-level lane values are rebuilt per region from their wave-uniform copies (one select per level)
        // instead of living in six registers from the kernel's first instruction on
        int lane_t = tid;
        if ((TUNE / 100) & 8) asm volatile("" : "+v"(lane_t));
        const int lane_g = lane_t & 63;
        int g_H = r_H_k, g_W = r_W_k, g_st = r_st_k, g_wh = r_wh_k, g_ww = r_ww_k, g_row0 = r_row0_k;
        if ((TUNE / 100) & 8) {
            g_H = g_W = g_wh = g_ww = 1;
            g_st = g_row0 = 0;
#pragma unroll
            for (int l = 0; l < KL; ++l)
                if (lane_g == l) {
                    g_H = __builtin_amdgcn_readfirstlane(Hs[l]);
                    g_W = __builtin_amdgcn_readfirstlane(Ws[l]);
                    g_st = __builtin_amdgcn_readfirstlane(sts[l]);
                    g_wh = Wn::wh(l);
                    g_ww = Wn::ww(l);
                    g_row0 = Wn::row0(l);
                }
        }
        const int r_H = g_H, r_W = g_W, r_st = g_st, r_wh = g_wh, r_ww = g_ww, r_row0 = g_row0;
        (void)r_row0;
        // ---- the region's queries: on level lq an exact rectangle (lanes 0 .. KL-1 of every wave work it out; the counts
        //      become wave-uniform through readlane, the rest is fetched per lane with __shfl: no LDS table, no barrier)
        int r_ylo = 0, r_xlo = 0, r_w = 1, r_cnt = 0;
        float r_invw = 1.f;
        // window origins: where the region centre maps to on each level, minus half the window
        const float pcy = 0.5f * (float)(y0b + y1b) / (float)Hb, pcx = 0.5f * (float)(x0b + x1b) / (float)Wb;      // centre of the region as it is
        const int r_wy0 = (int)floorf(pcy * (float)r_H - 0.5f) - r_wh / 2 + 1;
        const int r_wx0 = (int)floorf(pcx * (float)r_W - 0.5f) - r_ww / 2 + 1;
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
        int my_wy0[NPASS], my_wx0[NPASS];
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            my_wy0[p] = kLean ? 0 : __shfl(r_wy0, myl[p], 64);
            my_wx0[p] = kLean ? 0 : __shfl(r_wx0, myl[p], 64);
        }
        auto win_origin = [&](int p, int l, int &oy, int &ox) {      // window origin of level l = lvlc(p).l
            if constexpr (!kLean) {
                oy = my_wy0[p];
                ox = my_wx0[p];
            } else {
                const int la = min(2 * p, KL - 1), lb = min(2 * p + 1, KL - 1);
                oy = l == lb ? __builtin_amdgcn_readfirstlane(wy0[lb]) : __builtin_amdgcn_readfirstlane(wy0[la]);
                ox = l == lb ? __builtin_amdgcn_readfirstlane(wx0[lb]) : __builtin_amdgcn_readfirstlane(wx0[la]);
            }
        };

        // ---- rounds: G queries at a time, 8 lanes each
        const int nrounds_all = (nq_total + G - 1) / G;
        const int round0 = part * nrounds_all / nparts, nrounds = (part + 1) * nrounds_all / nparts;      // my share of the rounds (all of them unless the unit is split)
        auto slot_query = [&](int s) -> int {      // s-th query of the region (levels in order) or -1
            const bool ok = s < nq_total;
            int lq = 0;
#pragma unroll
            for (int l = 0; l < KL - 1; ++l)
                if (lq == l && s >= cnt[l]) { s -= cnt[l]; lq = l + 1; }
            const int yl = __shfl(r_ylo, lq, 64), xl = __shfl(r_xlo, lq, 64), w_ = __shfl(r_w, lq, 64);
            const int st_ = __shfl(r_st, lq, 64), W_ = __shfl(r_W, lq, 64);
            // s / w_ through the reciprocal: s < 2^15 and (s + 0.5) / w_ is at least 0.5 / w_ away from an integer
            const int dy = (int)(((float)s + 0.5f) * __shfl(r_invw, lq, 64));
            return ok ? st_ + (yl + dy) * W_ + xl + (s - dy * w_) : -1;
        };
        // raw sample data of a round: loaded one round ahead of its use
        // (TUNE + 1600: the loads' results stay as they arrive and the location arithmetic of the fused prologue runs at the START of
        //  the round that uses them -- done where the loads are issued it waits for them a round early)
        constexpr bool kSplitLoad = ((TUNE / 100) & 16) != 0;
        typename IO::RawXY rr[NPASS];
        float rx[NPASS], ry[NPASS], ra[NPASS];
        float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
        int q = slot_query(round0 * G + (lane_t >> 3));
        auto load_round = [&](int qq, int j8) {      // (j8: the caller's copy -- the lean builds rebuild it per round)
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const int k = j8 + 8 * p;
                rx[p] = ry[p] = ra[p] = 0.f;
                if ((SEMIDETR_BRFREE && kView && KLP % 8 == 0) || (qq >= 0 && k < KLP)) {
                    // (kView: inside my image, 32-bit index arithmetic on the image's view of the tensors)
                    const int qc = (SEMIDETR_BRFREE && kView) ? max(qq, 0) : qq;
                    const typename std::conditional<kView, unsigned, int64_t>::type nq = kView ? (int64_t)qc : (int64_t)n * Lq + qc, row = nq * M + m;
                    if constexpr (kSplitLoad) {
                        rr[p] = iov.load_xy_raw(row, nq, KLP, k, k / P);
                    } else {
                        const LvlC c = lvlc(p);
                        iov.load_xy(row, nq, KLP, k, c.l, P, c.H, c.W, rx[p], ry[p]);
                    }
                    ra[p] = iov.load_w(row, KLP, k);
                }
            }
            if (GATHER && qq >= 0)
                go = *reinterpret_cast<const float4 *>(gout + (((int64_t)n * Lq + qq) * M + m) * kD + 4 * j8);
        };
        load_round(q, lane_t & 7);

        lap(0);                            // 0: region set-up (rectangles, round-0 loads issued)
        // ---- stage the windows through registers: all loads of a thread first, then its stores
        {
            constexpr int RPS = NT / 8;                                   // window rows covered per step (8 lanes per row)
            constexpr int kMaxSteps = (Wn::zrow + RPS - 1) / RPS + KL;
            float4 sv[kMaxSteps];
            unsigned smk[MASK ? kMaxSteps : 1];      // MASK: the staged rows' padding bytes, loaded beside the rows (no dependent load)
            int tids = tid;    // lean builds: the per-thread window coordinates below are rebuilt per region, not kept in registers
            if (kLean) asm volatile("" : "+v"(tids));
            const int ocs = tids >> 3, j8s = tids & 7;
            const unsigned lane_bs = (unsigned)(m * kD + 4 * j8s) * 4u;
            if (DBG != 2) {
                int nst = 0;
#pragma unroll
                for (int l = 0; l < KL; ++l) {
                    const int ww_ = Wn::ww(l), rows_ = Wn::rows(l);
                    int r = ocs, wy = ocs / ww_, wx = ocs - wy * ww_;        // this thread's row inside level l's window
#pragma unroll
                    for (int s = 0; s < (rows_ + RPS - 1) / RPS; ++s) {
                        const int py = wy0[l] + wy, px = wx0[l] + wx;
                        // (MASK, round 6: a summarised level's padded rows -- y >= vh or x >= vw, vh <= H, vw <= W -- are not loaded at all:
                        //  the out-of-range offset returns the zeros value.masked_fill(mask, 0) asks for, and the store phase has nothing to undo)
                        const int Hv = (MASK && ves[l] >= 0) ? (ves[l] & 0xffff) : Hs[l], Wv = (MASK && ves[l] >= 0) ? (int)((unsigned)ves[l] >> 16) : Ws[l];
                        const bool ok = r < rows_ && (unsigned)py < (unsigned)Hv && (unsigned)px < (unsigned)Wv;
                        const unsigned goff = ok ? (unsigned)(sts[l] + py * Ws[l] + px) * row_bytes + lane_bs : kOob;
                        // (a level with a summary needs no bytes: wave-uniform branch; a row that is not loaded reads byte 0 and ignores it)
                        if constexpr (MASK) {
                            smk[nst] = 0u;
                            if (kBytes && ves[l] < 0) smk[nst] = mask_n[ok ? sts[l] + py * Ws[l] + px : 0];
                        }
                        sv[nst++] = buf_ld4(vr, goff);
                        r += RPS;
                        wx += RPS % ww_;
                        wy += RPS / ww_;
                        if (wx >= ww_) { wx -= ww_; ++wy; }
                    }
                }
            }
            lap(1);                        // 1: staging loads issued
            __syncthreads();               // every wave is done with the previous region's windows
            lap(2);                        // 2: waiting for the other waves at the region boundary
            if (DBG != 2) {
                int ist = 0;
#pragma unroll
                for (int l = 0; l < KL; ++l) {
                    const int ww_ = Wn::ww(l), rows_ = Wn::rows(l);
                    int r = ocs, wy = ocs / ww_, wx = ocs - wy * ww_;
                    (void)wy; (void)wx;
#pragma unroll
                    for (int s = 0; s < (rows_ + RPS - 1) / RPS; ++s) {
                        if constexpr (MASK) {      // a padded pixel's row is staged as zeros: value.masked_fill(mask, 0).  Summarised levels never
                            // loaded theirs (above); a level whose mask has another form carries the bytes it loaded beside the rows
                            if (kBytes && ves[l] < 0 && smk[ist] != 0) sv[ist] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                        if (r < rows_) *reinterpret_cast<float4 *>(lds + (Wn::row0(l) + r) * 128 + j8s * 16) = sv[ist];
                        ++ist;
                        r += RPS;
                    }
                }
            }
            if (ocs < Wn::zrows) {
                float z = 0.f;
                if (kLean) asm volatile("" : "+v"(z));      // (a zero the compiler cannot keep in four registers for the whole kernel)
                *reinterpret_cast<float4 *>(lds + kZ0 + ocs * 128 + j8s * 16) = make_float4(z, z, z, z);
            }
            if constexpr (kTab) {      // lanes 0 .. KL-1 of the first wave hold level `lane`'s constants and this region's window origin
                if (tids < KL) {
                    ltab[tids * 3 + 0] = make_int4(r_H, r_W, r_st, r_row0);
                    ltab[tids * 3 + 1] = make_int4(r_wh - 1, r_ww - 1, r_ww, r_wy0);
                    // .w (round 6): the level's VALID extent, rows | columns << 16 -- the padding mask's summary {vh, vw} (vh <= H, vw <= W:
                    // a corner is real iff y < vh and x < vw), the level's own size without a mask; bit 31: the mask of this level is not of
                    // that form, its bytes decide.  Corner validity then costs the masked kernel what it costs the unmasked one.
                    int valid = r_H | (r_W << 16);
                    if constexpr (MASK) {
                        int ve_l = -1;
#pragma unroll
                        for (int l = 0; l < KL; ++l) ve_l = tids == l ? __builtin_amdgcn_readfirstlane(ves[l]) : ve_l;
                        valid = ve_l >= 0 ? ve_l : (valid | (int)0x80000000);
                    }
                    ltab[tids * 3 + 2] = make_int4(r_wx0, __float_as_int((float)r_H), __float_as_int((float)r_W), valid);
                }
            }
            if constexpr (kQList) {    // (every lane runs slot_query: its shuffles read lanes 0 .. KL-1)
                const int lim = min(nq_total, kRwQList);
                for (int s0 = 0; s0 < lim; s0 += NT) {
                    const int qv = slot_query(s0 + tids);
                    if (s0 + tids < lim) qlist[s0 + tids] = qv;
                }
            }
        }
        lap(3);                            // 3: windows stored (incl. the wait for the staging loads)
        __syncthreads();                   // the windows are complete
        lap(4);                            // 4: waiting for the other waves' stores

        for (int round = round0; round < nrounds; ++round) {
            // queries are dealt out in order, so a wave without one in this round has none in the later ones either: it leaves the
            // loop (no workgroup barrier inside) instead of spending issue slots on empty octets -- a region's 340 queries fill
            // 3.5 rounds of 96
            if (!__any(q >= 0)) break;
            // TUNE + 800: the thread index goes through an empty asm once per round, so that what is derived from it (octet and
            // window addresses, the lane's byte offset) is rebuilt here instead of living in registers through the whole kernel
            int tid_r = tid;
            if ((TUNE / 100) & 8) asm volatile("" : "+v"(tid_r));
            const int oc = tid_r >> 3, j8 = tid_r & 7, cls = (tid_r >> 4) & 1, lane = tid_r & 63;
            const unsigned lane_b = (unsigned)(m * kD + 4 * j8) * 4u;
            char *const orec = recs + oc * kOctBytes;
            const char *const wbase = lds + j8 * 16;
            (void)lane;
            // ---- geometry of my samples -> records
            float sx[NPASS], sy[NPASS], sa[NPASS];
            const float4 mygo = go;
            unsigned fbm = 0;                     // bit p: my sample of pass p is valid but leaves its window
            // kTab: the constants of my two levels (lanes 0-3: level 2 p, lanes 4-7: level 2 p + 1; the table is padded to an even count)
            // (each pass reads them where it starts: held for both passes from the round's top they cost 24 registers)
            int4 T0[NPASS], T1[NPASS], T2[NPASS];
            const int4 *const te = ltab + (j8 >> 2) * 3;
            auto lvlt = [&](int p) -> LvlC {      // lvlc(p) from the table
                if constexpr (kTab) return LvlC{min((j8 + 8 * p) / P, KL - 1), T0[p].x, T0[p].y, T0[p].z, T1[p].x, T1[p].y, T1[p].z, T0[p].w};
                else return lvlc(p);
            };
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                if constexpr (kTab) {
                    sx[p] = sy[p] = 0.f;          // (finished in the geometry loop, beside the table reads)
                } else if constexpr (kSplitLoad) {
                    sx[p] = sy[p] = 0.f;
                    if (q >= 0 && j8 + 8 * p < KLP) {
                        const LvlC c = lvlc(p);
                        io.finish_xy_raw(rr[p], P, c.H, c.W, sx[p], sy[p]);
                    }
                } else {
                    sx[p] = rx[p];
                    sy[p] = ry[p];
                }
                sa[p] = ra[p];
            }
            if (IO::kSoftmax) {
                // the softmax of the fused prologue over the row's L*P logits: they sit in NPASS registers of the octet's
                // 8 lanes (row_softmax expects LP consecutive lanes), so reduce by hand
                float mx = -__builtin_huge_valf();
#pragma unroll
                for (int p = 0; p < NPASS; ++p)
                    if (j8 + 8 * p < KLP) mx = fmaxf(mx, sa[p]);
                mx = fmaxf(mx, dpp_mov<0xB1>(mx));
                mx = fmaxf(mx, dpp_mov<0x4E>(mx));
                mx = fmaxf(mx, dpp_mov<0x141>(mx));
                float sum = 0.f;
#pragma unroll
                for (int p = 0; p < NPASS; ++p) {
                    sa[p] = (j8 + 8 * p < KLP) ? __expf(sa[p] - mx) : 0.f;      // as row_softmax: the kernels of one op agree
                    sum += sa[p];
                }
                sum = group8_sum(sum);
                const float inv = fast_rcp(sum);
#pragma unroll
                for (int p = 0; p < NPASS; ++p) sa[p] = sa[p] * inv;
            }
            if (sampled && round == 0 && q >= 0) {     // (workgroup-uniform branch; the first round's G queries are sample enough --
                                                       //  counting in every round made the sampled workgroups the launch's stragglers)
                int lq = 0;                        // the query's own level: levels tile [0, S) in order
#pragma unroll
                for (int l = 1; l < KL; ++l) lq = q >= sts[l] ? l : lq;
                const int Wq = __shfl(r_W, lq, 64), Hq = __shfl(r_H, lq, 64), pix = q - __shfl(r_st, lq, 64);
                const int qy = (int)(((float)pix + 0.5f) / (float)Wq), qx = pix - qy * Wq;      // pix < 2^23: exact
                const float cx = ((float)qx + 0.5f) / (float)Wq, cy = ((float)qy + 0.5f) / (float)Hq;
#pragma unroll
                for (int p = 0; p < NPASS; ++p) {
                    const LvlC c = lvlc(p);
                    if (j8 + 8 * p < KLP && c.l >= 1) {
                        float fx = sx[p], fy = sy[p];
                        if constexpr (kTab) io.finish_xy_raw(rr[p], P, c.H, c.W, fx, fy);      // (not finished yet, see above)
                        st_total += 1u;
                        st_far += (fabsf((fx - cx) * (float)c.W) > kFarPx || fabsf((fy - cy) * (float)c.H) > kFarPx) ? 1u : 0u;
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const int k = j8 + 8 * p;
                if constexpr (kTab) {
                    T0[p] = te[6 * p + 0];
                    T1[p] = te[6 * p + 1];
                    T2[p] = te[6 * p + 2];
                    if (q >= 0 && k < KLP) io.finish_xy_raw(rr[p], P, T0[p].x, T0[p].y, sx[p], sy[p]);
                }
                const LvlC c = lvlt(p);
                int c_wy0, c_wx0;
                if constexpr (kTab) {
                    c_wy0 = T1[p].w;
                    c_wx0 = T2[p].x;
                } else {
                    win_origin(p, c.l, c_wy0, c_wx0);
                }
                const float Hf = kTab ? __int_as_float(T2[p].y) : (float)c.H, Wf = kTab ? __int_as_float(T2[p].z) : (float)c.W;
                const float h = sub_rn(mul_rn(sy[p], Hf), 0.5f), w = sub_rn(mul_rn(sx[p], Wf), 0.5f);
                const bool inside = q >= 0 && k < KLP && h > -1.f && w > -1.f && h < Hf && w < Wf;
                const float h0f = floorf(h), w0f = floorf(w);
                const float lh = sub_rn(h, h0f), lw = sub_rn(w, w0f);
                const int wy = (int)h0f - c_wy0, wx = (int)w0f - c_wx0;
                const bool inwin = inside && (unsigned)wy < (unsigned)c.wh1 && (unsigned)wx < (unsigned)c.ww1;
                const bool fine = FG && c.l == 0;             // level 0 without a window: global corner offsets instead
                if (inside && !inwin && !fine) fbm |= 1u << p;
                const float a = sa[p], hh = 1.f - lh, hw = 1.f - lw;
                // reading order: column sw first (sw = parity of the top-left row ^ octet class), see the header
                const int rl = c.row0 + wy * c.ww + wx;      // window row of the top-left corner
                const int sw = (rl ^ cls) & 1;
                // first = (top, column sw), partner = (top, column !sw); the bottom rows follow at + pitch (an immediate in
                // the loop).  A sample that is not served from the window reads the zero rows (also at + its pitch).
                const unsigned first = inwin ? (unsigned)(rl + sw) * 128u : kZ0 + (unsigned)cls * 128u;
                const unsigned partner = inwin ? (unsigned)(rl + 1 - sw) * 128u : kZ0 + (unsigned)(1 - cls) * 128u;
                if (fine && k < P) {
                    // {4 corner byte offsets (kOob: outside the level / no sample), 4 weights}: the plain kernel's record
                    const int h0 = (int)h0f, w0 = (int)w0f;
                    bool top = h0 >= 0, bot = h0 + 1 <= c.H - 1, lef = w0 >= 0, rig = w0 + 1 <= c.W - 1;
                    if constexpr (MASK && kTab) {      // (rows / columns of the level that are real: four unsigned compares, mask or no mask)
                        const unsigned Hv = (unsigned)T2[p].w & 0x7fffu, Wv = ((unsigned)T2[p].w >> 16) & 0x7fffu;
                        top = (unsigned)h0 < Hv; bot = (unsigned)(h0 + 1) < Hv; lef = (unsigned)w0 < Wv; rig = (unsigned)(w0 + 1) < Wv;
                    }
                    const unsigned base = (unsigned)(c.st + h0 * c.W + w0) * row_bytes;      // may wrap for -1: unused then
                    const unsigned wrow = (unsigned)c.W * row_bytes;
                    bool c_tl = inside && top && lef, c_tr = inside && top && rig, c_bl = inside && bot && lef, c_br = inside && bot && rig;
                    if constexpr (MASK) {
                        // padded corners read as zero.  With the level's summary: two compares per corner; without: its bytes
                        // (kTab: the summary already is the valid extent above; bit 31 of the table word = this level's bytes decide)
                        const int ve0 = kTab ? 0 : ves[0], vh0 = kTab ? -1 : ext_vh(ve0), vw0 = kTab ? -1 : ext_vw(ve0);
                        bool bytes = inside;
                        if constexpr (kTab) bytes = kBytes && inside && T2[p].w < 0;
                        if (vh0 >= 0) {
                            const bool py0 = h0 >= vh0, py1 = h0 + 1 >= vh0, px0 = w0 >= vw0, px1 = w0 + 1 >= vw0;
                            c_tl = c_tl && !(py0 || px0);
                            c_tr = c_tr && !(py0 || px1);
                            c_bl = c_bl && !(py1 || px0);
                            c_br = c_br && !(py1 || px1);
                        } else if (bytes) {
                            const unsigned char *gp = mask_of_image() + (c.st + h0 * c.W + w0);
                            c_tl = c_tl && !gp[0];      // (evaluated left to right: a corner outside the level is never dereferenced)
                            c_tr = c_tr && !gp[1];
                            c_bl = c_bl && !gp[c.W];
                            c_br = c_br && !gp[c.W + 1];
                        }
                    }
                    *reinterpret_cast<uint4 *>(orec + kFineAt + k * 32) = make_uint4(
                        c_tl ? base : kOob, c_tr ? base + row_bytes : kOob, c_bl ? base + wrow : kOob, c_br ? base + wrow + row_bytes : kOob);
                    if (!GATHER)
                        *reinterpret_cast<float4 *>(orec + kFineAt + k * 32 + 16) = inside
                            ? make_float4(a * (hh * hw), a * (hh * lw), a * (lh * hw), a * (lh * lw)) : make_float4(0.f, 0.f, 0.f, 0.f);
                    else      // the gather wants the fractions and the attention weight (zeros unless the sample counts: no 0 * NaN)
                        *reinterpret_cast<float4 *>(orec + kFineAt + k * 32 + 16) =
                            inside ? make_float4(lw, lh, a, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (k < KLP && !(FG && k < P)) {
                    if constexpr (kWide) *reinterpret_cast<uint2 *>(orec + kOffAt + (k - kRec0) * 8) = make_uint2(first, partner);
                    else *reinterpret_cast<unsigned *>(orec + kOffAt + (k - kRec0) * 4) = (first >> 4) | ((partner >> 4) << 16);
                    if (!GATHER) {
                        const float wl_t = inwin ? a * (hh * hw) : 0.f, wl_b = inwin ? a * (lh * hw) : 0.f;
                        const float wr_t = inwin ? a * (hh * lw) : 0.f, wr_b = inwin ? a * (lh * lw) : 0.f;
                        // weights in reading order: top sw, top !sw, bottom sw, bottom !sw
                        *reinterpret_cast<float4 *>(orec + (k - kRec0) * 16) = sw ? make_float4(wr_t, wl_t, wr_b, wl_b)
                                                                        : make_float4(wl_t, wr_t, wl_b, wr_b);
                        if constexpr (kCompact) {
                            // a sample outside its window reads the zero rows, so its record only has to be finite: overwritten (same
                            // lane, same address, program order) with the compact form {lw, lh, a, (h0 + 1) | (w0 + 1) << 15} -- an
                            // integer below 2^30 is a finite float.  Waves without such a sample skip the branch.
                            if (inside && !inwin)
                                *reinterpret_cast<float4 *>(orec + (k - kRec0) * 16) =
                                    make_float4(lw, lh, a, __int_as_float(((int)h0f + 1) | (((int)w0f + 1) << 15)));
                        }
                    } else {
                        // a NaN location must not leak through 0 * NaN: everything zero unless served from the window
                        *reinterpret_cast<float4 *>(orec + (k - kRec0) * 16) =
                            make_float4(inwin ? lw : 0.f, inwin ? lh : 0.f, inwin ? a : 0.f, __int_as_float(sw));
                    }
                }
            }
            const int q_cur = q;
            lap(5);                                // 5: geometry + record writes (incl. the wait for the round's raw data)
            // ---- next round's raw data (its latency hides behind this round's compute)
            if (kQList && nq_total <= kRwQList) {      // (workgroup-uniform)
                const int sn = (round + 1) * G + oc;
                q = sn < nq_total ? qlist[sn] : -1;
            } else {
                q = slot_query((round + 1) * G + oc);
            }
            if ((SEMIDETR_BRFREE && kView) || round + 1 < nrounds) load_round(q, j8);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my wave's records are written

            // out-of-window samples of this wave: per octet a mask over (pass, lane)
            unsigned gmask = 0, allmask = 0;
            int n_out = 0;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const unsigned long long bal = __ballot((fbm >> p) & 1u);
                const unsigned long long balv = __ballot(q_cur >= 0 && j8 + 8 * p < KLP && !(FG && j8 + 8 * p < P));
                gmask |= ((unsigned)(bal >> (lane & 56)) & 0xffu) << (8 * p);
                allmask |= ((unsigned)(balv >> (lane & 56)) & 0xffu) << (8 * p);
                n_out += __popcll(bal);
            }
            const bool plain_round = n_out * 3 > 64 * NPASS && DBG != 4 && DBG != 8;
            if (plain_round) gmask = allmask;      // every sample of the round takes the global path below
            if (DBG == 4 || DBG == 8) gmask = 0;
            if constexpr (kCompact) {
                if (plain_round) {      // (wave-uniform, rare) the records of the samples INSIDE their windows are in window form: redo them compact
#pragma unroll
                    for (int p = 0; p < NPASS; ++p) {
                        const int k = j8 + 8 * p;
                        if (q_cur >= 0 && k < KLP && !(FG && k < P)) {
                            const int4 t0 = te[6 * p + 0];
                            const float Hf = (float)t0.x, Wf = (float)t0.y;
                            const float h = sub_rn(mul_rn(sy[p], Hf), 0.5f), w = sub_rn(mul_rn(sx[p], Wf), 0.5f);
                            const bool inside = h > -1.f && w > -1.f && h < Hf && w < Wf;
                            const float h0f = floorf(h), w0f = floorf(w);
                            *reinterpret_cast<float4 *>(orec + (k - kRec0) * 16) =
                                inside ? make_float4(sub_rn(w, w0f), sub_rn(h, h0f), sa[p], __int_as_float(((int)h0f + 1) | (((int)w0f + 1) << 15)))
                                       : make_float4(0.f, 0.f, 0.f, __int_as_float(0x40000000));      // (bit 30: no sample; the float 2.0)
                            if constexpr (kLoopAlways) {      // ... and its two window offsets at the zero rows: the window loop runs over them (adds 0 x finite)
                                const unsigned z0 = kZ0 + (unsigned)cls * 128u, z1 = kZ0 + (unsigned)(1 - cls) * 128u;
                                if constexpr (kWide) *reinterpret_cast<uint2 *>(orec + kOffAt + (k - kRec0) * 8) = make_uint2(z0, z1);
                                else *reinterpret_cast<unsigned *>(orec + kOffAt + (k - kRec0) * 4) = (z0 >> 4) | ((z1 >> 4) << 16);
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }

            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            float m_a[NPASS], m_x[NPASS], m_y[NPASS];      // GATHER: d/d attn, d/d x, d/d y of my samples
#pragma unroll
            for (int p = 0; p < NPASS; ++p) m_a[p] = m_x[p] = m_y[p] = 0.f;
            // one sample of the gather: partial dots over my 4 channels with the four corners -> the three small gradients,
            // summed over the 8 lanes of the octet; the lane that owns sample k keeps them.  V1..V4 = top-left, top-right,
            // bottom-left, bottom-right.
#define RW_DOT4(V_) (mygo.x * (V_).x + mygo.y * (V_).y + mygo.z * (V_).z + mygo.w * (V_).w)
#define RW_GATHER_STEP(P_, J_, V1_, V2_, V3_, V4_, LW_, LH_, A_)                                            \
            RW_GATHER_DOTS(P_, J_, RW_DOT4(V1_), RW_DOT4(V2_), RW_DOT4(V3_), RW_DOT4(V4_), LW_, LH_, A_)
#define RW_GATHER_DOTS(P_, J_, D1_, D2_, D3_, D4_, LW_, LH_, A_)                                               \
            do {                                                                                               \
                const float d1_ = (D1_), d2_ = (D2_), d3_ = (D3_), d4_ = (D4_);                                \
                const float hh_ = 1.f - (LH_), hw_ = 1.f - (LW_);                                              \
                const float t_ = hw_ * d1_ + (LW_) * d2_, b_ = hw_ * d3_ + (LW_) * d4_;                        \
                float pa_ = hh_ * t_ + (LH_) * b_;                                                             \
                float px_ = (A_) * (hh_ * (d2_ - d1_) + (LH_) * (d4_ - d3_));                                  \
                float py_ = (A_) * (b_ - t_);                                                                  \
                pa_ = group8_sum(pa_);                                                                         \
                px_ = group8_sum(px_);                                                                         \
                py_ = group8_sum(py_);                                                                         \
                const bool own_ = j8 == (J_);                                                                  \
                m_a[P_] = own_ ? pa_ : m_a[P_];                                                                \
                m_x[P_] = own_ ? px_ : m_x[P_];                                                                \
                m_y[P_] = own_ ? py_ : m_y[P_];                                                                \
            } while (0)
            // publish sample k's global corner offsets + geometry in slot `sl` of the octet (owner lane only)
            auto publish = [&](bool act, int k, int sl) {
                MaskExt pme = MaskExt{-1};      // MASK: the summary of sample k's level (fetched with every lane active: a
                if constexpr (MASK) {               //       shuffle inside the owner-lane branch would read disabled lanes)
                    const int lv = min(k / P, KL - 1);
                    int ve = ves[0];
#pragma unroll
                    for (int l = 1; l < KL; ++l) ve = lv == l ? __builtin_amdgcn_readfirstlane(ves[l]) : ve;      // (readfirstlane: a select chain over a
                                                                                                      //  small array otherwise becomes an indexed SCRATCH array)
                    pme = MaskExt{ve};
                }
                if (act && j8 == (k & 7)) {
                    const int p = k >> 3;
                    unsigned off[4];
                    float lw, lh;
                    float px_ = sx[0], py_ = sy[0], a_ = sa[0];
                    int H_, W_, st_;
                    if constexpr (kTab) {      // (the table entry of sample k's level: one read; a select over the passes' copies becomes a scratch array)
                        const int4 t0 = ltab[min(k / P, KL - 1) * 3];
                        H_ = t0.x; W_ = t0.y; st_ = t0.z;
#pragma unroll
                        for (int pp = 1; pp < NPASS; ++pp)
                            if (p == pp) { px_ = sx[pp]; py_ = sy[pp]; a_ = sa[pp]; }
                    } else {
                        const LvlC c0 = lvlc(0);
                        H_ = c0.H; W_ = c0.W; st_ = c0.st;
#pragma unroll
                        for (int pp = 1; pp < NPASS; ++pp)
                            if (p == pp) {
                                const LvlC c = lvlc(pp);
                                px_ = sx[pp]; py_ = sy[pp]; a_ = sa[pp]; H_ = c.H; W_ = c.W; st_ = c.st;
                            }
                    }
                    sample_setup_oob(px_, py_, H_, W_, st_, row_bytes, off, lw, lh);
                    if constexpr (MASK) mask_corners_oob(io, pme, n, px_, py_, H_, W_, st_, off);
                    *reinterpret_cast<uint4 *>(orec + kSlotAt + 32 * sl) = make_uint4(off[0], off[1], off[2], off[3]);
                    *reinterpret_cast<float4 *>(orec + kSlotAt + 32 * sl + 16) = make_float4(lw, lh, a_, 0.f);
                }
            };
            // consume a published sample: accumulate (forward) / take its three gradients (gather)
            auto consume_fwd = [&](const float4 &g, const float4 &v1, const float4 &v2, const float4 &v3, const float4 &v4) {
                const float hh = 1.f - g.y, hw = 1.f - g.x;
                const float w1 = g.z * (hh * hw), w2 = g.z * (hh * g.x), w3 = g.z * (g.y * hw), w4 = g.z * (g.y * g.x);
                acc.x = fmaf(w4, v4.x, fmaf(w3, v3.x, fmaf(w2, v2.x, fmaf(w1, v1.x, acc.x))));
                acc.y = fmaf(w4, v4.y, fmaf(w3, v3.y, fmaf(w2, v2.y, fmaf(w1, v1.y, acc.y))));
                acc.z = fmaf(w4, v4.z, fmaf(w3, v3.z, fmaf(w2, v2.z, fmaf(w1, v1.z, acc.z))));
                acc.w = fmaf(w4, v4.w, fmaf(w3, v3.w, fmaf(w2, v2.w, fmaf(w1, v1.w, acc.w))));
            };
            // ---- the first two out-of-window samples of every octet: corner loads issued NOW, consumed after the LDS loop
            //      (their latency hides behind it); further ones take the loop at the end
            constexpr int kPre = TUNE % 10, kSB = (TUNE / 10) % 10;
            bool pact[kPre > 0 ? kPre : 1];
            int pk[kPre > 0 ? kPre : 1];
            float4 pg[kPre > 0 ? kPre : 1], pv[kPre > 0 ? kPre : 1][4];
            const bool any_out = kPre > 0 && !plain_round && __any(gmask != 0);
            if (any_out) {
#pragma unroll
                for (int i = 0; i < kPre; ++i) {
                    pact[i] = gmask != 0;
                    pk[i] = pact[i] ? __ffs((int)gmask) - 1 : 0;
                    gmask &= gmask - 1;
                    publish(pact[i], pk[i], i);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < kPre; ++i) {
                    const uint4 o = *reinterpret_cast<const uint4 *>(orec + kSlotAt + 32 * i);
                    pg[i] = *reinterpret_cast<const float4 *>(orec + kSlotAt + 32 * i + 16);
                    // an octet without such a sample reads out of range: zeros, no memory access
                    pv[i][0] = buf_ld4(vr, (pact[i] ? o.x : kOob) + lane_b);
                    pv[i][1] = buf_ld4(vr, (pact[i] ? o.y : kOob) + lane_b);
                    pv[i][2] = buf_ld4(vr, (pact[i] ? o.z : kOob) + lane_b);
                    pv[i][3] = buf_ld4(vr, (pact[i] ? o.w : kOob) + lane_b);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // slots read before anything rewrites them
            }
            // ---- level 0 without a window: its 4 x 4 corner rows come through the vector-memory path, issued now and
            //      consumed after the LDS loop (the two pipes work side by side)
            constexpr int kFineN = (TUNE / 100) & 1 ? 1 : 2;      // level-0 samples in flight at a time (registers: 20 each)
            constexpr int kFineGroups = P / kFineN, kFineStep = (KLP - P) / kFineGroups > 0 ? (KLP - P) / kFineGroups : 1;
            float4 fv[FG ? kFineN : 1][4], fw[FG ? kFineN : 1];
            int fine_cur = 0;                               // first level-0 sample of the group in flight
            auto fine_issue = [&](int k0) {
                fine_cur = k0;
#pragma unroll
                for (int i = 0; i < kFineN; ++i) {
                    const uint4 o = *reinterpret_cast<const uint4 *>(orec + kFineAt + (k0 + i) * 32);
                    fw[i] = *reinterpret_cast<const float4 *>(orec + kFineAt + (k0 + i) * 32 + 16);
                    fv[i][0] = buf_ld4(vr, o.x + lane_b);
                    fv[i][1] = buf_ld4(vr, o.y + lane_b);
                    fv[i][2] = buf_ld4(vr, o.z + lane_b);
                    fv[i][3] = buf_ld4(vr, o.w + lane_b);
                }
            };
            auto fine_consume = [&]() {
#pragma unroll
                for (int i = 0; i < kFineN; ++i) {
                    if constexpr (GATHER) {      // corners as loaded: top-left, top-right, bottom-left, bottom-right; sample fine_cur + i of pass 0
                        RW_GATHER_STEP(0, fine_cur + i, fv[i][0], fv[i][1], fv[i][2], fv[i][3], fw[i].x, fw[i].y, fw[i].z);
                        continue;
                    }
                    if constexpr (((TUNE / 100) & 32) != 0) {
                        rw_fma4(acc, fw[i], fv[i][0], fv[i][1], fv[i][2], fv[i][3]);
                        continue;
                    }
                    acc.x = fmaf(fw[i].w, fv[i][3].x, fmaf(fw[i].z, fv[i][2].x, fmaf(fw[i].y, fv[i][1].x, fmaf(fw[i].x, fv[i][0].x, acc.x))));
                    acc.y = fmaf(fw[i].w, fv[i][3].y, fmaf(fw[i].z, fv[i][2].y, fmaf(fw[i].y, fv[i][1].y, fmaf(fw[i].x, fv[i][0].y, acc.y))));
                    acc.z = fmaf(fw[i].w, fv[i][3].z, fmaf(fw[i].z, fv[i][2].z, fmaf(fw[i].y, fv[i][1].z, fmaf(fw[i].x, fv[i][0].z, acc.z))));
                    acc.w = fmaf(fw[i].w, fv[i][3].w, fmaf(fw[i].z, fv[i][2].w, fmaf(fw[i].y, fv[i][1].w, fmaf(fw[i].x, fv[i][0].w, acc.w))));
                }
            };
            constexpr bool kNoFine = DBG == 5 || DBG == 7 || DBG == 8;
            constexpr bool kNoLoop = DBG == 3 || DBG == 7 || DBG == 8;
            if (FG && !kNoFine) fine_issue(0);
            lap(6);                                // 6: next round's loads issued, masks, pre-issued corner loads
            if ((kLoopAlways || !plain_round) && !kNoLoop) {
                // ---- the common case: every corner from LDS, in the order (top, sw), (top, !sw), (bottom, sw), (bottom, !sw)
#pragma unroll
                for (int k = FG ? P : 0; k < KLP; ++k) {
                    const int pitch = Wn::ww(k / P) * 128;
                    const float4 r = *reinterpret_cast<const float4 *>(orec + (k - kRec0) * 16);
                    const unsigned o = kWide ? 0u : *reinterpret_cast<const unsigned *>(orec + kOffAt + (k - kRec0) * 4);
                    const char *a0, *a1;
                    if constexpr (kWide) {
                        const uint2 ow = *reinterpret_cast<const uint2 *>(orec + kOffAt + (k - kRec0) * 8);
                        a0 = wbase + ow.x;
                        a1 = wbase + ow.y;
                    } else if constexpr (((TUNE / 100) & 32) != 0) {
                        // TUNE + 3200: window address = base + 16 x (16-bit offset) as ONE v_mad_u32_u16 each (op_sel picks the half of
                        // the packed word) instead of shift / mask / add: 2 instead of 5 VALU instructions per sample
                        unsigned u0, u1;
                        typedef const __attribute__((address_space(3))) char *lds_cptr;      // 32-bit LDS addresses, no base to add
                        const unsigned jb = (unsigned)(uintptr_t)(lds_cptr)wbase;
                        asm("v_mad_u32_u16 %0, %1, 16, %2" : "=v"(u0) : "v"(o), "v"(jb));
                        asm("v_mad_u32_u16 %0, %1, 16, %2 op_sel:[1,0,0,0]" : "=v"(u1) : "v"(o), "v"(jb));
                        a0 = (const char *)(lds_cptr)(uintptr_t)u0;
                        a1 = (const char *)(lds_cptr)(uintptr_t)u1;
                    } else {
                        a0 = wbase + ((o & 0xffffu) << 4);
                        a1 = wbase + ((o >> 16) << 4);
                    }
                    const float4 f1 = *reinterpret_cast<const float4 *>(a0);
                    const float4 f2 = *reinterpret_cast<const float4 *>(a1);
                    const float4 f3 = *reinterpret_cast<const float4 *>(a0 + pitch);
                    const float4 f4 = *reinterpret_cast<const float4 *>(a1 + pitch);
                    if (!GATHER) {
                        if constexpr (((TUNE / 100) & 32) != 0) {
                            rw_fma4(acc, r, f1, f2, f3, f4);
                        } else {
                            acc.x = fmaf(r.w, f4.x, fmaf(r.z, f3.x, fmaf(r.y, f2.x, fmaf(r.x, f1.x, acc.x))));
                            acc.y = fmaf(r.w, f4.y, fmaf(r.z, f3.y, fmaf(r.y, f2.y, fmaf(r.x, f1.y, acc.y))));
                            acc.z = fmaf(r.w, f4.z, fmaf(r.z, f3.z, fmaf(r.y, f2.z, fmaf(r.x, f1.z, acc.z))));
                            acc.w = fmaf(r.w, f4.w, fmaf(r.z, f3.w, fmaf(r.y, f2.w, fmaf(r.x, f1.w, acc.w))));
                        }
                        if (k % kSB == kSB - 1) __builtin_amdgcn_sched_barrier(0);      // bounds the registers of the unrolled loop
                    } else if (FG) {
                        // (reading order -> corner order on the four DOTS, not on the sixteen row values)
                        const bool sw = __float_as_int(r.w) != 0;
                        const float e1 = RW_DOT4(f1), e2 = RW_DOT4(f2), e3 = RW_DOT4(f3), e4 = RW_DOT4(f4);
                        RW_GATHER_DOTS(k >> 3, k & 7, sw ? e2 : e1, sw ? e1 : e2, sw ? e4 : e3, sw ? e3 : e4, r.x, r.y, r.z);
                        if (k % kSB == kSB - 1) __builtin_amdgcn_sched_barrier(0);
                    } else {
                        const bool sw = __float_as_int(r.w) != 0;
                        const float4 vtl = sw ? f2 : f1, vtr = sw ? f1 : f2, vbl = sw ? f4 : f3, vbr = sw ? f3 : f4;
                        RW_GATHER_STEP(k >> 3, k & 7, vtl, vtr, vbl, vbr, r.x, r.y, r.z);
                        // keep the scheduler from hoisting the whole unrolled loop's LDS reads (256 VGPRs and spills otherwise)
                        if (k % (kSB / 2 > 0 ? kSB / 2 : 1) == (kSB / 2 > 0 ? kSB / 2 : 1) - 1) __builtin_amdgcn_sched_barrier(0);
                    }
                    if (FG && !kNoFine && (k - P) % kFineStep == kFineStep - 1 && (k - P) / kFineStep < kFineGroups - 1) {
                        fine_consume();      // hand-over point of the level-0 samples: the group in flight is used, the next one issued
                        fine_issue(((k - P) / kFineStep + 1) * kFineN);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (FG && !kNoFine) {
                if ((!kLoopAlways && plain_round) || kNoLoop) {      // the LDS loop (and its hand-over points) was skipped
#pragma unroll
                    for (int g = 1; g < kFineGroups; ++g) { fine_consume(); fine_issue(g * kFineN); }
                }
                fine_consume();
            }
            lap(7);                                // 7: compute loop (LDS)
            if (any_out) {
#pragma unroll
                for (int i = 0; i < kPre; ++i) {
                    if (!GATHER) {
                        if (pact[i]) consume_fwd(pg[i], pv[i][0], pv[i][1], pv[i][2], pv[i][3]);
                    } else if (pact[i]) {
#pragma unroll
                        for (int pp = 0; pp < NPASS; ++pp)
                            if ((pk[i] >> 3) == pp)
                                RW_GATHER_STEP(pp, pk[i] & 7, pv[i][0], pv[i][1], pv[i][2], pv[i][3], pg[i].x, pg[i].y, pg[i].z);
                    }
                }
            }
            // ---- what is left (all samples in a plain round): per trip every octet takes its next such sample, the owner
            //      lane publishes its corner offsets and geometry, the octet loads the four corners
            while (__any(gmask != 0)) {
                // kTrip such samples per octet and trip (one slot each): 4 * kTrip corner loads in flight per lane, as in the
                // plain kernels -- a trip costs one global round trip whatever it carries
                constexpr int kTrip = ((TUNE / 100) & 4) ? 1 : 2;      // <= the octet's slots (four per trip measured no better: 222 vs 216 us; TUNE + 400: one, 20 VGPRs less)
                bool act2[kTrip];
                int k2[kTrip];
                if constexpr (kCompact) {
                    // the sample's compact record + its level's table entry -> four corner offsets, on all eight lanes alike
                    float4 g2[kTrip], v2[kTrip][4];
                    int4 tl[kTrip];
#pragma unroll
                    for (int i = 0; i < kTrip; ++i) {
                        act2[i] = gmask != 0;
                        k2[i] = act2[i] ? __ffs((int)gmask) - 1 : kRec0;
                        gmask &= gmask - 1;
                        g2[i] = *reinterpret_cast<const float4 *>(orec + (k2[i] - kRec0) * 16);
                        tl[i] = ltab[(k2[i] / P) * 3];
                        if constexpr (MASK) tl[i].w = ltab[(k2[i] / P) * 3 + 2].w;      // (the level's valid extent, see the table)
                    }
#pragma unroll
                    for (int i = 0; i < kTrip; ++i) {
                        const int pk = __float_as_int(g2[i].w);
                        const int h0 = (pk & 0x7fff) - 1, w0 = ((pk >> 15) & 0x7fff) - 1, H_ = tl[i].x, W_ = tl[i].y;
                        const bool ok = act2[i] && !(pk >> 30);
                        bool top = h0 >= 0, bot = h0 + 1 <= H_ - 1, lef = w0 >= 0, rig = w0 + 1 <= W_ - 1;
                        if constexpr (MASK) {
                            const unsigned Hv = (unsigned)tl[i].w & 0x7fffu, Wv = ((unsigned)tl[i].w >> 16) & 0x7fffu;
                            top = (unsigned)h0 < Hv; bot = (unsigned)(h0 + 1) < Hv; lef = (unsigned)w0 < Wv; rig = (unsigned)(w0 + 1) < Wv;
                        }
                        bool c_tl = ok && top && lef, c_tr = ok && top && rig, c_bl = ok && bot && lef, c_br = ok && bot && rig;
                        const int pix = tl[i].z + h0 * W_ + w0;
                        if constexpr (MASK) {      // padded corners read as zero: the valid extent above, or (bit 31) the level's bytes
                            if (kBytes && tl[i].w < 0 && ok) {
                                const unsigned char *gp = mask_of_image() + pix;
                                c_tl = c_tl && !gp[0];      // (left to right: a corner outside the level is never dereferenced)
                                c_tr = c_tr && !gp[1];
                                c_bl = c_bl && !gp[W_];
                                c_br = c_br && !gp[W_ + 1];
                            }
                        }
                        const unsigned base = (unsigned)pix * row_bytes + lane_b, wrow = (unsigned)W_ * row_bytes;      // (may wrap for -1: unused then)
                        v2[i][0] = buf_ld4(vr, c_tl ? base : kOob);
                        v2[i][1] = buf_ld4(vr, c_tr ? base + row_bytes : kOob);
                        v2[i][2] = buf_ld4(vr, c_bl ? base + wrow : kOob);
                        v2[i][3] = buf_ld4(vr, c_br ? base + wrow + row_bytes : kOob);
                    }
#pragma unroll
                    for (int i = 0; i < kTrip; ++i)
                        if (act2[i]) consume_fwd(g2[i], v2[i][0], v2[i][1], v2[i][2], v2[i][3]);
                    continue;
                }
#pragma unroll
                for (int i = 0; i < kTrip; ++i) {
                    act2[i] = gmask != 0;
                    k2[i] = act2[i] ? __ffs((int)gmask) - 1 : 0;
                    gmask &= gmask - 1;
                    publish(act2[i], k2[i], i);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float4 g2[kTrip], v2[kTrip][4];
#pragma unroll
                for (int i = 0; i < kTrip; ++i) {
                    const uint4 o = *reinterpret_cast<const uint4 *>(orec + kSlotAt + 32 * i);
                    g2[i] = *reinterpret_cast<const float4 *>(orec + kSlotAt + 32 * i + 16);
                    v2[i][0] = buf_ld4(vr, (act2[i] ? o.x : kOob) + lane_b);
                    v2[i][1] = buf_ld4(vr, (act2[i] ? o.y : kOob) + lane_b);
                    v2[i][2] = buf_ld4(vr, (act2[i] ? o.z : kOob) + lane_b);
                    v2[i][3] = buf_ld4(vr, (act2[i] ? o.w : kOob) + lane_b);
                }
#pragma unroll
                for (int i = 0; i < kTrip; ++i) {
                    if (!act2[i]) continue;
                    if (!GATHER) {
                        consume_fwd(g2[i], v2[i][0], v2[i][1], v2[i][2], v2[i][3]);
                    } else {
                        // k is a runtime value here: one instantiation per pass
#pragma unroll
                        for (int pp = 0; pp < NPASS; ++pp)
                            if ((k2[i] >> 3) == pp) RW_GATHER_STEP(pp, k2[i] & 7, v2[i][0], v2[i][1], v2[i][2], v2[i][3], g2[i].x, g2[i].y, g2[i].z);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // slots read before the next trip rewrites them
            }
            lap(8);                                // 8: out-of-window samples
            if (DBG == 1 && tid == 0) SEMIDETR_DBG_ADD(10, 1);
            // ---- results
            if (!GATHER) {
                if (q_cur >= 0) {
                    if constexpr (kView) st_stream4(out_n + (size_t)((unsigned)(q_cur * M + m) * (unsigned)kD + 4u * (unsigned)j8), acc);
                    else st_stream4(out + (((int64_t)n * Lq + q_cur) * M + m) * kD + 4 * j8, acc);
                }
            } else {
                float dot = 0.f;                   // fused epilogue: sum_k a_k g_k over the row
                if (IO::kSoftmax) {
#pragma unroll
                    for (int p = 0; p < NPASS; ++p) dot += (j8 + 8 * p < KLP) ? sa[p] * m_a[p] : 0.f;
                    dot = group8_sum(dot);
                }
                if (q_cur >= 0) {
                    const int64_t nq_c = (int64_t)n * Lq + q_cur, row_c = nq_c * M + m;
#pragma unroll
                    for (int p = 0; p < NPASS; ++p) {
                        const int k = j8 + 8 * p;
                        if (k < KLP) {
                            const LvlC c = lvlc(p);
                            const float4 res = make_float4(m_a[p], m_x[p] * (float)c.W, m_y[p] * (float)c.H, sa[p]);
                            io.store_with_dot(row_c, nq_c, KLP, k, c.l, P, c.H, c.W, res, dot);
                        }
                    }
                }
            }
#undef RW_GATHER_STEP
#undef RW_GATHER_DOTS
#undef RW_DOT4
            lap(9);                                // 9: results
        }
        if (DBG == 1 && tid == 0) SEMIDETR_DBG_ADD(11, 1);
    }
    };
    if constexpr (MASK && kTab) {
        bool any_bytes = false;
#pragma unroll
        for (int l = 0; l < KL; ++l) any_bytes = any_bytes || ves[l] < 0;
        if (any_bytes) run_regions(std::true_type());
        else run_regions(std::false_type());
    } else if constexpr (MASK) {      // (the five-level instantiation keeps ONE loop with the byte paths: twice it spilled at its 128 registers)
        run_regions(std::true_type());
    } else {
        run_regions(std::false_type());
    }
    if (sampled) fwd_stats_add(fs, st_far, st_total, 2u);
    // the launch's first workgroup hands the previous launch's counts to the host when it is done (see msda_fwd_d32)
    if (!GATHER && fs.on() && b == 0 && tid == 0) fwd_stats_publish(fs);
}
