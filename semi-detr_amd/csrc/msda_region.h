// grad_value for encoder self-attention, REGION-owned windowed scatter (fp32, D == 32, num_point == 4).  Included by
// msda.hip after msda_fast.h; the successor of msda_bwd_scatter_d32_win.
//
// The windowed kernel takes a 16 x 16 patch of queries of ONE level per workgroup.  Its row atomics (4.7 M per bs-4
// launch = 454 us of the atomic unit, the kernel's binding resource at 552 us) have two avoidable parts:
//   * every query level flushes its own copy of the rows it shares with the other levels' queries of the same image
//     region (a level-1 patch covers the area of four level-0 patches and flushes the same level-0 .. level-3 rows again);
//   * the footprint of a coarse-level patch on the fine levels is larger than the 32 x 32 window, so 19 % of all atomics
//     are window misses scattered one by one.
// Here a workgroup owns a REGION of the image -- a 16 x 16 tile of the finest level -- and takes the queries of ALL levels
// whose pixel centres lie in it (256 + 64 + 16 + 4 for a halving pyramid, <= kRegQ).  Their samples on level l all fall
// around the same spot, so ONE window per sampling level serves every query level: the rows are flushed once per region
// instead of once per (query level, patch), and coarse-level queries no longer miss.  Everything else is the windowed
// kernel's machinery: grad_out of the region's queries staged in LDS, (sample, corner) pairs bucketed by window row with
// integer LDS atomics (count -> scan -> fill), 64 streams of 16 lanes walking equal shares of the row-sorted entries with
// the row sum in registers, one full-line atomic per row run, misses scattered one by one (any sampling pattern is
// correct).
// A region with more than kRegQ queries (pyramids whose coarse levels are not much smaller) is processed in several
// passes over slices of its query list.
#pragma once

// NT threads; kRegQ queries per pass; region = RTH x RTW pixels of the finest level; WH x WW window per sampling level.
// Product (round 5): <512, 256, 8, 24, 24, 40> -- 8 x 24 regions, two workgroups per CU (74.5 KB of LDS each; the phases of one hide the
// barriers of the other).  Rounds 2-4 ran <512, 208 / 176, 8, 16, 24, 32> at two / three per CU: the optimum while the walk's compute
// was a co-limit; with the cheap flush path the atomic unit binds and fewer flushed rows win (msda.hip SEMIDETR_SCATTER_*).
template <int NT, int kRegQ, int WH, int WW, int FUSE = 0>
constexpr size_t reg_lds_bytes()
{
    return (size_t)(kRegQ * kPT * 4 + 8) * 8 + (FUSE ? (size_t)(kRegQ * kPT * 4 + 8) * 4 : 0) + (size_t)kRegQ * kD * 4 +
           (size_t)2 * WH * WW * 4 + (size_t)kRegQ * 4 + 8 * 4 + (NT / 64) * 4 + 4 * kMaxLevels * 4 + 64;
}

// AID (experiments build only): timing aids, results wrong -- 1: no dot phase, 2: no row atomics, 4: no result stores,
// 8: dot phase without its global loads, 16: dot phase without its LDS reads
template <typename IO, int NT, int kRegQ, int RTH, int RTW, int WH, int WW, int DBG = 0, int WU = 8, int FUSE = 0, int AID = 0>
__device__ __forceinline__ void reg_scatter_body(
    const int b, float4 *smem, const float *__restrict__ gout, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int regions_bound, float *__restrict__ gvalue,
    const float *__restrict__ value = nullptr)
{
    io.same_dims(S, M, L);
    constexpr int kWR = WH * WW, kNE = kRegQ * kPT * 4, SPT = (kRegQ * kPT + NT - 1) / NT, KC = (kWR + NT - 1) / NT;
    // WU >= 1000: PAIRED corners.  The left and the right corner of a sample's top (bottom) edge land on NEIGHBOURING window rows, so
    // one 16-byte entry {w_left, w_right, word} bucketed by the LEFT row serves both: half the entries to count, fill and walk; the
    // walk keeps the sums of rows r and r + 1 and hands the second one on when the next run is row r + 1 (flag in the run's last entry),
    // so rows are flushed as often as before.  A corner whose partner is missing (level border) or outside the window goes to the miss
    // list on its own, as out-of-window corners always did.
    constexpr bool kPair = WU >= 1000;
    static_assert(!kPair || FUSE == 0, "the fused-backward experiment indexes one entry per corner");
    // round 5: the entry word of the plain walk carries the destination as a PIXEL OFFSET from the window's first pixel (15 bits:
    // (window row) * W + column) instead of the window row index the counting sort uses -- the flush then is one bit-field extract and
    // one 24-bit multiply-add onto a per-level scalar base (global_atomic saddr form) instead of extract / multiply / extract / add /
    // 64-bit multiply-add / 64-bit shift-add inside the divergent branch that 46 % of the walk's steps take.  A level too wide for
    // 15 bits ((WH - 1) * W + WW > 32768, i.e. W > 1423) gets no window: all its corners take the one-by-one path (lvl_win below).
    // (The paired and fused experiments decode window rows and keep the old word.)  Measured (same box, tools/r05_ab_kern.sh): the
    // kernel without its row atomics 382 -> 328 us, with them 404 -> 398 us: it now runs at 89 % of what the atomic unit delivers
    // (3.68 M rows at 10.4 G rows/s = 354 us) -- only fewer flushed rows make it faster.
    constexpr bool kPixKey = !kPair && FUSE == 0;
    static_assert(kRegQ <= 512 && kWR <= (1 << 14), "entry packing: 9 bits query slot (x 128), 14 bits window row, sign bit = last");
    float2 *entries = reinterpret_cast<float2 *>(smem);   // [kNE + 8] front: bucketed {weight, last << 31 | window row << 16 | slot << 7};
                                                          // back: misses {weight, slot << 23 | pixel index}
    // FUSE: [kNE + 8] <grad_out row, value row> of every entry (dot phase -> combine), same indexing as `entries`
    float *dvals = reinterpret_cast<float *>(entries + kNE + 8);
    float *gtile = dvals + (FUSE ? kNE + 8 : 0);                          // [kRegQ * kD] grad_out rows of the region's queries
    int *cnt = reinterpret_cast<int *>(gtile + kRegQ * kD);
    int *start = cnt + kWR;
    int *qlist = start + kWR;                                             // [kRegQ] query index of every slot
    int (*stats2)[4] = reinterpret_cast<int (*)[4]>(qlist + kRegQ);
    int *wsum = reinterpret_cast<int *>(stats2 + 2);                      // [NT / 64]
    int *lv = wsum + NT / 64;                                             // [4][kMaxLevels]: y_lo, x_lo, width, slot offset

    constexpr int P = kPT;
    const int Lq = S, LP = L * P, rs = M * kD;
    const int m = (b % M + (b / M) / kScatterHeadRun) % M;
    const int slot0 = (b / M) % regions_bound, n = (b / M) / regions_bound;
    // the thread index is rebuilt where it is needed (wave number: wave-uniform, lives in a scalar register; lane number: two VALU) so
    // that neither it nor anything derived from it has to stay in a vector register across the whole kernel (see the region loop)
    const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    auto fresh_tid = [&]() {
        unsigned ones = ~0u;
        asm volatile("" : "+s"(ones));      // (an input the compiler cannot see through: the sum is not hoisted out of the loops and spilled)
        return wave_s * 64 + (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
    };
    const int tid = fresh_tid();

    // the finest level (most pixels) carries the region grid
    int lb = 0, Hb = (int)shapes[0], Wb = (int)shapes[1];
    for (int l = 1; l < L; ++l) {
        const int h = (int)shapes[2 * l], w = (int)shapes[2 * l + 1];
        if (h * w > Hb * Wb) { lb = l; Hb = h; Wb = w; }
    }
    (void)lb;
    const int nry = (Hb + RTH - 1) / RTH, nrx = (Wb + RTW - 1) / RTW;

    unsigned long long tmark = DBG ? __builtin_readcyclecounter() : 0ull;      // instrumented build: see msda_dest.h
    (void)tmark;                                                               // (the product build's SEMIDETR_DBG_ADD is empty)
    auto lap = [&](int slot_) {
        if (DBG && tid == 0) {
            const unsigned long long now = __builtin_readcyclecounter();
            SEMIDETR_DBG_ADD(slot_, now - tmark);
            tmark = now;
        }
    };
    const int Hb_k = Hb, Wb_k = Wb, nrx_k = nrx, nregions = nry * nrx;
    for (int reg = slot0; reg < nregions; reg += regions_bound) {
        // the thread index (again per pass below) and the region grid's sizes go through an empty asm: what is derived from them (LDS
        // addresses, piece offsets, float copies, the reciprocal of the division below) is then rebuilt per region / pass instead of
        // living in registers from the kernel's first instruction to its last (102 -> 90 VGPRs)
        int tid_r = fresh_tid(), Hb_p = Hb_k, Wb_p = Wb_k, nrx_p = nrx_k;
        asm volatile("" : "+v"(tid_r), "+s"(Hb_p), "+s"(Wb_p), "+s"(nrx_p));
        const int tid = tid_r, Hb = Hb_p, Wb = Wb_p, nrx = nrx_p;
        const int y0b = (reg / nrx) * RTH, x0b = (reg % nrx) * RTW;
        const int y1b = min(y0b + RTH, Hb), x1b = min(x0b + RTW, Wb);
        __syncthreads();                      // previous region fully done before its LDS state is reused
        // ---- the region's queries: on level lq the pixels whose centre maps into [y0b, y1b) x [x0b, x1b) of the finest
        //      level -- by(qy) = ((2 qy + 1) Hb) / (2 Hq) is monotone, so they form an exact rectangle
        if (tid < L) {
            const int Hq = (int)shapes[2 * tid], Wq = (int)shapes[2 * tid + 1];
            auto first = [](int bound, int nq_, int nb) {      // smallest q with ((2q+1) nb) / (2 nq_) >= bound
                const long long a = 2ll * nq_ * bound;
                const int cc = (int)((a + nb - 1) / nb);       // 2q + 1 >= ceil(a / nb)
                return min(nq_, cc >> 1);
            };
            const int ylo = first(y0b, Hq, Hb), yhi = y1b >= Hb ? Hq : first(y1b, Hq, Hb);
            const int xlo = first(x0b, Wq, Wb), xhi = x1b >= Wb ? Wq : first(x1b, Wq, Wb);
            lv[0 * kMaxLevels + tid] = ylo;
            lv[1 * kMaxLevels + tid] = xlo;
            lv[2 * kMaxLevels + tid] = max(xhi - xlo, 0);
            lv[3 * kMaxLevels + tid] = max(yhi - ylo, 0) * max(xhi - xlo, 0);      // count, turned into an offset below
        }
        __syncthreads();
        int nq_sum = 0;
        for (int l = 0; l < L; ++l) nq_sum += lv[3 * kMaxLevels + l];
        // (wave-uniform values that came through vector registers -- an LDS sum, two float divisions -- go back to scalar registers)
        const int nq_total = __builtin_amdgcn_readfirstlane(nq_sum);
        // region centre in normalised coordinates (pixel centres are (i + 0.5) / size)
        const float pcy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((y0b + 0.5f * RTH) / (float)Hb)));
        const float pcx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((x0b + 0.5f * RTW) / (float)Wb)));

        for (int q_base = 0; q_base < nq_total; q_base += kRegQ) {      // one pass unless the region has > kRegQ queries
            const int nq = min(kRegQ, nq_total - q_base);
            int tid_p = fresh_tid();
            asm volatile("" : "+v"(tid_p));
            const int tid = tid_p, lane = tid & 63, wv = tid >> 6;
            __syncthreads();
            for (int i = tid; i < nq; i += NT) {       // slot -> query index
                int s = q_base + i, lq = 0;
                while (s >= lv[3 * kMaxLevels + lq]) { s -= lv[3 * kMaxLevels + lq]; ++lq; }
                const int w = lv[2 * kMaxLevels + lq];
                qlist[i] = (int)starts[lq] + (lv[0 * kMaxLevels + lq] + s / w) * (int)shapes[2 * lq + 1] + lv[1 * kMaxLevels + lq] + s % w;
            }
            __syncthreads();
            // this thread's samples = (slot i, point p), sample index tid + sp * NT
            int qs[SPT];
            // (rows inside MY image, 32-bit: the image's view of the sampling tensors `ion` -- the launcher checks that an image's sampling data
            //  stay below 2^32 bytes; the fused-backward experiment keeps the batch-wide 64-bit rows for its stores)
            auto srow_of = [&](int q) {
                if constexpr (FUSE == 0) return (unsigned)q * (unsigned)M + (unsigned)m;
                else return ((int64_t)n * Lq + q) * M + m;
            };      // recomputed: two registers fewer per sample
            const IO ion = FUSE == 0 ? io.image_view(n, Lq, M, LP) : io;
            float sm_max[SPT], sm_inv[SPT];
            float dotsum[SPT];               // FUSE + fused prologue: sum over the levels of a_k * d/d a_k (softmax backward)
#pragma unroll
            for (int sp = 0; sp < SPT; ++sp) dotsum[sp] = 0.f;
#pragma unroll
            for (int sp = 0; sp < SPT; ++sp) {
                const int sidx = tid + sp * NT, i = sidx / P, p = sidx - i * P;
                qs[sp] = i < nq ? qlist[i] : -1;
                sm_max[sp] = 0.f;
                sm_inv[sp] = 1.f;
                if (IO::kSoftmax) {      // fused prologue: softmax statistics of the (query, head) row.  The thread's L logits
                    // (its point on every level) are loaded ONCE, all loads in flight together; the four points of the row sit
                    // in one quad (P == 4): two DPP steps per reduction.  __expf as in the forward / gather (row_softmax), so
                    // the three kernels of an op agree on the weights to the last bit.
                    constexpr int kLv = 8;       // levels held in registers (pyramids beyond that take the loops)
                    const auto srow = srow_of(qs[sp] >= 0 ? qs[sp] : qlist[0]);
                    float mx, sum = 0.f;
                    if (L <= kLv) {
                        float lg[kLv];
#pragma unroll
                        for (int l = 0; l < kLv; ++l) lg[l] = l < L ? ion.load_w(srow, LP, l * P + p) : -__builtin_huge_valf();
                        mx = lg[0];
#pragma unroll
                        for (int l = 1; l < kLv; ++l) mx = fmaxf(mx, lg[l]);
                        mx = fmaxf(mx, dpp_mov<0xB1>(mx));
                        mx = fmaxf(mx, dpp_mov<0x4E>(mx));
#pragma unroll
                        for (int l = 0; l < kLv; ++l) sum += l < L ? __expf(lg[l] - mx) : 0.f;
                    } else {
                        mx = -__builtin_huge_valf();
                        for (int l = 0; l < L; ++l) mx = fmaxf(mx, ion.load_w(srow, LP, l * P + p));
                        mx = fmaxf(mx, dpp_mov<0xB1>(mx));
                        mx = fmaxf(mx, dpp_mov<0x4E>(mx));
                        for (int l = 0; l < L; ++l) sum += __expf(ion.load_w(srow, LP, l * P + p) - mx);
                    }
                    sum += dpp_mov<0xB1>(sum);
                    sum += dpp_mov<0x4E>(sum);
                    sm_max[sp] = mx;
                    sm_inv[sp] = fast_rcp(sum);
                }
            }
            {   // stage grad_out of the queries, channels (c, c+16) interleaved; every load of a thread is issued before its
                // stores (a load -> store loop exposed the global latency once per 16 rows: 18 % of the kernel)
                constexpr int kPass = (kRegQ * 8 + NT - 1) / NT;               // float4 pieces per thread
                float4 v[kPass];
#pragma unroll
                for (int ps = 0; ps < kPass; ++ps) {
                    const int r = (tid >> 3) + ps * (NT / 8);
                    v[ps] = r < nq ? *reinterpret_cast<const float4 *>(gout + (((int64_t)n * Lq + qlist[r]) * M + m) * kD + 4 * (tid & 7))
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
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
            lap(0);                  // 0: region set-up (query list, softmax statistics, sampling data, grad_out staging)
            for (int l = 0; l < L; ++l) {
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)starts[l];
                const MaskExt me = io.mask_ext(n, l);      // (workgroup-uniform)
                // window: where the region centre maps to on this level, minus half the window
                const int y0 = (int)floorf(pcy * H - 0.5f) - WH / 2 + 1;
                const int x0 = (int)floorf(pcx * W - 0.5f) - WW / 2 + 1;
                int *stats = stats2[l & 1];
                if (tid < 4) stats[tid] = 0;
                for (int k = tid; k < kWR; k += NT) cnt[k] = 0;
                // ---- this thread's sample geometry, kept compact (the registers live across every phase of the level):
                //      bilinear fractions + attention, window index of the top-left corner, its pixel, and per corner one
                //      "exists" bit (0..3) and one "inside the window" bit (4..7); corner weights and window rows are
                //      recomputed where they are used
                int s_wi[SPT], s_pix[SPT], s_fl[SPT], rank[SPT][4];
                float gx[SPT], gy[SPT], ga[SPT];
                float s_lw[SPT], s_lh[SPT], s_a[SPT];      // FUSE: bilinear fractions and attention of the thread's samples
                const int base_pix = st + y0 * W + x0;     // window row r -> pixel base_pix + (r / WW) * W + r % WW
#pragma unroll
                for (int sp = 0; sp < SPT; ++sp) {
                    const int k = l * P + (tid + sp * NT) % P, qq = qs[sp] >= 0 ? qs[sp] : qlist[0];
                    const auto srow = srow_of(qq);
                    if constexpr (FUSE == 0) ion.load_xy(srow, (unsigned)qq, LP, k, l, P, H, W, gx[sp], gy[sp]);
                    else ion.load_xy(srow, (int64_t)n * Lq + qq, LP, k, l, P, H, W, gx[sp], gy[sp]);
                    ga[sp] = ion.load_w(srow, LP, k);
                }
#pragma unroll
                for (int sp = 0; sp < SPT; ++sp) {
                    int off[4] = {-1, -1, -1, -1};
                    float lw = 0.f, lh = 0.f, a = 0.f;
                    int h0 = 0, w0 = 0;
                    if (qs[sp] >= 0) {
                        const float x = gx[sp], y = gy[sp];
                        // (only the SIGNS of off[] are used below -- which corners exist: start 1 and stride 1 keep them and drop the multiplies)
                        if (sample_setup(x, y, H, W, 1, 1, off, lw, lh)) {
                            a = ga[sp];
                            if (IO::kSoftmax) a = __expf(a - sm_max[sp]) * sm_inv[sp];
                            h0 = (int)floorf(sub_rn(mul_rn(y, (float)H), 0.5f));
                            w0 = (int)floorf(sub_rn(mul_rn(x, (float)W), 0.5f));
                            // padded pixels receive no gradient (fused prologue, see RawIO)
                            mask_corners_idx(io, me, n, h0, w0, W, st + h0 * W + w0, off);
                        }
                    }
                    s_lw[sp] = lw;
                    s_lh[sp] = lh;
                    s_a[sp] = a;
                    const int wy = h0 - y0, wx = w0 - x0;
                    // (kPixKey: a level so wide that a window's pixel offsets need more than 15 bits -- W > 1423 -- has no window: every
                    //  corner takes the one-by-one path, correct and slow)
                    const bool lvl_win = !kPixKey || (WH - 1) * W + WW <= 32768;
                    const bool in_y0 = lvl_win && (unsigned)wy < (unsigned)WH, in_y1 = lvl_win && (unsigned)(wy + 1) < (unsigned)WH;
                    const bool in_x0 = (unsigned)wx < (unsigned)WW, in_x1 = (unsigned)(wx + 1) < (unsigned)WW;
                    s_wi[sp] = wy * WW + wx;
                    s_pix[sp] = st + h0 * W + w0;         // a corner that exists is this + (c & 1) + (c >> 1) * W
                    s_fl[sp] = (off[0] >= 0 ? 1 : 0) | (off[1] >= 0 ? 2 : 0) | (off[2] >= 0 ? 4 : 0) | (off[3] >= 0 ? 8 : 0) |
                               (in_y0 && in_x0 ? 16 : 0) | (in_y0 && in_x1 ? 32 : 0) | (in_y1 && in_x0 ? 64 : 0) |
                               (in_y1 && in_x1 ? 128 : 0);
                }
                auto corner_w = [&](int sp, float (&cwv)[4]) {
                    const float lw = s_lw[sp], lh = s_lh[sp], a = s_a[sp], hh = 1.f - lh, hwt = 1.f - lw;
                    cwv[0] = hh * hwt * a;
                    cwv[1] = hh * lw * a;
                    cwv[2] = lh * hwt * a;
                    cwv[3] = lh * lw * a;
                };
                lap(6);              // 6: sample geometry (global loads of sampling_loc / attn_weight)
                __syncthreads();                  // counters zeroed, previous level's walk finished
                lap(1);              // 1: waiting for the other waves' walk of the previous level
                // ---- bucket the in-window corners by window row (count), list the others as misses
#pragma unroll
                for (int sp = 0; sp < SPT; ++sp) {
                    const int i = (tid + sp * NT) / P;
                    float cwv[4];
                    corner_w(sp, cwv);
#pragma unroll
                    for (int cidx = 0; cidx < 4; ++cidx) {
                        const int wr = s_wi[sp] + (cidx & 1) + (cidx >> 1) * WW;
                        rank[sp][cidx] = -1;
                        // paired mode: both corners of the edge lie inside the window and at least one of them exists -> ONE count, on
                        // the left row (a corner outside the level or masked enters with weight zero; its row is never flushed)
                        const bool paired = kPair && (s_fl[sp] & (0x30 << (cidx & 2))) == (0x30 << (cidx & 2)) && (s_fl[sp] & (3 << (cidx & 2)));
                        if (paired) {
                            if (!(cidx & 1)) rank[sp][cidx] = atomicAdd(&cnt[wr], 1);
                            continue;
                        }
                        if (!(s_fl[sp] & (1 << cidx))) continue;          // corner outside the level (or masked)
                        if (!kPair && (s_fl[sp] & (16 << cidx))) {
                            rank[sp][cidx] = atomicAdd(&cnt[wr], 1);
                        } else {                          // pixel index (< 2^23, checked by the launcher) + slot
                            const int at = kNE - 1 - atomicAdd(&stats[1], 1);
                            entries[at] = make_float2(
                                cwv[cidx], __int_as_float((int)(((unsigned)i << 23) | (unsigned)(s_pix[sp] + (cidx & 1) + (cidx >> 1) * W))));
                            rank[sp][cidx] = at;          // FUSE: where the combine step finds this corner's dot product
                        }
                    }
                }
                __syncthreads();
                lap(2);              // 2: count
                // ---- exclusive scan of the kWR counters -> start[]  (thread t owns counters t*KC .. t*KC + KC - 1)
                {
                    int cv[KC], v = 0;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        cv[k] = tid * KC + k < kWR ? cnt[tid * KC + k] : 0;
                        v += cv[k];
                    }
                    int incl = v;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const int t = __builtin_amdgcn_ds_bpermute((lane - d) << 2, incl);      // (= __shfl_up without its own copy of the lane number)
                        if (lane >= d) incl += t;
                    }
                    if (lane == 63) wsum[wv] = incl;
                    __syncthreads();
                    int base = 0;
                    for (int w2 = 0; w2 < wv; ++w2) base += wsum[w2];
                    int run = base + incl - v;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const int j = tid * KC + k;
                        if (j < kWR) start[j] = run;
                        run += cv[k];
                    }
                    if (tid == NT - 1) stats[3] = run;            // total number of bucketed entries
                }
                __syncthreads();
                lap(3);              // 3: scan
                // ---- fill the buckets; the sign bit marks the last entry of its row
#pragma unroll
                for (int sp = 0; sp < SPT; ++sp) {
                    const int i = (tid + sp * NT) / P;
                    float cwv[4];
                    corner_w(sp, cwv);
#pragma unroll
                    for (int cidx = 0; cidx < 4; ++cidx) {
                        const int wr = s_wi[sp] + (cidx & 1) + (cidx >> 1) * WW;
                        if constexpr (kPair) {
                            // {w_left, w_right, last << 31 | next row not empty << 30 | left row << 16 | slot << 7}: the walk hands the
                            // right row's sum on to the run that follows when that run IS the right row
                            if (!(cidx & 1) && (s_fl[sp] & (0x30 << cidx)) == (0x30 << cidx) && (s_fl[sp] & (3 << cidx))) {
                                const int at = start[wr] + rank[sp][cidx];
                                const bool last = rank[sp][cidx] == cnt[wr] - 1;
                                const bool next_there = wr + 1 < kWR && cnt[wr + 1] > 0;
                                reinterpret_cast<float4 *>(entries)[at] = make_float4(
                                    (s_fl[sp] & (1 << cidx)) ? cwv[cidx] : 0.f, (s_fl[sp] & (2 << cidx)) ? cwv[cidx + 1] : 0.f,
                                    __int_as_float((last ? (int)0x80000000 : 0) | (last && next_there ? 0x40000000 : 0) | (wr << 16) | (i << 7)), 0.f);
                            }
                            continue;
                        }
                        if ((s_fl[sp] & (17 << cidx)) == (17 << cidx)) {      // exists and inside the window
                            const int at = start[wr] + rank[sp][cidx];
                            // kPixKey: the destination as pixel offset from the window's first pixel (see above)
                            const int key = kPixKey ? s_pix[sp] - base_pix + (cidx & 1) + (cidx >> 1) * W : wr;
                            entries[at] = make_float2(
                                cwv[cidx], __int_as_float((rank[sp][cidx] == cnt[wr] - 1 ? (int)0x80000000 : 0) | (key << 16) | (i << 7)));
                            rank[sp][cidx] = at;
                        }
                    }
                }
                __syncthreads();
                lap(4);              // 4: fill
                if constexpr (FUSE) {
                    // ---- dot phase: d_e = <grad_out row of the entry's query, value row of its corner> for EVERY entry, one
                    //      lane per entry.  The sorted list makes neighbouring lanes read the same value rows (L1 hits); the
                    //      grad_out row comes from the LDS tile.  Eight 16-byte pieces per operand, the piece order rotated by
                    //      the lane so that the ds_read_b128 of a lane group spread over the banks.  gtile holds channel
                    //      j + 16 h at position 2 j + h, so the 32-byte piece u pairs with value channels 4u.. and 16 + 4u..
                    const __amdgpu_buffer_rsrc_t vr = image_rsrc(value + (int64_t)n * S * M * kD, (unsigned)S * M * kD * 4u);
                    const unsigned head_b = (unsigned)m * kD * 4u, row_b = (unsigned)rs * 4u;
                    const char *gtb0 = reinterpret_cast<const char *>(gtile);
                    const int total = stats[3], nitems = total + stats[1];
                    const int nquads = FUSE == 2 ? (total + 3) >> 2 : 0;
                    // FUSE == 2: a lane takes FOUR consecutive entries of the sorted list and reloads the value row only when
                    // the row changes (runs average ~8 entries), misses keep one lane per entry
                    for (int it = tid; it < ((AID & 1) ? 0 : nquads); it += NT) {
                        const int e0 = 4 * it, e1 = min(e0 + 4, total);
                        float4 va[4], vb[4];
                        int prev = -1;
#pragma unroll 1
                        for (int e = e0; e < e1; ++e) {
                            const int pk = __float_as_int(entries[e].y);
                            const int wr = (pk >> 16) & 0x3fff;
                            if (wr != prev) {
                                const unsigned rowb = (unsigned)(base_pix + (wr / WW) * W + wr % WW) * row_b + head_b;
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const unsigned u = (unsigned)(k + lane) & 3u;
                                    va[k] = buf_ld4(vr, rowb + 16u * u);
                                    vb[k] = buf_ld4(vr, rowb + 64u + 16u * u);
                                }
                                prev = wr;
                            }
                            const unsigned slotb = (unsigned)(pk & 0xff80);
                            float acc = 0.f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const unsigned u = (unsigned)(k + lane) & 3u;
                                const float4 ga = *reinterpret_cast<const float4 *>(gtb0 + slotb + 32u * u);
                                const float4 gb = *reinterpret_cast<const float4 *>(gtb0 + slotb + 32u * u + 16u);
                                acc += ga.x * va[k].x + ga.y * vb[k].x + ga.z * va[k].y + ga.w * vb[k].y;
                                acc += gb.x * va[k].z + gb.y * vb[k].z + gb.z * va[k].w + gb.w * vb[k].w;
                            }
                            dvals[e] = acc;
                        }
                    }
                    for (int it = tid + (FUSE == 2 ? total : 0); it < ((AID & 1) ? 0 : nitems); it += NT) {
                        const bool front = it < total;
                        const int e = front ? it : kNE - 1 - (it - total);
                        const int pk = __float_as_int(entries[e].y);
                        const int wr = (pk >> 16) & 0x3fff;
                        const int pix = front ? base_pix + (wr / WW) * W + wr % WW : (pk & 0x7fffff);
                        const unsigned slotb = front ? (unsigned)(pk & 0xff80) : (((unsigned)pk >> 23) << 7);
                        const unsigned rowb = (unsigned)pix * row_b + head_b;
                        float4 va[4], vb[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned u = (unsigned)(k + lane) & 3u;
                            if (AID & 8) {
                                va[k] = make_float4(1.f, 2.f, 3.f, __uint_as_float(rowb + u));
                                vb[k] = va[k];
                                continue;
                            }
                            va[k] = buf_ld4(vr, rowb + 16u * u);
                            vb[k] = buf_ld4(vr, rowb + 64u + 16u * u);
                        }
                        float acc = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned u = (unsigned)(k + lane) & 3u;
                            const float4 ga = (AID & 16) ? make_float4(1.f, 2.f, 3.f, __uint_as_float(slotb + u))
                                                         : *reinterpret_cast<const float4 *>(gtb0 + slotb + 32u * u);
                            const float4 gb = (AID & 16) ? ga : *reinterpret_cast<const float4 *>(gtb0 + slotb + 32u * u + 16u);
                            acc += ga.x * va[k].x + ga.y * vb[k].x + ga.z * va[k].y + ga.w * vb[k].y;
                            acc += gb.x * va[k].z + gb.y * vb[k].z + gb.z * va[k].w + gb.w * vb[k].w;
                        }
                        dvals[e] = acc;
                    }
                    __syncthreads();
                    lap(7);          // 7: dot phase
                }
                // ---- owner computes: 64 streams of 16 lanes (lane l = channels l and l+16) each walk an equal share of
                //      the row-sorted entries, running row sum in two registers, one atomic pair per finished row
                {
                    constexpr int kStreams = NT / 16;
                    const int sid = tid >> 4, l16 = tid & 15;
                    const float2 *gt2 = reinterpret_cast<const float2 *>(gtile);
                    // entry word: bit 31 = last of its row, bits 16..29 = window row, bits 7..15 = query slot * 128 (= the byte
                    // offset of the slot's grad_out row in gtile: one and-or gives the lane's address)
                    const char *gtb = reinterpret_cast<const char *>(gtile) + l16 * 8;
                    auto gq_of = [&](float y) {
                        if ((AID & 256) != 0) return make_float2(y, 1.f);      // aid 256: no grad_out reads
                        return *reinterpret_cast<const float2 *>(gtb + (__float_as_int(y) & 0xff80));
                    };
                    float *gvs = gvalue + ((int64_t)n * S * M + m) * kD + l16;
                    // kPixKey: scalar base of the level's window (first pixel, this head; may lie before the map when the window hangs
                    // over its edge -- only pixels inside the level are ever flushed) + a 32-bit byte offset per lane
                    // (the flush takes the word's upper half as it is, last-of-row flag included: 0x8000 pixels are taken off the base)
                    const uint64_t gwin_v = reinterpret_cast<uint64_t>(gvalue) + (uint64_t)((((int64_t)n * S + base_pix - 0x8000) * M + m) * kD * 4);
                    const uint64_t gwin = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(gwin_v >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gwin_v);
                    const unsigned rs4 = (unsigned)rs * 4u, lane4 = (unsigned)l16 * 4u;
                    const uint64_t gimg_v = reinterpret_cast<uint64_t>(gvalue) + (uint64_t)(((int64_t)n * S * M + m) * kD * 4);      // image, head
                    const uint64_t gimg = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(gimg_v >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gimg_v);
                    (void)gimg;
                    const int total = (AID & 64) ? 0 : stats[3];      // aid 64: no walk at all
                    // WU >= 100: shares start on even entries (16-byte aligned: two entries per ds_read_b128, which moves twice
                    // the bytes per LDS cycle of the ds_read2_b64 the compiler picks for single entries) and the entries of batch
                    // i + 1 are requested before batch i is processed
                    constexpr bool kPipe = WU >= 100 && WU < 1000;
                    constexpr int WB = WU % 100;
                    const int lo = kPipe ? ((int)((int64_t)total * sid / kStreams) & ~1) : (int)((int64_t)total * sid / kStreams);
                    const int hi = kPipe ? (sid + 1 == kStreams ? total : ((int)((int64_t)total * (sid + 1) / kStreams) & ~1))
                                         : (int)((int64_t)total * (sid + 1) / kStreams);
                    int cur = -1;           // row of the most recent entry whose sum is still open, or -1
                    float2 accv = make_float2(0.f, 0.f);
                    // window row -> pixel by arithmetic (a table lookup here costs an LDS round trip inside the divergent flush
                    // branch, with every stream of the wavefront waiting on it)
                    auto flush = [&](int rowi) {
                        if (DBG && l16 == 0) SEMIDETR_DBG_ADD(12 + (l < 4 ? l : 3), 1);      // flushed rows by sampling level
                        if ((AID & 2) && accv.x != 1.2345e30f) return;
                        if constexpr (kPixKey) {
                            // byte offset of the lane's first channel from the window's first pixel: one 24-bit multiply-add; the two
                            // half-line atomics take the scalar base (the instructions fp_atomic_add compiles to, saddr form).
                            // 16-bit key x row pitch must stay below 2^32: M * 128 <= 4096 on every fast-path launch (heads_ok, msda.hip;
                            // launch_fast_backward repeats the bound next to this kernel's launch -- ADVICE r05).  gfx9 encodings.
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "msda_region.h: the flush path is written in gfx9 (CDNA) assembly"
#endif
                            unsigned voff;
                            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(voff) : "v"(rowi | 0x8000), "s"(rs4), "v"(lane4));
                            asm volatile("global_atomic_add_f32 %0, %1, %3\n\tglobal_atomic_add_f32 %0, %2, %3 offset:64"
                                         :: "v"(voff), "v"(accv.x), "v"(accv.y), "s"(gwin) : "memory");
                        } else {
                            float *pr = gvs + (int64_t)(base_pix + (rowi / WW) * W + rowi % WW) * rs;
                            fp_atomic_add(pr, accv.x);
                            fp_atomic_add(pr + 16, accv.y);
                        }
                    };
                    auto step = [&](const float2 &en, const float2 &gq) {
                        const int pk = (AID & 128) ? (__float_as_int(en.y) & 0x7fffffff) : __float_as_int(en.y);      // aid 128: no row ever ends
                        accv.x += en.x * gq.x;
                        accv.y += en.x * gq.y;
                        cur = kPixKey ? (int)((unsigned)pk >> 16) : (pk >> 16) & 0x3fff;      // (kPixKey: bit 15 = the last-of-row flag, see flush)
                        if (pk < 0) {
                            flush(cur);
                            accv = make_float2(0.f, 0.f);
                            cur = -1;
                        }
                    };
                    int e = lo;
                    if constexpr (kPair) {
                        // paired entries: sums of rows cur (accv) and cur + 1 (accr)
                        const float4 *e4 = reinterpret_cast<const float4 *>(entries);
                        float2 accr = make_float2(0.f, 0.f);
                        bool open_r = false;             // accr holds contributions (row cur + 1 is then a valid pixel)
                        auto flush2 = [&](int rowi, const float2 &a) {
                            if (DBG && l16 == 0) SEMIDETR_DBG_ADD(12 + (l < 4 ? l : 3), 1);      // flushed rows by sampling level
                            // a window row outside the level only ever collected zero weights (or NaN from a non-finite grad_out): skip it
                            if ((unsigned)(y0 + rowi / WW) >= (unsigned)H || (unsigned)(x0 + rowi % WW) >= (unsigned)W) return;
                            float *pr = gvs + (int64_t)(base_pix + (rowi / WW) * W + rowi % WW) * rs;
                            if ((AID & 2) && a.x != 1.2345e30f) return;
                            fp_atomic_add(pr, a.x);
                            fp_atomic_add(pr + 16, a.y);
                        };
                        auto pstep = [&](const float4 &en, const float2 &gq) {
                            const int pk = __float_as_int(en.z);
                            accv.x += en.x * gq.x;
                            accv.y += en.x * gq.y;
                            accr.x += en.y * gq.x;
                            accr.y += en.y * gq.y;
                            open_r = true;
                            cur = (pk >> 16) & 0x3fff;
                            if (pk < 0) {                // last entry of row cur's run
                                flush2(cur, accv);
                                if (pk & 0x40000000) {   // the next run is row cur + 1: its sum so far moves over
                                    accv = accr;
                                    cur = cur + 1;
                                } else {
                                    if (!(AID & 4)) flush2(cur + 1, accr);
                                    accv = make_float2(0.f, 0.f);
                                    cur = -1;
                                }
                                accr = make_float2(0.f, 0.f);
                                open_r = false;
                            }
                        };
                        constexpr int WB = WU % 100;
                        for (; e + WB <= hi; e += WB) {
                            float4 en[WB];
                            float2 gq[WB];
#pragma unroll
                            for (int u = 0; u < WB; ++u) en[u] = e4[e + u];
#pragma unroll
                            for (int u = 0; u < WB; ++u) gq[u] = gq_of(en[u].z);
#pragma unroll
                            for (int u = 0; u < WB; ++u) pstep(en[u], gq[u]);
                        }
                        if (e < hi) {       // tail of < WB entries
                            float4 en[WB];
                            float2 gq[WB];
#pragma unroll
                            for (int u = 0; u < WB; ++u) en[u] = e4[min(e + u, hi - 1)];
#pragma unroll
                            for (int u = 0; u < WB; ++u) gq[u] = gq_of(en[u].z);
#pragma unroll
                            for (int u = 0; u < WB; ++u)
                                if (e + u < hi) pstep(en[u], gq[u]);
                        }
                        if (cur >= 0) {
                            flush2(cur, accv);
                            if (open_r && !(AID & 4)) flush2(cur + 1, accr);
                            cur = -1;
                        }
                    } else if constexpr (kPipe) {
                        const float4 *e4 = reinterpret_cast<const float4 *>(entries);
                        float4 ea[WB / 2];
#pragma unroll
                        for (int u = 0; u < WB / 2; ++u) ea[u] = e4[(e >> 1) + u];        // (the list has 8 entries of slack)
                        for (; e + WB <= hi; e += WB) {
                            float2 gq[WB];
#pragma unroll
                            for (int u = 0; u < WB / 2; ++u) {
                                gq[2 * u] = gq_of(ea[u].y);
                                gq[2 * u + 1] = gq_of(ea[u].w);
                            }
                            float4 nx[WB / 2];
#pragma unroll
                            for (int u = 0; u < WB / 2; ++u) nx[u] = e4[min((e + WB) >> 1, (kNE + 8 - WB) >> 1) + u];
#pragma unroll
                            for (int u = 0; u < WB / 2; ++u) {
                                step(make_float2(ea[u].x, ea[u].y), gq[2 * u]);
                                step(make_float2(ea[u].z, ea[u].w), gq[2 * u + 1]);
                            }
#pragma unroll
                            for (int u = 0; u < WB / 2; ++u) ea[u] = nx[u];
                        }
                        if (e < hi) {       // tail of < WB entries: ea holds them (and whatever follows them in the list)
                            float2 en[WB], gq[WB];
#pragma unroll
                            for (int u = 0; u < WB / 2; ++u) {
                                en[2 * u] = make_float2(ea[u].x, ea[u].y);
                                en[2 * u + 1] = make_float2(ea[u].z, ea[u].w);
                            }
#pragma unroll
                            for (int u = 0; u < WB; ++u) gq[u] = gq_of(e + u < hi ? en[u].y : en[0].y);
#pragma unroll
                            for (int u = 0; u < WB; ++u)
                                if (e + u < hi) step(en[u], gq[u]);
                        }
                    } else {
                    for (; e + WU <= hi; e += WU) {
                        float2 en[WU], gq[WU];
#pragma unroll
                        for (int u = 0; u < WU; ++u) en[u] = entries[e + u];
#pragma unroll
                        for (int u = 0; u < WU; ++u) gq[u] = gq_of(en[u].y);
#pragma unroll
                        for (int u = 0; u < WU; ++u) step(en[u], gq[u]);
                    }
                    if (e < hi) {       // tail of < WU entries
                        float2 en[WU], gq[WU];
#pragma unroll
                        for (int u = 0; u < WU; ++u) en[u] = entries[min(e + u, hi - 1)];
#pragma unroll
                        for (int u = 0; u < WU; ++u) gq[u] = gq_of(en[u].y);
#pragma unroll
                        for (int u = 0; u < WU; ++u)
                            if (e + u < hi) step(en[u], gq[u]);
                    }
                    }
                    if (cur >= 0) flush(cur);
                    // ---- misses: one row update per (sample, corner), as the plain kernel does
                    const int nmiss = stats[1];
                    for (int mi = sid; mi < ((AID & 8) ? 0 : nmiss); mi += kStreams) {
                        const float2 en = entries[kNE - 1 - mi];
                        const int pk = __float_as_int(en.y);
                        const float2 g2 = gt2[((unsigned)pk >> 23) * 16 + l16];
                        if constexpr (kPixKey) {      // (in-image byte offsets fit 32 bits: checked by the launcher)
                            const unsigned voff = (unsigned)(pk & 0x7fffff) * rs4 + lane4;
                            asm volatile("global_atomic_add_f32 %0, %1, %3\n\tglobal_atomic_add_f32 %0, %2, %3 offset:64"
                                         :: "v"(voff), "v"(en.x * g2.x), "v"(en.x * g2.y), "s"(gimg) : "memory");
                        } else {
                            float *pr = gvs + (int64_t)(pk & 0x7fffff) * rs;
                            fp_atomic_add(pr, en.x * g2.x);
                            fp_atomic_add(pr + 16, en.x * g2.y);
                        }
                    }
                    if (DBG && tid == 0) SEMIDETR_DBG_ADD(10, nmiss);
                }
                lap(5);              // 5: walk + misses of wave 0 (the other waves' walk ends show up in the next lap 1)
                if constexpr (FUSE) {
                    // ---- combine: the two small gradients of the thread's samples from the four corner dot products
                    //      (ms_deform_im2col_cuda.cuh:123-158 in dot-product form, as msda_bwd_gather_d32 computes them)
#pragma unroll
                    for (int sp = 0; sp < SPT; ++sp) {
                        if (qs[sp] < 0) continue;
                        float d[4];
#pragma unroll
                        for (int cidx = 0; cidx < 4; ++cidx) d[cidx] = rank[sp][cidx] >= 0 ? dvals[rank[sp][cidx]] : 0.f;
                        const float lw = s_lw[sp], lh = s_lh[sp], a = s_a[sp];
                        const float hh = 1.f - lh, hwt = 1.f - lw;
                        const float pa = hh * hwt * d[0] + hh * lw * d[1] + lh * hwt * d[2] + lh * lw * d[3];
                        const float px = a * (hh * (d[1] - d[0]) + lh * (d[3] - d[2]));
                        const float py = a * (hwt * (d[2] - d[0]) + lw * (d[3] - d[1]));
                        const int k = l * P + (tid + sp * NT) % P;
                        const int64_t nq = (int64_t)n * Lq + qs[sp], srow = nq * M + m;
                        if ((AID & 4) && pa != 1.2345e30f) continue;
                        if constexpr ((AID & 32) != 0 && !IO::kSoftmax) {      // aid: streaming stores
                            st_stream2(io.gloc + (srow * LP + k) * 2, make_float2((float)W * px, (float)H * py));
                            st_stream1(io.gattn + srow * LP + k, pa);
                            continue;
                        }
                        io.store_xy(srow, nq, LP, k, l, P, H, W, (float)W * px, (float)H * py);
                        io.store_attn_partial(srow, LP, k, pa);      // final for the reference contract; d/d a_k for the fused prologue
                        dotsum[sp] += a * pa;
                    }
                    lap(8);          // 8: combine
                }
            }
            if constexpr (FUSE && IO::kSoftmax) {
                // fused prologue: softmax backward over the row, g_logit_k = a_k * (g_a_k - sum_j a_j g_a_j).  The L * P values
                // of a row are spread over the levels of the loop above, so they were parked in the output and are finished here
#pragma unroll
                for (int sp = 0; sp < SPT; ++sp) {
                    float dot = dotsum[sp];
                    dot += dpp_mov<0xB1>(dot);       // the four points of a (query, head) row sit in one quad
                    dot += dpp_mov<0x4E>(dot);
                    if (qs[sp] < 0) continue;
                    const int64_t srow = srow_of(qs[sp]);
                    const int p = (tid + sp * NT) % P;
                    for (int l = 0; l < L; ++l) {
                        const float a = __expf(io.load_w(srow, LP, l * P + p) - sm_max[sp]) * sm_inv[sp];
                        io.finish_attn(srow, LP, l * P + p, a, dot);
                    }
                }
            }
        }
    }
}

template <typename IO, int NT, int kRegQ, int RTH, int RTW, int WH, int WW, int DBG = 0, int WPE = 4, int WU = 8>
__global__ __launch_bounds__(NT, WPE) void msda_bwd_scatter_d32_reg(
    const float *__restrict__ gout, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const IO io, int S, int M, int L, int regions_bound, float *__restrict__ gvalue)
{
    io.same_dims(S, M, L);
    extern __shared__ float4 smem[];
#ifndef SEMIDETR_SCATTER_AID
#define SEMIDETR_SCATTER_AID 0      // tuning builds (tools/ab_build.sh -DSEMIDETR_SCATTER_AID=...): timing aids of the walk, results wrong
#endif
    reg_scatter_body<IO, NT, kRegQ, RTH, RTW, WH, WW, DBG, WU, 0, SEMIDETR_SCATTER_AID>((int)blockIdx.x, smem, gout, shapes, starts, io, S, M, L,
                                                          regions_bound, gvalue);
}

#if SEMIDETR_EXPERIMENTS
// timing aids (backward variants 6986 / 6987, the first one's results wrong): the PAIRED-corner scatter without / with its row atomics
template <typename IO, int AID>
__global__ __launch_bounds__(512, 6) void msda_bwd_scatter_d32_reg_pair_aid(
    const float *__restrict__ gout, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const IO io, int S, int M, int L, int regions_bound, float *__restrict__ gvalue)
{
    io.same_dims(S, M, L);
    extern __shared__ float4 smem[];
    reg_scatter_body<IO, 512, 176, 8, 16, 24, 32, 0, 1004, 0, AID>((int)blockIdx.x, smem, gout, shapes, starts, io, S, M, L, regions_bound, gvalue);
}
// timing aid (backward variant 6985, results wrong): the region scatter WITHOUT its row atomics -- what the compute side alone costs
template <typename IO>
__global__ __launch_bounds__(512, 4) void msda_bwd_scatter_d32_reg_noatomics(
    const float *__restrict__ gout, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const IO io, int S, int M, int L, int regions_bound, float *__restrict__ gvalue)
{
    io.same_dims(S, M, L);
    extern __shared__ float4 smem[];
    reg_scatter_body<IO, 512, 208, 8, 16, 24, 32, 0, 8, 0, 2>((int)blockIdx.x, smem, gout, shapes, starts, io, S, M, L, regions_bound, gvalue);
}

// EXPERIMENT (round 4, backward variants 6900-6911; measured and rejected): the WHOLE encoder backward in one
// kernel -- the region scatter above plus the two small gradients (reg_scatter_body<..., FUSE>).  Once the (sample, corner)
// pairs of a level are bucketed, their dot products <grad_out[query], value[corner]> take one pass with a lane per entry
// (the grad_out rows are in LDS already, the sorted order turns the value-row loads into L1 hits), and the owner of a sample
// combines its four corners (dvals) into d/d attention and d/d location: no msda_bwd_gather_d32 launch.  Parity-green, but at
// bs 4 766 us against 737 us for gather + scatter: the dot pass costs 256 us (its 8 x 16-byte loads per entry are bound by
// the L1 return path exactly like the gather's: 138 us; LDS reads with a two-way bank conflict 56 us), the piecewise result
// stores 79 us; a lane per FOUR entries with reloads on row changes (FUSE == 2) 1070 us.  grad_value must be zero on entry.
template <typename IO, int NT, int kRegQ, int RTH, int RTW, int WH, int WW, int DBG = 0, int WPE = 4, int WU = 8, int AID = 0, int FZ = 1>
__global__ __launch_bounds__(NT, WPE) void msda_bwd_enc_fused_d32(
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int regions_bound, float *__restrict__ gvalue)
{
    io.same_dims(S, M, L);
    extern __shared__ float4 smem[];
    reg_scatter_body<IO, NT, kRegQ, RTH, RTW, WH, WW, DBG, WU, FZ, AID>((int)blockIdx.x, smem, gout, shapes, starts, io, S, M, L,
                                                                   regions_bound, gvalue, value);
}
#endif

#if SEMIDETR_EXPERIMENTS
// EXPERIMENT (backward variant 697, slower).  ONE launch for both halves of the encoder backward (second attempt, after msda_bwd_enc_merged): region-scatter
// workgroups (LDS / instruction issue / atomics) and pairs of gather blocks (vector-memory path) share the CUs.  Roles
// are dealt out in GROUPS OF EIGHT consecutive workgroups -- consecutive workgroups go round robin to the 8 XCDs, so
// "every period-th workgroup scatters" with an even period had put all scatter workgroups on two XCDs.
template <typename IO, int KLP>
__global__ __launch_bounds__(512, 4) void msda_bwd_encreg_merged(
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int P, int regions_bound, int scatter_blocks,
    int gather_bound, int gather_blocks, int scatter_groups, int period, float *__restrict__ gvalue)
{
    io.same_dims(S, M, L);
    extern __shared__ float4 smem[];
    const int b = (int)blockIdx.x, g = b >> 3, lane8 = b & 7;
    const bool is_scatter = (g % period == 0) && (g / period < scatter_groups);
    if (is_scatter) {
        const int sb = (g / period) * 8 + lane8;
        if (sb < scatter_blocks)
            reg_scatter_body<IO, 512, 208, 8, 16, 24, 32>(sb, smem, gout, shapes, starts, io, S, M, L, regions_bound, gvalue);
        return;
    }
    const int ns_before = min((g + period - 1) / period, scatter_groups);       // scatter groups with index < g
    const int gi = (g - ns_before) * 8 + lane8;                                  // index among the gather workgroups
    const int half = (int)threadIdx.x >> 8;
    const int vb = 2 * gi + half;
    const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
    gather_body<IO, KLP, 408, true>(vb, (int)threadIdx.x & 255, smem + half * half_f4, vb < gather_blocks, gout, value,
                                    shapes, starts, io, S, M, L, S, P, gather_bound);
}
#endif
