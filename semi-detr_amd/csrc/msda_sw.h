// grad_value for encoder self-attention, region-owned scatter with CELL-sorted samples and a LANE-PER-ROW walk (fp32, D == 32,
// num_point == 4).  Round 5 -- MEASURED AND REJECTED (profiles/r05_cell_sorted_scatter.txt): compiled only into tuning builds
// (-DSEMIDETR_SCATTER_SW=1, tools/ab_build.sh), never into the product or the experiments library.  It was meant to succeed
// msda_bwd_scatter_d32_reg (ms_deform_im2col_cuda.cuh:87-159: the atomicAdd of w_corner * attn * grad_out into grad_value); results
// are correct (every encoder-backward parity test passes with it swapped in), it is two to three times slower.
//
// The region scatter (msda_region.h) buckets every (sample, corner) PAIR by its window row -- four counts, four fills and four
// 8-byte entries per sample -- and walks the row-sorted list with streams of 16 lanes (two channels per lane): ~2 wave-instructions
// per pair, 45.5 M pairs per bs-4 launch, 380 us before a single atomic is counted.  Two changes, both from what the lane-per-sample
// gather (msda_gw.h) taught:
//   * ONE entry per SAMPLE, bucketed by the window CELL of its top-left corner: a window row r receives the top-left weights of the
//     samples of cell r, the top-right ones of cell r - 1, the bottom-left ones of cell r - WW and the bottom-right ones of cell
//     r - WW - 1 -- and in a cell-sorted list cells r - 1, r (and r - WW - 1, r - WW) are NEIGHBOURS, so a row reads two contiguous
//     ranges.  A quarter of the counting / scanning / filling; the entry holds the four corner weights (zero for a corner outside the
//     level or padded) and the walk picks the component its cell implies;
//   * the walk keeps the region scatter's shape -- streams of 16 lanes, lane = channels (l, l + 16), the row's sum in two registers,
//     one full-line atomic pair per row -- but a stream owns whole ROWS: the same scan that places the cells also compacts the
//     non-empty rows (counts and row flags packed in one word), and the rows are dealt to the streams round-robin (the four streams of
//     a wave take four neighbouring rows).  No run detection, no "last of its row" flags, every row flushed exactly once.
//   (First version of this file: a LANE per row, the row's 32 channels in registers, grad_out rows read as ds_read_b128 like the
//   lane-per-sample gather's window rows.  Correct, and twice as slow as the region scatter -- 839 against 394 us: rows hold 2 .. 40
//   samples, so a wave waits for its longest row (its LDS reads issue for one active lane as for 64), and the transposed flush through
//   an LDS buffer serialised the row atomics behind the walk: 331 us with the entry loop removed.)
// Samples whose cell lies outside the window (or on its last row / column) scatter their corners one by one, like the region
// scatter's misses: any sampling pattern is correct.  Regions with more than Q queries are processed in passes.
#pragma once
#ifndef SEMIDETR_SW_DBG
#define SEMIDETR_SW_DBG 0      // timing aids (tuning builds, results wrong): 1 no entry loop, 2 no flush, 4 no misses, 8 no walk at all
#endif

template <int NT, int Q, int WH, int WW>
constexpr size_t sw_lds_bytes()
{
    // grad_out tile | pool: entries {4 weights} from the front, misses {weight, slot << 23 | pixel} from the back (a sample is one
    // 16-byte entry or at most four 8-byte misses) | slots | cnt | start (+ total) | query list | level table, wave sums, counters,
    // first row of every stream
    return (size_t)Q * kD * 4 + (size_t)Q * kPT * 32 + (size_t)((Q * kPT * 2 + 15) & ~15) + (size_t)WH * WW * 4 + (size_t)(WH * WW + 4) * 4 +
           (size_t)((WH * WW + 7) & ~7) * 2 + (size_t)Q * 4 + 4 * kMaxLevels * 4 + (NT / 64) * 4 + 2 * 8 * 4;
}
constexpr unsigned kSwNoCorner = 0xffffffffu;      // weight word of a corner that does not exist (outside the level / padded): never multiplied

template <typename IO, int NT, int Q, int RTH, int RTW, int WH, int WW, int WPE = 2>
__global__ __launch_bounds__(NT, WPE) void msda_sw_d32(
    const float *__restrict__ gout, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts, const IO io, int S,
    int M, int L, int regions_bound, float *__restrict__ gvalue)
{
    io.same_dims(S, M, L);
    constexpr int P = kPT, kWR = WH * WW, SPT = (Q * P + NT - 1) / NT, NW = NT / 64;
    static_assert(kWR < 32768 && Q * P < 32768, "cell counts and row flags share one 32-bit scan word");
    extern __shared__ __attribute__((aligned(128))) float4 smem[];
    char *const lds = reinterpret_cast<char *>(smem);
    float *const gtile = reinterpret_cast<float *>(lds);                                   // [Q][32] grad_out rows of the pass's queries
    float4 *const entw = reinterpret_cast<float4 *>(lds + Q * kD * 4);                     // pool front: {w_tl, w_tr, w_bl, w_br}, cell-sorted
    float2 *const miss_end = reinterpret_cast<float2 *>(lds + Q * kD * 4 + Q * P * 32);    // pool back: miss i lives at miss_end[-1 - i]
    unsigned short *const ents = reinterpret_cast<unsigned short *>(miss_end);             // [Q * P] query slot of the entry
    int *const cnt = reinterpret_cast<int *>(lds + Q * kD * 4 + Q * P * 32 + ((Q * P * 2 + 15) & ~15));      // [kWR] samples per cell
    int *const start = cnt + kWR;                                                          // [kWR + 1] exclusive scan (+ total)
    unsigned short *const rowlist = reinterpret_cast<unsigned short *>(start + kWR + 4);   // [kWR] non-empty rows, ascending
    int *const qlist = reinterpret_cast<int *>(rowlist + ((kWR + 7) & ~7));                // [Q]
    int *const lv = qlist + Q;                                                             // [4][kMaxLevels]
    int *const wsum = lv + 4 * kMaxLevels;                                                 // [NW]
    int (*const stats2)[8] = reinterpret_cast<int (*)[8]>(wsum + NW);                      // per level parity: [0] misses, [1] entries, [2] rows

    // (the thread index goes through an empty asm per region and per level: what is derived from it -- LDS addresses, sample slots,
    //  cell ranges -- is then rebuilt there instead of being hoisted into registers held for the whole kernel, msda_region.h)
    auto fresh_tid = [&]() {
        int t = (int)threadIdx.x;
        asm volatile("" : "+v"(t));
        return t;
    };
    const int Lq = S, LP = L * P, rs = M * kD;
    const int b = (int)blockIdx.x;
    const int m = (b % M + (b / M) / kScatterHeadRun) % M;
    const int slot0 = (b / M) % regions_bound, n = (b / M) / regions_bound;
    float *const gvs = gvalue + ((int64_t)n * S * M + m) * kD;

    // the finest level (most pixels) carries the region grid
    int Hb = (int)shapes[0], Wb = (int)shapes[1];
    for (int l = 1; l < L; ++l) {
        const int h = (int)shapes[2 * l], w = (int)shapes[2 * l + 1];
        if (h * w > Hb * Wb) { Hb = h; Wb = w; }
    }
    const int nry = (Hb + RTH - 1) / RTH, nrx = (Wb + RTW - 1) / RTW, nregions = nry * nrx;

    for (int reg = slot0; reg < nregions; reg += regions_bound) {
        const int tid = fresh_tid();
        const int y0b = (reg / nrx) * RTH, x0b = (reg % nrx) * RTW;
        const int y1b = min(y0b + RTH, Hb), x1b = min(x0b + RTW, Wb);
        __syncthreads();                      // previous region fully done before its LDS state is reused
        // ---- the region's queries: on level lq an exact rectangle (msda_region.h)
        if (tid < L) {
            const int Hq = (int)shapes[2 * tid], Wq = (int)shapes[2 * tid + 1];
            const int ylo = rw_first(y0b, Hq, Hb), yhi = y1b >= Hb ? Hq : rw_first(y1b, Hq, Hb);
            const int xlo = rw_first(x0b, Wq, Wb), xhi = x1b >= Wb ? Wq : rw_first(x1b, Wq, Wb);
            lv[0 * kMaxLevels + tid] = ylo;
            lv[1 * kMaxLevels + tid] = xlo;
            lv[2 * kMaxLevels + tid] = max(xhi - xlo, 0);
            lv[3 * kMaxLevels + tid] = max(yhi - ylo, 0) * max(xhi - xlo, 0);
        }
        __syncthreads();
        int nq_sum = 0;
        for (int l = 0; l < L; ++l) nq_sum += lv[3 * kMaxLevels + l];
        const int nq_total = __builtin_amdgcn_readfirstlane(nq_sum);
        const float pcy = (y0b + 0.5f * RTH) / (float)Hb, pcx = (x0b + 0.5f * RTW) / (float)Wb;      // region centre, normalised

        for (int q_base = 0; q_base < nq_total; q_base += Q) {      // one pass unless the region has > Q queries
            const int nq = min(Q, nq_total - q_base);
            __syncthreads();
            for (int c = tid; c < kWR; c += NT) cnt[c] = 0;      // (later levels find them zeroed by the previous level's fill phase)
            if (tid < 16) stats2[0][tid] = 0;
            for (int i = tid; i < nq; i += NT) {       // slot -> query index
                int s = q_base + i, lq = 0;
                while (s >= lv[3 * kMaxLevels + lq]) { s -= lv[3 * kMaxLevels + lq]; ++lq; }
                const int w = lv[2 * kMaxLevels + lq];
                qlist[i] = (int)starts[lq] + (lv[0 * kMaxLevels + lq] + s / w) * (int)shapes[2 * lq + 1] + lv[1 * kMaxLevels + lq] + s % w;
            }
            __syncthreads();
            // ---- my samples = (slot i, point p), sample index tid + sp * NT; fused prologue: softmax statistics of the (query, head) row
            int qs[SPT];
            float sm_max[SPT], sm_inv[SPT];
            auto srow_of = [&](int q) { return ((int64_t)n * Lq + q) * M + m; };
#pragma unroll
            for (int sp = 0; sp < SPT; ++sp) {
                const int sidx = tid + sp * NT, i = sidx / P, p = sidx - i * P;
                qs[sp] = i < nq ? qlist[i] : -1;
                sm_max[sp] = 0.f;
                sm_inv[sp] = 1.f;
                if (IO::kSoftmax) {      // the thread's point on every level, the four points of the row sit in one quad (msda_region.h)
                    const int64_t srow = srow_of(qs[sp] >= 0 ? qs[sp] : qlist[0]);
                    float mx = -__builtin_huge_valf(), sum = 0.f;
                    for (int l = 0; l < L; ++l) mx = fmaxf(mx, io.load_w(srow, LP, l * P + p));
                    mx = fmaxf(mx, dpp_mov<0xB1>(mx));
                    mx = fmaxf(mx, dpp_mov<0x4E>(mx));
                    for (int l = 0; l < L; ++l) sum += __expf(io.load_w(srow, LP, l * P + p) - mx);
                    sum += dpp_mov<0xB1>(sum);
                    sum += dpp_mov<0x4E>(sum);
                    sm_max[sp] = mx;
                    sm_inv[sp] = fast_rcp(sum);
                }
            }
            {   // stage grad_out of the queries: every load of a thread before its stores
                constexpr int kPass = (Q * 8 + NT - 1) / NT;               // float4 pieces per thread
                float4 v[kPass];
#pragma unroll
                for (int ps = 0; ps < kPass; ++ps) {
                    const int r = (tid >> 3) + ps * (NT / 8);
                    v[ps] = r < nq ? *reinterpret_cast<const float4 *>(gout + (((int64_t)n * Lq + qlist[r]) * M + m) * kD + 4 * (tid & 7))
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int ps = 0; ps < kPass; ++ps) {
                    const int r = (tid >> 3) + ps * (NT / 8);
                    if (r < nq) {      // channels (c, c + 16) side by side: a walking lane reads its two channels with one ds_read_b64
                        float *dst = gtile + r * kD;
                        const int c0 = 4 * (tid & 7);
                        dst[((c0 + 0) & 15) * 2 + ((c0 + 0) >> 4)] = v[ps].x;
                        dst[((c0 + 1) & 15) * 2 + ((c0 + 1) >> 4)] = v[ps].y;
                        dst[((c0 + 2) & 15) * 2 + ((c0 + 2) >> 4)] = v[ps].z;
                        dst[((c0 + 3) & 15) * 2 + ((c0 + 3) >> 4)] = v[ps].w;
                    }
                }
            }
            for (int l = 0; l < L; ++l) {
                const int tid = fresh_tid(), lane = tid & 63, wv = tid >> 6;
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)starts[l];
                const MaskExt me = io.mask_ext(n, l);      // (workgroup-uniform)
                // window: where the region centre maps to on this level, minus half the window
                const int y0 = (int)floorf(pcy * H - 0.5f) - WH / 2 + 1;
                const int x0 = (int)floorf(pcx * W - 0.5f) - WW / 2 + 1;
                const int base_pix = st + y0 * W + x0;     // window row r -> pixel base_pix + (r / WW) * W + r % WW
                // ---- my samples on this level: loads first
                float gx[SPT], gy[SPT], ga[SPT];
#pragma unroll
                for (int sp = 0; sp < SPT; ++sp) {
                    const int k = l * P + (tid + sp * NT) % P, qq = qs[sp] >= 0 ? qs[sp] : qlist[0];
                    const int64_t srow = srow_of(qq);
                    io.load_xy(srow, (int64_t)n * Lq + qq, LP, k, l, P, H, W, gx[sp], gy[sp]);
                    ga[sp] = io.load_w(srow, LP, k);
                }
                int *const stats = stats2[l & 1];
                __syncthreads();                  // the previous level's walk and misses are done with the pool / start / rowlist
                // ---- geometry: cell of the top-left corner, four corner weights; count per cell
                int cell[SPT], rank[SPT];
                float4 cw[SPT];
#pragma unroll
                for (int sp = 0; sp < SPT; ++sp) {
                    cell[sp] = -1;
                    rank[sp] = 0;
                    cw[sp] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (qs[sp] < 0) continue;
                    int off[4];
                    float lw, lh;
                    if (!sample_setup(gx[sp], gy[sp], H, W, st, 1, off, lw, lh)) continue;      // off = pixel index of the corner or -1
                    float a = ga[sp];
                    if (IO::kSoftmax) a = __expf(a - sm_max[sp]) * sm_inv[sp];
                    const int h0 = (int)floorf(sub_rn(mul_rn(gy[sp], (float)H), 0.5f)), w0 = (int)floorf(sub_rn(mul_rn(gx[sp], (float)W), 0.5f));
                    mask_corners_idx(io, me, n, h0, w0, W, st + h0 * W + w0, off);      // padded pixels receive no gradient
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    const float none = __uint_as_float(kSwNoCorner);
                    cw[sp] = make_float4(off[0] >= 0 ? hh * hw * a : none, off[1] >= 0 ? hh * lw * a : none, off[2] >= 0 ? lh * hw * a : none,
                                         off[3] >= 0 ? lh * lw * a : none);
                    const int wy = h0 - y0, wx = w0 - x0;
                    if ((unsigned)wy < (unsigned)(WH - 1) && (unsigned)wx < (unsigned)(WW - 1)) {      // all four rows inside the window
                        cell[sp] = wy * WW + wx;
                        rank[sp] = atomicAdd(&cnt[cell[sp]], 1);
                    } else {                          // misses: corner by corner, {weight, slot << 23 | pixel (< 2^23, checked by the launcher)}
                        const int i = (tid + sp * NT) / P;
                        const float wv4[4] = {cw[sp].x, cw[sp].y, cw[sp].z, cw[sp].w};
#pragma unroll
                        for (int ci = 0; ci < 4; ++ci)
                            if (off[ci] >= 0) miss_end[-1 - atomicAdd(&stats[0], 1)] = make_float2(wv4[ci], __int_as_float((int)(((unsigned)i << 23) | (unsigned)off[ci])));
                    }
                }
                __syncthreads();
                // ---- ONE exclusive scan places the cells' entries AND compacts the non-empty rows: word = samples of the cell | (row has
                //      any contributor) << 16.  Thread t owns cells t * KC .. t * KC + KC - 1.
                {
                    constexpr int KC = (kWR + NT - 1) / NT;
                    int cv[KC], v = 0;
#pragma unroll
                    for (int kk = 0; kk < KC; ++kk) {
                        const int c = tid * KC + kk;
                        int word = 0;
                        if (c < kWR) {
                            const int up = c >= WW ? cnt[c - WW] : 0, upl = c >= WW + 1 ? cnt[c - WW - 1] : 0, lf = c >= 1 ? cnt[c - 1] : 0;
                            word = cnt[c] | ((cnt[c] + up + upl + lf) > 0 ? 0x10000 : 0);
                        }
                        cv[kk] = word;
                        v += word;
                    }
                    int incl = v;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const int t = __builtin_amdgcn_ds_bpermute((lane - d) << 2, incl);
                        if (lane >= d) incl += t;
                    }
                    if (lane == 63) wsum[wv] = incl;
                    __syncthreads();
                    int base = 0;
                    for (int w2 = 0; w2 < wv; ++w2) base += wsum[w2];
                    int run = base + incl - v;
#pragma unroll
                    for (int kk = 0; kk < KC; ++kk) {
                        const int c = tid * KC + kk;
                        if (c < kWR) {
                            start[c] = run & 0xffff;
                            if (cv[kk] & 0x10000) rowlist[run >> 16] = (unsigned short)c;
                        }
                        run += cv[kk];
                    }
                    if (tid == NT - 1) {
                        start[kWR] = run & 0xffff;
                        stats[1] = run & 0xffff;
                        stats[2] = (int)((unsigned)run >> 16);
                    }
                }
                __syncthreads();
                // ---- fill: my samples' entries at their cell's offset + rank (the cell counters are dead now: cleared for the next level,
                //      like the other parity's counters)
                for (int c = tid; c < kWR; c += NT) cnt[c] = 0;
                if (tid < 8) stats2[(l + 1) & 1][tid] = 0;
                const int nrows = stats[2], nmiss = stats[0];
#pragma unroll
                for (int sp = 0; sp < SPT; ++sp)
                    if (cell[sp] >= 0) {
                        const int at = start[cell[sp]] + rank[sp];
                        entw[at] = cw[sp];
                        ents[at] = (unsigned short)((tid + sp * NT) / P);
                    }
                __syncthreads();
                // ---- walk: streams of 16 lanes (lane l16 = channels l16, l16 + 16), each over the rows dealt to it
                if (!(SEMIDETR_SW_DBG & 8)) {
                    constexpr int NS = NT / 16;
                    const int sid = tid >> 4, l16 = tid & 15;
                    const float *wflat = reinterpret_cast<const float *>(entw);
                    const float2 *gt2 = reinterpret_cast<const float2 *>(gtile) + l16;
                    float *const gv16 = gvs + l16;
                    // non-empty rows are dealt round-robin: the four streams of a wave take four NEIGHBOURING rows (similar lengths)
                    for (int k = sid; k < nrows; k += NS) {
                        const int r = (int)rowlist[k];
                        // two contiguous entry ranges: cells (r - WW - 1, r - WW) -> bottom-right / bottom-left weights, cells (r - 1, r) ->
                        // top-right / top-left weights (cells of a window's last column / row hold no entries: no wrap-around)
                        const int a0 = r >= WW ? start[max(r - WW - 1, 0)] : 0, a1 = r >= WW ? start[r - WW + 1] : 0, sa = r >= WW ? start[r - WW] : 0;
                        const int b0 = start[max(r - 1, 0)], b1 = start[r + 1], sb = start[r];
                        float2 acc = make_float2(0.f, 0.f);
                        // eight entries per trip: their weights and slots first, then the eight grad_out pieces, then the FMAs (one entry
                        // at a time the walk paid two dependent LDS round trips per entry: 936 us)
                        const int la = a1 - a0, total = (SEMIDETR_SW_DBG & 1) ? 0 : la + (b1 - b0);
                        constexpr int WU = 8;
                        for (int i0 = 0; i0 < total; i0 += WU) {
                            float w[WU];
                            int sl[WU];
#pragma unroll
                            for (int u = 0; u < WU; ++u) {
                                const int i = min(i0 + u, total - 1);
                                const int e = i < la ? a0 + i : b0 + (i - la);
                                const int comp = i < la ? (e < sa ? 3 : 2) : (e < sb ? 1 : 0);
                                w[u] = wflat[e * 4 + comp];
                                sl[u] = (int)ents[e];
                            }
                            float2 g[WU];
#pragma unroll
                            for (int u = 0; u < WU; ++u) g[u] = gt2[sl[u] * 16];
#pragma unroll
                            for (int u = 0; u < WU; ++u) {
                                // (a corner that does not exist is never multiplied: 0 x inf; the tail repeats the last entry, unused)
                                const bool ok = i0 + u < total && __float_as_uint(w[u]) != kSwNoCorner;
                                acc.x = ok ? fmaf(w[u], g[u].x, acc.x) : acc.x;
                                acc.y = ok ? fmaf(w[u], g[u].y, acc.y) : acc.y;
                            }
                        }
                        const int py = y0 + r / WW, px = x0 + r % WW;
                        if (!(SEMIDETR_SW_DBG & 2) && (unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W) {      // (a row outside the level had only weightless corners)
                            float *pr = gv16 + (int64_t)(base_pix + (r / WW) * W + r % WW) * rs;
                            fp_atomic_add(pr, acc.x);
                            fp_atomic_add(pr + 16, acc.y);
                        }
                    }
                }
                // ---- misses: one row update per (sample, corner), 32 lanes per row
                for (int mi = tid >> 5; mi < ((SEMIDETR_SW_DBG & 4) ? 0 : nmiss); mi += NT / 32) {
                    const float2 en = miss_end[-1 - mi];
                    const int pk = __float_as_int(en.y);
                    const int ch = lane & 31;
                    fp_atomic_add(gvs + (int64_t)(pk & 0x7fffff) * rs + ch, en.x * gtile[((unsigned)pk >> 23) * kD + (ch & 15) * 2 + (ch >> 4)]);
                }
            }
        }
    }
}
