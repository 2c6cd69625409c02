// Measured-and-rejected D == 32 kernels that used to sit between the product kernels of msda_fast.h ("the
// product headers should contain exactly the kernels the code-object test names").  Included by msda.hip right after
// msda_fast.h, ONLY when SEMIDETR_EXPERIMENTS is set; every kernel here is reachable through semidetr_msda_set_variant
// (msda_experiments.h) and nothing in libsemidetr_hip.so.
#pragma once

// ---------------------------------------------------------------------------------------------
// grad_value for encoder self-attention (num_query == spatial_size: the queries ARE the pixels of the
// multi-scale map), fp32, D == 32, num_point == 4.  Runs after msda_bwd_gather_d32.
//
// The L2 atomic unit is the bottleneck of the plain backward (4 full-row atomics per sample; measured ceiling
// 10.4 G full-row fp32 atomics/s chip-wide, tools/atomic_probe.hip), but in the encoder neighbouring queries
// sample neighbouring pixels, so most of those atomics hit the same few rows.  LDS float atomics are no way
// out: ds_add_f32 runs lane-serially on gfx950 (~97 clk per 32-lane row vs ~9 for ds_add_u32,
// tools/lds_atomic_probe.hip).  So the scatter is turned into an OWNER-COMPUTES gather inside the workgroup,
// using only integer LDS atomics:
//   * a workgroup takes a TH x TW patch of query pixels of one level and one head; grad_out of the patch is
//     staged in LDS once (128 rows x 128 B);
//   * per sampling level it places a WH x WW window of value rows where the patch's own pixels map to on that
//     level, and buckets every (sample, corner) pair that falls inside the window by target row:
//     count (ds_add_rtn_u32) -> exclusive scan -> fill {corner weight x attention weight, row, query};
//   * the bucketed entries are sorted by row; every half-wave (lane = channel) walks an equal share of them,
//     keeps the running row sum  sum_i w_i * grad_out[q_i][c]  in a register and issues ONE full-line global
//     atomic per row run;
//   * corners outside the window (rare in the encoder) are put on a miss list and scattered one full-line
//     atomic each, exactly like the plain kernel -- any sampling pattern is correct, locality only decides speed.
// ---------------------------------------------------------------------------------------------

// TH x TW = query patch (pixels), WH x WW = value-row window per sampling level (both compile time).
// Every thread owns SPT = TH*TW*4 / 512 (query, point) samples of the patch.
// LDS of one windowed-scatter workgroup, carved from the dynamic allocation (so that the merged encoder backward launch
// can give the same bytes to a pair of gather blocks instead): sizes in bytes
template <int TH, int TW, int WH, int WW>
constexpr size_t win_lds_bytes()
{
    return (size_t)(TH * TW * kPT * 4 + 8) * 8 + (size_t)TH * TW * kD * 4 + (size_t)3 * WH * WW * 4 + 8 * 4 + (kWinThreads / 64) * 4 + 16;
}

template <typename IO, int TH, int TW, int WH, int WW>
__device__ __forceinline__ void win_scatter_body(
    const int b, float4 *smem, const float *__restrict__ gout, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int tiles_bound, float *__restrict__ gvalue)
{
    constexpr int kTQ = TH * TW, kWR = WH * WW, kNE = kTQ * kPT * 4, SPT = kTQ * kPT / kWinThreads;
    static_assert(SPT * kWinThreads == kTQ * kPT && SPT >= 1, "whole samples per thread");
    static_assert(kWR <= 2 * kWinThreads, "scan assigns two counters per thread");
    static_assert(kTQ <= 256 && kWR <= (1 << 14), "entry packing: 8 bits query, row above");
    float2 *entries = reinterpret_cast<float2 *>(smem);   // [kNE + 8] front: bucketed {weight, last << 30 | window row << 8 | query}
                                                          //        (+8: batch reads may run past a share's end, unused);
                                                          // back : misses {weight, query << 24 | pixel index}
    float *gtile = reinterpret_cast<float *>(entries + kNE + 8);          // [kTQ * kD] grad_out rows of the patch
    int *cnt = reinterpret_cast<int *>(gtile + kTQ * kD);                // per window row: count,
    int *start = cnt + kWR, *rowoff = start + kWR;                       //   first entry, element offset
    int (*stats2)[4] = reinterpret_cast<int (*)[4]>(rowoff + kWR);       // [2][4] stats double-buffered by level parity: a fast
    int *wsum = reinterpret_cast<int *>(stats2 + 2);                     //   wavefront may start level l+1 while others still read l's

    constexpr int P = kPT;
    const int Lq = S, LP = L * P, rs = M * kD;
    const int m = (b % M + (b / M) / kScatterHeadRun) % M;
    const int slot = (b / M) % tiles_bound, n = (b / M) / tiles_bound;
    const int tid = threadIdx.x, hw = tid >> 5, c = tid & 31, lane = tid & 63, wv = tid >> 6;

    // Patches are taken in REVERSE enumeration order: the coarse levels' patches come first.  They are the slow ones
    // (their footprint on the fine levels exceeds the window, so many of their corners go through the miss list),
    // and a launch is only a few waves of workgroups deep -- the slow ones must not be the last to start.
    int total_tiles = 0;
    for (int l = 0; l < L; ++l)
        total_tiles += (((int)shapes[2 * l] + TH - 1) / TH) * (((int)shapes[2 * l + 1] + TW - 1) / TW);
    for (int tile_f = slot; tile_f < total_tiles; tile_f += tiles_bound) {
        const Patch pt = find_patch<TH, TW>(total_tiles - 1 - tile_f, shapes, starts, L);
        if (pt.Hq == 0) break;
        // this thread's samples = (query i, point p) of the patch, sample index tid + sp * 512
        int qs[SPT];
        int64_t srow[SPT];
        float sm_max[SPT], sm_inv[SPT];
#pragma unroll
        for (int sp = 0; sp < SPT; ++sp) {
            const int sidx = tid + sp * kWinThreads, i = sidx / P, p = sidx - i * P;
            qs[sp] = patch_query<TW>(pt, i);
            srow[sp] = qs[sp] >= 0 ? ((int64_t)n * Lq + qs[sp]) * M + m : 0;
            // fused prologue: softmax statistics of the (query, head) row, once per patch.  The row's L*P logits
            // belong to the P threads of the query (a quad for P = 4), L each: quad reductions.
            sm_max[sp] = 0.f;
            sm_inv[sp] = 1.f;
            if (IO::kSoftmax) {
                float mx = -__builtin_huge_valf();
                if (qs[sp] >= 0)
                    for (int l = 0; l < L; ++l) mx = fmaxf(mx, io.load_w(srow[sp], LP, l * P + p));
                mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
                float sum = 0.f;
                if (qs[sp] >= 0)
                    for (int l = 0; l < L; ++l) sum += expf(io.load_w(srow[sp], LP, l * P + p) - mx);
                sum += __shfl_xor(sum, 1, 64);
                sum += __shfl_xor(sum, 2, 64);
                sm_max[sp] = mx;
                sm_inv[sp] = 1.f / sum;
            }
        }
        // patch centre in normalised coordinates (pixel centres are (i + 0.5) / size)
        const float pcy = (pt.y0 + 0.5f * TH) / (float)pt.Hq, pcx = (pt.x0 + 0.5f * TW) / (float)pt.Wq;
        __syncthreads();                      // previous patch fully done before its LDS state is reused
        for (int r = hw; r < kTQ; r += kWinThreads / 32) {      // stage grad_out of the patch, channels (c, c+16)
            const int rq = patch_query<TW>(pt, r);                        // interleaved: lane l of a 16-lane stream
            gtile[r * kD + (c & 15) * 2 + (c >> 4)] =                     // reads both with one ds_read_b64
                rq >= 0 ? gout[(((int64_t)n * Lq + rq) * M + m) * kD + c] : 0.f;
        }
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)starts[l];
            // window: where the patch centre maps to on this level, minus half the window
            const int y0 = (int)floorf(pcy * H - 0.5f) - WH / 2 + 1;
            const int x0 = (int)floorf(pcx * W - 0.5f) - WW / 2 + 1;
            int *stats = stats2[l & 1];
            if (tid < 4) stats[tid] = 0;
            for (int k = tid; k < kWR; k += kWinThreads) cnt[k] = 0;
            // ---- this thread's sample geometry: per corner weight, window row (or -1: miss, -2: no corner)
            float cw[SPT][4];
            int wrow[SPT][4], rank[SPT][4], pix[SPT][4];
#pragma unroll
            for (int sp = 0; sp < SPT; ++sp) {
                const int sidx = tid + sp * kWinThreads, p = sidx % P;
                int off[4] = {-1, -1, -1, -1};
                float lw = 0.f, lh = 0.f, a = 0.f;
                int h0 = 0, w0 = 0;
                if (qs[sp] >= 0) {
                    const int k = l * P + p;
                    float x, y;
                    io.load_xy(srow[sp], (int64_t)n * Lq + qs[sp], LP, k, l, P, H, W, x, y);
                    if (sample_setup(x, y, H, W, st, rs, off, lw, lh)) {
                        a = io.load_w(srow[sp], LP, k);
                        if (IO::kSoftmax) a = expf(a - sm_max[sp]) * sm_inv[sp];
                        // the top-left corner (h0, w0) exactly as sample_setup derived it
                        h0 = (int)floorf(sub_rn(mul_rn(y, (float)H), 0.5f));
                        w0 = (int)floorf(sub_rn(mul_rn(x, (float)W), 0.5f));
                    }
                }
                const int wy = h0 - y0, wx = w0 - x0;
                const bool in_y0 = (unsigned)wy < (unsigned)WH, in_y1 = (unsigned)(wy + 1) < (unsigned)WH;
                const bool in_x0 = (unsigned)wx < (unsigned)WW, in_x1 = (unsigned)(wx + 1) < (unsigned)WW;
                const int wi = wy * WW + wx;
                const float hh = 1.f - lh, hwt = 1.f - lw;
                const float cwv[4] = {hh * hwt * a, hh * lw * a, lh * hwt * a, lh * lw * a};
                const bool inw[4] = {in_y0 && in_x0, in_y0 && in_x1, in_y1 && in_x0, in_y1 && in_x1};
                const int wr[4] = {wi, wi + 1, wi + WW, wi + WW + 1};
#pragma unroll
                for (int cidx = 0; cidx < 4; ++cidx) {
                    cw[sp][cidx] = cwv[cidx];
                    wrow[sp][cidx] = off[cidx] < 0 ? -2 : (inw[cidx] ? wr[cidx] : -1);
                    pix[sp][cidx] = off[cidx] / rs;      // pixel index inside the image (misses only)
                    rank[sp][cidx] = 0;
                }
            }
            __syncthreads();                  // counters zeroed, previous level's walk finished
            // ---- bucket the in-window corners by window row (count), list the others as misses
#pragma unroll
            for (int sp = 0; sp < SPT; ++sp) {
                const int i = (tid + sp * kWinThreads) / P;
#pragma unroll
                for (int cidx = 0; cidx < 4; ++cidx) {
                    if (wrow[sp][cidx] >= 0) rank[sp][cidx] = atomicAdd(&cnt[wrow[sp][cidx]], 1);
                    else if (wrow[sp][cidx] == -1)   // keep the pixel index (< 2^24, checked by the launcher) + query
                        entries[kNE - 1 - atomicAdd(&stats[1], 1)] = make_float2(
                            cw[sp][cidx], __int_as_float((int)(((unsigned)i << 24) | (unsigned)pix[sp][cidx])));
                }
            }
            __syncthreads();
            // ---- exclusive scan of the kWR counters -> start[]  (thread t owns counters 2t, 2t+1)
            {
                const int j0 = tid * 2;
                const int c0 = j0 < kWR ? cnt[j0] : 0, c1 = j0 + 1 < kWR ? cnt[j0 + 1] : 0;
                const int v = c0 + c1;
                int incl = v;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int t = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += t;
                }
                if (lane == 63) wsum[wv] = incl;
                __syncthreads();
                int base = 0;
                for (int w2 = 0; w2 < wv; ++w2) base += wsum[w2];
                const int excl = base + incl - v;
                if (j0 < kWR) start[j0] = excl;
                if (j0 + 1 < kWR) start[j0 + 1] = excl + c0;
                // element offset of the window row's pixel inside the image slice (only used for touched rows, which
                // are always pixels of the level)
                if (j0 < kWR) rowoff[j0] = (st + (y0 + j0 / WW) * W + x0 + j0 % WW) * rs;
                if (j0 + 1 < kWR) rowoff[j0 + 1] = (st + (y0 + (j0 + 1) / WW) * W + x0 + (j0 + 1) % WW) * rs;
                if (tid == kWinThreads - 1) stats[3] = excl + v;            // total number of bucketed entries
            }
            __syncthreads();
            // ---- fill the buckets; bit 30 marks the last entry of its row
#pragma unroll
            for (int sp = 0; sp < SPT; ++sp) {
                const int i = (tid + sp * kWinThreads) / P;
#pragma unroll
                for (int cidx = 0; cidx < 4; ++cidx) {
                    const int wr = wrow[sp][cidx];
                    if (wr >= 0)
                        entries[start[wr] + rank[sp][cidx]] = make_float2(
                            cw[sp][cidx], __int_as_float((rank[sp][cidx] == cnt[wr] - 1 ? (1 << 30) : 0) | (wr << 8) | i));
                }
            }
            __syncthreads();
            // ---- owner computes: 32 streams of 16 lanes (lane l = channels l and l+16) each walk an equal share
            //      of the row-sorted entries, keep the running row sum in two registers and flush a finished row
            //      with two half-line atomics (64 contiguous bytes each = the same 2 atomic units as one full row)
            {
                constexpr int kStreams = kWinThreads / 16;
                const int sid = tid >> 4, l16 = tid & 15;
                const float2 *gt2 = reinterpret_cast<const float2 *>(gtile);
                float *gvs = gvalue + ((int64_t)n * S * M + m) * kD + l16;
                const int total = stats[3];
                const int lo = (int)((int64_t)total * sid / kStreams);
                const int hi = (int)((int64_t)total * (sid + 1) / kStreams);
                // acc += w * grad_out[q]; an entry flagged "last of its row" flushes the row sum (the row's element
                // offset comes from the rowoff table) and clears it.  A share that ends in the middle of a row
                // flushes its partial sum at the end; one that starts in the middle simply starts from zero.
                int cur = -1;           // row of the most recent entry whose sum is still open, or -1
                float2 accv = make_float2(0.f, 0.f);
                auto flush = [&](int rowi) {
                    float *pr = gvs + rowoff[rowi];
                    fp_atomic_add(pr, accv.x);
                    fp_atomic_add(pr + 16, accv.y);
                };
                auto step = [&](const float2 &en, const float2 &gq) {
                    const int pk = __float_as_int(en.y);
                    accv.x += en.x * gq.x;
                    accv.y += en.x * gq.y;
                    cur = (pk >> 8) & 0x3fffff;
                    if (pk & (1 << 30)) {
                        flush(cur);
                        accv = make_float2(0.f, 0.f);
                        cur = -1;
                    }
                };
                // software pipeline: 8 independent entry reads, then 8 independent grad_out reads, then the short
                // dependent chain.  Full batches run without bounds checks.
                int e = lo;
                for (; e + 8 <= hi; e += 8) {
                    float2 en[8], gq[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) en[u] = entries[e + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) gq[u] = gt2[(__float_as_int(en[u].y) & 255) * 16 + l16];
#pragma unroll
                    for (int u = 0; u < 8; ++u) step(en[u], gq[u]);
                }
                if (e < hi) {       // tail of < 8 entries (reads stay inside the padded array)
                    float2 en[8], gq[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) en[u] = entries[e + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) gq[u] = gt2[(__float_as_int(en[u].y) & 255) * 16 + l16];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (e + u < hi) step(en[u], gq[u]);
                }
                if (cur >= 0) flush(cur);
                // ---- misses: one row update per (sample, corner), as the plain kernel does
                const int nmiss = stats[1];
                for (int mi = sid; mi < nmiss; mi += kStreams) {
                    const float2 en = entries[kNE - 1 - mi];
                    const int pk = __float_as_int(en.y);
                    const float2 g2 = gt2[((unsigned)pk >> 24) * 16 + l16];
                    float *pr = gvs + (int64_t)(pk & 0xffffff) * rs;
                    fp_atomic_add(pr, en.x * g2.x);
                    fp_atomic_add(pr + 16, en.x * g2.y);
                }
            }
        }
    }
}

template <typename IO, int TH, int TW, int WH, int WW>
__global__ __launch_bounds__(kWinThreads, 4) void msda_bwd_scatter_d32_win(
    const float *__restrict__ gout, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const IO io, int S, int M, int L, int tiles_bound, float *__restrict__ gvalue)
{
    extern __shared__ float4 smem[];
    win_scatter_body<IO, TH, TW, WH, WW>((int)blockIdx.x, smem, gout, shapes, starts, io, S, M, L, tiles_bound, gvalue);
}

// ONE launch for both halves of the encoder backward: every `period`-th workgroup runs the windowed scatter, the others
// run two 4 x 8-patch gather blocks each.  The halves write disjoint outputs and lean on different units (LDS / issue /
// atomics vs the vector-memory path); in one launch they share the CUs from the first microsecond instead of queueing,
// and without the cross-stream events that made two-stream overlap lose.  All workgroups carry the scatter's LDS size,
// so any mix of two fits a CU.
template <typename IO, int KLP, int TH, int TW, int WH, int WW>
__global__ __launch_bounds__(kWinThreads, 4) void msda_bwd_enc_merged(
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int P, int tiles_bound, int scatter_blocks,
    int gather_bound, int gather_blocks, int period, float *__restrict__ gvalue)
{
    extern __shared__ float4 smem[];
    const int b = (int)blockIdx.x;
    // blocks 0, period, 2*period, ... (while scatter blocks remain) are scatter blocks
    const int ns_before = min((b + period - 1) / period, scatter_blocks);      // scatter blocks with index < b
    const bool is_scatter = (b % period == 0) && (b / period < scatter_blocks);
    if (is_scatter) {
        win_scatter_body<IO, TH, TW, WH, WW>(b / period, smem, gout, shapes, starts, io, S, M, L, tiles_bound, gvalue);
        return;
    }
    const int gi = b - ns_before;                                              // index among the gather workgroups
    const int half = (int)threadIdx.x >> 8;
    const int vb = 2 * gi + half;
    const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
    gather_body<IO, KLP, 408, true>(vb, (int)threadIdx.x & 255, smem + half * half_f4, vb < gather_blocks, gout, value,
                                    shapes, starts, io, S, M, L, S, P, gather_bound);
}

// ---------------------------------------------------------------------------------------------
// Encoder self-attention forward with the COARSE LEVELS LDS-RESIDENT (fp32, D == 32).
//
// The patch kernel above is bound by the vector-memory path, not by HBM (TA busy ~80 %, 5.8 GB of corner rows per
// bs-4 launch through the L1 at ~64 B/clk/CU).  The only lever is to take corner reads off that
// path.  A whole coarse level of ONE (image, head) is small -- 13 x 21 pixels x 128 B = 35 KB at 800 x 1333 -- so a
// workgroup that stays with one (image, head) can keep it in LDS for its whole life and serve every sample of that
// level (P of the L*P samples of every query: 25 % of all corner reads for the DINO pyramid) with ds_read_b128.
// Differences from the LDS-window experiments that lost twice: nothing is re-staged per patch (no
// halo traffic, no per-level barriers, no miss bookkeeping -- a resident level is resident completely), and the
// fine levels keep their 16 independent buffer loads in flight exactly as before.
//   * 512 threads = two 4 x 8 query patches at a time; a workgroup takes kResGroup consecutive patches of one
//     (image, head) -- re-staging the resident rows costs 35 KB per 8 patches, +1.7 % of the 262 KB of corner rows a
//     patch reads -- and workgroups are numbered / rotated over heads exactly like the plain patch kernel, so the
//     dispatch order, L2 locality and XCD balance that kernel was tuned for are kept.  (A first version with
//     persistent workgroups pinned to one (image, head) for the whole launch had every XCD working on all 32 (image,
//     head) slices at once and was slower than the plain kernel: 290 vs 245 us at bs 4.)
//   * resident levels = the longest suffix of the pyramid whose pixels fit kResRows rows (chosen on the device from
//     the level table); requires SEMIDETR_MSDA_QUERIES_ARE_PIXELS (levels tile [0, S) contiguously).
//   * corners outside a resident level read a zero row kept behind the resident rows (the op's zero padding).
// ---------------------------------------------------------------------------------------------
constexpr int kResRows = 320;            // 40 KB of value rows per workgroup
template <typename IO, int kResGroup = 8>   // patches per workgroup (even: two are processed at a time)
__global__ __launch_bounds__(512, 4) void msda_fwd_d32_res(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const IO io, int S, int M, int L, int P, int groups_per_image, int res_rows_max, float *__restrict__ out)
{
    constexpr int RPB = 32, PH = 4, PW = 8;
    extern __shared__ float4 smem[];
    const int LP = L * P, LPP = LP + 1, Lq = S;
    float4 *res = smem;                                              // (kResRows + 1) rows x 8 float4
    int4 *rec_off_all = reinterpret_cast<int4 *>(smem + (kResRows + 1) * 8);
    float4 *rec_w_all = smem + (kResRows + 1) * 8 + 2 * RPB * LPP;

    const int b = blockIdx.x, g = b / M;
    const int m = (b % M + g / (kHeadRun / kResGroup > 0 ? kHeadRun / kResGroup : 1)) % M;   // head rotation as in tile_of_block
    const int n = g / groups_per_image, slot = g % groups_per_image;
    const int tid = threadIdx.x, sub = tid >> 8, t = tid & 255;
    int4 *rec_off = rec_off_all + sub * RPB * LPP;
    float4 *rec_w = rec_w_all + sub * RPB * LPP;
    const int rs = M * kD;

    // ---- resident suffix of the pyramid
    int res_from = L, res_rows = 0;
    for (int l = L - 1; l >= 0; --l) {
        const int hw_ = (int)shapes[2 * l] * (int)shapes[2 * l + 1];
        if (res_rows + hw_ > res_rows_max) break;
        res_rows += hw_;
        res_from = l;
    }
    const int res_start = res_from < L ? (int)starts[res_from] : S;
    {
        const float4 *src = reinterpret_cast<const float4 *>(value + ((int64_t)n * S + res_start) * rs + m * kD);
        for (int i = tid; i < res_rows * 8; i += 512) res[i] = src[(int64_t)(i >> 3) * (rs / 4) + (i & 7)];
        if (tid < 8) res[res_rows * 8 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const unsigned zero_row = (unsigned)res_rows * 128u;
    const int kres = res_from * P;                                   // samples k >= kres are served from LDS

    const __amdgpu_buffer_rsrc_t vr = image_rsrc(value + (int64_t)n * S * M * kD, (unsigned)S * M * kD * 4u);
    // groups_per_image is a grid sizing hint: a workgroup takes patch groups slot, slot + hint, ... (see msda_fwd_d32)
    for (int tile = slot * kResGroup + sub;; tile += (tile % kResGroup >= kResGroup - 2) ? (groups_per_image - 1) * kResGroup + 2 : 2) {
        // both halves of the workgroup must take the same number of barriers: a half without a patch idles through them
        const Patch pt = find_patch<PH, PW>(tile, shapes, starts, L);
        const Patch p0 = sub ? find_patch<PH, PW>(tile - 1, shapes, starts, L) : pt;
        if (p0.Hq == 0) return;                                      // the pair's first patch does not exist: done
        __syncthreads();                                             // previous pair done with the records (and res loaded)
        // ---- phase 1: sample records
        if (pt.Hq)
            for (int s = t; s < RPB * LP; s += 256) {
                const int r = s / LP, k = s - r * LP;
                const int q = patch_query<PW>(pt, r);
                unsigned off[4] = {kOob, kOob, kOob, kOob};
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                const int l = k / P;
                const bool resident = l >= res_from;
                if (resident) off[0] = off[1] = off[2] = off[3] = zero_row;
                if (q >= 0) {
                    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)starts[l];
                    const int64_t nq = (int64_t)n * Lq + q, row = nq * M + m;
                    float x, y, lw, lh;
                    io.load_xy(row, nq, LP, k, l, P, H, W, x, y);
                    const float a = row_softmax(io, row, LP, k, io.load_w(row, LP, k));
                    unsigned o4[4];
                    // resident levels: row pitch 128 B, pixel index relative to the first resident pixel
                    if (sample_setup_oob(x, y, H, W, resident ? st - res_start : st, resident ? 128u : (unsigned)rs * 4u,
                                         o4, lw, lh)) {
                        const float hh = 1.f - lh, hw = 1.f - lw;
                        w = make_float4(a * (hh * hw), a * (hh * lw), a * (lh * hw), a * (lh * lw));
#pragma unroll
                        for (int i = 0; i < 4; ++i) off[i] = (resident && o4[i] == kOob) ? zero_row : o4[i];
                    }
                }
                rec_off[r * LPP + k] = make_int4((int)off[0], (int)off[1], (int)off[2], (int)off[3]);
                rec_w[r * LPP + k] = w;
            }
        __syncthreads();
        if (!pt.Hq) continue;
        // ---- phase 2: gather + weighted sum; fine levels through the buffer path, resident levels from LDS
        const int r = t >> 3, j = t & 7;
        const int q = patch_query<PW>(pt, r);
        const unsigned lane_b = (unsigned)(m * kD + 4 * j) * 4u;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int4 *ro = rec_off + r * LPP;
        const float4 *rw = rec_w + r * LPP;
        auto fma4 = [&](const float4 &w, const float4 &v1, const float4 &v2, const float4 &v3, const float4 &v4) {
            acc.x += w.x * v1.x + w.y * v2.x + w.z * v3.x + w.w * v4.x;
            acc.y += w.x * v1.y + w.y * v2.y + w.z * v3.y + w.w * v4.y;
            acc.z += w.x * v1.z + w.y * v2.z + w.z * v3.z + w.w * v4.z;
            acc.w += w.x * v1.w + w.y * v2.w + w.z * v3.w + w.w * v4.w;
        };
        const char *resb = reinterpret_cast<const char *>(res) + j * 16;
#pragma unroll 4
        for (int k = 0; k < kres; ++k) {
            const int4 o = ro[k];
            const float4 w = rw[k];
            fma4(w, buf_ld4(vr, (unsigned)o.x + lane_b), buf_ld4(vr, (unsigned)o.y + lane_b),
                 buf_ld4(vr, (unsigned)o.z + lane_b), buf_ld4(vr, (unsigned)o.w + lane_b));
        }
#pragma unroll 4
        for (int k = kres; k < LP; ++k) {
            const int4 o = ro[k];
            const float4 w = rw[k];
            fma4(w, *reinterpret_cast<const float4 *>(resb + o.x), *reinterpret_cast<const float4 *>(resb + o.y),
                 *reinterpret_cast<const float4 *>(resb + o.z), *reinterpret_cast<const float4 *>(resb + o.w));
        }
        if (q >= 0) {
            const int64_t row = ((int64_t)n * Lq + q) * M + m;
            *reinterpret_cast<float4 *>(out + row * kD + 4 * j) = acc;
        }
    }
}

template <typename IO>
__global__ __launch_bounds__(kLvlThreads) void msda_bwd_scatter_d32_lvl(
    const float *__restrict__ gout, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts, const IO io,
    int S, int M, int L, int Lq, int P, int chunks, int chunk_q, float *__restrict__ gvalue)
{
    extern __shared__ float4 smem[];
    lvl_scatter_body<IO>((int)blockIdx.x, smem, gout, shapes, starts, io, S, M, L, Lq, P, chunks, chunk_q, gvalue);
}

// ONE launch for both halves of the backward of an arbitrary query set: workgroups [0, scatter_blocks) run the
// level-aggregated scatter above, the rest run the gather (two 256-thread gather blocks per 512-thread workgroup).  The
// halves write disjoint outputs (grad_value vs grad_sampling_loc / grad_attn_weight) and stress different units (LDS +
// atomics vs the vector-memory path), so they overlap instead of queueing -- without the cross-stream events that made
// the side-stream experiment lose.  Scatter workgroups come first: they are the long pole.
template <typename IO, int KLP>
__global__ __launch_bounds__(kLvlThreads) void msda_bwd_lvl_merged(
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int Lq, int P, int chunks, int chunk_q,
    int scatter_blocks, int gather_tiles, int gather_blocks, float *__restrict__ gvalue)
{
    extern __shared__ float4 smem[];
    if ((int)blockIdx.x < scatter_blocks) {
        lvl_scatter_body<IO>((int)blockIdx.x, smem, gout, shapes, starts, io, S, M, L, Lq, P, chunks, chunk_q, gvalue);
        return;
    }
    const int half = (int)threadIdx.x >> 8;
    const int vb = 2 * ((int)blockIdx.x - scatter_blocks) + half;
    const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;       // LDS of one gather block, in float4
    gather_body<IO, KLP, 0>(vb, (int)threadIdx.x & 255, smem + half * half_f4, vb < gather_blocks, gout, value, shapes,
                            starts, io, S, M, L, Lq, P, gather_tiles);
}

// EXPERIMENT (backward variant 902): the merged launch WITHOUT the hipMemsetAsync in front of it.  Roles are dealt out by an
// atomic TICKET taken when a workgroup starts (not by blockIdx, so nothing depends on the dispatch order): the first
// ceil(gather_blocks / 2) tickets zero a slice of grad_value each, release it (agent-scope fence: the lines leave the XCD's
// L2 -- the row atomics are performed memory-side), count themselves done and run the gather; every later ticket is a scatter
// workgroup, which sorts its samples as usual and only then -- right before its first row atomic -- waits for the done
// count.  Whoever holds a scatter ticket knows that all fillers already RUN (they drew their tickets earlier) and fillers
// never wait for anything: no co-residency assumption, no deadlock.  The wait is bounded anyway (`spin_limit` polls, then
// it proceeds and raises g_dest_dbg[15] -- results are wrong in that case, a hang is worse).  The last workgroup to finish
// zeroes the three counters, so a slot is clean again when the kernel ends.
struct FillWait {
    const unsigned *done;
    unsigned need;
    int spin_limit;
    __device__ __forceinline__ void operator()() const
    {
        if (threadIdx.x == 0) {
            int it = 0;
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                if (++it >= spin_limit) { SEMIDETR_DBG_ADD(15, 1); break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
};

template <typename IO, int KLP>
__global__ __launch_bounds__(kLvlThreads) void msda_bwd_lvl_coop(
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int Lq, int P, int chunks, int chunk_q,
    int scatter_blocks, int gather_tiles, int gather_blocks, float *__restrict__ gvalue, float4 *__restrict__ zero,
    int64_t zero_n4, unsigned *__restrict__ sync, int spin_limit)
{
    extern __shared__ float4 smem[];
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&sync[0], 1u);
    __syncthreads();
    const unsigned ticket = s_ticket, fillers = (unsigned)(gather_blocks + 1) / 2;
    if (ticket < fillers) {
        const int64_t per = (zero_n4 + fillers - 1) / fillers;
        const int64_t lo = (int64_t)ticket * per, hi = lo + per < zero_n4 ? lo + per : zero_n4;
        // write-through stores (sc0 sc1): the zeros go to memory, where the row atomics are performed, without the L2
        // write-back an agent-scope release fence costs (buffer_wbl2 of the whole XCD L2 per wave: 36 -> 87 us at the
        // micro-benchmark shape); vmcnt(0) = written
        {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f z = {0.f, 0.f, 0.f, 0.f};
            for (int64_t i = lo + threadIdx.x; i < hi; i += kLvlThreads) {
                float4 *dst = zero + i;
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(z) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&sync[1], 1u);
        const int half = (int)threadIdx.x >> 8;
        const int vb = 2 * (int)ticket + half;
        const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
        gather_body<IO, KLP, 0>(vb, (int)threadIdx.x & 255, smem + half * half_f4, vb < gather_blocks, gout, value, shapes,
                                starts, io, S, M, L, Lq, P, gather_tiles);
    } else if ((int)(ticket - fillers) < scatter_blocks) {
        lvl_scatter_body<IO, FillWait>((int)(ticket - fillers), smem, gout, shapes, starts, io, S, M, L, Lq, P, chunks, chunk_q,
                                       gvalue, FillWait{sync + 1, fillers, spin_limit});
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned f = atomicAdd(&sync[2], 1u);
        if (f == gridDim.x - 1) {        // last one out: every ticket is drawn, every filler counted
            atomicExch(&sync[0], 0u);
            atomicExch(&sync[1], 0u);
            atomicExch(&sync[2], 0u);
        }
    }
}
