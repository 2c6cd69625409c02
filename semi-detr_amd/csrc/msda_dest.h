// grad_value for encoder self-attention, DESTINATION-OWNED (fp32, D == 32).  Included by msda.hip after msda_fast.h.
//
// Why: the windowed kernel (msda_bwd_scatter_d32_win) is owned by a patch of SOURCE queries, so every patch flushes the
// halo rows it shares with its neighbours, and every query level flushes its own copy of the rows it hits on every
// sampling level: 6.6 flushed rows per (query, head) at the 800x1333 encoder shape where one would do (measured:
// 604 MB of row atomics for a 91 MB grad_value, profiles/r01_msda_enc_bs4_pmc_WRITE_SIZE.txt), and even with
// infinitely large patches a source-owned formulation cannot get below 4 rows per query (a level-1 query patch
// touches 4 level-0 rows per query, a level-2 patch 16, ...).  Here a workgroup owns a TH x TW tile of grad_value
// rows of ONE (image, head, level) and goes looking for the samples that land there:
//
//   enumerate  the candidate queries of ALL query levels: those whose own pixel centre, mapped to the tile's level,
//              lies within R pixels of the tile (an integer predicate every workgroup evaluates identically); for each
//              (query, point) of the tile's level compute the sample; keep it when one of its corners is a row of the
//              tile.  Kept samples go to an LDS list {lw, lh, attention, query} by wave-aggregated compaction;
//   sort       the list by the sample's top-left CELL (counting sort with integer LDS atomics: ds_add_f32 is
//              lane-serial on gfx950, tools/lds_atomic_probe.hip) -- one insert per sample, not per corner;
//   accumulate one half-wave (lane = channel) per cell walks the cell's samples, reads grad_out[query] (one 128-byte
//              row per sample, from L1/L2) and keeps the four corner sums in registers; cells are processed in four
//              colour classes ((y & 1, x & 1)) whose 2 x 2 footprints are disjoint, so the corner sums are added to the
//              LDS tile with plain read-modify-write -- no float atomics anywhere;
//   flush      every touched row of the tile ONCE, as one full-line global atomic (the atomic is only there because a
//              coarse tile's candidates are split over several workgroups, and for the out-of-reach path below).
//
// A corner is owned by the workgroup of the tile it lies in if that workgroup enumerates the query (reach <= R), else
// by the query's HOME workgroup (the tile its own pixel maps to), which scatters it with one full-line atomic like the
// plain kernel.  Any sampling pattern is therefore correct; locality only decides speed.
//
// Rows flushed per launch at the encoder shape, bs 4: ~1.2 M (N*S*M = 711 k rows, coarse tiles split 3..11 ways)
// against 4.7 M for the windowed kernel.  The price is the halo on the QUERY side: (TH + 2R)(TW + 2R) / (TH TW)
// candidates per kept sample on the fine levels (4x at 16 x 16, R = 8) -- 12 bytes of sampling_loc / attn_weight and
// ~35 VALU instructions each, no grad_out traffic.
#pragma once

constexpr int kDestThreads = 512;
// instrumented build (DBG == 2): per-phase cycle counts summed over workgroups, read back by semidetr_debug_counters
// g_dest_dbg[16] (debug / instrumentation counters read by semidetr_debug_counters) is defined in msda_fast.h
constexpr int kDestMaxLevels = 8;        // LDS tables of the kernel (pyramids with more levels take the windowed kernel)

template <typename IO, int TH, int TW, int R, int DBG = 0>
__global__ __launch_bounds__(kDestThreads, 4) void msda_bwd_dest_d32(
    const float *__restrict__ gout, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const IO io, int S, int M, int L, int P, int units_bound, float *__restrict__ gvalue)
{
    constexpr int NT = kDestThreads, EMAX = 2048, BATCH = 1024, CAPS = 8192, MISSMAX = 128, ML = kDestMaxLevels;
    // cells (top-left corners) iy in [0, TH], ix in [0, TW] (tile-local corner + 1), numbered COLOUR-MAJOR: colour =
    // (iy & 1) * 2 + (ix & 1); cells of one colour have disjoint 2 x 2 footprints and are contiguous in the sorted list
    constexpr int kRows = TH * TW, NXC = (TW + 2) / 2, CC = NXC * ((TH + 2) / 2), kCells = 4 * CC;
    static_assert(kCells <= NT, "the scan gives one cell counter to each thread");
    static_assert(EMAX <= 0x8000, "order[] keeps the first-of-cell flag in bit 15");
    static_assert(EMAX % NT == 0 && BATCH % NT == 0 && BATCH <= EMAX, "whole iterations");
    static_assert((TW & (TW - 1)) == 0, "row -> (y, x) by shifts");
    __shared__ float acc[kRows * kD];                 // the tile of grad_value rows being built
    __shared__ float4 ent[EMAX];                      // kept samples {lw, lh, attention, bits(query)}
    __shared__ unsigned short cellid[EMAX], order[EMAX];
    __shared__ float4 miss[MISSMAX];                  // out-of-reach corners {weight, bits(query), bits(pixel), -}
    __shared__ int cnt[kCells], start[kCells + 1];
    __shared__ unsigned char touched[kRows];
    __shared__ int lv_h[ML], lv_w[ML], lv_st[ML], lv_units[ML], lv_splits[ML];
    __shared__ int rect[ML][4], pre[ML + 1];          // candidate rectangle / sample prefix per query level
    __shared__ int ctl[4], wsum[NT / 64];             // ctl[0..2]: rotating per-batch list counters, ctl[3]: miss counter

    const int Lq = S, LP = L * P, rs = M * kD;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hw = tid >> 5, c = tid & 31;
    const int b = blockIdx.x;
    const int m = (b % M + (b / M) / kScatterHeadRun) % M;
    const int slot = (b / M) % units_bound, n = (b / M) / units_bound;

    if (tid < L) {
        lv_h[tid] = (int)shapes[2 * tid];
        lv_w[tid] = (int)shapes[2 * tid + 1];
        lv_st[tid] = (int)starts[tid];
    }
    if (tid < 4) ctl[tid] = 0;
    __syncthreads();
    if (tid < L) {      // units of level `tid`: tiles x splits, splits from an upper bound of a tile's candidate count
        const int H = lv_h[tid], W = lv_w[tid];
        int cand = 0;
        for (int lq = 0; lq < L; ++lq) {
            const int Hq = lv_h[lq], Wq = lv_w[lq];
            cand += min(Hq, ((TH + 2 * R) * Hq) / H + 6) * min(Wq, ((TW + 2 * R) * Wq) / W + 6) * P;
        }
        const int sp = (cand + CAPS - 1) / CAPS;
        lv_splits[tid] = sp;
        lv_units[tid] = ((H + TH - 1) / TH) * ((W + TW - 1) / TW) * sp;
    }
    __syncthreads();
    int total_units = 0;
    for (int l = 0; l < L; ++l) total_units += lv_units[l];

    const float *gb = gout + ((int64_t)n * Lq * M + m) * kD + c;           // + query * rs
    float *gvb = gvalue + ((int64_t)n * S * M + m) * kD + c;               // + pixel * rs

    int bi = 0;                                       // batches enumerated so far by this workgroup
    for (int unit = slot; unit < total_units; unit += units_bound) {
        // ---- unit -> (level, tile, split); coarse levels first (their units carry the most samples)
        int l = L - 1, u = unit;
        while (u >= lv_units[l]) { u -= lv_units[l]; --l; }
        const int H = lv_h[l], W = lv_w[l], st = lv_st[l], splits = lv_splits[l];
        const int tile = u / splits, split = u - tile * splits;
        const int ntx = (W + TW - 1) / TW;
        const int ty = tile / ntx, tx = tile - ty * ntx;
        const int ty0 = ty * TH, tx0 = tx * TW;
        const int THc = min(TH, H - ty0), TWc = min(TW, W - tx0);          // rows of the tile that exist
        const int lo_y = ty0 - R, hi_y = ty0 + TH - 1 + R, lo_x = tx0 - R, hi_x = tx0 + TW - 1 + R;
        __syncthreads();                      // previous unit completely done with the LDS state
        if (tid < L) {                        // candidate rectangle on query level `tid` (a superset; the exact integer
            const int Hq = lv_h[tid], Wq = lv_w[tid];                      // predicate is evaluated per candidate)
            const int ay = max(lo_y, 0), by = min(hi_y, H - 1), ax = max(lo_x, 0), bx = min(hi_x, W - 1);
            const int y0 = max(0, (ay * Hq) / H - 1), y1 = min(Hq - 1, ((by + 1) * Hq + H - 1) / H + 1);
            const int x0 = max(0, (ax * Wq) / W - 1), x1 = min(Wq - 1, ((bx + 1) * Wq + W - 1) / W + 1);
            rect[tid][0] = y0; rect[tid][1] = x0; rect[tid][2] = y1 - y0 + 1; rect[tid][3] = x1 - x0 + 1;
        }
        for (int i = tid; i < kRows * kD; i += NT) acc[i] = 0.f;
        if (tid < kRows) touched[tid] = 0;
        __syncthreads();
        if (tid == 0) {
            int s = 0;
            for (int lq = 0; lq < L; ++lq) { pre[lq] = s; s += rect[lq][2] * rect[lq][3] * P; }
            pre[L] = s;
        }
        __syncthreads();
        const int c_lo = split * CAPS, c_hi = min(pre[L], c_lo + CAPS);
        unsigned long long tmark = DBG >= 2 ? __builtin_readcyclecounter() : 0ull;
        const float inv_p = 1.0f / (float)P;

        // one sort + accumulate round over the `ne` samples listed so far
        auto lap = [&](int slot_) {
            if (DBG >= 2 && tid == 0) {
                const unsigned long long now = __builtin_readcyclecounter();
                SEMIDETR_DBG_ADD(slot_, now - tmark);
                tmark = now;
            } else if (DBG >= 2) {
                tmark = 0;
            }
        };
        auto round = [&](int ne) {
            lap(0);                  // 0: enumeration since the last lap
            if (DBG >= 2 && tid == 0) { atomicAdd(&g_dest_dbg[8], 1ull); atomicAdd(&g_dest_dbg[9], (unsigned long long)ne); }
            if (DBG == 1) {          // timing aid: enumeration only
                __syncthreads();
                if (tid == 0) ctl[3] = 0;
                __syncthreads();
                return;
            }
            if (tid < kCells) cnt[tid] = 0;
            __syncthreads();
            int rk[EMAX / NT];
#pragma unroll
            for (int i = 0; i < EMAX / NT; ++i) {
                const int e = tid + i * NT;
                rk[i] = e < ne ? atomicAdd(&cnt[cellid[e]], 1) : 0;
            }
            __syncthreads();
            {   // exclusive scan of the cell counters
                const int v = tid < kCells ? cnt[tid] : 0;
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
                if (tid < kCells) start[tid] = base + incl - v;
                if (tid == 0) start[kCells] = ne;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < EMAX / NT; ++i) {
                const int e = tid + i * NT;
                if (e < ne) order[start[cellid[e]] + rk[i]] = (unsigned short)(e | (rk[i] == 0 ? 0x8000 : 0));
            }
            __syncthreads();
            lap(1);                  // 1: counting sort
            // ---- accumulate.  64 groups of 8 lanes (lane j = channels 4j..4j+3) each take an equal share of the
            //      colour's sorted entries -- snapped to cell boundaries: a group owns the cells whose FIRST entry lies in
            //      its share -- and stream through them eight at a time (8 entry reads, then 8 independent grad_out row
            //      loads in flight, then the short dependent chain), keeping the four corner sums of the current cell in
            //      registers and adding them to the LDS tile when the cell ends.  Cells of one colour have disjoint
            //      footprints, so the read-modify-write needs no atomics; a barrier separates the colours.
            {
                constexpr int G = NT / 8;
                const int grp = tid >> 3, j = tid & 7;
                const float4 *gb4 = reinterpret_cast<const float4 *>(gout + ((int64_t)n * Lq * M + m) * kD) + j;
                const int rs4 = rs / 4;
                float4 *acc4 = reinterpret_cast<float4 *>(acc) + j;
#pragma unroll 1
                for (int col = 0; col < 4; ++col) {
                    const int e0 = start[col * CC], e1 = start[(col + 1) * CC];
                    const int len = e1 - e0;
                    int i = e0 + (int)((int64_t)len * grp / G);
                    const int hi = e0 + (int)((int64_t)len * (grp + 1) / G);
                    while (i < e1 && !(order[i] & 0x8000)) ++i;      // tail of a cell that belongs to the previous share
                    float4 a00 = make_float4(0.f, 0.f, 0.f, 0.f), a01 = a00, a10 = a00, a11 = a00;
                    int cur = -1;
                    auto deposit = [&]() {
                        if (cur < 0) return;
                        const int idx = cur - col * CC;
                        const int cy = 2 * (idx / NXC) + (col >> 1) - 1, cx = 2 * (idx % NXC) + (col & 1) - 1;
                        const bool y0 = (unsigned)cy < (unsigned)TH, y1 = (unsigned)(cy + 1) < (unsigned)TH;
                        const bool x0 = (unsigned)cx < (unsigned)TW, x1 = (unsigned)(cx + 1) < (unsigned)TW;
                        const int r00 = cy * TW + cx;
                        auto rmw = [&](int r, const float4 &a) {
                            float4 v = acc4[r * 8];
                            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
                            acc4[r * 8] = v;
                            if (j == 0) touched[r] = 1;
                        };
                        if (y0 && x0) rmw(r00, a00);
                        if (y0 && x1) rmw(r00 + 1, a01);
                        if (y1 && x0) rmw(r00 + TW, a10);
                        if (y1 && x1) rmw(r00 + TW + 1, a11);
                        a00 = a01 = a10 = a11 = make_float4(0.f, 0.f, 0.f, 0.f);
                    };
                    bool done = i >= hi;
                    while (!done && i < e1) {
                        unsigned short o[8];
                        float4 g[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) o[u] = order[min(i + u, e1 - 1)];
#pragma unroll
                        for (int u = 0; u < 8; ++u)      // only the query index is needed before the loads are issued
                            g[u] = DBG == 3 ? make_float4(1.f, 1.f, 1.f, 1.f)       // timing aid: no grad_out loads
                                            : gb4[(int64_t)__float_as_int(ent[o[u] & 0x7fff].w) * rs4];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (done || i + u >= e1) { done = true; continue; }
                            if (o[u] & 0x8000) {
                                if (i + u >= hi) { done = true; continue; }
                                deposit();
                                cur = cellid[o[u] & 0x7fff];
                            }
                            const float4 en = ent[o[u] & 0x7fff];
                            const float lw = en.x, lh = en.y, a = en.z;
                            const float hh = 1.f - lh, hwt = 1.f - lw;
                            const float w1 = hh * hwt * a, w2 = hh * lw * a, w3 = lh * hwt * a, w4 = lh * lw * a;
                            a00.x += w1 * g[u].x; a00.y += w1 * g[u].y; a00.z += w1 * g[u].z; a00.w += w1 * g[u].w;
                            a01.x += w2 * g[u].x; a01.y += w2 * g[u].y; a01.z += w2 * g[u].z; a01.w += w2 * g[u].w;
                            a10.x += w3 * g[u].x; a10.y += w3 * g[u].y; a10.z += w3 * g[u].z; a10.w += w3 * g[u].w;
                            a11.x += w4 * g[u].x; a11.y += w4 * g[u].y; a11.z += w4 * g[u].z; a11.w += w4 * g[u].w;
                        }
                        i += 8;
                    }
                    deposit();
                    __syncthreads();
                }
            }
            lap(2);                  // 2: walk + read-modify-write, four colours
            // ---- out-of-reach corners: one full-line atomic each, exactly like the plain kernel
            const int nmiss = min(ctl[3], MISSMAX);
            for (int mi = hw; mi < nmiss; mi += NT / 32) {
                const float4 ms = miss[mi];
                const float g = gb[(int64_t)__float_as_int(ms.y) * rs];
                fp_atomic_add(gvb + (int64_t)__float_as_int(ms.z) * rs, ms.x * g);
            }
            __syncthreads();
            if (DBG >= 2 && tid == 0) atomicAdd(&g_dest_dbg[10], (unsigned long long)nmiss);
            if (tid == 0) ctl[3] = 0;
            __syncthreads();
            lap(3);                  // 3: misses
        };

        // The list position of a kept sample = samples listed before this batch (`ne`, a register every thread keeps in
        // step) + a per-batch LDS counter.  Three counters rotate so that the one batch i+2 will use can be zeroed
        // after the barrier of batch i without racing the threads still reading batch i's count.
        int ne = 0;
        for (int base = c_lo; base < c_hi;) {
            int *ctr = &ctl[bi % 3];
#pragma unroll
            for (int it = 0; it < BATCH / NT; ++it) {
                const int idx = base + it * NT + tid;
                bool push = false;
                float4 en = make_float4(0.f, 0.f, 0.f, 0.f);
                int cell = 0;
                if (idx < c_hi) {
                    int lq = 0;
                    while (idx >= pre[lq + 1]) ++lq;
                    const int rem = idx - pre[lq];
                    const int sq = (int)(((float)rem + 0.5f) * inv_p), p = rem - sq * P;        // exact: rem < 2^22
                    const int rw = rect[lq][3];
                    const int qyr = (int)(((float)sq + 0.5f) * (1.0f / (float)rw));
                    const int qy = rect[lq][0] + qyr, qx = rect[lq][1] + sq - qyr * rw;
                    const int Hq = lv_h[lq], Wq = lv_w[lq];
                    // the query's own pixel centre on the tile's level (integer pixel)
                    const int ry = ((2 * qy + 1) * H) / (2 * Hq), rx = ((2 * qx + 1) * W) / (2 * Wq);
                    if (ry >= lo_y && ry <= hi_y && rx >= lo_x && rx <= hi_x) {
                        const int q = lv_st[lq] + qy * Wq + qx;
                        const int64_t nq = (int64_t)n * Lq + q, row = nq * M + m;
                        const int k = l * P + p;
                        float x, y;
                        io.load_xy(row, nq, LP, k, l, P, H, W, x, y);
                        const float h = sub_rn(mul_rn(y, (float)H), 0.5f), w = sub_rn(mul_rn(x, (float)W), 0.5f);
                        if (h > -1.f && w > -1.f && h < (float)H && w < (float)W) {
                            const int h0 = (int)floorf(h), w0 = (int)floorf(w);
                            const int cy = h0 - ty0, cx = w0 - tx0;
                            const bool iy0 = (unsigned)cy < (unsigned)THc, iy1 = (unsigned)(cy + 1) < (unsigned)THc;
                            const bool ix0 = (unsigned)cx < (unsigned)TWc, ix1 = (unsigned)(cx + 1) < (unsigned)TWc;
                            const bool mine = (iy0 || iy1) && (ix0 || ix1);
                            const bool home = (ry / TH == ty) && (rx / TW == tx);
                            if (mine || home) {
                                float a = io.load_w(row, LP, k);
                                if (IO::kSoftmax) {
                                    float mx = a;
                                    for (int j = 0; j < LP; ++j) mx = fmaxf(mx, io.load_w(row, LP, j));
                                    float sum = 0.f;
                                    for (int j = 0; j < LP; ++j) sum += expf(io.load_w(row, LP, j) - mx);
                                    a = expf(a - mx) / sum;
                                }
                                const float lh = sub_rn(h, (float)h0), lw = sub_rn(w, (float)w0);
                                if (mine) {
                                    push = true;
                                    en = make_float4(lw, lh, a, __int_as_float(q));
                                    cell = (((cy + 1) & 1) * 2 + ((cx + 1) & 1)) * CC + ((cy + 1) >> 1) * NXC + ((cx + 1) >> 1);
                                }
                                if (home) {      // corners nobody else will pick up
                                    const float hh = 1.f - lh, hwt = 1.f - lw;
                                    const float cwv[4] = {hh * hwt * a, hh * lw * a, lh * hwt * a, lh * lw * a};
#pragma unroll
                                    for (int ci = 0; ci < 4; ++ci) {
                                        const int yy = h0 + (ci >> 1), xx = w0 + (ci & 1);
                                        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;      // zero padding
                                        const int oy = (yy / TH) * TH, ox = (xx / TW) * TW;         // owner tile origin
                                        if (oy == ty0 && ox == tx0) continue;                        // in my tile
                                        if (ry >= oy - R && ry <= oy + TH - 1 + R && rx >= ox - R && rx <= ox + TW - 1 + R)
                                            continue;                                                // its owner sees q
                                        const int pix = st + yy * W + xx;
                                        const int mi = atomicAdd(&ctl[3], 1);
                                        if (mi < MISSMAX) {
                                            miss[mi] = make_float4(cwv[ci], __int_as_float(q), __int_as_float(pix), 0.f);
                                        } else {     // list full (pathological sampling pattern): scatter it right here
                                            const float *gq = gout + ((int64_t)nq * M + m) * kD;
                                            float *dst = gvalue + (((int64_t)n * S + pix) * M + m) * kD;
                                            for (int ch = 0; ch < kD; ++ch) fp_atomic_add(dst + ch, cwv[ci] * gq[ch]);
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                // wave-aggregated append to the sample list
                const unsigned long long bal = __ballot(push);
                const int nb = __popcll(bal);
                int wbase = 0;
                if (lane == 0 && nb) wbase = ne + atomicAdd(ctr, nb);
                wbase = __shfl(wbase, 0, 64);
                if (push) {
                    const int pos = wbase + __popcll(bal & ((1ull << lane) - 1ull));
                    ent[pos] = en;
                    cellid[pos] = (unsigned short)cell;
                }
            }
            base += BATCH;
            __syncthreads();
            ne += *ctr;
            if (tid == 0) ctl[(bi + 2) % 3] = 0;
            ++bi;
            if (base >= c_hi || ne + BATCH > EMAX) {
                round(ne);
                ne = 0;
            }
        }

        lap(4);                      // 4: tail of the unit loop (nothing)
        // ---- flush: every touched row of the tile once
        for (int r = hw; r < kRows; r += NT / 32) {
            const int y = r / TW, x = r % TW;
            if (y < THc && x < TWc && touched[r])
                fp_atomic_add(gvb + (int64_t)(st + (ty0 + y) * W + tx0 + x) * rs, acc[r * kD + c]);
        }
        lap(5);                      // 5: flush
        if (DBG >= 2 && tid == 0) atomicAdd(&g_dest_dbg[11], 1ull);     // units
    }
}
