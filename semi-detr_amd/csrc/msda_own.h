// EXPERIMENT (backward variant 910), compiled only with SEMIDETR_EXPERIMENTS: grad_value for arbitrary query sets by
// owner-computes tiles -- no zero fill, no float atomics.  Parity-green, SLOWER than the fill + level-aggregated atomics it
// was meant to replace (micro-benchmark shape 37.8 us against 36.3; decoder bs 4 / Lq 1100: 236 us against 162): with the
// zero fill and the atomics gone the launch is a latency chain per workgroup -- loads, count, scan, zero rows, fill, walk
// with grad_out rows fetched from L2 -- of ~12 us at two resident workgroups per CU (instrumented: count + scan 6.7 us, zero
// rows 3.2, fill 7.5, walk 11.6 for 1088 tiles), and 9 samples per thread (Lq * P = 4400) do not fit 128 VGPRs without
// spilling.  See DESIGN.md section 7, round 3.
#pragma once

// ---------------------------------------------------------------------------------------------
// grad_value for arbitrary query sets by OWNER-COMPUTES: no float atomics, no zero fill in front of the launch.
//
// The level-aggregated scatter above still pays a hipMemsetAsync (8 us for the 45 MB of the micro-benchmark shape) and
// then 174 k full-row atomics that all start at the same moment, once every workgroup has sorted its samples (28 us, of
// which the atomic unit alone needs 17).  A query set this small can be turned around: a workgroup OWNS a tile of rows of
// one (image, head, level) -- T consecutive pixels, T = the level's size / 16 rounded to 32, between 32 and 512 -- and
// scans ALL Lq * P samples the (image, head) throws at that level (<= kOwnCap, 9 per thread at most: that is what bounds
// the path), keeps the (sample, corner) pairs landing in its tile, buckets them by row with integer LDS atomics (count ->
// scan -> fill, as everywhere in this file), and lets 32 streams of 16 lanes walk the row-sorted entries with the row sum
// in registers.  Every row belongs to exactly one workgroup and, inside it, to exactly one stream (the streams' shares are
// moved to row boundaries), so a finished row is STORED; rows nobody sampled are stored as zeros: the result is complete
// without a fill, and summation order inside a row is the deterministic rank order of the count phase.  grad_out rows are
// read where they are (L2; 16 lanes x float2 = one 128-byte row per entry, 8 in flight per stream).
// If the tile's entries exceed the LDS list (kOwnCap) the rows are processed in sub-ranges whose entries fit, each with its
// own fill + walk -- one row never exceeds the list because a sample touches a row at most once and Lq * P <= kOwnCap.
// ---------------------------------------------------------------------------------------------
constexpr int kOwnThreads = 512;
constexpr int kOwnRows = 512;            // rows one workgroup owns at most (== kOwnThreads: one counter per thread in the scan)
constexpr int kOwnCap = 4608;            // entries in LDS at a time = the largest Lq * P the path takes

__host__ __device__ inline int own_tile_rows(int R)
{
    const int t = ((R + 15) / 16 + 31) & ~31;
    return t < 32 ? 32 : (t > kOwnRows ? kOwnRows : t);
}
// upper bound of the tiles of one (image, head): <= 16 per level whose tile is not clamped to kOwnRows, R / kOwnRows + 1 otherwise
inline int own_tiles_bound(int S, int L) { return (S + kOwnRows - 1) / kOwnRows + 17 * L; }
constexpr size_t kOwnLdsBytes = (size_t)kOwnCap * 8 + (size_t)kOwnRows * 4 + (size_t)(kOwnRows + 1) * 4 + 12;

template <typename IO, int SPT>
__device__ __forceinline__ void own_scatter_body(
    int b, float4 *smem, const float *__restrict__ gout, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int Lq, int P, int tiles_bound,
    float *__restrict__ gvalue)
{
    constexpr int NT = kOwnThreads, kStreams = NT / 16;
    static_assert(NT == kOwnRows, "the scan gives every thread one counter");
    float2 *entries = reinterpret_cast<float2 *>(smem);
    int *cnt = reinterpret_cast<int *>(entries + kOwnCap);
    int *start = cnt + kOwnRows;                       // kOwnRows + 1 prefix sums
    __shared__ int wsum[NT / 64];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int LP = L * P, rs = M * kD;
    // block -> (head, tile slot, image): heads fastest (consecutive workgroups = the 8 heads of one pixel range)
    const int m = b % M; b /= M;
    int slot = b % tiles_bound;
    const int n = b / tiles_bound;
    int l = 0, H = 0, W = 0, T = 0;
    for (; l < L; ++l) {
        H = (int)shapes[2 * l]; W = (int)shapes[2 * l + 1];
        T = own_tile_rows(H * W);
        const int nt = (H * W + T - 1) / T;
        if (slot < nt) break;
        slot -= nt;
    }
    if (l == L) return;                                // the grid is an upper bound
    const int R = H * W, st = (int)starts[l], t0 = slot * T, Ta = min(T, R - t0);

    cnt[tid] = 0;
    // ---- this thread's samples (query i, point p), sample index tid + sp * NT: loads first, arithmetic after
    const int nsamp = Lq * P;
    float sx[SPT], sy[SPT], sa[SPT];
#pragma unroll
    for (int sp = 0; sp < SPT; ++sp) {
        const int sidx = min(tid + sp * NT, nsamp - 1);
        const int i = sidx / P, p = sidx - i * P;
        const int64_t nqi = (int64_t)n * Lq + i, row = nqi * M + m;
        const int k = l * P + p;
        io.load_xy(row, nqi, LP, k, l, P, H, W, sx[sp], sy[sp]);
        sa[sp] = io.load_w(row, LP, k);
        if (IO::kSoftmax) {
            float mx = sa[sp];
            for (int j = 0; j < LP; ++j) mx = fmaxf(mx, io.load_w(row, LP, j));
            float sum = 0.f;
            for (int j = 0; j < LP; ++j) sum += expf(io.load_w(row, LP, j) - mx);
            sa[sp] = expf(sa[sp] - mx) / sum;
        }
    }
    // tile-local rows + corner weights of sample sp (recomputed in the fill pass: 5 registers per sample stay live, not 13)
    auto corners = [&](int sp, int (&crow)[4], float (&cw)[4]) {
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) { crow[ci] = -1; cw[ci] = 0.f; }
        int off[4];
        float lw, lh;
        if (tid + sp * NT >= nsamp || !sample_setup(sx[sp], sy[sp], H, W, 0, 1, off, lw, lh)) return;   // off = level-local pixel or -1
        const float a = sa[sp];
        const float hh = 1.f - lh, hwt = 1.f - lw;
        cw[0] = hh * hwt * a; cw[1] = hh * lw * a; cw[2] = lh * hwt * a; cw[3] = lh * lw * a;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            const int r = off[ci] - t0;
            if (off[ci] >= 0 && (unsigned)r < (unsigned)Ta && !(io.has_mask() && io.masked(n, st + off[ci])))
                crow[ci] = r;                          // padded pixels receive no gradient (their rows are stored as zeros)
        }
    };
    unsigned rank01[SPT], rank23[SPT];
    __syncthreads();                                   // counters zeroed
#pragma unroll
    for (int sp = 0; sp < SPT; ++sp) {
        int crow[4];
        float cw[4];
        corners(sp, crow, cw);
        unsigned rk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
            if (crow[ci] >= 0) rk[ci] = (unsigned)atomicAdd(&cnt[crow[ci]], 1);
        rank01[sp] = rk[0] | (rk[1] << 16);
        rank23[sp] = rk[2] | (rk[3] << 16);
        if (SPT > 3) __builtin_amdgcn_sched_barrier(0);        // keep the samples' temporaries from piling up (spills at SPT = 9)
    }
    __syncthreads();
    {   // exclusive scan of the kOwnRows counters (rows past Ta stay 0)
        const int local = cnt[tid];
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
        start[tid] = base + incl - local;
        if (tid == NT - 1) start[NT] = base + incl;
    }
    __syncthreads();
    float *gvt = gvalue + (((int64_t)n * S + st + t0) * M + m) * kD;       // row r of the tile: + r * rs
    // rows nobody sampled: zeros (8 lanes x float4 per row)
    for (int r = tid >> 3; r < Ta; r += NT / 8)
        if (cnt[r] == 0) *reinterpret_cast<float4 *>(gvt + (int64_t)r * rs + 4 * (tid & 7)) = make_float4(0.f, 0.f, 0.f, 0.f);

    const int sid = tid >> 4, l16 = tid & 15;
    const float2 *g2 = reinterpret_cast<const float2 *>(gout + ((int64_t)n * Lq * M + m) * kD) + l16;   // query q: + q * (rs / 2)
    float2 *dst = reinterpret_cast<float2 *>(gvt) + l16;
    const int rs2 = rs / 2;
    int ra = 0;
    while (ra < Ta) {                                  // uniform: sub-ranges of rows whose entries fit the list (normally one)
        const int ebase = start[ra];
        int rb = Ta;
        if (start[Ta] - ebase > kOwnCap) {
            int lo = ra + 1, hi = Ta;                  // a single row always fits (a sample touches a row at most once)
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (start[mid] - ebase <= kOwnCap) lo = mid; else hi = mid - 1;
            }
            rb = lo;
        }
#pragma unroll
        for (int sp = 0; sp < SPT; ++sp) {
            const int q = (tid + sp * NT) / P;
            int crow[4];
            float cw[4];
            corners(sp, crow, cw);
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                const int r = crow[ci];
                if (r >= ra && r < rb) {
                    const int rk = (int)(((ci < 2 ? rank01[sp] : rank23[sp]) >> (16 * (ci & 1))) & 0xffffu);
                    entries[start[r] - ebase + rk] =
                        make_float2(cw[ci], __int_as_float((rk == cnt[r] - 1 ? (int)0x80000000 : 0) | (r << 16) | q));
                }
            }
            if (SPT > 3) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        {   // walk: equal shares moved forward to the next row boundary -> every row is summed and stored by one stream
            const int total = start[rb] - ebase;
            auto aligned = [&](int x) {
                if (x <= 0) return 0;
                while (x < total && __float_as_int(entries[x - 1].y) >= 0) ++x;
                return min(x, total);
            };
            const int lo = aligned((int)((int64_t)total * sid / kStreams));
            const int hi = aligned((int)((int64_t)total * (sid + 1) / kStreams));
            float2 acc = make_float2(0.f, 0.f);
            for (int e = lo; e < hi; e += 8) {
                float2 en[8], gq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) en[u] = entries[min(e + u, hi - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) gq[u] = g2[(int64_t)(__float_as_int(en[u].y) & 0xffff) * rs2];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (e + u >= hi) break;
                    const int pk = __float_as_int(en[u].y);
                    acc.x += en[u].x * gq[u].x;
                    acc.y += en[u].x * gq[u].y;
                    if (pk < 0) {
                        dst[(int64_t)((pk >> 16) & 0x7fff) * rs2] = acc;
                        acc = make_float2(0.f, 0.f);
                    }
                }
            }
        }
        ra = rb;
        if (ra < Ta) __syncthreads();                  // the list is refilled
    }
}

// ONE launch: owner-computes grad_value workgroups [0, scatter_blocks) + gather workgroups (two 256-thread gather blocks per
// 512-thread workgroup) for grad_sampling_loc / grad_attn_weight, as in msda_bwd_lvl_merged -- and nothing in front of it.
template <typename IO, int KLP, int SPT>
__global__ __launch_bounds__(kOwnThreads, 4) void msda_bwd_own_merged(
    const float *__restrict__ gout, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ starts, const IO io, int S, int M, int L, int Lq, int P, int tiles_bound,
    int scatter_blocks, int gather_tiles, int gather_blocks, float *__restrict__ gvalue)
{
    extern __shared__ float4 smem[];
    const int gwgs = (gather_blocks + 1) / 2;          // gather workgroups FIRST: theirs is the longer dependency chain
    if ((int)blockIdx.x >= gwgs) {
        own_scatter_body<IO, SPT>((int)blockIdx.x - gwgs, smem, gout, shapes, starts, io, S, M, L, Lq, P, tiles_bound, gvalue);
        return;
    }
    const int half = (int)threadIdx.x >> 8;
    const int vb = 2 * (int)blockIdx.x + half;
    const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;       // LDS of one gather block, in float4
    gather_body<IO, KLP, 0>(vb, (int)threadIdx.x & 255, smem + half * half_f4, vb < gather_blocks, gout, value, shapes,
                            starts, io, S, M, L, Lq, P, gather_tiles);
}

