// Encoder self-attention forward with the corner rows served from an LDS WINDOW (fp32, D == 32, num_point == 4, L * P == 16).
// Included by msda.hip after msda_fast.h.  EXPERIMENT (forward variant 600), NOT the default: third attempt at this idea
// (DESIGN.md 2.1 has all three).  Sized by a probe first -- tools/lds_window_probe.hip: the same synthetic corner traffic takes
// 200 us through the vector-memory path and 137 us through a 324-row window with three workgroups per CU -- and built for
// occupancy: a 256-thread workgroup, <= 324 window rows (41.5 KB) + 8 KB of records = three workgroups per CU, 84 VGPRs.
// Measured at the encoder shape, bs 4 (profiles/r02_lds_window_forward.txt): 375 / 459 / 685 us at a sample spread of 1 / 2 / 4
// pixels against 238 / 244 / 257 us for msda_fwd_d32<1, 4, 408>: per level a dependent chain (location load -> sample geometry ->
// record -> barrier -> record read -> corner reads -> barrier) with 12 waves per CU to hide it, and a fall-back path whose
// cost grows with the spread.
//
// A workgroup takes an 8 x 8 patch of query pixels of one level and one head.  Per sampling level it
//   * places a window where the patch maps to on that level, +-kLwMargin pixels, and copies those value rows (128 B each;
//     rows outside the level are zeros: the buffer bounds check does the zero padding) into LDS;
//   * writes one record per (query, point): either {LDS byte offset of the top-left corner} when the sample's 2 x 2
//     footprint lies inside the window, or the four global byte offsets the plain kernel uses (any sampling pattern is
//     correct; locality only decides speed).  A window that would not fit (coarse query level sampling a fine level) turns
//     the whole level into the plain path;
//   * lets each 8-lane group (lane = 4 channels) accumulate two queries' four samples: 4 x ds_read_b128 per sample, or
//     4 x buffer_load_dwordx4 for the direct ones.
#pragma once

constexpr int kLwQ = 64, kLwRows = 324, kLwMargin = 5;
constexpr unsigned kLwMark = 0xFFFFFFFEu;       // never a byte offset (those are multiples of 16) and not kOob

constexpr size_t lw_lds_bytes() { return (size_t)kLwRows * 128 + (size_t)256 * 16 * 2; }

template <typename IO>
__global__ __launch_bounds__(256, 3) void msda_fwd_d32_lw(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts, const IO io,
    int S, int M, int L, int tiles_per_image, float *__restrict__ out)
{
    static_assert(!IO::kSoftmax, "LocAttnIO only (the fused prologue keeps the plain kernel)");
    constexpr int P = kPT;
    extern __shared__ float4 smem[];
    float4 *win = smem;                                                   // [kLwRows][8]
    int4 *rec_o = reinterpret_cast<int4 *>(smem + kLwRows * 8);           // [256] per (query slot, point)
    float4 *rec_w = reinterpret_cast<float4 *>(rec_o + 256);              // [256] corner weights x attention

    const Tile t = tile_of_block(M, tiles_per_image, kLwQ);
    const int rs = M * kD, LP = L * P, Lq = S;
    const int tid = threadIdx.x, g = tid >> 3, j = tid & 7;
    const __amdgpu_buffer_rsrc_t vr = image_rsrc(value + (int64_t)t.n * S * M * kD, (unsigned)S * M * kD * 4u);
    const unsigned lane_b = (unsigned)(t.m * kD + 4 * j) * 4u;
    const unsigned head_b = (unsigned)(t.m * kD) * 4u;

    for (int tile = t.q0 / kLwQ;; tile += tiles_per_image) {
        const Patch pt = find_patch<8, 8>(tile, shapes, starts, L);
        if (pt.Hq == 0) return;
        const int rq = tid >> 2, p = tid & 3;                  // this thread's record: query slot, point
        const int qrec = patch_query<8>(pt, rq);
        const int y1 = min(pt.y0 + 7, pt.Hq - 1), x1 = min(pt.x0 + 7, pt.Wq - 1);
        float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)starts[l];
            // window rectangle on this level (a heuristic: whatever it misses takes the direct path)
            const float sy = (float)H / (float)pt.Hq, sx = (float)W / (float)pt.Wq;
            const int wy_lo = (int)floorf((pt.y0 + 0.5f) * sy - 0.5f) - kLwMargin;
            const int wy_hi = (int)floorf((y1 + 0.5f) * sy - 0.5f) + kLwMargin;       // rows h0 and h0 + 1 for |offset| < margin
            const int wx_lo = (int)floorf((pt.x0 + 0.5f) * sx - 0.5f) - kLwMargin;
            const int wx_hi = (int)floorf((x1 + 0.5f) * sx - 0.5f) + kLwMargin;
            const int wh = wy_hi - wy_lo + 1, ww = wx_hi - wx_lo + 1;
            const bool use_win = wh * ww <= kLwRows && wh <= 18 && ww <= 32;
            __syncthreads();                                   // previous level / patch done with window and records
            // ---- every global load of the level is issued before anything waits: the window lines (row slot s = tid / 8 is
            //      the window column, ww <= 32; one load per line and thread, lines beyond the window read out of range) and the
            //      sample's location / weight.  As three rounds of loads + a dependent record load this was four serial global
            //      round trips per level with only 12 waves per CU to hide them: 720-800 us.
            float x = 0.f, y = 0.f, a = 0.f;
            if (qrec >= 0) {
                const int k = l * P + p;
                const int64_t nq = (int64_t)t.n * Lq + qrec, row = nq * M + t.m;
                io.load_xy(row, nq, LP, k, l, P, H, W, x, y);
                a = io.load_w(row, LP, k);
            }
            // the window goes global -> LDS directly (buffer_load ... lds: no staging registers, no ds_write pass; the LDS
            // destination is wave-uniform base + lane x 16 B, which is exactly this layout: a wave holds 8 columns x 8 pieces
            // of one window line = 1 KB contiguous).  Columns beyond the window are masked out, rows outside the level read
            // as zeros through the buffer bounds check.
            if (use_win) {
                const int sx_ = tid >> 3, px = wx_lo + sx_;
                const bool col_ok = (unsigned)px < (unsigned)W;
                const unsigned step = (unsigned)W * (unsigned)rs * 4u;
                unsigned goff = (unsigned)(st + wy_lo * W + px) * (unsigned)rs * 4u + lane_b;      // wraps while py < 0: unused then
                typedef __attribute__((address_space(3))) void lds_void;
                char *line = reinterpret_cast<char *>(win) + (tid >> 6) * 1024;
                if (sx_ < ww) {
                    for (int u = 0; u < wh; ++u) {
                        const bool ok = col_ok && (unsigned)(wy_lo + u) < (unsigned)H;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(vr, (lds_void *)line, 16, ok ? goff : kOob, 0, 0, 0);
                        goff += step;
                        line += ww * 128;
                    }
                }
            }
            {   // this thread's record
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                int4 ro = make_int4((int)kOob, (int)kOob, (int)kOob, (int)kOob);
                if (qrec >= 0) {
                    const float hf = sub_rn(mul_rn(y, (float)H), 0.5f), wf = sub_rn(mul_rn(x, (float)W), 0.5f);
                    if (hf > -1.f && wf > -1.f && hf < (float)H && wf < (float)W) {       // the op's range test (.cuh:288)
                        const float h0f = floorf(hf), w0f = floorf(wf);
                        const int h0 = (int)h0f, w0 = (int)w0f;
                        const float lh = sub_rn(hf, h0f), lw = sub_rn(wf, w0f);
                        const float hh = 1.f - lh, hw = 1.f - lw;
                        w = make_float4(a * (hh * hw), a * (hh * lw), a * (lh * hw), a * (lh * lw));
                        if (use_win && h0 >= wy_lo && h0 < wy_hi && w0 >= wx_lo && w0 < wx_hi) {
                            ro = make_int4(((h0 - wy_lo) * ww + (w0 - wx_lo)) * 128, (int)kLwMark, 0, 0);
                        } else {                                // rare: the plain kernel's four global offsets
                            unsigned off[4];
                            float lw2, lh2;
                            sample_setup_oob(x, y, H, W, st, (unsigned)rs * 4u, off, lw2, lh2);
                            ro = make_int4((int)(off[0] + head_b), (int)(off[1] + head_b), (int)(off[2] + head_b), (int)(off[3] + head_b));
                        }
                    }
                }
                rec_o[tid] = ro;
                rec_w[tid] = w;
            }
            __syncthreads();
            const char *wb = reinterpret_cast<const char *>(win) + j * 16;
            const int pitch = ww * 128;
            auto consume = [&](const int slot, const int p0, float4 &ac) {      // samples p0, p0 + 1 of query `slot`
                int4 ro[2];
                float4 rw[2], v[2][4];
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    ro[pp] = rec_o[slot * 4 + p0 + pp];
                    rw[pp] = rec_w[slot * 4 + p0 + pp];
                }
                const bool d0 = (unsigned)ro[0].y != kLwMark, d1 = (unsigned)ro[1].y != kLwMark;
                if (!__any(d0 || d1)) {               // the common case, wave-uniform: LDS only
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const char *b0 = wb + ro[pp].x;
                        v[pp][0] = *reinterpret_cast<const float4 *>(b0);
                        v[pp][1] = *reinterpret_cast<const float4 *>(b0 + 128);
                        v[pp][2] = *reinterpret_cast<const float4 *>(b0 + pitch);
                        v[pp][3] = *reinterpret_cast<const float4 *>(b0 + pitch + 128);
                    }
                } else {                              // some lane group has a sample outside the window: one corner at a
                                                      // time (rare path; kept small in registers)
#pragma unroll 1
                    for (int pp = 0; pp < 2; ++pp) {
                        const int4 r = pp == 0 ? ro[0] : ro[1];
                        const bool lds = (unsigned)r.y == kLwMark;
#pragma unroll 1
                        for (int c = 0; c < 4; ++c) {
                            const int goff = c == 0 ? r.x : (c == 1 ? r.y : (c == 2 ? r.z : r.w));
                            const int loff = r.x + (c & 1) * 128 + (c >> 1) * pitch;
                            float4 vv;
                            if (lds) vv = *reinterpret_cast<const float4 *>(wb + loff);
                            else vv = buf_ld4(vr, (unsigned)goff + 16u * j);
                            const float4 wq = pp == 0 ? rw[0] : rw[1];
                            const float wc = c == 0 ? wq.x : (c == 1 ? wq.y : (c == 2 ? wq.z : wq.w));
                            ac.x += wc * vv.x; ac.y += wc * vv.y; ac.z += wc * vv.z; ac.w += wc * vv.w;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    return;
                }
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const float4 w = rw[pp];
                    ac.x += w.x * v[pp][0].x + w.y * v[pp][1].x + w.z * v[pp][2].x + w.w * v[pp][3].x;
                    ac.y += w.x * v[pp][0].y + w.y * v[pp][1].y + w.z * v[pp][2].y + w.w * v[pp][3].y;
                    ac.z += w.x * v[pp][0].z + w.y * v[pp][1].z + w.z * v[pp][2].z + w.w * v[pp][3].z;
                    ac.w += w.x * v[pp][0].w + w.y * v[pp][1].w + w.z * v[pp][2].w + w.w * v[pp][3].w;
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            __builtin_amdgcn_sched_barrier(0);      // keep the window copy's registers out of the consume phase ...
            consume(g, 0, acc[0]);
            consume(g, 2, acc[0]);
            consume(32 + g, 0, acc[1]);
            consume(32 + g, 2, acc[1]);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int q = patch_query<8>(pt, half * 32 + g);
            if (q >= 0) *reinterpret_cast<float4 *>(out + (((int64_t)t.n * Lq + q) * M + t.m) * kD + 4 * j) = acc[half];
        }
    }
}
