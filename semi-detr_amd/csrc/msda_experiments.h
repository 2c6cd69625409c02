// TEST / TUNING dispatch of the fp32 / D == 32 MSDA kernels: every kernel variant that was ever measured against the
// defaults, selectable through semidetr_msda_set_variant (codes: DESIGN.md 2.3b).  Compiled ONLY into
// libsemidetr_hip_exp.so (-DSEMIDETR_EXPERIMENTS=1); the product library has neither these kernels nor the variant state.
// Included by msda.hip inside its anonymous namespace.
#pragma once      // msda.hip includes <atomic> at file scope (this header sits inside its anonymous namespace)

std::atomic<int> g_fwd_variant_a{0}, g_bwd_variant_a{0};
// the launchers below read the variant codes once per call
#define g_fwd_variant (g_fwd_variant_a.load(std::memory_order_relaxed))
#define g_bwd_variant (g_bwd_variant_a.load(std::memory_order_relaxed))

// ---- region-window kernels (msda_rw.h): one launcher for the forward and the gather -------------------------------
template <typename IO, int NT, int RTH, int RTW, int H0, int HC, int KL, bool GATHER, int DBG = 0, int TUNE = 42>
int launch_rw(hipStream_t st, const float *grad_out, const float *value, const int64_t *spatial_shapes,
              const int64_t *level_start, const IO &io, int N, int S, int M, float *out, float4 *zero, int64_t zero_n4)
{
    auto kern = &msda_rw_d32<IO, NT, RTH, RTW, H0, HC, KL, GATHER, DBG, TUNE>;
    if (int rc = allow_big_lds(kern, rw_lds_bytes<NT, RTH, RTW, H0, HC, KL, TUNE>(), "msda region-window kernel")) return rc;
    // grid sizing hint: the finest level of a DETR pyramid holds ~3/4 of the pixels; a workgroup takes regions slot,
    // slot + bound, ... so any bound >= 1 is correct (the level table lives in device memory)
    const int rpx = RTH * RTW;
    const int bound = ((S * 3 / 4 + rpx - 1) / rpx) * 9 / 8 + 2 * KL;
    const int64_t grid = (int64_t)N * bound * M;
    SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda: grid too large");
    constexpr size_t lds = rw_lds_bytes<NT, RTH, RTW, H0, HC, KL, TUNE>();
    static_assert(lds <= 160 * 1024, "region-window configuration does not fit the LDS");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, st, grad_out, value, spatial_shapes, level_start, io, S, M,
                       bound, out, zero, zero_n4, FwdStats{nullptr, 0u}, 0);
    return semidetr::launch_status(GATHER ? "msda_rw_d32<gather>" : "msda_rw_d32<forward>");
}

// variant code -> configuration {threads, region, margins}; 0 = the default configuration
template <typename IO, int KL, bool GATHER>
int launch_rw_cfg(int cfg, hipStream_t st, const float *grad_out, const float *value, const int64_t *spatial_shapes,
                  const int64_t *level_start, const IO &io, int N, int S, int M, float *out, float4 *zero, int64_t zero_n4)
{
#define RW(NT_, RH_, RW_, H0_, HC_) RWT(NT_, RH_, RW_, H0_, HC_, 0, 42)
#define RWT(NT_, RH_, RW_, H0_, HC_, DBG_, TUNE_) \
    launch_rw<IO, NT_, RH_, RW_, H0_, HC_, KL, GATHER, DBG_, TUNE_>(st, grad_out, value, spatial_shapes, level_start, io, N, S, M, out, zero, zero_n4)
    if constexpr (KL == 4) {
        switch (cfg) {
        case 0: return RWT(512, 8, 16, 4, 5, 0, 40);
        case 1: if constexpr (!GATHER) return RWT(256, 16, 16, -1, 4, 0, 40); else return RWT(512, 8, 16, 4, 5, 0, 20);      // gather: a sample per scheduling barrier
        case 2: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 4, 0, 40); else return RWT(512, 8, 16, 4, 4, 0, 20);
        case 3: if constexpr (!GATHER) return RWT(256, 8, 16, -1, 3, 0, 40); else return RWT(512, 8, 16, 4, 5, 0, 60);
        case 4: if constexpr (!GATHER) return RWT(256, 8, 8, -1, 4, 0, 40); else return RWT(768, 24, 16, -1, 5, 0, 1920);      // gather: the forward's product shape (level 0 through global loads)
        case 5: if constexpr (!GATHER) return RWT(256, 8, 16, -1, 4, 0, 40); else return RWT(768, 24, 16, -1, 5, 0, 2720);      // level 0 through global loads; gather: everything thread-derived rebuilt
        case 6: if constexpr (!GATHER) return RWT(512, 8, 16, -1, 5, 0, 40); else return RWT(768, 16, 16, -1, 6, 0, 1920);
        case 15: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 5, 0, 40); else break;      // the product's shape with margin 5 (134 KB of LDS)
        case 16: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 6, 0, 40); else break;      // margin 6: 159 KB of LDS
        case 17: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 5, 1, 40); else break;      // ... instrumented
        case 18: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 6, 0, 20); else break;      // round 4's first product configuration (512 threads)
        case 19: if constexpr (!GATHER) return RWT(256, 16, 16, -1, 6, 0, 20); else break;      // ... with 256-thread workgroups
        // round 4, second look: does the kernel want more waves?  1024-thread workgroups = 16 waves per CU (128 VGPRs); their octet
        // records take 66 KB, so the margin drops to 4 -- compared at equal margin (730 = 512 threads)
        // TUNE + 100: one level-0 sample in flight instead of two; + 200: lean registers (level constants re-selected where they
        // are used, staging coordinates rebuilt per region) -- 160 instead of 256 VGPRs, which is what lets a 768-thread workgroup run
        case 30: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 4, 0, 20); else break;
        case 31: if constexpr (!GATHER) return RWT(1024, 16, 16, -1, 5, 0, 1120); else break;    // + 800: everything thread-derived rebuilt per round / region: 125 VGPRs
        case 40: if constexpr (!GATHER) return RWT(768, 16, 16, -1, 6, 0, 1120); else break;
        case 41: if constexpr (!GATHER) return RWT(1024, 16, 16, -1, 5, 0, 1110); else break;
        case 42: if constexpr (!GATHER) return RWT(768, 16, 16, -1, 6, 0, 5120); else break;     // + 3200: window addresses by v_mad_u32_u16, FMAs with explicit op_sel
        case 43: if constexpr (!GATHER) return RWT(768, 16, 16, -1, 6, 0, 1920); else break;     // the product configuration (= 734 + 1600)
        case 44: if constexpr (!GATHER) return RWT(704, 16, 16, -1, 6, 0, 1920); else break;     // 11 waves: a region's 340 queries fill 3.86 rounds of 88
        case 45: if constexpr (!GATHER) return RWT(768, 16, 32, -1, 4, 0, 1920); else break;     // 16 x 32 regions: 680 queries = 7.1 rounds, 1.2 instead of 2.9 staged rows per query; margin 4
        case 46: if constexpr (!GATHER) return RWT(704, 16, 32, -1, 5, 0, 1920); else break;     // ... margin 5 fits beside 88 octets' records
        case 47: if constexpr (!GATHER) return RWT(768, 32, 16, -1, 4, 0, 1920); else break;
        case 48: if constexpr (!GATHER) return RWT(768, 24, 16, -1, 5, 0, 1920); else break;     // the PRODUCT configuration: 24 x 16 regions, 510 queries = 5.3 rounds, margin 5
        case 49: if constexpr (!GATHER) return RWT(704, 32, 16, -1, 5, 0, 1920); else break;
        case 51: if constexpr (!GATHER) return RWT(768, 24, 16, -1, 5, 0, 1921); else break;     // + ONE out-of-window sample per octet pre-issued before the LDS loop
        case 52: if constexpr (!GATHER) return RWT(768, 24, 16, -1, 5, 0, 2721); else break;
        case 53: if constexpr (!GATHER) return RWT(768, 24, 16, -1, 5, 0, 2722); else break;
        case 54: if constexpr (!GATHER) return RWT(768, 24, 16, -1, 5, 1, 2721); else break;     // ... instrumented
        case 55: if constexpr (!GATHER) return RWT(768, 22, 16, -1, 5, 0, 1920); else break;     // region heights by how evenly their queries fill rounds of 96: 22 rows = 4.87
        case 56: if constexpr (!GATHER) return RWT(768, 25, 16, -1, 5, 0, 1920); else break;     // 25 rows = 5.53 (and four region rows exactly on a 100-row level)
        case 57: if constexpr (!GATHER) return RWT(768, 20, 16, -1, 5, 0, 1920); else break;     // 20 rows = 4.43
        case 50: if constexpr (!GATHER) return RWT(768, 24, 16, -1, 5, 1, 1920); else break;     // the product configuration, instrumented (tools/archive/r03_rw_dbg.py 750)
        case 32: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 6, 0, 220); else break;      // the product shape, lean
        case 33: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 6, 0, 240); else break;      // ... four samples between barriers again
        case 34: if constexpr (!GATHER) return RWT(768, 16, 16, -1, 6, 0, 320); else break;      // the PRODUCT configuration: 12 waves per CU, margin 6
        case 35: if constexpr (!GATHER) return RWT(768, 16, 16, -1, 6, 0, 310); else break;
        case 36: if constexpr (!GATHER) return RWT(768, 16, 16, -1, 5, 0, 320); else break;
        case 37: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 6, 0, 320); else break;
        case 38: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 6, 0, 340); else break;
        case 39: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 6, 0, 120); else break;
        case 12: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 4, 1, 40); else break;      // the PRODUCT configuration (702), instrumented
        case 13: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 4, 2, 40); else break;      // ... windows not staged (timing aid)
        case 14: if constexpr (!GATHER) return RWT(512, 16, 16, -1, 4, 3, 40); else break;      // ... LDS loop skipped (timing aid)
        case 7: if constexpr (!GATHER) return RWT(256, 8, 16, -1, 4, 1, 40); else return RWT(512, 8, 16, 4, 5, 1, 40);
        case 8: if constexpr (!GATHER) return RWT(256, 8, 16, -1, 4, 2, 40); else return RWT(512, 8, 16, 4, 5, 2, 40);
        case 9: if constexpr (!GATHER) return RWT(256, 8, 16, -1, 4, 3, 40); else return RWT(512, 8, 16, 4, 5, 3, 40);
        default: break;
        }
    }
    if constexpr (KL == 5 && !GATHER) {      // round 4: the product's shape for five levels (level 0 global, 16 x 16 regions)
        if (cfg == 15) return RWT(512, 16, 16, -1, 4, 0, 20);
        if (cfg == 18) return RWT(512, 16, 16, -1, 5, 0, 20);
        if (cfg == 34) return RWT(768, 16, 16, -1, 4, 0, 310);      // lean registers, 12 waves per CU (margin 5 does not fit beside 96 octets' records)
        if (cfg == 35) return RWT(640, 16, 16, -1, 5, 0, 310);      // 10 waves, margin 5
        if (cfg == 36) return RWT(512, 16, 16, -1, 5, 0, 320);
        if (cfg == 40) return RWT(768, 16, 16, -1, 4, 0, 1120);
        if (cfg == 41) return RWT(1024, 16, 16, -1, 4, 0, 1110);
        if (cfg == 48) return RWT(960, 24, 16, -1, 4, 0, 1110);      // 24 x 16 regions as for four levels: 15 waves is what fits beside the windows
        if (cfg == 49) return RWT(896, 24, 16, -1, 4, 0, 1110);
    }
    if constexpr (KL == 4) return RWT(512, 8, 16, 4, 5, 0, 40);
    else return RWT(512, 8, 16, 4, 4, 0, 40);      // five levels: the margin-5 windows do not fit 160 KB
#undef RWT
#undef RW
}

// ---- fast-path launchers, shared by the reference contract (LocAttnIO) and the fused prologue (RawIO) ----
template <typename IO>
int exp_launch_fast_forward(hipStream_t st, const float *value, const int64_t *spatial_shapes,
                        const int64_t *level_start, const IO &io, int N, int S, int M, int L, int Lq, int P,
                        int flags, float *out)
{
    const bool pixels = (flags & SEMIDETR_MSDA_QUERIES_ARE_PIXELS) != 0;
    SEMIDETR_REQUIRE(!pixels || Lq == S, SEMIDETR_E_BADARG,
                     "msda_forward: SEMIDETR_MSDA_QUERIES_ARE_PIXELS needs num_query == spatial_size");
    const int split = pick_split(g_fwd_variant % 10, N, Lq, M);
    const int rpb = 32 / split;
    const int tiles = (Lq + rpb - 1) / rpb;
    const int64_t grid = (int64_t)N * tiles * M;
    SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_forward: grid too large");
    const size_t lds = (size_t)rpb * (L * P + 1) * 32;
#define LAUNCH_FWD(SP, UN, PT, TILES)                                                                       \
    hipLaunchKernelGGL((msda_fwd_d32<SP, UN, PT, IO>), dim3((unsigned)((int64_t)N * (TILES) * M)), dim3(256), \
                       lds, st, value, spatial_shapes, level_start, io, S, M, L, Lq, P, (TILES), out)
    if constexpr (!IO::kSoftmax) {
        if (g_fwd_variant == 600 && pixels && L * P == 16 && P == kPT) {
            // LDS-window forward (msda_lw.h): 8 x 8 query patches, three workgroups per CU
            const int bound = (S + 63) / 64 * 5 / 4 + 4 * L;
            SEMIDETR_REQUIRE((int64_t)N * bound * M < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_forward: grid too large");
            hipLaunchKernelGGL((msda_fwd_d32_lw<IO>), dim3((unsigned)(N * bound * M)), dim3(256), lw_lds_bytes(), st, value,
                               spatial_shapes, level_start, io, S, M, L, bound, out);
            g_last_kernels = "msda_fwd_d32_lw";
            return semidetr::launch_status("msda_fwd_d32_lw");
        }
    }
    if (g_fwd_variant >= 720 && g_fwd_variant <= 729 && pixels) {
        // producer / consumer forward (msda_fwd_d32_ws): 4 consumer waves + 1 producer wave, the grid sized to the chip
        // ((variant - 720) workgroups per CU, 0 -> 4).  Parity-green; measured 237.6 us (3 per CU) against 230.3 us for the
        // default at bs 4 -- taking the record phase off the consumers' critical path buys nothing, the vector-memory path is
        // limited by its misses in flight, not by how many waves feed it.
        const int per_cu = g_fwd_variant == 720 ? 4 : g_fwd_variant - 720;
        const int bound = (S + 31) / 32 * 5 / 4 + 4 * L;
        const size_t wlds = (size_t)2 * 2 * 32 * (L * P + 1) * 16;
        SEMIDETR_REQUIRE(wlds <= 64 * 1024, SEMIDETR_E_BADARG, "msda_forward: the producer / consumer kernel needs L * P <= 63");
        const int hint = std::max(1, std::min(bound, (256 * per_cu + N * M - 1) / (N * M)));
        hipLaunchKernelGGL((msda_fwd_d32_ws<IO>), dim3((unsigned)((int64_t)N * hint * M)), dim3(kWsThreads), wlds, st, value,
                           spatial_shapes, level_start, io, S, M, L, Lq, P, hint, out);
        g_last_kernels = "msda_fwd_d32_ws";
        return semidetr::launch_status("msda_fwd_d32_ws");
    }
    if ((g_fwd_variant >= 700 && g_fwd_variant <= 719) || (g_fwd_variant >= 730 && g_fwd_variant <= 759)) {
        SEMIDETR_REQUIRE(pixels && P == kPT && (L == 4 || L == 5), SEMIDETR_E_BADARG,
                         "msda_forward: the region-window kernel needs SEMIDETR_MSDA_QUERIES_ARE_PIXELS, num_point == 4, 4 or 5 levels");
        g_last_kernels = "msda_rw_d32";
        if (L == 4)
            return launch_rw_cfg<IO, 4, false>(g_fwd_variant - 700, st, nullptr, value, spatial_shapes, level_start, io, N, S, M, out, nullptr, 0);
        return launch_rw_cfg<IO, 5, false>(g_fwd_variant - 700, st, nullptr, value, spatial_shapes, level_start, io, N, S, M, out, nullptr, 0);
    }
    if (g_fwd_variant >= 500 && g_fwd_variant <= 505) {
        SEMIDETR_REQUIRE(pixels, SEMIDETR_E_BADARG, "msda_forward: the resident-level kernel needs SEMIDETR_MSDA_QUERIES_ARE_PIXELS");
        // 500: 8 patches per workgroup, coarse levels resident; 501: same schedule, nothing resident (control);
        // 502 / 503: 4 patches per workgroup resident / control; 504 / 505: 2 patches
        const int grp = g_fwd_variant <= 501 ? 8 : (g_fwd_variant <= 503 ? 4 : 2);
        const int res_max = (g_fwd_variant & 1) ? 0 : kResRows;
        const int G = ((S + 31) / 32 * 5 / 4 + 4 * L + grp - 1) / grp;      // grid sizing hint as for the patch kernel
        const size_t rlds = (size_t)(kResRows + 1) * 128 + (size_t)2 * 32 * (L * P + 1) * 32;
        SEMIDETR_REQUIRE(rlds <= 160 * 1024, SEMIDETR_E_BADARG, "msda_forward: too many samples per query for the resident-level kernel");
#define LAUNCH_RES(GRP)                                                                                              \
        do {                                                                                                             \
            static bool lds_ok = false; /* dynamic LDS above 64 KB has to be allowed once per kernel */                \
            if (!lds_ok) {                                                                                               \
                const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_d32_res<IO, GRP>),   \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      \
                if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_forward: hipFuncSetAttribute: %s", hipGetErrorString(ae)); \
                lds_ok = true;                                                                                           \
            }                                                                                                            \
            hipLaunchKernelGGL((msda_fwd_d32_res<IO, GRP>), dim3((unsigned)(N * M * G)), dim3(512), rlds, st, value,     \
                               spatial_shapes, level_start, io, S, M, L, P, G, res_max, out);                          \
        } while (0)
        if (grp == 8) LAUNCH_RES(8);
        else if (grp == 4) LAUNCH_RES(4);
        else LAUNCH_RES(2);
#undef LAUNCH_RES
        g_last_kernels = "msda_fwd_d32_res";
        return semidetr::launch_status("msda_fwd_d32_res");
    }
    if ((pixels && g_fwd_variant == 0) || g_fwd_variant == 408 || g_fwd_variant == 804 || g_fwd_variant == 216) {
        SEMIDETR_REQUIRE(pixels, SEMIDETR_E_BADARG, "msda_forward: patch tiling needs SEMIDETR_MSDA_QUERIES_ARE_PIXELS");
        // grid sizing hint: about the number of 32-pixel patches of a usual pyramid (ragged edges included)
        const int bound = (S + 31) / 32 * 5 / 4 + 4 * L;
        SEMIDETR_REQUIRE((int64_t)N * bound * M < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_forward: grid too large");
        const size_t lds = (size_t)32 * (L * P + 1) * 32;
        // measured at the 800x1333 encoder shape, bs 4, with the head rotation of tile_of_block: 4x8 246 us, 8x4 252,
        // 2x16 253 (before the rotation: strips 299, 4x8 284, 8x4 281, 2x16 286)
        if (g_fwd_variant == 408) LAUNCH_FWD(1, 4, 408, bound);
        else if (g_fwd_variant == 216) LAUNCH_FWD(1, 4, 216, bound);
        else if (g_fwd_variant == 804) LAUNCH_FWD(1, 4, 804, bound);
        else LAUNCH_FWD(1, 4, 408, bound);
        g_last_kernels = "msda_fwd_d32<1, 4, 408";
        return semidetr::launch_status("msda_fwd_d32<patch>");
    }
    const int unroll = g_fwd_variant >= 10 && g_fwd_variant < 100 ? g_fwd_variant / 10 : 4;
    if (split == 1) { if (unroll == 2) LAUNCH_FWD(1, 2, 0, tiles); else if (unroll == 1) LAUNCH_FWD(1, 1, 0, tiles); else LAUNCH_FWD(1, 4, 0, tiles); }
    else if (split == 2) LAUNCH_FWD(2, 4, 0, tiles);
    else LAUNCH_FWD(4, 4, 0, tiles);
#undef LAUNCH_FWD
    g_last_kernels = split == 1 ? "msda_fwd_d32<1, 4, 0" : (split == 2 ? "msda_fwd_d32<2, 4, 0" : "msda_fwd_d32<4, 4, 0");
    return semidetr::launch_status("msda_fwd_d32");
}

// ---- strips backward (any query set).  Experiment (variants 808 / 832): SPLIT in two launches so that the zero fill of
// grad_value overlaps the half that does not need it: a side stream (one per host thread, joined back before the call returns to the caller's
// stream order) runs the fill while the caller's stream runs msda_bwd_gather_d32 (grad_sampling_loc / grad_attn_weight:
// reads value corners, never touches grad_value); the scatter-only instantiation of msda_bwd_d32 follows once both are
// done.  At the BASELINE micro-benchmark shape the 45.5 MB fill is 7-8 us of a 43 us backward.  Works under stream
// capture (the side stream forks from and joins the capturing stream through events).
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int device = -1;
};
thread_local SideStream g_side;

int side_stream(SideStream **out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward: hipGetDevice: %s", hipGetErrorString(e));
    if (g_side.device != dev) {      // first call of this thread on this device (objects of another device are leaked: rare)
        e = hipStreamCreateWithFlags(&g_side.s, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g_side.fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g_side.join, hipEventDisableTiming);
        if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward: side stream: %s", hipGetErrorString(e));
        g_side.device = dev;
    }
    *out = &g_side;
    return SEMIDETR_OK;
}

// zero-fill `bytes` at `ptr` on the side stream, ordered after everything already queued on `st`
int fill_on_side(hipStream_t st, void *ptr, size_t bytes, SideStream **side)
{
    if (int rc = side_stream(side)) return rc;
    hipError_t e = hipEventRecord((*side)->fork, st);
    if (e == hipSuccess) e = hipStreamWaitEvent((*side)->s, (*side)->fork, 0);
    if (e == hipSuccess) e = hipMemsetAsync(ptr, 0, bytes, (*side)->s);
    if (e == hipSuccess) e = hipEventRecord((*side)->join, (*side)->s);
    if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward fill on the side stream: %s", hipGetErrorString(e));
    return SEMIDETR_OK;
}
int join_side(hipStream_t st, SideStream *side)
{
    const hipError_t e = hipStreamWaitEvent(st, side->join, 0);
    if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward join: %s", hipGetErrorString(e));
    return SEMIDETR_OK;
}

template <typename IO>
int exp_launch_strips_backward(hipStream_t st, size_t fill, const float *grad_out, const float *value,
                           const int64_t *spatial_shapes, const int64_t *level_start, const IO &io, int N, int S, int M,
                           int L, int Lq, int P, int rpb, int tiles, unsigned grid, size_t lds, float *grad_value)
{
    // default: ONE kernel after the fill on the caller's stream.  The split below lost on MI355X (measured, r02): the
    // fork / join through events costs more than the 7-8 us of fill it hides -- micro-benchmark backward 42.8 -> 63.1 us,
    // decoder bs 4 242 -> 267 us, encoder bs 4 875 -> 893 us.  Kept selectable (808 / 832) as the evidence.
    // level-aggregated scatter + gather in ONE merged launch: default for launches with at least 512 (image, query) pairs
    // (measured: decoder bs 4 / Lq 1100 240 -> 163 us, bs 1 66 -> 54 us, BASELINE micro-benchmark shape 43.2 -> 39.7 us;
    // as two launches -- gather, then scatter -- the micro-benchmark shape gained nothing: 8 + 27 us against 35 us fused)
    if ((g_bwd_variant == 900 || (g_bwd_variant == 0 && (int64_t)N * Lq >= 512)) && P <= 8) {
        // fill, then ONE launch: level-aggregated scatter workgroups + gather workgroups side by side
        hipError_t e = hipMemsetAsync(grad_value, 0, fill, st);
        if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward memset: %s", hipGetErrorString(e));
        // chunks of <= kLvlQ queries; small launches are cut finer so that at least ~128 scatter workgroups exist.  Measured
        // at the micro-benchmark shape (N=2, Lq=300: 64 (image, head, level) triples): 1 chunk 42.1 us, 2 chunks 36.1 us,
        // 4 chunks 40.1 us, 8 chunks 38.7 us -- bigger chunks aggregate more, a single one leaves the chip idle.
        int chunks = (Lq + kLvlQ - 1) / kLvlQ;
        static const int target_wgs = getenv("SEMIDETR_LVL_WGS") ? atoi(getenv("SEMIDETR_LVL_WGS")) : 128;   // tuning aid
        const int want = (target_wgs + N * L * M - 1) / (N * L * M);
        chunks = std::max(chunks, std::min(want, (Lq + 63) / 64));
        const int chunk_q = (Lq + chunks - 1) / chunks;
        const int gt = (Lq + 31) / 32;                                   // gather: 32 query rows per 256-thread block
        const int64_t gblocks = (int64_t)N * gt * M, sblocks = (int64_t)N * chunks * L * M;
        const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
        const size_t slds = std::max((size_t)kLvlQ * kD * 4 + ((size_t)chunk_q * P * 4 + 8) * 8 + (size_t)2 * kLvlRows * 4,
                                     2 * half_f4 * 16);
        const int64_t grid = sblocks + (gblocks + 1) / 2;
        SEMIDETR_REQUIRE(grid < INT32_MAX && slds <= 159 * 1024, SEMIDETR_E_TOOLARGE, "msda_backward: merged launch too large");
#define LAUNCH_MERGED(KLP_)                                                                                          \
        do {                                                                                                             \
            static bool lds_ok = false; /* dynamic LDS above 64 KB has to be allowed once per kernel (static LDS too) */ \
            if (!lds_ok) {                                                                                               \
                const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_lvl_merged<IO, KLP_>), \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);      \
                if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_backward: hipFuncSetAttribute: %s", hipGetErrorString(ae)); \
                lds_ok = true;                                                                                           \
            }                                                                                                            \
            hipLaunchKernelGGL((msda_bwd_lvl_merged<IO, KLP_>), dim3((unsigned)grid), dim3(kLvlThreads), slds, st, grad_out, \
                               value, spatial_shapes, level_start, io, S, M, L, Lq, P, chunks, chunk_q, (int)sblocks, gt, \
                               (int)gblocks, grad_value);                                                              \
        } while (0)
        if (L * P == 16) LAUNCH_MERGED(16);
        else LAUNCH_MERGED(0);
#undef LAUNCH_MERGED
        g_last_kernels = "fillBufferAligned+msda_bwd_lvl_merged";
        return semidetr::launch_status("msda_bwd_lvl_merged");
    }
    if (g_bwd_variant == 910 && P <= 8 && (int64_t)Lq * P <= kOwnCap && (reinterpret_cast<uintptr_t>(grad_value) & 15) == 0) {
        // EXPERIMENT: owner-computes grad_value (stores only: no fill, no float atomics) + gather workgroups, ONE launch
        // (msda_own.h).  Requires the levels to tile the value rows exactly (rows no level covers are not written): the
        // tests that select it use canonical pyramids.
        const int tb = own_tiles_bound(S, L);
        const int gt = (Lq + 31) / 32;
        const int64_t gblocks = (int64_t)N * gt * M, sblocks = (int64_t)N * tb * M;
        const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
        const size_t slds = std::max(kOwnLdsBytes, 2 * half_f4 * 16);
        const int64_t grid = sblocks + (gblocks + 1) / 2;
        SEMIDETR_REQUIRE(grid < INT32_MAX && slds <= 159 * 1024, SEMIDETR_E_TOOLARGE, "msda_backward: merged launch too large");
#define LAUNCH_OWN(KLP_, SPT_)                                                                                       \
        do {                                                                                                             \
            auto kern = &msda_bwd_own_merged<IO, KLP_, SPT_>;                                                            \
            if (int rc = allow_big_lds(kern, slds, "msda_backward")) return rc;                                          \
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kOwnThreads), slds, st, grad_out, value, spatial_shapes, \
                               level_start, io, S, M, L, Lq, P, tb, (int)sblocks, gt, (int)gblocks, grad_value);        \
        } while (0)
        const int spt = (Lq * P + kOwnThreads - 1) / kOwnThreads;
        if (L * P == 16) { if (spt <= 3) LAUNCH_OWN(16, 3); else LAUNCH_OWN(16, 9); }
        else             { if (spt <= 3) LAUNCH_OWN(0, 3);  else LAUNCH_OWN(0, 9); }
#undef LAUNCH_OWN
        g_last_kernels = "msda_bwd_own_merged";
        return semidetr::launch_status("msda_bwd_own_merged");
    }
    if (g_bwd_variant == 902 && P <= 8 && (reinterpret_cast<uintptr_t>(grad_value) & 15) == 0) {
        // EXPERIMENT: cooperative zero fill inside the merged launch (msda_bwd_lvl_coop), no hipMemsetAsync.  The three
        // counters of a launch live in a library-owned device buffer (64 slots handed out round robin; the kernel leaves its
        // slot zeroed).  Not capturable on its first call (hipMalloc), not meant for concurrent replays of one captured graph.
        static unsigned *sync_buf = nullptr;
        static int sync_dev = -1, next_slot = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!sync_buf || dev != sync_dev) {
            unsigned *p = nullptr;
            hipError_t e = hipMalloc(&p, 64 * 4 * sizeof(unsigned));
            if (e == hipSuccess) e = hipMemset(p, 0, 64 * 4 * sizeof(unsigned));
            if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward: sync buffer: %s", hipGetErrorString(e));
            sync_buf = p;
            sync_dev = dev;
        }
        unsigned *slot = sync_buf + 4 * (next_slot++ & 63);
        int chunks = (Lq + kLvlQ - 1) / kLvlQ;
        const int want = (128 + N * L * M - 1) / (N * L * M);
        chunks = std::max(chunks, std::min(want, (Lq + 63) / 64));
        const int chunk_q = (Lq + chunks - 1) / chunks;
        const int gt = (Lq + 31) / 32;
        const int64_t gblocks = (int64_t)N * gt * M, sblocks = (int64_t)N * chunks * L * M;
        const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
        const size_t slds = std::max((size_t)kLvlQ * kD * 4 + ((size_t)chunk_q * P * 4 + 8) * 8 + (size_t)2 * kLvlRows * 4,
                                     2 * half_f4 * 16);
        const int64_t grid = sblocks + (gblocks + 1) / 2;
        SEMIDETR_REQUIRE(grid < INT32_MAX && slds <= 159 * 1024, SEMIDETR_E_TOOLARGE, "msda_backward: merged launch too large");
#define LAUNCH_COOP(KLP_)                                                                                            \
        do {                                                                                                             \
            static bool lds_ok = false;                                                                                  \
            if (!lds_ok) {                                                                                               \
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_lvl_coop<IO, KLP_>),                  \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);                      \
                lds_ok = true;                                                                                           \
            }                                                                                                            \
            hipLaunchKernelGGL((msda_bwd_lvl_coop<IO, KLP_>), dim3((unsigned)grid), dim3(kLvlThreads), slds, st, grad_out,   \
                               value, spatial_shapes, level_start, io, S, M, L, Lq, P, chunks, chunk_q, (int)sblocks, gt, \
                               (int)gblocks, grad_value, reinterpret_cast<float4 *>(grad_value), (int64_t)(fill / 16),  \
                               slot, 1 << 20);                                                                         \
        } while (0)
        if (L * P == 16) LAUNCH_COOP(16);
        else LAUNCH_COOP(0);
#undef LAUNCH_COOP
        g_last_kernels = "msda_bwd_lvl_coop";
        return semidetr::launch_status("msda_bwd_lvl_coop");
    }
    if (g_bwd_variant == 901 && P <= 8) {
        // experiment: NO memset -- the gather launch zero-fills grad_value as a side job, the level-aggregated scatter
        // follows as its own launch
        int chunks = (Lq + kLvlQ - 1) / kLvlQ;
        const int want = (128 + N * L * M - 1) / (N * L * M);
        chunks = std::max(chunks, std::min(want, (Lq + 63) / 64));
        const int chunk_q = (Lq + chunks - 1) / chunks;
        const int gt = (Lq + 31) / 32;
        const size_t glds = (size_t)32 * (L * P + 1) * 32 + 2 * kMaxLevels * sizeof(float);
        SEMIDETR_REQUIRE(fill % 16 == 0, SEMIDETR_E_BADARG, "msda_backward: grad_value size not a multiple of 16 bytes");
        if (L * P == 16)
            hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt,
                               reinterpret_cast<float4 *>(grad_value), (int64_t)(fill / 16));
        else
            hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 0>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt,
                               reinterpret_cast<float4 *>(grad_value), (int64_t)(fill / 16));
        if (int rc = semidetr::launch_status("msda_bwd_gather_d32")) return rc;
        const size_t slds = (size_t)kLvlQ * kD * 4 + ((size_t)chunk_q * P * 4 + 8) * 8 + (size_t)2 * kLvlRows * 4;
        static bool lds_ok = false;
        if (!lds_ok) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_scatter_d32_lvl<IO>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
            lds_ok = true;
        }
        hipLaunchKernelGGL((msda_bwd_scatter_d32_lvl<IO>), dim3((unsigned)((int64_t)N * chunks * L * M)), dim3(kLvlThreads), slds, st,
                           grad_out, spatial_shapes, level_start, io, S, M, L, Lq, P, chunks, chunk_q, grad_value);
        g_last_kernels = "msda_bwd_gather_d32+msda_bwd_scatter_d32_lvl";
        return semidetr::launch_status("msda_bwd_scatter_d32_lvl");
    }
    const bool fused = g_bwd_variant != 808 && g_bwd_variant != 832;
    if (fused) {
        hipError_t e = hipMemsetAsync(grad_value, 0, fill, st);
        if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward memset: %s", hipGetErrorString(e));
        if (rpb == 32)
            hipLaunchKernelGGL((msda_bwd_d32<32, IO>), dim3(grid), dim3(256), lds, st, grad_out, value, spatial_shapes,
                               level_start, io, S, M, L, Lq, P, tiles, grad_value);
        else
            hipLaunchKernelGGL((msda_bwd_d32<8, IO>), dim3(grid), dim3(256), lds, st, grad_out, value, spatial_shapes,
                               level_start, io, S, M, L, Lq, P, tiles, grad_value);
        g_last_kernels = rpb == 32 ? "fillBufferAligned+msda_bwd_d32<32" : "fillBufferAligned+msda_bwd_d32<8";
        return semidetr::launch_status("msda_bwd_d32");
    }
    SideStream *side = nullptr;
    if (int rc = fill_on_side(st, grad_value, fill, &side)) return rc;
    {   // the two small gradients (32 query rows per workgroup)
        const int gt = (Lq + 31) / 32;
        const size_t glds = (size_t)32 * (L * P + 1) * 32 + 2 * kMaxLevels * sizeof(float);
        if (L * P == 16)
            hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt);
        else
            hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 0>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt);
        const int grc = semidetr::launch_status("msda_bwd_gather_d32");
        if (int rc = join_side(st, side)) return rc;            // never leave the side stream un-joined
        if (grc) return grc;
    }
    if (rpb == 32)
        hipLaunchKernelGGL((msda_bwd_d32<32, IO, true>), dim3(grid), dim3(256), lds, st, grad_out, value, spatial_shapes,
                           level_start, io, S, M, L, Lq, P, tiles, grad_value);
    else
        hipLaunchKernelGGL((msda_bwd_d32<8, IO, true>), dim3(grid), dim3(256), lds, st, grad_out, value, spatial_shapes,
                           level_start, io, S, M, L, Lq, P, tiles, grad_value);
    g_last_kernels = rpb == 32 ? "fillBufferAligned+msda_bwd_gather_d32+msda_bwd_d32<32" : "fillBufferAligned+msda_bwd_gather_d32+msda_bwd_d32<8";
    return semidetr::launch_status("msda_bwd_d32<scatter>");
}

template <typename IO>
int exp_launch_fast_backward(hipStream_t st, const float *grad_out, const float *value, const int64_t *spatial_shapes,
                         const int64_t *level_start, const IO &io, int N, int S, int M, int L, int Lq, int P,
                         int flags, float *grad_value)
{
    const bool pixels = (flags & SEMIDETR_MSDA_QUERIES_ARE_PIXELS) != 0;
    SEMIDETR_REQUIRE(!pixels || Lq == S, SEMIDETR_E_BADARG,
                     "msda_backward: SEMIDETR_MSDA_QUERIES_ARE_PIXELS needs num_query == spatial_size");
    const size_t fill = sizeof(float) * (size_t)N * S * M * kD;
    const bool win_ok = P == kPT;
    if ((pixels && S < (1 << 24) && win_ok && g_bwd_variant == 0) || ((g_bwd_variant >= 64 && g_bwd_variant <= 74) || (g_bwd_variant >= 690 && g_bwd_variant <= 699) || (g_bwd_variant >= 6900 && g_bwd_variant <= 7009))) {
        SEMIDETR_REQUIRE(pixels && S < (1 << 24), SEMIDETR_E_BADARG,
                         "msda_backward: the self-attention kernels need SEMIDETR_MSDA_QUERIES_ARE_PIXELS and spatial_size < 2^24");
        // the unrolled gather launch clears grad_value as a side job (no hipMemsetAsync): the scatter that accumulates into it
        // is the NEXT launch.  Measured (tools/archive/r02_fillgather_try.sh): encoder bs 4 774 -> 766 us, bs 1 204 -> 201 us; 6991
        // forces the memset.  (The same idea for arbitrary query sets -- gather + fill, then the level scatter as a second
        // launch, variant 901 -- loses against the merged launch: micro-benchmark 36.2 -> 41-44 us, decoder bs 4 161 -> 169.)
        const bool rw_gather = g_bwd_variant >= 7000 && g_bwd_variant <= 7009 && P == kPT && (L == 4 || L == 5);
        const bool gather_kernel_runs = !(g_bwd_variant == 697 || g_bwd_variant == 68 || (g_bwd_variant >= 6900 && g_bwd_variant <= 6919));
        const bool fill_in_gather = (L * P == 16 || rw_gather) && gather_kernel_runs && g_bwd_variant != 6991 && g_bwd_variant != 66 &&
                                    g_bwd_variant != 67 && g_bwd_variant != 6962 && g_bwd_variant != 6952 && g_bwd_variant != 6948 &&
                                    (reinterpret_cast<uintptr_t>(grad_value) & 15) == 0;
        if (!fill_in_gather) {
            hipError_t e = hipMemsetAsync(grad_value, 0, fill, st);
            if (e != hipSuccess) return semidetr::fail((int)e, "msda_backward memset: %s", hipGetErrorString(e));
        }
        if (g_bwd_variant >= 6900 && g_bwd_variant <= 6919 && P == kPT && S < (1 << 23)) {
            // round 4: the whole encoder backward in ONE kernel after the fill (msda_bwd_enc_fused_d32); 6901 = instrumented
            const int rbound = (S + 127) / 128 * 5 / 4 + 4 * L;
            const int64_t rgrid = (int64_t)N * rbound * M;
            SEMIDETR_REQUIRE(rgrid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
            const size_t rlds = reg_lds_bytes<512, 208, 24, 32, 1>();
#define LAUNCH_FUSED(DBG_, AID_) LAUNCH_FUSEDZ(DBG_, AID_, 1)
#define LAUNCH_FUSEDZ(DBG_, AID_, FZ_)                                                                                     \
            do {                                                                                                             \
                auto kern = &msda_bwd_enc_fused_d32<IO, 512, 208, 8, 16, 24, 32, DBG_, 4, 8, AID_, FZ_>;                     \
                if (int rc = allow_big_lds(kern, rlds, "msda_backward")) return rc;                                          \
                hipLaunchKernelGGL(kern, dim3((unsigned)rgrid), dim3(512), rlds, st, grad_out, value, spatial_shapes,        \
                                   level_start, io, S, M, L, rbound, grad_value);                                            \
            } while (0)
            if (g_bwd_variant == 6901) LAUNCH_FUSED(1, 0);
            else if (g_bwd_variant == 6902) LAUNCH_FUSED(0, 1);       // timing aids (results wrong), see reg_scatter_body
            else if (g_bwd_variant == 6903) LAUNCH_FUSED(0, 2);
            else if (g_bwd_variant == 6904) LAUNCH_FUSED(0, 4);
            else if (g_bwd_variant == 6905) LAUNCH_FUSED(0, 8);
            else if (g_bwd_variant == 6906) LAUNCH_FUSED(0, 16);
            else if (g_bwd_variant == 6907) LAUNCH_FUSED(0, 7);
            else if (g_bwd_variant == 6908) LAUNCH_FUSED(0, 24);
            else if (g_bwd_variant == 6909) LAUNCH_FUSEDZ(0, 0, 2);   // four consecutive entries per lane in the dot phase
            else if (g_bwd_variant == 6910) LAUNCH_FUSEDZ(0, 32, 1);  // streaming result stores
            else if (g_bwd_variant == 6911) LAUNCH_FUSEDZ(0, 32, 2);
            else LAUNCH_FUSED(0, 0);
#undef LAUNCH_FUSED
#undef LAUNCH_FUSEDZ
            g_last_kernels = "fillBufferAligned+msda_bwd_enc_fused_d32";
            return semidetr::launch_status("msda_bwd_enc_fused_d32");
        }
        if (g_bwd_variant == 697 && L * P == 16 && P == kPT && S < (1 << 23)) {
            // experiment: region scatter + gather in ONE launch, roles dealt out in groups of eight workgroups
            const int rbound = (S + 127) / 128 * 5 / 4 + 4 * L;
            const int gbound = (S + 31) / 32 * 5 / 4 + 4 * L;
            const int64_t sblocks = (int64_t)N * rbound * M, gblocks = (int64_t)N * gbound * M, gwgs = (gblocks + 1) / 2;
            const int64_t sgroups = (sblocks + 7) / 8, ggroups = (gwgs + 7) / 8;
            const int period = (int)std::max<int64_t>(2, (sgroups + ggroups) / sgroups);
            const int64_t groups = std::max(sgroups + ggroups, (sgroups - 1) * period + 1);
            SEMIDETR_REQUIRE(groups * 8 < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
            const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
            const size_t mlds = std::max(reg_lds_bytes<512, 208, 24, 32>(), 2 * half_f4 * 16);
            auto kern = &msda_bwd_encreg_merged<IO, 16>;
            static bool lds_ok = false;
            if (!lds_ok) {
                const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
                if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_backward: hipFuncSetAttribute: %s", hipGetErrorString(ae));
                lds_ok = true;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)(groups * 8)), dim3(512), mlds, st, grad_out, value, spatial_shapes,
                               level_start, io, S, M, L, P, rbound, (int)sblocks, gbound, (int)gblocks, (int)sgroups, period,
                               grad_value);
            g_last_kernels = "fillBufferAligned+msda_bwd_encreg_merged";
            return semidetr::launch_status("msda_bwd_encreg_merged");
        }
        if (g_bwd_variant == 68 && L * P == 16 && P == kPT) {
            // Experiment (variant 68): ONE launch, windowed-scatter workgroups interleaved with pairs of gather blocks
            // (msda_bwd_enc_merged).  Measured at bs 4: 1771 us against 879 us for the two launches -- every workgroup of
            // the launch carries the scatter's 77 KB of LDS, so a CU holds two workgroups in any mix and the gather, which
            // needs 16-24 waves per CU to stream, is left with 8.  (The same trick WINS for arbitrary query sets, where
            // the scatter's LDS is small enough for the halves to share CUs at full occupancy: msda_bwd_lvl_merged.)
            const int tiles_bound = (S + 255) / 256 * 5 / 4 + 4 * L;
            const int gbound = (S + 31) / 32 * 5 / 4 + 4 * L;
            const int64_t sblocks = (int64_t)N * tiles_bound * M, gblocks = (int64_t)N * gbound * M;
            const int64_t gwgs = (gblocks + 1) / 2, total = sblocks + gwgs;
            const int period = (int)std::max<int64_t>(2, total / sblocks);         // every period-th workgroup scatters
            // all scatter indices 0, period, 2*period, ... must fall inside the grid
            const int64_t grid = std::max(total, (sblocks - 1) * period + 1);
            SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
            const size_t half_f4 = (size_t)2 * 32 * (L * P + 1) + 2 * kMaxLevels / 4 + 4;
            const size_t mlds = std::max(win_lds_bytes<16, 16, 32, 32>(), 2 * half_f4 * 16);
            static bool lds_ok = false;
            if (!lds_ok) {
                const hipError_t ae = hipFuncSetAttribute(
                    reinterpret_cast<const void *>(&msda_bwd_enc_merged<IO, 16, 16, 16, 32, 32>),
                    hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
                if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_backward: hipFuncSetAttribute: %s", hipGetErrorString(ae));
                lds_ok = true;
            }
            hipLaunchKernelGGL((msda_bwd_enc_merged<IO, 16, 16, 16, 32, 32>), dim3((unsigned)grid), dim3(kWinThreads), mlds, st,
                               grad_out, value, spatial_shapes, level_start, io, S, M, L, P, tiles_bound, (int)sblocks, gbound,
                               (int)gblocks, period, grad_value);
            g_last_kernels = "fillBufferAligned+msda_bwd_enc_merged";
            return semidetr::launch_status("msda_bwd_enc_merged");
        }
        if (rw_gather) {   // gather half through the region windows (msda_rw.h)
            float4 *z = fill_in_gather ? reinterpret_cast<float4 *>(grad_value) : nullptr;
            const int rc = L == 4 ? launch_rw_cfg<IO, 4, true>(g_bwd_variant - 7000, st, grad_out, value, spatial_shapes, level_start,
                                                               io, N, S, M, nullptr, z, (int64_t)(fill / 16))
                                  : launch_rw_cfg<IO, 5, true>(0, st, grad_out, value, spatial_shapes, level_start, io, N, S, M,
                                                               nullptr, z, (int64_t)(fill / 16));
            if (rc) return rc;
        } else
        {   // gather half: the two small gradients, streams like the forward
            const int gt = (Lq + 31) / 32;
            const size_t glds = (size_t)32 * (L * P + 1) * 32 + 2 * kMaxLevels * sizeof(float);
            // DINO (L * P == 16): sample loop unrolled, results in registers, 8 x 4 query patches like the forward;
            // measured at the encoder shape, bs 4: generic strips 370 us, unrolled strips 346 us; 66 / 67 force them
            const int gbound = (S + 31) / 32 * 5 / 4 + 4 * L;      // patch grid hint, see launch_fast_forward
            if (L * P == 16 && g_bwd_variant == 6962)            // tuning: 8 loads in flight, 6 waves per SIMD
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408, 6, 2>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound);
            else if (L * P == 16 && g_bwd_variant == 6952)       // 8 loads in flight, 5 waves per SIMD
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408, 5, 2>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound);
            else if (L * P == 16 && g_bwd_variant == 6948)       // timing aid: nothing stored
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408, 4, 104>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound);
            else if (L * P == 16 && g_bwd_variant == 920) {       // four lanes per query, 8 x 8 patches (msda_bwd_gather4_d32)
                const int gbound4 = (S + 63) / 64 * 5 / 4 + 4 * L;
                const size_t glds4 = (size_t)2 * 64 * (L * P + 1) * 16 + 2 * kMaxLevels * sizeof(float);
                hipLaunchKernelGGL((msda_bwd_gather4_d32<IO, 16>), dim3((unsigned)((int64_t)N * gbound4 * M)), dim3(256), glds4, st,
                                   grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound4,
                                   fill_in_gather ? reinterpret_cast<float4 *>(grad_value) : nullptr, (int64_t)(fill / 16));
            } else if (L * P == 16 && g_bwd_variant == 921)       // rolling window of 16 corner loads
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408, 4, 16>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound,
                                   fill_in_gather ? reinterpret_cast<float4 *>(grad_value) : nullptr, (int64_t)(fill / 16));
            else if (L * P == 16 && g_bwd_variant == 922)         // rolling window of 8 corner loads, 5 waves per SIMD
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408, 5, 8>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound,
                                   fill_in_gather ? reinterpret_cast<float4 *>(grad_value) : nullptr, (int64_t)(fill / 16));
            else if (fill_in_gather)
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound,
                                   reinterpret_cast<float4 *>(grad_value), (int64_t)(fill / 16));
            else if (L * P == 16 && g_bwd_variant != 66 && g_bwd_variant != 67)
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16, 408>), dim3((unsigned)((int64_t)N * gbound * M)), dim3(256), glds,
                                   st, grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gbound);
            else if (L * P == 16 && g_bwd_variant == 67)
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 16>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                                   grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt);
            else
                hipLaunchKernelGGL((msda_bwd_gather_d32<IO, 0>), dim3((unsigned)((int64_t)N * gt * M)), dim3(256), glds, st,
                                   grad_out, value, spatial_shapes, level_start, io, S, M, L, Lq, P, gt);
            if (int rc = semidetr::launch_status("msda_bwd_gather_d32")) return rc;
        }
        if ((g_bwd_variant == 0 || (g_bwd_variant >= 69 && g_bwd_variant <= 69) || (g_bwd_variant >= 690 && g_bwd_variant <= 699) || (g_bwd_variant >= 6900 && g_bwd_variant <= 7009)) && P == kPT && S < (1 << 23)) {
            // region-owned windowed scatter (msda_region.h): one workgroup per tile of the finest level, all query levels.
            // DEFAULT since round 2.  Measured at the 800x1333 encoder shape (backward incl. fill + gather): bs 4 886 us
            // (windowed kernel, variant 65) -> 867 us (16 x 16 regions, 1024 threads, 690) -> 823 us (8 x 16 regions, 512
            // threads, two workgroups per CU); bs 1 248 -> 226 -> 216 us; row atomics 590 MB -> 358 MB (16 x 16).
            const bool small = g_bwd_variant != 690;              // 8 x 16 regions, 512 threads, two workgroups per CU
            const int rpx = g_bwd_variant == 692 || g_bwd_variant == 693 ? 64 : (small ? 128 : 256);
            const int rbound = (S + rpx - 1) / rpx * 5 / 4 + 4 * L;
            const int64_t rgrid = (int64_t)N * rbound * M;
            SEMIDETR_REQUIRE(rgrid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
#define LAUNCH_REG(NT_, Q_, RH_, RW_, WH_, WW_) LAUNCH_REGW(NT_, Q_, RH_, RW_, WH_, WW_, 4)
#define LAUNCH_REGW(NT_, Q_, RH_, RW_, WH_, WW_, WPE_) LAUNCH_REGU(NT_, Q_, RH_, RW_, WH_, WW_, WPE_, 8)
#define LAUNCH_REGU(NT_, Q_, RH_, RW_, WH_, WW_, WPE_, WU_)                                                                   \
            do {                                                                                                         \
                static bool lds_ok = false;                                                                              \
                auto kern = &msda_bwd_scatter_d32_reg<IO, NT_, Q_, RH_, RW_, WH_, WW_, 0, WPE_, WU_>;                                  \
                if (!lds_ok) {                                                                                           \
                    const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                      \
                                                              hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);  \
                    if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_backward: hipFuncSetAttribute: %s", hipGetErrorString(ae)); \
                    lds_ok = true;                                                                                       \
                }                                                                                                        \
                const size_t rlds = reg_lds_bytes<NT_, Q_, WH_, WW_>();                                                  \
                hipLaunchKernelGGL(kern, dim3((unsigned)rgrid), dim3(NT_), rlds, st, grad_out, spatial_shapes, level_start, \
                                   io, S, M, L, rbound, grad_value);                                                    \
            } while (0)
            if (g_bwd_variant == 696) {                                             // instrumented build
                auto kern = &msda_bwd_scatter_d32_reg<IO, 512, 208, 8, 16, 24, 32, 1>;
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
                const size_t rlds = reg_lds_bytes<512, 208, 24, 32>();
                hipLaunchKernelGGL(kern, dim3((unsigned)rgrid), dim3(512), rlds, st, grad_out, spatial_shapes, level_start, io, S,
                                   M, L, rbound, grad_value);
            } else if (g_bwd_variant == 6988 || g_bwd_variant == 6989) {            // timing aids: no second-row flushes / no misses
                const size_t rlds = reg_lds_bytes<512, 176, 24, 32>();
                if (g_bwd_variant == 6988)
                    hipLaunchKernelGGL((msda_bwd_scatter_d32_reg_pair_aid<IO, 4>), dim3((unsigned)rgrid), dim3(512), rlds, st, grad_out, spatial_shapes, level_start, io, S, M, L, rbound, grad_value);
                else
                    hipLaunchKernelGGL((msda_bwd_scatter_d32_reg_pair_aid<IO, 8>), dim3((unsigned)rgrid), dim3(512), rlds, st, grad_out, spatial_shapes, level_start, io, S, M, L, rbound, grad_value);
            } else if (g_bwd_variant == 6986 || g_bwd_variant == 6987) {            // paired corners without / with the row atomics
                const size_t rlds = reg_lds_bytes<512, 176, 24, 32>();
                if (g_bwd_variant == 6986)
                    hipLaunchKernelGGL((msda_bwd_scatter_d32_reg_pair_aid<IO, 2>), dim3((unsigned)rgrid), dim3(512), rlds, st, grad_out, spatial_shapes, level_start, io, S, M, L, rbound, grad_value);
                else
                    hipLaunchKernelGGL((msda_bwd_scatter_d32_reg_pair_aid<IO, 0>), dim3((unsigned)rgrid), dim3(512), rlds, st, grad_out, spatial_shapes, level_start, io, S, M, L, rbound, grad_value);
            } else if (g_bwd_variant == 6960 || g_bwd_variant == 6961) {            // instrumented: the product shape / with paired corners
                const size_t rlds = reg_lds_bytes<512, 176, 24, 32>();
                if (g_bwd_variant == 6960) {
                    auto kern = &msda_bwd_scatter_d32_reg<IO, 512, 176, 8, 16, 24, 32, 1, 6, 8>;
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
                    hipLaunchKernelGGL(kern, dim3((unsigned)rgrid), dim3(512), rlds, st, grad_out, spatial_shapes, level_start, io, S, M, L, rbound, grad_value);
                } else {
                    auto kern = &msda_bwd_scatter_d32_reg<IO, 512, 176, 8, 16, 24, 32, 1, 6, 1004>;
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
                    hipLaunchKernelGGL(kern, dim3((unsigned)rgrid), dim3(512), rlds, st, grad_out, spatial_shapes, level_start, io, S, M, L, rbound, grad_value);
                }
            } else if (g_bwd_variant == 691) LAUNCH_REG(512, 208, 16, 8, 32, 24);   // tuning variants
            else if (g_bwd_variant == 692) LAUNCH_REG(256, 112, 8, 8, 24, 24);
            else if (g_bwd_variant == 693) LAUNCH_REG(512, 112, 8, 8, 24, 24);
            else if (g_bwd_variant == 694) LAUNCH_REG(512, 208, 8, 16, 32, 32);
            else if (g_bwd_variant == 695) LAUNCH_REG(768, 208, 8, 16, 24, 32);
            else if (g_bwd_variant == 698) LAUNCH_REGW(512, 176, 8, 16, 24, 32, 6);   // three workgroups per CU
            else if (g_bwd_variant == 699) LAUNCH_REGW(512, 176, 8, 16, 24, 32, 4);   // same LDS, register budget of two
            else if (g_bwd_variant == 6981) LAUNCH_REGU(512, 208, 8, 16, 24, 32, 4, 4);    // walk unrolled by 4
            else if (g_bwd_variant == 6982) LAUNCH_REGU(512, 208, 8, 16, 24, 32, 4, 16);   // ... by 16
            else if (g_bwd_variant == 6985) {                                              // timing aid: no row atomics
                auto kern = &msda_bwd_scatter_d32_reg_noatomics<IO>;
                const size_t rlds = reg_lds_bytes<512, 208, 24, 32>();
                if (int rc = allow_big_lds(kern, rlds, "msda_backward")) return rc;
                hipLaunchKernelGGL(kern, dim3((unsigned)rgrid), dim3(512), rlds, st, grad_out, spatial_shapes, level_start, io, S, M, L,
                                   rbound, grad_value);
            }
            else if (g_bwd_variant == 6983) LAUNCH_REGU(512, 208, 8, 16, 24, 32, 4, 108);  // b128 entry reads, next batch prefetched
            else if (g_bwd_variant == 6984) LAUNCH_REGU(512, 208, 8, 16, 24, 32, 4, 104);  // same, batches of 4
            else if (small) LAUNCH_REG(512, 208, 8, 16, 24, 32);
            else LAUNCH_REG(1024, 384, 16, 16, 32, 32);
#undef LAUNCH_REG
#undef LAUNCH_REGW
#undef LAUNCH_REGU
            g_last_kernels = rw_gather ? "msda_rw_d32<gather>+msda_bwd_scatter_d32_reg" : (fill_in_gather ? "msda_bwd_gather_d32+msda_bwd_scatter_d32_reg" : "fillBufferAligned+msda_bwd_gather_d32+msda_bwd_scatter_d32_reg");
            return semidetr::launch_status("msda_bwd_scatter_d32_reg");
        }
        // grad_value: destination-owned tiles (msda_dest.h) unless a windowed variant is forced (64..67) or the pyramid
        // has more levels than the kernel's LDS tables hold
        if (L <= kDestMaxLevels && g_bwd_variant >= 70 && g_bwd_variant <= 74) {
            // grid sizing hint: about 2.5 units per 256 rows of a usual 4-level pyramid (coarse tiles are split);
            // workgroups take units slot, slot + bound, ... so any bound >= 1 is correct
            const int bound = (S / 256 + 1) * 3 + 64;
            const int64_t grid = (int64_t)N * bound * M;
            SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
            if (g_bwd_variant == 74)
                hipLaunchKernelGGL((msda_bwd_dest_d32<IO, 16, 16, 8, 3>), dim3((unsigned)grid), dim3(kDestThreads), 0, st,
                                   grad_out, spatial_shapes, level_start, io, S, M, L, P, bound, grad_value);
            else if (g_bwd_variant == 73)
                hipLaunchKernelGGL((msda_bwd_dest_d32<IO, 16, 16, 8, 2>), dim3((unsigned)grid), dim3(kDestThreads), 0, st,
                                   grad_out, spatial_shapes, level_start, io, S, M, L, P, bound, grad_value);
            else if (g_bwd_variant == 72)
                hipLaunchKernelGGL((msda_bwd_dest_d32<IO, 16, 16, 8, 1>), dim3((unsigned)grid), dim3(kDestThreads), 0, st,
                                   grad_out, spatial_shapes, level_start, io, S, M, L, P, bound, grad_value);
            else if (g_bwd_variant == 71)
                hipLaunchKernelGGL((msda_bwd_dest_d32<IO, 16, 16, 6>), dim3((unsigned)grid), dim3(kDestThreads), 0, st,
                                   grad_out, spatial_shapes, level_start, io, S, M, L, P, bound, grad_value);
            else
                hipLaunchKernelGGL((msda_bwd_dest_d32<IO, 16, 16, 8>), dim3((unsigned)grid), dim3(kDestThreads), 0, st,
                                   grad_out, spatial_shapes, level_start, io, S, M, L, P, bound, grad_value);
            g_last_kernels = fill_in_gather ? "msda_bwd_gather_d32+msda_bwd_dest_d32" : "fillBufferAligned+msda_bwd_gather_d32+msda_bwd_dest_d32";
            return semidetr::launch_status("msda_bwd_dest_d32");
        }
        // windowed (source-owned) kernel: patches are enumerated on the device (the level table lives in device
        // memory); a workgroup takes patches slot, slot + tiles_bound, ... so any bound >= 1 is correct.
        // measured at the 800x1333 encoder shape, bs 4: 8x16 patches 687 us / 784 MB of row atomics, 16x16 patches
        // 593 us / 604 MB (fewer halo rows per query); 64 forces the small patch
        SEMIDETR_REQUIRE(P == kPT, SEMIDETR_E_BADARG, "msda_backward: the windowed kernel needs num_point == 4");
        const bool big = g_bwd_variant != 64;
        const int patch = big ? 256 : 128;
        const int tiles_bound = (S + patch - 1) / patch * 5 / 4 + 4 * L;
        const int64_t grid = (int64_t)N * tiles_bound * M;
        SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
        const size_t wlds = big ? win_lds_bytes<16, 16, 32, 32>() : win_lds_bytes<8, 16, 24, 32>();
#define ALLOW_LDS(KERNEL)                                                                                            \
        do {                                                                                                             \
            static bool lds_ok = false;                                                                                  \
            if (!lds_ok) {                                                                                               \
                const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL),                       \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);      \
                if (ae != hipSuccess) return semidetr::fail((int)ae, "msda_backward: hipFuncSetAttribute: %s", hipGetErrorString(ae)); \
                lds_ok = true;                                                                                           \
            }                                                                                                            \
        } while (0)
        if (big) {
            ALLOW_LDS((msda_bwd_scatter_d32_win<IO, 16, 16, 32, 32>));
            hipLaunchKernelGGL((msda_bwd_scatter_d32_win<IO, 16, 16, 32, 32>), dim3((unsigned)grid), dim3(kWinThreads),
                               wlds, st, grad_out, spatial_shapes, level_start, io, S, M, L, tiles_bound, grad_value);
        } else {
            ALLOW_LDS((msda_bwd_scatter_d32_win<IO, 8, 16, 24, 32>));
            hipLaunchKernelGGL((msda_bwd_scatter_d32_win<IO, 8, 16, 24, 32>), dim3((unsigned)grid), dim3(kWinThreads),
                               wlds, st, grad_out, spatial_shapes, level_start, io, S, M, L, tiles_bound, grad_value);
        }
        g_last_kernels = fill_in_gather ? "msda_bwd_gather_d32+msda_bwd_scatter_d32_win" : "fillBufferAligned+msda_bwd_gather_d32+msda_bwd_scatter_d32_win";
        return semidetr::launch_status("msda_bwd_scatter_d32_win");
    }
    // rows per workgroup: 32 normally, 8 when the problem is too small to fill 256 CUs with 32-row tiles
    int rpb = (int64_t)N * M * ((Lq + 31) / 32) >= 1024 ? 32 : 8;
    if (g_bwd_variant == 8 || g_bwd_variant == 32) rpb = g_bwd_variant % 100;
    if (g_bwd_variant == 808 || g_bwd_variant == 832) rpb = g_bwd_variant - 800;
    const int tiles = (Lq + rpb - 1) / rpb;
    const int64_t grid = (int64_t)N * tiles * M;
    SEMIDETR_REQUIRE(grid < INT32_MAX, SEMIDETR_E_TOOLARGE, "msda_backward: grid too large");
    const size_t lds = (size_t)rpb * (L * P + 1) * 32 + 2 * kMaxLevels * sizeof(float);
    return exp_launch_strips_backward<IO>(st, fill, grad_out, value, spatial_shapes, level_start, io, N, S, M, L, Lq, P, rpb,
                                      tiles, (unsigned)grid, lds, grad_value);
}


#undef g_fwd_variant
#undef g_bwd_variant
