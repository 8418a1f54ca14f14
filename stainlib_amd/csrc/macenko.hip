// macenko.hip -- host schedule + C ABI of the Macenko fit / transform (kernels: stats_kernels.hpp).
//
// One-launch-per-phase schedule (batches below kFusedMinTiles): tiles are processed in groups of up to 2 GiB of
// uint8 (bounds the workspace); per group 3 sweeps + 3 one-workgroup-per-tile finish kernels (+ the apply sweep
// for transform), all on the caller's stream, no host sync.  Larger batches run the persistent fused kernel.
#include <cassert>

#include "stats_kernels.hpp"
#include "sl_host.hpp"

using namespace sl;

namespace {

constexpr size_t kGroupBytes = (size_t)2 << 30;   // uint8 bytes of one tile group of the per-phase schedule (measured: big groups win --
                                                  // the one-workgroup-per-tile finish kernels need many tiles to fill the chip; cache reuse between sweeps does not matter)

// Defaults of SlParams.fused_min_tiles (documented in include/stainlib_hip.h): the measured crossovers of round 3's three-sweep fused
// kernel on an MI355X at its 1400 W power state (tools/crossover.py, profiles/r03_crossover.txt and r03_crossover_small.txt).  Below them one
// launch per phase wins.  1024^2 tiles: 1.11 vs 1.14 ms at 256, 1.64 vs 1.52 at 384.  Smaller tiles cross earlier (the per-phase launches
// of more than 256 tiles split into two groups' worth of finish launches): 512^2 0.42 vs 0.47 ms at 256 tiles, 0.58 vs 0.56 at 320, 0.64 vs
// 0.57 at 384; 256^2 0.235 vs 0.232 at 256, 0.35 vs 0.28 at 320; 700^2 0.78 vs 0.83 at 320, 0.87 vs 0.83 at 384.
// Round 4 (the fused kernel 10 % faster, the per-phase kernels as they were; profiles/r04_crossover*.txt): 1024^2 1.330 vs 1.371 ms at 320 tiles,
// 1.500 vs 1.401 at 384 -> 352; 700^2 0.815 vs 0.783 at 320 -> 320; 512^2 unchanged (0.386 vs 0.391 at 256, 0.548 vs 0.496 at 320).
constexpr int kFusedMinTiles = 352;         // tiles of 512 Ki pixels and more (round 3: 416)
constexpr int kFusedMinTilesMid = 320;      // 256 Ki < pixels < 512 Ki (round 3: 352)
constexpr int kFusedMinTilesSmall = 288;    // up to 256 Ki pixels
// The automatic split of a batch beyond the resident grid (plan_macenko): the remainder goes per phase when it is below these.  For small
// tiles a second, partly empty fused round overlaps the first one's tail and the split pays only for short remainders (512^2, 640 tiles:
// 0.91 split vs 0.97 fused, 768 tiles: 1.03 vs 0.98; 256^2: never).
constexpr int kSplitMaxRest = 208 /* round 4: 1024^2, 768 tiles 2.58 split vs 2.49 fused (remainder 256), 640 tiles 2.20 vs 2.37 (remainder 128); was 416 */, kSplitMaxRestMid = 224, kSplitMaxRestSmall = 192, kSplitMaxRestTiny = 0;   // tiny: up to 64 Ki pixels
inline int fused_min_default(long P) { return P >= (1L << 19) ? kFusedMinTiles : P > (1L << 18) ? kFusedMinTilesMid : kFusedMinTilesSmall; }
inline int split_max_rest(long P) {
    return P >= (1L << 19) ? kSplitMaxRest : P > (1L << 18) ? kSplitMaxRestMid : P > (1L << 16) ? kSplitMaxRestSmall : kSplitMaxRestTiny;
}
constexpr int kWideMinTiles = 192;          // (round 4, final tree: 192 tiles 0.788 vs 0.815 ms per phase, 160: 0.761 vs 0.709; was 208) Macenko, tiles of 256 Ki pixels and more: from here to #CU tiles the 1024-thread fused kernel (1024^2: 192 tiles 0.818 vs 0.798 ms per phase, 224: 0.858 vs 0.881, 256: 0.915 vs 0.974)
constexpr int kDictFusedMinTiles = 640;     // Vahadane: below it the dictionary sweeps run one launch per phase too (measured: 1024^2 tiles 3.70 vs
                                            // 3.85 ms at 512, 6.24 vs 5.76 at 768; in a fused launch of one tile per workgroup the few tiles
                                            // that need a third full sweep hold the whole launch, per phase they cost a short extra launch)
constexpr int kDictFusedMinTilesSmall = 192;   // ... for tiles below 512 Ki pixels (512^2: 0.82 vs 1.34 ms at 64 tiles, 1.53 vs 1.46 at 256)
constexpr int kDictFixedSweeps = 4;         // full sweeps launched after the sample stage; tiles that need more finish in k_dict_tail

#ifdef SL_DEVTOOLS                       // development build only (libstainlib_hip_dev.so, see tools/README.md): process-global knobs
unsigned g_debug_dyn_lds = 0;            // extra dynamic LDS per sweep workgroup (occupancy experiments)
long long* g_phase_clock = nullptr;      // see sl_debug_set_phase_clock
int g_debug_stop = 0;
#define SL_DYN_LDS g_debug_dyn_lds
#else
#define SL_DYN_LDS 0u
#endif

struct Layout {
    int parts, stride_log2, n_sample, sample_cap, G, cap_raw, cap_list, cap_ang;
    bool fused;
    bool wide;                              // fused with 1024-thread workgroups, one per CU (Macenko, batches of up to #CU tiles)
    int grid;                               // fused: workgroups launched
    int max_grid;                           // resident sweep workgroups of the device
    size_t off_M, off_maxC, off_status, off_partials, off_sample, off_cand, off_ang, off_list, off_state, off_dstate, off_mstate, off_diag, off_next, total;
};

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

Layout make_layout(int n, long P, int method = kMethodMacenko, int schedule = 0, int fused_min_tiles = 0) {
    Layout L;
    L.max_grid = max_resident_grid();
    L.parts = parts_for(P);
    {   // the persistent sweep kernels walk tiles x parts items: no more parts than it takes to give every workgroup ~4 items
        const long want = (4L * L.max_grid + n - 1) / (n > 0 ? n : 1);
        if (L.parts > want) L.parts = (int)(want < 1 ? 1 : want);
    }
    // The pixel sample steers the brackets (never the results): its SIZE sets how many pixels a bracket holds (the rank uncertainty
    // of a sample quantile goes with 1/sqrt(sample)), so Macenko tiles below 1 Mpixel are sampled more densely -- one pixel in 16 from
    // 256 Ki pixels down, up to the 16 Ki entries of a 1024^2 tile -- or their member lists overflow on real tissue (a soak over
    // 400 random tiles: 36 of 130 default-parameter tiles below 1 Mpixel lost the merged sweep to a full list at one pixel in 64).
    // Vahadane iterates over the sample: it keeps one in 64.
    L.stride_log2 = method == kMethodVahadane ? 6 : 4;
    while (((P + (1L << L.stride_log2) - 1) >> L.stride_log2) > kMaxSample) ++L.stride_log2;
    L.n_sample = (int)((P + (1L << L.stride_log2) - 1) >> L.stride_log2);
    assert(L.stride_log2 >= 4 && L.n_sample <= kMaxSample);     // sample_row / sample_pixel: cps_log2 = stride_log2 - 2 >= 2
    // the fused Macenko kernel's two-sweep schedule gathers a cluster sample into the same buffer (stats_twosweep.hpp)
    L.sample_cap = L.n_sample;
    if (method == kMethodMacenko && cluster_samples(P) > L.sample_cap) L.sample_cap = cluster_samples(P);
    assert(L.sample_cap <= kMaxSample);
    long g = (long)(kGroupBytes / (size_t)(3 * P));
    const long min_g = (1024 + L.parts - 1) / L.parts;      // keep >= ~1024 workgroups per sweep launch
    if (g < min_g) g = min_g;
    if (g > n) g = n;
    if (g < 1) g = 1;
    L.G = (int)g;
    const int min_fused = fused_min_tiles > 0 ? fused_min_tiles        // SlParams.fused_min_tiles; the defaults are the measured crossovers
                        : method == kMethodVahadane ? (P >= (1L << 19) ? kDictFusedMinTiles : kDictFusedMinTilesSmall)
                                                    : fused_min_default(P);
    L.fused = (schedule == 2) || (schedule != 1 && n >= min_fused);
    // Vahadane just beyond one resident grid: the second, mostly empty round of the persistent kernel costs more than one launch per phase
    // for the whole batch (1024^2: 640 tiles 5.65 vs 4.71 ms, 768 tiles 5.71 vs 6.20; 512^2: 576 tiles 1.93 vs 1.63, 768 tiles 1.98 vs 2.09)
    if (schedule == 0 && fused_min_tiles <= 0 && method == kMethodVahadane && P >= (1L << 18) && n > L.max_grid && 8L * n < 11L * L.max_grid)
        L.fused = false;
    // Macenko batches of no more tiles than CUs: the fused kernel with ONE 1024-thread workgroup per CU (schedule 3 forces it where it
    // fits; automatic from kWideMinTiles tiles of 256 Ki pixels and more -- below that one launch per phase fills the chip better)
    const int n_cu = L.max_grid / 2;
    L.wide = method == kMethodMacenko && n <= n_cu && (schedule == 3 || (schedule == 0 && fused_min_tiles <= 0 && n >= kWideMinTiles && P >= (1L << 18)));
    if (schedule == 3) L.fused = true;
    if (L.wide) L.fused = true;
    L.grid = n < L.max_grid ? n : L.max_grid;
    const size_t slots = L.fused ? (size_t)L.grid : (size_t)L.G;     // candidate buffers: per workgroup / per tile of a group
    size_t o = 0;
    L.off_M = o;        o = align_up(o + sizeof(double) * 6 * (size_t)n);
    L.off_maxC = o;     o = align_up(o + sizeof(double) * 2 * (size_t)n);
    L.off_status = o;   o = align_up(o + sizeof(int32_t) * (size_t)n);
    L.off_diag = o;     o = align_up(o + sizeof(int32_t) * (size_t)n);
    L.off_partials = o; o = align_up(o + sizeof(double) * 32 * (size_t)L.parts * L.G);     // 10 (Macenko) / 32 (Vahadane) per item
    L.off_sample = o;   o = align_up(o + sizeof(uint32_t) * (size_t)L.sample_cap * slots);
    // list capacities scale with the tile.  i.i.d. tiles: ~6 % of the pixels are raw candidates of the merged selection sweep (angle
    // ~2.5 %, concentrations ~4 %) and ~4.5 % end up in a bracket.  Real tissue fills them further -- the concentration brackets the
    // merged sweep widens by the box of stain matrices hold up to 8 % of the pixels of a stained-tissue tile with background, the raw
    // list 11 % (tools/merged_diag.py on windows of the ihc fixture) -- and a full list costs the tile its separate sweep: 1/6 and 1/8
    // of the pixels (round 3's first cut had 1/8 and 1/12: 13 of 400 soak tiles with default parameters still overflowed).
    L.cap_raw = (int)(P / 6 > kMinCapRaw ? P / 6 : kMinCapRaw);
    L.cap_list = (int)(P / 8 > kMinCapList ? P / 8 : kMinCapList);
    L.off_cand = o;     o = align_up(o + sizeof(uint32_t) * (size_t)L.cap_raw * slots);
    // the fused Macenko kernel's sweep behind the colour-cube mask keeps the angular candidates in a list of their own (RawDirect): ~2.5 %
    // of an i.i.d. tile at the default percentile, the two tails of 2 alpha % of the tissue on any tile -- but ALL of the tissue when a small sample leaves the brackets open: the raw list's own room, so that no tile loses the fast path to the split
    L.cap_ang = (L.fused && method == kMethodMacenko) ? L.cap_raw : 0;
    L.off_ang = o;      o = align_up(o + sizeof(uint32_t) * (size_t)L.cap_ang * slots);
    L.off_list = o;     o = align_up(o + sizeof(float) * 2 * (size_t)L.cap_list * slots);
    L.off_state = o;    o = align_up(o + sizeof(TileState) * (size_t)L.G);
    L.off_dstate = o;   o = align_up(o + sizeof(DictState) * (size_t)L.G);
    L.off_mstate = o;   o = align_up(o + sizeof(TileMerged) * (size_t)L.G);
    L.off_next = o;     o = align_up(o + sizeof(unsigned long long));      // the fused kernel's tile counter
    L.total = o;
    return L;
}

StatsArgs stats_args(const uint8_t* rgb, int g0, int m, long P, const SlParams& p, const Layout& L, char* ws) {
    StatsArgs a;
    a.rgb = rgb + (size_t)g0 * 3 * P;
    a.P = (int)P;
    a.parts = L.parts;
    a.n_items = m * L.parts;
    a.stride_log2 = L.stride_log2;
    a.n_sample = L.n_sample;
    a.ylimf = (float)y_limit_for_threshold(p.luminosity_threshold) - 2048.0f;   // exact: y_lim < 2^24
    a.lam = p.lasso_lambda;
    a.pct = p.angular_percentile;
    a.partials = (double*)(ws + L.off_partials);
    a.sample = (uint32_t*)(ws + L.off_sample);
    a.cap_raw = L.cap_raw;
    a.cap_list = L.cap_list;
    a.raw = (uint32_t*)(ws + L.off_cand);
    a.cand = (float*)(ws + L.off_list);
    a.state = (TileState*)(ws + L.off_state);
    a.dl_lambda = p.dl_lambda;
    a.dl_tol = p.dl_tol;
    a.dl_max_sweeps = p.dl_max_sweeps > 0 ? p.dl_max_sweeps : 1;
    a.tile0 = g0;
    a.dstate = (DictState*)(ws + L.off_dstate);
    a.sweeps_out = nullptr;
    a.mstate = nullptr;                      // set by the merged Macenko schedule only (the Vahadane groups reuse k_select<conc> / k_finish_conc)
    return a;
}

// Runs the statistics stages for tiles [g0, g0+m) of the batch; results land in M_all/maxC_all/status_all.
int run_stats_group(const uint8_t* rgb, int g0, int m, long P, const SlParams& p, const Layout& L, char* ws,
                    double* M_all, double* maxC_all, int32_t* status_all, hipStream_t s) {
    StatsArgs a = stats_args(rgb, g0, m, P, p, L, ws);
    const bool al = aligned4(a.rgb, P);
    const long items = (long)m * L.parts;
    const dim3 gs((unsigned)(items < L.max_grid ? items : L.max_grid)), bs(kSweepThreads), gf((unsigned)m), bf(kFinishThreads);
    SlProfile* prof = p.profile;
    a.mstate = (TileMerged*)(ws + L.off_mstate);
    const dim3 bm(kMFinishThreads);
    {
        ProfScope ps(prof, SL_PROF_MOMENTS, m, s);
        if (al) hipLaunchKernelGGL((k_moments<true>), gs, bs, SL_DYN_LDS, s, a);
        else    hipLaunchKernelGGL((k_moments<false>), gs, bs, SL_DYN_LDS, s, a);
    }
    { ProfScope ps(prof, SL_PROF_FINISH, m, s); hipLaunchKernelGGL(k_finish1m, gf, bm, 0, s, a); }
    {   // ONE selection sweep for the angular and the concentration candidates (as sweep 2 of the fused kernel)
        ProfScope ps(prof, SL_PROF_SELECT_ANGLE, m, s);
        if (al) hipLaunchKernelGGL((k_select<kStageMerged, true>), gs, bs, SL_DYN_LDS, s, a);
        else    hipLaunchKernelGGL((k_select<kStageMerged, false>), gs, bs, SL_DYN_LDS, s, a);
    }
    {
        ProfScope ps(prof, SL_PROF_FINISH, m, s);
        hipLaunchKernelGGL(k_finish2m, gf, bm, 0, s, a, M_all, maxC_all, status_all, p.fallbacks_out, g0);
    }
    {   // only the tiles whose exact stain matrix left the box the merged sweep assumed still have work here (none, normally)
        ProfScope ps(prof, SL_PROF_SELECT_CONC, m, s);
        if (al) hipLaunchKernelGGL((k_select<kStageConc, true>), gs, bs, SL_DYN_LDS, s, a);
        else    hipLaunchKernelGGL((k_select<kStageConc, false>), gs, bs, SL_DYN_LDS, s, a);
    }
    {
        ProfScope ps(prof, SL_PROF_FINISH, m, s);
        hipLaunchKernelGGL(k_finish_conc, gf, bf, 0, s, a, M_all, maxC_all, status_all, p.fallbacks_out, g0);
    }
    return launch_status();
}

// The same for Vahadane: dictionary sweeps, then the concentration stage of the Macenko schedule.
int run_dict_group(const uint8_t* rgb, int g0, int m, long P, const SlParams& p, const Layout& L, char* ws,
                   double* M_all, double* maxC_all, int32_t* status_all, int32_t* sweeps_out, hipStream_t s) {
    StatsArgs a = stats_args(rgb, g0, m, P, p, L, ws);
    a.sweeps_out = sweeps_out;
    const bool al = aligned4(a.rgb, P);
    const long items = (long)m * L.parts;
    const dim3 gs((unsigned)(items < L.max_grid ? items : L.max_grid)), bs(kSweepThreads), gf((unsigned)m), bf(kFinishThreads), bd(kDictFinishThreads);
    SlProfile* prof = p.profile;
    {   // the sample, gathered without a sweep, and the dictionary iterated on it
        ProfScope ps(prof, SL_PROF_FINISH, m, s);
        if (al) hipLaunchKernelGGL((k_dict_start<true>), gf, bd, 0, s, a);
        else    hipLaunchKernelGGL((k_dict_start<false>), gf, bd, 0, s, a);
    }
    const int fixed = a.dl_max_sweeps < kDictFixedSweeps ? a.dl_max_sweeps : kDictFixedSweeps;
    for (int i = 0; i < fixed; ++i) {
        {
            ProfScope ps(prof, SL_PROF_DICT, m, s);
            if (al) hipLaunchKernelGGL((k_dict<true>), gs, bs, 0, s, a);
            else    hipLaunchKernelGGL((k_dict<false>), gs, bs, 0, s, a);
        }
        { ProfScope ps(prof, SL_PROF_FINISH, m, s); hipLaunchKernelGGL(k_dict_finish, gf, bd, 0, s, a); }
    }
    {
        ProfScope ps(prof, SL_PROF_FINISH, m, s);
        if (al) hipLaunchKernelGGL((k_dict_tail<true>), gf, bd, 0, s, a);
        else    hipLaunchKernelGGL((k_dict_tail<false>), gf, bd, 0, s, a);
    }
    {
        ProfScope ps(prof, SL_PROF_SELECT_CONC, m, s);
        if (al) hipLaunchKernelGGL((k_select<kStageConc, true>), gs, bs, 0, s, a);
        else    hipLaunchKernelGGL((k_select<kStageConc, false>), gs, bs, 0, s, a);
    }
    {
        ProfScope ps(prof, SL_PROF_FINISH, m, s);
        hipLaunchKernelGGL(k_finish_conc, gf, bf, 0, s, a, M_all, maxC_all, status_all, p.fallbacks_out, g0);
    }
    return launch_status();
}

// The persistent schedule: one launch for the whole batch (fit only when out == nullptr).
int run_fused(int method, const uint8_t* rgb, uint8_t* out, int n, long P, const SlParams& p, const Layout& L, char* ws,
              const double* M_tgt, const double* maxC_tgt, double* M_all, double* maxC_all, int32_t* status_all,
              int32_t* sweeps_out, hipStream_t s) {
    FusedArgs a;
    a.rgb = rgb;
    a.out = out;
    a.n_tiles = n;
    a.P = (int)P;
    a.stride_log2 = L.stride_log2;
    a.n_sample = L.n_sample;
    a.ylimf = (float)y_limit_for_threshold(p.luminosity_threshold) - 2048.0f;   // exact: y_lim < 2^24
    a.lam = p.lasso_lambda;
    a.pct = p.angular_percentile;
    a.M_tgt = M_tgt;
    a.maxC_tgt = maxC_tgt;
    a.cap_raw = L.cap_raw;
    a.cap_list = L.cap_list;
    a.raw = (uint32_t*)(ws + L.off_cand);
    a.cap_ang = L.cap_ang;
    a.raw_ang = (uint32_t*)(ws + L.off_ang);
    a.cand = (float*)(ws + L.off_list);
    a.sample = (uint32_t*)(ws + L.off_sample);
    a.M_out = M_all;
    a.maxC_out = maxC_all;
    a.status_out = status_all;
    a.diag_out = p.fallbacks_out ? p.fallbacks_out : (int32_t*)(ws + L.off_diag);
    a.resweep_out = p.resweeps_out;
#ifdef SL_DEVTOOLS
    a.phase_clock = g_phase_clock;
    a.debug_stop = g_debug_stop;
#endif
    a.dl_lambda = p.dl_lambda;
    a.dl_tol = p.dl_tol;
    a.dl_max_sweeps = p.dl_max_sweeps > 0 ? p.dl_max_sweeps : 1;
    a.sweeps_out = sweeps_out;
    a.use_cube = p.prefilter == 1 ? 0 : (p.prefilter == 2 ? 2 : 1);
    a.cube_out = p.prefilter_out;
    a.sample_cap = L.sample_cap;
    // (tiles below 16 Ki pixels keep the three-sweep schedule: their sample would be a fifth of the tile)
    // (... and SlParams.prefilter = 1, "never behind the colour-cube mask", rules the merged sweep out: it has no per-pixel form)
    a.two_sweep = (method != kMethodMacenko || p.two_sweep == 1 || P < (1L << 14)) ? 0 : (p.two_sweep >= 2 && p.two_sweep <= 4 ? p.two_sweep : (p.prefilter == 1 ? 0 : 1));
    a.cl_lines = cluster_lines(P);
    a.cl_scale_log2 = 1;
    while (((long)a.cl_lines * kClusterPx << a.cl_scale_log2) < P && a.cl_scale_log2 < 30) ++a.cl_scale_log2;
    a.ts_out = p.twosweep_out;
    a.next_tile = (unsigned long long*)(ws + L.off_next);
    if (n > L.grid) zero_async(a.next_tile, sizeof(unsigned long long), s);
    const bool al = aligned4(rgb, P) && (!out || aligned4(out, P));
    ProfScope ps(p.profile, out ? SL_PROF_FUSED_TRANSFORM : SL_PROF_FUSED_FIT, n, s);
    // (the twelve instantiations of k_fused live in three translation units of their own -- fused_macenko.hip, fused_macenko_wide.hip,
    //  fused_vahadane.hip -- so that they compile side by side)
    if (method == kMethodMacenko && L.wide) launch_fused_macenko_wide(a, out != nullptr, al, (unsigned)L.grid, s);
    else if (method == kMethodMacenko) launch_fused_macenko(a, out != nullptr, al, (unsigned)L.grid, s);
    else launch_fused_vahadane(a, out != nullptr, al, (unsigned)L.grid, s);
    return launch_status();
}

int check_common(const void* rgb, int n, int h, int w, const void* ws, size_t ws_bytes, size_t need) {
    if (!rgb || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    if ((long)h * w > (1L << 30)) return SL_ERR_BADARG;
    if (!ws || ws_bytes < need) return SL_ERR_WORKSPACE;
    if (((uintptr_t)ws & 255u) != 0) return SL_ERR_WORKSPACE;
    return SL_OK;
}

// What a Macenko call does with its n tiles.  Automatic schedule only: a batch larger than the resident grid of the fused kernel
// whose last round would be mostly empty (n mod grid below the crossover) gives that remainder to the one-launch-per-phase
// schedule instead -- 640 tiles: one full fused round + 128 tiles per phase, 2.3 ms, where two fused rounds take 2.75 ms.  Results
// do not depend on the split (both schedules select the same values).
struct MacenkoPlan {
    Layout L;            // the whole batch (result arrays) and the fused part
    bool mixed;
    int n_fused;         // tiles [0, n_fused) fused, [n_fused, n) per phase (mixed only)
    Layout Lp;           // the per-phase remainder, placed behind L in the workspace
    size_t total;
};
MacenkoPlan plan_macenko(int n, long P, int schedule, int fused_min_tiles) {
    MacenkoPlan pl;
    pl.L = make_layout(n, P, kMethodMacenko, schedule, fused_min_tiles);
    pl.mixed = false;
    pl.n_fused = 0;
    pl.Lp = Layout{};
    pl.total = pl.L.total;
    if (schedule == 0 && pl.L.fused && n > pl.L.max_grid) {
        const int rest = n % pl.L.max_grid;
        const int max_rest = fused_min_tiles > 0 ? fused_min_tiles : split_max_rest(P);
        if (rest > 0 && rest < max_rest) {
            pl.mixed = true;
            pl.n_fused = n - rest;
            pl.Lp = make_layout(rest, P, kMethodMacenko, 1, 0);
            pl.total = pl.L.total + pl.Lp.total;
        }
    }
    return pl;
}

// fit (out == nullptr) or transform of a batch according to its plan
int run_macenko(const MacenkoPlan& pl, const uint8_t* rgb, uint8_t* out, int n, int h, int w, const SlParams& p, const double* M_tgt,
                const double* maxC_tgt, double* M_all, double* maxC_all, int32_t* st_all, char* ws, void* stream) {
    const long P = (long)h * w;
    const Layout& L = pl.L;
    int rc;
    if (L.fused && !pl.mixed)
        return run_fused(kMethodMacenko, rgb, out, n, P, p, L, ws, M_tgt, maxC_tgt, M_all, maxC_all, st_all, nullptr, (hipStream_t)stream);
    int first = 0;
    const Layout* Lg = &L;
    char* wsg = ws;
    if (pl.mixed) {
        rc = run_fused(kMethodMacenko, rgb, out, pl.n_fused, P, p, L, ws, M_tgt, maxC_tgt, M_all, maxC_all, st_all, nullptr, (hipStream_t)stream);
        if (rc) return rc;
        first = pl.n_fused;
        Lg = &pl.Lp;
        wsg = ws + L.total;
    }
    for (int g0 = first; g0 < n; g0 += Lg->G) {
        const int m = (n - g0) < Lg->G ? (n - g0) : Lg->G;
        rc = run_stats_group(rgb, g0, m, P, p, *Lg, wsg, M_all, maxC_all, st_all, (hipStream_t)stream);
        if (rc) return rc;
        if (out) {
            ProfScope ps(p.profile, SL_PROF_APPLY, m, (hipStream_t)stream);
            rc = sl_normalize_apply(rgb + (size_t)g0 * 3 * P, out + (size_t)g0 * 3 * P, m, h, w, M_all + 6 * (size_t)g0, maxC_all + 2 * (size_t)g0,
                                    M_tgt, maxC_tgt, p.lasso_lambda, nullptr, stream);
            if (rc) return rc;
        }
    }
    return SL_OK;
}

}  // namespace

extern "C" size_t sl_workspace_bytes(int op, int n_tiles, int h, int w) {
    if (n_tiles <= 0 || h <= 0 || w <= 0) return 0;
    switch (op) {
        case SL_OP_MACENKO_FIT:
        case SL_OP_MACENKO_TRANSFORM:
        {   // SlParams.schedule may force either schedule, and the automatic one may split the batch: size for the largest of the three
            // (a caller-set SlParams.fused_min_tiles can only move tiles between plans that are all covered: the split's remainder is
            //  sized as a per-phase batch of at most one resident grid)
            const long P = (long)h * w;
            const size_t a = make_layout(n_tiles, P, kMethodMacenko, 1).total, b = make_layout(n_tiles, P, kMethodMacenko, 2).total;
            size_t c = plan_macenko(n_tiles, P, 0, 0).total;
            const int mg = max_resident_grid();
            if (n_tiles > mg && n_tiles % mg) {                   // the largest remainder plan any fused_min_tiles could choose
                const size_t d = b + make_layout(n_tiles % mg, P, kMethodMacenko, 1).total;
                c = c > d ? c : d;
            }
            return (a > b ? a : b) > c ? (a > b ? a : b) : c;
        }
        case SL_OP_VAHADANE_FIT:
        case SL_OP_VAHADANE_TRANSFORM:
        {
            const size_t a = make_layout(n_tiles, (long)h * w, kMethodVahadane, 1).total, b = make_layout(n_tiles, (long)h * w, kMethodVahadane, 2).total;
            return a > b ? a : b;
        }
        case SL_OP_HED_AUGMENT:
            return (sizeof(unsigned long long) * (size_t)n_tiles + 255) & ~(size_t)255;
        case SL_OP_LAB_STATS:
            return lab_workspace_bytes(n_tiles);
        case SL_OP_TILE_MOMENTS:
            return (sizeof(double) * 10 * (size_t)parts_for((long)h * w) * (size_t)n_tiles + 255) & ~(size_t)255;
        default:
            return 0;
    }
}

// What the call these SlParams select needs: one plan, not the maximum over all of them.
extern "C" size_t sl_workspace_bytes_for(int op, int n_tiles, int h, int w, const SlParams* params) {
    if (n_tiles <= 0 || h <= 0 || w <= 0 || !params_ok(params)) return 0;
    const long P = (long)h * w;
    const int schedule = params ? params->schedule : 0, fmin = params ? params->fused_min_tiles : 0;
    switch (op) {
        case SL_OP_MACENKO_FIT:
        case SL_OP_MACENKO_TRANSFORM:
            return plan_macenko(n_tiles, P, schedule, fmin).total;
        case SL_OP_VAHADANE_FIT:
        case SL_OP_VAHADANE_TRANSFORM:
            return make_layout(n_tiles, P, kMethodVahadane, schedule, fmin).total;
        default:
            return sl_workspace_bytes(op, n_tiles, h, w);
    }
}

extern "C" int sl_macenko_fit(const uint8_t* rgb, int n, int h, int w, const SlParams* params, double* M_out,
                              double* maxC_out, int32_t* status, void* workspace, size_t workspace_bytes,
                              void* stream) {
    if (!params_ok(params)) return SL_ERR_BADARG;
    const long P = (long)h * w;
    const MacenkoPlan pl = (n > 0 && h > 0 && w > 0) ? plan_macenko(n, P, params ? params->schedule : 0, params ? params->fused_min_tiles : 0) : MacenkoPlan{};
    // this plan's own need: sl_workspace_bytes_for(op, n, h, w, params) (sl_workspace_bytes(), the maximum over every SlParams, always suffices)
    int rc = check_common(rgb, n, h, w, workspace, workspace_bytes, pl.total);
    if (rc) return rc;
    SlParams p;
    sl_default_params(&p);
    if (params) p = *params;
    char* ws = (char*)workspace;
    double* M_all = M_out ? M_out : (double*)(ws + pl.L.off_M);
    double* maxC_all = maxC_out ? maxC_out : (double*)(ws + pl.L.off_maxC);
    int32_t* st_all = status ? status : (int32_t*)(ws + pl.L.off_status);
    return run_macenko(pl, rgb, nullptr, n, h, w, p, nullptr, nullptr, M_all, maxC_all, st_all, ws, stream);
}

extern "C" int sl_macenko_transform(const uint8_t* rgb, uint8_t* out, int n, int h, int w, const SlParams* params,
                                    const double* M_tgt, const double* maxC_tgt, double* M_src_out,
                                    double* maxC_src_out, int32_t* status, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    if (!params_ok(params)) return SL_ERR_BADARG;
    const long P = (long)h * w;
    const MacenkoPlan pl = (n > 0 && h > 0 && w > 0) ? plan_macenko(n, P, params ? params->schedule : 0, params ? params->fused_min_tiles : 0) : MacenkoPlan{};
    // this plan's own need: sl_workspace_bytes_for(op, n, h, w, params) (sl_workspace_bytes(), the maximum over every SlParams, always suffices)
    int rc = check_common(rgb, n, h, w, workspace, workspace_bytes, pl.total);
    if (rc) return rc;
    if (!out || !M_tgt || !maxC_tgt) return SL_ERR_BADARG;
    SlParams p;
    sl_default_params(&p);
    if (params) p = *params;
    char* ws = (char*)workspace;
    double* M_all = M_src_out ? M_src_out : (double*)(ws + pl.L.off_M);
    double* maxC_all = maxC_src_out ? maxC_src_out : (double*)(ws + pl.L.off_maxC);
    int32_t* st_all = status ? status : (int32_t*)(ws + pl.L.off_status);
    return run_macenko(pl, rgb, out, n, h, w, p, M_tgt, maxC_tgt, M_all, maxC_all, st_all, ws, stream);
}

// Vahadane: the persistent kernel for large batches, one launch per phase below kDictFusedMinTiles tiles.
extern "C" int sl_vahadane_fit(const uint8_t* rgb, int n, int h, int w, const SlParams* params, double* M_out,
                               double* maxC_out, int32_t* status, int32_t* sweeps_out, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!params_ok(params)) return SL_ERR_BADARG;
    const long P = (long)h * w;
    const Layout L = (n > 0 && h > 0 && w > 0) ? make_layout(n, P, kMethodVahadane, params ? params->schedule : 0, params ? params->fused_min_tiles : 0) : Layout{};
    int rc = check_common(rgb, n, h, w, workspace, workspace_bytes, L.total);
    if (rc) return rc;
    SlParams p;
    sl_default_params(&p);
    if (params) p = *params;
    char* ws = (char*)workspace;
    double* M_all = M_out ? M_out : (double*)(ws + L.off_M);
    double* maxC_all = maxC_out ? maxC_out : (double*)(ws + L.off_maxC);
    int32_t* st_all = status ? status : (int32_t*)(ws + L.off_status);
    if (L.fused)
        return run_fused(kMethodVahadane, rgb, nullptr, n, P, p, L, ws, nullptr, nullptr, M_all, maxC_all, st_all, sweeps_out,
                         (hipStream_t)stream);
    for (int g0 = 0; g0 < n; g0 += L.G) {
        const int m = (n - g0) < L.G ? (n - g0) : L.G;
        rc = run_dict_group(rgb, g0, m, P, p, L, ws, M_all, maxC_all, st_all, sweeps_out, (hipStream_t)stream);
        if (rc) return rc;
    }
    return SL_OK;
}

extern "C" int sl_vahadane_transform(const uint8_t* rgb, uint8_t* out, int n, int h, int w, const SlParams* params,
                                     const double* M_tgt, const double* maxC_tgt, double* M_src_out,
                                     double* maxC_src_out, int32_t* status, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (!params_ok(params)) return SL_ERR_BADARG;
    const long P = (long)h * w;
    const Layout L = (n > 0 && h > 0 && w > 0) ? make_layout(n, P, kMethodVahadane, params ? params->schedule : 0, params ? params->fused_min_tiles : 0) : Layout{};
    int rc = check_common(rgb, n, h, w, workspace, workspace_bytes, L.total);
    if (rc) return rc;
    if (!out || !M_tgt || !maxC_tgt) return SL_ERR_BADARG;
    SlParams p;
    sl_default_params(&p);
    if (params) p = *params;
    char* ws = (char*)workspace;
    double* M_all = M_src_out ? M_src_out : (double*)(ws + L.off_M);
    double* maxC_all = maxC_src_out ? maxC_src_out : (double*)(ws + L.off_maxC);
    int32_t* st_all = status ? status : (int32_t*)(ws + L.off_status);
    if (L.fused)
        return run_fused(kMethodVahadane, rgb, out, n, P, p, L, ws, M_tgt, maxC_tgt, M_all, maxC_all, st_all, nullptr,
                         (hipStream_t)stream);
    for (int g0 = 0; g0 < n; g0 += L.G) {
        const int m = (n - g0) < L.G ? (n - g0) : L.G;
        rc = run_dict_group(rgb, g0, m, P, p, L, ws, M_all, maxC_all, st_all, nullptr, (hipStream_t)stream);
        if (rc) return rc;
        {
            ProfScope ps(p.profile, SL_PROF_APPLY, m, (hipStream_t)stream);
            rc = sl_normalize_apply(rgb + (size_t)g0 * 3 * P, out + (size_t)g0 * 3 * P, m, h, w,
                                    M_all + 6 * (size_t)g0, maxC_all + 2 * (size_t)g0, M_tgt, maxC_tgt,
                                    p.lasso_lambda, nullptr, stream);
        }
        if (rc) return rc;
    }
    return SL_OK;
}

#ifdef SL_DEVTOOLS
// Development aids, compiled ONLY into libstainlib_hip_dev.so (make dev): not part of the public header, backed by
// process-global state, never in the product library (tests/test_host_api.py checks the export list).
// where the per-tile state lives in the workspace
extern "C" SL_API int sl_debug_layout(int n, int h, int w, size_t* off_state, size_t* sizeof_state, int* group,
                               size_t* off_diag, int* fused) {
    const Layout L = make_layout(n, (long)h * w);
    if (off_state) *off_state = L.off_state;
    if (sizeof_state) *sizeof_state = sizeof(TileState);
    if (group) *group = L.G;
    if (off_diag) *off_diag = L.off_diag;
    if (fused) *fused = L.fused ? 1 : 0;
    return SL_OK;
}

extern "C" SL_API void sl_debug_set_phase_clock(long long* device_buf) { g_phase_clock = device_buf; }
extern "C" SL_API void sl_debug_set_stop(int phase) { g_debug_stop = phase; }
extern "C" SL_API void sl_debug_set_dyn_lds(unsigned bytes) { g_debug_dyn_lds = bytes; }


#ifdef SL_DEBUG_SUBCLK
extern "C" SL_API void sl_debug_bclk(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(sl::g_bclk), 128);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(sl::g_bclk), z, 128); }
}
#endif
#endif  // SL_DEVTOOLS
