// stats_fused.hpp -- the merged finish steps (fused_finish1/2), their per-phase kernels k_finish1m/2m, and the persistent k_fused.
// Part of stats_kernels.hpp (split by phase in round 4, no functional change); include that umbrella, not this file.
#pragma once
#include "stats_phase_kernels.hpp"
#include "stats_cube.hpp"
#include "stats_twosweep.hpp"

namespace sl {

// ------------------------------------------------------------------------------------------
// fused persistent schedule: one workgroup = one tile at a time, all phases
// ------------------------------------------------------------------------------------------
struct FusedArgs {
    const uint8_t* rgb;
    uint8_t* out;            // transform only
    int n_tiles;
    int P;
    int stride_log2;
    int n_sample;
    float ylimf;
    double lam;
    double pct;
    const double* M_tgt;     // transform only
    const double* maxC_tgt;  // transform only
    int cap_raw, cap_list;
    uint32_t* raw;           // [gridDim.x][cap_raw]
    int cap_ang;             // Macenko: capacity of the angular candidate list the cube sweep fills beside `raw` (RawDirect)
    uint32_t* raw_ang;       // [gridDim.x][cap_ang]
    float* cand;             // [gridDim.x][2][cap_list]
    uint32_t* sample;        // [gridDim.x][n_sample]
    double* M_out;           // [n_tiles][6]
    double* maxC_out;        // [n_tiles][2]
    int32_t* status_out;     // [n_tiles]
    int32_t* diag_out;       // [n_tiles] fallbacks (may be NULL)
    int32_t* resweep_out;    // [n_tiles] 1 when the tile needed the separate concentration sweep (may be NULL)
#ifdef SL_DEVTOOLS
    long long* phase_clock;  // [n_tiles][8] wall_clock64() at phase boundaries (development build only, may be NULL)
    int debug_stop;          // development build only: leave the tile after phase marker debug_stop-1 (0 = run everything)
#endif
    // Vahadane
    double dl_lambda;
    double dl_tol;
    int dl_max_sweeps;
    int32_t* sweeps_out;     // [n_tiles] (may be NULL)
    int use_cube;            // Macenko: let finish 1 build the colour-cube mask for sweep 2 (stats_cube.hpp); 0 = always the per-pixel sweep
    int32_t* cube_out;       // [n_tiles] 1 where sweep 2 ran behind the cube mask (diagnostics, may be NULL)
    // Macenko, two-sweep schedule (stats_twosweep.hpp)
    int two_sweep;           // 0 = off (three sweeps, the sample rides in sweep 1); 1 = on where phase 0 says it pays; 2 = on wherever an estimate
                             // exists (tests); 3 = as 2, and the plane check is made to fail (tests: drives the fallback); 4 = as 2 with the
                             // sample's plane tilted by 0.05 before use (tests: a check that fails on its own)
    int sample_cap;          // entries of a workgroup's sample buffer (>= n_sample and >= the cluster sample)
    int cl_lines;            // lines of the cluster sample (cluster_lines(P)); its entries: cl_lines * kClusterPx <= the sample buffer
    int cl_scale_log2;       // log2 of the pixels one entry of the cluster sample stands for (rounded up, >= 1)
    int32_t* ts_out;         // [n_tiles] kTs* (diagnostics, may be NULL)
    unsigned long long* next_tile;    // one word, zero before a launch of more tiles than workgroups: hands out the tiles beyond the first round
                                      // (NULL: a workgroup's tiles are blockIdx.x + k gridDim.x)
};

template <int NT>
struct FusedShared {
    RowTab tab;              // 64 KB, first member: LDS offset 0
    uint32_t stage[NT / 64][kStageWave];     // 1 KB per wave
    SelScratch S;            // 4 KB aligned (LDS offset 72 KB): between the finish steps S.hist holds the colour-cube mask, whose base the
                             // cube sweep ORs into its addresses
    alignas(8) unsigned int n_raw;     // head of the tile's raw list ...
    unsigned int n_ang;                // ... and, right behind it, of the angular list of the cube sweep: RawDirect advances both with one 64-bit atomic
    unsigned int overflow;
    double red[NT / 64][32];
    double sum[32];
    DictIter it;
    double Vd[6];
    double M[6];
    double maxC[2];
    float Vf[6];
    float lo[2], hi[2];
    float box[4];            // where the merged sweep assumes the two percentile angles (angle_brackets)
    float res[4];
    LassoK L;
    int status;
    int conc_done;           // the merged sweep's candidates settled maxC: sweep 3 is skipped
    float xmin;              // tissue_x_bound of the tile (merged sweep)
    int why;                 // why the merged sweep's concentration candidates were not used (SL_RESWEEP_*; 0 = they were)
    int use_cube;            // finish 1 built the colour-cube mask (in S.hist) and the sample says it pays: sweep 2 = select_sweep_cube
    MergedConc mk;
    TwoSweep ts;             // two-sweep schedule: the estimate phase 0 left for sweep 1 (ts.ok) and what the finish must verify
    int next_tile;           // the workgroup's next tile (k_fused)
};

// Every kernel that owns a FusedShared block declares it as its ONLY __shared__ object, so the block starts at LDS address 0
// (fused_lds_origin traps otherwise, once per workgroup).  The out-of-line phases receive the block as a generic pointer, from which
// the compiler cannot know any member's LDS address: every table gather then pays a v_add for base + offset -- 3 per pixel in sweep 1
// and in the apply sweep, measured in the ISA after the phases went out of line.  They use the constants below instead.
template <int NT>
struct FusedLds {
    static constexpr uint32_t tab = 0u;
    static constexpr uint32_t stage = (uint32_t)sizeof(RowTab);
    static constexpr uint32_t stage_wave = 4u * (uint32_t)kStageWave;
    static constexpr uint32_t hist = (uint32_t)(sizeof(RowTab) + sizeof(uint32_t) * (NT / 64) * kStageWave);
};
template <int NT>
__device__ __forceinline__ void fused_lds_origin(const FusedShared<NT>& sh) {
    static_assert(offsetof(FusedShared<NT>, tab) == FusedLds<NT>::tab && offsetof(FusedShared<NT>, stage) == FusedLds<NT>::stage &&
                  offsetof(FusedShared<NT>, S) == FusedLds<NT>::hist && offsetof(SelScratch, hist) == 0, "");
    if (lds_address(&sh) != 0u) __builtin_trap();
}

// (the finish steps inlined into the kernel again: measured +3.7 %)
#define SL_FINISH_ATTR __noinline__
// wave-uniform values arrive in VGPRs at an out-of-line function: back to SGPRs
template <class T>
__device__ __forceinline__ T* uni_ptr(T* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (T*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double uni_d(double x) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The colour-cube mask of the merged sweep (stats_cube.hpp) for the tile in *shp, after finish 1 has left brackets and thresholds there:
// per-channel tables in the sweeps' staging space, the mask where the finish steps' histogram lives (neither is in use between finish
// 1 and finish 2); sh.use_cube says whether sweep 2 runs behind it (bit 0) and carries the sampled share of ambiguous cells (bits 8..).
template <int NT>
__device__ SL_FINISH_ATTR void fused_cube_build(FusedShared<NT>* shp, uint32_t* samp_, int n_sample_, int cps_log2_, int P_, float ylimf_, int want_cube_) {
    FusedShared<NT>& sh = *shp;
    uint32_t* samp = uni_ptr(samp_);
    const int n_sample = __builtin_amdgcn_readfirstlane(n_sample_), cps_log2 = __builtin_amdgcn_readfirstlane(cps_log2_), P = __builtin_amdgcn_readfirstlane(P_);
    const int want_cube = __builtin_amdgcn_readfirstlane(want_cube_);
    const float ylimf = uni(ylimf_);
    const int tid = threadIdx.x;
    float* ctab = reinterpret_cast<float*>(&sh.stage[0][0]);
    static_assert(sizeof(sh.stage) >= sizeof(float) * kCubeTabFloats && sizeof(sh.S.hist) >= 4 * kCubeWords, "");
    const CubeConsts cc = cube_tables(view_of_b(sh.tab), sh.Vf, sh.hi[0], sh.lo[1], sh.mk, ctab, tid);
    __syncthreads();
    int share_pct;
    const bool pays = cube_worthwhile<NT>(samp, n_sample, cps_log2, P, ctab, cc, ylimf, &sh.S.misc[33], tid, share_pct);
    const bool go = pays || want_cube == 2;                                                   // block-uniform
    if (tid == 0) sh.use_cube = (go ? 1 : 0) | (share_pct << 8);                              // (the share rides along for prefilter_out)
    if (go) cube_mask<NT>(ctab, cc, ylimf, sh.S.hist, tid);
    __syncthreads();
}

// Finish 1 of the merged schedule after the eigenvectors: brackets and thresholds into sh.lo / sh.hi / sh.mk.
// The sample: n_sample entries; stride_log2 >= 4: the stratified sample sweep 1 left (one pixel per 2^stride_log2), some entries may be
// absent; stride_log2 < 0: the DENSE cluster sample of the two-sweep schedule (every entry present, one per 2^-stride_log2 pixels or so,
// brackets widened by the design effect).
template <int NT>
__device__ SL_FINISH_ATTR void fused_finish1(FusedShared<NT>* shp, uint32_t* samp_, int n_sample_, int stride_log2_, int P_, float ylimf_, double pct_, double lam_,
                                           long long* subclk_, int want_cube_) {
    FusedShared<NT>& sh = *shp;
    uint32_t* samp = uni_ptr(samp_);
    const int n_sample = __builtin_amdgcn_readfirstlane(n_sample_), P = __builtin_amdgcn_readfirstlane(P_);
    const int sl2 = __builtin_amdgcn_readfirstlane(stride_log2_);
    const bool dense = sl2 < 0;
    const int stride_log2 = dense ? -sl2 : sl2;                       // pixels one entry stands for (log2)
    const int cps_log2 = dense ? kDenseCps : stride_log2 - 2;
    const float zs = dense ? (float)sqrt(kClusterDeff) : 1.0f;
    const int want_cube = __builtin_amdgcn_readfirstlane(want_cube_);
    const float ylimf = uni(ylimf_);
    const double pct = uni_d(pct_), lam = uni_d(lam_);
    long long* subclk = uni_ptr(subclk_);
    const int tid = threadIdx.x;
#ifdef SL_DEBUG_SUBCLK
#define SL_SUB(j) { __syncthreads(); if (subclk && tid == 0) subclk[(j)] = wall_clock64(); }
#else
#define SL_SUB(j)
#endif
    (void)subclk;
    {
        SampleAngleKey key;                                           // (scoped: kept alive across the function it cost the bracket code registers)
        key.sample = samp; key.tab = view_of_b(sh.tab); key.cps_log2 = cps_log2; key.P = P; key.ylimf = ylimf;
        for (int i = 0; i < 6; ++i) key.V[i] = sh.Vf[i];
        float lo[2], hi[2];
        float box[4];
        angle_brackets<NT>(key, n_sample, pct, lo, hi, sh.S, box, zs);
        if (tid == 0) {
            sh.lo[0] = lo[0]; sh.hi[0] = hi[0]; sh.lo[1] = lo[1]; sh.hi[1] = hi[1];
            for (int i = 0; i < 4; ++i) sh.box[i] = box[i];
            sh.xmin = tissue_x_bound(sh.Vf, ylimf, view_of_b(sh.tab));
            sh.use_cube = 0;
        }
        __syncthreads();
    }
    SL_SUB(12);
    // ---------------- the box of stain matrices the sample leaves possible, concentration brackets under its centre
    if (tid < 64) merged_box(sh.Vd, sh.box, lam, tid, sh.mk);
    __syncthreads();
    SL_SUB(13);
    if (sh.mk.ok) {                                               // block-uniform
        SampleConcKey ckey;
        ckey.sample = samp; ckey.tab = view_of_b(sh.tab); ckey.L = sh.mk.Lc; ckey.cps_log2 = cps_log2;
        ckey.P = P; ckey.col = 0;
        float lo[2], hi[2];
        conc_brackets<NT>(ckey, n_sample, lo, hi, sh.S, zs);
        if (tid == 0) merged_thresholds(sh.mk, lo[0], lo[1], hi[0], hi[1]);
    } else if (tid == 0) {
        merged_thresholds(sh.mk, -INFINITY, -INFINITY, -INFINITY, -INFINITY);       // disarms the concentration test
    }
    __syncthreads();
    SL_SUB(4);
    // ---------------- the colour-cube mask of the merged sweep (out of line: inlined here, its unrolled mask loop pushed the bracket
    // code's register-resident sample keys into scratch -- both bracket steps took twice as long)
    const float hi0 = sh.hi[0], lo1 = sh.lo[1];
    if (want_cube && hi0 > -INFINITY && hi0 < INFINITY && lo1 > -INFINITY && lo1 < INFINITY)       // block-uniform
        fused_cube_build<NT>(&sh, samp, n_sample, cps_log2, P, ylimf, want_cube);
    SL_SUB(7);      // (slot 7 is otherwise written on the resweep path only)
    // ---------------- the per-pixel sweep: does the projection bound stand in for its tissue test?
    // The bound makes the sweep collect NON-tissue pixels too when they pass it and lie outside the cone.  On most
    // tiles those are few; a uniform bright-but-not-white background (say 245, 245, 245: not tissue, first projection above the
    // bound, direction outside the stains' cone) would put most of the tile on the candidate list and cost it the exact
    // fallback (measured: 21 ms per 512 such tiles).  The sample says beforehand: if the pixels the bound would add exceed
    // P/40, this tile's sweep keeps the per-pixel tissue test.  (Not needed behind the cube mask, which tests the tissue bound
    // of a cell exactly.)
    const float xm = sh.xmin;
    if (!(sh.use_cube & 1) && xm > -INFINITY && xm < INFINITY) {  // block-uniform
        SampleAngleKey key;
        key.sample = samp; key.tab = view_of_b(sh.tab); key.cps_log2 = cps_log2; key.P = P; key.ylimf = ylimf;
        for (int i = 0; i < 6; ++i) key.V[i] = sh.Vf[i];
        if (tid == 0) sh.S.misc[32] = 0;
        __syncthreads();
        uint32_t extra = 0;
        for (int b = tid; b < n_sample; b += NT) {
            if (!key.present(b, n_sample)) continue;
            const uint32_t w = as_global(samp)[b];
            const uint32_t r = w & 255u, g = (w >> 8) & 255u, bl = (w >> 16) & 255u;
            const bool tissue = is_tissue_f(key.tab.gam(r), key.tab.gam(g), key.tab.gam(bl), ylimf);
            const float ox = key.tab.odf(r), oy = key.tab.odf(g), oz = key.tab.odf(bl);
            const float x = fmaf(key.V[4], oz, fmaf(key.V[2], oy, key.V[0] * ox));
            const float p = angle_key(key.V, ox, oy, oz);
            extra += (!tissue && x > xm && !(p > hi0 && p < lo1)) ? 1u : 0u;
        }
        for (int o = 32; o > 0; o >>= 1) extra += __shfl_xor((int)extra, o, 64);
        if ((tid & 63) == 0 && extra) atomicAdd(&sh.S.misc[32], extra);
        __syncthreads();
        if (tid == 0 && ((unsigned long long)sh.S.misc[32] << stride_log2) > (unsigned long long)P / 40ull) sh.xmin = -INFINITY;
        __syncthreads();
    }
#undef SL_SUB
}

// ---- The sweeps of the fused kernel, each OUT OF LINE (round 4) like the finish steps since round 3: every one of them gets
// the kernel's whole register budget to itself.  Inlined side by side in the 128-register kernel, a change to one sweep (or to a
// finish step) moved spills and copies into the loops of the others -- the same source measured 1.70 or 1.83 ms depending on
// what else the kernel held.  Uniform arguments arrive in VGPRs and are re-read into SGPRs; per-tile constants come from *shp.

// Geometry of the sweeps inside the fused kernel: a 512-thread workgroup sweeps the whole tile; the 1024-thread variant (one
// workgroup per CU, batches of up to #CU tiles) lets its two halves sweep the two halves of the tile exactly as two 512-thread
// workgroups of the per-phase schedule would (part_range keeps the halves aligned to whole trips), so every (lane, trip) sums the
// same 16 pixels in every variant and schedule: the binary32 bursts are bit-identical, only the order of the binary64 additions
// differs -- as between the two schedules.
template <int NT>
__device__ __forceinline__ void fused_geometry(int nch, int tid, int& t, int& c0, int& c1) {
    constexpr int H = NT / kFusedThreads;
    static_assert(NT % kFusedThreads == 0, "");
    if (H == 1) { t = tid; c0 = 0; c1 = nch; return; }
    t = tid % kFusedThreads;
    part_range(nch, H, __builtin_amdgcn_readfirstlane(tid / kFusedThreads), c0, c1);      // (the half is wave-uniform: tell the compiler)
    c0 = __builtin_amdgcn_readfirstlane(c0); c1 = __builtin_amdgcn_readfirstlane(c1);
}

// Sweep 1: tissue test, moments, sample; leaves the ten wave sums in sh.red (the caller adds them up after a barrier).
template <int NT, bool ALIGNED>
__device__ __noinline__ void fused_sweep1(FusedShared<NT>* shp, const uint8_t* src_, uint32_t* samp_, int P_, float ylimf_, int stride_log2_, int stream_) {
    FusedShared<NT>& sh = *shp;
    const uint8_t* src = uni_ptr(src_);
    uint32_t* samp = uni_ptr(samp_);
    const int P = __builtin_amdgcn_readfirstlane(P_), stride_log2 = __builtin_amdgcn_readfirstlane(stride_log2_), stream = __builtin_amdgcn_readfirstlane(stream_);
    const float ylimf = uni(ylimf_);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TabReaderB TB = TabReaderB::make_at(FusedLds<NT>::tab);
    const int nch = (P + 3) >> 2;
    Moments mo;
    uint32_t n_tissue = 0;
    int t, c0, c1;
    fused_geometry<NT>(nch, tid, t, c0, c1);
    if (c0 >= c1) {                                              // (a half without pixels: wave-uniform)
    } else if (stream) moments_sweep_b<ALIGNED, kFusedTrip, true>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, stride_log2, samp, mo, n_tissue);
    else moments_sweep_b<ALIGNED, kFusedTrip, false>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, stride_log2, samp, mo, n_tissue);
    double v[10];
    mo.to_array(v, n_tissue, lane);
#pragma unroll
    for (int i = 0; i < 10; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0)
        for (int i = 0; i < 10; ++i) sh.red[wave][i] = v[i];
}

// Sweeps 2 (merged stage, per-pixel test) and 3 (concentration stage): classify against sh.lo / sh.hi, raw candidates to the tile's
// list through the wave's staging in sh.stage.  merged: constants from sh.Vf / sh.mk / sh.xmin; otherwise from sh.L.
template <int NT, bool ALIGNED>
__device__ __noinline__ void fused_select(FusedShared<NT>* shp, const uint8_t* src_, uint32_t* rawl_, int P_, int cap_raw_, float ylimf_, int stream_, int merged_) {
    FusedShared<NT>& sh = *shp;
    const uint8_t* src = uni_ptr(src_);
    uint32_t* rawl = uni_ptr(rawl_);
    const int P = __builtin_amdgcn_readfirstlane(P_), cap_raw = __builtin_amdgcn_readfirstlane(cap_raw_), stream = __builtin_amdgcn_readfirstlane(stream_);
    const int merged = __builtin_amdgcn_readfirstlane(merged_);
    const float ylimf = uni(ylimf_);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TabReaderB TB = TabReaderB::make_at(FusedLds<NT>::tab);
    const int nch = (P + 3) >> 2;
    int t, c0, c1;
    fused_geometry<NT>(nch, tid, t, c0, c1);
    SelConsts K;
    K.lo0 = uni(sh.lo[0]); K.hi0 = uni(sh.hi[0]); K.lo1 = uni(sh.lo[1]); K.hi1 = uni(sh.hi[1]);
    RawSink sink{(uint32_t)__builtin_amdgcn_readfirstlane((int)(FusedLds<NT>::stage + (uint32_t)wave * FusedLds<NT>::stage_wave)), 0u, rawl, &sh.n_raw, &sh.overflow, (uint32_t)cap_raw,
                 (uint32_t)kStageWave};
    if (merged) {
        for (int i = 0; i < 6; ++i) K.V[i] = in_vgpr(sh.Vf[i]);
        K.L.g12 = 0.0f;
        K.xmin = uni(sh.xmin);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            K.u[i][0] = in_vgpr(sh.mk.u[i][0]); K.u[i][1] = in_vgpr(sh.mk.u[i][1]); K.kt[i] = in_vgpr(sh.mk.kt[i]);
            K.eps[i] = in_vgpr(sh.mk.eps[i]); K.thr[i] = in_vgpr(sh.mk.thr[i]);
        }
        if (c0 >= c1) {                                          // (a half without pixels: wave-uniform)
        } else if (K.xmin > -INFINITY) {                         // block-uniform: the projection bound stands in for the tissue test
            if (stream) select_sweep<kStageMerged, ALIGNED, kFusedTrip, true, true>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, K, sink);
            else select_sweep<kStageMerged, ALIGNED, kFusedTrip, false, true>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, K, sink);
        } else {
            if (stream) select_sweep<kStageMerged, ALIGNED, kFusedTrip, true>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, K, sink);
            else select_sweep<kStageMerged, ALIGNED, kFusedTrip, false>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, K, sink);
        }
    } else {
        K.L = sh.L;
        K.xmin = -INFINITY;
        vgpr(K.L);
        if (c0 >= c1) {
        } else if (stream) select_sweep<kStageConc, ALIGNED, kFusedTrip, true>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, K, sink);
        else select_sweep<kStageConc, ALIGNED, kFusedTrip, false>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, K, sink);
    }
    sink.flush(lane);
}

// Sweep 4: the apply pass with the tile's (sh.M, sh.maxC).
template <int NT, bool ALIGNED>
__device__ __noinline__ void fused_apply(FusedShared<NT>* shp, const uint8_t* src_, uint8_t* dst_, int P_, const double* M_tgt_, const double* maxC_tgt_,
                                         double lam_, int stream_) {
    FusedShared<NT>& sh = *shp;
    const uint8_t* src = uni_ptr(src_);
    uint8_t* dst = uni_ptr(dst_);
    const double* M_tgt = uni_ptr(M_tgt_);
    const double* maxC_tgt = uni_ptr(maxC_tgt_);
    const int P = __builtin_amdgcn_readfirstlane(P_), stream = __builtin_amdgcn_readfirstlane(stream_);
    const double lam = uni_d(lam_);
    const int tid = threadIdx.x;
    const TabReaderB TB = TabReaderB::make_at(FusedLds<NT>::tab);
    const int nch = (P + 3) >> 2;
    ApplyK K;
    apply_consts(sh.M, sh.maxC, M_tgt, maxC_tgt, lam, K);
    int t, c0, c1;
    fused_geometry<NT>(nch, tid, t, c0, c1);
    if (c0 >= c1) return;                                        // (a half without pixels: wave-uniform)
    if (stream) {
        if (K.fast) apply_sweep<ALIGNED, true, TabReaderB, true>(src, dst, P, c0, c1, t, kFusedThreads, TB, K);
        else apply_sweep<ALIGNED, false, TabReaderB, true>(src, dst, P, c0, c1, t, kFusedThreads, TB, K);
    } else {
        if (K.fast) apply_sweep<ALIGNED, true, TabReaderB, false>(src, dst, P, c0, c1, t, kFusedThreads, TB, K);
        else apply_sweep<ALIGNED, false, TabReaderB, false>(src, dst, P, c0, c1, t, kFusedThreads, TB, K);
    }
}

// Sweep 2 behind the colour-cube mask, out of line like the finish steps: its registers are allocated apart from the other sweeps'
// (the fused kernel sits at its 128-register limit; inlined, this sweep made the others spill).  Constants come from *shp.
template <int NT, bool ALIGNED>
__device__ __noinline__ void fused_sweep2_cube(FusedShared<NT>* shp, const uint8_t* src_, uint32_t* rawl_, uint32_t* rawa_, int P_, int cap_raw_, int cap_ang_,
                                               float ylimf_, int stream_) {
    FusedShared<NT>& sh = *shp;
    const uint8_t* src = uni_ptr(src_);
    uint32_t* rawl = uni_ptr(rawl_);
    uint32_t* rawa = uni_ptr(rawa_);
    const int P = __builtin_amdgcn_readfirstlane(P_), cap_raw = __builtin_amdgcn_readfirstlane(cap_raw_), stream = __builtin_amdgcn_readfirstlane(stream_);
    const int cap_ang = __builtin_amdgcn_readfirstlane(cap_ang_);
    const float ylimf = uni(ylimf_);
    const int tid = threadIdx.x, wave = tid >> 6;
    const TabReaderB TB = TabReaderB::make_at(FusedLds<NT>::tab);
    const int nch = (P + 3) >> 2;
    SelConsts K;
    for (int i = 0; i < 6; ++i) K.V[i] = in_vgpr(sh.Vf[i]);
    K.L.g12 = 0.0f;
    K.xmin = -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        K.u[i][0] = in_vgpr(sh.mk.u[i][0]); K.u[i][1] = in_vgpr(sh.mk.u[i][1]); K.kt[i] = in_vgpr(sh.mk.kt[i]);
        K.eps[i] = in_vgpr(sh.mk.eps[i]); K.thr[i] = in_vgpr(sh.mk.thr[i]);
    }
    K.lo0 = uni(sh.lo[0]); K.hi0 = uni(sh.hi[0]); K.lo1 = uni(sh.lo[1]); K.hi1 = uni(sh.hi[1]);
    static_assert(offsetof(FusedShared<NT>, n_ang) == offsetof(FusedShared<NT>, n_raw) + 4 && offsetof(FusedShared<NT>, n_raw) % 8 == 0, "");
    const RawDirect direct{rawl, rawa, reinterpret_cast<unsigned long long*>(&sh.n_raw), (uint32_t)cap_raw, (uint32_t)cap_ang};
    static_assert(offsetof(FusedShared<NT>, S) % 4096 == 0 && offsetof(SelScratch, hist) == 0, "the cube mask must be 4 KB aligned");
    const uint32_t bits_lds = FusedLds<NT>::hist;
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(FusedLds<NT>::stage + (uint32_t)wave * FusedLds<NT>::stage_wave));
    int t, c0, c1;
    fused_geometry<NT>(nch, tid, t, c0, c1);
    if (c0 >= c1) return;                                        // (a half without pixels: wave-uniform)
    if (stream) select_sweep_cube<ALIGNED, kFusedTrip, true>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, K, bits_lds, ring_lds, direct);
    else select_sweep_cube<ALIGNED, kFusedTrip, false>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, K, bits_lds, ring_lds, direct);
}

// Finish 2 of the merged schedule for the tile whose state sits in *shp: returns the number of slow exact fallbacks.
// Leaves sh.M, sh.status, sh.conc_done (and sh.maxC, sh.L when conc_done) behind and the row table rebuilt.
// ts (two-sweep schedule): the brackets in sh.lo / sh.hi were carried over from an estimate of the eigenvectors; when they do not hold
// the wanted ranks (or the angular list is incomplete) the function returns -1 at once -- the caller then takes the three-sweep route
// from the exact moments -- instead of falling back to the exact selection over the whole tile.
template <int NT>
__device__ SL_FINISH_ATTR int fused_finish2(FusedShared<NT>* shp, const uint8_t* src_, uint32_t* rawl_, uint32_t* rawa_, float* cand0_, float* cand1_, int P_,
                                          int cap_raw_, int cap_ang_, int cap_list_, float ylimf_, double pct_, double lam_, long long* subclk_, int ts_ = 0) {
    FusedShared<NT>& sh = *shp;
    const uint8_t* src = uni_ptr(src_);
    uint32_t* rawl = uni_ptr(rawl_);
    uint32_t* rawa = uni_ptr(rawa_);                              // null: one mixed list (the per-pixel sweeps); else the cube sweep's angular list
    const int cap_ang = __builtin_amdgcn_readfirstlane(cap_ang_);
    float* cand0 = uni_ptr(cand0_);
    float* cand1 = uni_ptr(cand1_);
    const int P = __builtin_amdgcn_readfirstlane(P_), cap_raw = __builtin_amdgcn_readfirstlane(cap_raw_), cap_list = __builtin_amdgcn_readfirstlane(cap_list_);
    const float ylimf = uni(ylimf_);
    const double pct = uni_d(pct_), lam = uni_d(lam_);
    long long* subclk = uni_ptr(subclk_);
    const int ts = __builtin_amdgcn_readfirstlane(ts_);
    const int tid = threadIdx.x, wave = tid >> 6;
    int fallbacks = 0;
#ifdef SL_DEBUG_SUBCLK
#define SL_SUB(j) { __syncthreads(); if (subclk && tid == 0) subclk[(j)] = wall_clock64(); }
#else
#define SL_SUB(j)
#endif
    (void)subclk;
    // ---------------- finish 2: exact angular percentiles -> M  (see "Finish 2 of the fused kernel" above wg_refine_s)
    const uint32_t T = (uint32_t)sh.sum[0];
    long long k[2];
    double gfrac[2];
    percentile_pos((double)T, 100.0 - pct, k[0], gfrac[0]);
    percentile_pos((double)T, pct, k[1], gfrac[1]);
    fin_tab_build(sh.tab);                                        // the row table's space: one-copy table + member staging
    const FinTab FT{FusedLds<NT>::tab};
    const uint32_t stage_lds = FusedLds<NT>::tab + kFinTabBytes + (uint32_t)wave * fin_stage_bytes(NT);
    const uint32_t stage_entries = fin_stage_bytes(NT) / 8u;      // two lists per wave
    AngleTileKey tkey;
    tkey.src = src; tkey.tab = FT.view(); tkey.ylimf = ylimf;
    WordAngleKey rkey;
    rkey.T = FT; rkey.ylimf = ylimf;
    for (int i = 0; i < 6; ++i) { tkey.V[i] = sh.Vf[i]; rkey.V[i] = sh.Vf[i]; }
    const bool complete = sh.n_raw <= (uint32_t)cap_raw && sh.overflow == 0;
    const uint32_t n_raw = sh.n_raw < (uint32_t)cap_raw ? sh.n_raw : (uint32_t)cap_raw;
    // the list the angular pass reads: the cube sweep's own (every tissue pixel outside the plain cone), else the mixed one
    const bool split = rawa != nullptr;                            // block-uniform
    const bool complete_a = split ? sh.n_ang <= (uint32_t)cap_ang : complete;
    const uint32_t n_a = split ? (sh.n_ang < (uint32_t)cap_ang ? sh.n_ang : (uint32_t)cap_ang) : n_raw;
    const uint32_t* lista = split ? rawa : rawl;
    const float los[2] = {sh.lo[0], sh.lo[1]}, his[2] = {sh.hi[0], sh.hi[1]};
    SL_SUB(2);
    const RefineOut ra = wg_refine_s(lista, (int)n_a, rkey, los[0], his[0], los[1], his[1], cand0, cand1, (uint32_t)cap_list, stage_lds,
                                     stage_entries, sh.S);
    SL_SUB(3);
    {
        // plain = tissue pixels the sweep did not collect as angular candidates: they sit between the two brackets (a mixed list
        // also holds pixels collected for their concentrations; those with an angle key count like any other candidate)
        const long long lt[2] = {(long long)ra.n_lt[0], (long long)T - (long long)ra.n_valid + (long long)ra.n_lt[1]};
        if (ts) {                                                 // block-uniform
            bool covered = complete_a;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const long long k2 = (k[b] + 1 < (long long)T) ? k[b] + 1 : k[b];
                covered = covered & (k[b] >= lt[b]) & (k2 < lt[b] + (long long)ra.n_in[b]) & (ra.n_in[b] <= (uint32_t)cap_list);
            }
            if (!covered) {
                fin_tab_expand<NT>(sh.tab);
                return -1;
            }
        }
        float res[4];
        stage_pick2<false>(cand0, cand1, ra.n_in, (uint32_t)cap_list, complete_a, los, his, lt, P, tkey, T, k, ra.ps, res, fallbacks, sh.S);
        if (tid == 0) { sh.res[0] = res[0]; sh.res[1] = res[1]; sh.res[2] = res[2]; sh.res[3] = res[3]; }
        __syncthreads();
    }
    SL_SUB(5);
    if (tid < 64) {
        double M[6];
        stain_matrix_from_angles(sh.Vd, sh.res, gfrac, M, tid);
        if (tid == 0) {
            for (int i = 0; i < 6; ++i) sh.M[i] = M[i];
            if (stain_matrix_singular(M)) sh.status = SL_TILE_DEGENERATE_COV;
            // the concentration candidates of the merged sweep are usable iff the exact M lies where the sweep assumed
            const bool use = sh.status == SL_TILE_OK && complete && merged_verify(sh.mk, M, lam);
            sh.conc_done = use ? 1 : 0;
            sh.why = use ? 0 : (!sh.mk.ok ? SL_RESWEEP_NO_BOX : (!complete ? SL_RESWEEP_LIST_FULL : SL_RESWEEP_OUTSIDE_BOX));
            if (use) { LassoK L; lasso_consts(M, lam, L); sh.L = L; }
        }
    }
    __syncthreads();
    SL_SUB(6);
    if (sh.conc_done) {                                           // block-uniform
        // ---------------- finish 2b: exact 99th percentiles of the concentrations from the same raw list
        long long kc;
        double gc;
        percentile_pos((double)P, 99.0, kc, gc);
        const long long kc2 = kc + 1 < (long long)P ? kc + 1 : kc;
        WordConcKey ckey2;
        ckey2.T = FT; ckey2.L = sh.L;
        const float cl[2] = {sh.mk.L[0], sh.mk.L[1]}, chh[2] = {sh.mk.H[0], sh.mk.H[1]};
        const RefineOut rc = wg_refine_s(rawl, (int)n_raw, ckey2, cl[0], chh[0], cl[1], chh[1], cand0, cand1, (uint32_t)cap_list,
                                         stage_lds, stage_entries, sh.S);
        SL_SUB(14);
#ifdef SL_DEBUG_SUBCLK
        if (subclk && tid == 0) {             // list sizes beside the clocks, x 100: the tools scale by 0.01 (slots 8..11 are clocks only on the resweep path)
            long long* q = subclk;
            q[8] = 100ll * n_a; q[9] = 100ll * n_raw; q[10] = 100ll * (ra.n_in[0] + ra.n_in[1]); q[11] = 100ll * (rc.n_in[0] + rc.n_in[1]);
        }
#endif
        const long long n_plain = (long long)P - (long long)sh.n_raw;       // proven below both brackets
        const long long clt[2] = {n_plain + rc.n_lt[0], n_plain + rc.n_lt[1]};
        bool covered = true;
#pragma unroll
        for (int col = 0; col < 2; ++col)
            covered = covered & (kc >= clt[col]) & (kc2 < clt[col] + (long long)rc.n_in[col]) & (rc.n_in[col] <= (uint32_t)cap_list);
        if (covered) {
            ConcTileKey ctk;
            ctk.src = src; ctk.tab = FT.view(); ctk.L = sh.L; ctk.col = 0;
            const long long kk[2] = {kc, kc};
            float res[4];
            stage_pick2<true>(cand0, cand1, rc.n_in, (uint32_t)cap_list, true, cl, chh, clt, P, ctk, (uint32_t)P, kk, rc.ps, res, fallbacks, sh.S);
            if (tid == 0) {
                sh.maxC[0] = np_lerp((double)res[0], (double)res[1], gc);   // normalizer.py:36,47
                sh.maxC[1] = np_lerp((double)res[2], (double)res[3], gc);
                if (!(sh.maxC[0] > 0.0) || !(sh.maxC[1] > 0.0)) sh.status = SL_TILE_ZERO_MAXC;
            }
        } else if (tid == 0) {
            sh.conc_done = 0;                                     // a bracket missed (or holds more members than a list): sweep 3 settles it
            sh.why = (rc.n_in[0] > (uint32_t)cap_list || rc.n_in[1] > (uint32_t)cap_list) ? SL_RESWEEP_LIST_FULL : SL_RESWEEP_BRACKET_MISSED;
        }
        __syncthreads();
        SL_SUB(15);
    }
    fin_tab_expand<NT>(sh.tab);                                   // the row table back for the sweeps to come
    return fallbacks;
#undef SL_SUB
}

// ------------------------------------------------------------------------------------------
// two-sweep schedule (stats_twosweep.hpp): phase 0 and the merged sweep, out of line like every other phase
// ------------------------------------------------------------------------------------------
// Phase 0: the cluster sample of the tile (into registers, and into samp[0 .. n_lines * kClusterPx) for the routes that fall back),
// its eigenvectors, the angular brackets under them, the two half-spaces of the plain cone, the box of stain matrices with its tilts, the
// concentration brackets under the box centre and the colour-cube mask of the merged sweep (S.hist).  Leaves sh.ts (ts.ok: sweep 1
// collects candidates), sh.mk, and sh.use_cube (share).
// Every workgroup of a launch starts here at the same moment and nothing streams meanwhile, so this phase is written for LATENCY: the
// sample words stay in registers (thread t holds entries t, t + NT, ...), each bracket set costs one evaluation of the keys into a
// fixed-range histogram and one wave_locate per rank (the register-resident windowed search of finish 1 needs four passes and three
// times the barriers: 33-45 us per set where this takes ~15; its brackets are tighter by a histogram bin -- 0.002 of pseudo-angle,
// ~1 % of a concentration -- which costs this schedule a few hundred candidates), and the fourth moments ride in the angle pass.
// Phase 0, one thread: the sample's eigenvectors, the plane's normal and the Gaussian tilt bound from sh.sum (out of line: the binary64
// Jacobi sweep wants three dozen registers of its own while every thread of fused_phase0 holds its 32 sample words -- inlined, 28 of
// them went to scratch around it; out of line they sit in callee-saved registers).
template <int NT>
__device__ __noinline__ void phase0_estimate(FusedShared<NT>* shp, int mode_) {
    FusedShared<NT>& sh = *shp;
    const int mode = __builtin_amdgcn_readfirstlane(mode_);
    double Vd[6], wv[3];
    float Vf[6];
    const int st = eigvecs_from_moments(sh.sum, Vd, Vf, wv);
    const double ns = sh.sum[0];
    // the tilt a Gaussian cloud of this size would show: v_j^T dC v_3 / (l_j - l_3), sd sqrt(l_j l_3 / n)
    const double l1 = wv[0], l2 = wv[1], l3 = wv[2] > 0.0 ? wv[2] : 0.0;
    bool ok = st == SL_TILE_OK && ns >= (double)kTsMinTissue && l2 > 1e-9 * l1 && l2 - l3 > 0.05 * l2;
    double tau = 0.0;
    if (ok) {
        tau = kTiltZ * sqrt(kClusterDeff / ns) * fmax(sqrt(l3 * l2) / (l2 - l3), sqrt(l3 * l1) / (l1 - l3));
        tau = fmax(tau, kTsMinTau);
        // Tissue with a strong third component (the ihc fixture: 1.3e-2 here; H&E-like tiles 2-4e-3): the tilt allowance then puts a quarter
        // of the pixels on the concentration list and the tile would decline at the END of this phase, 170 us later -- it leaves now.
        ok = tau <= (mode >= 2 ? kTsMaxTau : kTsAutoMaxTau);
    }
    double nd[3] = {Vd[2] * Vd[5] - Vd[4] * Vd[3], Vd[4] * Vd[1] - Vd[0] * Vd[5], Vd[0] * Vd[3] - Vd[2] * Vd[1]};
    const double nn = 1.0 / sqrt(nd[0] * nd[0] + nd[1] * nd[1] + nd[2] * nd[2]);
    if (mode == 4 && ok) {                                    // tests: the sample's plane tilted by 0.05 (second column towards the normal)
        double b[3];
        for (int c = 0; c < 3; ++c) b[c] = Vd[2 * c + 1] + 0.05 * nd[c] * nn;
        const double nb = 1.0 / sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
        for (int c = 0; c < 3; ++c) { Vd[2 * c + 1] = b[c] * nb; Vf[2 * c + 1] = (float)Vd[2 * c + 1]; }
        nd[0] = Vd[2] * Vd[5] - Vd[4] * Vd[3]; nd[1] = Vd[4] * Vd[1] - Vd[0] * Vd[5]; nd[2] = Vd[0] * Vd[3] - Vd[2] * Vd[1];
    }
    const double nn2 = 1.0 / sqrt(nd[0] * nd[0] + nd[1] * nd[1] + nd[2] * nd[2]);
    for (int i = 0; i < 6; ++i) { sh.ts.Vd[i] = Vd[i]; sh.Vd[i] = Vd[i]; sh.Vf[i] = Vf[i]; }
    for (int c = 0; c < 3; ++c) { sh.ts.nd[c] = nd[c] * nn2; sh.ts.fn[c] = (float)(nd[c] * nn2); }
    sh.ts.tau = tau;
    sh.ts.gH[0] = l1 - l3; sh.ts.gH[1] = l2 - l3;             // (scratch: the eigenvalue gaps for the fourth moments below)
    sh.res[0] = (float)(4.0 * sqrt(l3));                      // zref of ts_thresholds
    sh.res[1] = (float)(sh.sum[1] / ns); sh.res[2] = (float)(sh.sum[2] / ns); sh.res[3] = (float)(sh.sum[3] / ns);     // the sample's mean
    sh.status = ok ? SL_TILE_OK : SL_TILE_DEGENERATE_COV;     // (scratch here: sweep 1 sets the tile's real status)
}

// Phase 0, wave 0: the two half-spaces of the plain cone (normals from the inner bracket ends) and the box of stain matrices.  Out of line
// like phase0_estimate: its binary64 trigonometry made EVERY wave of fused_phase0 park 26 registers in scratch on the way past it.
template <int NT>
__device__ __noinline__ void phase0_cone_and_box(FusedShared<NT>* shp, float hi0_, float lo1_, double lam_) {
    FusedShared<NT>& sh = *shp;
    const float hi0 = uni(hi0_), lo1 = uni(lo1_);
    const double lam = uni_d(lam_);
    const int tid = threadIdx.x;
    // half-space normals: H-type (-sin a, cos a), L-type (sin a, -cos a) in the plane, then V~ n
    const double aH = angle_of_pseudo((double)hi0), aL = angle_of_pseudo((double)lo1);
    double sH, cH, sL, cL;
    sincos(aH, &sH, &cH);
    sincos(aL, &sL, &cL);
    if (tid == 0) {
        const double k2 = sh.ts.kappa2 + 6e-6;                // + the sweep's own binary32 rounding and that of the finish's keys
        for (int c = 0; c < 3; ++c) {
            const double gH = sh.ts.Vd[2 * c] * -sH + sh.ts.Vd[2 * c + 1] * cH, gL = sh.ts.Vd[2 * c] * sL + sh.ts.Vd[2 * c + 1] * -cL;
            sh.ts.gH[c] = gH; sh.ts.gL[c] = gL;
            sh.ts.fgH[c] = (float)(gH - k2); sh.ts.fgL[c] = (float)(gL - k2);
        }
        sh.ts.fk1 = (float)(sh.ts.kappa1 * (1.0 + 1e-6));
        sh.ts.lo0 = sh.lo[0]; sh.ts.hi1 = sh.hi[1];
    }
    // ---------------- the box of stain matrices (in-plane grid x tilts)
    ts_box(sh.ts.Vd, sh.ts.nd, sh.ts.tau, sh.box, lam, tid, sh.mk);
}

constexpr int kP0Bins = 1024;            // angle keys: pseudo-angle [-1, 1) in 1024 bins; concentrations: 512 bins per stain of c / (c + 1)
template <int NT>
__device__ SL_FINISH_ATTR void fused_phase0(FusedShared<NT>* shp, const uint8_t* src_, uint32_t* samp_, int P_, int n_lines_, float ylimf_, double pct_, double lam_,
                                          int mode_, int cap_raw_, int cap_ang_, int cap_list_, long long* subclk_) {
    FusedShared<NT>& sh = *shp;
    const uint8_t* src = uni_ptr(src_);
    uint32_t* samp = uni_ptr(samp_);
    const int P = __builtin_amdgcn_readfirstlane(P_), n_lines = __builtin_amdgcn_readfirstlane(n_lines_), mode = __builtin_amdgcn_readfirstlane(mode_);
    const int cap_raw = __builtin_amdgcn_readfirstlane(cap_raw_), cap_ang = __builtin_amdgcn_readfirstlane(cap_ang_), cap_list = __builtin_amdgcn_readfirstlane(cap_list_);
    const float ylimf = uni(ylimf_);
    const double pct = uni_d(pct_), lam = uni_d(lam_);
    long long* subclk = uni_ptr(subclk_);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_sample = n_lines * kClusterPx;
    constexpr int KPT = kMaxSample / NT;
    static_assert(KPT % 8 == 0, "");
#ifdef SL_DEBUG_SUBCLK
#define SL_SUB(j) { __syncthreads(); if (subclk && tid == 0) subclk[(j)] = wall_clock64(); }
#else
#define SL_SUB(j)
#endif
    (void)subclk;
    SL_SUB(0);
    if (tid == 0) { sh.ts.ok = 0; sh.ts.why = kTsNoEstimate; sh.mk.ok = 0; sh.use_cube = 0; }
    const TabView tab = view_of_b(sh.tab);
    // ---------------- the sample: gathered into registers (8 loads in flight), written out for the fallback routes, summed on the way
    uint32_t w[KPT];
    Moments mo;
    BurstMoments bm;                                              // (binary32 sums of a thread's <= 32 entries: the estimate needs no more)
    double cnt = 0.0;
    {
        const uint32_t nl = (uint32_t)(((3ll * P) >> 7) < 1 ? 1 : ((3ll * P) >> 7));
        const uint32_t wl = nl / (uint32_t)n_lines;                // lines per stratum (>= 1)
#pragma unroll
        for (int j0 = 0; j0 < KPT; j0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = min((j0 + u) * NT + tid, n_sample - 1);      // (clamped, never predicated)
                w[j0 + u] = cluster_word(src, P, wl, (uint32_t)b);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = (j0 + u) * NT + tid;
                if (b < n_sample) {
                    as_global(samp)[b] = w[j0 + u];
                    const uint32_t r = w[j0 + u] & 255u, g = (w[j0 + u] >> 8) & 255u, bl = (w[j0 + u] >> 16) & 255u;
                    if (is_tissue_f(tab.gam(r), tab.gam(g), tab.gam(bl), ylimf)) {
                        bm.add(tab.odf(r), tab.odf(g), tab.odf(bl));
                        cnt += 1.0;
                    }
                }
            }
        }
        bm.flush(mo);
    }
    {
        double v[10];
        mo.to_array(v, 0u, lane);
        v[0] = cnt;
#pragma unroll
        for (int i = 0; i < 10; ++i) v[i] = wave_sum(v[i]);
        if (lane == 0)
            for (int i = 0; i < 10; ++i) sh.red[wave][i] = v[i];
    }
    for (int i = tid; i < kP0Bins; i += NT) sh.S.hist[i] = 0;
    __syncthreads();
    SL_SUB(1);
    if (tid < 10) {
        double t = 0;
        for (int wv = 0; wv < NT / 64; ++wv) t += sh.red[wv][tid];
        sh.sum[tid] = t;
    }
    __syncthreads();
    if (tid == 0) phase0_estimate<NT>(shp, mode);
    __syncthreads();
    SL_SUB(2);
    if (sh.status != SL_TILE_OK) return;                          // block-uniform: no estimate
    // ---------------- ONE pass over the sample: the angle keys under V~ into a fixed-range histogram, and the fourth moments that give the
    // tilt's standard error: the eigenvector perturbation is v_j^T dC n / (l_j - l_3) with dC the sampling error of the covariance, whose
    // (j, 3) entry has variance sum (a_j a_3)^2 / n^2 (a = the centred pixel in the eigenbasis).  The Gaussian figure above misses what a
    // few saturated pixels in the sample do to it (measured on the i.i.d. bench batch: the exact plane 2-8 % outside a 7-sigma Gaussian
    // bound in one tile of 750).
    {
        float V[6], nf[3];
        for (int i = 0; i < 6; ++i) V[i] = sh.Vf[i];
        for (int c = 0; c < 3; ++c) nf[c] = sh.ts.fn[c];
        const float mx = sh.res[1], my = sh.res[2], mz = sh.res[3];
        float q1 = 0.0f, q2 = 0.0f;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int b = j * NT + tid;
            const uint32_t r = w[j] & 255u, g = (w[j] >> 8) & 255u, bl = (w[j] >> 16) & 255u;
            const float ox = tab.odf(r), oy = tab.odf(g), oz = tab.odf(bl);
            const bool tissue = (b < n_sample) & is_tissue_f(tab.gam(r), tab.gam(g), tab.gam(bl), ylimf);
            if (tissue) {
                const float key = angle_key(V, ox, oy, oz);
                const int bin = min(kP0Bins - 1, max(0, (int)((key + 1.0f) * (0.5f * kP0Bins))));
                atomicAdd(&sh.S.hist[bin], 1u);
                const float dx = ox - mx, dy = oy - my, dz = oz - mz;
                const float a1 = fmaf(V[4], dz, fmaf(V[2], dy, V[0] * dx)), a2 = fmaf(V[5], dz, fmaf(V[3], dy, V[1] * dx));
                const float a3 = fmaf(nf[2], dz, fmaf(nf[1], dy, nf[0] * dx));
                q1 = fmaf(a1 * a3, a1 * a3, q1);
                q2 = fmaf(a2 * a3, a2 * a3, q2);
            }
        }
        double d1 = wave_sum((double)q1), d2 = wave_sum((double)q2);
        if (lane == 0) { sh.red[wave][0] = d1; sh.red[wave][1] = d2; }
    }
    __syncthreads();
    {
        // ranks of the two percentiles -/+ z sigma (z widened by the design effect), as wg_brackets_regs takes them; waves 0..3 locate the
        // 6-sigma ranks, waves 4..7 the kBoxZ ones (the box when a 6-sigma end is open)
        const double n = sh.sum[0];
        const double zsd = sqrt(kClusterDeff);
        if (wave < 8) {
            const int which = wave & 3, b = which >> 1, upper = which & 1;
            const double q = (b == 0 ? 100.0 - pct : pct) / 100.0;
            const double z = (wave < 4 ? (double)kBracketZ : (double)kBoxZ) * zsd;
            const double r = q * (n - 1.0), sd = sqrt(fmax(q * (1.0 - q) * n, 0.0));
            const long long rk = upper ? (long long)ceil(r + z * sd) + 1 : (long long)floor(r - z * sd) - 1;
            const bool open = upper ? rk > (long long)n - 1 : rk < 0;
            uint32_t* o = &sh.S.misc[kLocateOut - 24 + 3 * wave];             // slots 16..39
            if (!open) wave_locate(sh.S.hist, kP0Bins, (uint32_t)rk, o, lane);
            else if (lane == 0) o[0] = 0xffffffffu;
        }
        if (tid == 0) {
            double t1 = 0.0, t2 = 0.0;
            for (int wv = 0; wv < NT / 64; ++wv) { t1 += sh.red[wv][0]; t2 += sh.red[wv][1]; }
            const double se = fmax(sqrt(t1) / (n * sh.ts.gH[0]), sqrt(t2) / (n * sh.ts.gH[1]));
            const double tau = fmax(sh.ts.tau, kTiltZ4 * sqrt(kClusterDeff) * se);
            sh.ts.tau = tau;
            sh.ts.kappa1 = tau;
            sh.ts.kappa2 = tau * tau + 1e-7;
            if (!(tau <= kTsMaxTau)) sh.status = SL_TILE_DEGENERATE_COV;
        }
    }
    __syncthreads();
    if (tid == 0) {
        // bracket ends = bin edges on the safe side; a rank in the first or last bin (keys beyond [-1, 1): x < 0) counts as open
        float e[8];
        for (int i = 0; i < 8; ++i) {
            const uint32_t bin = sh.S.misc[kLocateOut - 24 + 3 * i];
            const bool upper = i & 1;
            const bool open = bin == 0xffffffffu || bin == 0u || bin >= (uint32_t)(kP0Bins - 1);
            e[i] = open ? (upper ? INFINITY : -INFINITY) : -1.0f + (float)(bin + (upper ? 1u : 0u)) * (2.0f / kP0Bins);
        }
        sh.lo[0] = e[0]; sh.hi[0] = e[1]; sh.lo[1] = e[2]; sh.hi[1] = e[3];
        const bool closed = (e[0] > -INFINITY) & (e[1] < INFINITY) & (e[2] > -INFINITY) & (e[3] < INFINITY);
        if (closed) {
            for (int b = 0; b < 2; ++b) {
                const float m = 0.5f * (e[2 * b] + e[2 * b + 1]), r = (float)kBoxFrac * 0.5f * (e[2 * b + 1] - e[2 * b]);
                sh.box[2 * b] = m - r; sh.box[2 * b + 1] = m + r;
            }
        } else {
            for (int i = 0; i < 4; ++i) sh.box[i] = e[4 + i];
        }
    }
    for (int i = tid; i < kP0Bins; i += NT) sh.S.hist[i] = 0;     // (the locates are done: the barrier above)
    __syncthreads();
    SL_SUB(3);
    if (sh.status != SL_TILE_OK) return;
    // the plain cone must be closed on its inner sides and lie within +-90 degrees of the first eigenvector
    const float hi0 = sh.hi[0], lo1 = sh.lo[1];
    if (!(hi0 > -0.95f && hi0 < lo1 && lo1 < 0.95f)) return;      // block-uniform (NaN / open ends fail the comparisons)
    if (tid < 64) phase0_cone_and_box<NT>(shp, hi0, lo1, lam);
    __syncthreads();
    SL_SUB(4);
    // ---------------- concentration brackets under the box centre: both stains' keys as c / (c + 1) into 512 bins each
    if (sh.mk.ok) {                                               // block-uniform
        LassoK L = sh.mk.Lc;
        uint32_t nz1 = 0, nz2 = 0;                                // wave-uniform
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int b = j * NT + tid;
            float c1, c2;
            lasso2(L, tab.odf(w[j] & 255u), tab.odf((w[j] >> 8) & 255u), tab.odf((w[j] >> 16) & 255u), c1, c2);
            const int b1 = min(511, max(0, (int)(512.0f * c1 * __builtin_amdgcn_rcpf(c1 + 1.0f))));
            const int b2 = min(511, max(0, (int)(512.0f * c2 * __builtin_amdgcn_rcpf(c2 + 1.0f))));
            // (background pixels all have concentration 0: thousands of atomics on ONE bin took most of this step -- entries of bin 0 are
            //  counted off the ballot instead, one add per wave and row)
            const bool in = b < n_sample;
            const unsigned long long z1 = __builtin_amdgcn_ballot_w64(in & (b1 == 0)), z2 = __builtin_amdgcn_ballot_w64(in & (b2 == 0));
            if (in & (b1 != 0)) atomicAdd(&sh.S.hist[b1], 1u);
            if (in & (b2 != 0)) atomicAdd(&sh.S.hist[512 + b2], 1u);
            nz1 += (uint32_t)__popcll(z1); nz2 += (uint32_t)__popcll(z2);
        }
        if (lane == 0) { if (nz1) atomicAdd(&sh.S.hist[0], nz1); if (nz2) atomicAdd(&sh.S.hist[512], nz2); }
        __syncthreads();
        if (wave < 4) {                                           // ranks of the 99th percentile of ALL sample entries -/+ z sigma, per stain
            const int col = wave >> 1, upper = wave & 1;
            const double n = (double)n_sample, q = 0.99;
            const double z = (double)kBracketZ * sqrt(kClusterDeff);
            const double r = q * (n - 1.0), sd = sqrt(q * (1.0 - q) * n);
            const long long rk = upper ? (long long)ceil(r + z * sd) + 1 : (long long)floor(r - z * sd) - 1;
            const bool open = upper ? rk > (long long)n - 1 : rk < 0;
            uint32_t* o = &sh.S.misc[kLocateOut - 24 + 3 * wave];
            if (!open) wave_locate(sh.S.hist + 512 * col, 512, (uint32_t)rk, o, lane);
            else if (lane == 0) o[0] = 0xffffffffu;
        }
        __syncthreads();
        if (tid == 0) {
            float e[4];
            for (int i = 0; i < 4; ++i) {
                const uint32_t bin = sh.S.misc[kLocateOut - 24 + 3 * i];
                const bool upper = i & 1;
                const bool open = bin == 0xffffffffu || (upper && bin >= 511u);
                const float edge = (float)(bin + (upper ? 1u : 0u));                     // u = c / (c + 1) = edge / 512  ->  c = edge / (512 - edge)
                e[i] = open ? (upper ? INFINITY : -INFINITY) : (upper ? edge / (512.0f - edge) * (1.0f + 1e-5f) : edge / (512.0f - edge) * (1.0f - 1e-5f));
            }
            ts_thresholds(sh.mk, e[0], e[2], e[1], e[3], sh.res[0]);
        }
    }
    __syncthreads();
    SL_SUB(7);
    // No box of stain matrices (a small tissue sample: the widened ranks leave it; or a box over which the map changes too much): the
    // concentration candidates could not ride in the sweep, and the three-sweep route, whose stratified sample needs no widening, may
    // well have its box -- declined.
    if (!sh.mk.ok && mode < 2) return;                            // block-uniform (ts.why = kTsNoEstimate)
    if (!sh.mk.ok && tid == 0) ts_thresholds(sh.mk, -INFINITY, -INFINITY, -INFINITY, -INFINITY, 0.0f);       // (forced: the concentration test disarmed)
    if (tid == 0) {
        for (int i = 0; i < 2; ++i) {
            for (int c = 0; c < 3; ++c) sh.ts.W[i][c] = sh.mk.ok ? (float)sh.mk.C.W[i][c] : 0.0f;
            sh.ts.kt[i] = sh.mk.kt[i]; sh.ts.eps[i] = sh.mk.eps[i]; sh.ts.zeta[i] = (float)(sh.mk.zeta[i] * (1.0 + 1e-6)); sh.ts.thr[i] = sh.mk.thr[i];
        }
        sh.S.misc[34] = 0; sh.S.misc[35] = 0;
    }
    __syncthreads();
    // Will the lists hold what the sweep is going to collect?  The histograms of the sample's concentrations (still in S.hist) say how many
    // entries lie above each stain's threshold once it is lowered by what the box adds to a typical candidate (eps (ref_1 + ref_2) + zeta
    // zref: the exact test's own allowance at the bracket) -- an estimate, which is all a decision about SPEED needs.  Real tissue with a
    // strong second stain (the ihc fixture) puts a quarter of its pixels on the concentration list once the box has to cover a tilt as
    // well -- the uncertainty of the weaker stain's concentration under a heavily stained pixel is real -- and a list that overflows
    // costs the tile a separate concentration sweep on top (measured: 2.06 ms per 512 such tiles against 1.62 for three sweeps): such a
    // tile keeps the three-sweep schedule.  (A pass of the sweep's exact test over the sample gave the same verdicts for 57 us more.)
    if (sh.mk.ok) {                                               // block-uniform
        if (wave < 2) {
            const float ref = (sh.mk.H[0] < INFINITY ? sh.mk.H[0] : 2.0f * sh.mk.L[0] + 1.0f) + (sh.mk.H[1] < INFINITY ? sh.mk.H[1] : 2.0f * sh.mk.L[1] + 1.0f);
            const float cut = fmaxf(sh.mk.thr[wave] - sh.mk.eps[wave] * ref - (float)sh.mk.zeta[wave] * sh.res[0], 0.0f);
            const int bl = min(511, max(0, (int)(512.0f * cut / (cut + 1.0f))));
            uint32_t cntm = 0;
            for (int b = lane; b < 512; b += 64) cntm += b >= bl ? sh.S.hist[512 * wave + b] : 0u;
            for (int o = 32; o > 0; o >>= 1) cntm += (uint32_t)__shfl_xor((int)cntm, o, 64);
            if (lane == 0) atomicAdd(&sh.S.misc[35], cntm);
        }
        __syncthreads();
        const double scale = (double)P / (double)n_sample;
        const bool full = (double)sh.S.misc[35] * scale > 0.75 * (double)cap_raw || (double)sh.S.misc[35] * scale > 1.5 * (double)cap_list;
        if (full && mode < 2) {                                   // block-uniform
            if (tid == 0) sh.ts.why = kTsLists;
            return;
        }
    }
    (void)cap_ang;
    SL_SUB(5);
    // ---------------- the colour cube of the merged sweep
    {
        float* ctab = reinterpret_cast<float*>(&sh.stage[0][0]);
        static_assert(sizeof(sh.stage) >= sizeof(float) * kTsTabFloats, "");
        const TsCubeConsts cc = ts_cube_tables(view_of_b(sh.tab), sh.ts, ctab, tid);
        if (tid == 0) sh.S.misc[33] = 0;
        __syncthreads();
        // the share of the sample in ambiguous cells, from two of the register rows (1/16 of the entries)
        uint32_t amb = 0, seen = 0;
#pragma unroll
        for (int j = 0; j < KPT; j += KPT / 2) {
            if (j * NT + tid < n_sample) {
                amb += ts_cell_plain(ctab, cc, ylimf, w[j]) ? 0u : 1u;
                ++seen;
            }
        }
        uint32_t both = amb | (seen << 16);
        for (int o = 32; o > 0; o >>= 1) both += (uint32_t)__shfl_xor((int)both, o, 64);
        if (lane == 0 && both) atomicAdd(&sh.S.misc[33], both);
        __syncthreads();
        const uint32_t tot = sh.S.misc[33];
        const int share = (tot >> 16) ? (int)(100u * (tot & 0xffffu) / (tot >> 16)) : 100;
        const bool go = share <= kTsMaxSharePct || mode >= 2;       // block-uniform
        if (tid == 0) {
            sh.use_cube = (go ? 1 : 0) | (share << 8);
            sh.ts.ok = go ? 1 : 0;
            sh.ts.why = go ? kTsDirect : kTsShare;
        }
        if (go) ts_cube_mask<NT>(ctab, cc, ylimf, sh.S.hist, tid);
        __syncthreads();
    }
    SL_SUB(6);
#undef SL_SUB
}

// Sweep 1 of the two-sweep schedule: moments into sh.red (as fused_sweep1) and the candidates of all four order statistics into the
// tile's two lists (as fused_sweep2_cube), under the estimate in sh.ts.
template <int NT, bool ALIGNED>
__device__ __noinline__ void fused_sweep1c(FusedShared<NT>* shp, const uint8_t* src_, uint32_t* rawl_, uint32_t* rawa_, int P_, int cap_raw_, int cap_ang_,
                                           float ylimf_, int stream_) {
    FusedShared<NT>& sh = *shp;
    const uint8_t* src = uni_ptr(src_);
    uint32_t* rawl = uni_ptr(rawl_);
    uint32_t* rawa = uni_ptr(rawa_);
    const int P = __builtin_amdgcn_readfirstlane(P_), cap_raw = __builtin_amdgcn_readfirstlane(cap_raw_), stream = __builtin_amdgcn_readfirstlane(stream_);
    const int cap_ang = __builtin_amdgcn_readfirstlane(cap_ang_);
    const float ylimf = uni(ylimf_);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TabReaderB TB = TabReaderB::make_at(FusedLds<NT>::tab);
    const int nch = (P + 3) >> 2;
    TsSweepConsts K;
    // (wave-uniform constants stay in SGPRs although a vector operation with an SGPR operand issues at half rate: the exact test runs for the
    //  pixels of ambiguous cells only, and in VGPRs the 24 constants made the drains spill -- measured 1.475-1.50 ms against 1.425-1.45)
#define SL_TSK(x) uni(x)
#pragma unroll
    for (int c = 0; c < 3; ++c) { K.gH[c] = SL_TSK(sh.ts.fgH[c]); K.gL[c] = SL_TSK(sh.ts.fgL[c]); K.n[c] = SL_TSK(sh.ts.fn[c]); }
    K.k1 = SL_TSK(sh.ts.fk1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) K.W[i][c] = SL_TSK(sh.ts.W[i][c]);
        K.kt[i] = SL_TSK(sh.ts.kt[i]); K.eps[i] = SL_TSK(sh.ts.eps[i]); K.zeta[i] = SL_TSK(sh.ts.zeta[i]); K.thr[i] = uni(sh.ts.thr[i]);
    }
#undef SL_TSK
    const RawDirect direct{rawl, rawa, reinterpret_cast<unsigned long long*>(&sh.n_raw), (uint32_t)cap_raw, (uint32_t)cap_ang};
    const uint32_t bits_lds = FusedLds<NT>::hist;
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(FusedLds<NT>::stage + (uint32_t)wave * FusedLds<NT>::stage_wave));
    Moments mo;
    uint32_t n_tissue = 0;
    int t, c0, c1;
    fused_geometry<NT>(nch, tid, t, c0, c1);
    if (c0 >= c1) {                                              // (a half without pixels: wave-uniform)
    } else if (stream) ts_sweep<ALIGNED, kFusedTrip, true>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, K, bits_lds, ring_lds, direct, mo, n_tissue);
    else ts_sweep<ALIGNED, kFusedTrip, false>(src, P, c0, c1, t, kFusedThreads, TB, ylimf, K, bits_lds, ring_lds, direct, mo, n_tissue);
    double v[10];
    mo.to_array(v, n_tissue, lane);
#pragma unroll
    for (int i = 0; i < 10; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0)
        for (int i = 0; i < 10; ++i) sh.red[wave][i] = v[i];
}

// ---- the merged schedule, one launch per phase: k_moments -> k_finish1m -> k_select<merged> -> k_finish2m [-> k_select<conc>,
// k_finish_conc for the rare tile whose exact stain matrix left the assumed box] -> k_apply.  The finish kernels ARE the fused
// kernel's finish steps (fused_finish1 / fused_finish2) on a FusedShared block of their own, with the tile's state carried in
// TileState / TileMerged between the launches: both schedules select the same values by construction.
constexpr int kMFinishThreads = 1024;
static __global__ __launch_bounds__(kMFinishThreads) void k_finish1m(StatsArgs a) {
    __shared__ FusedShared<kMFinishThreads> sh;
    fused_lds_origin(sh);
    const int tile = blockIdx.x, tid = threadIdx.x;
    TileState& st = a.state[tile];
    sh.tab.fill_b();
    if (tid < 10) {                                   // fixed order => run-to-run identical sums
        double t = 0;
        for (int p = 0; p < a.parts; ++p) t += a.partials[((size_t)tile * a.parts + p) * 10 + tid];
        sh.sum[tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        double Vd[6];
        float Vf[6];
        sh.status = eigvecs_from_moments(sh.sum, Vd, Vf);
        for (int i = 0; i < 6; ++i) { sh.Vd[i] = Vd[i]; sh.Vf[i] = Vf[i]; }
        sh.mk.ok = 0;
        sh.xmin = -INFINITY;
    }
    __syncthreads();
    if (sh.status == SL_TILE_OK)                                               // block-uniform
        fused_finish1<kMFinishThreads>(&sh, a.sample + (size_t)tile * a.n_sample, a.n_sample, a.stride_log2, a.P, a.ylimf, a.pct, a.lam, nullptr, 0);
    __syncthreads();
    if (tid == 0) {
        st.status = sh.status;
        st.n_tissue = sh.sum[0];
        for (int i = 0; i < 6; ++i) { st.Vd[i] = sh.Vd[i]; st.Vf[i] = sh.Vf[i]; }
        st.lo[0] = sh.lo[0]; st.hi[0] = sh.hi[0]; st.lo[1] = sh.lo[1]; st.hi[1] = sh.hi[1];
        st.fallbacks = 0;
        st.n_raw = 0; st.overflow = 0;
        TileMerged& tm = a.mstate[tile];
        tm.mk = sh.mk;
        tm.xmin = sh.xmin;
        tm.conc_done = 0;
    }
}

static __global__ __launch_bounds__(kMFinishThreads) void k_finish2m(StatsArgs a, double* M_out, double* maxC_out, int32_t* status_out,
                                                                    int32_t* fallbacks_out, int tile0) {
    __shared__ FusedShared<kMFinishThreads> sh;
    fused_lds_origin(sh);
    const int tile = blockIdx.x, tid = threadIdx.x;
    TileState& st = a.state[tile];
    TileMerged& tm = a.mstate[tile];
    const bool bad = st.status != SL_TILE_OK;                                  // block-uniform
    int fallbacks = 0;
    if (!bad) {
        sh.tab.fill_b();
        if (tid == 0) {
            sh.status = SL_TILE_OK;
            sh.sum[0] = st.n_tissue;
            for (int i = 0; i < 6; ++i) { sh.Vd[i] = st.Vd[i]; sh.Vf[i] = st.Vf[i]; }
            sh.lo[0] = st.lo[0]; sh.hi[0] = st.hi[0]; sh.lo[1] = st.lo[1]; sh.hi[1] = st.hi[1];
            sh.n_raw = st.n_raw; sh.overflow = st.overflow;
            sh.mk = tm.mk;
            sh.conc_done = 0;
            sh.maxC[0] = sh.maxC[1] = nan_d();
        }
        __syncthreads();
        fallbacks = fused_finish2<kMFinishThreads>(&sh, a.rgb + (size_t)tile * a.P * 3, a.raw + (size_t)tile * a.cap_raw, nullptr,
                                                   a.cand + ((size_t)tile * 2 + 0) * a.cap_list, a.cand + ((size_t)tile * 2 + 1) * a.cap_list, a.P,
                                                   a.cap_raw, 0, a.cap_list, a.ylimf, a.pct, a.lam, nullptr);
        __syncthreads();
        const bool singular = sh.status == SL_TILE_DEGENERATE_COV;             // block-uniform
        const bool settled = sh.conc_done != 0 || singular;                    // nothing left for the concentration stage to do
        if (tid == 0) {
            st.status = sh.status;
            for (int i = 0; i < 6; ++i) st.M[i] = singular ? nan_d() : sh.M[i];
            st.maxC[0] = (singular || !sh.conc_done) ? nan_d() : sh.maxC[0];
            st.maxC[1] = (singular || !sh.conc_done) ? nan_d() : sh.maxC[1];
            st.fallbacks += fallbacks;
            st.n_raw = 0; st.overflow = 0;
            tm.conc_done = settled ? 1 : 0;
        }
        if (!settled) {
            // the exact matrix left the box the sweep assumed (or a bracket missed): brackets for the separate concentration sweep,
            // as k_finish_angle leaves them
            if (tid == 0) { LassoK L; lasso_consts(sh.M, a.lam, L); sh.L = L; }
            __syncthreads();
            SampleConcKey ckey;
            ckey.sample = a.sample + (size_t)tile * a.n_sample; ckey.tab = view_of_b(sh.tab); ckey.L = sh.L; ckey.cps_log2 = a.stride_log2 - 2;
            ckey.P = a.P; ckey.col = 0;
            float lo[2], hi[2];
            conc_brackets<kMFinishThreads>(ckey, a.n_sample, lo, hi, sh.S);
            if (tid == 0) { st.lo[0] = lo[0]; st.hi[0] = hi[0]; st.lo[1] = lo[1]; st.hi[1] = hi[1]; }
            return;                                                            // k_select<conc> / k_finish_conc take it from here
        }
    } else if (tid == 0) {
        for (int i = 0; i < 6; ++i) st.M[i] = nan_d();
        st.maxC[0] = st.maxC[1] = nan_d();
        tm.conc_done = 1;
    }
    __syncthreads();
    if (tid < 6 && M_out) M_out[(size_t)(tile0 + tile) * 6 + tid] = st.M[tid];
    if (tid < 2 && maxC_out) maxC_out[(size_t)(tile0 + tile) * 2 + tid] = st.maxC[tid];
    if (tid == 0 && status_out) status_out[tile0 + tile] = st.status;
    if (tid == 0 && fallbacks_out) fallbacks_out[tile0 + tile] = bad ? 0 : st.fallbacks;
}

// The separate concentration pass of a tile whose merged sweep did not settle maxC (Macenko: rare -- SL_RESWEEP_*; Vahadane: always):
// concentration brackets under the tile's M from the sample, sweep 3 (concentration select), finish 3 (the exact 99th percentiles ->
// sh.maxC).  Out of line: inlined, its fully unrolled sample loops left three dozen per-thread addresses live across the whole tile
// loop of k_fused -- 79 scratch stores per lane in the kernel's prologue.  stride_log2 < 0: the cluster sample of the two-sweep
// schedule.  Returns the fallbacks of the order statistics.
template <int NT, bool ALIGNED>
__device__ __noinline__ int fused_conc_resweep(FusedShared<NT>* shp, const uint8_t* src_, uint32_t* samp_, uint32_t* rawl_, float* cand0_, float* cand1_, int P_,
                                               int cap_raw_, int cap_list_, float ylimf_, double lam_, int stream_, int n_sample_, int stride_log2_,
                                               long long* clk_, long long* subclk_) {
    FusedShared<NT>& sh = *shp;
    const uint8_t* src = uni_ptr(src_);
    uint32_t* samp = uni_ptr(samp_);
    uint32_t* rawl = uni_ptr(rawl_);
    float* cand0 = uni_ptr(cand0_);
    float* cand1 = uni_ptr(cand1_);
    const int P = __builtin_amdgcn_readfirstlane(P_), cap_raw = __builtin_amdgcn_readfirstlane(cap_raw_), cap_list = __builtin_amdgcn_readfirstlane(cap_list_);
    const int stream = __builtin_amdgcn_readfirstlane(stream_), n_sample = __builtin_amdgcn_readfirstlane(n_sample_);
    const int stride_log2 = __builtin_amdgcn_readfirstlane(stride_log2_);
    const float ylimf = uni(ylimf_);
    const double lam = uni_d(lam_);
    long long* clk = uni_ptr(clk_);
    long long* subclk = uni_ptr(subclk_);
    const int tid = threadIdx.x;
    int fallbacks = 0;
#ifdef SL_DEBUG_SUBCLK
#define SL_SUB(j) { __syncthreads(); if (subclk && tid == 0) subclk[(j)] = wall_clock64(); }
#else
#define SL_SUB(j)
    (void)subclk;
#endif
    if (tid == 0) {
        LassoK L;
        lasso_consts(sh.M, lam, L);
        sh.L = L;
        sh.n_raw = 0; sh.overflow = 0;
    }
    __syncthreads();
    SL_SUB(7);
    {
        const bool dense = stride_log2 < 0;                            // block-uniform
        SampleConcKey ckey;
        ckey.sample = samp; ckey.tab = view_of_b(sh.tab); ckey.L = sh.L; ckey.cps_log2 = dense ? kDenseCps : stride_log2 - 2;
        ckey.P = P; ckey.col = 0;
        float lo[2], hi[2];
        conc_brackets<NT>(ckey, n_sample, lo, hi, sh.S, dense ? (float)sqrt(kClusterDeff) : 1.0f);
        if (tid == 0) { sh.lo[0] = lo[0]; sh.hi[0] = hi[0]; sh.lo[1] = lo[1]; sh.hi[1] = hi[1]; }
        __syncthreads();
    }
    if (clk && tid == 0) clk[4] = wall_clock64();
    fused_select<NT, ALIGNED>(shp, src, rawl, P, cap_raw, ylimf, stream, 0);
    __threadfence_block();
    __syncthreads();
    if (clk && tid == 0) clk[5] = wall_clock64();
    {
        long long k;
        double gfrac;
        percentile_pos((double)P, 99.0, k, gfrac);
        ConcTileKey tkey;
        tkey.src = src; tkey.tab = view_of_b(sh.tab); tkey.L = sh.L;
        RawConcKey2 rkey;
        rkey.raw = rawl; rkey.tab = view_of_b(sh.tab); rkey.L = sh.L;
        const bool complete = sh.n_raw <= (uint32_t)cap_raw && sh.overflow == 0;
        const uint32_t n_raw = sh.n_raw < (uint32_t)cap_raw ? sh.n_raw : (uint32_t)cap_raw;
        const float los[2] = {sh.lo[0], sh.lo[1]}, his[2] = {sh.hi[0], sh.hi[1]};
        const long long n_plain = (long long)P - (long long)sh.n_raw;        // plain = pixels not collected
        uint32_t n_lt[2], n_in[2];
        SL_SUB(8);
        wg_refine((int)n_raw, rkey, los, his, cand0, cand1, (uint32_t)cap_list, n_lt, n_in, sh.S);
        SL_SUB(9);
        for (int col = 0; col < 2; ++col) {
            tkey.col = col;
            float xa, xb;
            stage_order_stats(col ? cand1 : cand0, n_in[col], (uint32_t)cap_list, complete, los[col], his[col], n_plain + n_lt[col], P,
                              tkey, (uint32_t)P, k, xa, xb, fallbacks, sh.S);
            if (tid == 0) { sh.res[2 * col] = xa; sh.res[2 * col + 1] = xb; }
            __syncthreads();
            SL_SUB(10 + col);
        }
        if (tid == 0) {
            sh.maxC[0] = np_lerp((double)sh.res[0], (double)sh.res[1], gfrac);   // normalizer.py:36,47
            sh.maxC[1] = np_lerp((double)sh.res[2], (double)sh.res[3], gfrac);
            if (!(sh.maxC[0] > 0.0) || !(sh.maxC[1] > 0.0)) sh.status = SL_TILE_ZERO_MAXC;
        }
        __syncthreads();
    }
#undef SL_SUB
    return fallbacks;
}

enum { kMethodMacenko = 0, kMethodVahadane = 1 };

// NT = 512: two workgroups per CU (the throughput configuration).  NT = 1024 (Macenko, round 4): one workgroup per CU for batches of
// no more tiles than CUs -- the sweeps run as fast as two half-size workgroups would, the finish steps with twice the threads and
// nothing to hide behind; between ~190 and 256 tiles of 1024^2 that beats one launch per phase (fused_geometry keeps the sums identical).
template <int METHOD, bool TRANSFORM, bool ALIGNED, int NT>
static __global__ __launch_bounds__(NT, 4) void k_fused(FusedArgs a) {
    __shared__ FusedShared<NT> sh;
    fused_lds_origin(sh);
    [[maybe_unused]] const int tid = threadIdx.x;      // (the development macros; the glue below reads lane_id())
    const TabReaderB TB = TabReaderB::make(sh.tab);       // the 8-byte {gamma, od32} rows serve every sweep
    const int nch = (a.P + 3) >> 2;
    uint32_t* samp = a.sample + (size_t)blockIdx.x * a.sample_cap;
    uint32_t* rawl = a.raw + (size_t)blockIdx.x * a.cap_raw;
    uint32_t* rawa = a.raw_ang + (size_t)blockIdx.x * a.cap_ang;
    float* cand0 = a.cand + ((size_t)blockIdx.x * 2 + 0) * a.cap_list;
    float* cand1 = a.cand + ((size_t)blockIdx.x * 2 + 1) * a.cap_list;

    const bool stream = (size_t)a.P * 3 >= kStreamBytes;       // non-temporal tile accesses (uniform; see kStreamBytes)
    // sweeps 2/3: out of line (fused_select / fused_sweep2_cube); plain count and raw candidates into sh.*
    auto run_select = [&](bool merged, const uint8_t* src) {
        if (merged && (sh.use_cube & 1)) fused_sweep2_cube<NT, ALIGNED>(&sh, src, rawl, rawa, a.P, a.cap_raw, a.cap_ang, a.ylimf, stream ? 1 : 0);   // block-uniform
        else fused_select<NT, ALIGNED>(&sh, src, rawl, a.P, a.cap_raw, a.ylimf, stream ? 1 : 0, merged ? 1 : 0);
        __threadfence_block();
        __syncthreads();
    };

    // (Two workgroups live on a CU and the instruction arbiter serves the OLDER one's waves first: the first-launched workgroup of every CU
    //  runs its sweeps ~20 % faster than its partner and the launch ends when the slow half does.  s_setprio schemes -- finish steps on
    //  top, sweeps alternating or by age -- were measured in rounds 1, 4 and 5 and changed nothing: the tail is bound by bytes)
    // Tiles: the first by position, the later rounds from the launch's counter -- a workgroup whose tiles were quick (empty, background)
    // takes more of them (a batch of 2 048 tiles 512^2, every fourth one white: 2.15 -> 1.88 ms); a tile's results do not depend on who
    // computes them.  The counter is asked before the apply sweep, whose length hides the answer's way back; sh.next_tile is read after
    // the barrier that ends the tile.
    // (the thread's index, read afresh wherever the glue between the phases needs it: from one `tid` the compiler derived three dozen
    //  per-thread addresses before the tile loop and kept them in scratch across it -- 86 stores per lane, 0.56 KB, in a kernel whose
    //  scratch is written back to HBM once per tile)
    auto lane_id = []() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; };
    int ts_backoff = 0;                 // tiles this workgroup still skips the two-sweep attempt on (automatic mode; uniform)
    for (int tile = blockIdx.x; tile < a.n_tiles; tile = __builtin_amdgcn_readfirstlane(sh.next_tile)) {
        const size_t nbytes = (size_t)a.P * 3;
        const uint8_t* src = a.rgb + (size_t)tile * nbytes;
        int fallbacks = 0;
        int sweeps_used = 0;
#ifdef SL_DEVTOOLS
// (Macenko only: in k_fused<vahadane, transform, unaligned> the extra `continue` edges run into the hipcc bug described in the Makefile)
#define SL_PHASE(i) { if (METHOD == kMethodMacenko) { if (a.phase_clock && tid == 0) a.phase_clock[(size_t)tile * 8 + (i)] = wall_clock64(); if (a.debug_stop == (i) + 1) { if (tid == 0) sh.next_tile = tile + (int)gridDim.x; __syncthreads(); continue; } } }
#else
#define SL_PHASE(i)
#endif
#ifdef SL_DEBUG_SUBCLK
#define SL_SUB(j) { __syncthreads(); if (a.phase_clock && tid == 0) a.phase_clock[(size_t)a.n_tiles * 8 + (size_t)tile * 16 + (j)] = wall_clock64(); }
#else
#define SL_SUB(j)
#endif
        SL_PHASE(0);
        if (tile == (int)blockIdx.x) sh.tab.fill_b();       // the row table: written once, before the workgroup's first tile
        __syncthreads();

        if constexpr (METHOD == kMethodMacenko) {
            if (lane_id() == 0) {
                sh.n_raw = 0; sh.n_ang = 0; sh.overflow = 0;
                sh.conc_done = 0;
                sh.use_cube = 0;
                sh.ts.ok = 0; sh.ts.why = kTsOff; sh.ts.dense = 0;
            }
            __syncthreads();
            // ---------------- two-sweep schedule, phase 0: the cluster sample and everything the merged sweep needs (stats_twosweep.hpp)
            // Every workgroup of a launch starts at the same moment, and phase 0 streams nothing: its length is paid in full on a workgroup's
            // first tile (measured, interleaved: with the 290 us phase 0 of the first version the whole-batch route gained 0-6 % and a tile
            // that DECLINED lost 10-13 %; letting only the workgroup launched second on its CU try gave -3 % / +1-3 %;
            // with phase 0 at ~130 us every workgroup tries: -6.5 ... -7.4 % on tiles that take the route, +3 % on real tissue that leaves
            // after the eigen-solve, +7 % on the spatially smooth synthetic tiles that decline at its end).
            // Automatic mode backs off: a workgroup whose tile DECLINED the route in phase 0 (no estimate within the tilt limit, too many ambiguous
            // cells, lists predicted to overflow) does not try on its next kTsBackoff tiles -- the tiles of a batch are mostly of a kind, and a
            // declined attempt costs 50-160 us (real tissue, 2 048 tiles of 512^2: +9 % against three sweeps without the back-off).
            const bool ts_try = a.two_sweep >= 2 || (a.two_sweep == 1 && ts_backoff == 0);   // block-uniform (both workgroup sizes: the 1024-thread kernel gains 5-8 % at 192-256 tiles)
            if (ts_try) {
                fused_phase0<NT>(&sh, src, samp, a.P, a.cl_lines, a.ylimf, a.pct, a.lam, a.two_sweep, a.cap_raw, a.cap_ang, a.cap_list,
#ifdef SL_DEBUG_SUBCLK
                                 // (development: a third region of the clock buffer, only for the tool that allocates it: sl_debug_set_stop(-7))
                                 (a.phase_clock && a.debug_stop == -7) ? a.phase_clock + (size_t)a.n_tiles * 24 + (size_t)tile * 16 : nullptr
#else
                                 nullptr
#endif
                                 );
                __syncthreads();
            }
            const bool ts_on = sh.ts.ok != 0;                                 // block-uniform
            if (a.two_sweep == 1) ts_backoff = ts_try ? (ts_on ? 0 : kTsBackoff) : ts_backoff - 1;
            if (ts_try) { SL_PHASE(4); }      // (slot 4 is otherwise written on the resweep path only: the end of phase 0)
            // ---------------- sweep 1: moments (+ sample, or + the candidates of all four order statistics)
            if (ts_on) fused_sweep1c<NT, ALIGNED>(&sh, src, rawl, rawa, a.P, a.cap_raw, a.cap_ang, a.ylimf, stream ? 1 : 0);
            else fused_sweep1<NT, ALIGNED>(&sh, src, samp, a.P, a.ylimf, a.stride_log2, stream ? 1 : 0);   // (a declined tile: the stratified sample replaces the cluster sample)
            if (ts_on && lane_id() == 0) sh.ts.dense = 1;
            __threadfence_block();
            __syncthreads();
            if (lane_id() < 10) {
                double t = 0;
                for (int w = 0; w < NT / 64; ++w) t += sh.red[w][lane_id()];
                sh.sum[lane_id()] = t;
            }
            __syncthreads();
            SL_PHASE(1);
            // ---------------- finish 1: eigenvectors, angle brackets
            SL_SUB(0);
            if (lane_id() == 0) {
                double Vd[6];
                float Vf[6];
                sh.status = eigvecs_from_moments(sh.sum, Vd, Vf);
                for (int i = 0; i < 6; ++i) { sh.Vd[i] = Vd[i]; sh.Vf[i] = Vf[i]; }
                sh.conc_done = 0;
            }
            __syncthreads();
            SL_SUB(1);
            bool direct = false;                                              // block-uniform: the two-sweep route settled the tile's M
            if (sh.status == SL_TILE_OK && ts_on) {                           // block-uniform
                // ---------------- two-sweep finish: do the half-spaces the sweep tested against hold for the exact eigenvectors?
                if (lane_id() < 64) {
                    float br[4];
                    const bool okv = ts_verify(sh.ts, sh.Vd, br, lane_id(), a.two_sweep == 3);
                    if (lane_id() == 0) {
                        sh.lo[0] = br[0]; sh.hi[0] = br[1]; sh.lo[1] = br[2]; sh.hi[1] = br[3];
                        if (!okv) { sh.ts.ok = 0; sh.ts.why = kTsPlane; }
                    }
                }
                __syncthreads();
                if (sh.ts.ok) {                                               // block-uniform
                    const int fb = fused_finish2<NT>(&sh, src, rawl, rawa, cand0, cand1, a.P, a.cap_raw, a.cap_ang, a.cap_list, a.ylimf, a.pct, a.lam,
#ifdef SL_DEBUG_SUBCLK
                                                     a.phase_clock ? a.phase_clock + (size_t)a.n_tiles * 8 + (size_t)tile * 16 : nullptr
#else
                                                     nullptr
#endif
                                                     , 1);
                    if (fb >= 0) { fallbacks += fb; direct = true; }
                    else if (lane_id() == 0) { sh.ts.ok = 0; sh.ts.why = kTsBracket; }
                }
                if (!direct) {                                                // the three-sweep route from the exact moments
                    if (lane_id() == 0) {
                        sh.n_raw = 0; sh.n_ang = 0; sh.overflow = 0;
                        sh.conc_done = 0;
                        sh.use_cube = 0;
                    }
                    __syncthreads();
                }
                SL_PHASE(2);
                SL_PHASE(3);
            }
            if (sh.status == SL_TILE_OK && !direct) {                         // block-uniform
                // ---------------- finish 1, the rest of it (out of line like finish 2): angle brackets, the box of stain matrices the
                // sample leaves possible, concentration brackets under its centre
                fused_finish1<NT>(&sh, samp, ts_on ? a.cl_lines * kClusterPx : a.n_sample, ts_on ? -a.cl_scale_log2 : a.stride_log2, a.P, a.ylimf, a.pct, a.lam,
#ifdef SL_DEBUG_SUBCLK
                                  a.phase_clock ? a.phase_clock + (size_t)a.n_tiles * 8 + (size_t)tile * 16 : nullptr
#else
                                  nullptr
#endif
                                  , a.use_cube);
                SL_PHASE(2);
                // ---------------- sweep 2: angle select + concentration select under the box
                run_select(true, src);
                SL_PHASE(3);
                // ---------------- finish 2 (out of line: its registers are allocated apart from the sweeps'): exact angular percentiles
                // -> M, then the concentration percentiles -> maxC from the same raw list
                fallbacks += fused_finish2<NT>(&sh, src, rawl, (sh.use_cube & 1) ? rawa : nullptr, cand0, cand1, a.P, a.cap_raw, a.cap_ang, a.cap_list, a.ylimf,
                                               a.pct, a.lam,
#ifdef SL_DEBUG_SUBCLK
                                               a.phase_clock ? a.phase_clock + (size_t)a.n_tiles * 8 + (size_t)tile * 16 : nullptr
#else
                                               nullptr
#endif
                                               );
            }
        } else {
            // ---------------- Vahadane: class-moment dictionary learning
            gather_sample<ALIGNED>(src, a.P, a.stride_log2, samp, a.n_sample, lane_id(), NT);
            if (lane_id() == 0) {
                dict_iter_init(sh.it);
                sh.n_raw = 0; sh.overflow = 0;
                sh.conc_done = 0;
            }
            __syncthreads();
            DictProgress pr{1, 0, 0, 0};
            if (stream) dict_learn<ALIGNED, NT, false, true>(src, a.P, nch, lane_id(), TB, a.ylimf, a.stride_log2, samp, a.n_sample, a.dl_lambda,
                                                             a.dl_tol, a.dl_max_sweeps, sh.it, sh.red, sh.sum, pr);
            else dict_learn<ALIGNED, NT, false, false>(src, a.P, nch, lane_id(), TB, a.ylimf, a.stride_log2, samp, a.n_sample, a.dl_lambda,
                                                       a.dl_tol, a.dl_max_sweeps, sh.it, sh.red, sh.sum, pr);
            sweeps_used = pr.sweeps_used;
            if (lane_id() == 0) {
                sh.status = sh.it.status;
                if (sh.status == SL_TILE_OK) {
                    dict_iter_stain_matrix(sh.it, sh.M);
                    if (stain_matrix_singular(sh.M)) sh.status = SL_TILE_DEGENERATE_COV;
                }
            }
        }
        __syncthreads();
        const bool bad = sh.status != SL_TILE_OK;                               // block-uniform
        const bool resweep = !bad && !sh.conc_done;                             // block-uniform: sweep 3 of the four-sweep schedule
        if (resweep) {
            // ---------------- concentration brackets from the sample, sweep 3 (concentration select), finish 3 (exact 99th percentiles -> maxC)
            fallbacks += fused_conc_resweep<NT, ALIGNED>(&sh, src, samp, rawl, cand0, cand1, a.P, a.cap_raw, a.cap_list, a.ylimf, a.lam, stream ? 1 : 0,
                                                         (METHOD == kMethodMacenko && sh.ts.dense != 0) ? a.cl_lines * kClusterPx : a.n_sample,
                                                         (METHOD == kMethodMacenko && sh.ts.dense != 0) ? -1 : a.stride_log2,
#ifdef SL_DEVTOOLS
                                                         (METHOD == kMethodMacenko && a.phase_clock) ? a.phase_clock + (size_t)tile * 8 : nullptr,
#else
                                                         nullptr,
#endif
#ifdef SL_DEBUG_SUBCLK
                                                         a.phase_clock ? a.phase_clock + (size_t)a.n_tiles * 8 + (size_t)tile * 16 : nullptr
#else
                                                         nullptr
#endif
                                                         );
        } else if (bad && sh.status != SL_TILE_ZERO_MAXC && lane_id() == 0) {     // (a zero maxC keeps its M and maxC, as after finish 3)
            for (int i = 0; i < 6; ++i) sh.M[i] = nan_d();
            sh.maxC[0] = sh.maxC[1] = nan_d();
        }
        __syncthreads();
        if (lane_id() == 0) sh.next_tile = a.n_tiles <= (int)gridDim.x ? a.n_tiles : a.next_tile ? (int)gridDim.x + (int)atomicAdd(a.next_tile, 1ull) : tile + (int)gridDim.x;
        if (lane_id() == 0 && a.resweep_out) a.resweep_out[tile] = (METHOD == kMethodMacenko && resweep) ? (sh.why ? sh.why : SL_RESWEEP_NO_BOX) : 0;
        if (lane_id() < 6 && a.M_out) a.M_out[(size_t)tile * 6 + lane_id()] = sh.M[lane_id()];
        if (lane_id() < 2 && a.maxC_out) a.maxC_out[(size_t)tile * 2 + lane_id()] = sh.maxC[lane_id()];
        if (lane_id() == 0 && a.status_out) a.status_out[tile] = sh.status;
        if (lane_id() == 0 && a.diag_out) a.diag_out[tile] = fallbacks;
        if (lane_id() == 0 && a.sweeps_out) a.sweeps_out[tile] = sweeps_used;
        if (METHOD == kMethodMacenko && lane_id() == 0 && a.cube_out) a.cube_out[tile] = (sh.status == SL_TILE_OK || sh.status == SL_TILE_ZERO_MAXC) ? sh.use_cube : 0;
        if (METHOD == kMethodMacenko && lane_id() == 0 && a.ts_out) a.ts_out[tile] = sh.ts.why;
        SL_PHASE(6);
        // ---------------- sweep 4: apply
        if (TRANSFORM) {
            uint8_t* dst = a.out + (size_t)tile * nbytes;
            if (sh.status != SL_TILE_OK) {       // block-uniform (sh.status is final: barrier above); includes a zero maxC found in finish 3
                for (int c = lane_id(); c < nch; c += 8 * NT) {          // eight chunks in flight per lane (one at a time: 340 us per Mpixel tile; now ~60)
                    Chunk ch[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) ch[k] = load_chunk_clamped<ALIGNED, false>(src, nbytes, c + k * NT, nch);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (c + k * NT < nch) store_chunk<ALIGNED>(dst, nbytes, c + k * NT, ch[k]);
                }
            } else {
                fused_apply<NT, ALIGNED>(&sh, src, dst, a.P, a.M_tgt, a.maxC_tgt, a.lam, stream ? 1 : 0);
            }
        }
        __syncthreads();     // sh.* is reused by the next tile
        SL_PHASE(7);
#undef SL_PHASE
#undef SL_SUB
    }
}

// host-side launchers of the twelve instantiations, one translation unit per family (transform: with the apply sweep; aligned: 4-byte
// aligned tiles of a multiple of four pixels)
void launch_fused_macenko(const FusedArgs& a, bool transform, bool aligned, unsigned grid, hipStream_t s);
void launch_fused_macenko_wide(const FusedArgs& a, bool transform, bool aligned, unsigned grid, hipStream_t s);
void launch_fused_vahadane(const FusedArgs& a, bool transform, bool aligned, unsigned grid, hipStream_t s);

}  // namespace sl
