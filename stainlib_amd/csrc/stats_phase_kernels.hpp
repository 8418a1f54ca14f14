// stats_phase_kernels.hpp -- one launch per phase: k_moments, k_select, k_finish_conc, k_dict*.
// Part of stats_kernels.hpp (split by phase in round 4, no functional change); include that umbrella, not this file.
#pragma once
#include "stats_dict.hpp"

namespace sl {

// ------------------------------------------------------------------------------------------
// multi-kernel schedule
// ------------------------------------------------------------------------------------------
// chunk range of part `part` of a tile: spans are multiples of one sweep trip of a workgroup (kSweepThreads x kPhaseTrip =
// 2048 chunks), so that every wave row is 64-aligned AND every lane's trips cover the same pixels as in the fused kernel
// (the binary32 burst sums of moments_sweep_b are then identical in both schedules); trailing parts may be empty
__device__ __forceinline__ void part_range(int nch, int parts, int part, int& c0, int& c1, int align_trips = 1) {
    const int kAlign = kSweepThreads * kPhaseTrip * align_trips;     // (the dictionary sweeps sum over kDictBurstTrips trips)
    const int span = (((nch + parts - 1) / parts) + kAlign - 1) / kAlign * kAlign;
    c0 = min(nch, part * span);
    c1 = min(nch, c0 + span);
}

// The sweep kernels of this schedule are persistent too: at most 2 workgroups per CU, each filling its 64 KB table
// once and then walking (tile, part) items blockIdx.x, +gridDim.x, ...  (StatsArgs.n_items = tiles x parts).
template <bool ALIGNED>
static __global__ __launch_bounds__(kSweepThreads, 4) void k_moments(StatsArgs a) {
    __shared__ RowTab s_tab;
    __shared__ double s_red[kSweepThreads / 64][10];
    s_tab.fill_b();
    __syncthreads();
    const TabReaderB T = TabReaderB::make(s_tab);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int tile = item / a.parts, part = item % a.parts;
        const uint8_t* src = a.rgb + (size_t)tile * a.P * 3;
        uint32_t* samp = a.sample + (size_t)tile * a.n_sample;
        int c0, c1;
        part_range((a.P + 3) >> 2, a.parts, part, c0, c1);
        Moments mo;
        uint32_t n_tissue = 0;
        if (c0 >= c1) {                            // an empty trailing part (block-uniform): its partial sums are zeros
        } else if ((size_t)a.P * 3 >= kStreamBytes)      // uniform: non-temporal tile loads for big tiles (see kStreamBytes)
            moments_sweep_b<ALIGNED, kPhaseTrip, true>(src, a.P, c0, c1, tid, kSweepThreads, T, a.ylimf, a.stride_log2, samp, mo, n_tissue);
        else
            moments_sweep_b<ALIGNED, kPhaseTrip, false>(src, a.P, c0, c1, tid, kSweepThreads, T, a.ylimf, a.stride_log2, samp, mo, n_tissue);
        double v[10];
        mo.to_array(v, n_tissue, lane);
#pragma unroll
        for (int i = 0; i < 10; ++i) v[i] = wave_sum(v[i]);
        if (lane == 0)
            for (int i = 0; i < 10; ++i) s_red[tid >> 6][i] = v[i];
        __syncthreads();
        if (tid < 10) {
            double t = 0;
            for (int w = 0; w < kSweepThreads / 64; ++w) t += s_red[w][tid];
            a.partials[((size_t)tile * a.parts + part) * 10 + tid] = t;
        }
        __syncthreads();                         // s_red is reused by the next item
    }
}

template <int STAGE, bool ALIGNED>
static __global__ __launch_bounds__(kSweepThreads, 4) void k_select(StatsArgs a) {
    __shared__ RowTab s_tab;
    __shared__ uint32_t s_stage[kSweepThreads / 64][kStageWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (STAGE == kStageConc && a.mstate) {   // merged Macenko schedule: normally every tile is settled already -- leave before the table is built
        bool any = false;
        for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
            const int tile = item / a.parts;
            any = any | (a.state[tile].status == SL_TILE_OK && !a.mstate[tile].conc_done);
        }
        if (!any) return;                    // block-uniform
    }
    s_tab.fill_b();
    __syncthreads();
    const TabReaderB T = TabReaderB::make(s_tab);
    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int tile = item / a.parts, part = item % a.parts;
        TileState& st = a.state[tile];
        if (st.status != SL_TILE_OK) continue;                             // block-uniform
        if (STAGE == kStageConc && a.mstate && a.mstate[tile].conc_done) continue;      // (merged schedule: the tile's maxC is settled)
        SelConsts K;
        K.xmin = -INFINITY;
        if (STAGE == kStageMerged) {
            const TileMerged& tm = a.mstate[tile];
            for (int i = 0; i < 6; ++i) K.V[i] = in_vgpr(st.Vf[i]);
            K.L.g12 = 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                K.u[i][0] = in_vgpr(tm.mk.u[i][0]); K.u[i][1] = in_vgpr(tm.mk.u[i][1]); K.kt[i] = in_vgpr(tm.mk.kt[i]);
                K.eps[i] = in_vgpr(tm.mk.eps[i]); K.thr[i] = in_vgpr(tm.mk.thr[i]);
            }
            K.xmin = uni(tm.xmin);
        } else {
            lasso_consts(st.M, a.lam, K.L);
            vgpr(K.L);
        }
        K.lo0 = uni(st.lo[0]); K.hi0 = uni(st.hi[0]); K.lo1 = uni(st.lo[1]); K.hi1 = uni(st.hi[1]);
        const uint8_t* src = a.rgb + (size_t)tile * a.P * 3;
        int c0, c1;
        part_range((a.P + 3) >> 2, a.parts, part, c0, c1);
        RawSink sink{(uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(s_stage[wave])), 0u, a.raw + (size_t)tile * a.cap_raw, &st.n_raw, &st.overflow, (uint32_t)a.cap_raw, (uint32_t)kStageWave};
        const bool stream = (size_t)a.P * 3 >= kStreamBytes;
        if (STAGE == kStageMerged && K.xmin > -INFINITY) {                   // block-uniform: the projection bound stands in for the tissue test
            if (stream) select_sweep<kStageMerged, ALIGNED, kPhaseTrip, true, true>(src, a.P, c0, c1, tid, kSweepThreads, T, a.ylimf, K, sink);
            else select_sweep<kStageMerged, ALIGNED, kPhaseTrip, false, true>(src, a.P, c0, c1, tid, kSweepThreads, T, a.ylimf, K, sink);
        } else {
            if (stream) select_sweep<STAGE, ALIGNED, kPhaseTrip, true>(src, a.P, c0, c1, tid, kSweepThreads, T, a.ylimf, K, sink);
            else select_sweep<STAGE, ALIGNED, kPhaseTrip, false>(src, a.P, c0, c1, tid, kSweepThreads, T, a.ylimf, K, sink);
        }
        sink.flush(lane);
    }
}

static __global__ SL_FINISH_BOUNDS void k_finish_conc(StatsArgs a, double* M_out, double* maxC_out,
                                                                       int32_t* status_out, int32_t* fallbacks_out, int tile0) {
    __shared__ SmallTab s_tab;
    __shared__ SelScratch S;
    __shared__ float s_res[4];
    __shared__ LassoK s_L;
    const int tile = blockIdx.x, tid = threadIdx.x;
    TileState& st = a.state[tile];
    if (a.mstate && a.mstate[tile].conc_done) return;             // block-uniform: settled (and written out) by k_finish2m
    const bool bad = st.status != SL_TILE_OK;
    if (!bad) {
        s_tab.fill();
        if (tid == 0) { LassoK L; lasso_consts(st.M, a.lam, L); s_L = L; }
        __syncthreads();
        long long k;
        double gfrac;
        percentile_pos((double)a.P, 99.0, k, gfrac);
        int fallbacks = 0;
        ConcTileKey tkey;
        tkey.src = a.rgb + (size_t)tile * a.P * 3;
        tkey.tab = view_of(s_tab);
        tkey.L = s_L;
        RawConcKey2 rkey;
        rkey.raw = a.raw + (size_t)tile * a.cap_raw; rkey.tab = view_of(s_tab); rkey.L = s_L;
        const bool complete = st.n_raw <= (uint32_t)a.cap_raw && st.overflow == 0;
        const uint32_t n_raw = st.n_raw < (uint32_t)a.cap_raw ? st.n_raw : (uint32_t)a.cap_raw;
        float* cand0 = a.cand + ((size_t)tile * 2 + 0) * a.cap_list;
        float* cand1 = a.cand + ((size_t)tile * 2 + 1) * a.cap_list;
        const float los[2] = {st.lo[0], st.lo[1]}, his[2] = {st.hi[0], st.hi[1]};
        uint32_t n_lt[2], n_in[2];
        wg_refine((int)n_raw, rkey, los, his, cand0, cand1, (uint32_t)a.cap_list, n_lt, n_in, S);
        for (int col = 0; col < 2; ++col) {
            tkey.col = col;
            float xa, xb;
            stage_order_stats(col ? cand1 : cand0, n_in[col], (uint32_t)a.cap_list, complete, los[col], his[col], (long long)a.P - (long long)st.n_raw + n_lt[col],
                              a.P, tkey, (uint32_t)a.P, k, xa, xb, fallbacks, S);
            if (tid == 0) { s_res[2 * col] = xa; s_res[2 * col + 1] = xb; }
            __syncthreads();
        }
        if (tid == 0) {
            st.maxC[0] = np_lerp((double)s_res[0], (double)s_res[1], gfrac);   // normalizer.py:36,47
            st.maxC[1] = np_lerp((double)s_res[2], (double)s_res[3], gfrac);
            st.fallbacks += fallbacks;
            if (!(st.maxC[0] > 0.0) || !(st.maxC[1] > 0.0)) st.status = SL_TILE_ZERO_MAXC;
        }
    } else if (tid == 0) {
        st.maxC[0] = st.maxC[1] = nan_d();
    }
    __syncthreads();
    if (tid < 6 && M_out) M_out[(size_t)(tile0 + tile) * 6 + tid] = st.M[tid];
    if (tid < 2 && maxC_out) maxC_out[(size_t)(tile0 + tile) * 2 + tid] = st.maxC[tid];
    if (tid == 0 && status_out) status_out[tile0 + tile] = st.status;
    if (tid == 0 && fallbacks_out) fallbacks_out[tile0 + tile] = bad ? 0 : st.fallbacks;
}

// ---- Vahadane, one launch per phase: k_dict<first> + k_dict_finish(first) [the sample stage runs inside it], then a
// fixed number of (k_dict, k_dict_finish) pairs that skip settled tiles, then k_dict_tail: tiles that still move (rare)
// finish on one workgroup each; it also sets up the concentration stage, which then runs the Macenko kernels
// (k_select<kStageConc>, k_finish_conc, apply).
template <int NT>
struct DictScratch {
    double red[NT / 64][32];
    double sum[32];
    DictIter it;
};
struct DictState {
    DictIter it;
    DictProgress pr;
    int done;
    int pad_;
};

template <bool ALIGNED>
static __global__ __launch_bounds__(kSweepThreads, 4) void k_dict(StatsArgs a) {
    __shared__ RowTab s_tab;
    __shared__ double s_red[kSweepThreads / 64][32];
    s_tab.fill_b();
    __syncthreads();
    const TabReaderB T = TabReaderB::make(s_tab);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int tile = item / a.parts, part = item % a.parts;
        const DictState& ds = a.dstate[tile];
        if (ds.done) continue;                                              // block-uniform
        DictK Ld;
        dict_consts(ds.it.D, a.dl_lambda, Ld);
        const uint8_t* src = a.rgb + (size_t)tile * a.P * 3;
        int c0, c1;
        part_range((a.P + 3) >> 2, a.parts, part, c0, c1, kDictAlignTrips);
        DictWaveAcc acc;
        acc.begin(s_red[tid >> 6], lane);
        if (c0 >= c1) {                          // an empty trailing part (block-uniform): zeros
        } else if ((size_t)a.P * 3 >= kStreamBytes) dict_sweep_b<ALIGNED, kDictTrip, true>(src, a.P, c0, c1, tid, kSweepThreads, T, a.ylimf, Ld, acc);
        else dict_sweep_b<ALIGNED, kDictTrip, false>(src, a.P, c0, c1, tid, kSweepThreads, T, a.ylimf, Ld, acc);
        __syncthreads();
        if (tid < 31) {
            double t = 0;
            for (int w = 0; w < kSweepThreads / 64; ++w) t += s_red[w][tid];
            a.partials[((size_t)tile * a.parts + part) * 32 + tid] = t;
        }
        __syncthreads();                         // s_red is reused by the next item
    }
}

__device__ __forceinline__ void dict_finalize(const DictIter& it, TileState& st) {
    st.status = it.status;
    if (it.status == SL_TILE_OK) {
        dict_iter_stain_matrix(it, st.M);
        if (stain_matrix_singular(st.M)) st.status = SL_TILE_DEGENERATE_COV;
    }
    if (st.status != SL_TILE_OK) for (int i = 0; i < 6; ++i) st.M[i] = nan_d();
}

// (512 threads like the sweep kernels: the sample stage and the straggler sweeps then form the same binary32 bursts as the
// fused kernel)
constexpr int kDictFinishThreads = kSweepThreads;
// one workgroup per tile: gather the stratified sample, iterate the dictionary on it from the Ruifrok start
template <bool ALIGNED>
static __global__ __launch_bounds__(kDictFinishThreads) void k_dict_start(StatsArgs a) {
    __shared__ RowTab s_tab;
    __shared__ DictScratch<kDictFinishThreads> s_d;
    const int tile = blockIdx.x, tid = threadIdx.x;
    DictState& ds = a.dstate[tile];
    uint32_t* samp = a.sample + (size_t)tile * a.n_sample;
    s_tab.fill_b();
    gather_sample<ALIGNED>(a.rgb + (size_t)tile * a.P * 3, a.P, a.stride_log2, samp, a.n_sample, tid, kDictFinishThreads);
    if (tid == 0) dict_iter_init(s_d.it);
    __syncthreads();
    DictProgress pr{1, 0, 0, 0};
    const TabReaderB T = TabReaderB::make(s_tab);
    dict_learn<true, kDictFinishThreads, true>(nullptr, a.P, 0, tid, T, a.ylimf, a.stride_log2, samp, a.n_sample, a.dl_lambda, a.dl_tol,
                                           a.dl_max_sweeps, s_d.it, s_d.red, s_d.sum, pr);
    if (tid == 0) {
        const bool go = s_d.it.status == SL_TILE_OK && pr.stage == 2;
        ds.it = s_d.it;
        ds.pr = pr;
        ds.done = go ? 0 : 1;
        if (!go) dict_finalize(s_d.it, a.state[tile]);
    }
}

// one workgroup per tile: the dictionary update from the partial sums of a full-sweep launch
static __global__ __launch_bounds__(kDictFinishThreads) void k_dict_finish(StatsArgs a) {
    __shared__ DictScratch<kDictFinishThreads> s_d;
    const int tile = blockIdx.x, tid = threadIdx.x;
    DictState& ds = a.dstate[tile];
    if (ds.done) return;
    DictProgress pr = ds.pr;
    if (tid == 0) s_d.it = ds.it;
    if (tid < 31) {                                   // fixed order => run-to-run identical sums
        double t = 0;
        for (int p = 0; p < a.parts; ++p) t += a.partials[((size_t)tile * a.parts + p) * 32 + tid];
        s_d.sum[tid] = t;
    }
    __syncthreads();
    if (tid == 0) dict_iter_update(s_d.it, s_d.sum, a.dl_lambda, pr.stage, pr.outer, a.dl_tol);
    __syncthreads();
    const bool go = dict_advance(s_d.it, pr, a.dl_tol, tid) && pr.sweeps_used < a.dl_max_sweeps;
    if (tid == 0) {
        ds.it = s_d.it;
        ds.pr = pr;
        ds.done = go ? 0 : 1;
        if (!go) dict_finalize(s_d.it, a.state[tile]);
    }
}

template <bool ALIGNED>
static __global__ __launch_bounds__(kDictFinishThreads) void k_dict_tail(StatsArgs a) {
    __shared__ RowTab s_tab;
    __shared__ SelScratch S;
    __shared__ DictScratch<kDictFinishThreads> s_d;
    __shared__ LassoK s_L;
    __shared__ int s_status;
    const int tile = blockIdx.x, tid = threadIdx.x;
    DictState& ds = a.dstate[tile];
    TileState& st = a.state[tile];
    s_tab.fill_b();
    __syncthreads();
    if (!ds.done) {                                   // block-uniform: this tile needs more sweeps than the launches gave it
        DictProgress pr = ds.pr;
        if (tid == 0) s_d.it = ds.it;
        __syncthreads();
        const TabReaderB T = TabReaderB::make(s_tab);
        dict_learn<ALIGNED, kDictFinishThreads>(a.rgb + (size_t)tile * a.P * 3, a.P, (a.P + 3) >> 2, tid, T, a.ylimf, a.stride_log2,
                                            a.sample + (size_t)tile * a.n_sample, a.n_sample, a.dl_lambda, a.dl_tol,
                                            a.dl_max_sweeps, s_d.it, s_d.red, s_d.sum, pr);
        if (tid == 0) {
            ds.pr = pr;
            ds.done = 1;
            dict_finalize(s_d.it, st);
        }
        __syncthreads();
    }
    if (tid == 0) {
        st.fallbacks = 0;
        st.n_raw = 0; st.overflow = 0;
        if (a.sweeps_out) a.sweeps_out[a.tile0 + tile] = ds.pr.sweeps_used;
        s_status = st.status;
        if (st.status == SL_TILE_OK) { LassoK L; lasso_consts(st.M, a.lam, L); s_L = L; }
    }
    __syncthreads();
    if (s_status != SL_TILE_OK) return;               // block-uniform
    SampleConcKey ckey;
    ckey.sample = a.sample + (size_t)tile * a.n_sample;
    ckey.tab = view_of_b(s_tab);
    ckey.L = s_L;
    ckey.cps_log2 = a.stride_log2 - 2;
    ckey.P = a.P;
    ckey.col = 0;
    float lo[2], hi[2];
    conc_brackets<kDictFinishThreads>(ckey, a.n_sample, lo, hi, S);
    if (tid == 0) { st.lo[0] = lo[0]; st.hi[0] = hi[0]; st.lo[1] = lo[1]; st.hi[1] = hi[1]; }
}

}  // namespace sl
