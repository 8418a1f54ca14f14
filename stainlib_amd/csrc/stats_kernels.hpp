// stats_kernels.hpp -- per-tile statistics of the Macenko path and the fused transform (gfx950).
//
// Reference chain (macenko_stain_extractor.py:7-44, normalizer.py:34-36,45-50):
//   mask -> OD -> cov(3x3) -> eigh -> project -> arctan2 -> percentile(1,99) -> M
//   -> lasso concentrations of ALL pixels -> percentile(99) per stain -> rescale -> 255 exp(-C M_t)
//
// Four dependent streaming sweeps over the uint8 tile, nothing per-pixel stored in between:
//   sweep 1  moments   tissue test, 9 moment sums (binary32 bursts of one trip per lane, binary64 totals), one
//                      stratified-random sample pixel per `stride` pixels
//   finish 1           cov -> Jacobi eigh -> V ; sample -> brackets [lo,hi] that contain the 1st/99th
//                      angular order statistics with overwhelming probability
//   sweep 2  select    exact counts below / inside each bracket, bracket members ("candidates", ~1 %
//                      of the pixels) collected
//   finish 2           EXACT order statistics k, k+1 among the candidates -> numpy-style linear
//                      interpolation -> stain matrix M ; sample -> brackets for the 99th percentile of
//                      both concentration columns
//   sweep 3  select    same skeleton on the lasso concentrations of all pixels
//   finish 3           exact 99th percentiles -> maxC, status
//   sweep 4  apply     (transform only) OD -> lasso -> rescale -> exp -> truncate -> store
//
// If a bracket misses (probability ~1e-9 per tile) or overflows (heavy ties) the finish step falls
// back to an exact radix selection over the whole tile: results never depend on the sampling, only
// the speed does.  Order statistics are exact on the binary32 keys the sweeps compute
// (pseudo-angle, concentrations); interpolation, trigonometry, eigen-decomposition and moment sums
// are binary64.
//
// Two schedules share every device function below:
//   * k_macenko_fused : persistent kernel, ONE 1024-thread workgroup owns a tile through all phases
//     (phase hand-offs are __syncthreads, the sample lives in LDS, no launch boundaries, no
//     inter-workgroup traffic).  Used for batches large enough to fill the chip.
//   * k_moments / k_finish_* / k_select : one launch per phase with each tile split over several
//     workgroups.  Used for small batches, where a tile per workgroup would leave most CUs idle.
#pragma once
#include "../../include/stainlib_hip.h"
#include <type_traits>

#include "apply_kernels.hpp"

namespace sl {

constexpr int kMaxSample = 16384;   // samples per tile (<= P/64)
constexpr int kMinCapRaw = 65536;   // raw-pixel candidate capacity per tile and stage: max(this, P/6), set by the host
constexpr int kMinCapList = 16384;  // exact-key bracket members per list after the refine pass: max(this, P/8)
#ifdef SL_EXP_FIN512
constexpr int kFinishThreads = 512;
#else
constexpr int kFinishThreads = 1024;
#endif
#ifdef SL_EXP_FIN2
#define SL_FINISH_BOUNDS __launch_bounds__(kFinishThreads, 2)
#else
#define SL_FINISH_BOUNDS __launch_bounds__(kFinishThreads)
#endif
constexpr int kFusedThreads = 512;  // 2 resident workgroups per CU (<=128 VGPRs, <80 KB LDS each)
constexpr int kSweepThreads = 512;  // sweep kernels of the one-launch-per-phase schedule (same occupancy: 64 KB table each)
constexpr int kFusedTrip = 4;       // chunks per lane and sweep trip in the fused kernel (even)
constexpr int kBrkBatch = 8;       // sample words in flight per lane while the bracket keys are evaluated
constexpr int kPhaseTrip = 4;       // ... in the one-sweep kernels (32 Ki-pixel parts: 8 chunks per lane)
constexpr int kStageWave = 256;     // per-wave LDS staging entries for raw candidates (8 KB per 8 waves)
constexpr float kBracketZ = 6.0f;   // bracket half-width in standard deviations of the sample rank
constexpr float kAngleMargin = 2e-5f;  // safety margin of the cheap pseudo-angle test (keys carry ~1e-7)

struct TileState {
    // ---- after finish 1
    double n_tissue;
    double Vd[6];            // V[c][k], c = channel, k = 0 (largest eigenvalue), 1 (second)
    float Vf[6];
    float lo[2], hi[2];      // brackets of the current selection stage
    unsigned int pad0_;
    unsigned int n_raw;      // raw candidates appended (may exceed cap_raw => overflow)
    unsigned int overflow;   // a wave's staging buffer overflowed: the collected list is incomplete
    unsigned int pad_;
    // ---- after finish 2
    double M[6];
    // ---- after finish 3
    double maxC[2];
    int status;
    int fallbacks;           // order statistics that needed the slow exact path (diagnostics)
};

struct StatsArgs {
    const uint8_t* rgb;      // first tile of the group / batch
    int P;
    int parts;               // parts per tile (multi-kernel schedule)
    int n_items;             // tiles x parts of this group: the work list of the persistent sweep kernels
    int stride_log2;         // sampling stride = 1 << stride_log2 (>= 6)
    int n_sample;            // ceil(P / stride)
    float ylimf;             // tissue test threshold: y_lim - 2048 (see is_tissue_f)
    double lam;
    double pct;              // angular percentile
    double* partials;        // [tile][part][10]          (multi-kernel)
    uint32_t* sample;        // [tile][n_sample]          (multi-kernel)
    int cap_raw, cap_list;   // capacities of the two lists below (scale with the tile size)
    uint32_t* raw;           // [tile][cap_raw] raw candidate pixels (r | g<<8 | b<<16)
    float* cand;             // [tile][2][cap_list] bracket members (exact keys)
    TileState* state;        // [tile]                    (multi-kernel)
    // Vahadane (multi-kernel): partials are [tile][part][32] there
    double dl_lambda, dl_tol;
    int dl_max_sweeps;
    int tile0;               // first tile of the group within the batch (sweeps_out index)
    struct DictState* dstate;   // [tile]
    int32_t* sweeps_out;     // [n_tiles of the batch] (may be NULL)
    struct TileMerged* mstate;  // [tile] merged selection stage of the per-phase Macenko schedule
};


// Table access of the finish steps and key functors (few lookups, any layout): entry v of table f / g sits at LDS
// byte address base + v*stride + off_f / off_g (DS reads; a generic pointer would go the slower flat path).
struct TabView {
    uint32_t base; uint32_t stride, off_f, off_g;
#if defined(__HIP_DEVICE_COMPILE__)
    __device__ __forceinline__ float odf(uint32_t v) const { return *(SL_LDS const float*)(base + v * stride + off_f); }
    __device__ __forceinline__ float gam(uint32_t v) const { return *(SL_LDS const float*)(base + v * stride + off_g); }
#else
    float odf(uint32_t) const { return 0.0f; }
    float gam(uint32_t) const { return 0.0f; }
#endif
};
// LDS byte address of a pointer into shared memory: the low half of its flat address (the shared aperture sits in the
// high half).  Not the generic->local cast: that one carries a null check, which this hipcc mis-folds into an illegal
// v_cmp against src_shared_base when the pointer's origin is known.
__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)p; }
__device__ __forceinline__ TabView view_of(const RowTab& t) {
    const uint32_t c = (uint32_t)sizeof(TabEntry) * (threadIdx.x & (kTabCopies - 1));
    return TabView{lds_address(&t), (uint32_t)sizeof(TabEntry) * kTabCopies, c + 12u, c + 8u};
}
__device__ __forceinline__ TabView view_of_b(const RowTab& t) {                 // layout B: 32 x {gamma, od32}
    const uint32_t c = 8u * (threadIdx.x & 31u);
    return TabView{lds_address(&t), 256u, c + 4u, c};
}
// the 2 KB version for kernels that only run finish steps
struct SmallTab {
    float f[256], g[256];
    __device__ __forceinline__ void fill() {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) { f[i] = d_od_f32[i]; g[i] = (float)d_gamma[i]; }
    }
};
__device__ __forceinline__ TabView view_of(const SmallTab& t) { return TabView{lds_address(&t), 4u, 0u, 1024u}; }

// ---- stratified sample: one pixel per block of 2^stride_log2 pixels (2^cps_log2 chunks, cps_log2 >= 4) ----
// Which pixel is decided per HASH GROUP = the 64 chunks one wave covers with one load (or the whole block when
// it is larger): every lane of a wave row then shares the draw, so the sweep computes it on the scalar unit and
// pays one compare per chunk.  The draw picks a chunk of the block and pixel 0 or 3 of that chunk (the two a
// single shift extracts).  The sample only steers the brackets; results never depend on it.
__device__ __forceinline__ uint32_t sample_hash(uint32_t group) {
    uint32_t h = group * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    return h;
}
__device__ __forceinline__ int sample_group_shift(int cps_log2) { return cps_log2 > 6 ? cps_log2 : 6; }   // chunk index -> group
// pixel index kept for block b (may lie beyond the tile for the last block: then the entry is absent)
__device__ __forceinline__ long long sample_pixel(uint32_t b, int cps_log2) {
    const uint32_t chunk0 = b << cps_log2;
    const uint32_t h = sample_hash(chunk0 >> sample_group_shift(cps_log2));
    const uint32_t chunk = chunk0 + ((h >> 8) & ((1u << cps_log2) - 1u));
    return (long long)chunk * 4 + ((h >> 31) ? 3 : 0);
}
// only the LAST block of a tile can hold its draw beyond the tile: every other entry is present without looking
__device__ __forceinline__ bool sample_absent(int b, int cps_log2, int P) {
    return b >= ((P - 1) >> (cps_log2 + 2)) && sample_pixel((uint32_t)b, cps_log2) >= P;
}

// Monotone surrogate of arctan2(y, x) on (-pi, pi]: y/(|x|+|y|) in [-1,1] for x >= 0, mirrored to
// (1,2] / [-2,-1) for x < 0.  One v_rcp instead of an atan2f per pixel; arctan2 itself is evaluated
// in binary64 only for the selected order statistics.
__device__ __forceinline__ float pseudo_angle(float x, float y) {
    const float d = fabsf(x) + fabsf(y);
    float p = d > 0.0f ? y * __builtin_amdgcn_rcpf(d) : 0.0f;
    if (x < 0.0f) p = (y >= 0.0f ? 2.0f : -2.0f) - p;
    return p;
}
__device__ inline double angle_of_pseudo(double p) {
    if (fabs(p) <= 1.0) return atan2(p, 1.0 - fabs(p));
    const double pp = p > 0.0 ? 2.0 - p : -2.0 - p;
    return atan2(pp, -(1.0 - fabs(pp)));
}
__device__ __forceinline__ float angle_key(const float* V, float x, float y, float z) {
    // That = OD @ V  (macenko_stain_extractor.py:29)
    const float t0 = fmaf(V[4], z, fmaf(V[2], y, V[0] * x));
    const float t1 = fmaf(V[5], z, fmaf(V[3], y, V[1] * x));
    return pseudo_angle(t0, t1);
}
__device__ __forceinline__ float nan_f() { return __uint_as_float(0x7fc00000u); }
__device__ __forceinline__ double nan_d() { return __longlong_as_double(0x7ff8000000000000LL); }

// order-preserving 32-bit image of a binary32 key
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// ------------------------------------------------------------------------------------------
// workgroup-level exact selection (any blockDim that is a multiple of 64)
// ------------------------------------------------------------------------------------------
struct SelScratch {
    uint32_t hist[1024];
    uint32_t misc[64];        // [0,16) the windowed selection primitives; [16,64) the one-pass primitives (wg_refine_s, wg_pick2)
};

// Locate the histogram bin holding 0-based rank k: out = {bin, count below bin, count in bin}.
// All threads call; wave 0 works; ends with a barrier.  Requires k < sum(hist).
__device__ inline void wg_locate(const uint32_t* hist, int nb, uint32_t k, uint32_t* out) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int per = (nb + 63) >> 6;
        uint32_t s = 0;
        for (int j = 0; j < per; ++j) {
            const int idx = lane * per + j;
            if (idx < nb) s += hist[idx];
        }
        uint32_t inc = s;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        const uint32_t exc = inc - s;
        if (k >= exc && k < inc) {
            uint32_t acc = exc;
            for (int j = 0; j < per; ++j) {
                const int idx = lane * per + j;
                if (idx >= nb) break;
                const uint32_t hcnt = hist[idx];
                if (k < acc + hcnt) { out[0] = (uint32_t)idx; out[1] = acc; out[2] = hcnt; break; }
                acc += hcnt;
            }
        }
    }
    __syncthreads();
}

// Visit key_at(i) for i in [0, n) with 4 independent loads in flight per thread (the key functors
// read global/LDS memory; a plain loop would expose one full latency per element).
template <class KeyAt, class Fn>
__device__ __forceinline__ void wg_for_each_key(int n, const KeyAt& key_at, Fn fn) {
    const int bd = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * bd) {
        float f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * bd;
            f[u] = i < n ? key_at(i) : nan_f();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (f[u] == f[u]) fn(f2ord(f[u]));
    }
}

// Exact 0-based k-th smallest of the n keys key_at(i) (NaN = absent).  count_le = #keys <= result,
// n_valid = #non-NaN keys.  Narrowing windows in the ordered-integer domain: each pass histograms
// the live window into <= 1024 bins, so a pass contends on LDS atomics only under real ties.
// (results come back by value: a reference to a caller's local would reach this out-of-line function as a generic pointer to
//  private memory, and this hipcc mis-folds the null check of that cast into an illegal v_cmp with src_private_base)
struct SelResult { float x; uint32_t count_le, n_valid; };
template <class KeyAt>
__device__ __noinline__ SelResult wg_select(int n, KeyAt key_at, uint32_t k, SelScratch& S) {
    uint32_t count_le = 0, n_valid = 0;
    // pass 0: window = [min, max]
    if (threadIdx.x == 0) { S.misc[4] = 0xffffffffu; S.misc[5] = 0; S.misc[6] = 0; }
    __syncthreads();
    {
        uint32_t mn = 0xffffffffu, mx = 0, cnt = 0;
        wg_for_each_key(n, key_at, [&](uint32_t o) { mn = min(mn, o); mx = max(mx, o); ++cnt; });
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
            cnt += __shfl_xor((int)cnt, o, 64);
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&S.misc[4], mn); atomicMax(&S.misc[5], mx); atomicAdd(&S.misc[6], cnt); }
    }
    __syncthreads();
    uint32_t wlo = S.misc[4], whi = S.misc[5];
    n_valid = S.misc[6];
    __syncthreads();
    if (n_valid == 0) return SelResult{nan_f(), 0u, 0u};
    if (k >= n_valid) k = n_valid - 1;
    uint32_t below = 0, in_win = n_valid;
    for (int guard = 0; guard < 8; ++guard) {
        const uint32_t R = whi - wlo;
        if (R == 0) break;
        const int s = R < 1024u ? 0 : (32 - __clz(R) - 10);
        const int nb = (int)(R >> s) + 1;
        for (int i = threadIdx.x; i < nb; i += blockDim.x) S.hist[i] = 0;
        __syncthreads();
        wg_for_each_key(n, key_at, [&](uint32_t o) {
            if (o >= wlo && o <= whi) atomicAdd(&S.hist[(o - wlo) >> s], 1u);
        });
        __syncthreads();
        wg_locate(S.hist, nb, k - below, S.misc);
        const uint32_t b = S.misc[0];
        below += S.misc[1];
        in_win = S.misc[2];
        __syncthreads();
        const uint32_t nlo = wlo + (b << s);
        const uint32_t span = s ? ((1u << s) - 1u) : 0u;
        whi = (whi - nlo) < span ? whi : nlo + span;
        wlo = nlo;
        if (s == 0) break;
    }
    count_le = below + in_win;
    return SelResult{ord2f(wlo), count_le, n_valid};
}

// smallest key strictly greater than v (v itself if none)
template <class KeyAt>
__device__ __noinline__ float wg_next_above(int n, KeyAt key_at, float v, SelScratch& S) {
    if (threadIdx.x == 0) S.misc[7] = 0xffffffffu;
    __syncthreads();
    uint32_t best = 0xffffffffu;
    const uint32_t ov = f2ord(v);
    wg_for_each_key(n, key_at, [&](uint32_t o) { if (o > ov) best = min(best, o); });
    for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMin(&S.misc[7], best);
    __syncthreads();
    const uint32_t r = S.misc[7];
    __syncthreads();
    return r == 0xffffffffu ? v : ord2f(r);
}

// order statistics k and k2 = min(k+1, n_valid-1)
template <class KeyAt>
__device__ void wg_select_pair(int n, KeyAt key_at, uint32_t k, float& xa, float& xb, SelScratch& S) {
    const SelResult r = wg_select(n, key_at, k, S);
    xa = r.x;
    xb = (k + 1 < r.count_le || k + 1 >= r.n_valid) ? xa : wg_next_above(n, key_at, xa, S);
}

// min / max / count of the valid keys (ordered-integer domain)
template <class KeyAt>
__device__ __forceinline__ void wg_minmax(int n, const KeyAt& key_at, uint32_t& omin, uint32_t& omax, uint32_t& nv, SelScratch& S) {
    if (threadIdx.x == 0) { S.misc[4] = 0xffffffffu; S.misc[5] = 0; S.misc[6] = 0; }
    __syncthreads();
    uint32_t mn = 0xffffffffu, mx = 0, cnt = 0;
    wg_for_each_key(n, key_at, [&](uint32_t o) { mn = min(mn, o); mx = max(mx, o); ++cnt; });
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        cnt += __shfl_xor((int)cnt, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&S.misc[4], mn); atomicMax(&S.misc[5], mx); atomicAdd(&S.misc[6], cnt); }
    __syncthreads();
    omin = S.misc[4]; omax = S.misc[5]; nv = S.misc[6];
    __syncthreads();
}

// Two brackets (for percentiles of the FULL population) from the sample keys, in three passes: min/max, one
// shared 1024-bin histogram, one 4x256-bin refinement.  Each end is a bin edge on the safe side of the exact sample
// order statistic at rank -/+ z sigma (so the bracket is a hair wider than with exact sample quantiles, never
// narrower); an end opens to -inf/+inf when its rank leaves the sample.
// The keys sit in REGISTERS: thread t holds sample entries t, t + blockDim, ... of NSETS key sets as ordered
// integers (kAbsent = no key).  Bracket b is the pct[b]-th percentile of key set set_of[b].  The passes touch no
// memory but the LDS histogram: the sample is read and its keys are evaluated once.
constexpr uint32_t kAbsent = 0xffffffffu;
#ifdef SL_DEBUG_SUBCLK
__device__ unsigned long long g_bclk[16];      // development aid: wall-clock ticks per step of wg_brackets_regs, summed over calls
#define SL_BCLK(j) { __syncthreads(); if (threadIdx.x == 0) { const long long now_ = wall_clock64(); atomicAdd(&g_bclk[j], (unsigned long long)(now_ - bclk_t_)); bclk_t_ = now_; } }
#else
#define SL_BCLK(j)
#endif
template <int NSETS, int KPT, int NBR>
__device__ __forceinline__ void wg_brackets_regs(const uint32_t (&ord)[NSETS][KPT], const int (&set_of)[NBR],
                                                 const double (&pct)[NBR], float* lo, float* hi, SelScratch& S, float z = kBracketZ) {
    static_assert(2 * NBR <= 4, "four 256-bin refinement windows");
    uint32_t omin[NSETS], omax[NSETS], nv[NSETS];
#ifdef SL_DEBUG_SUBCLK
    long long bclk_t_ = wall_clock64();
#endif
#pragma unroll
    for (int s = 0; s < NSETS; ++s) {
        if (threadIdx.x == 0) { S.misc[4] = 0xffffffffu; S.misc[5] = 0; S.misc[6] = 0; }
        __syncthreads();
        uint32_t mn = 0xffffffffu, mx = 0, cnt = 0;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t o = ord[s][j];
            if (o != kAbsent) { mn = min(mn, o); mx = max(mx, o); ++cnt; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
            cnt += __shfl_xor((int)cnt, o, 64);
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&S.misc[4], mn); atomicMax(&S.misc[5], mx); atomicAdd(&S.misc[6], cnt); }
        __syncthreads();
        omin[s] = S.misc[4]; omax[s] = S.misc[5]; nv[s] = S.misc[6];
        __syncthreads();
    }
    SL_BCLK(0);
    uint32_t rank[2 * NBR], wlo[2 * NBR], whi[2 * NBR], below[2 * NBR];
    bool open[2 * NBR];
#pragma unroll
    for (int b = 0; b < NBR; ++b) {
        const uint32_t n = nv[set_of[b]];
        rank[2 * b] = rank[2 * b + 1] = 0; open[2 * b] = open[2 * b + 1] = true;
        wlo[2 * b] = wlo[2 * b + 1] = whi[2 * b] = whi[2 * b + 1] = below[2 * b] = below[2 * b + 1] = 0;
        if (n > 0) {
            const double q = pct[b] / 100.0;
            const double r = q * ((double)n - 1.0);
            const double sd = sqrt(fmax(q * (1.0 - q) * (double)n, 0.0));
            const long long rlo = (long long)floor(r - z * sd) - 1;
            const long long rhi = (long long)ceil(r + z * sd) + 1;
            open[2 * b] = rlo < 0;
            open[2 * b + 1] = rhi > (long long)n - 1;
            rank[2 * b] = open[2 * b] ? 0u : (uint32_t)rlo;
            rank[2 * b + 1] = open[2 * b + 1] ? n - 1 : (uint32_t)rhi;
        }
    }
    int s1[NSETS];
#pragma unroll
    for (int s = 0; s < NSETS; ++s) {                       // coarse pass per key set
        const uint32_t R = omax[s] - omin[s];
        s1[s] = (nv[s] == 0 || R < 1024u) ? 0 : (32 - __clz(R) - 10);
        if (nv[s] == 0) continue;                            // block-uniform
        const int nb1 = (int)(R >> s1[s]) + 1;
        for (int i = threadIdx.x; i < nb1; i += blockDim.x) S.hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t o = ord[s][j];
            if (o != kAbsent) atomicAdd(&S.hist[(o - omin[s]) >> s1[s]], 1u);
        }
        __syncthreads();
        SL_BCLK(1);
#pragma unroll
        for (int i = 0; i < 2 * NBR; ++i) {
            if (set_of[i >> 1] != s) continue;
            wg_locate(S.hist, nb1, rank[i], S.misc);
            wlo[i] = omin[s] + (S.misc[0] << s1[s]);
            const uint32_t span = s1[s] ? ((1u << s1[s]) - 1u) : 0u;
            whi[i] = (omax[s] - wlo[i]) < span ? omax[s] : wlo[i] + span;
            below[i] = S.misc[1];
            __syncthreads();
        }
    }
    SL_BCLK(2);
    bool any_refine = false;
#pragma unroll
    for (int s = 0; s < NSETS; ++s) any_refine = any_refine | (nv[s] > 0 && s1[s] > 0);
    if (any_refine) {                                        // every window into 256 bins (segments of the LDS histogram)
        for (int i = threadIdx.x; i < 1024; i += blockDim.x) S.hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            if (!(nv[s] > 0 && s1[s] > 0)) continue;
            const int s2 = s1[s] > 8 ? s1[s] - 8 : 0;
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t o = ord[s][j];
                if (o == kAbsent) continue;
#pragma unroll
                for (int i = 0; i < 2 * NBR; ++i)
                    if (set_of[i >> 1] == s && o >= wlo[i] && o <= whi[i]) atomicAdd(&S.hist[i * 256 + ((o - wlo[i]) >> s2)], 1u);
            }
        }
        __syncthreads();
        SL_BCLK(3);
#pragma unroll
        for (int i = 0; i < 2 * NBR; ++i) {
            const int s = set_of[i >> 1];
            if (!(nv[s] > 0 && s1[s] > 0)) continue;
            const int s2 = s1[s] > 8 ? s1[s] - 8 : 0;
            wg_locate(S.hist + i * 256, 256, rank[i] - below[i], S.misc);
            const uint32_t nlo = wlo[i] + (S.misc[0] << s2);
            const uint32_t span = s2 ? ((1u << s2) - 1u) : 0u;
            whi[i] = (whi[i] - nlo) < span ? whi[i] : nlo + span;
            wlo[i] = nlo;
            __syncthreads();
        }
    }
    SL_BCLK(4);
#pragma unroll
    for (int b = 0; b < NBR; ++b) {
        const bool none = nv[set_of[b]] == 0;
        lo[b] = (none || open[2 * b]) ? -INFINITY : ord2f(wlo[2 * b]);
        hi[b] = (none || open[2 * b + 1]) ? INFINITY : ord2f(whi[2 * b + 1]);
    }
}

// inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / broadcasts (gfx9)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// One wave: the bin of `hist[0..nb)` (nb a multiple of 64, <= 1024) holding 0-based rank k -> out = {bin, count below, count in bin};
// k beyond the total gives {nb - 1, total - count(last bin), count(last bin)}.
__device__ __forceinline__ void wave_locate(const uint32_t* hist, int nb, uint32_t k, uint32_t* out, int lane) {
    const int per = nb >> 6;
    uint32_t s = 0;
    for (int j = 0; j < per; ++j) s += hist[lane * per + j];
    const uint32_t inc = wave_inclusive_scan(s), exc = inc - s;
    const uint32_t total = (uint32_t)__shfl((int)inc, 63, 64);
    const uint32_t kk = total ? (k < total ? k : total - 1) : 0u;
    if (total == 0) { if (lane == 0) { out[0] = 0; out[1] = 0; out[2] = 0; } return; }
    if (kk >= exc && kk < inc) {
        uint32_t acc = exc;
        for (int j = 0; j < per; ++j) {
            const uint32_t h = hist[lane * per + j];
            if (kk < acc + h) { out[0] = (uint32_t)(lane * per + j); out[1] = acc; out[2] = h; break; }
            acc += h;
        }
    }
}

// Exact order statistics k and k2 = min(k+1, nv-1) of a SMALL key set (the bracket members): min/max,
// one 1024-bin histogram, then the keys of the bin holding rank k are gathered into LDS and ranked by
// brute force.  Falls back to the generic windowed selection when that bin holds more than 256 keys.
template <class KeyAt>
__device__ __forceinline__ void wg_select_pair_small(int n, const KeyAt& key_at, uint32_t k, float& xa, float& xb, SelScratch& S) {
    uint32_t omin, omax, nv;
    wg_minmax(n, key_at, omin, omax, nv, S);
    if (nv == 0) { xa = xb = nan_f(); return; }
    if (k >= nv) k = nv - 1;
    const uint32_t k2 = k + 1 < nv ? k + 1 : k;
    const uint32_t R = omax - omin;
    const int s1 = R < 1024u ? 0 : (32 - __clz(R) - 10);
    const int nb1 = (int)(R >> s1) + 1;
    for (int i = threadIdx.x; i < nb1; i += blockDim.x) S.hist[i] = 0;
    __syncthreads();
    wg_for_each_key(n, key_at, [&](uint32_t o) { atomicAdd(&S.hist[(o - omin) >> s1], 1u); });
    __syncthreads();
    wg_locate(S.hist, nb1, k, S.misc);
    const uint32_t bin = S.misc[0], below = S.misc[1], cnt = S.misc[2];
    __syncthreads();
    if (cnt > 256u) {                                     // heavy ties / degenerate spread: generic path
        wg_select_pair(n, key_at, k, xa, xb, S);
        return;
    }
    const uint32_t wlo = omin + (bin << s1);
    const uint32_t span = s1 ? ((1u << s1) - 1u) : 0u;
    const uint32_t whi = (omax - wlo) < span ? omax : wlo + span;
    if (threadIdx.x == 0) S.misc[9] = 0;
    __syncthreads();
    float* list = reinterpret_cast<float*>(S.hist);       // the histogram is no longer needed
    wg_for_each_key(n, key_at, [&](uint32_t o) {
        if (o >= wlo && o <= whi) list[atomicAdd(&S.misc[9], 1u)] = ord2f(o);
    });
    __syncthreads();
    if (threadIdx.x == 0) { S.misc[10] = 0; S.misc[11] = 0; }
    __syncthreads();
    if (threadIdx.x < cnt) {
        const float me = list[threadIdx.x];
        uint32_t r = 0;
        for (uint32_t j = 0; j < cnt; ++j) {
            const float o = list[j];
            r += (o < me || (o == me && j < threadIdx.x)) ? 1u : 0u;
        }
        if (r == k - below) S.misc[10] = __float_as_uint(me);
        if (r == k2 - below) S.misc[11] = __float_as_uint(me);
    }
    __syncthreads();
    xa = __uint_as_float(S.misc[10]);
    const bool same_bin = (k2 - below) < cnt;
    xb = __uint_as_float(S.misc[11]);
    __syncthreads();
    if (!same_bin) xb = wg_next_above(n, key_at, xa, S);
}

// ------------------------------------------------------------------------------------------
// finish-step arithmetic (thread 0)
// ------------------------------------------------------------------------------------------
// One Jacobi rotation annihilating a_pq of a symmetric 3x3 (r = the third index); every operand
// is a named scalar so that nothing is indexed dynamically (dynamic indexing would put the
// matrices in scratch memory and cost ~100 us of latency per tile on the single working lane).
__device__ __forceinline__ void jacobi_rot(double& app, double& aqq, double& apq, double& apr, double& aqr,
                                           double (&vp)[3], double (&vq)[3]) {
    if (apq == 0.0) return;
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
    app -= t * apq;
    aqq += t * apq;
    apq = 0.0;
    const double npr = c * apr - sn * aqr, nqr = sn * apr + c * aqr;
    apr = npr;
    aqr = nqr;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a = vp[i], b = vq[i];
        vp[i] = c * a - sn * b;
        vq[i] = sn * a + c * b;
    }
}

// sums = {n, Sx, Sy, Sz, Sxx, Sxy, Sxz, Syy, Syz, Szz} -> status, V (binary64 + binary32)
__device__ __forceinline__ int eigvecs_from_moments(const double* sum, double* Vd, float* Vf) {
    const double n = sum[0];
    int status = SL_TILE_OK;
    double v0[3] = {1, 0, 0}, v1[3] = {0, 1, 0}, v2[3] = {0, 0, 1};
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
    if (n < 1) status = SL_TILE_EMPTY_MASK;
    else if (n < 2) status = SL_TILE_DEGENERATE_COV;
    else {
        // np.cov(OD, rowvar=False): (sum xx^T - n mean mean^T) / (n - 1)   (macenko_stain_extractor.py:22)
        const double mx = sum[1] / n, my = sum[2] / n, mz = sum[3] / n, inv = 1.0 / (n - 1.0);
        a00 = (sum[4] - n * mx * mx) * inv; a01 = (sum[5] - n * mx * my) * inv; a02 = (sum[6] - n * mx * mz) * inv;
        a11 = (sum[7] - n * my * my) * inv; a12 = (sum[8] - n * my * mz) * inv; a22 = (sum[9] - n * mz * mz) * inv;
        for (int sweep = 0; sweep < 30; ++sweep) {
            const double off = fabs(a01) + fabs(a02) + fabs(a12);
            const double dia = fabs(a00) + fabs(a11) + fabs(a22);
            if (off <= 1e-300 || off <= 1e-22 * dia) break;
            jacobi_rot(a00, a11, a01, a02, a12, v0, v1);
            jacobi_rot(a00, a22, a02, a01, a12, v0, v2);
            jacobi_rot(a11, a22, a12, a01, a02, v1, v2);
        }
    }
    // eigh is ascending; the reference takes columns [2, 1] = largest, second largest (:24)
    double w0 = a00, w1 = a11, w2 = a22;
#define SL_SWAP_COL(wa, wb, va, vb) do { const double tw = wa; wa = wb; wb = tw; \
        for (int i_ = 0; i_ < 3; ++i_) { const double tv = va[i_]; va[i_] = vb[i_]; vb[i_] = tv; } } while (0)
    if (w0 > w1) SL_SWAP_COL(w0, w1, v0, v1);
    if (w1 > w2) SL_SWAP_COL(w1, w2, v1, v2);
    if (w0 > w1) SL_SWAP_COL(w0, w1, v0, v1);
#undef SL_SWAP_COL
    // Rank-deficient covariance (tissue of one or two distinct colours): the eigenvectors of the null space are whatever
    // round-off makes them -- in numpy as much as here -- and the two kernel schedules, which sum the moments in different
    // orders, would disagree completely.  Pick them canonically instead (the outputs stay finite like the reference's,
    // and are reproducible): no spread at all -> the first two axes; a line -> the unit vector orthogonal to it that is
    // closest to the coordinate axis the line is least aligned with.
    if (status == SL_TILE_OK) {
        const double scale = (sum[4] + sum[7] + sum[9]) / n;            // mean squared optical density: the round-off floor of cov is ~1e-15 of it
        if (!(w2 > 1e-12 * scale)) {
            v2[0] = 1; v2[1] = 0; v2[2] = 0; v1[0] = 0; v1[1] = 1; v1[2] = 0;
        } else if (!(w1 > 1e-12 * scale)) {
            int ax = 0;
            if (fabs(v2[1]) < fabs(v2[ax])) ax = 1;
            if (fabs(v2[2]) < fabs(v2[ax])) ax = 2;
            double u[3] = {-v2[ax] * v2[0], -v2[ax] * v2[1], -v2[ax] * v2[2]};
            u[ax] += 1.0;
            const double nu = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            for (int i = 0; i < 3; ++i) v1[i] = u[i] / nu;
        }
    }
    const double s2 = v2[0] < 0 ? -1.0 : 1.0, s1 = v1[0] < 0 ? -1.0 : 1.0;      // :26-27
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Vd[c * 2 + 0] = s2 * v2[c]; Vf[c * 2 + 0] = (float)(s2 * v2[c]);
        Vd[c * 2 + 1] = s1 * v1[c]; Vf[c * 2 + 1] = (float)(s1 * v1[c]);
    }
    return status;
}

// Two (numerically) parallel stain vectors -- tissue of a single colour, or a collapsed dictionary: the Gram matrix is
// singular, the concentrations are inf/NaN in the reference and depend on the last bit here.  Such a tile is reported as
// degenerate (status 2, passed through unchanged) instead of producing round-off-dependent output.
__device__ __forceinline__ bool stain_matrix_singular(const double* M) {
    const double g11 = M[0] * M[0] + M[1] * M[1] + M[2] * M[2], g22 = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    const double g12 = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    return !(g11 * g22 - g12 * g12 > 1e-8 * g11 * g22);
}

// pseudo-angle order statistics -> stain matrix (macenko_stain_extractor.py:33-44).  Called by a whole wave (the result is
// valid in every lane): the four arctan2 run in lanes 0-3 at once and the two sincos in lanes 0-1 -- this one-lane chain
// of binary64 library calls was 31 us of every tile's finish step; the same calls on the same arguments, bit for bit.
__device__ __forceinline__ void stain_matrix_from_angles(const double* Vd, const float* xs /*[4]*/, const double* gfrac, double* M, int lane) {
    const double ang = angle_of_pseudo((double)xs[lane & 3]);
    const int pair = (lane & 1) * 2;                      // even lanes: minPhi (xs[0], xs[1]); odd lanes: maxPhi (xs[2], xs[3])
    const double phi = np_lerp(__shfl(ang, pair, 64), __shfl(ang, pair + 1, 64), (lane & 1) ? gfrac[1] : gfrac[0]);
    double s, c;
    sincos(phi, &s, &c);
    const double s1 = __shfl(s, 0, 64), c1 = __shfl(c, 0, 64), s2 = __shfl(s, 1, 64), c2 = __shfl(c, 1, 64);
    double v1[3], v2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {                         // :36-37
        v1[c] = Vd[c * 2] * c1 + Vd[c * 2 + 1] * s1;
        v2[c] = Vd[c * 2] * c2 + Vd[c * 2 + 1] * s2;
    }
    const bool first = v1[0] > v2[0];                     // :40-43
    double h[3], e[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { h[c] = first ? v1[c] : v2[c]; e[c] = first ? v2[c] : v1[c]; }
    const double nh = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
    const double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) { M[c] = h[c] / nh; M[3 + c] = e[c] / ne; }   // :44
}

// ------------------------------------------------------------------------------------------
// ONE selection sweep for the angular AND the concentration percentiles (round 3)
// ------------------------------------------------------------------------------------------
// normalizer.py:45-47 computes the concentrations with the tile's own stain matrix M, and M is exact only once the angular
// order statistics are (macenko_stain_extractor.py:33-37): that dependency cost a whole sweep (collect angle candidates,
// finish, collect concentration candidates).  Both selection sweeps only PROVE pixels plain and append the rest as raw RGB
// whose exact keys the finish step evaluates, so the concentration test can run before M is known, against every M the
// sample leaves possible:
//   * the sample's angular brackets [lo, hi] bound the two percentile angles; a box of kBoxFrac of their width around
//     the mid-points (about +-3.6 sigma of the sample rank) is where the exact angles will fall in all but ~1e-3 of the tiles
//     (when a 6-sigma bracket is open -- a small tissue sample -- the box is a second pair of brackets at kBoxZ sigma);
//   * for M in that box the interior solution of a pixel is a(M; x) = T a(M~; x) + r with M~ the box centre (the rows of
//     G^-1 M always span the plane of V, so T is 2x2).  |T - I| <= eps and |r| <= rho over the box (nine grid points,
//     inflated) give  a_i(M; x) <= a~_i + eps_i (|a~_1| + |a~_2|) + rho_i  for every pixel;
//   * c_i <= max(0, a_i) when g12 >= 0, so  a~_i + eps_i (|a~_1| + |a~_2|) < L_i - rho_i  for both stains proves both
//     concentrations below their brackets [L_i, H_i] (the sample's brackets under M~, widened by the same bound);
//   * a~ = u t + k~ costs four FMAs on the two projections t = V^T od the angle test needs anyway.
// After the sweep the finish step computes the exact M, then CHECKS the assumption: T(M), r(M) against the eps, rho the sweep
// used (merged_verify).  If it holds, every uncollected pixel is proven below both brackets under the exact M and the exact
// keys of the collected ones complete the counts; if it does not (or a bracket missed, or a list overflowed) the tile takes
// sweep 3 of the four-sweep schedule with brackets from the exact M.  Results never depend on the box, the sample or the
// pre-filter: the order statistics are exact on the same binary32 keys either way.
constexpr double kBoxFrac = 0.6;        // box half-width as a fraction of the 6-sigma bracket half-width
constexpr double kBoxInflate = 1.25;    // safety factor on the nine-point maxima (curvature inside the box)
constexpr double kBoxMaxEps = 0.25;     // a box over which the map changes by more than this is not worth a merged sweep

struct LassoD { double W[2][3], k[2], g12; };          // a(M; x) = W x + k, binary64 (lasso_consts' interior solution)
__device__ __forceinline__ void lasso_affine_d(const double* M, double lam, LassoD& o) {
    const double g11 = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    const double g22 = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    const double g12 = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    const double det = g11 * g22 - g12 * g12;
    const double i11 = g22 / det, i12 = -g12 / det, i22 = g11 / det;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o.W[0][c] = i11 * M[c] + i12 * M[3 + c];
        o.W[1][c] = i12 * M[c] + i22 * M[3 + c];
    }
    o.k[0] = -lam * (i11 + i12);
    o.k[1] = -lam * (i12 + i22);
    o.g12 = g12;
}
// T (2x2), r with  A.W x + A.k = T (C.W x + C.k) + r  for every x (least squares over the rows; exact when the rows of both
// maps span the same plane)
__device__ __forceinline__ void relate_affine(const LassoD& A, const LassoD& C, double (&T)[2][2], double (&r)[2]) {
    double g[2][2], b[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            g[i][j] = C.W[i][0] * C.W[j][0] + C.W[i][1] * C.W[j][1] + C.W[i][2] * C.W[j][2];
            b[i][j] = A.W[i][0] * C.W[j][0] + A.W[i][1] * C.W[j][1] + A.W[i][2] * C.W[j][2];
        }
    const double rd = 1.0 / (g[0][0] * g[1][1] - g[0][1] * g[1][0]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        T[i][0] = (b[i][0] * g[1][1] - b[i][1] * g[1][0]) * rd;
        T[i][1] = (b[i][1] * g[0][0] - b[i][0] * g[0][1]) * rd;
        r[i] = A.k[i] - T[i][0] * C.k[0] - T[i][1] * C.k[1];
    }
}
// the stain matrix of two percentile angles (macenko_stain_extractor.py:36-44), one lane
__device__ __forceinline__ void stain_matrix_from_phi(const double* Vd, double phi_min, double phi_max, double* M) {
    double s1, c1, s2, c2;
    sincos(phi_min, &s1, &c1);
    sincos(phi_max, &s2, &c2);
    double v1[3], v2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v1[c] = Vd[c * 2] * c1 + Vd[c * 2 + 1] * s1;
        v2[c] = Vd[c * 2] * c2 + Vd[c * 2 + 1] * s2;
    }
    const bool first = v1[0] > v2[0];
    double h[3], e[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { h[c] = first ? v1[c] : v2[c]; e[c] = first ? v2[c] : v1[c]; }
    const double nh = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
    const double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) { M[c] = h[c] / nh; M[3 + c] = e[c] / ne; }
}

struct MergedConc {
    int ok;                    // the merged sweep collects concentration candidates for this tile
    int pad_;
    float u[2][2], kt[2];      // a~_i = u[i][0] t0 + u[i][1] t1 + kt[i],  t = Vf^T od
    float eps[2], thr[2];      // plain_i <=> a~_i + eps[i] (|a~_1| + |a~_2|) < thr[i]
    float L[2], H[2];          // brackets of the exact concentration keys
    double rho[2], eta[2];     // rho: bound on |r| + eta over the box; eta: rounding allowance of the binary32 evaluations
    LassoD C;                  // the box centre's map, binary64
    LassoK Lc;                 // the box centre's lasso constants (sample keys)
};

// per-tile state of the merged selection stage in the one-launch-per-phase schedule (the fused kernel keeps it in LDS)
struct TileMerged {
    MergedConc mk;
    float xmin;
    int conc_done;
};

// Called by one whole wave after the angular brackets are known: lanes 0..8 evaluate the 3 x 3 grid of the box.
// box = {lo0, hi0, lo1, hi1}: the intervals of pseudo-angle the two percentile angles are assumed to fall in (angle_brackets)
__device__ __forceinline__ void merged_box(const double* Vd, const float* box, double lam, int lane, MergedConc& mk) {
    const bool finite = (box[0] > -INFINITY) & (box[1] < INFINITY) & (box[2] > -INFINITY) & (box[3] < INFINITY);
    const int i0 = lane % 3, i1 = (lane / 3) % 3;
    const double m0 = 0.5 * ((double)box[0] + (double)box[1]), r0 = 0.5 * ((double)box[1] - (double)box[0]);
    const double m1 = 0.5 * ((double)box[2] + (double)box[3]), r1 = 0.5 * ((double)box[3] - (double)box[2]);
    const double p0 = finite ? m0 + (double)(i0 - 1) * r0 : -0.25;
    const double p1 = finite ? m1 + (double)(i1 - 1) * r1 : 0.25;
    double M[6];
    stain_matrix_from_phi(Vd, angle_of_pseudo(p0), angle_of_pseudo(p1), M);
    LassoD A, C;
    lasso_affine_d(M, lam, A);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) C.W[i][c] = __shfl(A.W[i][c], 4, 64);
        C.k[i] = __shfl(A.k[i], 4, 64);
    }
    C.g12 = __shfl(A.g12, 4, 64);
    double Mc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) Mc[i] = __shfl(M[i], 4, 64);
    double T[2][2], r[2];
    relate_affine(A, C, T, r);
    double e0 = fmax(fabs(T[0][0] - 1.0), fabs(T[0][1])), e1 = fmax(fabs(T[1][1] - 1.0), fabs(T[1][0]));
    double q0 = fabs(r[0]), q1 = fabs(r[1]);
    const bool lane_bad = (lane < 9) & !((e0 <= kBoxMaxEps) & (e1 <= kBoxMaxEps) & (q0 <= 1.0) & (q1 <= 1.0) & (A.g12 >= 0.0));
    const bool any_bad = __ballot(lane_bad) != 0ull;
    if (lane >= 9 || lane_bad) e0 = e1 = q0 = q1 = 0.0;
    for (int o = 8; o > 0; o >>= 1) {
        e0 = fmax(e0, __shfl_xor(e0, o, 64)); e1 = fmax(e1, __shfl_xor(e1, o, 64));
        q0 = fmax(q0, __shfl_xor(q0, o, 64)); q1 = fmax(q1, __shfl_xor(q1, o, 64));
    }
    if (lane == 0) {
        mk.ok = (finite && !any_bad) ? 1 : 0;
        mk.pad_ = 0;
        mk.C = C;
        LassoK Lc;
        lasso_consts(Mc, lam, Lc);
        mk.Lc = Lc;
        const double e[2] = {e0, e1}, q[2] = {q0, q1};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int k = 0; k < 2; ++k) mk.u[i][k] = (float)(C.W[i][0] * Vd[k] + C.W[i][1] * Vd[2 + k] + C.W[i][2] * Vd[4 + k]);
            mk.kt[i] = (float)C.k[i];
            mk.eta[i] = 4e-6 * (kOdMax * (fabs(C.W[i][0]) + fabs(C.W[i][1]) + fabs(C.W[i][2])) + fabs(C.k[i]) + 1.0);
            mk.eps[i] = (float)(kBoxInflate * e[i] + 1e-7);
            mk.rho[i] = kBoxInflate * q[i] + mk.eta[i];
        }
    }
}
// thread 0, after the sample's concentration brackets [lo, hi] under the box centre are known
__device__ __forceinline__ void merged_thresholds(MergedConc& mk, float lo0, float lo1, float hi0, float hi1) {
    const float lo[2] = {lo0, lo1}, hi[2] = {hi0, hi1};
    bool ok = mk.ok != 0;
    float ref[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) ref[i] = hi[i] < INFINITY ? hi[i] : 2.0f * lo[i] + 1.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float delta = mk.eps[i] * (ref[i] + 1.5f * ref[1 - i]) + (float)mk.rho[i];
        mk.L[i] = lo[i] - delta;
        mk.H[i] = hi[i] + delta;
        ok = ok & (mk.L[i] > 0.0f) & (lo[i] > -INFINITY);
        mk.thr[i] = mk.L[i] - (float)mk.rho[i] - 1e-6f * fabsf(mk.L[i]);
    }
    if (!ok) {
        mk.ok = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) { mk.u[i][0] = mk.u[i][1] = mk.kt[i] = mk.eps[i] = 0.0f; mk.thr[i] = INFINITY; mk.L[i] = mk.H[i] = INFINITY; }
    }
}
// thread 0, with the exact stain matrix: do the bounds the sweep relied on hold?
__device__ __forceinline__ bool merged_verify(const MergedConc& mk, const double* M, double lam) {
    if (!mk.ok) return false;
    LassoD A;
    lasso_affine_d(M, lam, A);
    double T[2][2], r[2];
    relate_affine(A, mk.C, T, r);
    bool ok = A.g12 >= 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double e = fmax(fabs(T[i][i] - 1.0), fabs(T[i][1 - i]));
        ok = ok & (e <= (double)mk.eps[i]) & (fabs(r[i]) + mk.eta[i] <= mk.rho[i]);
    }
    return ok;
}

// ------------------------------------------------------------------------------------------
// per-pixel bodies shared by both schedules
// ------------------------------------------------------------------------------------------
struct Moments {
    double sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    __device__ __forceinline__ void add(double x, double y, double z) {
        sx += x; sy += y; sz += z;
        sxx = fma(x, x, sxx); sxy = fma(x, y, sxy); sxz = fma(x, z, sxz);
        syy = fma(y, y, syy); syz = fma(y, z, syz); szz = fma(z, z, szz);
    }
    // v[0] = pixel count: n_wave is the wave-uniform count, credited to lane 0 so that a wave sum yields it
    __device__ __forceinline__ void to_array(double* v, uint32_t n_wave, int lane) const {
        v[0] = lane == 0 ? (double)n_wave : 0.0; v[1] = sx; v[2] = sy; v[3] = sz; v[4] = sxx; v[5] = sxy; v[6] = sxz;
        v[7] = syy; v[8] = syz; v[9] = szz;
    }
};

// The sums of ONE trip of one lane (kTrip chunks = 16 pixels) in binary32, then added to the binary64 totals: 9 fast
// FMAs per pixel instead of 9 binary64 ones (4 issue cycles each, both pipes blocked), and the optical densities come from
// the 8-byte {gamma, od32} rows (layout B: half the LDS time of the 16-byte rows, no table switch after the sweep).
// A trip's pixel set is the same in both schedules (part_range keeps parts trip-aligned), so the binary32 partial sums are
// bit-identical across schedules and batch sizes; only the order of the binary64 additions differs, as before.
// Measured against the binary64 reference: stain matrix error 1.6e-8 -> 4e-8 (test tolerance 2e-6).
struct BurstMoments {
    float sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    __device__ __forceinline__ void add(float x, float y, float z) {
        sx += x; sy += y; sz += z;
        sxx = fmaf(x, x, sxx); sxy = fmaf(x, y, sxy); sxz = fmaf(x, z, sxz);
        syy = fmaf(y, y, syy); syz = fmaf(y, z, syz); szz = fmaf(z, z, szz);
    }
    __device__ __forceinline__ void flush(Moments& m) {
        m.sx += (double)sx; m.sy += (double)sy; m.sz += (double)sz; m.sxx += (double)sxx; m.sxy += (double)sxy; m.sxz += (double)sxz;
        m.syy += (double)syy; m.syz += (double)syz; m.szz += (double)szz;
        sx = sy = sz = sxx = sxy = sxz = syy = syz = szz = 0.0f;
    }
};


// Sample bookkeeping of one chunk row (64 chunks starting at the wave-uniform, 64-aligned chunk `row0`): the lane
// whose chunk the draw selects stores pixel 0 or 3 of it.  Scalar hash, ~6 vector instructions per chunk.
template <bool ALIGNED>
__device__ __forceinline__ void sample_row(const Chunk& ch, int row0, int cc, int c1, int P, int cps_log2, uint32_t* samp) {
    const uint32_t h = sample_hash((uint32_t)row0 >> sample_group_shift(cps_log2));
    const uint32_t cmask = (1u << cps_log2) - 1u;
    const uint32_t sel = (h >> 8) & cmask;
    const bool last = (h >> 31) != 0;                                  // uniform: pixel 3 instead of pixel 0
    if (samp && ((cc < c1) & (((uint32_t)cc & cmask) == sel))) {
        const uint32_t v = (last ? ch.w2 : ch.w0) >> (last ? 8 : 0);   // stray top byte for pixel 0: readers ignore it
        if (ALIGNED || (size_t)cc * 4 + (last ? 3 : 0) < (size_t)P) samp[(uint32_t)cc >> cps_log2] = v;
    }
}

// Sweep 1 on the layout-B table ({gamma, od32} per byte): the structure of select_sweep (gathers of a chunk issued one chunk
// ahead of its arithmetic, next trip's chunks in flight), tissue test, binary32 burst sums flushed once per trip.
// c0 must be a multiple of 64; for schedule-independent bursts also of kTrip * nthreads (part_range guarantees it).
template <bool ALIGNED, int kTrip, bool STREAM = false>
__device__ __forceinline__ void moments_sweep_b(const uint8_t* src, int P, int c0, int c1, int t, int nthreads,
                                                const TabReaderB& T, float ylimf, int stride_log2, uint32_t* samp,
                                                Moments& mo, uint32_t& n_tissue) {
    const size_t nbytes = (size_t)P * 3;
    const int lane = t & 63;
    const int cps_log2 = stride_log2 - 2;          // chunks per sampling block
    const int w0 = __builtin_amdgcn_readfirstlane(c0 + (t & ~63));
    struct G { float2 v[12]; };
    auto fetch = [&](int cc) { return load_chunk_clamped<ALIGNED, STREAM>(src, nbytes, cc, c1); };
    auto gather = [&](const Chunk& ch) {
        G g;
#pragma unroll
        for (int i = 0; i < 12; ++i) g.v[i] = T.gam_odf(T.addr(ch, i));
        return g;
    };
    BurstMoments bm;
    auto compute = [&](auto tail_tag, const G& g, int cc) {
        constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const float2 er = g.v[3 * px], eg = g.v[3 * px + 1], eb = g.v[3 * px + 2];
            const bool tc = is_tissue_f(er.x, eg.x, eb.x, ylimf);
            if (!TAIL) {
                n_tissue += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(tc));
                if (tc) bm.add(er.y, eg.y, eb.y);
            } else {
                const bool inb = (cc < c1) & (ALIGNED | ((size_t)cc * 4 + px < (size_t)P));
                const unsigned long long m = __builtin_amdgcn_ballot_w64(tc) & __builtin_amdgcn_ballot_w64(inb);
                n_tissue += (uint32_t)__popcll(m);
                if (tc & inb) bm.add(er.y, eg.y, eb.y);
            }
        }
    };
    Chunk cur[kTrip], nx[kTrip];
#pragma unroll
    for (int k = 0; k < kTrip; ++k) { cur[k] = fetch(w0 + lane + k * nthreads); nx[k] = fetch(w0 + lane + (kTrip + k) * nthreads); }
    G g[2];
    g[0] = gather(cur[0]);
    auto trip = [&](auto tail_tag, int cb) {
#pragma unroll
        for (int k = 0; k < kTrip; ++k) sample_row<ALIGNED>(cur[k], cb + k * nthreads, cb + k * nthreads + lane, c1, P, cps_log2, samp);
#pragma unroll
        for (int k = 0; k < kTrip; ++k) {
            if (k + 1 < kTrip) {
                g[(k + 1) & 1] = gather(cur[k + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < kTrip; ++j) { cur[j] = nx[j]; nx[j] = fetch(cb + lane + (2 * kTrip + j) * nthreads); }
                g[0] = gather(cur[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            compute(tail_tag, g[k & 1], cb + k * nthreads + lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        bm.flush(mo);
    };
    const int lim = ALIGNED ? c1 : min(c1, P >> 2);          // chunks made of in-range pixels only
    int cb = w0;
    for (; cb + (kTrip - 1) * nthreads + 64 <= lim; cb += nthreads * kTrip) trip(std::false_type{}, cb);
    if (cb < c1) trip(std::true_type{}, cb);                    // at most one ragged trip per wave
}

enum { kStageConc = 1, kStageMerged = 2 };       // (the angle-only stage of rounds 1-2 went with the merged sweep)

struct SelConsts {          // everything VGPR-resident (in_vgpr)
    float V[6];
    LassoK L;
    float lo0, hi0, lo1, hi1;
    // merged stage (see MergedConc): at_i = u[i][0] t0 + u[i][1] t1 + kt[i] with t = V^T od; plain <=> at_i + eps[i] (|at_1| + |at_2|) < thr[i]
    float u[2][2], kt[2], eps[2], thr[2];
    // merged stage, XBOUND variant: every tissue pixel has t0 = V1 . od > xmin (see tissue_x_bound), so the sweep needs no gamma values
    float xmin;
};

// A lower bound on the first projection of every TISSUE pixel, valid when the first eigenvector has only positive components:
// tissue <=> 871 gR + 2929 gG + 296 gB < ylimf  =>  the smallest gamma is below ylimf / 4096  =>  one byte is <= b*, the largest
// byte whose gamma is  =>  one optical density is >= od(b*), and with all three weights positive and all densities > 0
// V1 . od >= min(V1) od(b*).  Returns -inf when no bound holds (the caller then keeps the per-pixel tissue test).
__device__ __forceinline__ float tissue_x_bound(const float* Vf /*[6]*/, float ylimf, const TabView& tab) {
    const float vmin = fminf(fminf(Vf[0], Vf[2]), Vf[4]);
    if (!(vmin > 0.0f)) return -INFINITY;
    const float gf = ylimf * (1.0f / 4096.0f);
    if (!(tab.gam(0) < gf)) return INFINITY;                 // no byte can make a pixel tissue
    int lo = 0, hi = 255;                                    // invariant: gam(lo) < gf; the tables are monotone
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab.gam((uint32_t)mid) < gf) lo = mid; else hi = mid - 1;
    }
    return vmin * tab.odf((uint32_t)lo) * (1.0f - 1e-6f);    // (the binary32 evaluation of V1 . od adds positive terms: relative error 2e-7)
}

// The selection sweeps.  A sweep does NOT evaluate the selection keys of every pixel.  A cheap conservative
// test proves, for ~94 % of the pixels, on which side of both brackets their keys fall ("plain").  The remaining
// pixels -- inside or near a bracket, or beyond the outer ends -- are appended as raw RGB to a per-tile list and
// resolved exactly by the finish step.
//   merged stage: the angle test (one key p for both brackets, tissue only: plain <=> hi0 < p < lo1, tested without the
//     division as  y > (hi0+eps) d  and  y < (lo1-eps) d  with d = x + |y|, x > 0) and, from the same two projections,
//     a conservative test on the concentrations under the box of stain matrices (MergedConc)
//   concentration stage (g12 >= 0): c_i <= max(0, a_i) exactly, so  a1 < lo0 and a2 < lo1  =>  both
//     keys lie below their brackets (needs lo > 0; otherwise nothing is plain)
// The plain pixels are not even counted: their number is (valid pixels of the stage) - (raw candidates).
// c0 must be a multiple of 64.  The LDS gathers of a chunk are issued one chunk ahead of its arithmetic.
template <int STAGE> struct SelGather;
template <> struct SelGather<kStageConc> { float v[12]; };        // od32 per byte
template <> struct SelGather<kStageMerged> { float2 v[12]; };
struct SelGatherOd { float v[12]; };                             // merged stage with the projection bound: od32 only

// XBOUND (merged stage only): angle candidates are the pixels with t0 > K.xmin outside the plain cone instead of the tissue
// pixels outside it -- a superset (the finish evaluates the tissue test of every candidate exactly) that costs three
// instructions less per pixel and reads 4-byte table entries.
template <int STAGE, bool ALIGNED, int kTrip, bool STREAM = false, bool XBOUND = false, class TR, class Sink>
__device__ __forceinline__ void select_sweep(const uint8_t* src, int P, int c0, int c1, int t, int nthreads,
                                             const TR& T, float ylimf, const SelConsts& K, Sink& sink) {
    const size_t nbytes = (size_t)P * 3;
    const int lane = t & 63;
    // thresholds of the cheap tests
    const float nhi0m = in_vgpr(-(K.hi0 + kAngleMargin)), nlo1m = in_vgpr(-(K.lo1 - kAngleMargin));
    const bool conc_ok = (K.L.g12 >= 0.0f) & (K.lo0 > 0.0f) & (K.lo1 > 0.0f);
    const float clo0 = conc_ok ? K.lo0 : -INFINITY, clo1 = conc_ok ? K.lo1 : -INFINITY;
    const int w0 = __builtin_amdgcn_readfirstlane(c0 + (t & ~63));
    auto fetch = [&](int cc) { return load_chunk_clamped<ALIGNED, STREAM>(src, nbytes, cc, c1); };   // dead lanes: see `live`
    static_assert(!XBOUND || STAGE == kStageMerged, "");
    using GatherT = std::conditional_t<XBOUND, SelGatherOd, SelGather<STAGE>>;
    const float xmin = in_vgpr(K.xmin);
    auto gather = [&](const Chunk& ch) {
        GatherT g;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if constexpr (STAGE != kStageConc && !XBOUND) g.v[i] = T.gam_odf(T.addr(ch, i));
            else g.v[i] = T.odf(T.addr(ch, i));
        }
        return g;
    };
    auto compute = [&](auto tail_tag, const Chunk& ch, const GatherT& g, int cc) {
        constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            // flagged = valid and not provably plain.  The lane mask is assembled from ballots of BARE compares.
            unsigned long long m;
            if constexpr (STAGE == kStageMerged && XBOUND) {
                const float er = g.v[3 * px], eg = g.v[3 * px + 1], eb = g.v[3 * px + 2];
                const float x = fmaf(K.V[4], eb, fmaf(K.V[2], eg, K.V[0] * er));
                const float y = fmaf(K.V[5], eb, fmaf(K.V[3], eg, K.V[1] * er));
                const float d = x + fabsf(y);
                const float t0 = fmaf(nhi0m, d, y), t1 = fmaf(nlo1m, d, y);
                const bool cone = fminf(t0, -t1) > 0.0f;                     // y > hi0m d, y < lo1m d  (x > xmin > 0 comes with `big`)
                const bool big = x > xmin;
                const float a1 = fmaf(K.u[0][1], y, fmaf(K.u[0][0], x, K.kt[0]));
                const float a2 = fmaf(K.u[1][1], y, fmaf(K.u[1][0], x, K.kt[1]));
                const float sa = fabsf(a1) + fabsf(a2);
                const bool g1 = fmaf(K.eps[0], sa, a1) >= K.thr[0], g2 = fmaf(K.eps[1], sa, a2) >= K.thr[1];
                m = (__builtin_amdgcn_ballot_w64(big) & ~__builtin_amdgcn_ballot_w64(cone)) | __builtin_amdgcn_ballot_w64(g1) |
                    __builtin_amdgcn_ballot_w64(g2);
            } else if constexpr (STAGE == kStageMerged) {
                // the angle test of sweep 2 and, from the same two projections, a conservative test on the concentrations
                // under a stain matrix that is only known to lie in a box around its sample estimate (MergedConc)
                const float2 er = g.v[3 * px], eg = g.v[3 * px + 1], eb = g.v[3 * px + 2];
                const bool tc = is_tissue_f(er.x, eg.x, eb.x, ylimf);
                const float x = fmaf(K.V[4], eb.y, fmaf(K.V[2], eg.y, K.V[0] * er.y));
                const float y = fmaf(K.V[5], eb.y, fmaf(K.V[3], eg.y, K.V[1] * er.y));
                const float d = x + fabsf(y);
                const float t0 = fmaf(nhi0m, d, y), t1 = fmaf(nlo1m, d, y);
                const bool pp = fminf(fminf(x, t0), -t1) > 0.0f;
                const float a1 = fmaf(K.u[0][1], y, fmaf(K.u[0][0], x, K.kt[0]));
                const float a2 = fmaf(K.u[1][1], y, fmaf(K.u[1][0], x, K.kt[1]));
                const float sa = fabsf(a1) + fabsf(a2);
                const bool g1 = fmaf(K.eps[0], sa, a1) >= K.thr[0], g2 = fmaf(K.eps[1], sa, a2) >= K.thr[1];
                m = (__builtin_amdgcn_ballot_w64(tc) & ~__builtin_amdgcn_ballot_w64(pp)) | __builtin_amdgcn_ballot_w64(g1) |
                    __builtin_amdgcn_ballot_w64(g2);
            } else {
                float a1, a2;
                lasso_interior(K.L, g.v[3 * px], g.v[3 * px + 1], g.v[3 * px + 2], a1, a2);
                const bool g1 = a1 >= clo0, g2 = a2 >= clo1;
                m = __builtin_amdgcn_ballot_w64(g1) | __builtin_amdgcn_ballot_w64(g2);
#ifdef SL_DEBUG_EXTRA_MATH
                {   // development aid: the same arithmetic once more (a VALU-bound sweep slows down in proportion)
                    float b1, b2;
                    lasso_interior(K.L, g.v[3 * px + 1], g.v[3 * px + 2], g.v[3 * px], b1, b2);
                    const bool h1 = b1 >= 1e30f, h2 = b2 >= 1e30f;
                    m |= __builtin_amdgcn_ballot_w64(h1) | __builtin_amdgcn_ballot_w64(h2);
                }
#endif
            }
            if (TAIL) {
                const bool inb = (cc < c1) & (ALIGNED | ((size_t)cc * 4 + px < (size_t)P));
                m &= __builtin_amdgcn_ballot_w64(inb);
            }
            sink.put(m, ch, px, lane);
        }
    };
    Chunk cur[kTrip], nx[kTrip];                             // see moments_sweep
#pragma unroll
    for (int k = 0; k < kTrip; ++k) { cur[k] = fetch(w0 + lane + k * nthreads); nx[k] = fetch(w0 + lane + (kTrip + k) * nthreads); }
    GatherT g[2];
    g[0] = gather(cur[0]);
    auto trip = [&](auto tail_tag, int cb) {
#pragma unroll
        for (int k = 0; k < kTrip; ++k) {
            const Chunk ch = cur[k];
            if (k + 1 < kTrip) {
                g[(k + 1) & 1] = gather(cur[k + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < kTrip; ++j) { cur[j] = nx[j]; nx[j] = fetch(cb + lane + (2 * kTrip + j) * nthreads); }
                g[0] = gather(cur[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            compute(tail_tag, ch, g[k & 1], cb + k * nthreads + lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int lim = ALIGNED ? c1 : min(c1, P >> 2);          // chunks made of in-range pixels only
    int cb = w0;
    for (; cb + (kTrip - 1) * nthreads + 64 <= lim; cb += nthreads * kTrip) trip(std::false_type{}, cb);
    if (cb < c1) trip(std::true_type{}, cb);                    // at most one ragged trip per wave
}

// ---- key functors handed BY VALUE to the selection primitives ----
// pseudo-angle of sample entry b (NaN: not tissue / beyond the tile)
struct SampleAngleKey {
    const uint32_t* sample; TabView tab; float V[6]; int cps_log2; int P; float ylimf;
    __device__ __forceinline__ float operator()(int b) const {
        if (sample_absent(b, cps_log2, P)) return nan_f();
        return of_word(sample[b]);
    }
    // the key of a sample word already in a register; branch-free (NaN = not a tissue pixel)
    __device__ __forceinline__ float of_word(uint32_t s) const {
        const uint32_t r = s & 255u, g = (s >> 8) & 255u, bl = (s >> 16) & 255u;
        const bool tissue = is_tissue_f(tab.gam(r), tab.gam(g), tab.gam(bl), ylimf);
        const float k = angle_key(V, tab.odf(r), tab.odf(g), tab.odf(bl));
        return tissue ? k : nan_f();
    }
    __device__ __forceinline__ bool present(int b, int n_sample) const { return b < n_sample && !sample_absent(b, cps_log2, P); }
};
// concentration `col` of sample entry b (all pixels, tissue or not)
struct SampleConcKey {
    const uint32_t* sample; TabView tab; LassoK L; int cps_log2; int P; int col;
    __device__ __forceinline__ void both(int b, float& c1, float& c2) const {      // NaN, NaN: entry absent
        if (sample_absent(b, cps_log2, P)) { c1 = c2 = nan_f(); return; }
        of_word(sample[b], c1, c2);
    }
    __device__ __forceinline__ void of_word(uint32_t s, float& c1, float& c2) const {
        lasso2(L, tab.odf(s & 255u), tab.odf((s >> 8) & 255u), tab.odf((s >> 16) & 255u), c1, c2);
    }
    __device__ __forceinline__ bool present(int b, int n_sample) const { return b < n_sample && !sample_absent(b, cps_log2, P); }
    __device__ __forceinline__ float operator()(int b) const {
        float c1, c2;
        both(b, c1, c2);
        return col == 0 ? c1 : c2;
    }
};
// keys of pixel p of a whole tile (exact fallback)
struct AngleTileKey {
    const uint8_t* src; TabView tab; float V[6]; float ylimf;
    __device__ __forceinline__ float operator()(int p) const {
        const uint32_t r = src[3 * (size_t)p], g = src[3 * (size_t)p + 1], b = src[3 * (size_t)p + 2];
        if (!is_tissue_f(tab.gam(r), tab.gam(g), tab.gam(b), ylimf)) return nan_f();
        return angle_key(V, tab.odf(r), tab.odf(g), tab.odf(b));
    }
};
struct ConcTileKey {
    const uint8_t* src; TabView tab; LassoK L; int col;
    __device__ __forceinline__ float operator()(int p) const {
        float c1, c2;
        lasso2(L, tab.odf(src[3 * (size_t)p]), tab.odf(src[3 * (size_t)p + 1]), tab.odf(src[3 * (size_t)p + 2]), c1, c2);
        return col == 0 ? c1 : c2;
    }
};
// exact keys (for bracket 0 and bracket 1) of raw candidate i
struct RawConcKey2 {
    const uint32_t* raw; TabView tab; LassoK L;
    __device__ __forceinline__ void operator()(int i, float& k0, float& k1) const {
        const uint32_t s = raw[i];
        lasso2(L, tab.odf(s & 255u), tab.odf((s >> 8) & 255u), tab.odf((s >> 16) & 255u), k0, k1);
    }
};

struct CandKey {
    const float* cand;
    __device__ __forceinline__ float operator()(int i) const { return cand[i]; }
};

// One pass over the raw candidates of a stage: exact key(s) of every raw pixel, #keys below each
// bracket, and the bracket members written compactly to cand[li][...] (<= cap_list each).
// key2(i, k0, k1) yields both keys of raw entry i.
template <class Key2>
__device__ __forceinline__ void wg_refine(int n_raw, const Key2& key2, const float* lo, const float* hi, float* cand0,
                                          float* cand1, uint32_t cap_list, uint32_t* n_lt /*[2]*/, uint32_t* n_in /*[2]*/,
                                          SelScratch& S, uint32_t* n_valid = nullptr /* entries whose first key is not NaN */) {
    if (threadIdx.x < 4) S.misc[12 + threadIdx.x] = 0;
    if (threadIdx.x == 4) S.misc[8] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t lt0 = 0, lt1 = 0, nv = 0;
    constexpr int U = 4;                                            // entries per lane and trip: one list-head update per trip (8: slower)
    const int step = (int)blockDim.x * U;
    for (int i0 = (int)(threadIdx.x - lane) * U; i0 < n_raw; i0 += step) {      // wave-uniform trip count
        float k0[U], k1[U];
        unsigned long long m0[U], m1[U];
        uint32_t tot0 = 0, tot1 = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 64 + lane;
            k0[u] = k1[u] = nan_f();
            if (i < n_raw) key2(i, k0[u], k1[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            lt0 += k0[u] < lo[0] ? 1u : 0u;
            lt1 += k1[u] < lo[1] ? 1u : 0u;
            nv += k0[u] == k0[u] ? 1u : 0u;
            m0[u] = __ballot((k0[u] >= lo[0]) & (k0[u] <= hi[0]));
            m1[u] = __ballot((k1[u] >= lo[1]) & (k1[u] <= hi[1]));
            tot0 += (uint32_t)__popcll(m0[u]);
            tot1 += (uint32_t)__popcll(m1[u]);
        }
        if (tot0) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&S.misc[14], tot0);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0[u], 0));
                if (((m0[u] >> lane) & 1ull) && pos < cap_list) cand0[pos] = k0[u];
                base += (uint32_t)__popcll(m0[u]);
            }
        }
        if (tot1) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&S.misc[15], tot1);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1[u], 0));
                if (((m1[u] >> lane) & 1ull) && pos < cap_list) cand1[pos] = k1[u];
                base += (uint32_t)__popcll(m1[u]);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { lt0 += __shfl_xor((int)lt0, o, 64); lt1 += __shfl_xor((int)lt1, o, 64); nv += __shfl_xor((int)nv, o, 64); }
    if (lane == 0) { if (lt0) atomicAdd(&S.misc[12], lt0); if (lt1) atomicAdd(&S.misc[13], lt1); if (nv) atomicAdd(&S.misc[8], nv); }
    __threadfence_block();
    __syncthreads();
    n_lt[0] = S.misc[12]; n_lt[1] = S.misc[13]; n_in[0] = S.misc[14]; n_in[1] = S.misc[15];
    if (n_valid) *n_valid = S.misc[8];
    __syncthreads();
}

// One pass over the n keys: how many lie below lo, how many inside [lo, hi], and the smallest and largest of those inside
// (ordered integers; 0xffffffff / 0 when none).  Ends with a barrier.
struct Census { uint32_t n_below, n_in, omin, omax; };
template <class KeyAt>
__device__ __noinline__ Census wg_bracket_census(int n, KeyAt key_at, float lo, float hi, SelScratch& S) {
    if (threadIdx.x == 0) { S.misc[4] = 0xffffffffu; S.misc[5] = 0; S.misc[6] = 0; S.misc[7] = 0; }
    __syncthreads();
    const uint32_t olo = f2ord(lo), ohi = f2ord(hi);
    uint32_t mn = 0xffffffffu, mx = 0, nb = 0, ni = 0;
    wg_for_each_key(n, key_at, [&](uint32_t o) {
        nb += o < olo ? 1u : 0u;
        if (o >= olo && o <= ohi) { ++ni; mn = min(mn, o); mx = max(mx, o); }
    });
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        nb += __shfl_xor((int)nb, o, 64);
        ni += __shfl_xor((int)ni, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&S.misc[4], mn); atomicMax(&S.misc[5], mx); atomicAdd(&S.misc[6], nb); atomicAdd(&S.misc[7], ni); }
    __syncthreads();
    const Census c{S.misc[6], S.misc[7], S.misc[4], S.misc[5]};
    __syncthreads();
    return c;
}

// Exact order statistics (k, k+1) of one bracket of a selection stage from the refined lists:
// lt = pixels proven or found below the bracket, n_in = members collected in cand[].  Falls back to
// exact selection over the whole tile when the bracket missed or a list was incomplete.
template <class TileKeyAt>
__device__ __forceinline__ void stage_order_stats(const float* cand, uint32_t n_in, uint32_t cap_list, bool complete, float lo, float hi,
                                                  long long lt, int P, const TileKeyAt& tile_key_at, uint32_t n,
                                                  long long k, float& xa, float& xb, int& fallbacks, SelScratch& S) {
    const long long k2 = (k + 1 < (long long)n) ? k + 1 : k;
    const bool covered = complete && k >= lt && k2 < lt + (long long)n_in;
    if (covered && lo == hi) {
        xa = xb = lo;                                  // every member of the bracket equals lo
    } else if (covered && n_in <= cap_list) {
        wg_select_pair_small((int)n_in, CandKey{cand}, (uint32_t)(k - lt), xa, xb, S);
        if (k2 == k) xb = xa;
    } else {                                           // exact, slow, rare
        // Mostly this is a run of ties (few-colour images: more equal keys than the lists hold).  One census pass over the
        // tile settles that case: if every key inside the bracket is the same value and both ranks fall on it, that value
        // is the answer; only otherwise the windowed selection (about six more passes) runs.
        const Census c = wg_bracket_census(P, tile_key_at, lo, hi, S);
        if (c.n_in > 0 && c.omin == c.omax && k >= (long long)c.n_below && k2 < (long long)c.n_below + (long long)c.n_in) {
            xa = xb = ord2f(c.omin);
        } else {
            wg_select_pair(P, tile_key_at, (uint32_t)k, xa, xb, S);
            if (k2 == k) xb = xa;
        }
        fallbacks += 1;
    }
}

// bin_b(k) = clamp((k - lo_b) sc_b, 0, 511): the 512-bin histogram of bracket b's members that wg_refine_s fills and wg_pick2 reads
struct PickScale { float lo[2], sc[2]; };
__device__ __forceinline__ int pick_bin(float k, float lo, float sc) { return min(511, max(0, (int)((k - lo) * sc))); }

// Exact order statistics krel[b] and krel[b] + 1 (0-based among the members of bracket b, both < n_in[b] unless has2[b] is
// false) for the brackets with want[b], from the histograms wg_refine_s left in S.hist: locate the bin of rank krel, gather
// that bin's keys (one pass over the member list, next trip in flight) and the smallest key beyond it, rank by brute force.
// done[b] = false when the bin holds more than 512 keys (ties / a degenerate spread): the caller takes the windowed path.
__device__ __forceinline__ void wg_pick2(const float* cand0, const float* cand1, const uint32_t* n_in, const bool* want, const uint32_t* krel,
                                         const PickScale& ps, float* xa /*[2]*/, float* xb /*[2]*/, bool* done /*[2]*/, SelScratch& S) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int b = 0; b < 2; ++b)
        if (wave == b && want[b]) wave_locate(S.hist + 512 * b, 512, krel[b], &S.misc[16 + 3 * b], lane);
    if (tid < 2) { S.misc[24 + tid] = 0; S.misc[26 + tid] = 0xffffffffu; }      // list fill, smallest key beyond the bin (ordered)
    __syncthreads();
    uint32_t bin[2], below[2], cnt[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) { bin[b] = S.misc[16 + 3 * b]; below[b] = S.misc[17 + 3 * b]; cnt[b] = S.misc[18 + 3 * b]; done[b] = want[b] && cnt[b] <= 512u && cnt[b] > 0u; }
    __syncthreads();                                                          // the histograms become the two key lists
    float* list = reinterpret_cast<float*>(S.hist);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (!done[b]) continue;                                               // block-uniform
        const float* cand = b ? cand1 : cand0;
        const int n = (int)n_in[b];
        constexpr int U = 4;
        const int bd = blockDim.x;
        uint32_t best = 0xffffffffu;
        float kn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = tid + u * bd; kn[u] = cand[i < n ? i : 0]; }
        for (int i0 = tid; i0 < n; i0 += U * bd) {
            float k[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                k[u] = kn[u];
                const int i = i0 + (U + u) * bd;
                kn[u] = cand[i < n ? i : 0];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + u * bd >= n) continue;
                const int kb = pick_bin(k[u], ps.lo[b], ps.sc[b]);
                if (kb == (int)bin[b]) { const uint32_t pos = atomicAdd(&S.misc[24 + b], 1u); if (pos < 512u) list[512 * b + pos] = k[u]; }
                else if (kb > (int)bin[b]) best = min(best, f2ord(k[u]));
            }
        }
        for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 64));
        if (lane == 0 && best != 0xffffffffu) atomicMin(&S.misc[26 + b], best);
    }
    __syncthreads();
    if (tid < 4) S.misc[28 + tid] = 0;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (!done[b]) continue;
        const uint32_t m = cnt[b], ra = krel[b] - below[b];
        if ((uint32_t)tid < m) {
            const float me = list[512 * b + tid];
            uint32_t r = 0;
            for (uint32_t j = 0; j < m; ++j) {
                const float o = list[512 * b + j];
                r += (o < me || (o == me && j < (uint32_t)tid)) ? 1u : 0u;
            }
            if (r == ra) S.misc[28 + 2 * b] = __float_as_uint(me);
            if (r == ra + 1) S.misc[29 + 2 * b] = __float_as_uint(me);
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (!done[b]) continue;
        xa[b] = __uint_as_float(S.misc[28 + 2 * b]);
        const bool same_bin = krel[b] - below[b] + 1 < cnt[b];
        const uint32_t nx = S.misc[26 + b];
        xb[b] = same_bin ? __uint_as_float(S.misc[29 + 2 * b]) : (nx != 0xffffffffu ? ord2f(nx) : xa[b]);
    }
    __syncthreads();
}

// Both brackets of a selection stage: the one-pass path where the bracket covers the wanted ranks and its member list is
// complete, the windowed / whole-tile paths of stage_order_stats otherwise.  lt[b] = pixels below bracket b (proven or counted).
template <bool TWO_COLS, class TileKeyAt>
__device__ __forceinline__ void stage_pick2(const float* cand0, const float* cand1, const uint32_t* n_in, uint32_t cap_list, bool complete,
                                            const float* lo, const float* hi, const long long* lt, int P, TileKeyAt tile_key_at, uint32_t n,
                                            const long long* k, const PickScale& ps, float* res /*[4]: xa0, xb0, xa1, xb1*/, int& fallbacks, SelScratch& S) {
    bool fast[2], has2[2];
    uint32_t krel[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const long long k2 = (k[b] + 1 < (long long)n) ? k[b] + 1 : k[b];
        has2[b] = k2 != k[b];
        const bool covered = complete && k[b] >= lt[b] && k2 < lt[b] + (long long)n_in[b];
        fast[b] = covered && lo[b] < hi[b] && n_in[b] <= cap_list;
        krel[b] = fast[b] ? (uint32_t)(k[b] - lt[b]) : 0u;
    }
    float xa[2] = {0, 0}, xb[2] = {0, 0};
    bool done[2] = {false, false};
    if (fast[0] | fast[1]) wg_pick2(cand0, cand1, n_in, fast, krel, ps, xa, xb, done, S);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (done[b]) {
            if (!has2[b]) xb[b] = xa[b];
        } else {
            if constexpr (TWO_COLS) tile_key_at.col = b;
            stage_order_stats(b ? cand1 : cand0, n_in[b], cap_list, complete, lo[b], hi[b], lt[b], P, tile_key_at, n, k[b], xa[b], xb[b], fallbacks, S);
        }
        res[2 * b] = xa[b]; res[2 * b + 1] = xb[b];
    }
}

// Raw candidates are staged per wave in LDS and written out in dense bursts; the tile's list head is
// touched once per burst.  Positions come from v_mbcnt on the row's lane mask: no atomics, no LDS round
// trip, the fill level stays in an SGPR.
// burst of a wave's staged candidates to the tile's list (cold: once per ~130 pixel rows; kept out of line so
// that the eight call sites of a trip stay small)
// buf_lds: LDS byte address of the wave's staging buffer (a flat pointer to LDS kept live across the sweep drives this
// hipcc into an illegal post-RA copy of src_shared_base)
__device__ __noinline__ void raw_flush(uint32_t buf_lds, uint32_t n, uint32_t* dst, unsigned int* head, uint32_t cap) {
#if defined(__HIP_DEVICE_COMPILE__)
    SL_LDS const uint32_t* buf = (SL_LDS const uint32_t*)buf_lds;
#else
    const uint32_t* buf = nullptr;
#endif
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(head, n);
    base = __builtin_amdgcn_readfirstlane(base);
    for (uint32_t i = lane; i < n; i += 64)
        if (base + i < cap) dst[base + i] = buf[i];
}

struct RawSink {
    uint32_t buf;               // LDS byte address of this wave's kStageWave entries
    uint32_t n;                 // wave-uniform fill
    uint32_t* dst;              // global raw list of the tile
    unsigned int* head;         // list head (LDS in the fused kernel, global otherwise)
    unsigned int* overflow;     // (unused by this sink: an over-full list shows as head > cap)
    uint32_t cap;               // capacity of dst
    uint32_t stage_cap;         // entries of the staging buffer (>= 64)
    __device__ __forceinline__ void flush(int) {
        if (n != 0) raw_flush(buf, n, dst, head, cap);
        n = 0;
    }
    // One pixel row of the wave: m = lane mask of the flagged lanes (a wave-uniform value).  Branch-free on the hot
    // path: the masked LDS write is an asm block that swaps EXEC itself (measured: the three branches per row of
    // the structured version cost more than all the arithmetic of the sweep).
    __device__ __forceinline__ void put(unsigned long long m, const Chunk& ch, int px, int lane) {
        put_value(m, chunk_pixel(ch, px) & 0xffffffu, lane);
    }
    // the same for any 32-bit value of the flagged lanes
    __device__ __forceinline__ void put_value(unsigned long long m, uint32_t value, int lane) {
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (__builtin_expect(n + cnt > stage_cap, 0)) flush(lane);   // rare, out of line; a row holds <= 64 entries
        // rank of this lane among the flagged lanes; the fill level joins the buffer address on the scalar unit (as v_mbcnt's
        // addend it cost a v_mov per row: two SGPR operands do not fit one VOP3)
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t sbase;                                      // buf + 4 n on the scalar unit (the compiler would fold it back into the vector side)
        asm("s_lshl2_add_u32 %0, %1, %2" : "=s"(sbase) : "s"(n), "s"(buf) : "scc");
        const uint32_t addr = sbase + 4u * rank;
        unsigned long long saved;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "s"(m), "v"(addr), "v"(value) : "memory");
#else
        (void)rank; (void)value;
#endif
        n += cnt;
    }
};

// ------------------------------------------------------------------------------------------
// Finish 2 of the fused kernel, one pass per key family (round 3).
//
// What the first version of this step cost was not its arithmetic but its memory round trips: every trip of the refine loop
// loaded its raw words, appended the bracket members to the global lists (an LDS atomic with return per list, then global
// stores) and -- vmcnt completes in order on gfx9 and the number of conditional stores is unknown at compile time -- waited
// for ALL of it at the top of the next trip: ~5 us per trip on a chip whose memory system is saturated by the neighbours'
// sweeps, 31 trips per pass.  Here the hot loop issues no global store at all:
//   * the 64 KB row table is not needed between the sweeps, so during finish 2 its space holds a 2 KB one-copy table
//     {gamma, od32}[256] (bank conflicts instead of 32 copies: the finish steps are not LDS bound) and, per wave, two staging
//     lists of 992 keys; a list is written out when it fills (about twice per wave and pass) and at the end;
//   * the next trip's raw words are in flight while a trip is evaluated;
//   * the members are counted into a 512-bin histogram per bracket on the way (masked ds_add, no return), from which
//     wg_pick2 takes the order statistics with ONE more pass over the member list instead of three;
//   * all counts are popcounts of ballots on the scalar unit.
// The row table is rebuilt (fill_b) before the next sweep.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kFinTabBytes = 2048;                                  // 256 x {gamma, od32}
__device__ __forceinline__ uint32_t fin_stage_bytes(int nthreads) { return (uint32_t)((sizeof(RowTab) - kFinTabBytes) / (size_t)(nthreads / 64)); }   // per wave

struct FinTab {                 // reader of the one-copy table at LDS byte address `base`
    uint32_t base;
    // byte offset of the entry of byte k (0..2) of a raw word r | g << 8 | b << 16
    __device__ __forceinline__ uint32_t addr(uint32_t w, int k) const { return k == 0 ? ((w << 3) & 0x7f8u) : ((w >> (8 * k - 3)) & 0x7f8u); }
#if defined(__HIP_DEVICE_COMPILE__)
    __device__ __forceinline__ float2 gam_odf(uint32_t a) const {
        const v2f v = *(SL_LDS const v2f*)(base + a);
        return make_float2(v.x, v.y);
    }
    __device__ __forceinline__ float odf(uint32_t a) const { return *(SL_LDS const float*)(base + a + 4u); }
#else
    float2 gam_odf(uint32_t) const { return float2{}; }
    float odf(uint32_t) const { return 0.0f; }
#endif
    __device__ __forceinline__ TabView view() const { return TabView{base, 8u, 4u, 0u}; }      // for the TabView key functors (exact fallbacks)
};
// all threads: builds the one-copy table in the first 2 KB of the row table from the row table itself (layout B)
__device__ __forceinline__ void fin_tab_build(RowTab& tab) {
    float2* p = reinterpret_cast<float2*>(tab.e);
    float2 e = make_float2(0.0f, 0.0f);
    if (threadIdx.x < 256) e = p[threadIdx.x * 32];
    __syncthreads();
    if (threadIdx.x < 256) p[threadIdx.x] = e;
    __syncthreads();
}

// all threads: the row table (layout B) back from the one-copy table -- LDS to LDS, the constants are not fetched again
template <int NT>
__device__ __forceinline__ void fin_tab_expand(RowTab& tab) {
    float2* p = reinterpret_cast<float2*>(tab.e);
    constexpr int PER = 256 * 32 / NT;                    // entries per thread; thread t writes t, t + NT, ...: values t/32 + j NT/32
    float2 e[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) e[j] = p[threadIdx.x / 32 + j * (NT / 32)];
    __syncthreads();                                      // every read of the one-copy table precedes the first write over it
#pragma unroll
    for (int j = 0; j < PER; ++j) p[threadIdx.x + j * NT] = e[j];
    __syncthreads();
}

// keys of a raw word (angle_key / lasso2 as in the tile-key functors: every path must select the same values)
struct WordAngleKey {           // one pseudo-angle serves both brackets; valid = tissue
    FinTab T; float V[6]; float ylimf;
    __device__ __forceinline__ void of_word(uint32_t s, float& k0, float& k1, bool& valid) const {
        const float2 er = T.gam_odf(T.addr(s, 0)), eg = T.gam_odf(T.addr(s, 1)), eb = T.gam_odf(T.addr(s, 2));
        valid = is_tissue_f(er.x, eg.x, eb.x, ylimf);
        k0 = k1 = angle_key(V, er.y, eg.y, eb.y);
    }
};
struct WordConcKey {
    FinTab T; LassoK L;
    __device__ __forceinline__ void of_word(uint32_t s, float& k0, float& k1, bool& valid) const {
        lasso2(L, T.odf(T.addr(s, 0)), T.odf(T.addr(s, 1)), T.odf(T.addr(s, 2)), k0, k1);
        valid = true;
    }
};

// ds_add_u32 of `one` at LDS byte address `addr` for the lanes of mask m (no return value, nothing to wait for)
__device__ __forceinline__ void lds_count_masked(unsigned long long m, uint32_t addr, uint32_t one) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_add_u32 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "s"(m), "v"(addr), "v"(one) : "memory");
#else
    (void)m; (void)addr; (void)one;
#endif
}

// One pass over the raw candidates: counts below each bracket, bracket members to cand0 / cand1 (through the wave's two
// staging lists at LDS byte address stage_lds, stage_entries keys each) and into the histograms of S.hist.
struct RefineOut { uint32_t n_lt[2], n_in[2], n_valid; PickScale ps; };
template <class WordKey2>
__device__ __forceinline__ RefineOut wg_refine_s(const uint32_t* raw, int n_raw, const WordKey2& key2, float lo0, float hi0, float lo1, float hi1,
                                                 float* cand0, float* cand1, uint32_t cap_list, uint32_t stage_lds, uint32_t stage_entries,
                                                 SelScratch& S) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 5) S.misc[8 + tid] = 0;                                // [8] valid, [9] lt0, [10] lt1, [11] in0, [12] in1
    for (int i = tid; i < 1024; i += blockDim.x) S.hist[i] = 0;
    RefineOut r;
    r.ps.lo[0] = lo0; r.ps.lo[1] = lo1;
    r.ps.sc[0] = (hi0 > lo0 && lo0 > -INFINITY && hi0 < INFINITY) ? 512.0f * 0.999999f / (hi0 - lo0) : 0.0f;
    r.ps.sc[1] = (hi1 > lo1 && lo1 > -INFINITY && hi1 < INFINITY) ? 512.0f * 0.999999f / (hi1 - lo1) : 0.0f;
    __syncthreads();
    stage_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)stage_lds);
    RawSink s0{stage_lds, 0u, reinterpret_cast<uint32_t*>(cand0), &S.misc[11], nullptr, cap_list, stage_entries};
    RawSink s1{stage_lds + 4u * stage_entries, 0u, reinterpret_cast<uint32_t*>(cand1), &S.misc[12], nullptr, cap_list, stage_entries};
    const float vlo0 = in_vgpr(lo0), vhi0 = in_vgpr(hi0), vlo1 = in_vgpr(lo1), vhi1 = in_vgpr(hi1);
    const float psl0 = in_vgpr(r.ps.lo[0]), psc0 = in_vgpr(r.ps.sc[0]), psl1 = in_vgpr(r.ps.lo[1]), psc1 = in_vgpr(r.ps.sc[1]);
    const uint32_t hist_lds = lds_address(S.hist);
    uint32_t one = 1u;
    asm("" : "+v"(one));
    uint32_t lt0 = 0, lt1 = 0, nv = 0;                               // wave-uniform
    constexpr int U = 4;                                             // raw words per lane and trip
    const int step = (int)blockDim.x * U;
    const int last = n_raw > 0 ? n_raw - 1 : 0;
    int i0 = (tid - lane) * U;                                       // wave-uniform trip count
    uint32_t wn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) wn[u] = raw[min(i0 + u * 64 + lane, last)];
    for (; i0 < n_raw; i0 += step) {
        uint32_t w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            w[u] = wn[u];
            wn[u] = raw[min(i0 + step + u * 64 + lane, last)];       // next trip (clamped, never predicated)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float k0, k1;
            bool valid;
            key2.of_word(w[u], k0, k1, valid);
            const bool inb = i0 + u * 64 + lane < n_raw;
            const unsigned long long mv = __builtin_amdgcn_ballot_w64(valid) & __builtin_amdgcn_ballot_w64(inb);
            const unsigned long long l0 = __builtin_amdgcn_ballot_w64(k0 < vlo0) & mv, l1 = __builtin_amdgcn_ballot_w64(k1 < vlo1) & mv;
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(k0 <= vhi0) & ~l0 & mv, m1 = __builtin_amdgcn_ballot_w64(k1 <= vhi1) & ~l1 & mv;
            nv += (uint32_t)__popcll(mv);
            lt0 += (uint32_t)__popcll(l0);
            lt1 += (uint32_t)__popcll(l1);
            lds_count_masked(m0, hist_lds + 4u * (uint32_t)pick_bin(k0, psl0, psc0), one);
            lds_count_masked(m1, hist_lds + 2048u + 4u * (uint32_t)pick_bin(k1, psl1, psc1), one);
            s0.put_value(m0, __float_as_uint(k0), lane);
            s1.put_value(m1, __float_as_uint(k1), lane);
        }
    }
    s0.flush(lane);
    s1.flush(lane);
    if (lane == 0) { if (nv) atomicAdd(&S.misc[8], nv); if (lt0) atomicAdd(&S.misc[9], lt0); if (lt1) atomicAdd(&S.misc[10], lt1); }
    __threadfence_block();
    __syncthreads();
    r.n_lt[0] = S.misc[9]; r.n_lt[1] = S.misc[10]; r.n_in[0] = S.misc[11]; r.n_in[1] = S.misc[12]; r.n_valid = S.misc[8];
    __syncthreads();
    return r;
}

// ------------------------------------------------------------------------------------------
// Vahadane: sparse-NMF dictionary (vahadane_stain_extractor.py:35-36, spams.trainDL K=2, lambda1,
// posAlpha, posD, unit-ball atoms) by CLASS MOMENTS.
//
// For a fixed dictionary D the exact non-negative code of a pixel is affine in its OD vector x once
// its active set is known: alpha = P_c (D x - lambda 1), c in {both atoms, atom 1 only, atom 2 only,
// none}.  Hence A = sum alpha alpha^T and B = sum x alpha^T -- all the online-dictionary-learning
// update needs (Mairal et al. 2010, Alg. 2) -- are closed-form functions of D and of the per-class
// moments {n_c, sum x, sum x x^T}.  One sweep over the tile classifies the pixels under the current D
// and accumulates 3 x 10 moment sums; one lane then iterates the block-coordinate dictionary update
// on those 30 numbers until it stalls (no pixel is touched); the next sweep re-classifies.  The
// fixed point is the one plain full-batch block-coordinate descent reaches (oracle:
// vahadane_dictionary), but in ~9 sweeps instead of ~90.
// ------------------------------------------------------------------------------------------
// ---- class moments in binary32 bursts --------------------------------------------------------------------------------
// The classification (which of the code's active sets a pixel falls in) and the nine moment products are binary32; a lane
// sums them over kDictBurstTrips trips (128 pixels), then the wave adds its 64 lanes' bursts (DPP, binary32) into its
// binary64 row of workgroup memory.  Per pixel: 12 fast FMAs + 5 compares to classify, 9 fast FMAs under the class's exec
// mask to accumulate -- the binary64 version issued 41 binary64 instructions (4 cycles each, both pipes blocked) and three
// 16-byte LDS gathers.  A pixel next to a class boundary may land on the other side than in exact arithmetic; the code is
// continuous across the boundary, so its contribution moves by its distance to the boundary (~1e-7): far below dl_tol.
// The bursts cover the same pixels in the fused kernel and in the per-phase kernels (parts are aligned to
// kDictBurstTrips trips of a 512-thread workgroup): both schedules iterate the same map.
constexpr int kDictTrip = 2;             // chunks per lane and trip in the dictionary sweeps (register pressure: 27 burst sums live)
constexpr int kDictBurstTrips = 16;      // trips per burst: 16 x 2 x 512 chunks = 64 Ki pixels per workgroup = 128 pixels per lane
constexpr int kDictAlignTrips = kDictBurstTrips * kDictTrip / 4;   // the same span in units of the sweep kernels' 4-chunk trips (part_range)
struct ClsBurst {
    float sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    __device__ __forceinline__ void add(float x, float y, float z) {
        sx += x; sy += y; sz += z;
        sxx = fmaf(x, x, sxx); sxy = fmaf(x, y, sxy); sxz = fmaf(x, z, sxz);
        syy = fmaf(y, y, syy); syz = fmaf(y, z, syz); szz = fmaf(z, z, szz);
    }
};

// sum over the 64 lanes of a wave, valid in lane 63 (DPP: no LDS traffic)
__device__ __forceinline__ float wave_total_f32(float v) {
#define SL_DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false))
    SL_DPP_ADD(0x111, 0xf);      // row_shr:1
    SL_DPP_ADD(0x112, 0xf);      // row_shr:2
    SL_DPP_ADD(0x114, 0xf);      // row_shr:4
    SL_DPP_ADD(0x118, 0xf);      // row_shr:8   -> lane 15 of every row holds the row's sum
    SL_DPP_ADD(0x142, 0xa);      // row_bcast:15 -> rows 1, 3
    SL_DPP_ADD(0x143, 0xc);      // row_bcast:31 -> rows 2, 3: lane 63 holds the wave's sum
#undef SL_DPP_ADD
    return v;
}

// the dictionary as the classification needs it (VGPR-resident)
struct DictK { float m1[3], m2[3], nlam, g11, g12, g22; };
__device__ __forceinline__ void dict_consts(const double* D, double lam, DictK& k) {
    for (int c = 0; c < 3; ++c) { k.m1[c] = in_vgpr((float)D[c]); k.m2[c] = in_vgpr((float)D[3 + c]); }
    k.nlam = in_vgpr((float)(-lam));
    k.g11 = in_vgpr((float)(D[0] * D[0] + D[1] * D[1] + D[2] * D[2]));
    k.g22 = in_vgpr((float)(D[3] * D[3] + D[4] * D[4] + D[5] * D[5]));
    k.g12 = in_vgpr((float)(D[0] * D[3] + D[1] * D[4] + D[2] * D[5]));
}

// the wave's binary64 row: [class][n, s(3), q(6)] for classes both / only-1 / only-2, then [30] = tissue pixels
struct DictWaveAcc {
    ClsBurst b[3];
    uint32_t n[3] = {0, 0, 0};            // wave-uniform counts of the current burst
    uint32_t n_tissue = 0;
    double* row;                          // 32 doubles of workgroup memory owned by this wave
    __device__ __forceinline__ void begin(double* r, int lane) {
        row = r;
        if (lane < 32) row[lane] = 0.0;
    }
    __device__ __forceinline__ void flush(int lane) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v[9] = {b[c].sx, b[c].sy, b[c].sz, b[c].sxx, b[c].sxy, b[c].sxz, b[c].syy, b[c].syz, b[c].szz};
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const float t = wave_total_f32(v[i]);
                if (lane == 63) row[10 * c + 1 + i] += (double)t;
            }
            if (lane == 63) row[10 * c] += (double)n[c];
            b[c] = ClsBurst{};
            n[c] = 0;
        }
        if (lane == 63) row[30] += (double)n_tissue;
        n_tissue = 0;
    }
    // one pixel: tissue = the lane's pixel counts; od = (x, y, z)
    // The active set from the NUMERATORS of the interior solution (b = D x - lambda; a = G^-1 b has the signs of
    // n1 = g22 b1 - g12 b2, n2 = g11 b2 - g12 b1): two nearly parallel atoms (early sweeps) make G^-1 large and a binary32
    // a = W x + k cancels catastrophically, the numerators do not.  only-1 holds when b1 > 0 and the gradient with respect
    // to the second code at (b1/g11, 0) is non-positive, i.e. n2 <= 0.
    __device__ __forceinline__ void pixel(const DictK& L, bool tissue, float x, float y, float z) {
        const float b1 = fmaf(L.m1[2], z, fmaf(L.m1[1], y, fmaf(L.m1[0], x, L.nlam)));
        const float b2 = fmaf(L.m2[2], z, fmaf(L.m2[1], y, fmaf(L.m2[0], x, L.nlam)));
        const float n1 = fmaf(L.g22, b1, -L.g12 * b2), n2 = fmaf(L.g11, b2, -L.g12 * b1);
        const bool both = tissue & (n1 >= 0.0f) & (n2 >= 0.0f);
        const bool only1 = tissue & !both & (b1 > 0.0f) & (n2 <= 0.0f);
        const bool only2 = tissue & !both & !only1 & (b2 > 0.0f);
        n_tissue += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(tissue));
        n[0] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(both));
        n[1] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(only1));
        n[2] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(only2));
        if (both) b[0].add(x, y, z);
        if (only1) b[1].add(x, y, z);
        if (only2) b[2].add(x, y, z);
    }
};

// classify every tissue pixel of chunks [c0,c1) under the dictionary L and accumulate the class moments into acc (its row
// must have been begun; the caller flushes nothing: the sweep ends flushed).  c0 must be a multiple of 64 (of
// kDictBurstTrips trips for schedule-independent bursts).  Structure of moments_sweep_b.
template <bool ALIGNED, int kTrip, bool STREAM = false>
__device__ __forceinline__ void dict_sweep_b(const uint8_t* src, int P, int c0, int c1, int t, int nthreads, const TabReaderB& T,
                                             float ylimf, const DictK& L, DictWaveAcc& acc) {
    const size_t nbytes = (size_t)P * 3;
    const int lane = t & 63;
    const int w0 = __builtin_amdgcn_readfirstlane(c0 + (t & ~63));
    struct G { float2 v[12]; };
    auto fetch = [&](int cc) { return load_chunk_clamped<ALIGNED, STREAM>(src, nbytes, cc, c1); };
    auto gather = [&](const Chunk& ch) {
        G g;
#pragma unroll
        for (int i = 0; i < 12; ++i) g.v[i] = T.gam_odf(T.addr(ch, i));
        return g;
    };
    auto compute = [&](auto tail_tag, const G& g, int cc) {
        constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const float2 er = g.v[3 * px], eg = g.v[3 * px + 1], eb = g.v[3 * px + 2];
            bool tissue = is_tissue_f(er.x, eg.x, eb.x, ylimf);
            if (TAIL) tissue = tissue & (cc < c1) & (ALIGNED | ((size_t)cc * 4 + px < (size_t)P));
            acc.pixel(L, tissue, er.y, eg.y, eb.y);
        }
    };
    // (no gather look-ahead here: the 27 burst sums leave no room for a second set of table values, and the sweep is bound by
    //  its ~60 vector instructions per pixel, not by the LDS latency the other waves of the SIMD cover)
    Chunk cur[kTrip], nx[kTrip];
#pragma unroll
    for (int k = 0; k < kTrip; ++k) { cur[k] = fetch(w0 + lane + k * nthreads); nx[k] = fetch(w0 + lane + (kTrip + k) * nthreads); }
    int trips = 0;
    auto trip = [&](auto tail_tag, int cb) {
#pragma unroll
        for (int k = 0; k < kTrip; ++k) {
            const G g = gather(cur[k]);
            compute(tail_tag, g, cb + k * nthreads + lane);
        }
#pragma unroll
        for (int j = 0; j < kTrip; ++j) { cur[j] = nx[j]; nx[j] = fetch(cb + lane + (2 * kTrip + j) * nthreads); }
        if (++trips == kDictBurstTrips) { acc.flush(lane); trips = 0; }          // wave-uniform
    };
    const int lim = ALIGNED ? c1 : min(c1, P >> 2);
    int cb = w0;
    for (; cb + (kTrip - 1) * nthreads + 64 <= lim; cb += nthreads * kTrip) trip(std::false_type{}, cb);
    if (cb < c1) trip(std::true_type{}, cb);
    if (trips) acc.flush(lane);
}

// The stratified sample of a tile WITHOUT a sweep: entry b is the pixel sample_row() would keep for block b, fetched directly
// (one 4-byte load per entry from a different 128-byte line each: ~2/3 of the tile's lines are touched, but nothing is
// computed).  The Vahadane path starts from it: the dictionary is first iterated on the sample, and every full sweep then
// starts near the fixed point.  Entries whose pixel lies beyond the tile stay unwritten (readers test sample_absent).
template <bool ALIGNED>
__device__ __forceinline__ void gather_sample(const uint8_t* src, int P, int stride_log2, uint32_t* samp, int n_sample, int t, int nthreads) {
    const int cps_log2 = stride_log2 - 2;
    for (int b = t; b < n_sample; b += nthreads) {
        const long long px = sample_pixel((uint32_t)b, cps_log2);
        if (px >= P) continue;
        const uint8_t* q = src + 3 * (size_t)px;
        uint32_t v;
        if (ALIGNED && (px & 3) == 0) v = *(const uint32_t*)q;                                   // pixel 0 of its chunk: {r, g, b, stray}
        else if (ALIGNED) v = *(const uint32_t*)(q - 1) >> 8;                                   // pixel 3: {b of pixel 2, r, g, b} >> 8
        else v = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16);
        samp[b] = v;
    }
}

// the same classification + accumulation over the tile's stratified SAMPLE (tissue entries only): a 1/64-cost stand-in for
// a full sweep, used to bring D close to its fixed point before touching the tile again.  Always walked by 512 "virtual
// lanes" (threads beyond 511 idle) so that the bursts do not depend on the workgroup size of the calling kernel.
__device__ __forceinline__ void dict_sweep_sample_b(const uint32_t* samp, int n_sample, int stride_log2, int P, int t,
                                                    const TabReaderB& T, float ylimf, const DictK& L, DictWaveAcc& acc) {
    const int lane = t & 63;
    const int cps_log2 = stride_log2 - 2;
    if (t < 512) {                                                      // wave-uniform
        for (int b0 = t & ~63; b0 < n_sample; b0 += 512) {
            const int b = b0 + lane;
            const bool have = b < n_sample && !sample_absent(b, cps_log2, P);
            const uint32_t s = samp[have ? b : 0];                      // unconditional load (n_sample >= 1); `have` masks the result
            const float2 er = T.gam_odf(T.addr(s, 0)), eg = T.gam_odf(T.addr(s, 1)), eb = T.gam_odf(T.addr(s, 2));
            acc.pixel(L, have & is_tissue_f(er.x, eg.x, eb.x, ylimf), er.y, eg.y, eb.y);
        }
    }
    acc.flush(lane);
}

// A (2x2) and B (3x2) of the dictionary update from the class moments m[c] = {n, s(3), q(6)}: with the codes of a
// class written as alpha = W x - w (W = P D, w = lam P 1, P the class's inverse Gram block),
//   A = sum_c  W S W' - (W s) w' - w (W s)' + n w w',     B = sum_c  S W' - s w'.
// Class 0 (both stains active) has a full P; classes 1 / 2 (one stain) have a single non-zero entry, so only
// A[0][0], B[:,0] resp. A[1][1], B[:,1] receive anything.  One lane runs this several hundred times per tile, so
// its latency is a fixed cost of every tile: the one-stain classes are written out (a third of the generic
// arithmetic) and the binary64 divisions (~100 dependent cycles each) are three reciprocals.
// WITH_SA: also 1' sum alpha (for dict_objective), from the same W s - n w the class blocks form anyway.
template <bool WITH_SA = false>
__device__ __forceinline__ void ab_from_class_moments(const double* mom /*[3][10]*/, const double (&D)[2][3], double lam,
                                                      double (&A)[2][2], double (&B)[3][2], double* sa = nullptr) {
    double sa_ = 0.0;
    const double g11 = D[0][0] * D[0][0] + D[0][1] * D[0][1] + D[0][2] * D[0][2];
    const double g22 = D[1][0] * D[1][0] + D[1][1] * D[1][1] + D[1][2] * D[1][2];
    const double g12 = D[0][0] * D[1][0] + D[0][1] * D[1][1] + D[0][2] * D[1][2];
    const double rdet = 1.0 / (g11 * g22 - g12 * g12), r11 = 1.0 / g11, r22 = 1.0 / g22;
    A[0][0] = A[0][1] = A[1][0] = A[1][1] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) B[k][0] = B[k][1] = 0.0;
    {   // ---- class 0: both active
        const double* m = mom;
        const double n = m[0];
        if (n > 0) {
            const double P00 = g22 * rdet, P01 = -g12 * rdet, P11 = g11 * rdet;
            const double s1[3] = {m[1], m[2], m[3]};
            const double S2[3][3] = {{m[4], m[5], m[6]}, {m[5], m[7], m[8]}, {m[6], m[8], m[9]}};
            double W[2][3], Ws1[2], WS2[2][3];
            const double w[2] = {lam * (P00 + P01), lam * (P01 + P11)};
#pragma unroll
            for (int k = 0; k < 3; ++k) { W[0][k] = P00 * D[0][k] + P01 * D[1][k]; W[1][k] = P01 * D[0][k] + P11 * D[1][k]; }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                Ws1[r] = W[r][0] * s1[0] + W[r][1] * s1[1] + W[r][2] * s1[2];
#pragma unroll
                for (int k = 0; k < 3; ++k) WS2[r][k] = W[r][0] * S2[0][k] + W[r][1] * S2[1][k] + W[r][2] * S2[2][k];
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = r; q < 2; ++q)
                    A[r][q] = WS2[r][0] * W[q][0] + WS2[r][1] * W[q][1] + WS2[r][2] * W[q][2] - Ws1[r] * w[q] - w[r] * Ws1[q] +
                              n * w[r] * w[q];
            A[1][0] = A[0][1];                                              // S is symmetric
            if (WITH_SA) sa_ += Ws1[0] + Ws1[1] - n * (w[0] + w[1]);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int r = 0; r < 2; ++r) B[k][r] = WS2[r][k] - s1[k] * w[r];
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {   // ---- class 1 + j: only stain j active, alpha_j = (D_j . x - lam) / g_jj
        const double* m = mom + 10 * (1 + j);
        const double n = m[0];
        if (n > 0) {
            const double rg = j == 0 ? r11 : r22, w = lam * rg;
            const double s1[3] = {m[1], m[2], m[3]};
            const double S2[3][3] = {{m[4], m[5], m[6]}, {m[5], m[7], m[8]}, {m[6], m[8], m[9]}};
            double W[3], WS2[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) W[k] = rg * D[j][k];
            const double Ws1 = W[0] * s1[0] + W[1] * s1[1] + W[2] * s1[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) WS2[k] = W[0] * S2[0][k] + W[1] * S2[1][k] + W[2] * S2[2][k];
            A[j][j] += WS2[0] * W[0] + WS2[1] * W[1] + WS2[2] * W[2] - 2.0 * Ws1 * w + n * w * w;
#pragma unroll
            for (int k = 0; k < 3; ++k) B[k][j] += WS2[k] - s1[k] * w;
            if (WITH_SA) sa_ += Ws1 - n * w;
        }
    }
    if (WITH_SA) *sa = sa_;
}

// The dictionary objective at D, up to a constant, from the class moments of D's OWN partition (the moments a sweep
// under D returns): with alpha the exact codes,
//   sum_i 1/2 |x_i - D' alpha_i|^2 + lam 1' alpha_i  =  1/2 sum |x_i|^2  -  tr(D B)  +  1/2 tr(G A)  +  lam 1' sum alpha.
// The first term does not depend on D; pixels without an active stain contribute to none of the others, so the three
// active classes' moments are all it takes -- and A, B are what the first pass of the update needs anyway
// (ab_from_class_moments<true> adds 1' sum alpha).  dict_iter_update holds the iteration to a monotone descent with it.
__device__ __forceinline__ double dict_objective(const double (&D)[2][3], double lam, const double (&A)[2][2], const double (&B)[3][2], double sa) {
    const double g11 = D[0][0] * D[0][0] + D[0][1] * D[0][1] + D[0][2] * D[0][2];
    const double g22 = D[1][0] * D[1][0] + D[1][1] * D[1][1] + D[1][2] * D[1][2];
    const double g12 = D[0][0] * D[1][0] + D[0][1] * D[1][1] + D[0][2] * D[1][2];
    double tdb = 0.0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) tdb = fma(D[j][k], B[k][j], tdb);
    return -tdb + 0.5 * (g11 * A[0][0] + 2.0 * g12 * A[0][1] + g22 * A[1][1]) + lam * sa;
}

#ifdef SL_DEBUG_INNER
__device__ unsigned long long g_dbg_inner[4];     // solves, passes, wall-clock ticks (development aid)
#endif
// one pass of the block-coordinate dictionary update on frozen class moments: D <- g(D)
__device__ __forceinline__ void dict_bcd_update(const double (&A)[2][2], const double (&B)[3][2], double (&D)[2][3]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (A[j][j] > 1e-300) {
            const double ra = 1.0 / A[j][j];
            double u[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                u[k] = (B[k][j] - (D[0][k] * A[0][j] + D[1][k] * A[1][j])) * ra + D[j][k];
                u[k] = fmax(u[k], 0.0);                               // posD
            }
            const double rn = 1.0 / fmax(sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1.0);   // unit ball (modeD=0)
#pragma unroll
            for (int k = 0; k < 3; ++k) D[j][k] = u[k] * rn;
        }
    }
}
__device__ __forceinline__ void dict_bcd_pass(const double* mom, double (&D)[2][3], double lam) {
    double A[2][2], B[3][2];
    ab_from_class_moments(mom, D, lam, A, B);
    dict_bcd_update(A, B, D);
}

// Iterate D <- g(D) until a pass moves D by less than inner_tol (the caller ties it to what the outer iteration still
// needs).  The plain iteration contracts at ~0.7 per pass (~37 passes); depth-1 Anderson mixing
//     D+ = g(D) - gamma (g(D) - g(D_prev)),  gamma = <f, f - f_prev> / |f - f_prev|^2,  f = g(D) - D
// removes the dominant mode (same fixed points: it stops only where g(D) = D).  A mixed step is taken only while the
// residual keeps shrinking and |gamma| is moderate; otherwise the pass is a plain one.  max_it = 1 is exactly one
// plain pass.  Returns the largest change of D over the whole call.
// G_first = g(D) of the incoming D, which the caller has from evaluating the objective there (the first pass is not computed twice).
__device__ __forceinline__ double dict_inner_solve(const double* mom, double (&D)[2][3], double lam, int max_it, double inner_tol, bool mix,
                                                   const double (&G_first)[2][3]) {
#ifdef SL_DEBUG_INNER
    const long long dbg_t0 = wall_clock64();
    int dbg_its = 0;
#endif
    double D0[2][3], gp[2][3], fp[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) { D0[j][k] = D[j][k]; gp[j][k] = 0.0; fp[j][k] = 0.0; }
    double fn_prev = 1e300;
    bool have_prev = false;
    for (int it = 0; it < max_it; ++it) {
        double G[2][3];
        if (it == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) G[j][k] = G_first[j][k];
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) G[j][k] = D[j][k];
            dict_bcd_pass(mom, G, lam);
        }
#ifdef SL_DEBUG_INNER
        ++dbg_its;
#endif
        double f[2][3], step = 0.0, fn = 0.0, fdf = 0.0, dfdf = 0.0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                f[j][k] = G[j][k] - D[j][k];
                step = fmax(step, fabs(f[j][k]));
                fn = fma(f[j][k], f[j][k], fn);
                const double df = f[j][k] - fp[j][k];
                fdf = fma(f[j][k], df, fdf);
                dfdf = fma(df, df, dfdf);
            }
        const bool last = step < inner_tol || it + 1 == max_it;
        double gamma = 0.0;
        if (mix && !last && have_prev && fn < fn_prev && dfdf > 1e-300) {
            gamma = fdf / dfdf;
            if (!(fabs(gamma) <= 20.0)) gamma = 0.0;
        }
        if (gamma != 0.0) {                           // (one lane runs this: a real branch, the plain pass skips the projection)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                double u[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) u[k] = fmax(G[j][k] - gamma * (G[j][k] - gp[j][k]), 0.0);   // the mixed point stays feasible
                const double rn = 1.0 / fmax(sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1.0);
#pragma unroll
                for (int k = 0; k < 3; ++k) D[j][k] = u[k] * rn;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) D[j][k] = G[j][k];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) { gp[j][k] = G[j][k]; fp[j][k] = f[j][k]; }
        fn_prev = fn;
        have_prev = true;
        if (step < inner_tol) break;
    }
#ifdef SL_DEBUG_INNER
    atomicAdd(&g_dbg_inner[0], 1ull); atomicAdd(&g_dbg_inner[1], (unsigned long long)dbg_its);
    atomicAdd(&g_dbg_inner[2], (unsigned long long)(wall_clock64() - dbg_t0));
#endif
    double delta = 0.0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) delta = fmax(delta, fabs(D[j][k] - D0[j][k]));
    return delta;
}

// The state of one tile's dictionary iteration (shared memory in the fused kernel, workspace in the per-phase schedule)
// The first update works on the partition of the Ruifrok start: solved to the end it collapses both atoms onto one
// direction (the sample stage then has to pull them apart again); a few passes keep them apart (measured: 12 -> 10
// solves per tile, 143 -> 108 passes) -- and they are PLAIN passes since late round 3: with the objective in hand
// (dict_iter_update) the mixed first step turned out to raise it on every tile of the bench batch (0.138 -> 0.146: both atoms
// pushed towards each other, the state the two soak failures started from) and to cost the sample stage two more
// iterations than six unmixed passes do (7 iterations / 103 passes -> 5 / 40 on i.i.d. tiles).
constexpr int kDictFirstCap = 6;
struct DictIter {
    double D[6];
    double Dprev[6];
    double delta, delta_prev;   // max-abs change of D by the last update and by the one before it
    double Facc;                // the lowest objective (dict_objective) an accepted iterate of this stage has shown
    int inner_cap;
    int status;
    int cycled;                 // the last update was a cycle break / a rejected step: its delta says nothing about the rate
    int mix;                    // the frozen-partition solves use Anderson mixing (off for the rest of the stage after a rejected mixed step)
    int first_pending;          // the next solve is the first one: kDictFirstCap plain passes
    int rej_cap;                // pass limit imposed by rejected steps; recovers fourfold per solve
    int last_mix, last_cap;     // what the last solve was: mixed or not, its pass limit
    int rejected;               // steps taken back so far (diagnostics)
    int pad_;
};
__device__ __forceinline__ void dict_iter_init(DictIter& it) {
    // deterministic start: Ruifrok's H and E optical-density vectors, unit norm
    const double h[3] = {0.65, 0.70, 0.29}, e[3] = {0.07, 0.99, 0.11};
    const double nh = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]), ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    for (int k = 0; k < 3; ++k) { it.D[k] = h[k] / nh; it.D[3 + k] = e[k] / ne; }
    it.status = SL_TILE_OK;
    it.delta = it.delta_prev = 1.0;
    it.cycled = 0;
    for (int k = 0; k < 6; ++k) it.Dprev[k] = 1e300;
    it.inner_cap = 500;
    it.Facc = 1e300;
    it.mix = 1; it.first_pending = 1; it.rej_cap = 500; it.last_mix = 0; it.last_cap = 0; it.rejected = 0; it.pad_ = 0;
}
// one lane: the dictionary update from the 31 class-moment sums of a sweep (sum[30] = tissue pixels seen).
// stage: 1 sample iteration, 2 full sweep; outer = steps already taken in this stage.
// goal = the change of D below which the caller stops iterating this stage: the frozen-partition solve runs to
// 1e-3 of it (at its ~0.7 linear rate the remaining error is ~2 steps), never below 1e-13.
__device__ __forceinline__ void dict_iter_update(DictIter& it, const double* sum, double lam, int stage, int outer, double goal) {
    if (sum[30] < 1.0) {
        if (stage != 1) it.status = SL_TILE_EMPTY_MASK;      // (an empty SAMPLE only ends the sample stage)
        it.delta_prev = it.delta;
        it.delta = 0.0;
        return;
    }
    double D[2][3];
    for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 3; ++k) D[j][k] = it.D[3 * j + k];
    // Safeguards.  The sums were taken under it.D's own partition, so they give the true objective there and say whether
    // every atom still has pixels that use it.  The target is DEFINED as the point the plain block-coordinate scheme
    // reaches from the Ruifrok start (oracle/stain_oracle.py vahadane_dictionary); the long frozen-partition solves and
    // their mixed steps are an acceleration of it that can leave its path while the partition is still far from final:
    //  - a step that RAISED the objective (beyond the binary32 bursts' noise) -- an over-extrapolated mixed step, seen
    //    on a smooth tile at lambda 0.2: objective +10 %;
    //  - a step that left an atom WITHOUT any pixel: a solve on a partition that no longer holds can shrink an atom
    //    inside the unit ball until no pixel's projection on it exceeds lambda.  A dead atom is never updated again (its
    //    A_jj is 0, in every scheme: a fixed point), and the objective may even have dropped on the way (seen on a
    //    26 x 186 window of real tissue: from 0.567 at the start to 0.170 with one atom dead; the target has 0.163).
    // Either step is taken back: D returns to the iterate before it (the next sweep re-evaluates its sums); a mixed step
    // costs the stage its mixing, an unmixed one three quarters of its passes (the limit recovers fourfold per solve).
    // At one unmixed pass the scheme IS the plain one and its steps stand, whatever they do.
    double A0[2][2], B0[3][2], sa0;
    ab_from_class_moments<true>(sum, D, lam, A0, B0, &sa0);
    const double F = dict_objective(D, lam, A0, B0, sa0);
    const bool dead = sum[0] + sum[10] <= 0.0 || sum[0] + sum[20] <= 0.0;
    const bool plain = !it.last_mix && it.last_cap <= 1;
    if ((dead || !(F <= it.Facc + 1e-6 * sum[30])) && !plain && it.Dprev[0] < 1e299) {      // (a NaN objective is a rejection too)
        for (int k = 0; k < 6; ++k) { it.D[k] = it.Dprev[k]; it.Dprev[k] = 1e300; }
        if (it.last_mix) it.mix = 0;
        else it.rej_cap = it.last_cap > 4 ? it.last_cap / 4 : 1;
        it.delta = it.delta_prev = 1.0;
        it.cycled = 1;
        ++it.rejected;
        return;
    }
    it.Facc = fmin(it.Facc, F);
    int cap = it.inner_cap < it.rej_cap ? it.inner_cap : it.rej_cap;
    if (it.first_pending && cap > kDictFirstCap) cap = kDictFirstCap;
    const bool mix = it.mix && !it.first_pending;
    it.last_mix = mix ? 1 : 0; it.last_cap = cap; it.first_pending = 0;
    if (it.rej_cap < 500) it.rej_cap = it.rej_cap * 4 < 500 ? it.rej_cap * 4 : 500;
    double G1[2][3];
    for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 3; ++k) G1[j][k] = D[j][k];
    dict_bcd_update(A0, B0, G1);                                   // the first pass, from the A and B the objective was read off
    const double delta = dict_inner_solve(sum, D, lam, cap, fmax(1e-3 * goal, 1e-13), mix, G1);
    // The frozen-partition solve is a Newton-like step on a piecewise-smooth map and can fall
    // into a 2-cycle between two partitions: the new iterate then returns to the one before
    // last.  In that case restart from the midpoint and shorten the inner solve; at one inner
    // iteration the scheme IS plain block-coordinate descent (monotone).
    double back = 0.0;
    for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 3; ++k) back = fmax(back, fabs(D[j][k] - it.Dprev[3 * j + k]));
    const bool cycling = outer >= 2 && back < 0.25 * delta && it.inner_cap > 1;
    if (cycling) it.inner_cap = it.inner_cap > 4 ? it.inner_cap / 4 : 1;
    for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 3; ++k) {
            const double cur = it.D[3 * j + k];
            it.Dprev[3 * j + k] = cur;
            it.D[3 * j + k] = cycling ? 0.5 * (D[j][k] + cur) : D[j][k];
        }
    it.delta_prev = it.delta;
    it.delta = delta;
    it.cycled = cycling ? 1 : 0;
}
// the sample stage is over: the full sweeps restart the cycle detector
__device__ __forceinline__ void dict_iter_restart(DictIter& it) {
    // (Dprev stays: the cycle test waits for two steps of the new stage, and a first full sweep that finds an atom dead can
    // still step back)
    it.inner_cap = 500;
    it.delta = it.delta_prev = 1.0;
    it.cycled = 0;
    it.Facc = 1e300;            // (another pixel set: the sample's objective says nothing about the tile's)
    it.mix = 1; it.first_pending = 0; it.rej_cap = 500; it.last_mix = 1; it.last_cap = 500;
}
// H first: swap when D[0,0] < D[1,0] (vahadane_stain_extractor.py:40-41), unit-norm rows (:43)
__device__ __forceinline__ void dict_iter_stain_matrix(const DictIter& it, double* M) {
    const bool swap = it.D[0] < it.D[3];
    double h[3], e[3];
    for (int k = 0; k < 3; ++k) { h[k] = swap ? it.D[3 + k] : it.D[k]; e[k] = swap ? it.D[k] : it.D[3 + k]; }
    const double nh = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]), ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    for (int k = 0; k < 3; ++k) { M[k] = h[k] / nh; M[3 + k] = e[k] / ne; }
}

struct DictProgress { int stage, outer, sample_its, sweeps_used; };   // workgroup-uniform
constexpr double kDictRateSafety = 4.0;    // the predicted contraction of the next full sweep is this times the quadratic rule's
constexpr double kDictRhoFast = 0.1;       // the a-posteriori stop needs delta_k / delta_(k-1) below this (measured ratios of full sweeps: 1e-2 ... 1e-4)
constexpr double kDictSampleTol = 1e-4;   // the sample stage ends when an update moves D by less than this (the sample itself is only good to ~1e-3: tighter buys no full sweep)

// workgroup-uniform bookkeeping after an update; returns false when the iteration is over (it.status / it.delta are
// read by every thread: call between barriers).  Ends with a barrier when the stage changes.
__device__ __forceinline__ bool dict_advance(DictIter& it, DictProgress& pr, double tol, int tid) {
    ++pr.outer;
    if (pr.stage != 1) ++pr.sweeps_used;
    if (it.status != SL_TILE_OK) return false;
    if (pr.stage == 1) {
        ++pr.sample_its;
        if (it.delta < kDictSampleTol || pr.sample_its >= 40) {                // sample fixed point reached: on to the tile
            pr.stage = 2; pr.outer = 0;
            __syncthreads();
            if (tid == 0) dict_iter_restart(it);
            __syncthreads();
        }
    } else {
        if (it.delta < tol) return false;
        // A-posteriori stop.  With rho = delta_k / delta_(k-1), the step the NEXT sweep would take -- the distance of D to
        // the fixed point -- is delta_k * rho / (1 - rho) for a linearly convergent iteration.  This one is Newton-like
        // (the frozen-partition solve is exact for its partition; measured error per full sweep on 1024^2 tiles:
        // 2e-3 -> 3e-5 -> 1e-8 -> 5e-14, i.e. the next ratio is about rho^2: 0.2-0.5 rho^2 on 256 tiles), so the next
        // ratio is taken as kDictRateSafety * rho^2, never better than rho itself.  When the estimate is below tol the
        // next sweep would only confirm it: stop.  Guards: two full sweeps taken, no cycle break among them.
        // tests/test_gpu_vahadane.py::test_vahadane_error_stays_within_the_tolerance holds the rule to its promise
        // against the converged oracle (measured: error <= 0.7 tol down to tol = 1e-8).
        // The rule presumes the Newton-like regime: it is applied only while the contraction is fast (rho < kDictRhoFast).  A tile
        // that converges merely linearly (a near-degenerate partition) keeps iterating until the step itself is below tol.
        const double rho = it.delta / it.delta_prev;
        if (pr.outer >= 2 && !it.cycled && rho < kDictRhoFast) {
            const double next_rate = fmin(rho, kDictRateSafety * rho * rho);
            if (it.delta * next_rate / (1.0 - next_rate) < tol) return false;
        }
    }
    return true;
}

// One workgroup iterates a tile's dictionary from (it, pr) until it settles.  Schedule (pr starts at stage 1 with the
// sample gathered: gather_sample): the fixed-point iteration on the 16 Ki-pixel sample until it settles (each step costs
// 1/64 of a sweep), then full sweeps from that warm start until the dictionary moves by less than tol: ~3 full sweeps
// instead of ~9 from the cold start.  (Round 1 spent one more full sweep up front, whose only lasting product was the
// sample.)  red/sum are workgroup scratch.  Ends with a barrier.  SAMPLE_ONLY: only the sample stage (the per-phase
// schedule runs the full sweeps as launches of their own).
template <bool ALIGNED, int NT, bool SAMPLE_ONLY = false, bool STREAM = false>
__device__ __forceinline__ void dict_learn(const uint8_t* src, int P, int nch, int tid, const TabReaderB& T, float ylimf,
                                           int stride_log2, uint32_t* samp, int n_sample, double lam, double tol, int max_sweeps,
                                           DictIter& it, double (*red)[32], double* sum, DictProgress& pr) {
    static_assert(SAMPLE_ONLY || NT == kSweepThreads, "full sweeps run with the 512-thread trip geometry of the sweep kernels");
    const int lane = tid & 63, wave = tid >> 6;
    while (pr.sweeps_used < max_sweeps && (!SAMPLE_ONLY || pr.stage == 1)) {
        DictK Ld;
        dict_consts(it.D, lam, Ld);
        __syncthreads();                                             // previous iteration's readers of red are done
        DictWaveAcc acc;
        acc.begin(red[wave], lane);
        if (SAMPLE_ONLY || pr.stage == 1)
            dict_sweep_sample_b(samp, n_sample, stride_log2, P, tid, T, ylimf, Ld, acc);
        else
            dict_sweep_b<ALIGNED, kDictTrip, STREAM>(src, P, 0, nch, tid, NT, T, ylimf, Ld, acc);
        __syncthreads();
        if (tid < 31) {
            double t = 0;
            for (int w = 0; w < NT / 64; ++w) t += red[w][tid];
            sum[tid] = t;
        }
        __syncthreads();
        if (tid == 0) dict_iter_update(it, sum, lam, pr.stage, pr.outer, pr.stage == 1 ? kDictSampleTol : tol);
        __syncthreads();
        if (!dict_advance(it, pr, tol, tid)) break;
    }
}

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

// brackets of both angular quantiles from the sample (THREADS = blockDim.x)
// box (optional, float[4] = {lo0, hi0, lo1, hi1}): where the merged sweep may assume the two percentile angles to fall.  When
// both 6-sigma brackets are closed it is their central kBoxFrac; when one is open (a small tissue sample: the rank minus 6 sigma
// leaves it) a second pair at kBoxZ sigma is located in the same register-resident keys -- costs a histogram pass only then.
constexpr float kBoxZ = 3.6f;
template <int THREADS>
__device__ __forceinline__ void angle_brackets(const SampleAngleKey& key, int n_sample, double pct, float* lo, float* hi,
                                               SelScratch& S, float* box = nullptr) {
    constexpr int KPT = kMaxSample / THREADS;
    uint32_t ord[1][KPT];
#ifdef SL_DEBUG_SUBCLK
    long long bclk_t_ = wall_clock64();
#endif
    // the sample words are loaded kBrkBatch at a time so that their latencies overlap (all 32 at once measured 1-2 % SLOWER end
    // to end: the extra live registers shift the allocator's spills into the sweep prologues; 8 gains 1.5 %)
    static_assert(KPT % kBrkBatch == 0, "");
#pragma unroll
    for (int j0 = 0; j0 < KPT; j0 += kBrkBatch) {
        uint32_t w[kBrkBatch];
#pragma unroll
        for (int g = 0; g < kBrkBatch; ++g) {
            const int b = (j0 + g) * THREADS + (int)threadIdx.x;
            w[g] = b < n_sample ? key.sample[b] : 0u;
        }
#pragma unroll
        for (int g = 0; g < kBrkBatch; ++g) {
            const int b = (j0 + g) * THREADS + (int)threadIdx.x;
            const float k = key.of_word(w[g]);
            ord[0][j0 + g] = (k == k && key.present(b, n_sample)) ? f2ord(k) : kAbsent;
        }
    }
    SL_BCLK(5);
    const int set_of[2] = {0, 0};
    const double p2[2] = {100.0 - pct, pct};          // minPhi, maxPhi (macenko_stain_extractor.py:33-34)
    wg_brackets_regs<1, KPT, 2>(ord, set_of, p2, lo, hi, S);
    if (box) {
        const bool closed = (lo[0] > -INFINITY) & (hi[0] < INFINITY) & (lo[1] > -INFINITY) & (hi[1] < INFINITY);     // block-uniform
        if (closed) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float m = 0.5f * (lo[b] + hi[b]), r = (float)kBoxFrac * 0.5f * (hi[b] - lo[b]);
                box[2 * b] = m - r; box[2 * b + 1] = m + r;
            }
        } else {
            // worth a second pass only if the kBoxZ-sigma ranks stay inside the sample (S.misc[6]: its valid keys, left by the first pass)
            const double n = (double)S.misc[6], q = p2[0] / 100.0;
            const bool inside = n > 0.0 && floor(q * (n - 1.0) - (double)kBoxZ * sqrt(fmax(q * (1.0 - q) * n, 0.0))) - 1.0 >= 0.0;   // block-uniform
            float blo[2] = {-INFINITY, -INFINITY}, bhi[2] = {INFINITY, INFINITY};
            __syncthreads();
            if (inside) wg_brackets_regs<1, KPT, 2>(ord, set_of, p2, blo, bhi, S, kBoxZ);
            box[0] = blo[0]; box[1] = bhi[0]; box[2] = blo[1]; box[3] = bhi[1];
        }
    }
}
// brackets of the 99th percentile of both concentration columns from the sample (normalizer.py:36,47)
template <int THREADS>
__device__ __forceinline__ void conc_brackets(const SampleConcKey& key, int n_sample, float* lo, float* hi, SelScratch& S) {
    constexpr int KPT = kMaxSample / THREADS;
    uint32_t ord[2][KPT];
#ifdef SL_DEBUG_SUBCLK
    long long bclk_t_ = wall_clock64();
#endif
    static_assert(KPT % kBrkBatch == 0, "");
#pragma unroll
    for (int j0 = 0; j0 < KPT; j0 += kBrkBatch) {
        uint32_t w[kBrkBatch];
#pragma unroll
        for (int g = 0; g < kBrkBatch; ++g) {
            const int b = (j0 + g) * THREADS + (int)threadIdx.x;
            w[g] = b < n_sample ? key.sample[b] : 0u;
        }
#pragma unroll
        for (int g = 0; g < kBrkBatch; ++g) {
            const int b = (j0 + g) * THREADS + (int)threadIdx.x;
            float c1, c2;
            key.of_word(w[g], c1, c2);
            const bool have = key.present(b, n_sample);
            ord[0][j0 + g] = (have && c1 == c1) ? f2ord(c1) : kAbsent;
            ord[1][j0 + g] = (have && c2 == c2) ? f2ord(c2) : kAbsent;
        }
    }
    SL_BCLK(6);
    const int set_of[2] = {0, 1};
    const double p2[2] = {99.0, 99.0};
    wg_brackets_regs<2, KPT, 2>(ord, set_of, p2, lo, hi, S);
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
#ifdef SL_EXP_DICT_DIAG      // development: sample iterations and rejected steps ride in the sweep count (x 100, x 10000)
        if (a.sweeps_out) a.sweeps_out[a.tile0 + tile] = ds.pr.sweeps_used + 100 * ds.pr.sample_its + 10000 * ds.it.rejected;
#else
        if (a.sweeps_out) a.sweeps_out[a.tile0 + tile] = ds.pr.sweeps_used;
#endif
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
};

template <int NT>
struct FusedShared {
    RowTab tab;              // 64 KB, first member: LDS offset 0
    uint32_t stage[NT / 64][kStageWave];     // 1 KB per wave
    unsigned int n_raw, overflow;
    SelScratch S;
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
    MergedConc mk;
};

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

// Finish 1 of the merged schedule after the eigenvectors: brackets and thresholds into sh.lo / sh.hi / sh.mk.
template <int NT>
__device__ __noinline__ void fused_finish1(FusedShared<NT>* shp, uint32_t* samp_, int n_sample_, int stride_log2_, int P_, float ylimf_, double pct_, double lam_,
                                           long long* subclk_) {
    FusedShared<NT>& sh = *shp;
    uint32_t* samp = uni_ptr(samp_);
    const int n_sample = __builtin_amdgcn_readfirstlane(n_sample_), stride_log2 = __builtin_amdgcn_readfirstlane(stride_log2_), P = __builtin_amdgcn_readfirstlane(P_);
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
        SampleAngleKey key;
        key.sample = samp; key.tab = view_of_b(sh.tab); key.cps_log2 = stride_log2 - 2; key.P = P; key.ylimf = ylimf;
        for (int i = 0; i < 6; ++i) key.V[i] = sh.Vf[i];
        float lo[2], hi[2];
        float box[4];
        angle_brackets<NT>(key, n_sample, pct, lo, hi, sh.S, box);
        if (tid == 0) {
            sh.lo[0] = lo[0]; sh.hi[0] = hi[0]; sh.lo[1] = lo[1]; sh.hi[1] = hi[1];
            for (int i = 0; i < 4; ++i) sh.box[i] = box[i];
            sh.xmin = tissue_x_bound(sh.Vf, ylimf, view_of_b(sh.tab));
            sh.S.misc[32] = 0;
        }
        __syncthreads();
        // The projection bound makes the sweep collect NON-tissue pixels too when they pass it and lie outside the cone.  On most
        // tiles those are few; a uniform bright-but-not-white background (say 245, 245, 245: not tissue, first projection above the
        // bound, direction outside the stains' cone) would put most of the tile on the candidate list and cost it the exact
        // fallback (measured: 21 ms per 512 such tiles).  The sample says beforehand: if the pixels the bound would add exceed
        // P/40, this tile's sweep keeps the per-pixel tissue test.
        const float xm = sh.xmin;
        if (xm > -INFINITY && xm < INFINITY) {                    // block-uniform
            const float hi0 = sh.hi[0], lo1 = sh.lo[1];
            uint32_t extra = 0;
            for (int b = tid; b < n_sample; b += NT) {
                if (!key.present(b, n_sample)) continue;
                const uint32_t w = samp[b];
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
    }
    SL_SUB(12);
    // ---------------- the box of stain matrices the sample leaves possible, concentration brackets under its centre
    if (tid < 64) merged_box(sh.Vd, sh.box, lam, tid, sh.mk);
    __syncthreads();
    SL_SUB(13);
    if (sh.mk.ok) {                                               // block-uniform
        SampleConcKey ckey;
        ckey.sample = samp; ckey.tab = view_of_b(sh.tab); ckey.L = sh.mk.Lc; ckey.cps_log2 = stride_log2 - 2;
        ckey.P = P; ckey.col = 0;
        float lo[2], hi[2];
        conc_brackets<NT>(ckey, n_sample, lo, hi, sh.S);
        if (tid == 0) merged_thresholds(sh.mk, lo[0], lo[1], hi[0], hi[1]);
    } else if (tid == 0) {
        merged_thresholds(sh.mk, -INFINITY, -INFINITY, -INFINITY, -INFINITY);       // disarms the concentration test
    }
    __syncthreads();
#undef SL_SUB
}

// Finish 2 of the merged schedule for the tile whose state sits in *shp: returns the number of slow exact fallbacks.
// Leaves sh.M, sh.status, sh.conc_done (and sh.maxC, sh.L when conc_done) behind and the row table rebuilt.
template <int NT>
__device__ __noinline__ int fused_finish2(FusedShared<NT>* shp, const uint8_t* src_, uint32_t* rawl_, float* cand0_, float* cand1_, int P_, int cap_raw_,
                                          int cap_list_, float ylimf_, double pct_, double lam_, long long* subclk_) {
    FusedShared<NT>& sh = *shp;
    const uint8_t* src = uni_ptr(src_);
    uint32_t* rawl = uni_ptr(rawl_);
    float* cand0 = uni_ptr(cand0_);
    float* cand1 = uni_ptr(cand1_);
    const int P = __builtin_amdgcn_readfirstlane(P_), cap_raw = __builtin_amdgcn_readfirstlane(cap_raw_), cap_list = __builtin_amdgcn_readfirstlane(cap_list_);
    const float ylimf = uni(ylimf_);
    const double pct = uni_d(pct_), lam = uni_d(lam_);
    long long* subclk = uni_ptr(subclk_);
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
    const FinTab FT{lds_address(&sh.tab)};
    const uint32_t stage_lds = lds_address(&sh.tab) + kFinTabBytes + (uint32_t)wave * fin_stage_bytes(NT);
    const uint32_t stage_entries = fin_stage_bytes(NT) / 8u;      // two lists per wave
    AngleTileKey tkey;
    tkey.src = src; tkey.tab = FT.view(); tkey.ylimf = ylimf;
    WordAngleKey rkey;
    rkey.T = FT; rkey.ylimf = ylimf;
    for (int i = 0; i < 6; ++i) { tkey.V[i] = sh.Vf[i]; rkey.V[i] = sh.Vf[i]; }
    const bool complete = sh.n_raw <= (uint32_t)cap_raw && sh.overflow == 0;
    const uint32_t n_raw = sh.n_raw < (uint32_t)cap_raw ? sh.n_raw : (uint32_t)cap_raw;
    const float los[2] = {sh.lo[0], sh.lo[1]}, his[2] = {sh.hi[0], sh.hi[1]};
    SL_SUB(2);
    const RefineOut ra = wg_refine_s(rawl, (int)n_raw, rkey, los[0], his[0], los[1], his[1], cand0, cand1, (uint32_t)cap_list, stage_lds,
                                     stage_entries, sh.S);
    SL_SUB(3);
    {
        // plain = tissue pixels the sweep did not collect: they sit between the two brackets (the list also holds
        // pixels collected for their concentrations; those with an angle key count like any other candidate)
        const long long lt[2] = {(long long)ra.n_lt[0], (long long)T - (long long)ra.n_valid + (long long)ra.n_lt[1]};
        float res[4];
        stage_pick2<false>(cand0, cand1, ra.n_in, (uint32_t)cap_list, complete, los, his, lt, P, tkey, T, k, ra.ps, res, fallbacks, sh.S);
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
            q[8] = 100ll * ra.n_valid; q[9] = 100ll * n_raw; q[10] = 100ll * (ra.n_in[0] + ra.n_in[1]); q[11] = 100ll * (rc.n_in[0] + rc.n_in[1]);
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

// ---- the merged schedule, one launch per phase: k_moments -> k_finish1m -> k_select<merged> -> k_finish2m [-> k_select<conc>,
// k_finish_conc for the rare tile whose exact stain matrix left the assumed box] -> k_apply.  The finish kernels ARE the fused
// kernel's finish steps (fused_finish1 / fused_finish2) on a FusedShared block of their own, with the tile's state carried in
// TileState / TileMerged between the launches: both schedules select the same values by construction.
constexpr int kMFinishThreads = 1024;
static __global__ __launch_bounds__(kMFinishThreads) void k_finish1m(StatsArgs a) {
    __shared__ FusedShared<kMFinishThreads> sh;
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
        fused_finish1<kMFinishThreads>(&sh, a.sample + (size_t)tile * a.n_sample, a.n_sample, a.stride_log2, a.P, a.ylimf, a.pct, a.lam, nullptr);
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
        fallbacks = fused_finish2<kMFinishThreads>(&sh, a.rgb + (size_t)tile * a.P * 3, a.raw + (size_t)tile * a.cap_raw,
                                                   a.cand + ((size_t)tile * 2 + 0) * a.cap_list, a.cand + ((size_t)tile * 2 + 1) * a.cap_list, a.P,
                                                   a.cap_raw, a.cap_list, a.ylimf, a.pct, a.lam, nullptr);
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

enum { kMethodMacenko = 0, kMethodVahadane = 1 };

// NT = 512: two workgroups per CU (the throughput configuration).  NT = 1024: one workgroup per CU, used when the batch
// has no more tiles than CUs -- the tile's latency halves (Vahadane below 257 tiles; Macenko runs per phase there).
template <int METHOD, bool TRANSFORM, bool ALIGNED, int NT>
static __global__ __launch_bounds__(NT, 4) void k_fused(FusedArgs a) {
    __shared__ FusedShared<NT> sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TabReaderB TB = TabReaderB::make(sh.tab);       // the 8-byte {gamma, od32} rows serve every sweep
    const int nch = (a.P + 3) >> 2;
    uint32_t* samp = a.sample + (size_t)blockIdx.x * a.n_sample;
    uint32_t* rawl = a.raw + (size_t)blockIdx.x * a.cap_raw;
    float* cand0 = a.cand + ((size_t)blockIdx.x * 2 + 0) * a.cap_list;
    float* cand1 = a.cand + ((size_t)blockIdx.x * 2 + 1) * a.cap_list;

    // sweeps 2/3 share this: classify against sh.lo/hi with constants K; plain count and raw candidates into sh.*
    const bool stream = (size_t)a.P * 3 >= kStreamBytes;       // non-temporal tile accesses (uniform; see kStreamBytes)
    auto run_select = [&](auto stage_tag, const uint8_t* src, SelConsts& K) {
        constexpr int STAGE = decltype(stage_tag)::value;
        K.lo0 = uni(sh.lo[0]); K.hi0 = uni(sh.hi[0]); K.lo1 = uni(sh.lo[1]); K.hi1 = uni(sh.hi[1]);
        RawSink sink{(uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(sh.stage[wave])), 0u, rawl, &sh.n_raw, &sh.overflow, (uint32_t)a.cap_raw,
                     (uint32_t)kStageWave};
        if (STAGE == kStageMerged && K.xmin > -INFINITY) {       // block-uniform: the projection bound stands in for the tissue test
            if (stream) select_sweep<kStageMerged, ALIGNED, kFusedTrip, true, true>(src, a.P, 0, nch, tid, NT, TB, a.ylimf, K, sink);
            else select_sweep<kStageMerged, ALIGNED, kFusedTrip, false, true>(src, a.P, 0, nch, tid, NT, TB, a.ylimf, K, sink);
        } else {
            if (stream) select_sweep<STAGE, ALIGNED, kFusedTrip, true>(src, a.P, 0, nch, tid, NT, TB, a.ylimf, K, sink);
            else select_sweep<STAGE, ALIGNED, kFusedTrip, false>(src, a.P, 0, nch, tid, NT, TB, a.ylimf, K, sink);
        }
        sink.flush(lane);
        __threadfence_block();
        __syncthreads();
    };

    // ---- sharing the CU.  Two workgroups live on a CU and the instruction arbiter serves the OLDER one's waves first: left alone,
    // the first-launched workgroup of every CU runs its sweeps ~20 % faster than its partner, whose latency-bound finish steps
    // stretch by half (measured: tile latency 1.50 vs 1.83 ms; the launch ends when the slow half does, with the CU
    // half empty for the last 0.3 ms).  s_setprio overrides age: the finish steps (few instructions, long dependent
    // latencies) always run at top priority, and the sweeps' priorities alternate between the two workgroups by sweep.
    uint32_t lds_alloc_;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds_alloc_));
    const bool younger = (lds_alloc_ & 0xffu) != 0u;          // the workgroup that got the upper half of the CU's LDS was launched second
#ifndef SL_PRIO_SCHEME
#define SL_PRIO_SCHEME 0
#endif
    auto prio_finish = [&]() { if (SL_PRIO_SCHEME >= 1) __builtin_amdgcn_s_setprio(3); };
    auto prio_sweep = [&](int which) {              // which: 0 moments, 1 select, 2 conc resweep / dictionary, 3 apply
        if (SL_PRIO_SCHEME == 1) __builtin_amdgcn_s_setprio(0);
        if (SL_PRIO_SCHEME == 2) {
            if (younger) __builtin_amdgcn_s_setprio(1);
            else if (which & 1) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
        if (SL_PRIO_SCHEME == 3) {
            if (younger) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
    };
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const size_t nbytes = (size_t)a.P * 3;
#ifdef SL_DEBUG_SAMETILE
        const uint8_t* src = a.rgb + (size_t)(tile & SL_DEBUG_SAMETILE) * nbytes;   // development aid: cache-resident input (0: one tile, 7: eight)
#else
        const uint8_t* src = a.rgb + (size_t)tile * nbytes;
#endif
        int fallbacks = 0;
        int sweeps_used = 0;
#ifdef SL_DEVTOOLS
// (Macenko only: in k_fused<vahadane, transform, unaligned> the extra `continue` edges run into the hipcc bug described in the Makefile)
#define SL_PHASE(i) { if (METHOD == kMethodMacenko) { if (a.phase_clock && tid == 0) a.phase_clock[(size_t)tile * 8 + (i)] = wall_clock64(); if (a.debug_stop == (i) + 1) { __syncthreads(); continue; } } }
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

        if (METHOD == kMethodMacenko) {
            // ---------------- sweep 1: moments + sample
            prio_sweep(0);
            {
                Moments mo;
                uint32_t n_tissue = 0;
                if (stream) moments_sweep_b<ALIGNED, kFusedTrip, true>(src, a.P, 0, nch, tid, NT, TB, a.ylimf, a.stride_log2, samp, mo, n_tissue);
                else moments_sweep_b<ALIGNED, kFusedTrip, false>(src, a.P, 0, nch, tid, NT, TB, a.ylimf, a.stride_log2, samp, mo, n_tissue);
                double v[10];
                mo.to_array(v, n_tissue, lane);
#pragma unroll
                for (int i = 0; i < 10; ++i) v[i] = wave_sum(v[i]);
                if (lane == 0)
                    for (int i = 0; i < 10; ++i) sh.red[wave][i] = v[i];
            }
            prio_finish();
            __syncthreads();
            if (tid < 10) {
                double t = 0;
                for (int w = 0; w < NT / 64; ++w) t += sh.red[w][tid];
                sh.sum[tid] = t;
            }
            __syncthreads();
            SL_PHASE(1);
            // ---------------- finish 1: eigenvectors, angle brackets
            SL_SUB(0);
            if (tid == 0) {
                double Vd[6];
                float Vf[6];
                sh.status = eigvecs_from_moments(sh.sum, Vd, Vf);
                for (int i = 0; i < 6; ++i) { sh.Vd[i] = Vd[i]; sh.Vf[i] = Vf[i]; }
                sh.n_raw = 0; sh.overflow = 0;
                sh.conc_done = 0;
            }
            __syncthreads();
            SL_SUB(1);
            if (sh.status == SL_TILE_OK) {                                    // block-uniform
                // ---------------- finish 1, the rest of it (out of line like finish 2): angle brackets, the box of stain matrices the
                // sample leaves possible, concentration brackets under its centre
                fused_finish1<NT>(&sh, samp, a.n_sample, a.stride_log2, a.P, a.ylimf, a.pct, a.lam,
#ifdef SL_DEBUG_SUBCLK
                                  a.phase_clock ? a.phase_clock + (size_t)a.n_tiles * 8 + (size_t)tile * 16 : nullptr
#else
                                  nullptr
#endif
                                  );
                SL_PHASE(2);
                // ---------------- sweep 2: angle select + concentration select under the box
                {
                    SelConsts K;
                    for (int i = 0; i < 6; ++i) K.V[i] = in_vgpr(sh.Vf[i]);
                    K.L.g12 = 0.0f;
                    K.xmin = uni(sh.xmin);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        K.u[i][0] = in_vgpr(sh.mk.u[i][0]); K.u[i][1] = in_vgpr(sh.mk.u[i][1]); K.kt[i] = in_vgpr(sh.mk.kt[i]);
                        K.eps[i] = in_vgpr(sh.mk.eps[i]); K.thr[i] = in_vgpr(sh.mk.thr[i]);
                    }
                    prio_sweep(1);
                    run_select(std::integral_constant<int, kStageMerged>{}, src, K);
                    prio_finish();
                }
                SL_PHASE(3);
                // ---------------- finish 2 (out of line: its registers are allocated apart from the sweeps'): exact angular percentiles
                // -> M, then the concentration percentiles -> maxC from the same raw list
                fallbacks += fused_finish2<NT>(&sh, src, rawl, cand0, cand1, a.P, a.cap_raw, a.cap_list, a.ylimf, a.pct, a.lam,
#ifdef SL_DEBUG_SUBCLK
                                               a.phase_clock ? a.phase_clock + (size_t)a.n_tiles * 8 + (size_t)tile * 16 : nullptr
#else
                                               nullptr
#endif
                                               );
            }
        } else {
            // ---------------- Vahadane: class-moment dictionary learning
            prio_sweep(0);
            gather_sample<ALIGNED>(src, a.P, a.stride_log2, samp, a.n_sample, tid, NT);
            if (tid == 0) {
                dict_iter_init(sh.it);
                sh.n_raw = 0; sh.overflow = 0;
                sh.conc_done = 0;
            }
            __syncthreads();
            DictProgress pr{1, 0, 0, 0};
            if (stream) dict_learn<ALIGNED, NT, false, true>(src, a.P, nch, tid, TB, a.ylimf, a.stride_log2, samp, a.n_sample, a.dl_lambda,
                                                             a.dl_tol, a.dl_max_sweeps, sh.it, sh.red, sh.sum, pr);
            else dict_learn<ALIGNED, NT, false, false>(src, a.P, nch, tid, TB, a.ylimf, a.stride_log2, samp, a.n_sample, a.dl_lambda,
                                                       a.dl_tol, a.dl_max_sweeps, sh.it, sh.red, sh.sum, pr);
            sweeps_used = pr.sweeps_used;
            if (tid == 0) {
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
            if (tid == 0) {
                LassoK L;
                lasso_consts(sh.M, a.lam, L);
                sh.L = L;
                sh.n_raw = 0; sh.overflow = 0;
            }
            __syncthreads();
            SL_SUB(7);
            // ---------------- concentration brackets from the sample
            {
                SampleConcKey ckey;
                ckey.sample = samp; ckey.tab = view_of_b(sh.tab); ckey.L = sh.L; ckey.cps_log2 = a.stride_log2 - 2;
                ckey.P = a.P; ckey.col = 0;
                float lo[2], hi[2];
                conc_brackets<NT>(ckey, a.n_sample, lo, hi, sh.S);
                if (tid == 0) { sh.lo[0] = lo[0]; sh.hi[0] = hi[0]; sh.lo[1] = lo[1]; sh.hi[1] = hi[1]; }
                __syncthreads();
            }
            SL_PHASE(4);
            // ---------------- sweep 3: concentration select
            {
                SelConsts K;
                K.L = sh.L;
                K.xmin = -INFINITY;
                vgpr(K.L);
                prio_sweep(2);
                run_select(std::integral_constant<int, kStageConc>{}, src, K);
                prio_finish();
            }
            SL_PHASE(5);
            // ---------------- finish 3: exact 99th percentiles -> maxC
            {
                long long k;
                double gfrac;
                percentile_pos((double)a.P, 99.0, k, gfrac);
                ConcTileKey tkey;
                tkey.src = src; tkey.tab = view_of_b(sh.tab); tkey.L = sh.L;
                RawConcKey2 rkey;
                rkey.raw = rawl; rkey.tab = view_of_b(sh.tab); rkey.L = sh.L;
                const bool complete = sh.n_raw <= (uint32_t)a.cap_raw && sh.overflow == 0;
                const uint32_t n_raw = sh.n_raw < (uint32_t)a.cap_raw ? sh.n_raw : (uint32_t)a.cap_raw;
                const float los[2] = {sh.lo[0], sh.lo[1]}, his[2] = {sh.hi[0], sh.hi[1]};
                const long long n_plain = (long long)a.P - (long long)sh.n_raw;        // plain = pixels not collected
                uint32_t n_lt[2], n_in[2];
                SL_SUB(8);
                wg_refine((int)n_raw, rkey, los, his, cand0, cand1, (uint32_t)a.cap_list, n_lt, n_in, sh.S);
                SL_SUB(9);
                for (int col = 0; col < 2; ++col) {
                    tkey.col = col;
                    float xa, xb;
                    stage_order_stats(col ? cand1 : cand0, n_in[col], (uint32_t)a.cap_list, complete, los[col], his[col], n_plain + n_lt[col], a.P,
                                      tkey, (uint32_t)a.P, k, xa, xb, fallbacks, sh.S);
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
        } else if (bad && sh.status != SL_TILE_ZERO_MAXC && tid == 0) {     // (a zero maxC keeps its M and maxC, as after finish 3)
            for (int i = 0; i < 6; ++i) sh.M[i] = nan_d();
            sh.maxC[0] = sh.maxC[1] = nan_d();
        }
        __syncthreads();
        if (tid == 0 && a.resweep_out) a.resweep_out[tile] = (METHOD == kMethodMacenko && resweep) ? (sh.why ? sh.why : SL_RESWEEP_NO_BOX) : 0;
        if (tid < 6 && a.M_out) a.M_out[(size_t)tile * 6 + tid] = sh.M[tid];
        if (tid < 2 && a.maxC_out) a.maxC_out[(size_t)tile * 2 + tid] = sh.maxC[tid];
        if (tid == 0 && a.status_out) a.status_out[tile] = sh.status;
        if (tid == 0 && a.diag_out) a.diag_out[tile] = fallbacks;
        if (tid == 0 && a.sweeps_out) a.sweeps_out[tile] = sweeps_used;
        SL_PHASE(6);
        // ---------------- sweep 4: apply
        if (TRANSFORM) {
            uint8_t* dst = a.out + (size_t)tile * nbytes;
            if (sh.status != SL_TILE_OK) {       // block-uniform (sh.status is final: barrier above); includes a zero maxC found in finish 3
                for (int c = tid; c < nch; c += NT) store_chunk<ALIGNED>(dst, nbytes, c, load_chunk<ALIGNED>(src, nbytes, c));
            } else {
                ApplyK K;
                apply_consts(sh.M, sh.maxC, a.M_tgt, a.maxC_tgt, a.lam, K);
                prio_sweep(3);
                if (stream) {
                    if (K.fast) apply_sweep<ALIGNED, true, TabReaderB, true>(src, dst, a.P, 0, nch, tid, NT, TB, K);
                    else apply_sweep<ALIGNED, false, TabReaderB, true>(src, dst, a.P, 0, nch, tid, NT, TB, K);
                } else {
                    if (K.fast) apply_sweep<ALIGNED, true, TabReaderB, false>(src, dst, a.P, 0, nch, tid, NT, TB, K);
                    else apply_sweep<ALIGNED, false, TabReaderB, false>(src, dst, a.P, 0, nch, tid, NT, TB, K);
                }
            }
        }
        __syncthreads();     // sh.* is reused by the next tile
        SL_PHASE(7);
#undef SL_PHASE
#undef SL_SUB
    }
}

}  // namespace sl
