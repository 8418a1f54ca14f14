// stats_selection.hpp -- workgroup-level exact selection primitives (radix select, register-resident brackets, small-list select).
// Part of stats_kernels.hpp (split by phase in round 4, no functional change); include that umbrella, not this file.
#pragma once
#include "stats_common.hpp"

namespace sl {

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
__device__ __forceinline__ void wave_locate(const uint32_t* hist, int nb, uint32_t k, uint32_t* out, int lane);   // (below)
constexpr int kLocateOut = 40;          // S.misc[kLocateOut + 3 i]: {bin, below, in bin} of rank i (wg_brackets_regs; slots 40..51)
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
        const int nb1r = (nb1 + 63) & ~63;                   // wave_locate reads whole multiples of 64 bins
        for (int i = threadIdx.x; i < nb1r; i += blockDim.x) S.hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t o = ord[s][j];
            if (o != kAbsent) atomicAdd(&S.hist[(o - omin[s]) >> s1[s]], 1u);
        }
        __syncthreads();
        SL_BCLK(1);
        // the set's ranks located side by side, one wave each (round 4: were four whole-workgroup rounds of wave 0 + barrier)
        {
            const int wave = (int)(threadIdx.x >> 6), nwaves = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63);
#pragma unroll
            for (int i = 0; i < 2 * NBR; ++i)
                if (set_of[i >> 1] == s && (i % nwaves) == wave) wave_locate(S.hist, nb1r, rank[i], S.misc + kLocateOut + 3 * i, lane);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2 * NBR; ++i) {
                if (set_of[i >> 1] != s) continue;
                wlo[i] = omin[s] + (S.misc[kLocateOut + 3 * i] << s1[s]);
                const uint32_t span = s1[s] ? ((1u << s1[s]) - 1u) : 0u;
                whi[i] = (omax[s] - wlo[i]) < span ? omax[s] : wlo[i] + span;
                below[i] = S.misc[kLocateOut + 3 * i + 1];
            }
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
        {
            const int wave = (int)(threadIdx.x >> 6), nwaves = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63);
#pragma unroll
            for (int i = 0; i < 2 * NBR; ++i) {
                const int s = set_of[i >> 1];
                if (nv[s] > 0 && s1[s] > 0 && (i % nwaves) == wave) wave_locate(S.hist + i * 256, 256, rank[i] - below[i], S.misc + kLocateOut + 3 * i, lane);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2 * NBR; ++i) {
                const int s = set_of[i >> 1];
                if (!(nv[s] > 0 && s1[s] > 0)) continue;
                const int s2 = s1[s] > 8 ? s1[s] - 8 : 0;
                const uint32_t nlo = wlo[i] + (S.misc[kLocateOut + 3 * i] << s2);
                const uint32_t span = s2 ? ((1u << s2) - 1u) : 0u;
                whi[i] = (whi[i] - nlo) < span ? whi[i] : nlo + span;
                wlo[i] = nlo;
            }
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

}  // namespace sl
