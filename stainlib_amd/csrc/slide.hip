// slide.hip -- building blocks of the POOLED slide-level mode (BASELINE.json configs[4]):
// every tile of a slide, on every rank, is normalised with the stain matrix / 99th-percentile concentrations
// the reference would compute from the vertical concatenation of all the tiles as one tall image.
//
// The statistics of that tall image are sums and order statistics over all its pixels, so each rank reduces
// its own tiles to a few numbers and the host combines them with small all-reduces (stainlib_amd/distributed.py):
//   sl_tile_moments            per-tile {n, sum od, sum od od^T}            -> summed over tiles and ranks -> V
//   sl_slide_key_histogram_sampled + sl_slide_key_window   the fast path: estimate the key on a 1/64 pixel sample,
//                              then ONE sweep counts the keys below a 65536-key window around the estimate and
//                              histograms the keys inside it (exact; falls back to the rounds below on a miss)
//   sl_slide_key_histogram     256-bin histogram of the next 8 key bits among the keys that match a prefix
//                              -> all-reduced; 4 rounds pin one exact order statistic of the binary32 key
//   sl_slide_key_histogram16   the low 16 bits under a 16-bit prefix in one sweep (2 x 65536 bins, global atomics:
//                              only ~0.2 % of the pixels match) -> 8 + 8 + 16 bits = three sweeps per stage
//   sl_slide_key_next_above    smallest key above a given key (only when the last histogram holds no successor)
// Keys are the ones the per-tile path selects on: the pseudo-angle of the projected OD (tissue pixels) and the
// two lasso concentrations (all pixels), as order-preserving uint32 of their binary32 value.  Nothing
// per-pixel is stored: every round is one more sweep over the uint8 tiles.
#include "stats_kernels.hpp"
#include "sl_host.hpp"
#include <cmath>
#include <cstring>
#include <type_traits>

using namespace sl;

namespace {

// Both order statistics a stage needs are pinned in the SAME sweeps: the two angular percentiles share the key and
// differ in the prefix; the two concentration columns come out of one lasso solve.
struct SlideArgs {
    const uint8_t* rgb;
    int P, parts, n_items;
    float ylimf;
    int keyset;              // SL_KEYSET_ANGLE / SL_KEYSET_CONC
    float V[6];              // angle: V[c*2+k]
    double M[6];             // concentrations
    double lam;
    uint32_t prefix[2];
    int prefix_bits;
    uint32_t above[2];       // next_above: keys strictly greater than these
    uint32_t window_lo[2];   // window mode: histogram of key - window_lo over [window_lo, window_lo + 65536), count of keys below
    float win_flo[2], win_fhi[2];   // the same bounds as binary32 values (-inf / +inf: no cheap zone test for that target)
    uint32_t sample_mask;    // 0: every pixel; 2^s - 1: one 64-chunk row in 2^s (stratified over rows and items)
    const double* dyn;       // NULL, or the device-resident state of a device-driven pooled computation (SL_POOL_*): the basis, prefixes
                             // and windows are read from it by the kernel instead of coming from the host
};

// ---- layout of the device-resident pool state (doubles; include/stainlib_hip.h SL_POOL_*).  Every rank holds an identical copy:
// each step consumes all-reduced data only.
enum {
    kPoolM = SL_POOL_M, kPoolMaxC = SL_POOL_MAXC, kPoolStatus = SL_POOL_STATUS, kPoolMiss = SL_POOL_MISS,
    kPoolT = 10, kPoolNpx = 11, kPoolVd = 12, kPoolVf = 18,
    kPoolK = 24, kPoolG = 26, kPoolTotalS = 28, kPoolKs = 30, kPoolBelow = 32, kPoolPrefix = 34, kPoolWinLo = 36, kPoolWinFlo = 38,
    kPoolWinFhi = 40, kPoolRes = 43
};
static_assert(kPoolRes + 4 <= SL_POOL_STATE_DOUBLES, "");

// the host-filled fields a device-driven launch takes from the pool state instead
template <int KEYSET>
__device__ __forceinline__ void args_from_pool_state(SlideArgs& a) {
    const double* d = a.dyn;
    if (!d) return;                                              // uniform
    if (KEYSET == SL_KEYSET_ANGLE) { for (int i = 0; i < 6; ++i) a.V[i] = (float)d[kPoolVf + i]; }
    else { for (int i = 0; i < 6; ++i) a.M[i] = d[kPoolM + i]; }
    for (int t = 0; t < 2; ++t) {
        a.prefix[t] = (uint32_t)d[kPoolPrefix + t];
        a.window_lo[t] = (uint32_t)d[kPoolWinLo + t];
        a.win_flo[t] = (float)d[kPoolWinFlo + t];
        a.win_fhi[t] = (float)d[kPoolWinFhi + t];
    }
}

// One target's bookkeeping of a lane: matching keys are counted in runs (neighbouring pixels mostly fall into the
// same bin, and in the first round nearly all keys do: per-key LDS atomics on one address would serialise).
struct BinRun {
    uint32_t bin = 0xffffffffu, run = 0;
    __device__ __forceinline__ void add(uint32_t* hist, uint32_t b) {
        if (b == bin) { ++run; return; }
        if (run) atomicAdd(&hist[bin], run);
        bin = b; run = 1;
    }
    __device__ __forceinline__ void flush(uint32_t* hist) { if (run) atomicAdd(&hist[bin], run); run = 0; }
};

// MODE 0: 256-bin histogram of the next 8 bits (LDS, merged into hist at the end); 1: smallest key above a.above;
// 2: the LOW 16 bits of the keys whose top 16 bits match, counted straight into hist[2][65536] with global atomics
// (a 16-bit prefix leaves ~0.2 % of the pixels: two rounds in one sweep).  (The window sweep is k_slide_window below.)
template <int KEYSET, int MODE, bool ALIGNED>
__global__ __launch_bounds__(kSweepThreads, 4) void k_slide_keys(SlideArgs a, unsigned long long* hist, uint32_t* min_out) {
    args_from_pool_state<KEYSET>(a);
    constexpr bool NEXT_ABOVE = MODE == 1;
    constexpr bool LOW16 = MODE == 2;
    __shared__ RowTab s_tab;
    __shared__ uint32_t s_hist[2][256];
    __shared__ uint32_t s_min[2];
    s_tab.fill_b();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) (&s_hist[0][0])[i] = 0;
    if (threadIdx.x < 2) s_min[threadIdx.x] = 0xffffffffu;
    __syncthreads();
    const TabReaderB T = TabReaderB::make(s_tab);
    const int tid = threadIdx.x;
    float V[6];
    LassoK L;
    if (KEYSET == SL_KEYSET_ANGLE) {
        for (int i = 0; i < 6; ++i) V[i] = in_vgpr(a.V[i]);
    } else {
        lasso_consts(a.M, a.lam, L);
        vgpr(L);
    }
    const int sh = 24 - a.prefix_bits;
    const bool all = a.prefix_bits == 0;
    const uint32_t p0 = a.prefix[0], p1 = a.prefix[1], hs = all ? 0u : (uint32_t)(32 - a.prefix_bits);
    BinRun r0, r1;
    uint32_t best0 = 0xffffffffu, best1 = 0xffffffffu;
    const size_t nbytes = (size_t)a.P * 3;
    // one chunk (4 pixels) of a lane: keys, then the mode's bookkeeping
    auto process = [&](const Chunk& in, int cc, int c1) {
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            bool have = (cc < c1) & (ALIGNED | ((size_t)cc * 4 + px < (size_t)a.P));
            uint32_t o0, o1;
            if (KEYSET == SL_KEYSET_ANGLE) {
                const float2 er = T.gam_odf(T.addr(in, 3 * px)), eg = T.gam_odf(T.addr(in, 3 * px + 1)), eb = T.gam_odf(T.addr(in, 3 * px + 2));
                have = have & is_tissue_f(er.x, eg.x, eb.x, a.ylimf);
                o0 = o1 = f2ord(angle_key(V, er.y, eg.y, eb.y));
            } else {
                float c1f, c2f;
                lasso2(L, T.odf(T.addr(in, 3 * px)), T.odf(T.addr(in, 3 * px + 1)), T.odf(T.addr(in, 3 * px + 2)), c1f, c2f);
                o0 = f2ord(c1f); o1 = f2ord(c2f);
            }
            if (!have) continue;
            if (NEXT_ABOVE) {
                if (o0 > a.above[0]) best0 = min(best0, o0);
                if (o1 > a.above[1]) best1 = min(best1, o1);
            } else if (LOW16) {
                if ((o0 >> 16) == p0) atomicAdd(&hist[o0 & 0xffffu], 1ull);
                if ((o1 >> 16) == p1) atomicAdd(&hist[65536u + (o1 & 0xffffu)], 1ull);
            } else {
                if (all || (o0 >> hs) == p0) r0.add(s_hist[0], (o0 >> sh) & 255u);
                if (all || (o1 >> hs) == p1) r1.add(s_hist[1], (o1 >> sh) & 255u);
            }
        }
    };
    if (a.sample_mask == 63u) {
        // 1/64 sample: every wave owns at most ONE 64-chunk row per (tile, part) item, so the loads of four items are
        // issued together (one row at a time is a chain of exposed memory latencies: 3x slower)
        const int wave = tid >> 6, lane = tid & 63;
        for (int it0 = blockIdx.x; it0 < a.n_items; it0 += 4 * (int)gridDim.x) {
            Chunk ch[4];
            int ccs[4], c1s[4];
            bool ok[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int item = it0 + j * (int)gridDim.x;
                ok[j] = item < a.n_items;                                       // workgroup-uniform
                ccs[j] = 0; c1s[j] = 0;
                if (ok[j]) {
                    const int tile = item / a.parts, part = item % a.parts;
                    int c0, c1;
                    part_range((a.P + 3) >> 2, a.parts, part, c0, c1);
                    const int r0 = c0 >> 6;
                    const int row = r0 + ((item - r0) & 63) + wave * 64;        // rows with (row ^ item) & 63 == 0
                    ok[j] = row * 64 < c1;                                      // wave-uniform
                    ccs[j] = row * 64 + lane; c1s[j] = c1;
                    if (ok[j]) ch[j] = load_chunk_clamped<ALIGNED>(a.rgb + (size_t)tile * nbytes, nbytes, ccs[j], c1);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ok[j]) process(ch[j], ccs[j], c1s[j]);
            // parts longer than 8 x 64 rows (few tiles per launch: the host clamps parts to ~2048 / n): the wave walks on
            // through its row class, so that the sample stays 1 row in 64 over the WHOLE part (not only its top)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!ok[j]) continue;
                const int tile = (it0 + j * (int)gridDim.x) / a.parts;
                for (int cc = ccs[j] + 8 * 64 * 64; (cc & ~63) < c1s[j]; cc += 8 * 64 * 64)
                    process(load_chunk_clamped<ALIGNED>(a.rgb + (size_t)tile * nbytes, nbytes, cc, c1s[j]), cc, c1s[j]);
            }
        }
    } else {
        for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
            const int tile = item / a.parts, part = item % a.parts;
            const uint8_t* src = a.rgb + (size_t)tile * nbytes;
            int c0, c1;
            part_range((a.P + 3) >> 2, a.parts, part, c0, c1);
            if (a.sample_mask == 0u) {
                // every pixel: the next trip's two chunks are requested before this trip's arithmetic
                if (c0 >= c1) continue;
                Chunk nxt[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) nxt[u] = load_chunk_clamped<ALIGNED>(src, nbytes, c0 + tid + u * kSweepThreads, c1);
                for (int c = c0 + tid; c < c1; c += kSweepThreads * 2) {
                    Chunk in[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        in[u] = nxt[u];
                        nxt[u] = load_chunk_clamped<ALIGNED>(src, nbytes, c + (2 + u) * kSweepThreads, c1);
                    }
                    process(in[0], c, c1);
                    process(in[1], c + kSweepThreads, c1);
                }
                continue;
            }
            for (int c = c0 + tid; c < c1; c += kSweepThreads * 2) {
                // sampling (wave-uniform: c0 and the wave's first chunk are multiples of 64)
                const bool take0 = ((((uint32_t)c >> 6) ^ (uint32_t)item) & a.sample_mask) == 0u;
                const bool take1 = (((((uint32_t)c + kSweepThreads) >> 6) ^ (uint32_t)item) & a.sample_mask) == 0u;
                if (!(take0 | take1)) continue;
                Chunk in[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) in[u] = load_chunk_clamped<ALIGNED>(src, nbytes, c + u * kSweepThreads, c1);
                if (take0) process(in[0], c, c1);
                if (take1) process(in[1], c + kSweepThreads, c1);
            }
        }
    }
    if (NEXT_ABOVE) {
        for (int o = 32; o > 0; o >>= 1) {
            best0 = min(best0, (uint32_t)__shfl_xor((int)best0, o, 64));
            best1 = min(best1, (uint32_t)__shfl_xor((int)best1, o, 64));
        }
        if ((tid & 63) == 0) { atomicMin(&s_min[0], best0); atomicMin(&s_min[1], best1); }
        __syncthreads();
        if (tid < 2 && s_min[tid] != 0xffffffffu) atomicMin(&min_out[tid], s_min[tid]);
    } else if (!LOW16) {
        r0.flush(s_hist[0]); r1.flush(s_hist[1]);
        __syncthreads();
        for (int i = tid; i < 512; i += blockDim.x) {
            const uint32_t v = (&s_hist[0][0])[i];
            if (v) atomicAdd(&hist[i], (unsigned long long)v);
        }
    }
}

// ---- the WINDOW sweep: a 65536-key window per target at an arbitrary position (hist[2][65536] of key - window_lo) plus the
// number of keys below it (hist[2 * 65536 + t]).  With the window centred on an estimate from a pixel sample, one sweep
// pins an order statistic.  Like the per-tile selection sweeps it does NOT evaluate the ordered key of every pixel: cheap
// tests in the binary32 domain prove on which side of both windows a pixel lies; proven pixels are counted on the scalar
// unit (ballot popcounts), the few others (inside or next to a window: ~0.1 %) take the exact path under the exec mask.
//   angle (one key p = y / (x + |y|), x > 0, for both windows, window 0 below window 1):
//       p < lo0 - eps | hi0 + eps < p < lo1 - eps | p > hi1 + eps      tested as y <> bound * d, without the division
//   concentrations (the keys ARE cheap: max(min(a, s), 0) when g12 >= 0): c < lo or c > hi, compared as binary32 --
//       equivalent to the ordered-integer comparison except at +-0, which neither strict test claims
struct WinAcc {
    unsigned long long l_nb0 = 0, l_nb1 = 0;     // per lane: keys below window 0 / window 1
};

template <int KEYSET, bool ALIGNED, int kTrip, bool STREAM>
__device__ __forceinline__ void window_sweep(const uint8_t* src, int P, int c0, int c1, int t, const TabReaderB& T, const SlideArgs& a,
                                             const float* V, const LassoK& L, const float* thr, unsigned long long* hist, WinAcc& acc) {
    constexpr int nthreads = kSweepThreads;
    const size_t nbytes = (size_t)P * 3;
    const int lane = t & 63;
    const int w0 = __builtin_amdgcn_readfirstlane(c0 + (t & ~63));
    const uint32_t wlo0 = a.window_lo[0], wlo1 = a.window_lo[1];
    struct G { float2 v[12]; };
    struct Gc { float v[12]; };
    using GT = typename std::conditional<KEYSET == SL_KEYSET_ANGLE, G, Gc>::type;
    const float ylimf = in_vgpr(a.ylimf);
    uint32_t lc0 = 0, lc1 = 0;                  // per lane: pixels proven below window 0 / (angle: between the windows; conc: below window 1)
    auto g_at = [](const GT& g, int i) { return g.v[i]; };
    auto fetch = [&](int cc) { return load_chunk_clamped<ALIGNED, STREAM>(src, nbytes, cc, c1); };
    auto gather = [&](const Chunk& ch) {
        GT g;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if constexpr (KEYSET == SL_KEYSET_ANGLE) g.v[i] = T.gam_odf(T.addr(ch, i));
            else g.v[i] = T.odf(T.addr(ch, i));
        }
        return g;
    };
    auto exact = [&](uint32_t o0, uint32_t o1, bool counted0, bool counted1) {     // under the exec mask of the unproven lanes
        acc.l_nb0 += ((o0 < wlo0) & !counted0) ? 1u : 0u;
        acc.l_nb1 += ((o1 < wlo1) & !counted1) ? 1u : 0u;
        const uint32_t d0 = o0 - wlo0, d1 = o1 - wlo1;
        if (o0 >= wlo0 && d0 < 65536u) atomicAdd(&hist[d0], 1ull);
        if (o1 >= wlo1 && d1 < 65536u) atomicAdd(&hist[65536u + d1], 1ull);
    };
    // Scalar instructions are as scarce as vector ones here (one issue slot per SIMD visit), so the proven pixels are
    // counted per LANE (one add-with-carry off the compare's mask each) and the unproven ones are looked for once per
    // CHUNK: the exact path of a chunk's four pixels sits behind one branch.
    auto compute = [&](auto tail_tag, const GT& g, int cc) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        bool flag[4], cnt0[4], cnt1[4];
        auto inb = [&](int px) { return (cc < c1) & (ALIGNED | ((size_t)cc * 4 + px < (size_t)P)); };
        auto xy = [&](float odr, float odg, float odb, float& x, float& y) {                 // the operations of angle_key
            x = fmaf(V[4], odb, fmaf(V[2], odg, V[0] * odr));
            y = fmaf(V[5], odb, fmaf(V[3], odg, V[1] * odr));
        };
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            if constexpr (KEYSET == SL_KEYSET_ANGLE) {
                const float2 er = g_at(g, 3 * px), eg = g_at(g, 3 * px + 1), eb = g_at(g, 3 * px + 2);
                // is_tissue_f as a number: tv > 0 <=> tissue (integers below 2^24: exact); it rides in the min chains
                float tv = ylimf - fmaf(871.0f, er.x, fmaf(2929.0f, eg.x, 296.0f * eb.x));
                if (TAIL) tv = inb(px) ? tv : -1.0f;
                float x, y;
                xy(er.y, eg.y, eb.y, x, y);
                const float d = x + fabsf(y);
                const float tl0 = fmaf(thr[0], d, y), th0 = fmaf(thr[1], d, y), tl1 = fmaf(thr[2], d, y), th1 = fmaf(thr[3], d, y);
                const bool z0 = fminf(fminf(x, -tl0), tv) > 0.0f;                      // tissue, x > 0, p < lo0 - eps: below both windows
                const bool z1 = fminf(fminf(fminf(x, th0), -tl1), tv) > 0.0f;          // between the windows
                const bool z2 = fminf(fminf(x, th1), tv) > 0.0f;                       // above both
                cnt0[px] = z0; cnt1[px] = z0 | z1;
                lc0 += z0 ? 1u : 0u;
                lc1 += z1 ? 1u : 0u;
                flag[px] = (tv > 0.0f) & !(z0 | z1 | z2);
            } else {
                float c1f, c2f;
                lasso2(L, g_at(g, 3 * px), g_at(g, 3 * px + 1), g_at(g, 3 * px + 2), c1f, c2f);
                bool b0 = c1f < thr[0], b1 = c2f < thr[2];
                bool fast = (b0 | (c1f > thr[1])) & (b1 | (c2f > thr[3]));
                if (TAIL) { const bool in = inb(px); b0 = b0 & in; b1 = b1 & in; fast = fast | !in; }
                cnt0[px] = b0; cnt1[px] = b1;
                lc0 += b0 ? 1u : 0u;
                lc1 += b1 ? 1u : 0u;
                flag[px] = !fast;
            }
        }
        if (flag[0] | flag[1] | flag[2] | flag[3]) {
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                if (!flag[px]) continue;
                if constexpr (KEYSET == SL_KEYSET_ANGLE) {
                    float x, y;
                    xy(g_at(g, 3 * px).y, g_at(g, 3 * px + 1).y, g_at(g, 3 * px + 2).y, x, y);
                    const uint32_t o = f2ord(pseudo_angle(x, y));
                    exact(o, o, false, false);
                } else {
                    float c1f, c2f;
                    lasso2(L, g_at(g, 3 * px), g_at(g, 3 * px + 1), g_at(g, 3 * px + 2), c1f, c2f);
                    exact(f2ord(c1f), f2ord(c2f), cnt0[px], cnt1[px]);
                }
            }
        }
    };
    Chunk cur[kTrip], nx[kTrip];
#pragma unroll
    for (int k = 0; k < kTrip; ++k) { cur[k] = fetch(w0 + lane + k * nthreads); nx[k] = fetch(w0 + lane + (kTrip + k) * nthreads); }
    GT g[2];
    g[0] = gather(cur[0]);
    auto trip = [&](auto tail_tag, int cb) {
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
    };
    const int lim = ALIGNED ? c1 : min(c1, P >> 2);          // chunks made of in-range pixels only
    int cb = w0;
    for (; cb + (kTrip - 1) * nthreads + 64 <= lim; cb += nthreads * kTrip) trip(std::false_type{}, cb);
    if (cb < c1) trip(std::true_type{}, cb);                    // at most one ragged trip per wave
    // angle: window 1 lies above window 0, so "below window 1" = below window 0 or between the two
    acc.l_nb0 += lc0;
    acc.l_nb1 += KEYSET == SL_KEYSET_ANGLE ? lc0 + lc1 : lc1;
}

template <int KEYSET, bool ALIGNED>
__global__ __launch_bounds__(kSweepThreads, 4) void k_slide_window(SlideArgs a, unsigned long long* hist) {
    // Device-driven chain: a stage that already ended in a miss (or a slide without tissue) has poisoned M -- every concentration key
    // would be the same NaN pattern, one window bin taking a global atomic per pixel: 3.3 s for 256 tiles of 1024^2 (measured, round 6:
    // a slide of four tiles repeated, whose angular keys tie beyond the 65 536-key window).  Nothing to select: the window stays empty,
    // sl_pool_resolve reports the miss, the caller takes the radix rounds.
    if (a.dyn && ((int)a.dyn[kPoolMiss] != 0 || (int)a.dyn[kPoolStatus] != SL_TILE_OK)) return;      // uniform
    args_from_pool_state<KEYSET>(a);
    __shared__ RowTab s_tab;
    s_tab.fill_b();
    __syncthreads();
    const TabReaderB T = TabReaderB::make(s_tab);
    const int tid = threadIdx.x;
    float V[6] = {0, 0, 0, 0, 0, 0}, thr[4];
    LassoK L{};
    if (KEYSET == SL_KEYSET_ANGLE) {
        for (int i = 0; i < 6; ++i) V[i] = in_vgpr(a.V[i]);
        // thr: -(lo0 - eps), -(hi0 + eps), -(lo1 - eps), -(hi1 + eps)   (t = y - bound * d)
        thr[0] = in_vgpr(-(a.win_flo[0] - kAngleMargin)); thr[1] = in_vgpr(-(a.win_fhi[0] + kAngleMargin));
        thr[2] = in_vgpr(-(a.win_flo[1] - kAngleMargin)); thr[3] = in_vgpr(-(a.win_fhi[1] + kAngleMargin));
    } else {
        lasso_consts(a.M, a.lam, L);
        vgpr(L);
        thr[0] = in_vgpr(a.win_flo[0]); thr[1] = in_vgpr(a.win_fhi[0]); thr[2] = in_vgpr(a.win_flo[1]); thr[3] = in_vgpr(a.win_fhi[1]);
    }
    WinAcc acc;
    const bool stream = (size_t)a.P * 3 >= kStreamBytes;      // uniform
    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int tile = item / a.parts, part = item % a.parts;
        const uint8_t* src = a.rgb + (size_t)tile * a.P * 3;
        int c0, c1;
        part_range((a.P + 3) >> 2, a.parts, part, c0, c1);
        if (c0 >= c1) continue;
        if (stream) window_sweep<KEYSET, ALIGNED, kPhaseTrip, true>(src, a.P, c0, c1, tid, T, a, V, L, thr, hist, acc);
        else window_sweep<KEYSET, ALIGNED, kPhaseTrip, false>(src, a.P, c0, c1, tid, T, a, V, L, thr, hist, acc);
    }
    const unsigned long long nb0 = wave_sum(acc.l_nb0), nb1 = wave_sum(acc.l_nb1);
    if ((tid & 63) == 0) { if (nb0) atomicAdd(&hist[2u * 65536u], nb0); if (nb1) atomicAdd(&hist[2u * 65536u + 1u], nb1); }
}

// per-tile moment sums in a fixed order (run-to-run identical): sweep -> partials -> one thread per (tile, moment).
// Persistent like the sweep kernels of the per-phase schedule: the 64 KB table is written once per workgroup.
template <bool ALIGNED>
__global__ __launch_bounds__(kSweepThreads, 4) void k_tile_moment_partials(const uint8_t* rgb, int P, int parts, int n_items, float ylimf, double* partials) {
    __shared__ RowTab s_tab;
    __shared__ double s_red[kSweepThreads / 64][10];
    s_tab.fill_b();
    __syncthreads();
    const TabReaderB T = TabReaderB::make(s_tab);
    const int tid = threadIdx.x, lane = tid & 63;
    const bool stream = (size_t)P * 3 >= kStreamBytes;            // uniform
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int tile = item / parts, part = item % parts;
        const uint8_t* src = rgb + (size_t)tile * P * 3;
        int c0, c1;
        part_range((P + 3) >> 2, parts, part, c0, c1);
        Moments mo;
        uint32_t n_tissue = 0;
        if (c0 >= c1) {                                           // an empty trailing part (block-uniform): zeros
        } else if (stream) moments_sweep_b<ALIGNED, kPhaseTrip, true>(src, P, c0, c1, tid, kSweepThreads, T, ylimf, 6, nullptr, mo, n_tissue);
        else moments_sweep_b<ALIGNED, kPhaseTrip, false>(src, P, c0, c1, tid, kSweepThreads, T, ylimf, 6, nullptr, mo, n_tissue);
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
            partials[((size_t)tile * parts + part) * 10 + tid] = t;
        }
        __syncthreads();                                          // s_red is reused by the next item
    }
}
__global__ void k_sum_partials(const double* partials, int n, int parts, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 10) return;
    const int tile = i / 10, m = i % 10;
    double t = 0;
    for (int p = 0; p < parts; ++p) t += partials[((size_t)tile * parts + p) * 10 + m];
    out[i] = t;
}

float ord_to_float_host(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

int fill_args(SlideArgs& a, const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset, const double* basis_host) {
    if (!rgb || n <= 0 || h <= 0 || w <= 0 || !basis_host) return SL_ERR_BADARG;
    if ((long)h * w > (1L << 30)) return SL_ERR_BADARG;
    if (keyset != SL_KEYSET_ANGLE && keyset != SL_KEYSET_CONC) return SL_ERR_BADARG;
    SlParams p;
    sl_default_params(&p);
    if (!params_ok(params)) return SL_ERR_BADARG;
    if (params) p = *params;
    a.rgb = rgb;
    a.P = h * w;
    a.parts = parts_for((long)h * w);
    const long want = (4L * max_resident_grid() + n - 1) / n;   // ~4 items per persistent workgroup
    if (a.parts > want) a.parts = (int)(want < 1 ? 1 : want);
    a.n_items = n * a.parts;
    a.ylimf = (float)y_limit_for_threshold(p.luminosity_threshold) - 2048.0f;
    a.keyset = keyset;
    a.lam = p.lasso_lambda;
    for (int i = 0; i < 6; ++i) { a.V[i] = (float)basis_host[i]; a.M[i] = basis_host[i]; }
    a.prefix[0] = a.prefix[1] = 0; a.prefix_bits = 0; a.above[0] = a.above[1] = 0;
    a.window_lo[0] = a.window_lo[1] = 0; a.sample_mask = 0;
    a.win_flo[0] = a.win_flo[1] = -INFINITY; a.win_fhi[0] = a.win_fhi[1] = INFINITY;
    a.dyn = nullptr;
    return SL_OK;
}

// ------------------------------------------------------------------------------------------
// Device-driven pooled statistics (round 3): the decisions the host took between the sweeps -- eigenvectors, which histogram bin
// holds the wanted rank, where the window goes, whether it caught the rank, the stain matrix -- are single-workgroup kernels on
// the pool state, so that a whole pooled computation is one enqueued chain of sweeps, small all-reduces and these steps with no
// host read-back in between (stainlib_amd/distributed.py PooledSlideStatistics; graph-capturable on one rank).
// ------------------------------------------------------------------------------------------
__global__ void k_pool_begin(const double* mom11, double* st, double pct) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < SL_POOL_STATE_DOUBLES; ++i) st[i] = 0.0;
    double Vd[6];
    float Vf[6];
    const int status = eigvecs_from_moments(mom11, Vd, Vf);
    st[kPoolStatus] = (double)status;
    st[kPoolT] = mom11[0];
    st[kPoolNpx] = mom11[10];
    for (int i = 0; i < 6; ++i) { st[kPoolVd + i] = Vd[i]; st[kPoolVf + i] = (double)Vf[i]; }
    long long k;
    double g;
    percentile_pos(mom11[0], 100.0 - pct, k, g);                 // minPhi, maxPhi (macenko_stain_extractor.py:33-34)
    st[kPoolK] = (double)k; st[kPoolG] = g;
    percentile_pos(mom11[0], pct, k, g);
    st[kPoolK + 1] = (double)k; st[kPoolG + 1] = g;
    for (int t = 0; t < 2; ++t) { st[kPoolWinFlo + t] = -INFINITY; st[kPoolWinFhi + t] = INFINITY; }
}

// One radix round of the SAMPLE estimate (all-reduced 2 x 256 histogram of the next 8 bits under the current prefix); after the third
// the 65536-key window is centred on the estimate (distributed.py window_rank_pairs does the same on the host).
__global__ void k_pool_pick(double* st, const unsigned long long* hist, int keyset, int round) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double N = keyset == SL_KEYSET_ANGLE ? st[kPoolT] : st[kPoolNpx];
    for (int t = 0; t < 2; ++t) {
        const unsigned long long* h = hist + 256 * t;
        if (round == 0) {
            unsigned long long tot = 0;
            for (int b = 0; b < 256; ++b) tot += h[b];
            st[kPoolTotalS + t] = (double)tot;
            double f = N > 1.0 ? st[kPoolK + t] / (N - 1.0) : 0.0;
            f = f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f);
            double ks = tot > 0 ? floor(f * ((double)tot - 1.0)) : 0.0;
            if (tot == 0) st[kPoolMiss] = (double)((int)st[kPoolMiss] | (keyset == SL_KEYSET_ANGLE ? 1 : 2));
            st[kPoolKs + t] = ks; st[kPoolBelow + t] = 0.0; st[kPoolPrefix + t] = 0.0;
        }
        const unsigned long long want = (unsigned long long)(st[kPoolKs + t] - st[kPoolBelow + t]);
        unsigned long long cum = 0;
        int b = 0;
        for (; b < 255; ++b) { if (cum + h[b] > want) break; cum += h[b]; }     // first bin with cum(b) > want
        st[kPoolBelow + t] += (double)cum;
        st[kPoolPrefix + t] = (double)((((unsigned long long)st[kPoolPrefix + t]) << 8) | (unsigned long long)b);
    }
    if (round == 2) {
        bool usable = true;
        float flo[2], fhi[2];
        for (int t = 0; t < 2; ++t) {
            const unsigned long long est = (((unsigned long long)st[kPoolPrefix + t]) << 8) | 0x80ull;
            long long lo = (long long)est - 32768;
            lo = lo < 0 ? 0 : (lo > (long long)(0xffffffffll - 65535ll) ? (long long)(0xffffffffll - 65535ll) : lo);
            st[kPoolWinLo + t] = (double)lo;
            flo[t] = ord2f((uint32_t)lo); fhi[t] = ord2f((uint32_t)lo + 65535u);
            if (!isfinite(flo[t]) || !isfinite(fhi[t])) usable = false;
        }
        if (keyset == SL_KEYSET_ANGLE && !(flo[0] <= flo[1] && fhi[0] <= fhi[1])) usable = false;
        for (int t = 0; t < 2; ++t) { st[kPoolWinFlo + t] = usable ? (double)flo[t] : -INFINITY; st[kPoolWinFhi + t] = usable ? (double)fhi[t] : INFINITY; }
        for (int t = 0; t < 2; ++t) { st[kPoolPrefix + t] = 0.0; }
    }
}

// The all-reduced window histogram (2 x 65536 bins + 2 counts below): the keys of ranks k and k + 1 of both targets, or a miss.
// Angle stage: the stain matrix (macenko_stain_extractor.py:33-44) and the ranks of the concentration stage; concentration
// stage: maxC (normalizer.py:36,47).  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_pool_resolve(double* st, const unsigned long long* win, int keyset, double lam) {
    __shared__ unsigned long long s_seg[1024];
    __shared__ unsigned long long s_cum[1024];
    __shared__ float s_res[4];
    __shared__ int s_miss;
    const int tid = threadIdx.x;
    if (tid == 0) s_miss = 0;
    if (tid < 4) s_res[tid] = 0.0f;
    const double N = keyset == SL_KEYSET_ANGLE ? st[kPoolT] : st[kPoolNpx];
    for (int t = 0; t < 2; ++t) {
        const unsigned long long* h = win + 65536 * t;
        unsigned long long seg = 0;
        for (int j = 0; j < 64; ++j) seg += h[tid * 64 + j];
        __syncthreads();
        s_seg[tid] = seg;
        __syncthreads();
        if (tid == 0) { unsigned long long c = 0; for (int i = 0; i < 1024; ++i) { s_cum[i] = c; c += s_seg[i]; } }   // exclusive
        __syncthreads();
        const unsigned long long inside = s_cum[1023] + s_seg[1023];
        const unsigned long long below = win[2 * 65536 + t];
        const double kd = st[kPoolK + t];
        const unsigned long long k = (unsigned long long)(kd < 0 ? 0 : (kd > N - 1.0 ? N - 1.0 : kd));
        const unsigned long long k1 = (double)(k + 1) <= N - 1.0 ? k + 1 : k;
        const bool covered = N >= 1.0 && below <= k && k1 < below + inside;
        if (!covered) { if (tid == 0) s_miss = 1; continue; }                  // uniform
        const uint32_t lo = (uint32_t)st[kPoolWinLo + t];
        for (int which = 0; which < 2; ++which) {
            const unsigned long long want = (which ? k1 : k) - below;           // rank inside the window
            if (want >= s_cum[tid] && want < s_cum[tid] + s_seg[tid]) {         // exactly one thread
                unsigned long long c = s_cum[tid];
                for (int j = 0; j < 64; ++j) {
                    const unsigned long long v = h[tid * 64 + j];
                    if (want < c + v) { s_res[2 * t + which] = ord2f(lo + (uint32_t)(tid * 64 + j)); break; }
                    c += v;
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // The apply pass is enqueued behind this chain before the host has looked at the state: whenever the state is unusable at the
    // END of the chain (a window missed a rank in either stage, an empty tissue mask, a degenerate covariance) the concentration
    // stage leaves NaN in (M, maxC), which k_apply treats like any tile with unusable statistics -- it passes the tiles through
    // unchanged instead of writing exp(NaN) bytes (round-3 advisor finding).
    auto poison = [&]() {
        const double nan = nan_d();
        for (int i = 0; i < 6; ++i) st[kPoolM + i] = nan;
        st[kPoolMaxC] = st[kPoolMaxC + 1] = nan;
    };
    if (s_miss) {
        if (tid == 0) {
            st[kPoolMiss] = (double)((int)st[kPoolMiss] | (keyset == SL_KEYSET_ANGLE ? 1 : 2));
            if (keyset != SL_KEYSET_ANGLE) poison();
        }
        return;
    }
    if (tid < 4) st[kPoolRes + tid] = (double)s_res[tid];
    if (keyset == SL_KEYSET_ANGLE) {
        __shared__ double s_Vd[6], s_g[2];
        if (tid < 6) s_Vd[tid] = st[kPoolVd + tid];
        if (tid < 2) s_g[tid] = st[kPoolG + tid];
        __syncthreads();
        if (tid < 64) {
            double M[6];
            stain_matrix_from_angles(s_Vd, s_res, s_g, M, tid);
            if (tid == 0) {
                for (int i = 0; i < 6; ++i) st[kPoolM + i] = M[i];
                if (stain_matrix_singular(M)) st[kPoolStatus] = (double)SL_TILE_DEGENERATE_COV;
                long long k;
                double g;
                percentile_pos(st[kPoolNpx], 99.0, k, g);                       // normalizer.py:36,47
                st[kPoolK] = st[kPoolK + 1] = (double)k;
                st[kPoolG] = st[kPoolG + 1] = g;
                for (int t = 0; t < 2; ++t) { st[kPoolWinFlo + t] = -INFINITY; st[kPoolWinFhi + t] = INFINITY; st[kPoolWinLo + t] = 0.0; }
            }
        }
    } else if (tid == 0) {
        for (int t = 0; t < 2; ++t) st[kPoolMaxC + t] = np_lerp((double)s_res[2 * t], (double)s_res[2 * t + 1], st[kPoolG + t]);
        if ((int)st[kPoolMiss] != 0 || (int)st[kPoolStatus] != SL_TILE_OK) poison();      // (an angle-stage miss, an empty mask, ...)
        (void)lam;
    }
}

template <int NEXT>
void launch_keys(const SlideArgs& a, bool al, unsigned long long* hist, uint32_t* min_out, hipStream_t s) {
    const int mg = max_resident_grid();
    const dim3 g((unsigned)(a.n_items < mg ? a.n_items : mg)), b(kSweepThreads);
    if (a.keyset == SL_KEYSET_ANGLE) {
        if (al) hipLaunchKernelGGL((k_slide_keys<SL_KEYSET_ANGLE, NEXT, true>), g, b, 0, s, a, hist, min_out);
        else    hipLaunchKernelGGL((k_slide_keys<SL_KEYSET_ANGLE, NEXT, false>), g, b, 0, s, a, hist, min_out);
    } else {
        if (al) hipLaunchKernelGGL((k_slide_keys<SL_KEYSET_CONC, NEXT, true>), g, b, 0, s, a, hist, min_out);
        else    hipLaunchKernelGGL((k_slide_keys<SL_KEYSET_CONC, NEXT, false>), g, b, 0, s, a, hist, min_out);
    }
}

}  // namespace

extern "C" int sl_tile_moments(const uint8_t* rgb, int n, int h, int w, const SlParams* params, double* moments_out,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!rgb || !moments_out || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    const long P = (long)h * w;
    if (P > (1L << 30)) return SL_ERR_BADARG;
    int parts = parts_for(P);
    const size_t need = sizeof(double) * 10 * (size_t)parts * (size_t)n;
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 7u)) return SL_ERR_WORKSPACE;
    const int mg = max_resident_grid();
    const long want = (4L * mg + n - 1) / n;                      // ~4 items per persistent workgroup
    if (parts > want) parts = (int)(want < 1 ? 1 : want);
    const long items = (long)n * parts;
    SlParams p;
    sl_default_params(&p);
    if (!params_ok(params)) return SL_ERR_BADARG;
    if (params) p = *params;
    const float ylimf = (float)y_limit_for_threshold(p.luminosity_threshold) - 2048.0f;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g((unsigned)(items < mg ? items : mg)), b(kSweepThreads);
    if (aligned4(rgb, P))
        hipLaunchKernelGGL((k_tile_moment_partials<true>), g, b, 0, s, rgb, (int)P, parts, (int)items, ylimf, (double*)workspace);
    else
        hipLaunchKernelGGL((k_tile_moment_partials<false>), g, b, 0, s, rgb, (int)P, parts, (int)items, ylimf, (double*)workspace);
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((n * 10 + 255) / 256)), dim3(256), 0, s, (const double*)workspace, n, parts,
                       moments_out);
    return launch_status();
}

extern "C" int sl_slide_key_histogram(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset,
                                      const double* basis, const uint32_t* prefixes, int prefix_bits,
                                      unsigned long long* hist, void* stream) {
    SlideArgs a;
    const int rc = fill_args(a, rgb, n, h, w, params, keyset, basis);
    if (rc) return rc;
    if (!hist || !prefixes || (prefix_bits != 0 && prefix_bits != 8 && prefix_bits != 16 && prefix_bits != 24)) return SL_ERR_BADARG;
    a.prefix[0] = prefixes[0]; a.prefix[1] = prefixes[1]; a.prefix_bits = prefix_bits;
    launch_keys<0>(a, aligned4(rgb, (long)h * w), hist, nullptr, (hipStream_t)stream);
    return launch_status();
}

extern "C" int sl_slide_key_histogram16(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset,
                                        const double* basis, const uint32_t* prefixes16, unsigned long long* hist16, void* stream) {
    SlideArgs a;
    const int rc = fill_args(a, rgb, n, h, w, params, keyset, basis);
    if (rc) return rc;
    if (!hist16 || !prefixes16) return SL_ERR_BADARG;
    a.prefix[0] = prefixes16[0]; a.prefix[1] = prefixes16[1]; a.prefix_bits = 16;
    launch_keys<2>(a, aligned4(rgb, (long)h * w), hist16, nullptr, (hipStream_t)stream);
    return launch_status();
}

// The sampled variants of the two histogram entry points: the same counts over one 64-chunk row in 2^sample_log2
// (an unbiased, stratified pixel sample), for estimating where an order statistic lies.
extern "C" int sl_slide_key_histogram_sampled(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset,
                                              const double* basis, const uint32_t* prefixes, int prefix_bits, int sample_log2,
                                              unsigned long long* hist, void* stream) {
    SlideArgs a;
    const int rc = fill_args(a, rgb, n, h, w, params, keyset, basis);
    if (rc) return rc;
    if (!hist || !prefixes || sample_log2 < 0 || sample_log2 > 12) return SL_ERR_BADARG;
    if (prefix_bits != 0 && prefix_bits != 8 && prefix_bits != 16 && prefix_bits != 24) return SL_ERR_BADARG;
    a.prefix[0] = prefixes[0]; a.prefix[1] = prefixes[1]; a.prefix_bits = prefix_bits;
    a.sample_mask = (1u << sample_log2) - 1u;
    launch_keys<0>(a, aligned4(rgb, (long)h * w), hist, nullptr, (hipStream_t)stream);
    return launch_status();
}

extern "C" int sl_slide_key_window(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset,
                                   const double* basis, const uint32_t* window_lo, unsigned long long* hist_below, void* stream) {
    SlideArgs a;
    const int rc = fill_args(a, rgb, n, h, w, params, keyset, basis);
    if (rc) return rc;
    if (!hist_below || !window_lo) return SL_ERR_BADARG;
    a.window_lo[0] = window_lo[0]; a.window_lo[1] = window_lo[1];
    // the windows as binary32 values for the cheap zone tests; a target whose bounds are not ordinary numbers gets none
    bool usable = true;
    for (int t = 0; t < 2; ++t) {
        const uint32_t hi = window_lo[t] > 0xffffffffu - 65535u ? 0xffffffffu : window_lo[t] + 65535u;
        a.win_flo[t] = ord_to_float_host(window_lo[t]);
        a.win_fhi[t] = ord_to_float_host(hi);
        if (!std::isfinite(a.win_flo[t]) || !std::isfinite(a.win_fhi[t])) usable = false;
    }
    // angle: both targets share the key, and the zone tests assume window 0 lies below window 1
    if (keyset == SL_KEYSET_ANGLE && !(a.win_flo[0] <= a.win_flo[1] && a.win_fhi[0] <= a.win_fhi[1])) usable = false;
    // concentrations: the binary32 comparison stands in for the ordered-integer one only for c >= +0 (g12 < 0 included: exact keys either way)
    if (!usable)
        for (int t = 0; t < 2; ++t) { a.win_flo[t] = -INFINITY; a.win_fhi[t] = INFINITY; }
    {
        const bool al = aligned4(rgb, (long)h * w);
        const int mg = max_resident_grid();
        const dim3 g((unsigned)(a.n_items < mg ? a.n_items : mg)), b(kSweepThreads);
        hipStream_t s = (hipStream_t)stream;
        if (keyset == SL_KEYSET_ANGLE) {
            if (al) hipLaunchKernelGGL((k_slide_window<SL_KEYSET_ANGLE, true>), g, b, 0, s, a, hist_below);
            else    hipLaunchKernelGGL((k_slide_window<SL_KEYSET_ANGLE, false>), g, b, 0, s, a, hist_below);
        } else {
            if (al) hipLaunchKernelGGL((k_slide_window<SL_KEYSET_CONC, true>), g, b, 0, s, a, hist_below);
            else    hipLaunchKernelGGL((k_slide_window<SL_KEYSET_CONC, false>), g, b, 0, s, a, hist_below);
        }
    }
    return launch_status();
}

extern "C" int sl_slide_key_next_above(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset,
                                       const double* basis, const uint32_t* key_ords, uint32_t* min_out, void* stream) {
    SlideArgs a;
    const int rc = fill_args(a, rgb, n, h, w, params, keyset, basis);
    if (rc) return rc;
    if (!min_out || !key_ords) return SL_ERR_BADARG;
    a.above[0] = key_ords[0]; a.above[1] = key_ords[1];
    launch_keys<1>(a, aligned4(rgb, (long)h * w), nullptr, min_out, (hipStream_t)stream);
    return launch_status();
}


// ---- device-driven pooled statistics: see k_pool_* above and include/stainlib_hip.h ----
extern "C" int sl_pool_begin(const double* moments11, const SlParams* params, double* state, void* stream) {
    if (!moments11 || !state) return SL_ERR_BADARG;
    SlParams p;
    sl_default_params(&p);
    if (!params_ok(params)) return SL_ERR_BADARG;
    if (params) p = *params;
    hipLaunchKernelGGL(k_pool_begin, dim3(1), dim3(64), 0, (hipStream_t)stream, moments11, state, p.angular_percentile);
    return launch_status();
}

extern "C" int sl_pool_histogram(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset, const double* state,
                                 int round, int sample_log2, unsigned long long* hist, void* stream) {
    static const double dummy_basis[6] = {1, 0, 0, 0, 1, 0};
    SlideArgs a;
    const int rc = fill_args(a, rgb, n, h, w, params, keyset, dummy_basis);
    if (rc) return rc;
    if (!hist || !state || round < 0 || round > 3 || sample_log2 < 0 || sample_log2 > 12) return SL_ERR_BADARG;
    a.dyn = state;
    a.prefix_bits = 8 * round;
    a.sample_mask = (1u << sample_log2) - 1u;
    launch_keys<0>(a, aligned4(rgb, (long)h * w), hist, nullptr, (hipStream_t)stream);
    return launch_status();
}

extern "C" int sl_pool_pick(double* state, int keyset, int round, const unsigned long long* hist_reduced, void* stream) {
    if (!state || !hist_reduced || round < 0 || round > 2 || (keyset != SL_KEYSET_ANGLE && keyset != SL_KEYSET_CONC)) return SL_ERR_BADARG;
    hipLaunchKernelGGL(k_pool_pick, dim3(1), dim3(64), 0, (hipStream_t)stream, state, hist_reduced, keyset, round);
    return launch_status();
}

extern "C" int sl_pool_window(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset, const double* state,
                              unsigned long long* hist_below, void* stream) {
    static const double dummy_basis[6] = {1, 0, 0, 0, 1, 0};
    SlideArgs a;
    const int rc = fill_args(a, rgb, n, h, w, params, keyset, dummy_basis);
    if (rc) return rc;
    if (!hist_below || !state) return SL_ERR_BADARG;
    a.dyn = state;
    const bool al = aligned4(rgb, (long)h * w);
    const int mg = max_resident_grid();
    const dim3 g((unsigned)(a.n_items < mg ? a.n_items : mg)), b(kSweepThreads);
    hipStream_t s = (hipStream_t)stream;
    if (keyset == SL_KEYSET_ANGLE) {
        if (al) hipLaunchKernelGGL((k_slide_window<SL_KEYSET_ANGLE, true>), g, b, 0, s, a, hist_below);
        else    hipLaunchKernelGGL((k_slide_window<SL_KEYSET_ANGLE, false>), g, b, 0, s, a, hist_below);
    } else {
        if (al) hipLaunchKernelGGL((k_slide_window<SL_KEYSET_CONC, true>), g, b, 0, s, a, hist_below);
        else    hipLaunchKernelGGL((k_slide_window<SL_KEYSET_CONC, false>), g, b, 0, s, a, hist_below);
    }
    return launch_status();
}

extern "C" int sl_pool_resolve(double* state, int keyset, const unsigned long long* window_reduced, const SlParams* params, void* stream) {
    if (!state || !window_reduced || (keyset != SL_KEYSET_ANGLE && keyset != SL_KEYSET_CONC)) return SL_ERR_BADARG;
    SlParams p;
    sl_default_params(&p);
    if (!params_ok(params)) return SL_ERR_BADARG;
    if (params) p = *params;
    hipLaunchKernelGGL(k_pool_resolve, dim3(1), dim3(1024), 0, (hipStream_t)stream, state, window_reduced, keyset, p.lasso_lambda);
    return launch_status();
}
