// stats_finish.hpp -- finish-step primitives: key functors, refine passes, bracket census, histogram picks, the raw-candidate sink, sample brackets.
// Part of stats_kernels.hpp (split by phase in round 4, no functional change); include that umbrella, not this file.
#pragma once
#include "stats_selection.hpp"
#include "stats_sweeps.hpp"

namespace sl {

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
        for (int u = 0; u < U; ++u) { const int i = tid + u * bd; kn[u] = as_global(cand)[i < n ? i : 0]; }
        for (int i0 = tid; i0 < n; i0 += U * bd) {
            float k[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                k[u] = kn[u];
                const int i = i0 + (U + u) * bd;
                kn[u] = as_global(cand)[i < n ? i : 0];
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
// that the eight call sites of a trip stay small).  Round 4 tried it INLINED: -0.7 % with the sweeps inlined in the kernel (and
// spills in their loops as soon as anything else changed); with every phase out of line nothing, neither in the sweeps nor in the
// finish steps' refine passes -- and one instantiation of the select sweep hung with it (hazard 3).  It stays a call.
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
        if (base + i < cap) as_global(dst)[base + i] = buf[i];
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
    for (int u = 0; u < U; ++u) wn[u] = as_global(raw)[min(i0 + u * 64 + lane, last)];
    for (; i0 < n_raw; i0 += step) {
        uint32_t w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            w[u] = wn[u];
            wn[u] = as_global(raw)[min(i0 + step + u * 64 + lane, last)];       // next trip (clamped, never predicated)
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

// brackets of both angular quantiles from the sample (THREADS = blockDim.x)
// box (optional, float[4] = {lo0, hi0, lo1, hi1}): where the merged sweep may assume the two percentile angles to fall.  When
// both 6-sigma brackets are closed it is their central kBoxFrac; when one is open (a small tissue sample: the rank minus 6 sigma
// leaves it) a second pair at kBoxZ sigma is located in the same register-resident keys -- costs a histogram pass only then.
constexpr float kBoxZ = 3.6f;
template <int THREADS>
// zs: scale of the bracket half-widths (a clustered sample's ranks scatter by sqrt(design effect) more than an independent one's)
__device__ __forceinline__ void angle_brackets(const SampleAngleKey& key, int n_sample, double pct, float* lo, float* hi,
                                               SelScratch& S, float* box = nullptr, float zs = 1.0f) {
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
            w[g] = b < n_sample ? as_global(key.sample)[b] : 0u;
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
    wg_brackets_regs<1, KPT, 2>(ord, set_of, p2, lo, hi, S, kBracketZ * zs);
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
            const bool inside = n > 0.0 && floor(q * (n - 1.0) - (double)(kBoxZ * zs) * sqrt(fmax(q * (1.0 - q) * n, 0.0))) - 1.0 >= 0.0;   // block-uniform
            float blo[2] = {-INFINITY, -INFINITY}, bhi[2] = {INFINITY, INFINITY};
            __syncthreads();
            if (inside) wg_brackets_regs<1, KPT, 2>(ord, set_of, p2, blo, bhi, S, kBoxZ * zs);
            box[0] = blo[0]; box[1] = bhi[0]; box[2] = blo[1]; box[3] = bhi[1];
        }
    }
}
// brackets of the 99th percentile of both concentration columns from the sample (normalizer.py:36,47)
template <int THREADS>
__device__ __forceinline__ void conc_brackets(const SampleConcKey& key, int n_sample, float* lo, float* hi, SelScratch& S, float zs = 1.0f) {
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
            w[g] = b < n_sample ? as_global(key.sample)[b] : 0u;
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
    wg_brackets_regs<2, KPT, 2>(ord, set_of, p2, lo, hi, S, kBracketZ * zs);
}

}  // namespace sl
