// stats_sweeps.hpp -- per-pixel sweeps shared by both schedules: moments + sample (sweep 1), the selection sweeps (merged / concentration).
// Part of stats_kernels.hpp (split by phase in round 4, no functional change); include that umbrella, not this file.
#pragma once
#include "stats_linalg.hpp"

namespace sl {

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
// whose chunk the draw selects stores pixel 0 or 3 of it.  Everything that depends on the row alone is scalar (hash, draw, the
// row's slot in the sample); a lane contributes its two constants (SampleLane), so a chunk costs one compare, one v_perm and the
// predicated store -- 2 vector instructions where the per-lane form of rounds 1-3 (chunk index, mask, two compares, select, two
// shifts, address) cost 9, i.e. 1.75 of sweep 1's 22 per pixel (ISA count, round 4).  The stored words are the same.
struct SampleLane {
    uint32_t lm;        // lane & (cmask & 63): compared with the draw's low bits
    uint32_t idx;       // lane >> cps_log2 (0 when a sampling block spans whole rows): the lane's slot relative to the row's
    __device__ __forceinline__ static SampleLane make(int lane, int cps_log2) {
        const uint32_t cmask = (1u << cps_log2) - 1u;
        return SampleLane{(uint32_t)lane & cmask & 63u, cps_log2 <= 6 ? (uint32_t)lane >> cps_log2 : 0u};
    }
};
template <bool ALIGNED, bool TAIL>
__device__ __forceinline__ void sample_row(const Chunk& ch, int row0, int lane, const SampleLane& sl, int c1, int P, int cps_log2, uint32_t* samp) {
    const uint32_t h = sample_hash((uint32_t)row0 >> sample_group_shift(cps_log2));
    const uint32_t cmask = (1u << cps_log2) - 1u;
    const uint32_t sel = (h >> 8) & cmask;
    const bool last = (h >> 31) != 0;                                  // uniform: pixel 3 instead of pixel 0
    const bool row_hit = ((((uint32_t)row0 ^ sel) & cmask) >> 6) == 0u;   // uniform: the draw falls into this row (always, up to 64 chunks per block)
    if (samp && row_hit && sl.lm == (sel & 63u)) {
        // pixel 0: w0 as it is (stray top byte: readers ignore it); pixel 3: w2 >> 8 -- one v_perm with a uniform selector
        const uint32_t v = __builtin_amdgcn_perm(ch.w2, ch.w0, last ? 0x0c070605u : 0x03020100u);
        bool ok = true;
        if (TAIL) {
            const int cc = row0 + lane;
            ok = (cc < c1) & (ALIGNED | ((size_t)cc * 4 + (last ? 3 : 0) < (size_t)P));
        }
        if (ok) as_global(samp)[((uint32_t)row0 >> cps_log2) + sl.idx] = v;
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
    const SampleLane slane = SampleLane::make(lane, cps_log2);
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
        for (int k = 0; k < kTrip; ++k) sample_row<ALIGNED, decltype(tail_tag)::value>(cur[k], cb + k * nthreads, lane, slane, c1, P, cps_log2, samp);
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

}  // namespace sl
