// stats_dict.hpp -- Vahadane: class-moment dictionary learning (sweeps, one-lane solve, iteration control).
// Part of stats_kernels.hpp (split by phase in round 4, no functional change); include that umbrella, not this file.
#pragma once
#include "stats_finish.hpp"

namespace sl {

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

}  // namespace sl
