// stats_linalg.hpp -- thread-0 arithmetic of the finish steps: Jacobi eigh, stain matrix from angles, the box of stain matrices of the merged sweep.
// Part of stats_kernels.hpp (split by phase in round 4, no functional change); include that umbrella, not this file.
#pragma once
#include "stats_common.hpp"

namespace sl {

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
// w_out (optional, [3]): the eigenvalues, largest first (after the canonical choice of a null space: as computed)
__device__ __forceinline__ int eigvecs_from_moments(const double* sum, double* Vd, float* Vf, double* w_out = nullptr) {
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
    if (w_out) { w_out[0] = w2; w_out[1] = w1; w_out[2] = w0; }
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
    // two-sweep schedule (stats_twosweep.hpp): the box also covers a TILT of the eigenvector plane against the sample's.  The exact map
    // then has a component E_i = (A.W_i . nrm) nrm^T off the plane of the centre's map: a_i(M; x) <= ... + zeta_i |nrm . x|
    int tilted;                // 0: the box was built under the exact eigenvectors (three-sweep schedule): no such component
    int pad2_;
    double nrm[3];             // unit normal of the SAMPLE's eigenvector plane
    double zeta[2];            // bound on |A.W_i . nrm| over the box
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
        mk.tilted = 0; mk.pad2_ = 0;
        mk.nrm[0] = mk.nrm[1] = mk.nrm[2] = 0.0; mk.zeta[0] = mk.zeta[1] = 0.0;
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
        if (mk.tilted) ok = ok & (fabs(A.W[i][0] * mk.nrm[0] + A.W[i][1] * mk.nrm[1] + A.W[i][2] * mk.nrm[2]) <= mk.zeta[i]);
    }
    return ok;
}

}  // namespace sl
