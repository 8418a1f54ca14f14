// stats_twosweep.hpp -- the two-read-sweep Macenko schedule (round 5): phase 0 (cluster sample, its eigenvectors, brackets, box, colour-cube
// mask), the merged moments + candidates sweep, and the verification the finish runs once the exact eigenvectors are known.
// Part of stats_kernels.hpp; include that umbrella, not this file.
//
// Why a sweep fewer.  macenko_stain_extractor.py:22-34 needs the eigenvectors V of ALL tissue pixels before a single angle can be taken,
// so the three-sweep schedule reads the tile once for the moments and once more to collect the candidates of the angular (and, since
// round 3, concentration) order statistics.  Both selection sweeps only ever PROVE pixels plain; what they cannot prove is collected raw
// and evaluated exactly by the finish.  So the proof may just as well run against an ESTIMATE V~ of the eigenvectors, made before the
// first sweep from a sample -- provided the finish can show afterwards that the proof also holds for the exact V:
//   * the plain cone between the two angular brackets is two half-planes of the projected pixel t = V^T od, i.e. two HALF-SPACES of
//     optical-density space, gH . od > 0 and gL . od > 0 with gH = V~ nH, gL = V~ nL: 3-vectors that make no reference to the basis
//     inside the plane -- an in-plane rotation between V~ and V (the large part of the sampling error: ~ sqrt(l2 / l1 / n)) moves
//     nothing.  What does matter is the TILT of the plane.  With n the exact unit normal, g' = g - n (n . g) lies in plane(V) and
//         g' . od = g . od - (n . g)(n . od)  >=  g . od - |n . g| (|n~ . od| + |n - n~|_inf S),      S = od_r + od_g + od_b,
//     so a pixel with  g . od > kappa1 |n~ . od| + kappa2 S  is on the positive side of the line { t : (V^T g) . t = 0 } of the EXACT
//     projection as soon as  |n . g| <= kappa1  and  |n . g| |n - n~|_inf <= kappa2  -- two numbers the finish checks (ts_verify).
//     n~ . od is the pixel's coordinate off the sample's plane (the third principal component: small for every stained pixel), which is
//     why the margin costs few candidates (tools/two_sweep_sim.py: 3.4-4.0 % of the tissue against 3.0-3.5 % with the exact V);
//   * the box of stain matrices of round 3 (merged_box) gets five tilts of the plane beside its nine in-plane grid points; the exact
//     map's component off the sample's plane is E_i = (A.W_i . n~) n~^T exactly, so the concentration test gains one term
//     zeta_i |n~ . od|, and merged_verify one check.
// A tile whose checks fail -- or whose sample gives no usable estimate, or whose colour cube would leave too many pixels to the exact
// test -- takes the three-sweep schedule from its exact moments on: results never depend on the estimate (SlParams.twosweep_out says
// what happened).
#pragma once
#include "stats_cube.hpp"

namespace sl {

constexpr double kClusterDeff = 2.0;     // design effect assumed for the cluster sample's rank statistics (brackets widen by its square root)
constexpr double kTiltZ = 5.0;           // a-priori tilt bound tau = kTiltZ x the standard error a Gaussian cloud of the sample's size would give
constexpr double kTiltZ4 = 4.0;          // ... and kTiltZ4 x the standard error the sample's own fourth moments give (whichever is larger)
constexpr double kTsMinTau = 2e-4, kTsMaxTau = 0.05;
constexpr int kTsBackoff = 3;            // automatic mode: tiles a workgroup does not try on after one of its tiles declined in phase 0 (k_fused)
constexpr double kTsAutoMaxTau = 6e-3;   // automatic mode: a Gaussian tilt bound above this leaves phase 0 right after the eigen-solve (see fused_phase0)
constexpr int kTsMinTissue = 256;        // tissue entries the sample must hold for an estimate
constexpr int kTsMaxSharePct = 40;       // above this share of sample pixels in ambiguous cells the two-sweep schedule is declined (measured, interleaved: i.i.d. tiles at
                                         // 13 % gain 6 %, spatially smooth synthetic tiles at 60 % lose 11 % against the three-sweep schedule)
constexpr int kTsFn = 9;                 // functionals per channel and cell index of the two-sweep cube (ts_cube_tables)
constexpr int kTsTabFloats = kTsFn * 3 * 32;

// what became of the two-sweep attempt of a tile (SlParams.twosweep_out)
enum { kTsDirect = 1, kTsOff = 0, kTsNoEstimate = -1, kTsShare = -2, kTsPlane = -3, kTsBracket = -4, kTsLists = -5 };

struct TwoSweep {
    int ok;                 // phase 0 left an estimate: sweep 1 collects candidates under it
    int dense;              // the tile's sample buffer holds the cluster sample (sweep 1 ran without the in-sweep sampler)
    int pad_;
    int why;                // kTs*
    double Vd[6];           // V~[c][k]
    double nd[3];           // unit normal of plane(V~)
    double gH[3], gL[3];    // unit normals of the two half-spaces of the plain cone (V~ nH, V~ nL)
    double kappa1, kappa2;  // what ts_verify demands of the exact plane
    double tau;             // a-priori tilt bound
    float lo0, hi1;         // outer ends of the angular brackets: pseudo-angles under V~ (may be open)
    // the sweep's constants, binary32
    float fgH[3], fgL[3];   // gH - (kappa2 + rounding allowance) 1, likewise gL:  tH = fgH . od = gH . od - kappa2' S
    float fn[3], fk1;       // n~, kappa1 (rounded up)
    float W[2][3], kt[2];   // the box centre's interior solution a~_i = W_i . od + kt_i
    float eps[2], zeta[2], thr[2];       // plain_i  <=>  a~_i + eps_i (|a~_1| + |a~_2|) + zeta_i |n~ . od| < thr_i
};

// ------------------------------------------------------------------------------------------
// the cluster sample
// ------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(1))) U32U { uint32_t v; };
// entry b of the cluster sample as r | g << 8 | b << 16 (top byte 0); wl = lines per stratum (P >= 2)
__device__ __forceinline__ uint32_t cluster_word(const uint8_t* src, int P, uint32_t wl, uint32_t b) {
    const uint32_t i = b / (uint32_t)kClusterPx, j = b % (uint32_t)kClusterPx;
    const uint32_t h = sample_hash(i);
    const uint32_t line = i * wl + __umulhi(h << 8, wl);           // a draw in [0, wl) off 24 hash bits (no runtime division)
    uint32_t px = (line * 128u + 2u) / 3u + (h >> 28) % 3u + (uint32_t)kClusterStep * j;      // (3 P <= 3 x 2^30 fits 32 bits)
    px = px < (uint32_t)(P - 2) ? px : (uint32_t)(P - 2);                                       // 4 bytes from 3 px stay inside the tile
    return ((SL_GLOBAL const U32U*)(as_global(src) + 3 * (size_t)px))->v & 0xffffffu;
}
// ------------------------------------------------------------------------------------------
// the box of stain matrices under a tilted plane (one whole wave; lanes 0..44 = 9 grid points x 5 tilts)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void ts_box(const double* Vd, const double* nd, double tau, const float* box, double lam, int lane, MergedConc& mk) {
    const bool finite = (box[0] > -INFINITY) & (box[1] < INFINITY) & (box[2] > -INFINITY) & (box[3] < INFINITY);
    const int g = lane % 9, var = (lane / 9) % 5;
    const int i0 = g % 3, i1 = g / 3;
    // the basis of this lane's plane: V~ itself, or one of its columns tilted by +-tau towards the normal (Gram-Schmidt)
    double u1[3], u2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { u1[c] = Vd[2 * c]; u2[c] = Vd[2 * c + 1]; }
    {
        const double s = (var == 1 || var == 3) ? tau : ((var == 2 || var == 4) ? -tau : 0.0);
        const bool first = var == 1 || var == 2;
        double a[3], b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { a[c] = (first ? u1[c] : u2[c]) + s * nd[c]; b[c] = first ? u2[c] : u1[c]; }
        const double na = 1.0 / sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] *= na;
        const double ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) b[c] -= ab * a[c];
        const double nb = 1.0 / sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) { b[c] *= nb; u1[c] = first ? a[c] : b[c]; u2[c] = first ? b[c] : a[c]; }
    }
    double U[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) { U[2 * c] = u1[c]; U[2 * c + 1] = u2[c]; }
    const double m0 = 0.5 * ((double)box[0] + (double)box[1]), r0 = 0.5 * ((double)box[1] - (double)box[0]);
    const double m1 = 0.5 * ((double)box[2] + (double)box[3]), r1 = 0.5 * ((double)box[3] - (double)box[2]);
    const double p0 = finite ? m0 + (double)(i0 - 1) * r0 : -0.25;
    const double p1 = finite ? m1 + (double)(i1 - 1) * r1 : 0.25;
    double M[6];
    stain_matrix_from_phi(U, angle_of_pseudo(p0), angle_of_pseudo(p1), M);
    LassoD A, C;
    lasso_affine_d(M, lam, A);
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                 // the centre: grid point 4 of the untilted plane = lane 4
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
    double z0 = fabs(A.W[0][0] * nd[0] + A.W[0][1] * nd[1] + A.W[0][2] * nd[2]), z1 = fabs(A.W[1][0] * nd[0] + A.W[1][1] * nd[1] + A.W[1][2] * nd[2]);
    const bool live = lane < 45;
    const bool lane_bad = live & !((e0 <= kBoxMaxEps) & (e1 <= kBoxMaxEps) & (q0 <= 1.0) & (q1 <= 1.0) & (A.g12 >= 0.0) & (z0 <= 1.0) & (z1 <= 1.0));
    const bool any_bad = __ballot(lane_bad) != 0ull;
    if (!live || lane_bad) e0 = e1 = q0 = q1 = z0 = z1 = 0.0;
    for (int o = 32; o > 0; o >>= 1) {
        e0 = fmax(e0, __shfl_xor(e0, o, 64)); e1 = fmax(e1, __shfl_xor(e1, o, 64));
        q0 = fmax(q0, __shfl_xor(q0, o, 64)); q1 = fmax(q1, __shfl_xor(q1, o, 64));
        z0 = fmax(z0, __shfl_xor(z0, o, 64)); z1 = fmax(z1, __shfl_xor(z1, o, 64));
    }
    if (lane == 0) {
        mk.ok = (finite && !any_bad) ? 1 : 0;
        mk.pad_ = 0;
        mk.C = C;
        LassoK Lc;
        lasso_consts(Mc, lam, Lc);
        mk.Lc = Lc;
        mk.tilted = 1; mk.pad2_ = 0;
        const double e[2] = {e0, e1}, q[2] = {q0, q1}, z[2] = {z0, z1};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int k = 0; k < 2; ++k) mk.u[i][k] = (float)(C.W[i][0] * Vd[k] + C.W[i][1] * Vd[2 + k] + C.W[i][2] * Vd[4 + k]);
            mk.kt[i] = (float)C.k[i];
            mk.eta[i] = 4e-6 * (kOdMax * (fabs(C.W[i][0]) + fabs(C.W[i][1]) + fabs(C.W[i][2])) + fabs(C.k[i]) + 1.0);
            mk.eps[i] = (float)(kBoxInflate * e[i] + 1e-7);
            mk.rho[i] = kBoxInflate * q[i] + mk.eta[i];
            mk.zeta[i] = kBoxInflate * z[i] + 1e-7;
            mk.nrm[i] = nd[i];
        }
        mk.nrm[2] = nd[2];
    }
}
// thread 0, after the sample's concentration brackets [lo, hi] under the box centre: merged_thresholds with the tilt term.
// zref: how far off the sample's plane the pixels near the brackets may sit (a few standard deviations of the third component)
__device__ __forceinline__ void ts_thresholds(MergedConc& mk, float lo0, float lo1, float hi0, float hi1, float zref) {
    const float lo[2] = {lo0, lo1}, hi[2] = {hi0, hi1};
    bool ok = mk.ok != 0;
    float ref[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) ref[i] = hi[i] < INFINITY ? hi[i] : 2.0f * lo[i] + 1.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float delta = mk.eps[i] * (ref[i] + 1.5f * ref[1 - i]) + (float)mk.rho[i] + (float)mk.zeta[i] * zref;
        mk.L[i] = lo[i] - delta;
        mk.H[i] = hi[i] + delta;
        ok = ok & (mk.L[i] > 0.0f) & (lo[i] > -INFINITY);
        mk.thr[i] = mk.L[i] - (float)mk.rho[i] - 1e-6f * fabsf(mk.L[i]);
    }
    if (!ok) {
        mk.ok = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) { mk.u[i][0] = mk.u[i][1] = mk.kt[i] = mk.eps[i] = 0.0f; mk.thr[i] = INFINITY; mk.L[i] = mk.H[i] = INFINITY; mk.zeta[i] = 0.0; }
    }
}

// ------------------------------------------------------------------------------------------
// the colour cube of the two-sweep tests (the construction of stats_cube.hpp on other functionals)
// ------------------------------------------------------------------------------------------
// ctab[f][ch][k]:  0 luminance sum at the cell's lowest byte (exact)   1 min tH   2 min tL   3 min z   4 max z   (z = n~ . od)
//                  5 min a1   6 max a1   7 min a2   8 max a2   (without the constants kt)
struct TsCubeConsts { float k1, s_ang, kt[2], eps[2], zeta[2], thr[2]; };
__device__ __forceinline__ TsCubeConsts ts_cube_tables(const TabView& tab, const TwoSweep& ts, float* ctab, int tid) {
    TsCubeConsts cc;
    cc.k1 = ts.fk1;
    float sg = 0.0f;
    for (int c = 0; c < 3; ++c) sg = fmaxf(sg, fmaxf(fabsf(ts.fgH[c]), fabsf(ts.fgL[c])));
    float sn = fabsf(ts.fn[0]) + fabsf(ts.fn[1]) + fabsf(ts.fn[2]);
    cc.s_ang = 8e-6f * ((float)kOdMax * (3.0f * sg + ts.fk1 * sn) + 1.0f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        cc.kt[i] = ts.kt[i]; cc.eps[i] = ts.eps[i]; cc.zeta[i] = ts.zeta[i];
        const float ca = fabsf(ts.W[i][0]) + fabsf(ts.W[i][1]) + fabsf(ts.W[i][2]);
        cc.thr[i] = ts.thr[i] - 8e-6f * ((float)kOdMax * (ca * (1.0f + 2.0f * ts.eps[i]) + ts.zeta[i] * sn) + fabsf(ts.kt[i]) + 1.0f);
    }
    for (int e = tid; e < kTsTabFloats; e += (int)blockDim.x) {
        const int f = e / 96, ch = (e / 32) % 3, k = e & 31;
        const float od_lo = tab.odf((uint32_t)(8 * k + 7)), od_hi = tab.odf((uint32_t)(8 * k));   // od falls with the byte
        float out;
        if (f == 0) {
            const float w = ch == 0 ? 871.0f : (ch == 1 ? 2929.0f : 296.0f);
            out = w * tab.gam((uint32_t)(8 * k));
        } else {
            float c;
            switch (f) {
                case 1: c = ts.fgH[ch]; break;
                case 2: c = ts.fgL[ch]; break;
                case 3: case 4: c = ts.fn[ch]; break;
                case 5: case 6: c = ts.W[0][ch]; break;
                default: c = ts.W[1][ch]; break;
            }
            const bool want_max = (f == 4) | (f == 6) | (f == 8);
            const float a = c * od_lo, b = c * od_hi;
            out = want_max ? fmaxf(a, b) : fminf(a, b);
        }
        ctab[e] = out;
    }
    return cc;
}
// the verdict of one cell from its ten sums v[f] (kt already added to 5..8)
__device__ __forceinline__ bool ts_cell_verdict(const float* v, const TsCubeConsts& cc, float ylimf) {
    const bool no_tissue = v[0] >= ylimf;
    const float zabs = fmaxf(fabsf(v[3]), fabsf(v[4]));
    const bool cone = fmaf(-cc.k1, zabs, fminf(v[1], v[2])) > cc.s_ang;
    const float sa = fmaxf(fabsf(v[5]), fabsf(v[6])) + fmaxf(fabsf(v[7]), fabsf(v[8]));
    const bool conc1 = fmaf(cc.zeta[0], zabs, fmaf(cc.eps[0], sa, v[6])) < cc.thr[0], conc2 = fmaf(cc.zeta[1], zabs, fmaf(cc.eps[1], sa, v[8])) < cc.thr[1];
    return (no_tissue || cone) && conc1 && conc2;                    // NaN anywhere => not plain
}
__device__ __forceinline__ bool ts_cell_plain(const float* ctab, const TsCubeConsts& cc, float ylimf, uint32_t p) {
    const int r5 = (int)((p >> 3) & 31u), g5 = (int)((p >> 11) & 31u), b5 = (int)((p >> 19) & 31u);
    float v[kTsFn];
#pragma unroll
    for (int f = 0; f < kTsFn; ++f)
        v[f] = ctab[(f * 3 + 1) * 32 + g5] + ctab[(f * 3 + 2) * 32 + b5] + (f >= 5 ? cc.kt[(f - 5) >> 1] : 0.0f) + ctab[(f * 3) * 32 + r5];
    return ts_cell_verdict(v, cc, ylimf);
}
template <int NT>
__device__ __forceinline__ void ts_cube_mask(const float* ctab, const TsCubeConsts& cc, float ylimf, uint32_t* bits, int tid) {
    for (int w = tid; w < kCubeWords; w += NT) {
        const int g5 = w & 31, b5 = w >> 5;
        float part[kTsFn];
#pragma unroll
        for (int f = 0; f < kTsFn; ++f) part[f] = ctab[(f * 3 + 1) * 32 + g5] + ctab[(f * 3 + 2) * 32 + b5] + (f >= 5 ? cc.kt[(f - 5) >> 1] : 0.0f);
        uint32_t word = 0;
        for (int r5 = 0; r5 < 32; ++r5) {
            float v[kTsFn];
#pragma unroll
            for (int f = 0; f < kTsFn; ++f) v[f] = part[f] + ctab[(f * 3) * 32 + r5];
            word |= ts_cell_verdict(v, cc, ylimf) ? 0u : (1u << r5);
        }
        bits[w] = word;
    }
}
// share of the (dense) sample in ambiguous cells, every kCubeShareStep-th row of it; all threads, two barriers
template <int NT>
__device__ __forceinline__ int ts_cube_share(const uint32_t* samp, int n_sample, const float* ctab, const TsCubeConsts& cc, float ylimf, unsigned int* counter, int tid) {
    if (tid == 0) *counter = 0;
    __syncthreads();
    uint32_t amb = 0, seen = 0;
    for (int b = tid; b < n_sample; b += kCubeShareStep * NT) {
        amb += ts_cell_plain(ctab, cc, ylimf, as_global(samp)[b] & 0xffffffu) ? 0u : 1u;
        ++seen;
    }
    uint32_t both = amb | (seen << 16);
    for (int o = 32; o > 0; o >>= 1) both += (uint32_t)__shfl_xor((int)both, o, 64);
    if ((tid & 63) == 0 && both) atomicAdd(counter, both);
    __syncthreads();
    const uint32_t tot = *counter;
    return (tot >> 16) ? (int)(100u * (tot & 0xffffu) / (tot >> 16)) : 100;
}

// ------------------------------------------------------------------------------------------
// sweep 1 of the two-sweep schedule: exact moments AND the candidates of all four order statistics
// ------------------------------------------------------------------------------------------
// The per-trip structure of moments_sweep_b (binary32 burst sums of the same 16 pixels per lane and trip: the moments are bit-identical
// to every other schedule's) with the cube test of select_sweep_cube run on the trip's four chunks while the first table gather of the
// trip is in flight.  Pixels of ambiguous cells go to the wave's ring in LDS and are re-tested exactly two rows at a time (the drain);
// what the exact test flags goes to the tile's two candidate lists (RawDirect).  No sample bookkeeping: the sample exists already.
struct TsSweepConsts {      // wave-uniform (SGPRs): the exact test runs for the pixels of ambiguous cells only
    float gH[3], gL[3], n[3], k1;
    float W[2][3], kt[2], eps[2], zeta[2], thr[2];
};
template <bool ALIGNED, int kTrip, bool STREAM>
__device__ __forceinline__ void ts_sweep(const uint8_t* src, int P, int c0, int c1, int t, int nthreads, const TabReaderB& T, float ylimf,
                                         const TsSweepConsts& K, uint32_t bits_lds, uint32_t ring_lds, const RawDirect& out, Moments& mo,
                                         uint32_t& n_tissue) {
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
    // ---- the candidate side
    uint32_t rn = 0;                                             // ring fill, wave-uniform
    struct G3 { float2 r, g, b; };
    auto gather3 = [&](uint32_t q) { return G3{T.gam_odf(T.addr(q, 0)), T.gam_odf(T.addr(q, 1)), T.gam_odf(T.addr(q, 2))}; };
    struct M2 { unsigned long long a, c; };
    auto flags = [&](const G3& e) -> M2 {
        const bool tc = is_tissue_f(e.r.x, e.g.x, e.b.x, ylimf);
        const float z = fmaf(K.n[2], e.b.y, fmaf(K.n[1], e.g.y, K.n[0] * e.r.y));
        const float tH = fmaf(K.gH[2], e.b.y, fmaf(K.gH[1], e.g.y, K.gH[0] * e.r.y));
        const float tL = fmaf(K.gL[2], e.b.y, fmaf(K.gL[1], e.g.y, K.gL[0] * e.r.y));
        const bool pp = fmaf(-K.k1, fabsf(z), fminf(tH, tL)) > 0.0f;
        const float a1 = fmaf(K.W[0][2], e.b.y, fmaf(K.W[0][1], e.g.y, fmaf(K.W[0][0], e.r.y, K.kt[0])));
        const float a2 = fmaf(K.W[1][2], e.b.y, fmaf(K.W[1][1], e.g.y, fmaf(K.W[1][0], e.r.y, K.kt[1])));
        const float sa = fabsf(a1) + fabsf(a2);
        const bool g1 = fmaf(K.zeta[0], fabsf(z), fmaf(K.eps[0], sa, a1)) >= K.thr[0], g2 = fmaf(K.zeta[1], fabsf(z), fmaf(K.eps[1], sa, a2)) >= K.thr[1];
        return M2{__builtin_amdgcn_ballot_w64(tc) & ~__builtin_amdgcn_ballot_w64(pp), __builtin_amdgcn_ballot_w64(g1) | __builtin_amdgcn_ballot_w64(g2)};
    };
    auto ring_read = [&](uint32_t i) -> uint32_t {
#if defined(__HIP_DEVICE_COMPILE__)
        return *(SL_LDS const uint32_t*)(ring_lds + 4u * i);
#else
        return i;
#endif
    };
    auto drain = [&]() {
        while (rn >= 128u) {                                     // wave-uniform
            rn -= 128u;
            const uint32_t q0 = ring_read(rn + (uint32_t)lane), q1 = ring_read(rn + 64u + (uint32_t)lane);
            const G3 e0 = gather3(q0), e1 = gather3(q1);
            const M2 f0 = flags(e0), f1 = flags(e1);
            out.put2(f0.a, f0.c, q0, f1.a, f1.c, q1, lane);
        }
    };
    auto pixel_mask = [&](uint32_t p) -> bool {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t w = *(SL_LDS const uint32_t*)(((p >> 12) & 0xF80u) | (((p >> 9) & 0x7Cu) | bits_lds));
#else
        const uint32_t w = 0;
#endif
        return ((w >> cube_bit(p)) & 1u) != 0u;
    };
    auto cube = [&](auto tail_tag, const Chunk& ch, int cc) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        uint32_t p[4];
        bool amb[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            p[px] = chunk_pixel(ch, px);
            amb[px] = pixel_mask(px == 0 ? ch.w0 : p[px]);
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            unsigned long long m = __builtin_amdgcn_ballot_w64(amb[px]);
            if (TAIL) {
                const bool inb = (cc < c1) & (ALIGNED | ((size_t)cc * 4 + px < (size_t)P));
                m &= __builtin_amdgcn_ballot_w64(inb);
            }
            lds_append_masked(ring_lds, rn, m, p[px] & 0xffffffu);
            rn += (uint32_t)__popcll(m);
            if (px & 1) drain();
        }
    };
    Chunk cur[kTrip], nx[kTrip];
#pragma unroll
    for (int k = 0; k < kTrip; ++k) { cur[k] = fetch(w0 + lane + k * nthreads); nx[k] = fetch(w0 + lane + (kTrip + k) * nthreads); }
    auto trip = [&](auto tail_tag, int cb) {
        G g[2];
        g[0] = gather(cur[0]);
#pragma unroll
        for (int k = 0; k < kTrip; ++k) cube(tail_tag, cur[k], cb + k * nthreads + lane);
#pragma unroll
        for (int k = 0; k < kTrip; ++k) {
            if (k + 1 < kTrip) {
                g[(k + 1) & 1] = gather(cur[k + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < kTrip; ++j) { cur[j] = nx[j]; nx[j] = fetch(cb + lane + (2 * kTrip + j) * nthreads); }
            }
            __builtin_amdgcn_sched_barrier(0);
            compute(tail_tag, g[k & 1], cb + k * nthreads + lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        bm.flush(mo);
    };
    const int lim = ALIGNED ? c1 : min(c1, P >> 2);
    int cb = w0;
    for (; cb + (kTrip - 1) * nthreads + 64 <= lim; cb += nthreads * kTrip) trip(std::false_type{}, cb);
    if (cb < c1) trip(std::true_type{}, cb);
    {   // what is left in the ring: fewer than 128 entries (stale entries beyond the fill are valid pixels: read by every lane, masked out)
        const uint32_t q0 = ring_read((uint32_t)lane), q1 = ring_read(64u + (uint32_t)lane);
        const G3 e0 = gather3(q0 & 0xffffffu), e1 = gather3(q1 & 0xffffffu);
        const unsigned long long l0 = __builtin_amdgcn_ballot_w64((uint32_t)lane < rn), l1 = __builtin_amdgcn_ballot_w64(64u + (uint32_t)lane < rn);
        const M2 f0 = flags(e0), f1 = flags(e1);
        out.put2(f0.a & l0, f0.c & l0, q0, f1.a & l1, f1.c & l1, q1, lane);
        rn = 0;
    }
}

// ------------------------------------------------------------------------------------------
// the finish's check of the angular half-spaces against the exact eigenvectors (one whole wave; the result is valid in every lane)
// ------------------------------------------------------------------------------------------
// Vd: the exact V[c][k].  On success br = {lo0, hi0, lo1, hi1}: the brackets of the exact binary32 pseudo-angle keys -- every tissue
// pixel the sweep did not collect as an angular candidate has its key strictly between hi0 and lo1.
__device__ __forceinline__ bool ts_verify(const TwoSweep& ts, const double* Vd, float* br /*[4]*/, int lane, bool force_fail) {
    // exact unit normal n = v1 x v2, oriented like the sample's
    double n[3] = {Vd[2] * Vd[5] - Vd[4] * Vd[3], Vd[4] * Vd[1] - Vd[0] * Vd[5], Vd[0] * Vd[3] - Vd[2] * Vd[1]};
    const double nn = 1.0 / sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    const double sgn = (n[0] * ts.nd[0] + n[1] * ts.nd[1] + n[2] * ts.nd[2]) < 0.0 ? -nn : nn;
    double dn = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { n[c] *= sgn; dn = fmax(dn, fabs(n[c] - ts.nd[c])); }
    const double cH = fabs(n[0] * ts.gH[0] + n[1] * ts.gH[1] + n[2] * ts.gH[2]), cL = fabs(n[0] * ts.gL[0] + n[1] * ts.gL[1] + n[2] * ts.gL[2]);
    const double cmax = fmax(cH, cL);
    bool ok = (cmax <= ts.kappa1) & (cmax * dn <= ts.kappa2) & !force_fail;
    // the lines of the exact projection: q = V^T g.  H-type normal (-sin a, cos a): ray (q_y, -q_x); L-type (sin a, -cos a): ray (-q_y, q_x)
    const double qH[2] = {Vd[0] * ts.gH[0] + Vd[2] * ts.gH[1] + Vd[4] * ts.gH[2], Vd[1] * ts.gH[0] + Vd[3] * ts.gH[1] + Vd[5] * ts.gH[2]};
    const double qL[2] = {Vd[0] * ts.gL[0] + Vd[2] * ts.gL[1] + Vd[4] * ts.gL[2], Vd[1] * ts.gL[0] + Vd[3] * ts.gL[1] + Vd[5] * ts.gL[2]};
    const double rHx = qH[1], rHy = -qH[0], rLx = -qL[1], rLy = qL[0];
    ok = ok & (rHx > 1e-3) & (rLx > 1e-3);                       // both rays within +-90 degrees of the first eigenvector: no wrap of the key
    const double pH = rHy / (fabs(rHx) + fabs(rHy)), pL = rLy / (fabs(rLx) + fabs(rLy));
    ok = ok & (pH + 4.0 * (double)kAngleMargin < pL);
    // the outer ends: the sample's, carried over by the rotation inside the plane (ray of pseudo-angle p under V~ -> V^T V~ ray), padded;
    // they only decide how many members a bracket holds -- the counts verify them
    const double pad = 4.0 * ts.tau + 1e-3;
    double ends[2] = {(double)ts.lo0, (double)ts.hi1};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const double p = ends[e];
        if (p > -1.0 && p < 1.0) {
            const double dx = 1.0 - fabs(p), dy = p;
            double v3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) v3[c] = ts.Vd[2 * c] * dx + ts.Vd[2 * c + 1] * dy;
            const double x = Vd[0] * v3[0] + Vd[2] * v3[1] + Vd[4] * v3[2], y = Vd[1] * v3[0] + Vd[3] * v3[1] + Vd[5] * v3[2];
            ends[e] = x > 0.0 ? y / (x + fabs(y)) + (e == 0 ? -pad : pad) : (e == 0 ? -INFINITY : INFINITY);
        } else {
            ends[e] = e == 0 ? -INFINITY : INFINITY;
        }
    }
    br[0] = (float)ends[0];
    br[1] = (float)pH - kAngleMargin;
    br[2] = (float)pL + kAngleMargin;
    br[3] = (float)ends[1];
    if (!(br[0] < br[1])) br[0] = -INFINITY;
    if (!(br[3] > br[2])) br[3] = INFINITY;
    (void)lane;
    return ok;
}

}  // namespace sl
