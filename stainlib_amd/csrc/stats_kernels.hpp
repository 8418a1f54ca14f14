// stats_kernels.hpp -- per-tile statistics of the Macenko path and the fused transform (gfx950).
//
// Reference chain (macenko_stain_extractor.py:7-44, normalizer.py:34-36,45-50):
//   mask -> OD -> cov(3x3) -> eigh -> project -> arctan2 -> percentile(1,99) -> M
//   -> lasso concentrations of ALL pixels -> percentile(99) per stain -> rescale -> 255 exp(-C M_t)
//
// Four dependent streaming sweeps over the uint8 tile, nothing per-pixel stored in between:
//   sweep 1  moments   tissue test, 9 binary64 moment sums, one stratified-random sample pixel per
//                      `stride` pixels
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
#include "apply_kernels.hpp"

namespace sl {

constexpr int kMaxSample = 16384;   // samples per tile (<= P/64)
constexpr int kCapList = 16384;     // candidate capacity per bracket list
constexpr int kFinishThreads = 1024;
constexpr int kFusedThreads = 1024;
constexpr int kWaveStage = 512;     // per-wave LDS staging entries per list (multi-kernel select)
constexpr float kBracketZ = 6.0f;   // bracket half-width in standard deviations of the sample rank

struct TileState {
    // ---- after finish 1
    double n_tissue;
    double Vd[6];            // V[c][k], c = channel, k = 0 (largest eigenvalue), 1 (second)
    float Vf[6];
    float lo[2], hi[2];      // brackets of the current selection stage
    unsigned int lt[2];      // keys <  lo   (per list)
    unsigned int le[2];      // keys <= hi
    unsigned int ncand[2];   // appended candidates (may exceed kCapList => overflow)
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
    int parts;               // workgroups per tile (multi-kernel schedule)
    int stride_log2;         // sampling stride = 1 << stride_log2 (>= 6)
    int n_sample;            // ceil(P / stride)
    uint32_t y_lim;
    double lam;
    double pct;              // angular percentile
    double* partials;        // [tile][part][10]          (multi-kernel)
    uint32_t* sample;        // [tile][n_sample]          (multi-kernel)
    float* cand;             // [tile or workgroup][2][kCapList]
    TileState* state;        // [tile]                    (multi-kernel)
};

// One 16-byte LDS entry per byte value: everything a sweep needs from one ds_read_b128.
struct __attribute__((aligned(16))) TabEntry { double od; uint32_t gamma; float odf; };

__device__ __forceinline__ void fill_tab(TabEntry* s_tab) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        TabEntry e; e.od = d_od_f64[i]; e.gamma = d_gamma[i]; e.odf = d_od_f32[i];
        s_tab[i] = e;
    }
}

// Which pixel of sampling block b is kept (same function in the sweep and in the finish steps).
__device__ __forceinline__ uint32_t sample_offset(uint32_t b, int stride_log2) {
    uint32_t h = b * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    return h >> (32 - stride_log2);
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

// numpy.percentile(method='linear') position for n values (numpy/lib/function_base.py _quantile):
// virtual index = n*q + (alpha + q*(1-alpha-beta)) - 1 with alpha = beta = 1.
__device__ inline void percentile_pos(double n, double pct, long long& k, double& g) {
    const double q = pct / 100.0;
    double vi = n * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;
    if (vi < 0) vi = 0;
    if (vi > n - 1) vi = n - 1;
    const double f = floor(vi);
    k = (long long)f;
    g = vi - f;
}
__device__ inline double np_lerp(double a, double b, double t) {
    const double d = b - a;
    return t >= 0.5 ? b - d * (1.0 - t) : a + d * t;
}

// ------------------------------------------------------------------------------------------
// workgroup-level exact selection (any blockDim that is a multiple of 64)
// ------------------------------------------------------------------------------------------
struct SelScratch {
    uint32_t hist[1024];
    uint32_t misc[16];
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

// Exact 0-based k-th smallest of the n keys key_at(i) (NaN = absent).  count_le = #keys <= result,
// n_valid = #non-NaN keys.  Narrowing windows in the ordered-integer domain: each pass histograms
// the live window into <= 1024 bins, so a pass contends on LDS atomics only under real ties.
template <class KeyAt>
__device__ float wg_select(int n, KeyAt key_at, uint32_t k, uint32_t& count_le, uint32_t& n_valid, SelScratch& S) {
    // pass 0: window = [min, max]
    if (threadIdx.x == 0) { S.misc[4] = 0xffffffffu; S.misc[5] = 0; S.misc[6] = 0; }
    __syncthreads();
    {
        uint32_t mn = 0xffffffffu, mx = 0, cnt = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float f = key_at(i);
            if (f == f) { const uint32_t o = f2ord(f); mn = min(mn, o); mx = max(mx, o); ++cnt; }
        }
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
    if (n_valid == 0) { count_le = 0; return nan_f(); }
    if (k >= n_valid) k = n_valid - 1;
    uint32_t below = 0, in_win = n_valid;
    for (int guard = 0; guard < 8; ++guard) {
        const uint32_t R = whi - wlo;
        if (R == 0) break;
        const int s = R < 1024u ? 0 : (32 - __clz(R) - 10);
        const int nb = (int)(R >> s) + 1;
        for (int i = threadIdx.x; i < nb; i += blockDim.x) S.hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float f = key_at(i);
            if (f == f) {
                const uint32_t o = f2ord(f);
                if (o >= wlo && o <= whi) atomicAdd(&S.hist[(o - wlo) >> s], 1u);
            }
        }
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
    return ord2f(wlo);
}

// smallest key strictly greater than v (v itself if none)
template <class KeyAt>
__device__ float wg_next_above(int n, KeyAt key_at, float v, SelScratch& S) {
    if (threadIdx.x == 0) S.misc[7] = 0xffffffffu;
    __syncthreads();
    uint32_t best = 0xffffffffu;
    const uint32_t ov = f2ord(v);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float f = key_at(i);
        if (f == f) {
            const uint32_t o = f2ord(f);
            if (o > ov) best = min(best, o);
        }
    }
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
    uint32_t cle, nv;
    xa = wg_select(n, key_at, k, cle, nv, S);
    xb = (k + 1 < cle || k + 1 >= nv) ? xa : wg_next_above(n, key_at, xa, S);
}

// number of non-NaN keys
template <class KeyAt>
__device__ uint32_t wg_count_valid(int n, KeyAt key_at, SelScratch& S) {
    if (threadIdx.x == 0) S.misc[8] = 0;
    __syncthreads();
    uint32_t cnt = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float f = key_at(i);
        cnt += (f == f) ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor((int)cnt, o, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&S.misc[8], cnt);
    __syncthreads();
    const uint32_t r = S.misc[8];
    __syncthreads();
    return r;
}

// Bracket [lo, hi] around rank pct/100*(nv-1) of the nv valid sample keys: exact sample order
// statistics at rank -/+ z sigma; an end opens to -inf/+inf when its rank leaves the sample.
template <class KeyAt>
__device__ void wg_sample_bracket(int n, KeyAt key_at, uint32_t nv, double pct, float& lo, float& hi, SelScratch& S) {
    if (nv == 0) { lo = -INFINITY; hi = INFINITY; return; }
    uint32_t cle, nv2;
    const double q = pct / 100.0;
    const double r = q * ((double)nv - 1.0);
    const double sd = sqrt(fmax(q * (1.0 - q) * (double)nv, 0.0));
    const long long rlo = (long long)floor(r - kBracketZ * sd) - 1;
    const long long rhi = (long long)ceil(r + kBracketZ * sd) + 1;
    lo = rlo < 0 ? -INFINITY : wg_select(n, key_at, (uint32_t)rlo, cle, nv2, S);
    hi = rhi > (long long)nv - 1 ? INFINITY : wg_select(n, key_at, (uint32_t)rhi, cle, nv2, S);
}

// ------------------------------------------------------------------------------------------
// finish-step arithmetic (thread 0)
// ------------------------------------------------------------------------------------------
__device__ inline void jacobi_eigh3(double A[3][3], double w[3], double V[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        const double dia = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * dia) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int r = 0; r < 3; ++r) {       // A <- A J
                    const double arp = A[r][p], arq = A[r][q];
                    A[r][p] = c * arp - s * arq;
                    A[r][q] = s * arp + c * arq;
                }
                for (int r = 0; r < 3; ++r) {       // A <- J^T A
                    const double apr = A[p][r], aqr = A[q][r];
                    A[p][r] = c * apr - s * aqr;
                    A[q][r] = s * apr + c * aqr;
                }
                for (int r = 0; r < 3; ++r) {       // V <- V J
                    const double vrp = V[r][p], vrq = V[r][q];
                    V[r][p] = c * vrp - s * vrq;
                    V[r][q] = s * vrp + c * vrq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

// sums = {n, Sx, Sy, Sz, Sxx, Sxy, Sxz, Syy, Syz, Szz} -> status, V (binary64 + binary32)
__device__ inline int eigvecs_from_moments(const double* sum, double* Vd, float* Vf) {
    const double n = sum[0];
    int status = SL_TILE_OK;
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, w[3] = {0, 0, 0};
    if (n < 1) status = SL_TILE_EMPTY_MASK;
    else if (n < 2) status = SL_TILE_DEGENERATE_COV;
    else {
        // np.cov(OD, rowvar=False): (sum xx^T - n mean mean^T) / (n - 1)   (macenko_stain_extractor.py:22)
        const double m[3] = {sum[1] / n, sum[2] / n, sum[3] / n};
        double C[3][3];
        C[0][0] = sum[4] - n * m[0] * m[0]; C[0][1] = sum[5] - n * m[0] * m[1]; C[0][2] = sum[6] - n * m[0] * m[2];
        C[1][1] = sum[7] - n * m[1] * m[1]; C[1][2] = sum[8] - n * m[1] * m[2]; C[2][2] = sum[9] - n * m[2] * m[2];
        C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) C[i][j] /= (n - 1.0);
        jacobi_eigh3(C, w, V);
    }
    // eigh is ascending; the reference takes columns [2, 1] = largest, second largest (:24)
    int o[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (w[o[j]] > w[o[j + 1]]) { const int t = o[j]; o[j] = o[j + 1]; o[j + 1] = t; }
    const int sel[2] = {o[2], o[1]};
    for (int k = 0; k < 2; ++k) {
        const double sgn = V[0][sel[k]] < 0 ? -1.0 : 1.0;          // :26-27
        for (int c = 0; c < 3; ++c) {
            Vd[c * 2 + k] = sgn * V[c][sel[k]];
            Vf[c * 2 + k] = (float)(sgn * V[c][sel[k]]);
        }
    }
    return status;
}

// pseudo-angle order statistics -> stain matrix (macenko_stain_extractor.py:33-44)
__device__ inline void stain_matrix_from_angles(const double* Vd, const float* xs /*[4]*/, const double* gfrac, double* M) {
    const double minPhi = np_lerp(angle_of_pseudo((double)xs[0]), angle_of_pseudo((double)xs[1]), gfrac[0]);
    const double maxPhi = np_lerp(angle_of_pseudo((double)xs[2]), angle_of_pseudo((double)xs[3]), gfrac[1]);
    double v1[3], v2[3];
    for (int c = 0; c < 3; ++c) {                         // :36-37
        v1[c] = Vd[c * 2] * cos(minPhi) + Vd[c * 2 + 1] * sin(minPhi);
        v2[c] = Vd[c * 2] * cos(maxPhi) + Vd[c * 2 + 1] * sin(maxPhi);
    }
    const double* h = v1[0] > v2[0] ? v1 : v2;            // :40-43
    const double* e = v1[0] > v2[0] ? v2 : v1;
    const double nh = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
    const double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    for (int c = 0; c < 3; ++c) { M[c] = h[c] / nh; M[3 + c] = e[c] / ne; }   // :44
}

// ------------------------------------------------------------------------------------------
// per-pixel bodies shared by both schedules
// ------------------------------------------------------------------------------------------
struct Moments {
    double sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    uint32_t cnt = 0;
    __device__ __forceinline__ void add(double x, double y, double z) {
        cnt += 1;
        sx += x; sy += y; sz += z;
        sxx = fma(x, x, sxx); sxy = fma(x, y, sxy); sxz = fma(x, z, sxz);
        syy = fma(y, y, syy); syz = fma(y, z, syz); szz = fma(z, z, szz);
    }
    __device__ __forceinline__ void to_array(double* v) const {
        v[0] = (double)cnt; v[1] = sx; v[2] = sy; v[3] = sz; v[4] = sxx; v[5] = sxy; v[6] = sxz;
        v[7] = syy; v[8] = syz; v[9] = szz;
    }
};

// sweep 1 over chunks [c0, c1) with `nthreads` cooperating threads (thread index t)
template <bool ALIGNED, class SampleStore>
__device__ __forceinline__ void moments_sweep(const uint8_t* src, int P, int c0, int c1, int t, int nthreads,
                                              const TabEntry* s_tab, uint32_t y_lim, int stride_log2,
                                              SampleStore store_sample, Moments& mo) {
    const size_t nbytes = (size_t)P * 3;
    const int cps_log2 = stride_log2 - 2;          // chunks per sampling block
    for (int c = c0 + t; c < c1; c += nthreads * 2) {
        Chunk in[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cc = c + u * nthreads;
            in[u] = cc < c1 ? load_chunk<ALIGNED>(src, nbytes, cc) : Chunk{0xffffffffu, 0xffffffffu, 0xffffffffu};
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cc = c + u * nthreads;
            const bool live = cc < c1;
            const uint32_t b = (uint32_t)cc >> cps_log2;        // stratified sample: block b keeps pixel b*stride+off
            const uint32_t off = sample_offset(b, stride_log2);
            const bool has_sample = live && ((off >> 2) == ((uint32_t)cc & ((1u << cps_log2) - 1)));
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const uint32_t r = chunk_byte(in[u], 3 * px), g = chunk_byte(in[u], 3 * px + 1),
                               bb = chunk_byte(in[u], 3 * px + 2);
                const TabEntry er = s_tab[r], eg = s_tab[g], eb = s_tab[bb];
                const bool inb = live && (ALIGNED || (size_t)cc * 4 + px < (size_t)P);
                const bool tissue = inb && is_tissue(er.gamma, eg.gamma, eb.gamma, y_lim);
                if (tissue) mo.add(er.od, eg.od, eb.od);
                if (has_sample && (off & 3) == (uint32_t)px && inb)
                    store_sample(b, r | (g << 8) | (bb << 16) | ((tissue ? 1u : 0u) << 24));
            }
        }
    }
}

enum { kStageAngle = 0, kStageConc = 1 };

struct SelConsts {
    float V[6];
    LassoK L;
    float lo0, hi0, lo1, hi1;
};

// sweeps 2/3 over chunks [c0, c1): wave-uniform trip count so that ballots see every lane.
// Sink: push(list, flag, key, lane) collects bracket members.
template <int STAGE, bool ALIGNED, class Sink>
__device__ __forceinline__ void select_sweep(const uint8_t* src, int P, int c0, int c1, int t, int nthreads,
                                             const TabEntry* s_tab, uint32_t y_lim, const SelConsts& K, Sink& sink,
                                             uint32_t (&cnt)[4]) {
    const size_t nbytes = (size_t)P * 3;
    const int lane = t & 63;
    for (int cb = c0 + (t & ~63); cb < c1; cb += nthreads * 2) {
        Chunk in[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cc = cb + lane + u * nthreads;
            in[u] = cc < c1 ? load_chunk<ALIGNED>(src, nbytes, cc) : Chunk{0xffffffffu, 0xffffffffu, 0xffffffffu};
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cc = cb + lane + u * nthreads;
            const bool live = cc < c1;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const uint32_t r = chunk_byte(in[u], 3 * px), g = chunk_byte(in[u], 3 * px + 1),
                               bb = chunk_byte(in[u], 3 * px + 2);
                const TabEntry er = s_tab[r], eg = s_tab[g], eb = s_tab[bb];
                const bool inb = live && (ALIGNED || (size_t)cc * 4 + px < (size_t)P);
                float k0, k1;
                bool valid;
                if (STAGE == kStageAngle) {
                    valid = inb && is_tissue(er.gamma, eg.gamma, eb.gamma, y_lim);
                    k0 = k1 = angle_key(K.V, er.odf, eg.odf, eb.odf);
                } else {
                    valid = inb;
                    lasso2(K.L, er.odf, eg.odf, eb.odf, k0, k1);
                }
                const bool b_lt0 = valid & (k0 < K.lo0), b_le0 = valid & (k0 <= K.hi0);
                const bool b_lt1 = valid & (k1 < K.lo1), b_le1 = valid & (k1 <= K.hi1);
                cnt[0] += __popcll(__ballot(b_lt0));
                cnt[1] += __popcll(__ballot(b_le0));
                cnt[2] += __popcll(__ballot(b_lt1));
                cnt[3] += __popcll(__ballot(b_le1));
                sink.push(0, b_le0 & !b_lt0, k0, lane);
                sink.push(1, b_le1 & !b_lt1, k1, lane);
            }
            sink.after_chunk(lane);
        }
    }
}

// Exact order statistics (k, k+1) of one list of a selection stage: from the collected candidates
// when the bracket verified, else by exact selection over the whole tile (rare).
template <class TileKeyAt>
__device__ void stage_order_stats(const float* cand, float lo, float hi, uint32_t lt, uint32_t le, uint32_t nc,
                                  int P, TileKeyAt tile_key_at, uint32_t n, long long k, float& xa, float& xb,
                                  int& fallbacks, SelScratch& S) {
    const long long k2 = (k + 1 < (long long)n) ? k + 1 : k;
    const long long in = (long long)le - (long long)lt;
    const bool covered = (k >= (long long)lt) && (k2 < (long long)lt + in);
    if (covered && lo == hi) {               // every member of the bracket equals lo
        xa = xb = lo;
    } else if (covered && (long long)nc == in && nc <= (uint32_t)kCapList) {
        auto at = [&](int i) -> float { return cand[i]; };
        wg_select_pair((int)nc, at, (uint32_t)(k - lt), xa, xb, S);
        if (k2 == k) xb = xa;
    } else {                                 // exact, slow, rare
        wg_select_pair(P, tile_key_at, (uint32_t)k, xa, xb, S);
        if (k2 == k) xb = xa;
        fallbacks += 1;
    }
}

// ------------------------------------------------------------------------------------------
// multi-kernel schedule
// ------------------------------------------------------------------------------------------
template <bool ALIGNED>
static __global__ __launch_bounds__(kWG) void k_moments(StatsArgs a) {
    __shared__ TabEntry s_tab[256];
    __shared__ double s_red[kWG / 64][10];
    fill_tab(s_tab);
    __syncthreads();
    const int tile = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
    const int tid = threadIdx.x;
    const uint8_t* src = a.rgb + (size_t)tile * a.P * 3;
    uint32_t* samp = a.sample + (size_t)tile * a.n_sample;
    const int nch = (a.P + 3) >> 2;
    const int span = (nch + a.parts - 1) / a.parts;
    const int c0 = part * span, c1 = min(nch, c0 + span);
    Moments mo;
    auto store = [&](uint32_t b, uint32_t v) { samp[b] = v; };
    moments_sweep<ALIGNED>(src, a.P, c0, c1, tid, kWG, s_tab, a.y_lim, a.stride_log2, store, mo);
    double v[10];
    mo.to_array(v);
#pragma unroll
    for (int i = 0; i < 10; ++i) v[i] = wave_sum(v[i]);
    if ((tid & 63) == 0)
        for (int i = 0; i < 10; ++i) s_red[tid >> 6][i] = v[i];
    __syncthreads();
    if (tid < 10) {
        double t = 0;
        for (int w = 0; w < kWG / 64; ++w) t += s_red[w][tid];
        a.partials[((size_t)tile * a.parts + part) * 10 + tid] = t;
    }
}

static __global__ __launch_bounds__(kFinishThreads) void k_finish_moments(StatsArgs a) {
    __shared__ SelScratch S;
    __shared__ double s_sum[10];
    __shared__ float s_V[6];
    const int tile = blockIdx.x, tid = threadIdx.x;
    TileState& st = a.state[tile];
    if (tid < 10) {                                   // fixed order => run-to-run identical sums
        double t = 0;
        for (int p = 0; p < a.parts; ++p) t += a.partials[((size_t)tile * a.parts + p) * 10 + tid];
        s_sum[tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        double Vd[6];
        float Vf[6];
        st.status = eigvecs_from_moments(s_sum, Vd, Vf);
        for (int i = 0; i < 6; ++i) { st.Vd[i] = Vd[i]; st.Vf[i] = Vf[i]; s_V[i] = Vf[i]; }
        st.n_tissue = s_sum[0];
        st.fallbacks = 0;
        for (int i = 0; i < 2; ++i) { st.lt[i] = 0; st.le[i] = 0; st.ncand[i] = 0; }
    }
    __syncthreads();
    const uint32_t* samp = a.sample + (size_t)tile * a.n_sample;
    float Vr[6];
    for (int i = 0; i < 6; ++i) Vr[i] = s_V[i];
    auto key_at = [&](int b) -> float {
        const long long pix = ((long long)b << a.stride_log2) + sample_offset(b, a.stride_log2);
        if (pix >= a.P) return nan_f();
        const uint32_t s = samp[b];
        if (!(s >> 24)) return nan_f();
        return angle_key(Vr, d_od_f32[s & 255u], d_od_f32[(s >> 8) & 255u], d_od_f32[(s >> 16) & 255u]);
    };
    float lo0, hi0, lo1, hi1;
    const uint32_t nv = wg_count_valid(a.n_sample, key_at, S);
    wg_sample_bracket(a.n_sample, key_at, nv, 100.0 - a.pct, lo0, hi0, S);   // minPhi (:33)
    wg_sample_bracket(a.n_sample, key_at, nv, a.pct, lo1, hi1, S);           // maxPhi (:34)
    if (tid == 0) { st.lo[0] = lo0; st.hi[0] = hi0; st.lo[1] = lo1; st.hi[1] = hi1; }
}

struct WaveStageSink {              // per-wave candidate staging in LDS, flushed with ONE global atomic
    float* buf[2];
    uint32_t n[2];
    float* dst[2];
    unsigned int* counter[2];
    __device__ __forceinline__ void push(int li, bool flag, float v, int lane) {
        const unsigned long long m = __ballot(flag);
        if (m) {
            const uint32_t pos = n[li] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (flag) buf[li][pos] = v;
            n[li] += __popcll(m);
        }
    }
    __device__ __forceinline__ void flush(int li, int lane) {
        if (n[li] == 0) return;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(counter[li], n[li]);
        base = __builtin_amdgcn_readfirstlane(base);
        for (uint32_t i = lane; i < n[li]; i += 64)
            if (base + i < (uint32_t)kCapList) dst[li][base + i] = buf[li][i];
        n[li] = 0;
    }
    // a chunk adds at most 4*64 entries per list: keep that much room (wave-uniform test)
    __device__ __forceinline__ void after_chunk(int lane) {
        if (n[0] > kWaveStage - 256) flush(0, lane);
        if (n[1] > kWaveStage - 256) flush(1, lane);
    }
};

template <int STAGE, bool ALIGNED>
static __global__ __launch_bounds__(kWG) void k_select(StatsArgs a) {
    __shared__ TabEntry s_tab[256];
    __shared__ float s_stage[2][kWG / 64][kWaveStage];
    fill_tab(s_tab);
    const int tile = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TileState& st = a.state[tile];
    if (st.status == SL_TILE_EMPTY_MASK || st.status == SL_TILE_DEGENERATE_COV) return;   // block-uniform
    SelConsts K;
    if (STAGE == kStageAngle) {
        for (int i = 0; i < 6; ++i) K.V[i] = uni(st.Vf[i]);
    } else {
        lasso_consts(st.M, a.lam, K.L);
        uni(K.L);
    }
    K.lo0 = uni(st.lo[0]); K.hi0 = uni(st.hi[0]); K.lo1 = uni(st.lo[1]); K.hi1 = uni(st.hi[1]);
    __syncthreads();
    const uint8_t* src = a.rgb + (size_t)tile * a.P * 3;
    const int nch = (a.P + 3) >> 2;
    const int span = (nch + a.parts - 1) / a.parts;
    const int c0 = part * span, c1 = min(nch, c0 + span);
    WaveStageSink sink;
    for (int li = 0; li < 2; ++li) {
        sink.buf[li] = s_stage[li][wave];
        sink.n[li] = 0;
        sink.dst[li] = a.cand + ((size_t)tile * 2 + li) * kCapList;
        sink.counter[li] = &st.ncand[li];
    }
    uint32_t cnt[4] = {0, 0, 0, 0};
    select_sweep<STAGE, ALIGNED>(src, a.P, c0, c1, tid, kWG, s_tab, a.y_lim, K, sink, cnt);
    sink.flush(0, lane);
    sink.flush(1, lane);
    if (lane == 0) {
        if (cnt[0]) atomicAdd(&st.lt[0], cnt[0]);
        if (cnt[1]) atomicAdd(&st.le[0], cnt[1]);
        if (cnt[2]) atomicAdd(&st.lt[1], cnt[2]);
        if (cnt[3]) atomicAdd(&st.le[1], cnt[3]);
    }
}

// key of pixel p of a tile for the exact fallback
struct AngleTileKey {
    const uint8_t* src; const float* V; uint32_t y_lim;
    __device__ __forceinline__ float operator()(int p) const {
        const uint32_t r = src[3 * (size_t)p], g = src[3 * (size_t)p + 1], b = src[3 * (size_t)p + 2];
        if (!is_tissue(d_gamma[r], d_gamma[g], d_gamma[b], y_lim)) return nan_f();
        return angle_key(V, d_od_f32[r], d_od_f32[g], d_od_f32[b]);
    }
};
struct ConcTileKey {
    const uint8_t* src; const LassoK* L; int col;
    __device__ __forceinline__ float operator()(int p) const {
        float c1, c2;
        lasso2(*L, d_od_f32[src[3 * (size_t)p]], d_od_f32[src[3 * (size_t)p + 1]], d_od_f32[src[3 * (size_t)p + 2]], c1, c2);
        return col == 0 ? c1 : c2;
    }
};

static __global__ __launch_bounds__(kFinishThreads) void k_finish_angle(StatsArgs a) {
    __shared__ SelScratch S;
    __shared__ float s_res[4];
    __shared__ LassoK s_L;
    const int tile = blockIdx.x, tid = threadIdx.x;
    TileState& st = a.state[tile];
    if (st.status == SL_TILE_EMPTY_MASK || st.status == SL_TILE_DEGENERATE_COV) {
        if (tid < 6) st.M[tid] = nan_d();
        return;
    }
    float Vr[6];
    for (int i = 0; i < 6; ++i) Vr[i] = st.Vf[i];
    const uint8_t* src = a.rgb + (size_t)tile * a.P * 3;
    const AngleTileKey tkey{src, Vr, a.y_lim};
    const uint32_t T = (uint32_t)st.n_tissue;
    long long k[2];
    double gfrac[2];
    percentile_pos((double)T, 100.0 - a.pct, k[0], gfrac[0]);
    percentile_pos((double)T, a.pct, k[1], gfrac[1]);
    int fallbacks = 0;
    for (int li = 0; li < 2; ++li) {
        float xa, xb;
        stage_order_stats(a.cand + ((size_t)tile * 2 + li) * kCapList, st.lo[li], st.hi[li], st.lt[li], st.le[li],
                          st.ncand[li], a.P, tkey, T, k[li], xa, xb, fallbacks, S);
        if (tid == 0) { s_res[2 * li] = xa; s_res[2 * li + 1] = xb; }
        __syncthreads();
    }
    if (tid == 0) {
        double M[6];
        stain_matrix_from_angles(st.Vd, s_res, gfrac, M);
        for (int i = 0; i < 6; ++i) st.M[i] = M[i];
        st.fallbacks += fallbacks;
        LassoK L;
        lasso_consts(M, a.lam, L);
        s_L = L;
        for (int i = 0; i < 2; ++i) { st.lt[i] = 0; st.le[i] = 0; st.ncand[i] = 0; }
    }
    __syncthreads();
    // brackets for np.percentile(C, 99, axis=0) from the sample (all pixels, tissue or not)
    const LassoK L = s_L;
    const uint32_t* samp = a.sample + (size_t)tile * a.n_sample;
    for (int col = 0; col < 2; ++col) {
        auto key_at = [&](int b) -> float {
            const long long pix = ((long long)b << a.stride_log2) + sample_offset(b, a.stride_log2);
            if (pix >= a.P) return nan_f();
            const uint32_t s = samp[b];
            float c1, c2;
            lasso2(L, d_od_f32[s & 255u], d_od_f32[(s >> 8) & 255u], d_od_f32[(s >> 16) & 255u], c1, c2);
            return col == 0 ? c1 : c2;
        };
        float lo, hi;
        const uint32_t nv = wg_count_valid(a.n_sample, key_at, S);
        wg_sample_bracket(a.n_sample, key_at, nv, 99.0, lo, hi, S);
        if (tid == 0) { st.lo[col] = lo; st.hi[col] = hi; }
    }
}

static __global__ __launch_bounds__(kFinishThreads) void k_finish_conc(StatsArgs a, double* M_out, double* maxC_out,
                                                                       int32_t* status_out, int tile0) {
    __shared__ SelScratch S;
    __shared__ float s_res[4];
    __shared__ LassoK s_L;
    const int tile = blockIdx.x, tid = threadIdx.x;
    TileState& st = a.state[tile];
    const bool bad = st.status == SL_TILE_EMPTY_MASK || st.status == SL_TILE_DEGENERATE_COV;
    if (!bad) {
        if (tid == 0) { LassoK L; lasso_consts(st.M, a.lam, L); s_L = L; }
        __syncthreads();
        long long k;
        double gfrac;
        percentile_pos((double)a.P, 99.0, k, gfrac);
        int fallbacks = 0;
        const uint8_t* src = a.rgb + (size_t)tile * a.P * 3;
        for (int col = 0; col < 2; ++col) {
            const ConcTileKey tkey{src, &s_L, col};
            float xa, xb;
            stage_order_stats(a.cand + ((size_t)tile * 2 + col) * kCapList, st.lo[col], st.hi[col], st.lt[col],
                              st.le[col], st.ncand[col], a.P, tkey, (uint32_t)a.P, k, xa, xb, fallbacks, S);
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
    uint32_t y_lim;
    double lam;
    double pct;
    const double* M_tgt;     // transform only
    const double* maxC_tgt;  // transform only
    float* cand;             // [gridDim.x][2][kCapList]
    double* M_out;           // [n_tiles][6]
    double* maxC_out;        // [n_tiles][2]
    int32_t* status_out;     // [n_tiles]
    int32_t* diag_out;       // [n_tiles] fallbacks (may be NULL)
};

struct LdsSink {                    // the tile belongs to this workgroup: the list heads live in LDS
    unsigned int* ncand;            // LDS [2]
    float* dst[2];                  // global
    __device__ __forceinline__ void push(int li, bool flag, float v, int lane) {
        const unsigned long long m = __ballot(flag);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&ncand[li], (unsigned int)__popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (flag && pos < (uint32_t)kCapList) dst[li][pos] = v;
        }
    }
    __device__ __forceinline__ void after_chunk(int) {}
};

struct FusedShared {
    TabEntry tab[256];
    uint32_t sample[kMaxSample];
    SelScratch S;
    double red[kFusedThreads / 64][10];
    double sum[10];
    double Vd[6];
    double M[6];
    double maxC[2];
    float Vf[6];
    float lo[2], hi[2];
    unsigned int lt[2], le[2], ncand[2];
    float res[4];
    LassoK L;
    int status;
};

template <bool TRANSFORM, bool ALIGNED>
static __global__ __launch_bounds__(kFusedThreads) void k_macenko_fused(FusedArgs a) {
    __shared__ FusedShared sh;
    const int tid = threadIdx.x, lane = tid & 63;
    fill_tab(sh.tab);
    __syncthreads();
    const int nch = (a.P + 3) >> 2;
    float* cand0 = a.cand + ((size_t)blockIdx.x * 2 + 0) * kCapList;
    float* cand1 = a.cand + ((size_t)blockIdx.x * 2 + 1) * kCapList;

    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const size_t nbytes = (size_t)a.P * 3;
        const uint8_t* src = a.rgb + (size_t)tile * nbytes;
        int fallbacks = 0;

        // ---------------- sweep 1: moments + sample (into LDS)
        {
            Moments mo;
            auto store = [&](uint32_t b, uint32_t v) { sh.sample[b] = v; };
            moments_sweep<ALIGNED>(src, a.P, 0, nch, tid, kFusedThreads, sh.tab, a.y_lim, a.stride_log2, store, mo);
            double v[10];
            mo.to_array(v);
#pragma unroll
            for (int i = 0; i < 10; ++i) v[i] = wave_sum(v[i]);
            if (lane == 0)
                for (int i = 0; i < 10; ++i) sh.red[tid >> 6][i] = v[i];
        }
        __syncthreads();
        if (tid < 10) {
            double t = 0;
            for (int w = 0; w < kFusedThreads / 64; ++w) t += sh.red[w][tid];
            sh.sum[tid] = t;
        }
        __syncthreads();
        // ---------------- finish 1
        if (tid == 0) {
            double Vd[6];
            float Vf[6];
            sh.status = eigvecs_from_moments(sh.sum, Vd, Vf);
            for (int i = 0; i < 6; ++i) { sh.Vd[i] = Vd[i]; sh.Vf[i] = Vf[i]; }
            for (int i = 0; i < 2; ++i) { sh.lt[i] = 0; sh.le[i] = 0; sh.ncand[i] = 0; }
        }
        __syncthreads();
        const bool bad = sh.status == SL_TILE_EMPTY_MASK || sh.status == SL_TILE_DEGENERATE_COV;   // block-uniform
        if (!bad) {
            SelConsts K;
            for (int i = 0; i < 6; ++i) K.V[i] = uni(sh.Vf[i]);
            {
                auto key_at = [&](int b) -> float {
                    const long long pix = ((long long)b << a.stride_log2) + sample_offset(b, a.stride_log2);
                    if (pix >= a.P) return nan_f();
                    const uint32_t s = sh.sample[b];
                    if (!(s >> 24)) return nan_f();
                    return angle_key(K.V, d_od_f32[s & 255u], d_od_f32[(s >> 8) & 255u], d_od_f32[(s >> 16) & 255u]);
                };
                const uint32_t nv = wg_count_valid(a.n_sample, key_at, sh.S);
                wg_sample_bracket(a.n_sample, key_at, nv, 100.0 - a.pct, K.lo0, K.hi0, sh.S);
                wg_sample_bracket(a.n_sample, key_at, nv, a.pct, K.lo1, K.hi1, sh.S);
            }
            // ---------------- sweep 2: angle select
            {
                LdsSink sink{sh.ncand, {cand0, cand1}};
                uint32_t cnt[4] = {0, 0, 0, 0};
                select_sweep<kStageAngle, ALIGNED>(src, a.P, 0, nch, tid, kFusedThreads, sh.tab, a.y_lim, K, sink, cnt);
                if (lane == 0) {
                    if (cnt[0]) atomicAdd(&sh.lt[0], cnt[0]);
                    if (cnt[1]) atomicAdd(&sh.le[0], cnt[1]);
                    if (cnt[2]) atomicAdd(&sh.lt[1], cnt[2]);
                    if (cnt[3]) atomicAdd(&sh.le[1], cnt[3]);
                }
            }
            __threadfence_block();
            __syncthreads();
            // ---------------- finish 2: exact angular percentiles -> M
            {
                const uint32_t T = (uint32_t)sh.sum[0];
                long long k[2];
                double gfrac[2];
                percentile_pos((double)T, 100.0 - a.pct, k[0], gfrac[0]);
                percentile_pos((double)T, a.pct, k[1], gfrac[1]);
                const AngleTileKey tkey{src, K.V, a.y_lim};
                const float los[2] = {K.lo0, K.lo1}, his[2] = {K.hi0, K.hi1};
                for (int li = 0; li < 2; ++li) {
                    float xa, xb;
                    stage_order_stats(li ? cand1 : cand0, los[li], his[li], sh.lt[li], sh.le[li], sh.ncand[li], a.P,
                                      tkey, T, k[li], xa, xb, fallbacks, sh.S);
                    if (tid == 0) { sh.res[2 * li] = xa; sh.res[2 * li + 1] = xb; }
                    __syncthreads();
                }
                if (tid == 0) {
                    double M[6];
                    stain_matrix_from_angles(sh.Vd, sh.res, gfrac, M);
                    for (int i = 0; i < 6; ++i) sh.M[i] = M[i];
                    LassoK L;
                    lasso_consts(M, a.lam, L);
                    sh.L = L;
                    for (int i = 0; i < 2; ++i) { sh.lt[i] = 0; sh.le[i] = 0; sh.ncand[i] = 0; }
                }
                __syncthreads();
            }
            K.L = sh.L;
            uni(K.L);
            {
                float lo[2], hi[2];
                for (int col = 0; col < 2; ++col) {
                    auto key_at = [&](int b) -> float {
                        const long long pix = ((long long)b << a.stride_log2) + sample_offset(b, a.stride_log2);
                        if (pix >= a.P) return nan_f();
                        const uint32_t s = sh.sample[b];
                        float c1, c2;
                        lasso2(K.L, d_od_f32[s & 255u], d_od_f32[(s >> 8) & 255u], d_od_f32[(s >> 16) & 255u], c1, c2);
                        return col == 0 ? c1 : c2;
                    };
                    const uint32_t nv = wg_count_valid(a.n_sample, key_at, sh.S);
                    wg_sample_bracket(a.n_sample, key_at, nv, 99.0, lo[col], hi[col], sh.S);
                }
                K.lo0 = lo[0]; K.hi0 = hi[0]; K.lo1 = lo[1]; K.hi1 = hi[1];
            }
            // ---------------- sweep 3: concentration select
            {
                LdsSink sink{sh.ncand, {cand0, cand1}};
                uint32_t cnt[4] = {0, 0, 0, 0};
                select_sweep<kStageConc, ALIGNED>(src, a.P, 0, nch, tid, kFusedThreads, sh.tab, a.y_lim, K, sink, cnt);
                if (lane == 0) {
                    if (cnt[0]) atomicAdd(&sh.lt[0], cnt[0]);
                    if (cnt[1]) atomicAdd(&sh.le[0], cnt[1]);
                    if (cnt[2]) atomicAdd(&sh.lt[1], cnt[2]);
                    if (cnt[3]) atomicAdd(&sh.le[1], cnt[3]);
                }
            }
            __threadfence_block();
            __syncthreads();
            // ---------------- finish 3: exact 99th percentiles -> maxC
            {
                long long k;
                double gfrac;
                percentile_pos((double)a.P, 99.0, k, gfrac);
                const float los[2] = {K.lo0, K.lo1}, his[2] = {K.hi0, K.hi1};
                for (int col = 0; col < 2; ++col) {
                    const ConcTileKey tkey{src, &sh.L, col};
                    float xa, xb;
                    stage_order_stats(col ? cand1 : cand0, los[col], his[col], sh.lt[col], sh.le[col], sh.ncand[col],
                                      a.P, tkey, (uint32_t)a.P, k, xa, xb, fallbacks, sh.S);
                    if (tid == 0) { sh.res[2 * col] = xa; sh.res[2 * col + 1] = xb; }
                    __syncthreads();
                }
                if (tid == 0) {
                    sh.maxC[0] = np_lerp((double)sh.res[0], (double)sh.res[1], gfrac);   // normalizer.py:36,47
                    sh.maxC[1] = np_lerp((double)sh.res[2], (double)sh.res[3], gfrac);
                    if (!(sh.maxC[0] > 0.0) || !(sh.maxC[1] > 0.0)) sh.status = SL_TILE_ZERO_MAXC;
                }
                __syncthreads();
            }
        } else if (tid == 0) {
            for (int i = 0; i < 6; ++i) sh.M[i] = nan_d();
            sh.maxC[0] = sh.maxC[1] = nan_d();
        }
        __syncthreads();
        if (tid < 6 && a.M_out) a.M_out[(size_t)tile * 6 + tid] = sh.M[tid];
        if (tid < 2 && a.maxC_out) a.maxC_out[(size_t)tile * 2 + tid] = sh.maxC[tid];
        if (tid == 0 && a.status_out) a.status_out[tile] = sh.status;
        if (tid == 0 && a.diag_out) a.diag_out[tile] = fallbacks;

        // ---------------- sweep 4: apply
        if (TRANSFORM) {
            uint8_t* dst = a.out + (size_t)tile * nbytes;
            if (bad) {
                for (int c = tid; c < nch; c += kFusedThreads) store_chunk<ALIGNED>(dst, nbytes, c, load_chunk<ALIGNED>(src, nbytes, c));
            } else {
                LassoK L = sh.L;
                uni(L);
                ReconK R;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const double ratio = a.maxC_tgt[i] / sh.maxC[i];                       // normalizer.py:48
#pragma unroll
                    for (int c = 0; c < 3; ++c) R.q[i][c] = uni((float)(-1.4426950408889634 * ratio * a.M_tgt[3 * i + c]));
                }
                for (int c = tid; c < nch; c += kFusedThreads * kU) {
                    Chunk in[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int cc = c + u * kFusedThreads;
                        in[u] = cc < nch ? load_chunk<ALIGNED>(src, nbytes, cc) : Chunk{0, 0, 0};
                    }
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int cc = c + u * kFusedThreads;
                        uint32_t ob[12];
#pragma unroll
                        for (int px = 0; px < 4; ++px) {
                            const float x = sh.tab[chunk_byte(in[u], 3 * px + 0)].odf;
                            const float y = sh.tab[chunk_byte(in[u], 3 * px + 1)].odf;
                            const float z = sh.tab[chunk_byte(in[u], 3 * px + 2)].odf;
                            float c1, c2, v[3];
                            lasso2(L, x, y, z, c1, c2);
                            recon_px<false>(R, c1, c2, v);
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) ob[3 * px + ch] = trunc_u8(v[ch]);
                        }
                        Chunk o;
                        o.w0 = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
                        o.w1 = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
                        o.w2 = ob[8] | (ob[9] << 8) | (ob[10] << 16) | (ob[11] << 24);
                        if (cc < nch) store_chunk<ALIGNED>(dst, nbytes, cc, o);
                    }
                }
            }
        }
        __syncthreads();     // sh.* is reused by the next tile
    }
}

}  // namespace sl
