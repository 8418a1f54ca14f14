// stats_kernels.hpp -- per-tile statistics of the Macenko path (gfx950).
//
// Reference chain (macenko_stain_extractor.py:7-44, normalizer.py:34-36,45-47):
//   mask -> OD -> cov(3x3) -> eigh -> project -> arctan2 -> percentile(1,99) -> M
//   -> lasso concentrations of ALL pixels -> percentile(99) per stain
//
// MI355X schedule: three streaming sweeps over the uint8 tile, each followed by a tiny
// one-workgroup-per-tile "finish" kernel; nothing per-pixel is stored.
//
//   k_moments  (sweep 1)  tissue test, 9 binary64 moment sums per lane -> fixed-order partials;
//                         also drops one stratified-random sample pixel per `stride` pixels.
//   k_finish_moments      partials -> cov -> Jacobi eigh -> V ; sample -> sort -> brackets that
//                         contain the 1st/99th angular order statistics with overwhelming
//                         probability.
//   k_select<ANGLE>  (sweep 2)  exact counts below/inside each bracket + the bracket members
//                         ("candidates", ~1% of the pixels) appended via per-wave LDS staging.
//   k_finish_angle        sort candidates -> EXACT order statistics k, k+1 -> numpy-style linear
//                         interpolation -> stain matrix M ; sample -> brackets for the 99th
//                         percentile of both concentration columns.
//   k_select<CONC>   (sweep 3)  same skeleton on the lasso concentrations of all pixels.
//   k_finish_conc         exact 99th percentiles -> maxC, status.
//
// If a bracket misses (probability ~1e-9 per tile) or overflows (heavy ties), the finish
// kernel falls back to an exact radix select over the whole tile done by that one workgroup:
// results never depend on the sampling, only the speed does.
//
// Exactness: order statistics are exact on the binary32 keys the sweeps compute (pseudo-angle,
// concentrations); the interpolation, trigonometry, eigen-decomposition and moment sums are
// binary64.  Bounds: HBM (sweeps re-read the tile from L2 / Infinity Cache when the host keeps
// the tile group resident); 3 B/px per sweep.
#pragma once
#include "../../include/stainlib_hip.h"
#include "apply_kernels.hpp"

namespace sl {

constexpr int kMaxSample = 16384;   // samples per tile (<= P/64)
constexpr int kCapList = 16384;     // candidate capacity per bracket list
constexpr int kFinishThreads = 1024;
constexpr int kWaveStage = 512;     // per-wave LDS staging entries per list
constexpr float kBracketZ = 6.0f;   // bracket half-width in standard deviations of the sample rank

struct TileState {
    // ---- after k_finish_moments
    double n_tissue;
    double Vd[6];            // V[c][k], c = channel, k = 0 (largest eigenvalue), 1 (second)
    float Vf[6];
    float lo[2], hi[2];      // brackets of the current selection stage
    unsigned int lt[2];      // keys <  lo   (per list)
    unsigned int le[2];      // keys <= hi
    unsigned int ncand[2];   // appended candidates (may exceed kCapList => overflow)
    // ---- after k_finish_angle
    double M[6];
    // ---- after k_finish_conc
    double maxC[2];
    int status;
    int fallbacks;           // how many order statistics needed the slow exact path (diagnostics)
};

struct StatsArgs {
    const uint8_t* rgb;      // first tile of the group
    int P;
    int parts;
    int stride_log2;         // sampling stride = 1 << stride_log2 (>= 6)
    int n_sample;            // ceil(P / stride)
    uint32_t y_lim;
    double lam;
    double pct;              // angular percentile
    double* partials;        // [tile][part][10]
    uint32_t* sample;        // [tile][n_sample]
    float* cand;             // [tile][2][kCapList]
    TileState* state;        // [tile]
};

// One 16-byte LDS entry per byte value: everything a sweep needs from one ds_read_b128.
struct __attribute__((aligned(16))) TabEntry { double od; uint32_t gamma; float odf; };

__device__ __forceinline__ void fill_tab(TabEntry* s_tab) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        TabEntry e; e.od = d_od_f64[i]; e.gamma = d_gamma[i]; e.odf = d_od_f32[i];
        s_tab[i] = e;
    }
}

// Which pixel of sampling block b is kept (same function in the sweep and in the finish kernels).
__device__ __forceinline__ uint32_t sample_offset(uint32_t b, int stride_log2) {
    uint32_t h = b * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    return h >> (32 - stride_log2);
}

// Monotone surrogate of arctan2(y, x) on (-pi, pi]: y/(|x|+|y|) in [-1,1] for x >= 0, mirrored
// to (1,2] / [-2,-1) for x < 0.  One v_rcp instead of an atan2f per pixel; arctan2 itself is
// evaluated in binary64 only for the two selected order statistics.
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

// ------------------------------------------------------------------------------------------
// sweep 1: moments + sample
// ------------------------------------------------------------------------------------------
template <bool ALIGNED>
static __global__ __launch_bounds__(kWG) void k_moments(StatsArgs a) {
    __shared__ TabEntry s_tab[256];
    __shared__ double s_red[kWG / 64][10];
    fill_tab(s_tab);
    __syncthreads();
    const int tile = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
    const int tid = threadIdx.x;
    const size_t nbytes = (size_t)a.P * 3;
    const uint8_t* src = a.rgb + (size_t)tile * nbytes;
    uint32_t* samp = a.sample + (size_t)tile * a.n_sample;
    const int nch = (a.P + 3) >> 2;
    const int span = (nch + a.parts - 1) / a.parts;
    const int c0 = part * span, c1 = min(nch, c0 + span);
    const int cps_log2 = a.stride_log2 - 2;          // chunks per sampling block

    double sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    uint32_t cnt = 0;
    for (int c = c0 + tid; c < c1; c += kWG * 2) {
        Chunk in[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cc = c + u * kWG;
            in[u] = cc < c1 ? load_chunk<ALIGNED>(src, nbytes, cc) : Chunk{0xffffffffu, 0xffffffffu, 0xffffffffu};
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cc = c + u * kWG;
            const bool live = cc < c1;
            // stratified sample: block b keeps pixel b*stride + off
            const uint32_t b = (uint32_t)cc >> cps_log2;
            const uint32_t off = sample_offset(b, a.stride_log2);
            const bool has_sample = live && ((off >> 2) == ((uint32_t)cc & ((1u << cps_log2) - 1)));
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const uint32_t r = chunk_byte(in[u], 3 * px), g = chunk_byte(in[u], 3 * px + 1),
                               bb = chunk_byte(in[u], 3 * px + 2);
                const TabEntry er = s_tab[r], eg = s_tab[g], eb = s_tab[bb];
                const bool inb = live && (ALIGNED || (size_t)cc * 4 + px < (size_t)a.P);
                const bool tissue = inb && is_tissue(er.gamma, eg.gamma, eb.gamma, a.y_lim);
                if (tissue) {
                    const double x = er.od, y = eg.od, z = eb.od;
                    cnt += 1;
                    sx += x; sy += y; sz += z;
                    sxx = fma(x, x, sxx); sxy = fma(x, y, sxy); sxz = fma(x, z, sxz);
                    syy = fma(y, y, syy); syz = fma(y, z, syz); szz = fma(z, z, szz);
                }
                if (has_sample && (off & 3) == (uint32_t)px && inb)
                    samp[b] = r | (g << 8) | (bb << 16) | ((tissue ? 1u : 0u) << 24);
            }
        }
    }
    double v[10] = {(double)cnt, sx, sy, sz, sxx, sxy, sxz, syy, syz, szz};
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

// ------------------------------------------------------------------------------------------
// helpers for the one-workgroup finish kernels
// ------------------------------------------------------------------------------------------
// In-LDS bitonic sort, ascending; n2 = power of two >= number of valid entries (rest = +inf).
__device__ inline void lds_bitonic_sort(float* s, int n2) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const float x = s[i], y = s[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { s[i] = y; s[l] = x; }
                }
            }
            __syncthreads();
        }
    }
}
__device__ __forceinline__ int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

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

// Bracket [lo,hi] around sample rank q*(ns-1) in the ascending sample s[0..ns).
__device__ inline void bracket_from_sample(const float* s, int ns, double pct, float& lo, float& hi) {
    if (ns <= 0) { lo = -INFINITY; hi = INFINITY; return; }
    const double q = pct / 100.0;
    const double r = q * (ns - 1);
    const double sd = sqrt(fmax(q * (1.0 - q) * ns, 0.0));
    const long long rlo = (long long)floor(r - kBracketZ * sd) - 1;
    const long long rhi = (long long)ceil(r + kBracketZ * sd) + 1;
    lo = rlo < 0 ? -INFINITY : s[rlo];
    hi = rhi > ns - 1 ? INFINITY : s[rhi];
}

// Exact k-th smallest (0-based) of key(pixel) over a tile by one workgroup: 4-pass radix select
// on the order-preserving 32-bit image of the binary32 key.  Pixels whose key is NaN are skipped
// (non-tissue in the angle stage).  Also returns how many keys are <= the result.
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

template <class KeyFn>
__device__ float wg_select_exact(const uint8_t* src, int P, KeyFn key, unsigned long long k,
                                 unsigned long long& count_le, uint32_t* s_hist /*[256]*/, uint32_t* s_misc /*[4]*/) {
    uint32_t prefix = 0, pmask = 0;
    unsigned long long below = 0;            // keys strictly below the current prefix range
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) s_hist[i] = 0;
        __syncthreads();
        for (int p = threadIdx.x; p < P; p += blockDim.x) {
            const float f = key(src[3 * (size_t)p], src[3 * (size_t)p + 1], src[3 * (size_t)p + 2]);
            if (f == f) {
                const uint32_t o = f2ord(f);
                if ((o & pmask) == prefix) atomicAdd(&s_hist[(o >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long acc = below;
            int b = 0;
            for (; b < 256; ++b) {
                if (acc + s_hist[b] > k) break;
                acc += s_hist[b];
            }
            if (b > 255) b = 255;
            s_misc[0] = (uint32_t)b;
            s_misc[1] = (uint32_t)(acc & 0xffffffffu);
            s_misc[2] = (uint32_t)(acc >> 32);
            s_misc[3] = s_hist[b];
        }
        __syncthreads();
        prefix |= s_misc[0] << shift;
        pmask |= 0xffu << shift;
        below = ((unsigned long long)s_misc[2] << 32) | s_misc[1];
        count_le = below + s_misc[3];
        __syncthreads();
    }
    return ord2f(prefix);
}

// smallest key strictly greater than v (or v itself if none)
template <class KeyFn>
__device__ float wg_next_above(const uint8_t* src, int P, KeyFn key, float v, uint32_t* s_misc) {
    if (threadIdx.x == 0) s_misc[0] = 0xffffffffu;
    __syncthreads();
    uint32_t best = 0xffffffffu;
    const uint32_t ov = f2ord(v);
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        const float f = key(src[3 * (size_t)p], src[3 * (size_t)p + 1], src[3 * (size_t)p + 2]);
        if (f == f) {
            const uint32_t o = f2ord(f);
            if (o > ov && o < best) best = o;
        }
    }
    atomicMin(&s_misc[0], best);
    __syncthreads();
    const uint32_t r = s_misc[0];
    __syncthreads();
    return r == 0xffffffffu ? v : ord2f(r);
}

// Order statistics k and k+1 of a selection stage for list `li`, from the sorted candidates when
// the bracket verified, else by the exact slow path.  Returns the two values in (xa, xb).
template <class KeyFn>
__device__ void stage_order_stats(const StatsArgs& a, int tile, int li, TileState& st, float* s_sort,
                                  uint32_t* s_hist, uint32_t* s_misc, KeyFn key, unsigned long long n,
                                  long long k, float& xa, float& xb, int& fallbacks) {
    const long long k2 = (k + 1 < (long long)n) ? k + 1 : k;
    const float lo = st.lo[li], hi = st.hi[li];
    const long long lt = st.lt[li], in = (long long)st.le[li] - (long long)st.lt[li];
    const long long nc = st.ncand[li];
    const bool covered = (k >= lt) && (k2 < lt + in);
    bool ok = false;
    if (covered && lo == hi) {               // every member of the bracket equals lo
        xa = xb = lo;
        ok = true;
    } else if (covered && nc == in && nc <= kCapList) {
        const float* cand = a.cand + ((size_t)tile * 2 + li) * kCapList;
        const int n2 = next_pow2((int)nc);
        for (int i = threadIdx.x; i < n2; i += blockDim.x) s_sort[i] = i < nc ? cand[i] : INFINITY;
        __syncthreads();
        lds_bitonic_sort(s_sort, n2);
        xa = s_sort[k - lt];
        xb = s_sort[k2 - lt];
        __syncthreads();
        ok = true;
    }
    if (!ok) {                               // exact, slow, rare
        const uint8_t* src = a.rgb + (size_t)tile * a.P * 3;
        unsigned long long cle = 0;
        xa = wg_select_exact(src, a.P, key, (unsigned long long)k, cle, s_hist, s_misc);
        xb = (k2 == k || (unsigned long long)k2 < cle) ? xa : wg_next_above(src, a.P, key, xa, s_misc);
        fallbacks += 1;
    }
}

// ------------------------------------------------------------------------------------------
// finish 1: partials -> covariance -> eigenvectors ; sample -> angle brackets
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

static __global__ __launch_bounds__(kFinishThreads) void k_finish_moments(StatsArgs a) {
    __shared__ float s_sort[kMaxSample];
    __shared__ double s_sum[10];
    __shared__ float s_V[6];
    __shared__ int s_n;
    const int tile = blockIdx.x, tid = threadIdx.x;
    TileState& st = a.state[tile];
    if (tid < 10) {                                   // fixed order => run-to-run identical sums
        double t = 0;
        for (int p = 0; p < a.parts; ++p) t += a.partials[((size_t)tile * a.parts + p) * 10 + tid];
        s_sum[tid] = t;
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (tid == 0) {
        const double n = s_sum[0];
        int status = SL_TILE_OK;
        double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, w[3] = {0, 0, 0};
        if (n < 1) status = SL_TILE_EMPTY_MASK;
        else if (n < 2) status = SL_TILE_DEGENERATE_COV;
        else {
            // np.cov(OD, rowvar=False): (sum xx^T - n mean mean^T) / (n - 1)
            const double m[3] = {s_sum[1] / n, s_sum[2] / n, s_sum[3] / n};
            double C[3][3];
            C[0][0] = s_sum[4] - n * m[0] * m[0]; C[0][1] = s_sum[5] - n * m[0] * m[1]; C[0][2] = s_sum[6] - n * m[0] * m[2];
            C[1][1] = s_sum[7] - n * m[1] * m[1]; C[1][2] = s_sum[8] - n * m[1] * m[2]; C[2][2] = s_sum[9] - n * m[2] * m[2];
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
                st.Vd[c * 2 + k] = sgn * V[c][sel[k]];
                st.Vf[c * 2 + k] = (float)(sgn * V[c][sel[k]]);
                s_V[c * 2 + k] = (float)(sgn * V[c][sel[k]]);
            }
        }
        st.n_tissue = n;
        st.status = status;
        st.fallbacks = 0;
        for (int i = 0; i < 2; ++i) { st.lt[i] = 0; st.le[i] = 0; st.ncand[i] = 0; }
    }
    __syncthreads();
    // pseudo-angles of the tissue samples
    const uint32_t* samp = a.sample + (size_t)tile * a.n_sample;
    float Vr[6];
    for (int i = 0; i < 6; ++i) Vr[i] = s_V[i];
    for (int b = tid; b < a.n_sample; b += blockDim.x) {
        const long long pix = ((long long)b << a.stride_log2) + sample_offset(b, a.stride_log2);
        if (pix >= a.P) continue;
        const uint32_t s = samp[b];
        if (!(s >> 24)) continue;
        const float kx = angle_key(Vr, d_od_f32[s & 255u], d_od_f32[(s >> 8) & 255u], d_od_f32[(s >> 16) & 255u]);
        s_sort[atomicAdd(&s_n, 1)] = kx;
    }
    __syncthreads();
    const int ns = s_n;
    const int n2 = next_pow2(max(ns, 1));
    for (int i = ns + tid; i < n2; i += blockDim.x) s_sort[i] = INFINITY;
    __syncthreads();
    lds_bitonic_sort(s_sort, n2);
    if (tid == 0) {
        bracket_from_sample(s_sort, ns, 100.0 - a.pct, st.lo[0], st.hi[0]);   // minPhi (:33)
        bracket_from_sample(s_sort, ns, a.pct, st.lo[1], st.hi[1]);           // maxPhi (:34)
    }
}

// ------------------------------------------------------------------------------------------
// sweeps 2 and 3: count + collect around the brackets
// ------------------------------------------------------------------------------------------
enum { kStageAngle = 0, kStageConc = 1 };

struct WaveStager {                 // per-wave candidate staging in LDS, flushed with ONE global atomic
    float* buf;                     // [kWaveStage]
    uint32_t n;
    __device__ __forceinline__ void push(bool flag, float v, int lane) {
        const unsigned long long m = __ballot(flag);
        if (m) {
            const uint32_t pos = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (flag) buf[pos] = v;
            n += __popcll(m);
        }
    }
    __device__ __forceinline__ void flush(float* dst, unsigned int* counter, int lane) {
        if (n == 0) return;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(counter, n);
        base = __builtin_amdgcn_readfirstlane(base);
        for (uint32_t i = lane; i < n; i += 64)
            if (base + i < (uint32_t)kCapList) dst[base + i] = buf[i];
        n = 0;
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
    if (st.status == SL_TILE_EMPTY_MASK || st.status == SL_TILE_DEGENERATE_COV) return;   // uniform
    float V[6];
    LassoK L;
    if (STAGE == kStageAngle) {
        for (int i = 0; i < 6; ++i) V[i] = uni(st.Vf[i]);
    } else {
        lasso_consts(st.M, a.lam, L);
        uni(L);
    }
    const float lo0 = uni(st.lo[0]), hi0 = uni(st.hi[0]), lo1 = uni(st.lo[1]), hi1 = uni(st.hi[1]);
    __syncthreads();

    const size_t nbytes = (size_t)a.P * 3;
    const uint8_t* src = a.rgb + (size_t)tile * nbytes;
    const int nch = (a.P + 3) >> 2;
    const int span = (nch + a.parts - 1) / a.parts;
    const int c0 = part * span, c1 = min(nch, c0 + span);
    WaveStager w0{s_stage[0][wave], 0}, w1{s_stage[1][wave], 0};
    float* cand0 = a.cand + ((size_t)tile * 2 + 0) * kCapList;
    float* cand1 = a.cand + ((size_t)tile * 2 + 1) * kCapList;
    uint32_t lt0 = 0, le0 = 0, lt1 = 0, le1 = 0;     // wave-uniform counters

    // the loop bound is made wave-uniform so that ballots see every lane
    const int c_first = c0 + (tid & ~63);
    for (int cb = c_first; cb < c1; cb += kWG * 2) {
        Chunk in[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cc = cb + lane + u * kWG;
            in[u] = cc < c1 ? load_chunk<ALIGNED>(src, nbytes, cc) : Chunk{0xffffffffu, 0xffffffffu, 0xffffffffu};
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cc = cb + lane + u * kWG;
            const bool live = cc < c1;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const uint32_t r = chunk_byte(in[u], 3 * px), g = chunk_byte(in[u], 3 * px + 1),
                               bb = chunk_byte(in[u], 3 * px + 2);
                const TabEntry er = s_tab[r], eg = s_tab[g], eb = s_tab[bb];
                const bool inb = live && (ALIGNED || (size_t)cc * 4 + px < (size_t)a.P);
                float k0, k1;
                bool valid;
                if (STAGE == kStageAngle) {
                    valid = inb && is_tissue(er.gamma, eg.gamma, eb.gamma, a.y_lim);
                    k0 = k1 = angle_key(V, er.odf, eg.odf, eb.odf);
                } else {
                    valid = inb;
                    lasso2(L, er.odf, eg.odf, eb.odf, k0, k1);
                }
                const bool b_lt0 = valid & (k0 < lo0), b_le0 = valid & (k0 <= hi0);
                const bool b_lt1 = valid & (k1 < lo1), b_le1 = valid & (k1 <= hi1);
                lt0 += __popcll(__ballot(b_lt0));
                le0 += __popcll(__ballot(b_le0));
                lt1 += __popcll(__ballot(b_lt1));
                le1 += __popcll(__ballot(b_le1));
                w0.push(b_le0 & !b_lt0, k0, lane);
                w1.push(b_le1 & !b_lt1, k1, lane);
            }
            // a chunk adds at most 4*64 entries per list: keep that much room (wave-uniform test)
            if (w0.n > kWaveStage - 256) w0.flush(cand0, &st.ncand[0], lane);
            if (w1.n > kWaveStage - 256) w1.flush(cand1, &st.ncand[1], lane);
        }
    }
    w0.flush(cand0, &st.ncand[0], lane);
    w1.flush(cand1, &st.ncand[1], lane);
    if (lane == 0) {
        if (lt0) atomicAdd(&st.lt[0], lt0);
        if (le0) atomicAdd(&st.le[0], le0);
        if (lt1) atomicAdd(&st.lt[1], lt1);
        if (le1) atomicAdd(&st.le[1], le1);
    }
}

// ------------------------------------------------------------------------------------------
// finish 2: exact angular percentiles -> stain matrix ; sample -> concentration brackets
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(kFinishThreads) void k_finish_angle(StatsArgs a) {
    __shared__ float s_sort[kCapList];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_misc[4];
    __shared__ float s_res[4];
    __shared__ LassoK s_L;
    const int tile = blockIdx.x, tid = threadIdx.x;
    TileState& st = a.state[tile];
    if (st.status == SL_TILE_EMPTY_MASK || st.status == SL_TILE_DEGENERATE_COV) {
        if (tid < 6) st.M[tid] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    float Vr[6];
    for (int i = 0; i < 6; ++i) Vr[i] = st.Vf[i];
    const uint32_t y_lim = a.y_lim;
    auto key = [&](uint32_t r, uint32_t g, uint32_t b) -> float {
        if (!is_tissue(d_gamma[r], d_gamma[g], d_gamma[b], y_lim)) return __uint_as_float(0x7fc00000u);
        return angle_key(Vr, d_od_f32[r], d_od_f32[g], d_od_f32[b]);
    };
    const unsigned long long T = (unsigned long long)st.n_tissue;
    long long k[2];
    double gfrac[2];
    percentile_pos((double)T, 100.0 - a.pct, k[0], gfrac[0]);
    percentile_pos((double)T, a.pct, k[1], gfrac[1]);
    int fallbacks = 0;
    float xa, xb;
    for (int li = 0; li < 2; ++li) {
        stage_order_stats(a, tile, li, st, s_sort, s_hist, s_misc, key, T, k[li], xa, xb, fallbacks);
        if (tid == 0) { s_res[2 * li] = xa; s_res[2 * li + 1] = xb; }
        __syncthreads();
    }
    if (tid == 0) {
        const double minPhi = np_lerp(angle_of_pseudo((double)s_res[0]), angle_of_pseudo((double)s_res[1]), gfrac[0]);
        const double maxPhi = np_lerp(angle_of_pseudo((double)s_res[2]), angle_of_pseudo((double)s_res[3]), gfrac[1]);
        double v1[3], v2[3];
        for (int c = 0; c < 3; ++c) {                         // :36-37
            v1[c] = st.Vd[c * 2] * cos(minPhi) + st.Vd[c * 2 + 1] * sin(minPhi);
            v2[c] = st.Vd[c * 2] * cos(maxPhi) + st.Vd[c * 2 + 1] * sin(maxPhi);
        }
        const double* h = v1[0] > v2[0] ? v1 : v2;            // :40-43
        const double* e = v1[0] > v2[0] ? v2 : v1;
        const double nh = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
        const double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        for (int c = 0; c < 3; ++c) { st.M[c] = h[c] / nh; st.M[3 + c] = e[c] / ne; }   // :44
        st.fallbacks += fallbacks;
        LassoK L;
        lasso_consts(st.M, a.lam, L);
        s_L = L;
        for (int i = 0; i < 2; ++i) { st.lt[i] = 0; st.le[i] = 0; st.ncand[i] = 0; }
    }
    __syncthreads();
    // brackets for np.percentile(C, 99, axis=0) from the sample (all pixels, tissue or not)
    const LassoK L = s_L;
    const uint32_t* samp = a.sample + (size_t)tile * a.n_sample;
    for (int col = 0; col < 2; ++col) {
        if (tid == 0) s_misc[0] = 0;
        __syncthreads();
        for (int b = tid; b < a.n_sample; b += blockDim.x) {
            const long long pix = ((long long)b << a.stride_log2) + sample_offset(b, a.stride_log2);
            if (pix >= a.P) continue;
            const uint32_t s = samp[b];
            float c1, c2;
            lasso2(L, d_od_f32[s & 255u], d_od_f32[(s >> 8) & 255u], d_od_f32[(s >> 16) & 255u], c1, c2);
            s_sort[atomicAdd(&s_misc[0], 1u)] = col == 0 ? c1 : c2;
        }
        __syncthreads();
        const int ns = (int)s_misc[0];
        const int n2 = next_pow2(max(ns, 1));
        for (int i = ns + tid; i < n2; i += blockDim.x) s_sort[i] = INFINITY;
        __syncthreads();
        lds_bitonic_sort(s_sort, n2);
        if (tid == 0) bracket_from_sample(s_sort, ns, 99.0, st.lo[col], st.hi[col]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// finish 3: exact 99th percentiles of the two concentration columns
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(kFinishThreads) void k_finish_conc(StatsArgs a, double* M_out, double* maxC_out,
                                                                int32_t* status_out, int tile0) {
    __shared__ float s_sort[kCapList];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_misc[4];
    __shared__ float s_res[4];
    const int tile = blockIdx.x, tid = threadIdx.x;
    TileState& st = a.state[tile];
    const bool bad = st.status == SL_TILE_EMPTY_MASK || st.status == SL_TILE_DEGENERATE_COV;
    if (!bad) {
        LassoK L;
        lasso_consts(st.M, a.lam, L);
        long long k;
        double gfrac;
        percentile_pos((double)a.P, 99.0, k, gfrac);
        int fallbacks = 0;
        for (int col = 0; col < 2; ++col) {
            auto key = [&](uint32_t r, uint32_t g, uint32_t b) -> float {
                float c1, c2;
                lasso2(L, d_od_f32[r], d_od_f32[g], d_od_f32[b], c1, c2);
                return col == 0 ? c1 : c2;
            };
            float xa, xb;
            stage_order_stats(a, tile, col, st, s_sort, s_hist, s_misc, key, (unsigned long long)a.P, k, xa, xb, fallbacks);
            if (tid == 0) { s_res[2 * col] = xa; s_res[2 * col + 1] = xb; }
            __syncthreads();
        }
        if (tid == 0) {
            st.maxC[0] = np_lerp((double)s_res[0], (double)s_res[1], gfrac);
            st.maxC[1] = np_lerp((double)s_res[2], (double)s_res[3], gfrac);
            st.fallbacks += fallbacks;
            if (!(st.maxC[0] > 0.0) || !(st.maxC[1] > 0.0)) st.status = SL_TILE_ZERO_MAXC;
        }
    } else if (tid == 0) {
        st.maxC[0] = st.maxC[1] = __longlong_as_double(0x7ff8000000000000LL);
    }
    __syncthreads();
    if (tid < 6 && M_out) M_out[(size_t)(tile0 + tile) * 6 + tid] = st.M[tid];
    if (tid < 2 && maxC_out) maxC_out[(size_t)(tile0 + tile) * 2 + tid] = st.maxC[tid];
    if (tid == 0 && status_out) status_out[tile0 + tile] = st.status;
}

}  // namespace sl
