// lab.hip -- the OpenCV 8-bit Lab family: ReinhardStainNormalizer (normalization/normalizer.py:54-94),
// LuminosityStandardizer (utils/stain_utils.py:50-67) and the LAB helpers (utils/stain_utils.py:146-194), plus
// convert_OD_to_RGB (utils/stain_utils.py:114-124).
//
// cv2.cvtColor(uint8, COLOR_RGB2LAB / COLOR_LAB2RGB) is OpenCV's integer RGB2Lab_b / Lab2RGBinteger: table lookups and
// 32-bit integer arithmetic, restated here from the published color_lab.cpp (tables: gen_tables.py).  Everything the
// reference computes AROUND those conversions is a function of 256-entry histograms:
//   np.percentile(I, 90) over all bytes            -> byte histogram                      (standardize_brightness)
//   cv2.meanStdDev of L8/2.55, a8-128, b8-128      -> histograms of the three Lab bytes   (get_mean_std)
//   np.percentile(L8, 95)                          -> histogram of L8                     (LuminosityStandardizer)
// and each per-pixel float expression (I*255.0/p; ((x-mean)*(tstd/std)+tmean)*2.55, clip, truncate; 255*L/p) has a
// uint8 argument, i.e. is a 256-entry table evaluated ONCE per tile in binary64 with the reference's operation order.
// So a tile costs two histogram sweeps (3 B/px read each) and one map sweep (3 B/px read + 3 B/px written); no per-pixel
// floating point at all.  Roofline: HBM; the LDS histogram atomics bound the two statistics sweeps in practice.
#include "apply_kernels.hpp"
#include "sl_host.hpp"

namespace sl {

static __device__ const uint16_t d_lab_cbrt[3072] = {SL_LAB_CBRT_VALUES};   // OpenCV LabCbrtTab_b
static __device__ const uint16_t d_lab_yf[512] = {SL_LAB_YF_VALUES};        // OpenCV LabToYF_b: (y, f(y)) per L8
static __device__ const uint8_t d_inv_gamma[4096] = {SL_INV_GAMMA_VALUES};  // OpenCV sRGBInvGammaTab_b

struct LabTabs {
    uint16_t gamma[256];
    uint16_t cbrt[3072];
    uint16_t yf[512];
    uint8_t invg[4096];
    __device__ __forceinline__ void fill() {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) gamma[i] = (uint16_t)d_gamma[i];
        for (int i = threadIdx.x; i < 3072; i += blockDim.x) cbrt[i] = d_lab_cbrt[i];
        for (int i = threadIdx.x; i < 512; i += blockDim.x) yf[i] = d_lab_yf[i];
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) invg[i] = d_inv_gamma[i];
    }
};

// the static tables a sweep needs, without the ones its per-tile tables (LabTileTabs) already contain
struct LabCbrt {              // RGB -> Lab8 behind a per-tile gamma table
    uint16_t cbrt[3072];
    __device__ __forceinline__ void fill() {
        for (int i = threadIdx.x; i < 3072 / 2; i += blockDim.x) ((uint32_t*)cbrt)[i] = ((const uint32_t*)d_lab_cbrt)[i];
    }
};
struct LabCbrtInv {           // ... and back through per-tile yf / a / b tables
    uint16_t cbrt[3072];
    uint8_t invg[4096];
    __device__ __forceinline__ void fill() {
        for (int i = threadIdx.x; i < 3072 / 2; i += blockDim.x) ((uint32_t*)cbrt)[i] = ((const uint32_t*)d_lab_cbrt)[i];
        for (int i = threadIdx.x; i < 4096 / 4; i += blockDim.x) ((uint32_t*)invg)[i] = ((const uint32_t*)d_inv_gamma)[i];
    }
};

// Compiler hazard (hipcc 7.2, gfx950): clamp(x >> n, 0, 255) of two values is selected as ONE v_ashr_pk_u8_i32, whose result the
// compiler then ORs into a word as if bits 31:16 were zero -- on the hardware they are not (found by the exhaustive Lab test:
// bytes 2 of every packed word came out with stray bits).  The empty asm keeps the shift and the clamp apart.
__device__ __forceinline__ int sat8(int v) {
    asm("" : "+v"(v));
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// OpenCV RGB2Lab_b::operator(): coefficients cvRound(4096 * sRGB2XYZ_D65[i][j] / whitePt[i]), lab_shift 12, lab_shift2 15.
// R, G, Bc are the gamma-table values of the three bytes (the sweeps read them from a per-tile table that already
// contains the brightness table: gamma[lut[v]]).
template <class T>
__device__ __forceinline__ void gamma_to_lab8(const T& t, int R, int G, int Bc, int& L, int& A, int& B) {
    // (24-bit multiplies: the gamma values stay below 2^11, the cube-root values below 2^16, every product below 2^31 -- the compiler
    //  cannot see the ranges behind the table reads and would issue twelve full-width multiplies at a quarter of the vector rate)
    const int fX = t.cbrt[(__mul24(R, 1777) + __mul24(G, 1541) + __mul24(Bc, 778) + 2048) >> 12];
    const int fY = t.cbrt[(__mul24(R, 871) + __mul24(G, 2929) + __mul24(Bc, 296) + 2048) >> 12];
    const int fZ = t.cbrt[(__mul24(R, 73) + __mul24(G, 448) + __mul24(Bc, 3575) + 2048) >> 12];
    L = sat8((__mul24(296, fY) - 1336934 + 16384) >> 15);
    A = sat8((__mul24(500, fX - fY) + 128 * 32768 + 16384) >> 15);
    B = sat8((__mul24(200, fY - fZ) + 128 * 32768 + 16384) >> 15);
}
__device__ __forceinline__ void rgb_to_lab8(const LabTabs& t, uint32_t r, uint32_t g, uint32_t b, int& L, int& A, int& B) {
    gamma_to_lab8(t, t.gamma[r], t.gamma[g], t.gamma[b], L, A, B);
}

// OpenCV abToXZ_b[i - minABvalue] evaluated instead of stored (36864 entries): C integer arithmetic, division truncates.
// Both branches are evaluated for every lane; the cubic one only counts for i > 3390, where every operand is positive and
// the two divisions by 16384 are plain shifts (as signed divisions they cost a sign fix-up each).
// The truncating division by 841 (a full-width multiply-high and a multiply-low at a quarter of the vector rate each, plus
// the sign fix-up) is done in binary32 instead: x = 108 i is exact there (|x| < 2^23), x * (1/841) is off by at most
// 1.3e-4 in the range the linear branch is used (|x / 841| < 1100), and exact quotients keep 1/841 = 1.19e-3 away from the
// integers they do not hit -- so truncating x/841 pushed half of 1/841 away from zero gives C's quotient for every x.
__device__ __forceinline__ int ab_to_xz(int i) {
    const float xf = (float)__mul24(i, 108);                  // (|i| < 2^17: a 24-bit multiply, the full-width one runs at quarter rate)
    const int lin = (int)fmaf(xf, 1.0f / 841.0f, copysignf(0.5f / 841.0f, xf)) - 290;     // (i * 108) / 841 - 290;  290 = BASE*16/116*108/841
    const uint32_t u = (uint32_t)i;                       // (i < 2^17, (i*i) >> 14 < 2^20: 24-bit multiplies, the full-width ones run at quarter rate)
    const int cub = (int)(__umul24(__umul24(u, u) >> 14, u) >> 14);
    return i <= 3390 ? lin : cub;
}

// OpenCV Lab2RGBinteger::process: coefficients cvRound(4096 * XYZ2sRGB_D65[i][j] * whitePt[j]), shift 14.
// lab_adiv / lab_bdiv: the a and b bytes on the table's scale (128*BASE/500 = 4194, 128*BASE/200 = 10485).
__device__ __forceinline__ int lab_adiv(int a) { return ((5 * a * 53687 + 128) >> 13) - 4194; }
__device__ __forceinline__ int lab_bdiv(int b) { return ((b * 41943 + 16) >> 9) - 10485 + 1; }
template <class T>
__device__ __forceinline__ void yf_to_rgb(const T& t, int y, int ify, int adiv, int bdiv, uint32_t& r, uint32_t& g, uint32_t& bl) {
    const int x = ab_to_xz(ify + adiv), z = ab_to_xz(ify - bdiv);
    // x, z < 2^17 (ab_to_xz of an argument below 2^15 + 2^14), y < 2^15: 24-bit multiplies (the compiler cannot see the ranges and
    // would issue full-width ones at a quarter of the rate); every sum stays below 2^31 as in OpenCV's int arithmetic
    int ro = (__mul24(12615, x) - __mul24(6296, y) - __mul24(2223, z) + 8192) >> 14;
    int go = (__mul24(-3773, x) + __mul24(7684, y) + __mul24(185, z) + 8192) >> 14;
    int bo = (__mul24(217, x) - __mul24(836, y) + __mul24(4715, z) + 8192) >> 14;
    ro = ro < 0 ? 0 : (ro > 4095 ? 4095 : ro);
    go = go < 0 ? 0 : (go > 4095 ? 4095 : go);
    bo = bo < 0 ? 0 : (bo > 4095 ? 4095 : bo);
    r = t.invg[ro]; g = t.invg[go]; bl = t.invg[bo];
}
__device__ __forceinline__ void lab8_to_rgb(const LabTabs& t, int L, int a, int b, uint32_t& r, uint32_t& g, uint32_t& bl) {
    yf_to_rgb(t, t.yf[2 * L], t.yf[2 * L + 1], lab_adiv(a), lab_bdiv(b), r, g, bl);
}

__device__ __forceinline__ Chunk pack12(const uint32_t (&ob)[12]) {
    Chunk o;
    o.w0 = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
    o.w1 = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
    o.w2 = ob[8] | (ob[9] << 8) | (ob[10] << 16) | (ob[11] << 24);
    return o;
}

// uint8(clip(x, 0, 255)) of numpy: clip, then truncate toward zero (NaN -> 0)
__device__ __forceinline__ uint32_t clip_trunc_u8(double x) { return (uint32_t)fmin(fmax(x, 0.0), 255.0); }

// Per-tile scratch (workspace): histograms as uint64 (3 * 2^30 bytes per tile overflow 32 bits)
struct LabScratch {
    unsigned long long bytes[256];      // all byte values of the tile
    unsigned long long lab_l[256];      // L8 of the (optionally brightness-standardised) tile
    unsigned long long ab[4];           // sum a8, sum a8^2, sum b8, sum b8^2 (a - 128 and b - 128 are affine in the byte: no histogram needed)
    unsigned long long tissue;          // pixels of the standardised tile passing the luminosity test
    unsigned long long pad_[3];
};

// np.percentile(values, pct) (linear interpolation) of the integer population described by a 256-bin histogram; one thread
__device__ inline double percentile_of_hist(const unsigned long long* hist, double pct) {
    unsigned long long n = 0;
    for (int v = 0; v < 256; ++v) n += hist[v];
    if (n == 0) return nan("");
    long long k;
    double g;
    percentile_pos((double)n, pct, k, g);
    const unsigned long long k2 = (unsigned long long)k + 1 < n ? (unsigned long long)k + 1 : (unsigned long long)k;
    int va = -1, vb = -1;
    unsigned long long cum = 0;
    for (int v = 0; v < 256; ++v) {
        cum += hist[v];
        if (va < 0 && cum > (unsigned long long)k) va = v;
        if (vb < 0 && cum > k2) { vb = v; break; }
    }
    return np_lerp((double)va, (double)vb, g);
}

constexpr int kLabWG = 256;

// ---- sweep A: histogram of all byte values ------------------------------------------------------------------------
// kHistCopies sub-histograms per wave, chosen by the lane (lane & 3): neighbouring lanes read neighbouring pixels, whose bytes are
// often equal, and lanes that hit one counter of one copy serialise (measured: ~13 cycles per wave instruction with one copy).
constexpr int kHistCopies = 4;
template <bool ALIGNED>
static __global__ __launch_bounds__(kLabWG) void k_byte_hist(const uint8_t* __restrict__ rgb, int P, int parts, LabScratch* __restrict__ sc) {
    __shared__ uint32_t s_h[kLabWG / 64][kHistCopies][256];
    for (int i = threadIdx.x; i < (kLabWG / 64) * kHistCopies * 256; i += kLabWG) (&s_h[0][0][0])[i] = 0;
    __syncthreads();
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts, wave = threadIdx.x >> 6;
    uint32_t* h = s_h[wave][threadIdx.x & (kHistCopies - 1)];
    const size_t nbytes = (size_t)P * 3;
    const uint8_t* src = rgb + (size_t)tile * nbytes;
    const int nch = (P + 3) >> 2;
    const int span = (nch + parts - 1) / parts;
    const int c0 = part * span, c1 = min(nch, c0 + span);
    for (int c = c0 + (int)threadIdx.x; c < c1; c += kLabWG) {
        const Chunk in = load_chunk<ALIGNED>(src, nbytes, c);
#pragma unroll
        for (int i = 0; i < 12; ++i)
            if (ALIGNED || (size_t)c * 12 + i < nbytes) atomicAdd(&h[chunk_byte(in, i)], 1u);
    }
    __syncthreads();
    const int v = threadIdx.x;
    unsigned long long t = 0;
    for (int w = 0; w < kLabWG / 64; ++w)
        for (int k = 0; k < kHistCopies; ++k) t += s_h[w][k][v];
    if (t) atomicAdd(&sc[tile].bytes[v], t);
}

// the brightness table of a tile: uint8(clip(v * 255.0 / p, 0, 255)) (stain_utils.py:193-194); identity when !standardize
__device__ __forceinline__ void fill_brightness_lut(uint8_t* lut, const LabScratch& sc, int standardize, double* s_p) {
    if (threadIdx.x == 0) *s_p = standardize ? percentile_of_hist(sc.bytes, 90.0) : nan("");
    __syncthreads();
    const double p = *s_p;
    for (int v = threadIdx.x; v < 256; v += blockDim.x)
        lut[v] = standardize ? (uint8_t)clip_trunc_u8((double)v * 255.0 / p) : (uint8_t)v;
    __syncthreads();
}

// largest L8 that still counts as tissue, +1:  L8 / 255.0 < threshold  (stain_utils.py:42-43)
__device__ __forceinline__ int l8_limit(double thr) {
    int lim = 0;
    for (int v = 0; v < 256; ++v)
        if ((double)v / 255.0 < thr) lim = v + 1;
    return lim;
}

// Per-tile composed tables (workspace, behind the LabScratch array): written ONCE per tile by k_lab_pre / k_lab_tables and loaded by every
// workgroup of the sweeps that follow.  (Round 5: until then every workgroup of k_lab_hist / k_lab_map rebuilt them -- a serial
// percentile over 256 global counters, three mean/std loops in binary64, five table passes -- before it touched a pixel: ~20 us of a
// workgroup's ~450, which also kept the sweeps from being cut into workgroups small enough to fill the last round.)
struct LabTileTabs {
    uint16_t g[256];          // byte -> gamma value, through the brightness table where the op standardises
    uint32_t yf[256];         // L8 -> (y, f(y)) of the MAPPED L byte, packed
    int ad[256], bd[256];     // a8 / b8 -> the mapped byte on abToXZ's scale (MODE 0)
    uint8_t lut[256];         // the brightness table (identity when the op does not standardise)
    double p;                 // the percentile the op reports (p90 of the bytes / the L percentile of MODE 1)
    double pad_;
};

// after sweep A: the brightness table and the gamma values behind it (what sweep B needs)
static __global__ __launch_bounds__(256) void k_lab_pre(const LabScratch* __restrict__ sc, int standardize, LabTileTabs* __restrict__ tt) {
    __shared__ uint8_t s_lut[256];
    __shared__ double s_p;
    const int tile = blockIdx.x, tid = threadIdx.x;
    fill_brightness_lut(s_lut, sc[tile], standardize, &s_p);
    tt[tile].lut[tid] = s_lut[tid];
    tt[tile].g[tid] = (uint16_t)d_gamma[s_lut[tid]];
    if (tid == 0) tt[tile].p = s_p;
}

// ---- sweep B: histograms of the Lab bytes of the (standardised) tile ------------------------------------------------
template <bool ALIGNED>
static __global__ __launch_bounds__(kLabWG) void k_lab_hist(const uint8_t* __restrict__ rgb, int P, int parts, int want_ab, double thr,
                                                            LabScratch* __restrict__ sc, const LabTileTabs* __restrict__ tt) {
    __shared__ LabCbrt s_t;
    __shared__ uint32_t s_h[kLabWG / 64][256];
    __shared__ uint16_t s_g[256];
    __shared__ unsigned long long s_tissue, s_ab[4];
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts, wave = threadIdx.x >> 6;
    s_t.fill();
    for (int i = threadIdx.x; i < (kLabWG / 64) * 256; i += kLabWG) (&s_h[0][0])[i] = 0;
    if (threadIdx.x == 0) s_tissue = 0;
    if (threadIdx.x < 4) s_ab[threadIdx.x] = 0;
    for (int v = threadIdx.x; v < 256; v += kLabWG) s_g[v] = tt[tile].g[v];     // brightness table and gamma table in one lookup (k_lab_pre)
    __syncthreads();
    const int lim = l8_limit(thr);
    const size_t nbytes = (size_t)P * 3;
    const uint8_t* src = rgb + (size_t)tile * nbytes;
    const int nch = (P + 3) >> 2;
    const int span = (nch + parts - 1) / parts;
    const int c0 = part * span, c1 = min(nch, c0 + span);
    uint32_t n_tissue = 0, sa = 0, sb = 0;                       // a thread sees <= 2^30 / 256 pixels: the plain sums fit 32 bits
    unsigned long long saa = 0, sbb = 0;
    for (int c = c0 + (int)threadIdx.x; c < c1; c += kLabWG) {
        const Chunk in = load_chunk<ALIGNED>(src, nbytes, c);
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            if (!ALIGNED && (size_t)c * 4 + px >= (size_t)P) break;
            int L, A, B;
            gamma_to_lab8(s_t, s_g[chunk_byte(in, 3 * px)], s_g[chunk_byte(in, 3 * px + 1)], s_g[chunk_byte(in, 3 * px + 2)], L, A, B);
            atomicAdd(&s_h[wave][L], 1u);
            sa += (uint32_t)A; saa += (uint32_t)(A * A); sb += (uint32_t)B; sbb += (uint32_t)(B * B);
            n_tissue += L < lim ? 1u : 0u;
        }
    }
    unsigned long long red[5] = {n_tissue, sa, saa, sb, sbb};
#pragma unroll
    for (int i = 0; i < 5; ++i) red[i] = wave_sum(red[i]);
    if ((threadIdx.x & 63) == 0) {
        if (red[0]) atomicAdd(&s_tissue, red[0]);
        if (want_ab) for (int i = 0; i < 4; ++i) atomicAdd(&s_ab[i], red[1 + i]);
    }
    __syncthreads();
    const int v = threadIdx.x;
    unsigned long long t = 0;
    for (int w = 0; w < kLabWG / 64; ++w) t += s_h[w][v];
    if (t) atomicAdd(&sc[tile].lab_l[v], t);
    if (want_ab && threadIdx.x < 4) atomicAdd(&sc[tile].ab[threadIdx.x], s_ab[threadIdx.x]);
    if (threadIdx.x == 0 && s_tissue) atomicAdd(&sc[tile].tissue, s_tissue);
}

// cv2.meanStdDev of a lab_split plane (stain_utils.py:153-157): sums in binary64, population variance clamped at 0.  One thread.
// L: value(v) = binary32 v / 2.55f as lab_split makes it, from the histogram of L8.  a, b: value = byte - 128, so the two
// sums follow exactly from the integer sums of the byte and its square (every term is an integer below 2^53).
__device__ inline void mean_std_of_scratch(const LabScratch& sc, int channel, double& mean, double& sd) {
    double n = 0, s1 = 0, s2 = 0;
    for (int v = 0; v < 256; ++v) {
        const double c = (double)sc.lab_l[v];
        n += c;
        if (channel == 0) {
            const double x = (double)((float)v / 2.55f);
            s1 += c * x; s2 += c * x * x;
        }
    }
    if (channel != 0) {
        const double sv = (double)sc.ab[2 * (channel - 1)], svv = (double)sc.ab[2 * (channel - 1) + 1];
        s1 = sv - 128.0 * n;
        s2 = svv - 256.0 * sv + 16384.0 * n;
    }
    mean = s1 / n;
    const double var = s2 / n - mean * mean;
    sd = sqrt(var > 0.0 ? var : 0.0);
}

// stats_out[tile] = {p90, mean L, a, b, std L, a, b, tissue}
static __global__ __launch_bounds__(64) void k_lab_stats(const LabScratch* __restrict__ sc, int standardize, double* __restrict__ stats_out) {
    const int tile = blockIdx.x, t = threadIdx.x;
    double* o = stats_out + 8 * (size_t)tile;
    if (t < 3) {
        double m, s;
        mean_std_of_scratch(sc[tile], t, m, s);
        o[1 + t] = m; o[4 + t] = s;
    } else if (t == 3) {
        o[0] = standardize ? percentile_of_hist(sc[tile].bytes, 90.0) : nan("");
        o[7] = (double)sc[tile].tissue;
    }
}

// ---- sweep C: the map.  MODE 0 Reinhard transform, 1 LuminosityStandardizer, 2 standardize_brightness only ----------
struct LabMapArgs {
    const uint8_t* rgb; uint8_t* out; int P, parts;
    const LabScratch* sc;
    LabTileTabs* tt;                                            // [n] per-tile tables (k_lab_pre, k_lab_tables)
    const double* target_means; const double* target_stds;      // MODE 0 (device, 3 each)
    int mask_background; double thr;                            // MODE 0
    double percentile;                                          // MODE 1
    double* p_out;                                              // MODE 1 / 2 (may be NULL)
};

// The per-tile tables of the map sweep, one workgroup per tile (what every workgroup of the sweep used to rebuild): normalizer.py:81-83 /
// stain_utils.py:65 evaluated in binary64 on the 256 possible bytes, composed with the conversion tables the pixel path would
// look up next.  MODE 0: needs k_lab_pre's brightness table; MODE 1: identity brightness.
template <int MODE>
static __global__ __launch_bounds__(256) void k_lab_tables(LabMapArgs a) {
    __shared__ uint8_t s_ch[3][256];          // per-channel Lab byte tables (MODE 0: all three; MODE 1: L only)
    __shared__ double s_p;
    __shared__ double s_ms[6];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const LabScratch& sc = a.sc[tile];
    LabTileTabs& tt = a.tt[tile];
    if (MODE == 0) {
        if (tid < 3) {
            double m, s;
            mean_std_of_scratch(sc, tid, m, s);
            s_ms[tid] = m; s_ms[3 + tid] = s;
        }
        __syncthreads();
        // normalizer.py:81-83 in binary64: ((x - mean) * (tstd / std)) + tmean; merge_back (stain_utils.py:168-171): * 2.55 resp.
        // + 128.0, clip, truncate.  x is the binary32 value lab_split produced, promoted.
        for (int ch = 0; ch < 3; ++ch) {
            const double ratio = a.target_stds[ch] / s_ms[3 + ch];
            const double x = ch == 0 ? (double)((float)tid / 2.55f) : (double)((float)tid - 128.0f);
            const double nrm = ((x - s_ms[ch]) * ratio) + a.target_means[ch];
            s_ch[ch][tid] = (uint8_t)clip_trunc_u8(ch == 0 ? nrm * 2.55 : nrm + 128.0);
        }
    } else {
        if (tid == 0) s_p = percentile_of_hist(sc.lab_l, a.percentile);
        __syncthreads();
        s_ch[0][tid] = (uint8_t)clip_trunc_u8(255.0 * (double)tid / s_p);          // stain_utils.py:65: 255 * L_float / p
        if (tid == 0) tt.p = s_p;
        tt.g[tid] = (uint16_t)d_gamma[tid];
        tt.lut[tid] = (uint8_t)tid;
    }
    __syncthreads();
    const int L2 = s_ch[0][tid], A2 = MODE == 0 ? s_ch[1][tid] : tid, B2 = MODE == 0 ? s_ch[2][tid] : tid;
    tt.yf[tid] = (uint32_t)d_lab_yf[2 * L2] | ((uint32_t)d_lab_yf[2 * L2 + 1] << 16);
    tt.ad[tid] = lab_adiv(A2);
    tt.bd[tid] = lab_bdiv(B2);
}

// The sweep: per-tile tables from k_lab_pre / k_lab_tables; two chunks per lane in flight.
template <int MODE, bool ALIGNED>
static __global__ __launch_bounds__(kLabWG) void k_lab_map(LabMapArgs a) {
    __shared__ LabCbrtInv s_t;
    __shared__ uint8_t s_lut[256];            // brightness table (MODE 2)
    __shared__ uint16_t s_g[256];
    __shared__ uint32_t s_yf[256];
    __shared__ int s_ad[256], s_bd[256];
    const int tile = blockIdx.x / a.parts, part = blockIdx.x % a.parts, tid = threadIdx.x;
    const LabTileTabs& tt = a.tt[tile];
    if (MODE != 2) {
        s_t.fill();
        s_g[tid] = tt.g[tid];
        s_yf[tid] = tt.yf[tid];
        if (MODE == 0) { s_ad[tid] = tt.ad[tid]; s_bd[tid] = tt.bd[tid]; }
    } else {
        s_lut[tid] = tt.lut[tid];
    }
    if (MODE != 0 && part == 0 && tid == 0 && a.p_out) a.p_out[tile] = tt.p;
    __syncthreads();
    const int lim = MODE == 0 ? l8_limit(a.thr) : 256;
    // background (normalizer.py:86-90): 254 + 0 on the L/2.55 scale -> clips to 255; a = b = 0 + 128
    const uint32_t yf_bg = (uint32_t)d_lab_yf[2 * 255] | ((uint32_t)d_lab_yf[2 * 255 + 1] << 16);
    const int ad_bg = lab_adiv(128), bd_bg = lab_bdiv(128);
    const bool mask_bg = MODE == 0 && a.mask_background;
    const size_t nbytes = (size_t)a.P * 3;
    const uint8_t* src = a.rgb + (size_t)tile * nbytes;
    uint8_t* dst = a.out + (size_t)tile * nbytes;
    const int nch = (a.P + 3) >> 2;
    const int span = (nch + a.parts - 1) / a.parts;
    const int c0 = part * span, c1 = min(nch, c0 + span);
    if (c0 >= c1) return;
    auto one = [&](const Chunk& in, int c) {
        uint32_t ob[12];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const uint32_t r = chunk_byte(in, 3 * px), g = chunk_byte(in, 3 * px + 1), b = chunk_byte(in, 3 * px + 2);
            if (MODE == 2) {
                ob[3 * px] = s_lut[r]; ob[3 * px + 1] = s_lut[g]; ob[3 * px + 2] = s_lut[b];
                continue;
            }
            int L, A, B;
            gamma_to_lab8(s_t, s_g[r], s_g[g], s_g[b], L, A, B);
            uint32_t yf = s_yf[L];
            int ad = MODE == 0 ? s_ad[A] : lab_adiv(A), bd = MODE == 0 ? s_bd[B] : lab_bdiv(B);     // (MODE 1 leaves a and b alone: three operations each)
            if (mask_bg) {
                const bool bg = !(L < lim);
                yf = bg ? yf_bg : yf; ad = bg ? ad_bg : ad; bd = bg ? bd_bg : bd;
            }
            yf_to_rgb(s_t, (int)(yf & 0xffffu), (int)(yf >> 16), ad, bd, ob[3 * px], ob[3 * px + 1], ob[3 * px + 2]);
        }
        store_chunk<ALIGNED>(dst, nbytes, c, pack12(ob));
    };
    // (the next trip's two chunks are requested before this trip's arithmetic; clamped loads, never predicated)
    Chunk n0 = load_chunk_clamped<ALIGNED, true>(src, nbytes, c0 + tid, c1), n1 = load_chunk_clamped<ALIGNED, true>(src, nbytes, c0 + tid + kLabWG, c1);
    for (int c = c0 + tid; c < c1; c += 2 * kLabWG) {
        const Chunk i0 = n0, i1 = n1;
        n0 = load_chunk_clamped<ALIGNED, true>(src, nbytes, c + 2 * kLabWG, c1);
        n1 = load_chunk_clamped<ALIGNED, true>(src, nbytes, c + 3 * kLabWG, c1);
        one(i0, c);
        if (c + kLabWG < c1) one(i1, c + kLabWG);
    }
}

// ---- plain conversions -------------------------------------------------------------------------------------------
// DIR 0: RGB -> Lab8 bytes; 1: Lab8 -> RGB bytes
template <int DIR, bool ALIGNED>
static __global__ __launch_bounds__(kLabWG) void k_lab_convert(const uint8_t* __restrict__ in_img, uint8_t* __restrict__ out_img, int P, int parts) {
    __shared__ LabTabs s_t;
    s_t.fill();
    __syncthreads();
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    const size_t nbytes = (size_t)P * 3;
    const uint8_t* src = in_img + (size_t)tile * nbytes;
    uint8_t* dst = out_img + (size_t)tile * nbytes;
    const int nch = (P + 3) >> 2;
    const int span = (nch + parts - 1) / parts;
    const int c0 = part * span, c1 = min(nch, c0 + span);
    for (int c = c0 + (int)threadIdx.x; c < c1; c += kLabWG) {
        const Chunk in = load_chunk<ALIGNED>(src, nbytes, c);
        uint32_t ob[12];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const uint32_t v0 = chunk_byte(in, 3 * px), v1 = chunk_byte(in, 3 * px + 1), v2 = chunk_byte(in, 3 * px + 2);
            if (DIR == 0) {
                int L, A, B;
                rgb_to_lab8(s_t, v0, v1, v2, L, A, B);
                ob[3 * px] = (uint32_t)L; ob[3 * px + 1] = (uint32_t)A; ob[3 * px + 2] = (uint32_t)B;
            } else {
                lab8_to_rgb(s_t, (int)v0, (int)v1, (int)v2, ob[3 * px], ob[3 * px + 1], ob[3 * px + 2]);
            }
        }
        store_chunk<ALIGNED>(dst, nbytes, c, pack12(ob));
    }
}

// lab_split (stain_utils.py:146-158): binary32 planes
static __global__ __launch_bounds__(kLabWG) void k_lab_split(const uint8_t* __restrict__ rgb, size_t n_px, float* __restrict__ I1,
                                                             float* __restrict__ I2, float* __restrict__ I3) {
    __shared__ LabTabs s_t;
    s_t.fill();
    __syncthreads();
    for (size_t p = blockIdx.x * (size_t)kLabWG + threadIdx.x; p < n_px; p += (size_t)gridDim.x * kLabWG) {
        int L, A, B;
        rgb_to_lab8(s_t, rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2], L, A, B);
        I1[p] = (float)L / 2.55f; I2[p] = (float)A - 128.0f; I3[p] = (float)B - 128.0f;
    }
}

// merge_back (stain_utils.py:160-172)
template <class T>
static __global__ __launch_bounds__(kLabWG) void k_lab_merge(const T* __restrict__ I1, const T* __restrict__ I2, const T* __restrict__ I3,
                                                             size_t n_px, uint8_t* __restrict__ out) {
    __shared__ LabTabs s_t;
    s_t.fill();
    __syncthreads();
    for (size_t p = blockIdx.x * (size_t)kLabWG + threadIdx.x; p < n_px; p += (size_t)gridDim.x * kLabWG) {
        const T l = I1[p] * (T)2.55, a = I2[p] + (T)128.0, b = I3[p] + (T)128.0;      // in the planes' own precision, like numpy
        uint32_t r, g, bl;
        lab8_to_rgb(s_t, (int)clip_trunc_u8((double)l), (int)clip_trunc_u8((double)a), (int)clip_trunc_u8((double)b), r, g, bl);
        out[3 * p] = (uint8_t)r; out[3 * p + 1] = (uint8_t)g; out[3 * p + 2] = (uint8_t)bl;
    }
}

// convert_OD_to_RGB (stain_utils.py:114-124), binary64 like the reference
static __global__ __launch_bounds__(kLabWG) void k_od_to_rgb(const double* __restrict__ od, size_t n, uint8_t* __restrict__ out,
                                                             int32_t* __restrict__ negative) {
    bool neg = false;
    for (size_t i = blockIdx.x * (size_t)kLabWG + threadIdx.x; i < n; i += (size_t)gridDim.x * kLabWG) {
        const double v = od[i];
        neg = neg | (v < 0.0);
        out[i] = (uint8_t)(255.0 * exp(-1.0 * fmax(v, 1e-6)));
    }
    if (negative && __any(neg) && (threadIdx.x & 63) == 0) atomicOr(negative, 1);
}

}  // namespace sl

using namespace sl;

namespace {

// Workgroups per tile of the Lab sweeps: every workgroup merges a 256-bin histogram into the tile's with global atomics (sweeps A, B) or
// fills 14 KB of LDS tables (sweep C) before it touches a pixel, so a tile is split only as far as it takes to put ~8 workgroups on
// every CU.  Measured on 1 250 tiles of 512^2 (round 5, per-tile tables precomputed): 1 250 / 2 500 / 3 750 / 8 750 / 17 500 workgroups:
// k_lab_hist 529 / 526 / 525 / 629 / 882 us, k_lab_map<0> 909 / 903 / 927 / 1 016 / - us.
int lab_parts(int n, long P) {
    const long want = (2048 + n - 1) / n;
    const long nch = (P + 3) >> 2;
    long most = nch / (16L * kLabWG);
    if (most < 1) most = 1;
    return (int)(want < 1 ? 1 : (want < most ? want : most));
}

size_t lab_scratch_bytes(int n) { return (sizeof(LabScratch) * (size_t)n + 255) & ~(size_t)255; }
size_t lab_ws_bytes(int n) { return lab_scratch_bytes(n) + ((sizeof(LabTileTabs) * (size_t)n + 255) & ~(size_t)255); }
LabTileTabs* lab_tabs_of(void* ws, int n) { return (LabTileTabs*)((char*)ws + lab_scratch_bytes(n)); }

int lab_check(const void* rgb, const void* out, int n, int h, int w, const void* ws, size_t ws_bytes) {
    if (!rgb || !out || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    if ((long)h * w > (1L << 30)) return SL_ERR_BADARG;
    if (!ws || ws_bytes < lab_ws_bytes(n) || ((uintptr_t)ws & 7u)) return SL_ERR_WORKSPACE;
    return SL_OK;
}

// the two statistics sweeps shared by the entry points below (and k_lab_pre between them: the tile's brightness table)
int lab_statistics(const uint8_t* rgb, int n, long P, int standardize, int want_ab, double thr, LabScratch* sc, hipStream_t s) {
    zero_async(sc, sizeof(LabScratch) * (size_t)n, s);
    LabTileTabs* tt = lab_tabs_of(sc, n);
    const int parts = lab_parts(n, P);
    const dim3 grid((unsigned)((long)n * parts)), block(kLabWG);
    const bool al = aligned4(rgb, P);
    if (standardize) {
        if (al) hipLaunchKernelGGL((k_byte_hist<true>), grid, block, 0, s, rgb, (int)P, parts, sc);
        else    hipLaunchKernelGGL((k_byte_hist<false>), grid, block, 0, s, rgb, (int)P, parts, sc);
    }
    hipLaunchKernelGGL(k_lab_pre, dim3((unsigned)n), dim3(256), 0, s, sc, standardize, tt);
    if (want_ab >= 0) {
        if (al) hipLaunchKernelGGL((k_lab_hist<true>), grid, block, 0, s, rgb, (int)P, parts, want_ab, thr, sc, tt);
        else    hipLaunchKernelGGL((k_lab_hist<false>), grid, block, 0, s, rgb, (int)P, parts, want_ab, thr, sc, tt);
    }
    return launch_status();
}

template <int MODE>
int lab_map(const LabMapArgs& a, int n, hipStream_t s) {
    if (MODE != 2) hipLaunchKernelGGL((k_lab_tables<MODE>), dim3((unsigned)n), dim3(256), 0, s, a);
    const dim3 grid((unsigned)((long)n * a.parts)), block(kLabWG);
    if (aligned4(a.rgb, a.P) && aligned4(a.out, a.P)) hipLaunchKernelGGL((k_lab_map<MODE, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_lab_map<MODE, false>), grid, block, 0, s, a);
    return launch_status();
}

}  // namespace

namespace sl { size_t lab_workspace_bytes(int n) { return lab_ws_bytes(n); } }      // for sl_workspace_bytes (macenko.hip)

extern "C" int sl_rgb_to_lab8(const uint8_t* rgb, uint8_t* lab_out, int n, int h, int w, void* stream) {
    if (!rgb || !lab_out || n <= 0 || h <= 0 || w <= 0 || (long)h * w > (1L << 30)) return SL_ERR_BADARG;
    const long P = (long)h * w;
    const int parts = parts_for(P);
    const dim3 grid((unsigned)((long)n * parts)), block(kLabWG);
    if (aligned4(rgb, P) && aligned4(lab_out, P)) hipLaunchKernelGGL((k_lab_convert<0, true>), grid, block, 0, (hipStream_t)stream, rgb, lab_out, (int)P, parts);
    else hipLaunchKernelGGL((k_lab_convert<0, false>), grid, block, 0, (hipStream_t)stream, rgb, lab_out, (int)P, parts);
    return launch_status();
}

extern "C" int sl_lab8_to_rgb(const uint8_t* lab, uint8_t* rgb_out, int n, int h, int w, void* stream) {
    if (!lab || !rgb_out || n <= 0 || h <= 0 || w <= 0 || (long)h * w > (1L << 30)) return SL_ERR_BADARG;
    const long P = (long)h * w;
    const int parts = parts_for(P);
    const dim3 grid((unsigned)((long)n * parts)), block(kLabWG);
    if (aligned4(lab, P) && aligned4(rgb_out, P)) hipLaunchKernelGGL((k_lab_convert<1, true>), grid, block, 0, (hipStream_t)stream, lab, rgb_out, (int)P, parts);
    else hipLaunchKernelGGL((k_lab_convert<1, false>), grid, block, 0, (hipStream_t)stream, lab, rgb_out, (int)P, parts);
    return launch_status();
}

extern "C" int sl_lab_split(const uint8_t* rgb, int n, int h, int w, float* I1, float* I2, float* I3, void* stream) {
    if (!rgb || !I1 || !I2 || !I3 || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    const size_t n_px = (size_t)n * h * w;
    const size_t blocks = (n_px + kLabWG - 1) / kLabWG;
    hipLaunchKernelGGL(k_lab_split, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(kLabWG), 0, (hipStream_t)stream, rgb, n_px, I1, I2, I3);
    return launch_status();
}

extern "C" int sl_lab_merge(const void* I1, const void* I2, const void* I3, int is_f64, int n, int h, int w, uint8_t* rgb_out, void* stream) {
    if (!I1 || !I2 || !I3 || !rgb_out || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    const size_t n_px = (size_t)n * h * w;
    const size_t blocks = (n_px + kLabWG - 1) / kLabWG;
    const dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192)), block(kLabWG);
    if (is_f64) hipLaunchKernelGGL((k_lab_merge<double>), grid, block, 0, (hipStream_t)stream, (const double*)I1, (const double*)I2, (const double*)I3, n_px, rgb_out);
    else hipLaunchKernelGGL((k_lab_merge<float>), grid, block, 0, (hipStream_t)stream, (const float*)I1, (const float*)I2, (const float*)I3, n_px, rgb_out);
    return launch_status();
}

extern "C" int sl_od_to_rgb(const double* od, size_t n_values, uint8_t* rgb_out, int32_t* negative_flag, void* stream) {
    if (!od || !rgb_out || n_values == 0) return SL_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (negative_flag) zero_async(negative_flag, sizeof(int32_t), s);
    const size_t blocks = (n_values + kLabWG - 1) / kLabWG;
    hipLaunchKernelGGL(k_od_to_rgb, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(kLabWG), 0, s, od, n_values, rgb_out, negative_flag);
    return launch_status();
}

extern "C" int sl_standardize_brightness(const uint8_t* rgb, uint8_t* out, int n, int h, int w, double* p_out, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    int rc = lab_check(rgb, out, n, h, w, workspace, workspace_bytes);
    if (rc) return rc;
    const long P = (long)h * w;
    hipStream_t s = (hipStream_t)stream;
    LabScratch* sc = (LabScratch*)workspace;
    rc = lab_statistics(rgb, n, P, 1, -1, 0.0, sc, s);
    if (rc) return rc;
    LabMapArgs a{};
    a.rgb = rgb; a.out = out; a.P = (int)P; a.parts = lab_parts(n, P); a.sc = sc; a.tt = lab_tabs_of(sc, n); a.p_out = p_out;
    return lab_map<2>(a, n, s);
}

extern "C" int sl_reinhard_stats(const uint8_t* rgb, int n, int h, int w, int standardize, double* stats_out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    int rc = lab_check(rgb, stats_out, n, h, w, workspace, workspace_bytes);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    LabScratch* sc = (LabScratch*)workspace;
    rc = lab_statistics(rgb, n, (long)h * w, standardize ? 1 : 0, 1, 0.8, sc, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_lab_stats, dim3((unsigned)n), dim3(64), 0, s, sc, standardize ? 1 : 0, stats_out);
    return launch_status();
}

extern "C" int sl_reinhard_transform(const uint8_t* rgb, uint8_t* out, int n, int h, int w, const double* target_means,
                                     const double* target_stds, int mask_background, double luminosity_threshold,
                                     double* stats_out, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = lab_check(rgb, out, n, h, w, workspace, workspace_bytes);
    if (rc) return rc;
    if (!target_means || !target_stds) return SL_ERR_BADARG;
    const long P = (long)h * w;
    hipStream_t s = (hipStream_t)stream;
    LabScratch* sc = (LabScratch*)workspace;
    rc = lab_statistics(rgb, n, P, 1, 1, luminosity_threshold, sc, s);
    if (rc) return rc;
    if (stats_out) hipLaunchKernelGGL(k_lab_stats, dim3((unsigned)n), dim3(64), 0, s, sc, 1, stats_out);
    LabMapArgs a{};
    a.rgb = rgb; a.out = out; a.P = (int)P; a.parts = lab_parts(n, P); a.sc = sc; a.tt = lab_tabs_of(sc, n);
    a.target_means = target_means; a.target_stds = target_stds;
    a.mask_background = mask_background ? 1 : 0; a.thr = luminosity_threshold;
    return lab_map<0>(a, n, s);
}

extern "C" int sl_luminosity_standardize(const uint8_t* rgb, uint8_t* out, int n, int h, int w, double percentile, double* p_out,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    int rc = lab_check(rgb, out, n, h, w, workspace, workspace_bytes);
    if (rc) return rc;
    const long P = (long)h * w;
    hipStream_t s = (hipStream_t)stream;
    LabScratch* sc = (LabScratch*)workspace;
    rc = lab_statistics(rgb, n, P, 0, 0, 0.8, sc, s);
    if (rc) return rc;
    LabMapArgs a{};
    a.rgb = rgb; a.out = out; a.P = (int)P; a.parts = lab_parts(n, P); a.sc = sc; a.tt = lab_tabs_of(sc, n); a.percentile = percentile; a.p_out = p_out;
    return lab_map<1>(a, n, s);
}
