// slide.hip -- building blocks of the POOLED slide-level mode (BASELINE.json configs[4]):
// every tile of a slide, on every rank, is normalised with the stain matrix / 99th-percentile concentrations
// the reference would compute from the vertical concatenation of all the tiles as one tall image.
//
// The statistics of that tall image are sums and order statistics over all its pixels, so each rank reduces
// its own tiles to a few numbers and the host combines them with small all-reduces (stainlib_amd/distributed.py):
//   sl_tile_moments            per-tile {n, sum od, sum od od^T}            -> summed over tiles and ranks -> V
//   sl_slide_key_histogram     256-bin histogram of the next 8 key bits among the keys that match a prefix
//                              -> all-reduced; 4 rounds pin one exact order statistic of the binary32 key
//   sl_slide_key_next_above    smallest key above a given key (the k+1-th value when the k-th is unique)
// Keys are the ones the per-tile path selects on: the pseudo-angle of the projected OD (tissue pixels) and the
// two lasso concentrations (all pixels), as order-preserving uint32 of their binary32 value.  Nothing
// per-pixel is stored: every round is one more sweep over the uint8 tiles.
#include "stats_kernels.hpp"
#include "sl_host.hpp"

using namespace sl;

namespace {

constexpr int kSlideThreads = 256;

struct SlideArgs {
    const uint8_t* rgb;
    int P, parts;
    float ylimf;
    int key;                 // SL_KEY_*
    float V[6];              // angle: V[c*2+k]
    double M[6];             // concentrations
    double lam;
    uint32_t prefix;
    int prefix_bits;
    uint32_t above;          // next_above: keys strictly greater than this
};

// ordered-integer key of pixel (r,g,b); kAbsent when the pixel does not take part (not tissue)
__device__ __forceinline__ uint32_t slide_key(const SlideArgs& a, const TabView& tab, const LassoK& L, uint32_t r, uint32_t g, uint32_t b) {
    if (a.key == SL_KEY_ANGLE) {
        if (!is_tissue_f(tab.gam(r), tab.gam(g), tab.gam(b), a.ylimf)) return kAbsent;
        return f2ord(angle_key(a.V, tab.odf(r), tab.odf(g), tab.odf(b)));
    }
    float c1, c2;
    lasso2(L, tab.odf(r), tab.odf(g), tab.odf(b), c1, c2);
    return f2ord(a.key == SL_KEY_CONC0 ? c1 : c2);
}

template <bool NEXT_ABOVE>
__global__ __launch_bounds__(kSlideThreads) void k_slide_keys(SlideArgs a, unsigned long long* hist, uint32_t* min_out) {
    __shared__ SmallTab s_tab;
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_min;
    s_tab.fill();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_hist[i] = 0;
    if (threadIdx.x == 0) s_min = 0xffffffffu;
    __syncthreads();
    const TabView tab = view_of(s_tab);
    LassoK L;
    if (a.key != SL_KEY_ANGLE) lasso_consts(a.M, a.lam, L); else L.g12 = 0.0f;
    const int tile = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
    const uint8_t* src = a.rgb + (size_t)tile * a.P * 3;
    const int span = (a.P + a.parts - 1) / a.parts;
    const int p0 = part * span, p1 = min(a.P, p0 + span);
    uint32_t best = 0xffffffffu;
    const int sh = 24 - a.prefix_bits;
    for (int p = p0 + threadIdx.x; p < p1; p += kSlideThreads) {
        const uint32_t o = slide_key(a, tab, L, src[3 * (size_t)p], src[3 * (size_t)p + 1], src[3 * (size_t)p + 2]);
        if (o == kAbsent) continue;
        if (NEXT_ABOVE) {
            if (o > a.above) best = min(best, o);
        } else if (a.prefix_bits == 0 || (o >> (32 - a.prefix_bits)) == a.prefix) {
            atomicAdd(&s_hist[(o >> sh) & 255u], 1u);
        }
    }
    if (NEXT_ABOVE) {
        for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 64));
        if ((threadIdx.x & 63) == 0) atomicMin(&s_min, best);
        __syncthreads();
        if (threadIdx.x == 0 && s_min != 0xffffffffu) atomicMin(min_out, s_min);
    } else {
        __syncthreads();
        for (int i = threadIdx.x; i < 256; i += blockDim.x)
            if (s_hist[i]) atomicAdd(&hist[i], (unsigned long long)s_hist[i]);
    }
}

// per-tile moment sums in a fixed order (run-to-run identical): sweep -> partials -> one thread per (tile, moment)
__global__ __launch_bounds__(kSweepThreads, 4) void k_tile_moment_partials(const uint8_t* rgb, int P, int parts, float ylimf, double* partials) {
    __shared__ RowTab s_tab;
    __shared__ double s_red[kSweepThreads / 64][10];
    s_tab.fill();
    __syncthreads();
    const TabReader T = TabReader::make(s_tab);
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint8_t* src = rgb + (size_t)tile * P * 3;
    int c0, c1;
    part_range((P + 3) >> 2, parts, part, c0, c1);
    Moments mo;
    uint32_t n_tissue = 0;
    if ((P & 3) == 0 && ((uintptr_t)rgb & 3u) == 0)
        moments_sweep<true, kPhaseTrip>(src, P, c0, c1, tid, kSweepThreads, T, ylimf, 6, nullptr, mo, n_tissue);
    else
        moments_sweep<false, kPhaseTrip>(src, P, c0, c1, tid, kSweepThreads, T, ylimf, 6, nullptr, mo, n_tissue);
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
}
__global__ void k_sum_partials(const double* partials, int n, int parts, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 10) return;
    const int tile = i / 10, m = i % 10;
    double t = 0;
    for (int p = 0; p < parts; ++p) t += partials[((size_t)tile * parts + p) * 10 + m];
    out[i] = t;
}

int fill_args(SlideArgs& a, const uint8_t* rgb, int n, int h, int w, const SlParams* params, int key, const double* basis_host) {
    if (!rgb || n <= 0 || h <= 0 || w <= 0 || !basis_host) return SL_ERR_BADARG;
    if ((long)h * w > (1L << 30)) return SL_ERR_BADARG;
    if (key != SL_KEY_ANGLE && key != SL_KEY_CONC0 && key != SL_KEY_CONC1) return SL_ERR_BADARG;
    SlParams p;
    sl_default_params(&p);
    if (params) p = *params;
    a.rgb = rgb;
    a.P = h * w;
    a.parts = parts_for((long)h * w);
    a.ylimf = (float)y_limit_for_threshold(p.luminosity_threshold) - 2048.0f;
    a.key = key;
    a.lam = p.lasso_lambda;
    for (int i = 0; i < 6; ++i) { a.V[i] = (float)basis_host[i]; a.M[i] = basis_host[i]; }
    a.prefix = 0; a.prefix_bits = 0; a.above = 0;
    return SL_OK;
}

}  // namespace

extern "C" int sl_tile_moments(const uint8_t* rgb, int n, int h, int w, const SlParams* params, double* moments_out,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!rgb || !moments_out || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    const long P = (long)h * w;
    if (P > (1L << 30)) return SL_ERR_BADARG;
    const int parts = parts_for(P);
    const size_t need = sizeof(double) * 10 * (size_t)parts * (size_t)n;
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 7u)) return SL_ERR_WORKSPACE;
    SlParams p;
    sl_default_params(&p);
    if (params) p = *params;
    const float ylimf = (float)y_limit_for_threshold(p.luminosity_threshold) - 2048.0f;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_tile_moment_partials, dim3((unsigned)((long)n * parts)), dim3(kSweepThreads), 0, s, rgb, (int)P, parts, ylimf,
                       (double*)workspace);
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((n * 10 + 255) / 256)), dim3(256), 0, s, (const double*)workspace, n, parts,
                       moments_out);
    return launch_status();
}

extern "C" int sl_slide_key_histogram(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int key,
                                      const double* basis, uint32_t prefix, int prefix_bits, unsigned long long* hist,
                                      void* stream) {
    SlideArgs a;
    const int rc = fill_args(a, rgb, n, h, w, params, key, basis);
    if (rc) return rc;
    if (!hist || (prefix_bits != 0 && prefix_bits != 8 && prefix_bits != 16 && prefix_bits != 24)) return SL_ERR_BADARG;
    a.prefix = prefix; a.prefix_bits = prefix_bits;
    hipLaunchKernelGGL((k_slide_keys<false>), dim3((unsigned)((long)n * a.parts)), dim3(kSlideThreads), 0, (hipStream_t)stream, a, hist,
                       (uint32_t*)nullptr);
    return launch_status();
}

extern "C" int sl_slide_key_next_above(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int key,
                                       const double* basis, uint32_t key_ord, uint32_t* min_out, void* stream) {
    SlideArgs a;
    const int rc = fill_args(a, rgb, n, h, w, params, key, basis);
    if (rc) return rc;
    if (!min_out) return SL_ERR_BADARG;
    a.above = key_ord;
    hipLaunchKernelGGL((k_slide_keys<true>), dim3((unsigned)((long)n * a.parts)), dim3(kSlideThreads), 0, (hipStream_t)stream, a,
                       (unsigned long long*)nullptr, min_out);
    return launch_status();
}
