// hed.hip -- HedColorAugmenter.transform (stainlib/augmentation/augmenter.py:276-331) for uint8 tiles.
//
// Reference chain: mean cutoff test -> skimage.color.rgb2hed -> per-channel x*(1+sigma)+bias ->
// hed2rgb -> clip [0,1] -> *255 -> truncate.  With scikit-image 0.18 semantics
// (colorconv.py:1448-1454, 1511-1518) the two 3x3 products and the affine fold into
//     ln rgb' = ln(max(rgb,1e-6)) @ (H diag(1+sigma) R) + ln(1e-6) * (bias @ R)
// so a pixel costs 3 LDS lookups, 9 FMA, 3 exp: the sweep is HBM-bound at 3 B read + 3 B written.
// The cutoff test needs the tile mean BEFORE the transform; instead of a separate 3 B/px sweep the
// transform runs speculatively while the same sweep sums the bytes, and a fix-up kernel copies the
// (rare) tiles that fail the test.  >=0.19 semantics (stains clamped at 0) keep the two products apart; the presumed
// <=0.17 semantics (the version environment.yml:107 pins: -ln(rgb+2), exp(.) - 2; restated from memory, unpinned) fold the
// same way with another table and another weight on the bias term; the base-10 reading of it is kept as an experiment.
#include "apply_kernels.hpp"
#include "sl_host.hpp"

namespace sl {

struct HedConst { double H[9]; double R[9]; double log2_base; };   // hed_from_rgb, rgb_from_hed (row-major); MODE 2: log2 of the logarithm's base

template <int MODE, bool ALIGNED>
static __global__ __launch_bounds__(kWG) void k_hed(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ out, int P,
                                                    int parts, const double* __restrict__ sigma,
                                                    const double* __restrict__ bias, HedConst hc,
                                                    unsigned long long* __restrict__ sums) {
    __shared__ float s_x[256];      // ln(max(v/255, 1e-6)) * log2(e)-free: plain natural log, binary32
    __shared__ unsigned long long s_sum;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    const double Ladj = log(1e-6);
    if (MODE == 2) {
        s_x[tid] = (float)log2((double)tid / 255.0 + 2.0);                 // scikit-image <= 0.17: log(rgb + 2), kept in base 2
    } else {
        const double v = tid == 0 ? 1e-6 : fmax((double)tid / 255.0, 1e-6);
        s_x[tid] = (float)log(v);
    }
    if (tid == 0) s_sum = 0;
    // per-tile constants in binary64, then binary32 in VGPRs (a VALU op with an SGPR operand issues at half rate)
    float A[3][3], b[3], Hs[3][3], Rs[3][3], sc[3], bi[3];
    const double* sg = sigma + 3 * (size_t)tile;
    const double* bs = bias + 3 * (size_t)tile;
    const double kL2E = 1.4426950408889634;
    if (MODE == 0 || MODE == 2) {
        // MODE 2 (<= 0.17): stains = -log_B(x+2) @ H, rgb' = B^(-stains' @ R) - 2 (B = e, or 10 in the experimental reading), i.e.
        //     log2(rgb'+2) = log2(x+2) @ (H diag(1+sigma) R) - log2(B) * (bias @ R)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double acc = 0;
                for (int j = 0; j < 3; ++j) acc += hc.H[3 * k + j] * (1.0 + sg[j]) * hc.R[3 * j + c];
                A[k][c] = in_vgpr((float)(MODE == 2 ? acc : acc * kL2E));
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
            for (int j = 0; j < 3; ++j) acc += bs[j] * hc.R[3 * j + c];
            b[c] = in_vgpr((float)(MODE == 2 ? -hc.log2_base * acc : Ladj * acc * kL2E));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Hs[k][c] = in_vgpr((float)(hc.H[3 * k + c] / Ladj));           // stains = ln(rgb)/ln(1e-6) @ H
                Rs[k][c] = in_vgpr((float)(hc.R[3 * k + c] * Ladj * kL2E));    // log2 rgb = ln(1e-6) * stains @ R * log2(e)
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) { sc[c] = in_vgpr((float)(1.0 + sg[c])); bi[c] = in_vgpr((float)bs[c]); }
    }
    __syncthreads();

    const size_t nbytes = (size_t)P * 3;
    const uint8_t* src = rgb + (size_t)tile * nbytes;
    uint8_t* dst = out + (size_t)tile * nbytes;
    const int nch = (P + 3) >> 2;
    const int span = (nch + parts - 1) / parts;
    const int c0 = part * span, c1 = min(nch, c0 + span);
    uint32_t bsum = 0;
    for (int c = c0 + tid; c < c1; c += kWG * kU) {
        Chunk in[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int cc = c + u * kWG;
            in[u] = load_chunk_clamped<ALIGNED, true>(src, nbytes, cc, c1);   // single pass: non-temporal; no predicated load (see load_chunk_clamped); dead lanes masked below
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int cc = c + u * kWG;
            const uint32_t one = cc < c1 ? 0x01010101u : 0u;                 // lanes past the end add nothing to the byte sum
            bsum = __builtin_amdgcn_udot4(in[u].w0, one, bsum, false);
            bsum = __builtin_amdgcn_udot4(in[u].w1, one, bsum, false);
            bsum = __builtin_amdgcn_udot4(in[u].w2, one, bsum, false);
            float tv[12];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const float x0 = s_x[chunk_byte(in[u], 3 * px)], x1 = s_x[chunk_byte(in[u], 3 * px + 1)],
                            x2 = s_x[chunk_byte(in[u], 3 * px + 2)];
                float l[3];
                if (MODE == 0 || MODE == 2) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) l[ch] = fmaf(x2, A[2][ch], fmaf(x1, A[1][ch], fmaf(x0, A[0][ch], b[ch])));
                } else {
                    float st[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        st[k] = fmaf(x2, Hs[2][k], fmaf(x1, Hs[1][k], x0 * Hs[0][k]));
                        st[k] = fmaxf(st[k], 0.0f);                        // scikit-image >= 0.19
                        st[k] = fmaf(st[k], sc[k], bi[k]);                 // augmenter.py:298-316
                    }
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) l[ch] = fmaf(st[2], Rs[2][ch], fmaf(st[1], Rs[1][ch], st[0] * Rs[0][ch]));
                }
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    tv[3 * px + ch] = MODE == 2 ? fmaf(255.0f, __builtin_amdgcn_exp2f(l[ch]), -510.0f) : 255.0f * __builtin_amdgcn_exp2f(l[ch]);
            }
            // clip [0,1], *255, astype(uint8) (augmenter.py:320-325): max(min(255 x, 255), 0) truncated = the saturating pack
            const Chunk o = pack_trunc_fast(tv);
            if (cc < c1) store_chunk<ALIGNED, true>(dst, nbytes, cc, o);
        }
    }
    unsigned long long ws = wave_sum((unsigned long long)bsum);
    if ((tid & 63) == 0) atomicAdd(&s_sum, ws);
    __syncthreads();
    if (tid == 0) atomicAdd(&sums[tile], s_sum);
}

// Tiles whose mean/255 is outside [lo, hi] are returned unchanged (augmenter.py:293,331).
template <bool ALIGNED>
static __global__ __launch_bounds__(kWG) void k_hed_fixup(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ out, int P,
                                                          int parts, const unsigned long long* __restrict__ sums,
                                                          double lo, double hi, int32_t* __restrict__ applied) {
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    // the reference averages float32 values then divides by 255 (augmenter.py:291); here the byte sum is exact
    const double mean = (double)sums[tile] / (3.0 * (double)P) / 255.0;
    const bool ok = (lo <= mean) && (mean <= hi);
    if (part == 0 && threadIdx.x == 0 && applied) applied[tile] = ok ? 1 : 0;
    if (ok) return;
    const size_t nbytes = (size_t)P * 3;
    const uint8_t* src = rgb + (size_t)tile * nbytes;
    uint8_t* dst = out + (size_t)tile * nbytes;
    const int nch = (P + 3) >> 2;
    const int span = (nch + parts - 1) / parts;
    const int c0 = part * span, c1 = min(nch, c0 + span);
    for (int c = c0 + threadIdx.x; c < c1; c += kWG) store_chunk<ALIGNED>(dst, nbytes, c, load_chunk<ALIGNED>(src, nbytes, c));
}


// ---- float patches (augmenter.py:288-289, 319-320): values in [0,1], binary64 in and out -------------
template <int MODE>
static __global__ __launch_bounds__(kWG) void k_hed_f64(const double* __restrict__ rgb, double* __restrict__ out, int P,
                                                        int parts, const double* __restrict__ sigma,
                                                        const double* __restrict__ bias, HedConst hc,
                                                        double* __restrict__ sums) {
    __shared__ double s_sum;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    if (tid == 0) s_sum = 0;
    __syncthreads();
    const double Ladj = log(1e-6);
    const double* sg = sigma + 3 * (size_t)tile;
    const double* bs = bias + 3 * (size_t)tile;
    const int span = (P + parts - 1) / parts;
    const int p0 = part * span, p1 = min(P, p0 + span);
    const double* src = rgb + (size_t)tile * P * 3;
    double* dst = out + (size_t)tile * P * 3;
    double acc = 0;
    for (int p = p0 + tid; p < p1; p += kWG) {
        const double r = src[3 * (size_t)p], g = src[3 * (size_t)p + 1], b = src[3 * (size_t)p + 2];
        acc += r + g + b;
        double x[3];
        if (MODE == 2) { x[0] = -log(r + 2.0); x[1] = -log(g + 2.0); x[2] = -log(b + 2.0); }             // presumed <= 0.17
        else if (MODE == 3) { x[0] = -log10(r + 2.0); x[1] = -log10(g + 2.0); x[2] = -log10(b + 2.0); }   // experimental base-10 reading
        else { x[0] = log(fmax(r, 1e-6)) / Ladj; x[1] = log(fmax(g, 1e-6)) / Ladj; x[2] = log(fmax(b, 1e-6)) / Ladj; }
        double st[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            st[k] = x[0] * hc.H[k] + x[1] * hc.H[3 + k] + x[2] * hc.H[6 + k];       // separate_stains
            if (MODE == 1) st[k] = fmax(st[k], 0.0);
            st[k] = st[k] * (1.0 + sg[k]) + bs[k];                                  // augmenter.py:298-316
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v;
            if (MODE == 2) {
                v = exp(-(st[0] * hc.R[c] + st[1] * hc.R[3 + c] + st[2] * hc.R[6 + c])) - 2.0;
            } else if (MODE == 3) {
                v = pow(10.0, -(st[0] * hc.R[c] + st[1] * hc.R[3 + c] + st[2] * hc.R[6 + c])) - 2.0;
            } else {
                const double lr = -(st[0] * (-Ladj)) * hc.R[c] - (st[1] * (-Ladj)) * hc.R[3 + c] - (st[2] * (-Ladj)) * hc.R[6 + c];
                v = exp(lr);
            }
            dst[3 * (size_t)p + c] = fmin(fmax(v, 0.0), 1.0);                       // combine_stains + clip
        }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) atomicAdd(&s_sum, acc);
    __syncthreads();
    if (tid == 0) atomicAdd(&sums[tile], s_sum);
}

static __global__ __launch_bounds__(kWG) void k_hed_f64_fixup(const double* __restrict__ rgb, double* __restrict__ out,
                                                              int P, int parts, const double* __restrict__ sums,
                                                              double lo, double hi, int32_t* __restrict__ applied) {
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    const double mean = sums[tile] / (3.0 * (double)P);                              // np.mean(patch), augmenter.py:289
    const bool ok = (lo <= mean) && (mean <= hi);
    if (part == 0 && threadIdx.x == 0 && applied) applied[tile] = ok ? 1 : 0;
    if (ok) return;
    const int span = (P + parts - 1) / parts;
    const int p0 = part * span, p1 = min(P, p0 + span);
    for (size_t i = 3 * (size_t)p0 + threadIdx.x; i < 3 * (size_t)p1; i += kWG)
        out[(size_t)tile * P * 3 + i] = rgb[(size_t)tile * P * 3 + i];
}

// convert_RGB_to_OD (stain_utils.py:101-112) materialised
static __global__ __launch_bounds__(kWG) void k_rgb_to_od(const uint8_t* __restrict__ rgb, size_t nbytes, double* __restrict__ od) {
    for (size_t i = blockIdx.x * (size_t)kWG + threadIdx.x; i < nbytes; i += (size_t)gridDim.x * kWG) od[i] = d_od_f64[rgb[i]];
}

}  // namespace sl

using namespace sl;

namespace {
void inv3(const double* m, double* o) {
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    o[0] = (e * i - f * h) / det; o[1] = (c * h - b * i) / det; o[2] = (b * f - c * e) / det;
    o[3] = (f * g - d * i) / det; o[4] = (a * i - c * g) / det; o[5] = (c * d - a * f) / det;
    o[6] = (d * h - e * g) / det; o[7] = (b * g - a * h) / det; o[8] = (a * e - b * d) / det;
}
}  // namespace

extern "C" int sl_hed_augment(const uint8_t* rgb, uint8_t* out, int n, int h, int w, const double* sigma,
                              const double* bias, double cutoff_lo, double cutoff_hi, int skimage_mode,
                              int32_t* applied, void* workspace, size_t workspace_bytes, void* stream) {
    if (!rgb || !out || !sigma || !bias || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    if (skimage_mode < SL_HED_SKIMAGE_018 || skimage_mode > SL_HED_EXPERIMENTAL_LOG10) return SL_ERR_BADARG;
    const long P = (long)h * w;
    if (P > (1L << 30)) return SL_ERR_BADARG;
    if (!workspace || workspace_bytes < sizeof(unsigned long long) * (size_t)n || ((uintptr_t)workspace & 7u))
        return SL_ERR_WORKSPACE;
    HedConst hc;
    // skimage.color.rgb_from_hed (colorconv.py:475-478), hed_from_rgb = inv(.)
    const double R[9] = {0.65, 0.70, 0.29, 0.07, 0.99, 0.11, 0.27, 0.57, 0.78};
    for (int i = 0; i < 9; ++i) hc.R[i] = R[i];
    inv3(R, hc.H);
    hc.log2_base = skimage_mode == SL_HED_EXPERIMENTAL_LOG10 ? 3.321928094887362 : 1.4426950408889634;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* sums = (unsigned long long*)workspace;
    zero_async(sums, sizeof(unsigned long long) * (size_t)n, s);
    const int parts = parts_for(P);
    const dim3 grid((unsigned)((long)n * parts)), block(kWG);
    const bool al = aligned4(rgb, P) && aligned4(out, P);
#define SL_GO(M, A) hipLaunchKernelGGL((k_hed<M, A>), grid, block, 0, s, rgb, out, (int)P, parts, sigma, bias, hc, sums)
    if (skimage_mode == SL_HED_SKIMAGE_018)      { if (al) SL_GO(0, true); else SL_GO(0, false); }
    else if (skimage_mode == SL_HED_SKIMAGE_019) { if (al) SL_GO(1, true); else SL_GO(1, false); }
    else                                         { if (al) SL_GO(2, true); else SL_GO(2, false); }
#undef SL_GO
    if (al) hipLaunchKernelGGL((k_hed_fixup<true>), grid, block, 0, s, rgb, out, (int)P, parts, sums, cutoff_lo, cutoff_hi, applied);
    else    hipLaunchKernelGGL((k_hed_fixup<false>), grid, block, 0, s, rgb, out, (int)P, parts, sums, cutoff_lo, cutoff_hi, applied);
    return launch_status();
}

extern "C" int sl_hed_augment_f64(const double* rgb, double* out, int n, int h, int w, const double* sigma,
                                  const double* bias, double cutoff_lo, double cutoff_hi, int skimage_mode,
                                  int32_t* applied, void* workspace, size_t workspace_bytes, void* stream) {
    if (!rgb || !out || !sigma || !bias || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    if (skimage_mode < SL_HED_SKIMAGE_018 || skimage_mode > SL_HED_EXPERIMENTAL_LOG10) return SL_ERR_BADARG;
    const long P = (long)h * w;
    if (P > (1L << 30)) return SL_ERR_BADARG;
    if (!workspace || workspace_bytes < sizeof(double) * (size_t)n || ((uintptr_t)workspace & 7u)) return SL_ERR_WORKSPACE;
    HedConst hc;
    const double R[9] = {0.65, 0.70, 0.29, 0.07, 0.99, 0.11, 0.27, 0.57, 0.78};
    for (int i = 0; i < 9; ++i) hc.R[i] = R[i];
    inv3(R, hc.H);
    hc.log2_base = 0.0;
    hipStream_t s = (hipStream_t)stream;
    double* sums = (double*)workspace;
    zero_async(sums, sizeof(double) * (size_t)n, s);
    const int parts = parts_for(P);
    const dim3 grid((unsigned)((long)n * parts)), block(kWG);
    if (skimage_mode == SL_HED_SKIMAGE_018) hipLaunchKernelGGL((k_hed_f64<0>), grid, block, 0, s, rgb, out, (int)P, parts, sigma, bias, hc, sums);
    else if (skimage_mode == SL_HED_SKIMAGE_019) hipLaunchKernelGGL((k_hed_f64<1>), grid, block, 0, s, rgb, out, (int)P, parts, sigma, bias, hc, sums);
    else if (skimage_mode == SL_HED_SKIMAGE_017) hipLaunchKernelGGL((k_hed_f64<2>), grid, block, 0, s, rgb, out, (int)P, parts, sigma, bias, hc, sums);
    else hipLaunchKernelGGL((k_hed_f64<3>), grid, block, 0, s, rgb, out, (int)P, parts, sigma, bias, hc, sums);
    hipLaunchKernelGGL(k_hed_f64_fixup, grid, block, 0, s, rgb, out, (int)P, parts, sums, cutoff_lo, cutoff_hi, applied);
    return launch_status();
}

extern "C" int sl_rgb_to_od(const uint8_t* rgb, int n, int h, int w, double* od_out, void* stream) {
    if (!rgb || !od_out || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    const size_t nbytes = (size_t)n * h * w * 3;
    const size_t blocks = (nbytes + kWG - 1) / kWG;
    hipLaunchKernelGGL(k_rgb_to_od, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(kWG), 0, (hipStream_t)stream, rgb, nbytes, od_out);
    return launch_status();
}
