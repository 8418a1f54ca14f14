// apply.hip -- C-ABI launchers of the per-pixel sweeps in apply_kernels.hpp.
#include "apply_kernels.hpp"
#include "sl_host.hpp"

using namespace sl;

extern "C" int sl_normalize_apply(const uint8_t* rgb, uint8_t* out, int n, int h, int w, const double* M_src,
                                  const double* maxC_src, const double* M_tgt, const double* maxC_tgt,
                                  double lasso_lambda, float* prequant, void* stream) {
    if (!rgb || !out || !M_src || !maxC_src || !M_tgt || !maxC_tgt || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    const long P = (long)h * w;
    if (P > (1L << 30)) return SL_ERR_BADARG;
    const int parts = parts_for(P);
    const dim3 grid((unsigned)((long)n * parts)), block(kWG);
    hipStream_t s = (hipStream_t)stream;
    const bool al = aligned4(rgb, P) && aligned4(out, P);
#define SL_GO(A, Q)                                                                                         \
    hipLaunchKernelGGL((k_apply<A, Q>), grid, block, 0, s, rgb, out, (int)P, parts, M_src, maxC_src, M_tgt, \
                       maxC_tgt, lasso_lambda, prequant)
    if (al) { if (prequant) SL_GO(true, true); else SL_GO(true, false); }
    else    { if (prequant) SL_GO(false, true); else SL_GO(false, false); }
#undef SL_GO
    return launch_status();
}

extern "C" int sl_stain_augment(const uint8_t* rgb, uint8_t* out, int n, int h, int w, const double* M,
                                const double* alpha_beta, int augment_background, const SlParams* params,
                                void* stream) {
    if (!rgb || !out || !M || !alpha_beta || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    SlParams p;
    sl_default_params(&p);
    if (!params_ok(params)) return SL_ERR_BADARG;
    if (params) p = *params;
    const long P = (long)h * w;
    if (P > (1L << 30)) return SL_ERR_BADARG;
    int parts = parts_for(P);
    const int max_grid = max_resident_grid();
    {   // persistent workgroups walk tiles x parts items: no more parts than it takes to give every workgroup ~4 items
        const long want = (4L * max_grid + n - 1) / n;
        if (parts > want) parts = (int)(want < 1 ? 1 : want);
    }
    const long items = (long)n * parts;
    const dim3 grid((unsigned)(items < max_grid ? items : max_grid)), block(kAugThreads);
    const uint32_t y_lim = y_limit_for_threshold(p.luminosity_threshold);
    hipStream_t s = (hipStream_t)stream;
    if (aligned4(rgb, P) && aligned4(out, P))
        hipLaunchKernelGGL((k_stain_augment<true>), grid, block, 0, s, rgb, out, (int)P, parts, (int)items, M, alpha_beta,
                           augment_background, y_lim, p.lasso_lambda);
    else
        hipLaunchKernelGGL((k_stain_augment<false>), grid, block, 0, s, rgb, out, (int)P, parts, (int)items, M, alpha_beta,
                           augment_background, y_lim, p.lasso_lambda);
    return launch_status();
}

extern "C" int sl_grayscale_augment(const uint8_t* rgb, uint8_t* out, int n, int h, int w, const double* alpha_beta,
                                   void* stream) {
    if (!rgb || !out || !alpha_beta || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    const long P = (long)h * w;
    if (P > (1L << 30)) return SL_ERR_BADARG;
    const int parts = parts_for(P);
    const dim3 grid((unsigned)((long)n * parts)), block(kWG);
    hipStream_t s = (hipStream_t)stream;
    if (aligned4(rgb, P) && aligned4(out, P)) hipLaunchKernelGGL((k_grayscale<true>), grid, block, 0, s, rgb, out, (int)P, parts, alpha_beta);
    else hipLaunchKernelGGL((k_grayscale<false>), grid, block, 0, s, rgb, out, (int)P, parts, alpha_beta);
    return launch_status();
}

extern "C" int sl_concentrations(const uint8_t* rgb, int n, int h, int w, const double* M, double lasso_lambda,
                                 float* C_out, void* stream) {
    if (!rgb || !M || !C_out || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    const long P = (long)h * w;
    if (P > (1L << 30)) return SL_ERR_BADARG;
    const int parts = parts_for(P);
    hipLaunchKernelGGL(k_concentrations, dim3((unsigned)((long)n * parts)), dim3(kWG), 0, (hipStream_t)stream, rgb,
                       (int)P, parts, M, lasso_lambda, C_out);
    return launch_status();
}

extern "C" int sl_tissue_mask(const uint8_t* rgb, int n, int h, int w, double luminosity_threshold,
                              uint8_t* mask_out, int64_t* counts, void* stream) {
    if (!rgb || n <= 0 || h <= 0 || w <= 0) return SL_ERR_BADARG;
    const long P = (long)h * w;
    if (P > (1L << 30)) return SL_ERR_BADARG;
    const int parts = parts_for(P);
    hipStream_t s = (hipStream_t)stream;
    if (counts) zero_async(counts, sizeof(int64_t) * (size_t)n, s);
    hipLaunchKernelGGL(k_tissue_mask, dim3((unsigned)((long)n * parts)), dim3(kWG), 0, s, rgb, (int)P, parts,
                       y_limit_for_threshold(luminosity_threshold), mask_out, (unsigned long long*)counts);
    return launch_status();
}
