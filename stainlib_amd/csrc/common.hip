// common.hip -- host-only pieces of the C ABI: version, errors, defaults, the luminosity
// threshold -> integer limit translation.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "sl_host.hpp"

namespace sl {

// OpenCV RGB2Lab_b (modules/imgproc/src/color_lab.cpp): L8 = sat8((296*fY - 1336934 + 2^14) >> 15)
// with fY = LabCbrtTab_b[idx] = round(2^15 * f(idx/2040)), f the CIE Lab cube-root curve.
// The reference keeps pixels with L8/255.0 < threshold (stain_utils.py:42-43).
uint32_t y_limit_for_threshold(double luminosity_threshold) {
    const double thr = luminosity_threshold;
    const float scale = 1.0f / (255.0f * 8.0f);
    int best = -1;
    for (int i = 0; i < 3072; ++i) {
        const double x = (double)(scale * (float)i);
        const double f = x < 216.0 / 24389.0 ? x * (841.0 / 108.0) + 16.0 / 116.0 : std::cbrt(x);
        const long fY = std::lrint(f * 32768.0);
        long L = (296 * fY - 1336934 + 16384) >> 15;
        L = L < 0 ? 0 : (L > 255 ? 255 : L);
        if ((double)L / 255.0 < thr) best = i;   // tables are monotone: keep the last that passes
    }
    return (uint32_t)(best + 1) << 12;
}

// Resident persistent-sweep workgroups of the current device: 2 per compute unit (512 threads, 128 VGPRs, < 80 KB of LDS
// each; 256 CUs on MI355X -- queried, not assumed: a partitioned or smaller part gets the grid it can hold).  Falls back
// to 256 CUs when no device is visible (sl_workspace_bytes is host-only and must answer without one).
int max_resident_grid() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        cus = 256;
    }
    return 2 * cus;
}

// Zeroes as a KERNEL (zero_async): captured into a HIP graph, hipMemsetAsync nodes did not stay ordered with the kernels around them on
// this ROCm (round 6: replays of the pooled chain zeroed block counts after the sweep had begun to write them) -- every counter an
// entry point clears before a kernel adds into it goes through here, so that the calls stay capture-safe (engine.Graphed).
static __global__ __launch_bounds__(256) void k_zero_words(uint32_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
void zero_async(void* p, size_t bytes, hipStream_t s) {                   // p 4-byte aligned, bytes a multiple of 4
    const size_t words = bytes / 4;
    if (!words) return;
    size_t blocks = (words + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t*)p, words);
}

}  // namespace sl

extern "C" int sl_version(void) { return SL_VERSION; }

extern "C" void sl_default_params(SlParams* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->struct_size = (uint32_t)sizeof(SlParams);
    p->luminosity_threshold = 0.8;
    p->angular_percentile = 99.0;
    p->lasso_lambda = 0.01;
    p->dl_lambda = 0.1;
    p->dl_max_sweeps = 200;
    p->dl_tol = 1e-7;
}

extern "C" const char* sl_error_string(int code) {
    static thread_local char buf[96];
    switch (code) {
        case SL_OK: return "ok";
        case SL_ERR_BADARG: return "bad argument";
        case SL_ERR_WORKSPACE: return "workspace missing or too small";
        case SL_ERR_NODEVICE: return "no usable HIP device";
        default: break;
    }
    if (code <= SL_ERR_HIP_BASE) {
        std::snprintf(buf, sizeof(buf), "HIP error %d: %s", SL_ERR_HIP_BASE - code,
                      hipGetErrorString((hipError_t)(SL_ERR_HIP_BASE - code)));
        return buf;
    }
    return "unknown error";
}
