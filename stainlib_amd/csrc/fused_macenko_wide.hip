// fused_macenko_wide.hip -- the Macenko, 1024-thread (one workgroup per CU) instantiations of the persistent fused kernel (stats_fused.hpp), in a translation unit of their own so
// that the three families compile side by side (each takes about a minute).
#include "stats_kernels.hpp"
#include "sl_host.hpp"

namespace sl {

void launch_fused_macenko_wide(const FusedArgs& a, bool transform, bool aligned, unsigned grid, hipStream_t s) {
    const dim3 g(grid), b(2 * kFusedThreads);
#define SL_GO(T, A) hipLaunchKernelGGL((k_fused<kMethodMacenko, T, A, 2 * kFusedThreads>), g, b, 0, s, a)
    if (transform) { if (aligned) SL_GO(true, true); else SL_GO(true, false); }
    else           { if (aligned) SL_GO(false, true); else SL_GO(false, false); }
#undef SL_GO
}

}  // namespace sl
