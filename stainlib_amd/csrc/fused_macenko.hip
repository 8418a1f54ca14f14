// fused_macenko.hip -- the Macenko, 512-thread (two workgroups per CU) instantiations of the persistent fused kernel (stats_fused.hpp), in a translation unit of their own so
// that the three families compile side by side (each takes about a minute).
#include "stats_kernels.hpp"
#include "sl_host.hpp"

namespace sl {

void launch_fused_macenko(const FusedArgs& a, bool transform, bool aligned, unsigned grid, hipStream_t s) {
    const dim3 g(grid), b(kFusedThreads);
#define SL_GO(T, A) hipLaunchKernelGGL((k_fused<kMethodMacenko, T, A, kFusedThreads>), g, b, 0, s, a)
    if (transform) { if (aligned) SL_GO(true, true); else SL_GO(true, false); }
    else           { if (aligned) SL_GO(false, true); else SL_GO(false, false); }
#undef SL_GO
}

}  // namespace sl
