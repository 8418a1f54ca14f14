// stubs.hip -- entry points whose kernels have not landed yet (removed as they do).
#include "sl_host.hpp"
extern "C" int sl_vahadane_fit(const uint8_t*, int, int, int, const SlParams*, double*, double*, int32_t*, int32_t*, void*, size_t, void*) { return SL_ERR_BADARG; }
extern "C" int sl_vahadane_transform(const uint8_t*, uint8_t*, int, int, int, const SlParams*, const double*, const double*, double*, double*, int32_t*, void*, size_t, void*) { return SL_ERR_BADARG; }
