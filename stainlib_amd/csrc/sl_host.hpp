// sl_host.hpp -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stainlib_hip.h"

namespace sl {

inline int hip_err(hipError_t e) { return e == hipSuccess ? SL_OK : SL_ERR_HIP_BASE - (int)e; }

#define SL_HIP_TRY(expr)                                   \
    do {                                                   \
        hipError_t e__ = (expr);                           \
        if (e__ != hipSuccess) return ::sl::hip_err(e__);  \
    } while (0)

inline int launch_status() { return hip_err(hipGetLastError()); }

// An SlParams handed over by the caller: NULL (defaults) or a struct of THIS header's size (SlParams.struct_size, set by sl_default_params).
// (and with its enumerated fields inside their ranges: an unknown two_sweep used to run as "automatic" without a word)
inline bool params_ok(const SlParams* p) { return !p || (p->struct_size == (uint32_t)sizeof(SlParams) && p->two_sweep >= 0 && p->two_sweep <= 4); }

// Largest index into OpenCV's LabCbrtTab_b whose 8-bit L satisfies L/255.0 < threshold, +1,
// shifted to the fixed-point scale the kernels compare against (see is_tissue()).
// 0 means "no pixel is tissue".
uint32_t y_limit_for_threshold(double luminosity_threshold);

// resident persistent-sweep workgroups of the current device (2 per CU; see common.hip)
int max_resident_grid();

// zeroes `bytes` (a multiple of 4) at the 4-byte aligned p with a kernel on s: see common.hip (hipMemsetAsync is not capture-safe here)
void zero_async(void* p, size_t bytes, hipStream_t s);

// workspace of the Lab family (lab.hip)
size_t lab_workspace_bytes(int n_tiles);

inline bool aligned4(const void* p, long pixels_per_tile) {
    return ((uintptr_t)p & 3u) == 0 && (pixels_per_tile & 3) == 0;
}

// Workgroups a tile of P pixels is split into for the streaming sweeps: ~32 Ki pixels each,
// at least 1, so that a batch of >=32 tiles launches >>256 workgroups.
inline int parts_for(long P) {
    long p = (P + 32767) / 32768;
    return (int)(p < 1 ? 1 : p);
}

// Brackets one launch with two caller-provided events when the class is selected (see SlProfile).
struct ProfScope {
    SlProfile* p;
    hipStream_t s;
    bool on;
    ProfScope(SlProfile* prof, int cls, int tiles, hipStream_t stream) : p(prof), s(stream), on(false) {
        if (p && (p->mask & cls) && p->events && p->used + 2 <= p->capacity) {
            on = true;
            if (p->tags) p->tags[p->used / 2] = cls;
            if (p->tiles) p->tiles[p->used / 2] = tiles;
            (void)hipEventRecord((hipEvent_t)p->events[p->used], s);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord((hipEvent_t)p->events[p->used + 1], s);
            p->used += 2;
        }
    }
};

}  // namespace sl
