// stats_common.hpp -- constants, per-tile state, table views, the stratified sample, pseudo-angle and ordered keys.
// Part of stats_kernels.hpp (split by phase in round 4, no functional change); include that umbrella, not this file.
#pragma once
#include "../../include/stainlib_hip.h"
#include <type_traits>

#include "apply_kernels.hpp"

namespace sl {

constexpr int kMaxSample = 16384;   // samples per tile (<= P/64)
constexpr int kMinCapRaw = 65536;   // raw-pixel candidate capacity per tile and stage: max(this, P/6), set by the host
constexpr int kMinCapList = 16384;  // exact-key bracket members per list after the refine pass: max(this, P/8)
constexpr int kFinishThreads = 1024;
#define SL_FINISH_BOUNDS __launch_bounds__(kFinishThreads)
constexpr int kFusedThreads = 512;  // 2 resident workgroups per CU (<=128 VGPRs, <80 KB LDS each)
constexpr int kSweepThreads = 512;  // sweep kernels of the one-launch-per-phase schedule (same occupancy: 64 KB table each)
constexpr int kFusedTrip = 4;       // chunks per lane and sweep trip in the fused kernel (even)
constexpr int kBrkBatch = 8;       // sample words in flight per lane while the bracket keys are evaluated
constexpr int kPhaseTrip = 4;       // ... in the one-sweep kernels (32 Ki-pixel parts: 8 chunks per lane)
constexpr int kStageWave = 256;     // per-wave LDS staging entries for raw candidates (8 KB per 8 waves)
constexpr float kBracketZ = 6.0f;   // bracket half-width in standard deviations of the sample rank
constexpr float kAngleMargin = 2e-5f;  // safety margin of the cheap pseudo-angle test (keys carry ~1e-7)

struct TileState {
    // ---- after finish 1
    double n_tissue;
    double Vd[6];            // V[c][k], c = channel, k = 0 (largest eigenvalue), 1 (second)
    float Vf[6];
    float lo[2], hi[2];      // brackets of the current selection stage
    unsigned int pad0_;
    unsigned int n_raw;      // raw candidates appended (may exceed cap_raw => overflow)
    unsigned int overflow;   // a wave's staging buffer overflowed: the collected list is incomplete
    unsigned int pad_;
    // ---- after finish 2
    double M[6];
    // ---- after finish 3
    double maxC[2];
    int status;
    int fallbacks;           // order statistics that needed the slow exact path (diagnostics)
};

struct StatsArgs {
    const uint8_t* rgb;      // first tile of the group / batch
    int P;
    int parts;               // parts per tile (multi-kernel schedule)
    int n_items;             // tiles x parts of this group: the work list of the persistent sweep kernels
    int stride_log2;         // sampling stride = 1 << stride_log2 (>= 4: Macenko tiles below 1 Mpixel; 6 from 1 Mpixel on and for Vahadane)
    int n_sample;            // ceil(P / stride)
    float ylimf;             // tissue test threshold: y_lim - 2048 (see is_tissue_f)
    double lam;
    double pct;              // angular percentile
    double* partials;        // [tile][part][10]          (multi-kernel)
    uint32_t* sample;        // [tile][n_sample]          (multi-kernel)
    int cap_raw, cap_list;   // capacities of the two lists below (scale with the tile size)
    uint32_t* raw;           // [tile][cap_raw] raw candidate pixels (r | g<<8 | b<<16)
    float* cand;             // [tile][2][cap_list] bracket members (exact keys)
    TileState* state;        // [tile]                    (multi-kernel)
    // Vahadane (multi-kernel): partials are [tile][part][32] there
    double dl_lambda, dl_tol;
    int dl_max_sweeps;
    int tile0;               // first tile of the group within the batch (sweeps_out index)
    struct DictState* dstate;   // [tile]
    int32_t* sweeps_out;     // [n_tiles of the batch] (may be NULL)
    struct TileMerged* mstate;  // [tile] merged selection stage of the per-phase Macenko schedule
};


// Table access of the finish steps and key functors (few lookups, any layout): entry v of table f / g sits at LDS
// byte address base + v*stride + off_f / off_g (DS reads; a generic pointer would go the slower flat path).
struct TabView {
    uint32_t base; uint32_t stride, off_f, off_g;
#if defined(__HIP_DEVICE_COMPILE__)
    __device__ __forceinline__ float odf(uint32_t v) const { return *(SL_LDS const float*)(base + v * stride + off_f); }
    __device__ __forceinline__ float gam(uint32_t v) const { return *(SL_LDS const float*)(base + v * stride + off_g); }
#else
    float odf(uint32_t) const { return 0.0f; }
    float gam(uint32_t) const { return 0.0f; }
#endif
};
// LDS byte address of a pointer into shared memory: the low half of its flat address (the shared aperture sits in the
// high half).  Not the generic->local cast: that one carries a null check, which this hipcc mis-folds into an illegal
// v_cmp against src_shared_base when the pointer's origin is known.
__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)p; }
__device__ __forceinline__ TabView view_of(const RowTab& t) {
    const uint32_t c = (uint32_t)sizeof(TabEntry) * (threadIdx.x & (kTabCopies - 1));
    return TabView{lds_address(&t), (uint32_t)sizeof(TabEntry) * kTabCopies, c + 12u, c + 8u};
}
__device__ __forceinline__ TabView view_of_b(const RowTab& t) {                 // layout B: 32 x {gamma, od32}
    const uint32_t c = 8u * (threadIdx.x & 31u);
    return TabView{lds_address(&t), 256u, c + 4u, c};
}
// the 2 KB version for kernels that only run finish steps
struct SmallTab {
    float f[256], g[256];
    __device__ __forceinline__ void fill() {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) { f[i] = d_od_f32[i]; g[i] = (float)d_gamma[i]; }
    }
};
__device__ __forceinline__ TabView view_of(const SmallTab& t) { return TabView{lds_address(&t), 4u, 0u, 1024u}; }

// ---- stratified sample: one pixel per block of 2^stride_log2 pixels (2^cps_log2 chunks, cps_log2 >= 2, i.e. stride_log2 >= 4;
// make_layout asserts it) ----
// With cps_log2 < 6 a 64-chunk wave row holds several blocks (16 at cps_log2 = 2) that all share ONE draw -- the same chunk offset
// and pixel choice in each: the sample is still one pixel per block, only less randomised within a row than at cps_log2 >= 6.
// Which pixel is decided per HASH GROUP = the 64 chunks one wave covers with one load (or the whole block when
// it is larger): every lane of a wave row then shares the draw, so the sweep computes it on the scalar unit and
// pays one compare per chunk.  The draw picks a chunk of the block and pixel 0 or 3 of that chunk (the two a
// single shift extracts).  The sample only steers the brackets; results never depend on it.
__device__ __forceinline__ uint32_t sample_hash(uint32_t group) {
    uint32_t h = group * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    return h;
}
__device__ __forceinline__ int sample_group_shift(int cps_log2) { return cps_log2 > 6 ? cps_log2 : 6; }   // chunk index -> group
// pixel index kept for block b (may lie beyond the tile for the last block: then the entry is absent)
__device__ __forceinline__ long long sample_pixel(uint32_t b, int cps_log2) {
    const uint32_t chunk0 = b << cps_log2;
    const uint32_t h = sample_hash(chunk0 >> sample_group_shift(cps_log2));
    const uint32_t chunk = chunk0 + ((h >> 8) & ((1u << cps_log2) - 1u));
    return (long long)chunk * 4 + ((h >> 31) ? 3 : 0);
}
// only the LAST block of a tile can hold its draw beyond the tile: every other entry is present without looking
// (cps_log2 == kDenseCps: a DENSE sample -- the cluster sample of the two-sweep schedule, stats_twosweep.hpp -- whose n_sample entries all exist)
constexpr int kDenseCps = 64;
__device__ __forceinline__ bool sample_absent(int b, int cps_log2, int P) {
    if (cps_log2 >= kDenseCps) return false;
    return b >= ((P - 1) >> (cps_log2 + 2)) && sample_pixel((uint32_t)b, cps_log2) >= P;
}

// ---- cluster sample (round 5, the two-sweep Macenko schedule): a sample that must exist BEFORE the tile's first sweep cannot ride in
// a sweep, and 16 Ki scattered 4-byte reads fetch 16 Ki x 128 bytes = two thirds of a sweep.  So: n_lines blocks of 128 bytes (the
// fetch unit of the L2) spread evenly over the tile, the block of each sample line drawn by hash inside its stratum, and
// kClusterPx pixels of each (every kClusterStep-th: they span 36 of the block's 42): 2 048 lines x 8 pixels = 16 Ki entries for 256 KB
// of fetches (0.25 B/px of a 1024^2 tile).  Neighbouring pixels of real tissue are correlated: the brackets made of this sample are
// widened by a design effect (kClusterDeff) -- the sample steers, results never depend on it.
constexpr int kClusterLines = 2048, kClusterPx = 8, kClusterStep = 5;
__host__ __device__ inline int cluster_lines(long P) {
    const long nl = (3 * P) >> 7;
    return (int)(nl < 1 ? 1 : (nl < kClusterLines ? nl : kClusterLines));
}
__host__ __device__ inline int cluster_samples(long P) { return cluster_lines(P) * kClusterPx; }
// pixel index of entry b (always inside the tile)
__device__ __forceinline__ int cluster_pixel(int b, int P, int n_lines) {
    const int i = b / kClusterPx, j = b % kClusterPx;
    const long long nl = ((3ll * P) >> 7) < 1 ? 1 : ((3ll * P) >> 7);
    const long long l0 = (long long)i * nl / n_lines, l1 = (long long)(i + 1) * nl / n_lines;     // the stratum of line i (l1 > l0: n_lines <= nl)
    const uint32_t h = sample_hash((uint32_t)i);
    const long long line = l0 + (long long)((h >> 8) % (uint32_t)(l1 - l0));
    const long long px = (line * 128 + 2) / 3 + (long long)((h >> 28) % 3u) + kClusterStep * j;   // first whole pixel of the block + 0..2 + 5 j: <= +37, inside the block
    return (int)(px < (long long)P ? px : (long long)P - 1);
}

// Monotone surrogate of arctan2(y, x) on (-pi, pi]: y/(|x|+|y|) in [-1,1] for x >= 0, mirrored to
// (1,2] / [-2,-1) for x < 0.  One v_rcp instead of an atan2f per pixel; arctan2 itself is evaluated
// in binary64 only for the selected order statistics.
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
__device__ __forceinline__ float nan_f() { return __uint_as_float(0x7fc00000u); }
__device__ __forceinline__ double nan_d() { return __longlong_as_double(0x7ff8000000000000LL); }

// order-preserving 32-bit image of a binary32 key
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

}  // namespace sl
