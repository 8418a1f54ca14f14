// apply.hip -- the per-pixel "OD + reconstruction" family of sweeps (gfx950).
//
//   k_apply          normalizer.py:46-50   u8 -> OD -> 2-atom lasso -> rescale -> 255*exp(-C@Mt) -> u8
//   k_stain_augment  augmenter.py:428-449  same skeleton, C*alpha+beta on tissue, clip
//   k_concentrations stain_utils.py:69-78  materialise C (only for the Python attribute)
//   k_tissue_mask    stain_utils.py:32-48  materialise the mask / count it
//
// Roofline: HBM.  Algorithmic traffic of k_apply is 3 B read + 3 B written per pixel.
#pragma once
#include "sl_device.hpp"
#include <type_traits>

namespace sl {

// Tiles whose byte size is not a multiple of 4 (or whose batch is not 4-byte aligned) take the same dwordx3
// accesses at unaligned addresses (gfx950 serves them at full speed; checked on hardware).  Only the ragged last
// chunk of a tile needs care: its load starts early enough to stay inside the tile and is shifted into place,
// its store goes byte by byte.  No predicated load anywhere (see load_chunk_clamped).
struct __attribute__((packed, aligned(1))) ChunkU { uint32_t w0, w1, w2; };
typedef uint32_t sl_u32x3 __attribute__((ext_vector_type(3)));      // (16 bytes wide: never do pointer arithmetic on it)
typedef uint32_t sl_u32x3u __attribute__((ext_vector_type(3), aligned(1)));   // the same at any byte address
// STREAM = non-temporal accesses.  The fused kernel turns them on for tiles of kStreamBytes or more: every sweep streams
// the tile once and nothing is re-read before 3 MB x 512 workgroups have passed through the caches (measured at 1024^2:
// +2.5 %, the stores alone +0.5 %); smaller tiles stay cacheable -- a 256^2 tile (192 KB) is re-read by the next sweep
// from the L2.  The choice is made once per sweep (two instantiations), not per load (that cost more than it gave).
constexpr size_t kStreamBytes = (size_t)1 << 19;

template <bool ALIGNED, bool STREAM = false>
__device__ __forceinline__ Chunk load_chunk(const uint8_t* tile, size_t nbytes, int c) {
    if (ALIGNED) {
        if (STREAM) {
            const sl_u32x3 t = __builtin_nontemporal_load((SL_GLOBAL const sl_u32x3*)(as_global(tile) + (size_t)c * 12));
            return Chunk{t.x, t.y, t.z};
        }
        const sl_u32x3 t = *(SL_GLOBAL const sl_u32x3*)(as_global(tile) + (size_t)c * 12);
        return Chunk{t.x, t.y, t.z};
    } else {
        const size_t base = (size_t)c * 12;
        if (nbytes < 12) {                                   // a tile of fewer than 4 pixels (uniform)
            uint32_t w[3] = {0, 0, 0};
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const size_t at = base + i < nbytes - 1 ? base + i : nbytes - 1;
                const uint32_t b = as_global(tile)[at];
                w[i >> 2] |= (base + i < nbytes ? b : 0u) << (8 * (i & 3));
            }
            return Chunk{w[0], w[1], w[2]};
        }
        const size_t lim = nbytes - 12;
        const size_t at = base < lim ? base : lim;
        const sl_u32x3u uv = *(SL_GLOBAL const sl_u32x3u*)(as_global(tile) + at);
        const ChunkU u{uv.x, uv.y, uv.z};
        const uint32_t d = (uint32_t)(base - at);            // 0 except for a ragged last chunk (1..11 bytes too early)
        const uint32_t step = d >> 2, sh = 8u * (d & 3u);
        const uint32_t a0 = step == 0 ? u.w0 : (step == 1 ? u.w1 : u.w2);
        const uint32_t a1 = step == 0 ? u.w1 : (step == 1 ? u.w2 : 0u);
        const uint32_t a2 = step == 0 ? u.w2 : 0u;
        Chunk r;
        r.w0 = (uint32_t)((((unsigned long long)a1 << 32) | a0) >> sh);
        r.w1 = (uint32_t)((((unsigned long long)a2 << 32) | a1) >> sh);
        r.w2 = a2 >> sh;
        return r;                                            // bytes past the tile come out as zeros
    }
}

// Chunk c of [.., c1) for a sweep lane: lanes past the end re-read the last chunk (always a valid address, so the
// load needs no predicate and the sweep body stays free of divergent regions); they mask their results with
// `c < c1` themselves.  Requires c1 >= 1.
template <bool ALIGNED, bool STREAM = false>
__device__ __forceinline__ Chunk load_chunk_clamped(const uint8_t* tile, size_t nbytes, int c, int c1) {
    return load_chunk<ALIGNED, STREAM>(tile, nbytes, c < c1 ? c : c1 - 1);
}

template <bool ALIGNED, bool STREAM = false>
__device__ __forceinline__ void store_chunk(uint8_t* tile, size_t nbytes, int c, const Chunk& v) {
    if (ALIGNED) {
        if (STREAM) {
            sl_u32x3 t; t.x = v.w0; t.y = v.w1; t.z = v.w2;
            __builtin_nontemporal_store(t, (SL_GLOBAL sl_u32x3*)(as_global(tile) + (size_t)c * 12));
        } else {
            sl_u32x3 t; t.x = v.w0; t.y = v.w1; t.z = v.w2;
            *(SL_GLOBAL sl_u32x3*)(as_global(tile) + (size_t)c * 12) = t;
        }
    } else {
        const size_t base = (size_t)c * 12;
        if (base + 12 <= nbytes) {
            sl_u32x3u u; u.x = v.w0; u.y = v.w1; u.z = v.w2;
            *(SL_GLOBAL sl_u32x3u*)(as_global(tile) + base) = u;
        } else {
            for (int i = 0; i < 12; ++i)
                if (base + i < nbytes) as_global(tile)[base + i] = (uint8_t)chunk_byte(v, i);
        }
    }
}

// Truncating cast of normalizer.py:50 (`astype(np.uint8)`): toward zero, then modulo 256.
__device__ __forceinline__ uint32_t trunc_u8(float t) { return ((uint32_t)t) & 0xffu; }

// ---- the normalisation step shared by k_apply and sweep 4 of the fused kernel (bit-identical by construction) ----
// Issue count per pixel (the sweep is vector-issue bound): the lasso's max(min(a, s), 0) is a v_min_f32 and ONE v_fma_f32
// with the clamp modifier (fma_clamp01; round 2: three FMAs for s and a clamped v_min_f32) -- VOP3 clamp is [0, 1], so the
// concentrations are carried scaled by 2^-k, k chosen per tile from a
// bound on |a|, |s| over every optical density a byte can produce (exact: a power of two), and compensated in q.
// (Folding the factor 255 into the exponent -- 2^(e + log2 255), three multiplies fewer -- was measured and dropped: a
// pixel with zero concentrations must give EXACTLY 255 like the reference, and 2^(log2 255) comes out as 254.99998, which
// truncates to 254 on every background pixel.)
struct ApplyK {
    LassoK L;            // source stain matrix, weights scaled by 2^-k; affine parts VGPR-resident
    float q[2][3];       // -log2(e) * (maxC_tgt_i / maxC_src_i) * M_tgt[i][c] * 2^k, VGPR-resident
    bool fast;           // wave-uniform: g12 >= 0 and every q <= 0, i.e. 0 <= 255*2^e <= 255 for every pixel
};

constexpr double kOdMax = 5.541263545158426;      // -ln(1/255): the largest optical density of a byte

// 2^-k with 2^k >= every |a_i|, |s_i| the lasso can produce from byte optical densities (k in [0, 60])
__device__ __forceinline__ double lasso_unit_scale(const LassoK& k) {
    double b = 1.0;
    const float* rows[4] = {k.wa1, k.wa2, k.ws1, k.ws2};
    const float ks[4] = {k.ka1, k.ka2, k.ks1, k.ks2};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double v = kOdMax * (fabs((double)rows[r][0]) + fabs((double)rows[r][1]) + fabs((double)rows[r][2])) + fabs((double)ks[r]);
        b = v > b ? v : b;                                  // (NaN / inf from a singular matrix: the tile is reported, any scale does)
    }
    int e = 0;
    (void)frexp(b, &e);                                     // b = f * 2^e, f in [0.5, 1)  =>  b <= 2^e
    e = e < 0 ? 0 : (e > 60 ? 60 : e);
    return ldexp(1.0, -e);
}
__device__ __forceinline__ void scale_lasso(LassoK& k, float sc) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { k.wa1[c] *= sc; k.wa2[c] *= sc; k.ws1[c] *= sc; k.ws2[c] *= sc; }
    k.ka1 *= sc; k.ka2 *= sc; k.ks1 *= sc; k.ks2 *= sc;     // (g12, g22 only enter sign tests that are homogeneous in s1, s2)
}

// max(a + r m, 0) for results in [-1, 1]: v_fma_f32 with the VOP3 clamp modifier.  With m = min(a_other, 0) this is the lasso's
// max(0, min(a, s)) (sl_device.hpp: s = a + r a_other), one instruction where a separate s cost three FMAs and a clamped min.
__device__ __forceinline__ float fma_clamp01(float r, float m, float a) {
    float o;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(o) : "v"(r), "v"(m), "v"(a));
    return o;
}

// The lasso's min(a_other, 0) without a v_min_f32 (a 4-cycle instruction; multiplies and FMAs issue in 2): the concentrations are
// carried scaled into [-1, 1], so  -min(a, 0) = max(0, min(1, -a))  is ONE v_mul_f32 by -1 with the clamp modifier, and the FMA that
// follows takes r negated.  Bit-identical to the v_min form (a negation is exact; NaN clamps to 0 as fminf(NaN, 0) gives 0).
__device__ __forceinline__ float neg_part01(float a) {          // -min(a, 0) for |a| <= 1
    float o;
    asm("v_mul_f32 %0, -1.0, %1 clamp" : "=v"(o) : "v"(a));
    return o;
}
__device__ __forceinline__ float fnma_clamp01(float r, float m, float a) {     // max(a - r m, 0) for results in [-1, 1]
    float o;
    asm("v_fma_f32 %0, -%1, %2, %3 clamp" : "=v"(o) : "v"(r), "v"(m), "v"(a));
    return o;
}

__device__ __forceinline__ void apply_consts(const double* M_src, const double* maxC_src, const double* M_tgt,
                                             const double* maxC_tgt, double lam, ApplyK& K) {
    lasso_consts(M_src, lam, K.L);
    const double sc = lasso_unit_scale(K.L), inv = 1.0 / sc;       // powers of two
    scale_lasso(K.L, (float)sc);
    bool nonpos = true;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double ratio = maxC_tgt[i] / maxC_src[i];                          // normalizer.py:48
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float q = (float)(-1.4426950408889634 * ratio * M_tgt[3 * i + c] * inv);
            nonpos = nonpos & (q <= 0.0f);
            K.q[i][c] = in_vgpr(uni(q));
        }
    }
    K.fast = (bool)__builtin_amdgcn_readfirstlane((int)(nonpos & (K.L.g12 >= 0.0f)));
    vgpr(K.L);
}

// 255 * exp(-C @ M_tgt) of one pixel, before the cast
template <bool FAST>
__device__ __forceinline__ void apply_px(const ApplyK& K, float x, float y, float z, float (&t)[3]) {
    float c1, c2;
    if (FAST) {                                    // g12 >= 0: branch-free lasso (see lasso2)
        float a1, a2;
        lasso_interior(K.L, x, y, z, a1, a2);
        c1 = fnma_clamp01(K.L.r1, neg_part01(a2), a1);
        c2 = fnma_clamp01(K.L.r2, neg_part01(a1), a2);
    } else {
        lasso2(K.L, x, y, z, c1, c2);
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) t[ch] = 255.0f * __builtin_amdgcn_exp2f(fmaf(c1, K.q[0][ch], c2 * K.q[1][ch]));
}

// The truncating cast of normalizer.py:50 for 12 values in [0, 255] -> one chunk.  v_cvt_pk_u8_f32 converts,
// saturates and inserts the byte in one instruction but rounds per MODE.fp_round, so the block switches the
// binary32 rounding mode to toward-zero around the 12 conversions (measured: tools/ubench_issue.hip).
__device__ __forceinline__ Chunk pack_trunc_fast(const float (&t)[12]) {
    Chunk o;
    asm volatile(
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
        "s_nop 0\n\t"
        "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\t"
        "v_cvt_pk_u8_f32 %1, %7, 0, 0\n\t"
        "v_cvt_pk_u8_f32 %2, %11, 0, 0\n\t"
        "v_cvt_pk_u8_f32 %0, %4, 1, %0\n\t"
        "v_cvt_pk_u8_f32 %1, %8, 1, %1\n\t"
        "v_cvt_pk_u8_f32 %2, %12, 1, %2\n\t"
        "v_cvt_pk_u8_f32 %0, %5, 2, %0\n\t"
        "v_cvt_pk_u8_f32 %1, %9, 2, %1\n\t"
        "v_cvt_pk_u8_f32 %2, %13, 2, %2\n\t"
        "v_cvt_pk_u8_f32 %0, %6, 3, %0\n\t"
        "v_cvt_pk_u8_f32 %1, %10, 3, %1\n\t"
        "v_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
        : "=&v"(o.w0), "=&v"(o.w1), "=&v"(o.w2)
        : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]), "v"(t[8]), "v"(t[9]),
          "v"(t[10]), "v"(t[11]));
    return o;
}
// general case (a target matrix with negative entries can push values past 255): toward zero, then modulo 256
__device__ __forceinline__ Chunk pack_trunc_general(const float (&t)[12]) {
    uint32_t ob[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) ob[i] = trunc_u8(t[i]);
    Chunk o;
    o.w0 = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
    o.w1 = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
    o.w2 = ob[8] | (ob[9] << 8) | (ob[10] << 16) | (ob[11] << 24);
    return o;
}

// The apply sweep over chunks [c0, c1) of one tile with `nthreads` cooperating threads, table values from the
// row table.  Two chunks per trip; the next trip's chunks are in flight during the current one and the 12 table
// gathers of a chunk are issued one chunk ahead of its arithmetic.
template <bool ALIGNED, bool FAST, class TR, bool STREAM = false>
__device__ __forceinline__ void apply_sweep(const uint8_t* src, uint8_t* dst, int P, int c0, int c1, int t, int nthreads,
                                            const TR& T, const ApplyK& K) {
    const size_t nbytes = (size_t)P * 3;
    struct G { float v[12]; };
    auto fetch = [&](int cc) { return load_chunk_clamped<ALIGNED, STREAM>(src, nbytes, cc, c1); };
    auto gather = [&](const Chunk& ch) {
        G g;
#pragma unroll
        for (int i = 0; i < 12; ++i) g.v[i] = T.odf(T.addr(ch, i));
        return g;
    };
    auto compute = [&](const G& g, int cc) {
        float tv[12];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            float v[3];
            apply_px<FAST>(K, g.v[3 * px], g.v[3 * px + 1], g.v[3 * px + 2], v);
            tv[3 * px] = v[0]; tv[3 * px + 1] = v[1]; tv[3 * px + 2] = v[2];
        }
        const Chunk o = FAST ? pack_trunc_fast(tv) : pack_trunc_general(tv);
        if (cc < c1) store_chunk<ALIGNED, STREAM>(dst, nbytes, cc, o);
    };
    constexpr int N = 4;                                       // chunks per lane and trip; the next trip is in flight
    Chunk cur[N], nx[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { cur[k] = fetch(c0 + t + k * nthreads); nx[k] = fetch(c0 + t + (N + k) * nthreads); }
    G g[2];
    g[0] = gather(cur[0]);
    for (int c = c0 + t; c < c1; c += N * nthreads) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (k + 1 < N) {
                g[(k + 1) & 1] = gather(cur[k + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) { cur[j] = nx[j]; nx[j] = fetch(c + (2 * N + j) * nthreads); }
                g[0] = gather(cur[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            compute(g[k & 1], c + k * nthreads);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

constexpr int kU = 4;       // chunks in flight per lane per trip (plain sweeps: 4 x 12 B loads issued back to back)
constexpr int kUApply = 2;  // k_apply: 2 chunks per trip with the following trip prefetched

template <bool ALIGNED, bool PREQ>
static __global__ __launch_bounds__(kWG) void k_apply(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ out,
                                               int P, int parts, const double* __restrict__ M_src,
                                               const double* __restrict__ maxC_src,
                                               const double* __restrict__ M_tgt,
                                               const double* __restrict__ maxC_tgt, double lam,
                                               float* __restrict__ prequant) {
    __shared__ float s_od[256 * kRepl];
    fill_od_lut(s_od);
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    const int tid = threadIdx.x;
    const uint32_t lane32 = tid & (kRepl - 1);   // which LDS copy of the table this lane reads

    // per-tile constants, computed redundantly in binary64 by every lane
    ApplyK K;
    apply_consts(M_src + 6 * (size_t)tile, maxC_src + 2 * (size_t)tile, M_tgt, maxC_tgt, lam, K);
    __syncthreads();

    const size_t nbytes = (size_t)P * 3;
    const uint8_t* src = rgb + (size_t)tile * nbytes;
    uint8_t* dst = out + (size_t)tile * nbytes;
    const int nch = (P + 3) >> 2;
    const int span = (nch + parts - 1) / parts;
    const int c0 = part * span;
    const int c1 = min(nch, c0 + span);

    // A tile whose fit failed (empty tissue mask / degenerate covariance: M is NaN; a zero 99th-percentile concentration,
    // which the reference divides by, normalizer.py:48) is passed through unchanged; the caller sees why in status[].
    // (Block-uniform branch.)
    if (!(M_src[6 * (size_t)tile] == M_src[6 * (size_t)tile]) || !(maxC_src[2 * (size_t)tile] > 0.0) || !(maxC_src[2 * (size_t)tile + 1] > 0.0)) {
        for (int c = c0 + tid; c < c1; c += kWG) store_chunk<ALIGNED>(dst, nbytes, c, load_chunk<ALIGNED>(src, nbytes, c));
        return;
    }

    // software pipeline: the next trip's chunks are requested before this trip's arithmetic (measured +5 %)
    auto fetch = [&](int cc) { return load_chunk_clamped<ALIGNED, true>(src, nbytes, cc, c1); };    // single pass: non-temporal
    auto sweep = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        Chunk nxt[kUApply];
#pragma unroll
        for (int u = 0; u < kUApply; ++u) nxt[u] = fetch(c0 + tid + u * kWG);
        for (int c = c0 + tid; c < c1; c += kWG * kUApply) {
            Chunk in[kUApply];
#pragma unroll
            for (int u = 0; u < kUApply; ++u) {
                in[u] = nxt[u];
                nxt[u] = fetch(c + (kUApply + u) * kWG);
            }
#pragma unroll
            for (int u = 0; u < kUApply; ++u) {
                const int cc = c + u * kWG;
                float t[12];
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    const float x = lut(s_od, chunk_byte(in[u], 3 * px + 0), lane32);
                    const float y = lut(s_od, chunk_byte(in[u], 3 * px + 1), lane32);
                    const float z = lut(s_od, chunk_byte(in[u], 3 * px + 2), lane32);
                    float v[3];
                    apply_px<FAST>(K, x, y, z, v);
                    t[3 * px] = v[0]; t[3 * px + 1] = v[1]; t[3 * px + 2] = v[2];
                    if (PREQ) {
                        const size_t pix = (size_t)cc * 4 + px;
                        if (cc < c1 && pix < (size_t)P) {
                            float* pq = prequant + ((size_t)tile * P + pix) * 3;
                            pq[0] = v[0]; pq[1] = v[1]; pq[2] = v[2];
                        }
                    }
                }
                const Chunk o = FAST ? pack_trunc_fast(t) : pack_trunc_general(t);
                if (cc < c1) store_chunk<ALIGNED, true>(dst, nbytes, cc, o);
            }
        }
    };
    if (K.fast) sweep(std::true_type{}); else sweep(std::false_type{});
}

// StainAugmentor.pop: own stain matrix both ways, affine on the concentrations of tissue pixels
// (augmenter.py:435-443), clip (augmenter.py:447).  Persistent 512-thread workgroups over (tile, part) items with the
// row table in layout B: one conflict-free ds_read_b64 {gamma, od32} per byte, gathers issued one chunk ahead of the
// arithmetic, the next trip's chunks in flight (the structure of apply_sweep).
struct AugmentK {
    LassoK L;            // VGPR-resident, weights scaled by 2^-k (see ApplyK)
    float q[2][3];       // -log2(e) * M[i][c] * 2^k
    float al0, be0, al1, be1, ylimf;            // be_i scaled by 2^-k
};

template <bool ALIGNED, bool ALL, bool FAST, class TR>
__device__ __forceinline__ void augment_sweep(const uint8_t* src, uint8_t* dst, int P, int c0, int c1, int t, int nthreads,
                                              const TR& T, const AugmentK& K) {
    const size_t nbytes = (size_t)P * 3;
    struct G { float2 v[12]; };
    auto fetch = [&](int cc) { return load_chunk_clamped<ALIGNED, true>(src, nbytes, cc, c1); };
    auto gather = [&](const Chunk& ch) {
        G g;
#pragma unroll
        for (int i = 0; i < 12; ++i) g.v[i] = T.gam_odf(T.addr(ch, i));
        return g;
    };
    auto compute = [&](const G& g, int cc) {
        float tv[12];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const float2 er = g.v[3 * px], eg = g.v[3 * px + 1], eb = g.v[3 * px + 2];     // x = gamma, y = od32
            float a1, a2;
            if (FAST) {                                    // g12 >= 0: branch-free lasso (see lasso2, apply_px)
                float i1, i2;
                lasso_interior(K.L, er.y, eg.y, eb.y, i1, i2);
                a1 = fnma_clamp01(K.L.r1, neg_part01(i2), i1);
                a2 = fnma_clamp01(K.L.r2, neg_part01(i1), i2);
            } else {
                lasso2(K.L, er.y, eg.y, eb.y, a1, a2);
            }
            if (ALL) {
                a1 = fmaf(a1, K.al0, K.be0);
                a2 = fmaf(a2, K.al1, K.be1);
            } else {
                const bool tissue = is_tissue_f(er.x, eg.x, eb.x, K.ylimf);
                a1 = tissue ? fmaf(a1, K.al0, K.be0) : a1;
                a2 = tissue ? fmaf(a2, K.al1, K.be1) : a2;
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) tv[3 * px + ch] = 255.0f * __builtin_amdgcn_exp2f(fmaf(a1, K.q[0][ch], a2 * K.q[1][ch]));
        }
        const Chunk o = pack_trunc_fast(tv);        // values are >= 0; > 255 saturates = np.clip(.., 0, 255)
        if (cc < c1) store_chunk<ALIGNED, true>(dst, nbytes, cc, o);
    };
    constexpr int N = 2;                                       // chunks per lane and trip; the next trip is in flight
    Chunk cur[N], nx[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { cur[k] = fetch(c0 + t + k * nthreads); nx[k] = fetch(c0 + t + (N + k) * nthreads); }
    G g[2];
    g[0] = gather(cur[0]);
    for (int c = c0 + t; c < c1; c += N * nthreads) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (k + 1 < N) {
                g[(k + 1) & 1] = gather(cur[k + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) { cur[j] = nx[j]; nx[j] = fetch(c + (2 * N + j) * nthreads); }
                g[0] = gather(cur[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            compute(g[k & 1], c + k * nthreads);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

constexpr int kAugThreads = 512;

template <bool ALIGNED>
static __global__ __launch_bounds__(kAugThreads, 4) void k_stain_augment(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ out,
                                                       int P, int parts, int n_items, const double* __restrict__ M,
                                                       const double* __restrict__ alpha_beta, int augment_background,
                                                       uint32_t y_lim, double lam) {
    __shared__ RowTab s_tab;
    s_tab.fill_b();
    __syncthreads();
    const TabReaderB T = TabReaderB::make(s_tab);
    const int tid = threadIdx.x;
    const size_t nbytes = (size_t)P * 3;
    const int nch = (P + 3) >> 2;
    const int span = (((nch + parts - 1) / parts) + 63) & ~63;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int tile = item / parts, part = item % parts;
        // per-tile constants live in VGPRs (a VALU op with an SGPR operand issues at half rate on gfx950)
        AugmentK K;
        lasso_consts(M + 6 * (size_t)tile, lam, K.L);
        const double sc = lasso_unit_scale(K.L), inv = 1.0 / sc;      // powers of two (see ApplyK)
        scale_lasso(K.L, (float)sc);
        vgpr(K.L);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) K.q[i][c] = in_vgpr((float)(-1.4426950408889634 * M[6 * (size_t)tile + 3 * i + c] * inv));
        K.al0 = in_vgpr((float)alpha_beta[4 * (size_t)tile + 0]); K.be0 = in_vgpr((float)(alpha_beta[4 * (size_t)tile + 1] * sc));
        K.al1 = in_vgpr((float)alpha_beta[4 * (size_t)tile + 2]); K.be1 = in_vgpr((float)(alpha_beta[4 * (size_t)tile + 3] * sc));
        K.ylimf = in_vgpr((float)y_lim - 2048.0f);
        const int c0 = min(nch, part * span), c1 = min(nch, c0 + span);
        if (c0 >= c1) continue;
        const uint8_t* src = rgb + (size_t)tile * nbytes;
        uint8_t* dst = out + (size_t)tile * nbytes;
        const bool fast = (bool)__builtin_amdgcn_readfirstlane((int)(K.L.g12 >= 0.0f));    // every real stain matrix
        if (augment_background) {
            if (fast) augment_sweep<ALIGNED, true, true>(src, dst, P, c0, c1, tid, kAugThreads, T, K);
            else augment_sweep<ALIGNED, true, false>(src, dst, P, c0, c1, tid, kAugThreads, T, K);
        } else {
            if (fast) augment_sweep<ALIGNED, false, true>(src, dst, P, c0, c1, tid, kAugThreads, T, K);
            else augment_sweep<ALIGNED, false, false>(src, dst, P, c0, c1, tid, kAugThreads, T, K);
        }
    }
}

// GrayscaleAugmentor.pop (augmenter.py:390-401): skimage rgb2gray on the uint8 image (x * (1/255) in binary64, weights
// 0.2125 / 0.7154 / 0.0721), * alpha + beta, clip [0,1], * 255, truncate, replicated on three channels.  Binary64
// like the reference (the pass is HBM bound: 6 B/px against ~10 binary64 ops), so the truncation sees the same value.
template <bool ALIGNED>
static __global__ __launch_bounds__(kWG) void k_grayscale(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ out, int P,
                                                          int parts, const double* __restrict__ alpha_beta) {
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    const double alpha = alpha_beta[2 * (size_t)tile], beta = alpha_beta[2 * (size_t)tile + 1];
    const size_t nbytes = (size_t)P * 3;
    const uint8_t* src = rgb + (size_t)tile * nbytes;
    uint8_t* dst = out + (size_t)tile * nbytes;
    const int nch = (P + 3) >> 2;
    const int span = (nch + parts - 1) / parts;
    const int c0 = part * span, c1 = min(nch, c0 + span);
    const double k = 1.0 / 255;
    for (int c = c0 + (int)threadIdx.x; c < c1; c += kWG) {
        const Chunk in = load_chunk<ALIGNED>(src, nbytes, c);
        uint32_t v[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const double r = (double)chunk_byte(in, 3 * px) * k, g = (double)chunk_byte(in, 3 * px + 1) * k,
                         b = (double)chunk_byte(in, 3 * px + 2) * k;
            double y = __dadd_rn(__dadd_rn(__dmul_rn(r, 0.2125), __dmul_rn(g, 0.7154)), __dmul_rn(b, 0.0721));   // no contraction
            y = __dadd_rn(__dmul_rn(y, alpha), beta);
            y = fmin(fmax(y, 0.0), 1.0);
            v[px] = (uint32_t)fmin(fmax(y * 255.0, 0.0), 255.0);
        }
        Chunk o;
        o.w0 = v[0] * 0x010101u | (v[1] << 24);
        o.w1 = (v[1] * 0x0101u) | (v[2] * 0x01010000u);
        o.w2 = v[2] | (v[3] * 0x01010100u);
        store_chunk<ALIGNED>(dst, nbytes, c, o);
    }
}

static __global__ __launch_bounds__(kWG) void k_concentrations(const uint8_t* __restrict__ rgb, int P, int parts,
                                                        const double* __restrict__ M, double lam,
                                                        float* __restrict__ C_out) {
    __shared__ float s_od[256 * kRepl];
    fill_od_lut(s_od);
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    const uint32_t lane32 = threadIdx.x & (kRepl - 1);
    LassoK L;
    lasso_consts(M + 6 * (size_t)tile, lam, L);
    uni(L);
    __syncthreads();
    const int span = (P + parts - 1) / parts;
    const int p0 = part * span, p1 = min(P, p0 + span);
    const uint8_t* src = rgb + (size_t)tile * P * 3;
    for (int p = p0 + threadIdx.x; p < p1; p += kWG) {
        float a, b;
        lasso2(L, lut(s_od, src[3 * (size_t)p], lane32), lut(s_od, src[3 * (size_t)p + 1], lane32),
               lut(s_od, src[3 * (size_t)p + 2], lane32), a, b);
        reinterpret_cast<float2*>(C_out)[(size_t)tile * P + p] = make_float2(a, b);
    }
}

static __global__ __launch_bounds__(kWG) void k_tissue_mask(const uint8_t* __restrict__ rgb, int P, int parts,
                                                     uint32_t y_lim, uint8_t* __restrict__ mask_out,
                                                     unsigned long long* __restrict__ counts) {
    __shared__ uint32_t s_g[256 * kRepl];
    __shared__ unsigned long long s_cnt;
    fill_gamma_lut(s_g);
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    const uint32_t lane32 = threadIdx.x & (kRepl - 1);
    const int span = (P + parts - 1) / parts;
    const int p0 = part * span, p1 = min(P, p0 + span);
    const uint8_t* src = rgb + (size_t)tile * P * 3;
    unsigned long long n = 0;
    for (int p = p0 + threadIdx.x; p < p1; p += kWG) {
        const bool t = is_tissue(lut(s_g, src[3 * (size_t)p], lane32), lut(s_g, src[3 * (size_t)p + 1], lane32),
                                 lut(s_g, src[3 * (size_t)p + 2], lane32), y_lim);
        if (mask_out) mask_out[(size_t)tile * P + p] = t ? 1 : 0;
        n += t ? 1 : 0;
    }
    n = wave_sum(n);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, n);
    __syncthreads();
    if (threadIdx.x == 0 && counts) atomicAdd(&counts[tile], s_cnt);
}

}  // namespace sl

