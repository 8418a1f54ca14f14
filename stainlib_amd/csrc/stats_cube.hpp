// stats_cube.hpp -- the colour-cube pre-filter of the merged selection sweep (round 4).
// Part of stats_kernels.hpp; include that umbrella, not this file.
//
// The merged selection sweep (select_sweep<kStageMerged>) spends ~30 vector instructions per pixel on proving ~95 % of the pixels
// "plain" (angle key strictly between the two angular brackets, both concentrations below their brackets under every stain matrix
// of the box).  All of its tests are functions of the pixel's COLOUR alone, and every quantity in them is a linear functional of
// the three optical densities, which are monotone in the bytes.  So finish 1 can decide once per tile, for each of the 32^3 cells
// of the colour cube (8 byte values per channel and cell), whether EVERY colour of the cell is provably plain:
//   * a linear functional sum_c k_c od_c over a cell is bounded by choosing per channel the end of the cell's od range that k_c's
//     sign asks for -- separable, so three 32-entry tables per functional give the bound of a cell in two additions;
//   * no tissue in the cell  <=>  the luminance sum at the cell's LOWEST bytes is already past the threshold (exact integers);
//   * the cone test  hi0m d < y < lo1m d,  d = x + |y|  is piecewise linear: with L+- = (1 -+ h) y - h x one has
//     y - h d = (y >= 0 ? L+ : L-), which is min(L+, L-) for h >= 0 and max(L+, L-) for h < 0 -- bounded from the four per-cell
//     extremes of L+, L- (exactly for the concave case, conservatively for the convex one);
//   * the concentration tests  a_i + eps_i (|a_1| + |a_2|) < thr_i  are bounded with max a_i and max |a_j| over the cell.
// The result is a 4 KB bit mask (bit = 1: the cell holds a colour that is NOT provably plain: "ambiguous"), kept in LDS where the
// finish steps' histogram lives between them.  The sweep then costs, per pixel, the cell address, one LDS read and a bit test;
// the pixels of ambiguous cells (10 % of an i.i.d. tile, 8 % with white background, 16 % of real tissue, 45 % of the spatially
// smooth synthetic tiles: tools/cube_prefilter_estimate.py) are staged per wave in LDS and re-tested EXACTLY, 64 at a time with all
// lanes busy, by the very test of select_sweep; what that flags goes to the raw list as before.  Margins: the cell bounds carry
// twice the per-pixel test's angular margin plus an absolute allowance for their own binary32 rounding, so a pixel of a "plain" cell
// is plain for the per-pixel test and for the exact keys of the finish a fortiori.  Results never depend on the mask: a cell
// wrongly called ambiguous costs time only, and the finish steps verify counts and brackets as before.
// Finish 1 measures the share of sample pixels in ambiguous cells and keeps the per-pixel sweep when it exceeds kCubeMaxShare.
#pragma once
#include "stats_finish.hpp"

namespace sl {

constexpr int kCubeFn = 10;            // functionals per channel and cell index (see cube_tables)
constexpr int kCubeTabFloats = kCubeFn * 3 * 32;
constexpr int kCubeWords = 1024;       // 32 x 32 words (g5, b5), bit = r5
constexpr int kCubeMaxSharePct = 40;   // above this share of sample pixels in ambiguous cells the per-pixel sweep is cheaper
constexpr int kCubeShareStep = 16;     // the share is estimated on every 16th row of the sample (1 024 pixels of 16 Ki: +-1 %)
constexpr int kCubeRing = kStageWave;  // per-wave staging of ambiguous pixels = the wave's whole 1 KB: up to 127 left over + 2 pixel rows of 64

// word address (bytes) of a pixel's cell inside the mask and its bit: pixel = r | g << 8 | b << 16
// (The LDS bank of a mask read is g5.  Measured: an XOR swizzle of the word index with b5 -- banks spread even where the green bytes
// crowd into few values, one instruction more per pixel -- changes nothing: 1.62-1.65 ms either way, real tissue 1.72 vs 1.75.)
__device__ __forceinline__ uint32_t cube_word_offset(uint32_t p) { return ((p >> 9) & 0x7Cu) | ((p >> 12) & 0xF80u); }
__device__ __forceinline__ uint32_t cube_bit(uint32_t p) { return (p >> 3) & 31u; }

struct CubeConsts { float hi0m, lo1m, s_ang, kt[2], eps[2], thr[2]; };

// Per-channel, per-cell-index extremes of the ten functionals, ctab[f][ch][k] (LDS, kCubeTabFloats floats):
//   0 luminance sum at the cell's lowest byte (exact integer)      1 min x
//   2 min L+(hi0m)   3 min L-(hi0m)   4 max L+(lo1m)   5 max L-(lo1m)     (L+-(h) = (1 -+ h) y - h x)
//   6 min a1   7 max a1   8 min a2   9 max a2                              (without the constant kt)
// The workgroup's threads share the entries; all threads get the constants.  Ends WITHOUT a barrier.
__device__ __forceinline__ CubeConsts cube_tables(const TabView& tab, const float* Vf, float hi0, float lo1, const MergedConc& mk, float* ctab, int tid) {
    CubeConsts cc;
    cc.hi0m = hi0 + 2.0f * kAngleMargin;
    cc.lo1m = lo1 - 2.0f * kAngleMargin;
    float sum_abs = 0.0f;
    for (int i = 0; i < 6; ++i) sum_abs += fabsf(Vf[i]);
    cc.s_ang = 8e-6f * ((float)kOdMax * 2.0f * sum_abs + 1.0f);         // |1 -+ h| <= 2, |h| <= 2 on pseudo-angles
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        cc.kt[i] = mk.kt[i]; cc.eps[i] = mk.eps[i];
        float ca = 0.0f;
        for (int ch = 0; ch < 3; ++ch) ca += fabsf(fmaf(mk.u[i][1], Vf[2 * ch + 1], mk.u[i][0] * Vf[2 * ch]));
        cc.thr[i] = mk.thr[i] - 8e-6f * ((float)kOdMax * ca * (1.0f + 2.0f * mk.eps[i]) + fabsf(mk.kt[i]) + 1.0f);
    }
    for (int e = tid; e < kCubeTabFloats; e += (int)blockDim.x) {
        const int f = e / 96, ch = (e / 32) % 3, k = e & 31;
        const float v0 = Vf[2 * ch], v1 = Vf[2 * ch + 1];
        const float od_lo = tab.odf((uint32_t)(8 * k + 7)), od_hi = tab.odf((uint32_t)(8 * k));   // od falls with the byte
        float out;
        if (f == 0) {
            const float w = ch == 0 ? 871.0f : (ch == 1 ? 2929.0f : 296.0f);
            out = w * tab.gam((uint32_t)(8 * k));                       // gamma rises with the byte: the cell's smallest
        } else {
            float c;
            switch (f) {
                case 1: c = v0; break;
                case 2: c = (1.0f - cc.hi0m) * v1 - cc.hi0m * v0; break;
                case 3: c = (1.0f + cc.hi0m) * v1 - cc.hi0m * v0; break;
                case 4: c = (1.0f - cc.lo1m) * v1 - cc.lo1m * v0; break;
                case 5: c = (1.0f + cc.lo1m) * v1 - cc.lo1m * v0; break;
                case 6: case 7: c = fmaf(mk.u[0][1], v1, mk.u[0][0] * v0); break;
                default: c = fmaf(mk.u[1][1], v1, mk.u[1][0] * v0); break;
            }
            const bool want_max = (f == 4) | (f == 5) | (f == 7) | (f == 9);
            const float a = c * od_lo, b = c * od_hi;
            out = want_max ? fmaxf(a, b) : fminf(a, b);
        }
        ctab[e] = out;
    }
    return cc;
}

// one cell's verdict from the per-channel tables (what cube_mask evaluates for every cell, here for the cell of ONE pixel)
__device__ __forceinline__ bool cube_cell_plain(const float* ctab, const CubeConsts& cc, float ylimf, uint32_t p) {
    const int r5 = (int)((p >> 3) & 31u), g5 = (int)((p >> 11) & 31u), b5 = (int)((p >> 19) & 31u);
    float v[kCubeFn];
#pragma unroll
    for (int f = 0; f < kCubeFn; ++f)
        v[f] = ctab[(f * 3 + 1) * 32 + g5] + ctab[(f * 3 + 2) * 32 + b5] + (f >= 6 ? cc.kt[(f - 6) >> 1] : 0.0f) + ctab[(f * 3) * 32 + r5];
    const bool no_tissue = v[0] >= ylimf;
    const float t0lb = cc.hi0m >= 0.0f ? fminf(v[2], v[3]) : fmaxf(v[2], v[3]);
    const float t1ub = cc.lo1m >= 0.0f ? fminf(v[4], v[5]) : fmaxf(v[4], v[5]);
    const bool cone = fminf(fminf(v[1], t0lb), -t1ub) > cc.s_ang;
    const float sa = fmaxf(fabsf(v[6]), fabsf(v[7])) + fmaxf(fabsf(v[8]), fabsf(v[9]));
    const bool conc1 = fmaf(cc.eps[0], sa, v[7]) < cc.thr[0], conc2 = fmaf(cc.eps[1], sa, v[9]) < cc.thr[1];
    return (no_tissue || cone) && conc1 && conc2;
}

// The mask: thread t fills words t, t + NT, ... (word = g5 | b5 << 5, bit = r5).  Call after a barrier behind cube_tables.
// (Fully unrolled by the compiler, ~45 us per tile next to a sweeping partner; a rolled inner loop waits for its ten LDS reads in
// every iteration: 310 us.)
template <int NT>
__device__ __forceinline__ void cube_mask(const float* ctab, const CubeConsts& cc, float ylimf, uint32_t* bits, int tid) {
    const bool h_pos = cc.hi0m >= 0.0f, l_pos = cc.lo1m >= 0.0f;       // block-uniform
    for (int w = tid; w < kCubeWords; w += NT) {
        const int g5 = w & 31, b5 = w >> 5;
        float part[kCubeFn];
#pragma unroll
        for (int f = 0; f < kCubeFn; ++f) part[f] = ctab[(f * 3 + 1) * 32 + g5] + ctab[(f * 3 + 2) * 32 + b5] + (f >= 6 ? cc.kt[(f - 6) >> 1] : 0.0f);
        uint32_t word = 0;
        for (int r5 = 0; r5 < 32; ++r5) {
            float v[kCubeFn];
#pragma unroll
            for (int f = 0; f < kCubeFn; ++f) v[f] = part[f] + ctab[(f * 3) * 32 + r5];
            const bool no_tissue = v[0] >= ylimf;
            const float t0lb = h_pos ? fminf(v[2], v[3]) : fmaxf(v[2], v[3]);
            const float t1ub = l_pos ? fminf(v[4], v[5]) : fmaxf(v[4], v[5]);
            const bool cone = fminf(fminf(v[1], t0lb), -t1ub) > cc.s_ang;
            const float sa = fmaxf(fabsf(v[6]), fabsf(v[7])) + fmaxf(fabsf(v[8]), fabsf(v[9]));       // (the constants kt joined `part`)
            const bool conc1 = fmaf(cc.eps[0], sa, v[7]) < cc.thr[0], conc2 = fmaf(cc.eps[1], sa, v[9]) < cc.thr[1];
            const bool plain = (no_tissue || cone) && conc1 && conc2;    // NaN anywhere => not plain
            word |= plain ? 0u : (1u << r5);
        }
        bits[w] = word;
    }
}

// Share of the sample in ambiguous cells (all threads; two barriers): true when the cube sweep should run.  Evaluated from the
// per-channel tables for the cells of every kCubeShareStep-th row of the sample, BEFORE the mask is built: a tile that declines (the
// spatially smooth synthetic ones: 55 %) pays for 1 024 cells instead of 32 768.
template <int NT>
__device__ __forceinline__ bool cube_worthwhile(const uint32_t* samp, int n_sample, int cps_log2, int P, const float* ctab, const CubeConsts& cc, float ylimf,
                                                unsigned int* counter, int tid, int& share_pct) {
    if (tid == 0) *counter = 0;
    __syncthreads();
    uint32_t amb = 0, seen = 0;
    for (int b = tid; b < n_sample; b += kCubeShareStep * NT) {        // every kCubeShareStep-th row of the sample is plenty
        if (sample_absent(b, cps_log2, P)) continue;
        amb += cube_cell_plain(ctab, cc, ylimf, as_global(samp)[b] & 0xffffffu) ? 0u : 1u;
        ++seen;
    }
    uint32_t both = amb | (seen << 16);                                 // (at most 16 Ki / 16 entries in all: the halves cannot overflow)
    for (int o = 32; o > 0; o >>= 1) both += (uint32_t)__shfl_xor((int)both, o, 64);
    if ((tid & 63) == 0 && both) atomicAdd(counter, both);
    __syncthreads();
    const uint32_t tot = *counter;
    share_pct = (tot >> 16) ? (int)(100u * (tot & 0xffffu) / (tot >> 16)) : 100;
    return share_pct <= kCubeMaxSharePct;
}

// exec-masked append of `value` for the lanes of m at LDS byte address buf + 4 n (n wave-uniform): RawSink::put_value's core
__device__ __forceinline__ void lds_append_masked(uint32_t buf, uint32_t n, unsigned long long m, uint32_t value) {
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t sbase;
    asm("s_lshl2_add_u32 %0, %1, %2" : "=s"(sbase) : "s"((uint32_t)__builtin_amdgcn_readfirstlane((int)n)), "s"(buf) : "scc");
    const uint32_t addr = sbase + 4u * rank;
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "s"(m), "v"(addr), "v"(value) : "memory");
#else
    (void)rank; (void)value; (void)buf; (void)n;
#endif
}

// Where the exact test's candidates go: straight to the tile's lists (no second staging level: a drain flags ~50 pixels of 128 and
// pays ONE allocation for them).  TWO lists: the angular candidates (tissue outside the plain cone) and the concentration candidates
// (either stain's conservative test).  Each refine pass of finish 2 then reads the ~45 % / ~65 % of the candidates that concern it
// instead of all of them; a pixel that is both is stored twice.  The two heads share one 64-bit word: one atomic per drain.
struct RawDirect {
    uint32_t* dst_c;            // concentration candidates of the tile (the list the per-pixel sweeps fill with everything)
    uint32_t* dst_a;            // angular candidates of the tile
    unsigned long long* heads;  // LDS: low word = head of dst_c, high word = head of dst_a
    uint32_t cap_c, cap_a;      // capacities; a head beyond its capacity marks that list incomplete, as with RawSink
    __device__ __forceinline__ void alloc(uint32_t n_c, uint32_t n_a, int lane, uint32_t& base_c, uint32_t& base_a) const {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(heads, (unsigned long long)n_c | ((unsigned long long)n_a << 32));
        base_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        base_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
    }
    __device__ __forceinline__ static void store(uint32_t* dst, uint32_t cap, unsigned long long m, uint32_t q, uint32_t base, int lane) {
        const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (((m >> lane) & 1ull) && at < cap) as_global(dst)[at] = q;
    }
    // two pixel rows at once: masks {angle, conc} of each
    __device__ __forceinline__ void put2(unsigned long long a0, unsigned long long c0, uint32_t q0, unsigned long long a1, unsigned long long c1,
                                         uint32_t q1, int lane) const {
        const uint32_t nc0 = (uint32_t)__popcll(c0), nc1 = (uint32_t)__popcll(c1), na0 = (uint32_t)__popcll(a0), na1 = (uint32_t)__popcll(a1);
        uint32_t bc, ba;
        alloc(nc0 + nc1, na0 + na1, lane, bc, ba);
        store(dst_c, cap_c, c0, q0, bc, lane);
        store(dst_c, cap_c, c1, q1, bc + nc0, lane);
        store(dst_a, cap_a, a0, q0, ba, lane);
        store(dst_a, cap_a, a1, q1, ba + na0, lane);
    }
};

// The merged selection sweep behind the cube mask.  bits_lds: LDS byte address of the 4 KB mask; ring_lds: LDS byte address of this
// wave's kCubeRing staging entries (wave-uniform).  Pixels of ambiguous cells are appended to the ring; whenever it holds two full
// rows (checked every second pixel row) they are re-tested exactly, two rows side by side, by the test of select_sweep<kStageMerged>
// (the variant with the per-pixel tissue test), and what that flags goes to the tile's two candidate lists (RawDirect).  c0 must be a multiple of 64.
template <bool ALIGNED, int kTrip, bool STREAM, class TR>
__device__ __forceinline__ void select_sweep_cube(const uint8_t* src, int P, int c0, int c1, int t, int nthreads, const TR& T, float ylimf,
                                                  const SelConsts& K, uint32_t bits_lds, uint32_t ring_lds, const RawDirect& out) {
    const size_t nbytes = (size_t)P * 3;
    const int lane = t & 63;
    const float nhi0m = in_vgpr(-(K.hi0 + kAngleMargin)), nlo1m = in_vgpr(-(K.lo1 - kAngleMargin));
    const int w0 = __builtin_amdgcn_readfirstlane(c0 + (t & ~63));
    auto fetch = [&](int cc) { return load_chunk_clamped<ALIGNED, STREAM>(src, nbytes, cc, c1); };
    uint32_t rn = 0;                                             // ring fill, wave-uniform
    struct G3 { float2 r, g, b; };
    auto gather = [&](uint32_t q) { return G3{T.gam_odf(T.addr(q, 0)), T.gam_odf(T.addr(q, 1)), T.gam_odf(T.addr(q, 2))}; };
    struct M2 { unsigned long long a, c; };                      // lane masks: angular candidate, concentration candidate
    auto flags = [&](const G3& e) -> M2 {
        const bool tc = is_tissue_f(e.r.x, e.g.x, e.b.x, ylimf);
        const float x = fmaf(K.V[4], e.b.y, fmaf(K.V[2], e.g.y, K.V[0] * e.r.y));
        const float y = fmaf(K.V[5], e.b.y, fmaf(K.V[3], e.g.y, K.V[1] * e.r.y));
        const float d = x + fabsf(y);
        const float t0 = fmaf(nhi0m, d, y), t1 = fmaf(nlo1m, d, y);
        const bool pp = fminf(fminf(x, t0), -t1) > 0.0f;
        const float a1 = fmaf(K.u[0][1], y, fmaf(K.u[0][0], x, K.kt[0]));
        const float a2 = fmaf(K.u[1][1], y, fmaf(K.u[1][0], x, K.kt[1]));
        const float sa = fabsf(a1) + fabsf(a2);
        const bool g1 = fmaf(K.eps[0], sa, a1) >= K.thr[0], g2 = fmaf(K.eps[1], sa, a2) >= K.thr[1];
        return M2{__builtin_amdgcn_ballot_w64(tc) & ~__builtin_amdgcn_ballot_w64(pp), __builtin_amdgcn_ballot_w64(g1) | __builtin_amdgcn_ballot_w64(g2)};
    };
    auto ring_read = [&](uint32_t i) -> uint32_t {
#if defined(__HIP_DEVICE_COMPILE__)
        return *(SL_LDS const uint32_t*)(ring_lds + 4u * i);
#else
        return i;
#endif
    };
    auto drain = [&]() {
        while (rn >= 128u) {                                     // wave-uniform; two rows side by side: their LDS round trips overlap
            rn -= 128u;
            const uint32_t q0 = ring_read(rn + (uint32_t)lane), q1 = ring_read(rn + 64u + (uint32_t)lane);
            const G3 e0 = gather(q0), e1 = gather(q1);
            const M2 f0 = flags(e0), f1 = flags(e1);
            out.put2(f0.a, f0.c, q0, f1.a, f1.c, q1, lane);
        }
    };
    auto pixel_mask = [&](uint32_t p) -> bool {
#if defined(__HIP_DEVICE_COMPILE__)
        // (bits_lds is 4 KB aligned: the base joins the address in the and-or that builds it)
        const uint32_t w = *(SL_LDS const uint32_t*)(((p >> 12) & 0xF80u) | (((p >> 9) & 0x7Cu) | bits_lds));
#else
        const uint32_t w = 0;
#endif
        return ((w >> cube_bit(p)) & 1u) != 0u;
    };
    auto compute = [&](auto tail_tag, const Chunk& ch, int cc) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        uint32_t p[4];
        bool amb[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            p[px] = chunk_pixel(ch, px);
            amb[px] = pixel_mask(px == 0 ? ch.w0 : p[px]);        // (a stray top byte does not reach the cell address or the bit index)
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            unsigned long long m = __builtin_amdgcn_ballot_w64(amb[px]);
            if (TAIL) {
                const bool inb = (cc < c1) & (ALIGNED | ((size_t)cc * 4 + px < (size_t)P));
                m &= __builtin_amdgcn_ballot_w64(inb);
            }
            lds_append_masked(ring_lds, rn, m, p[px]);
            rn += (uint32_t)__popcll(m);
            if (px & 1) drain();                                 // at most 127 + 2 x 64 entries before it: kCubeRing holds them
        }
    };
    Chunk cur[kTrip], nx[kTrip];
#pragma unroll
    for (int k = 0; k < kTrip; ++k) { cur[k] = fetch(w0 + lane + k * nthreads); nx[k] = fetch(w0 + lane + (kTrip + k) * nthreads); }
    auto trip = [&](auto tail_tag, int cb) {
#pragma unroll
        for (int k = 0; k < kTrip; ++k) {
            const Chunk ch = cur[k];
            if (k + 1 == kTrip) {
#pragma unroll
                for (int j = 0; j < kTrip; ++j) { cur[j] = nx[j]; nx[j] = fetch(cb + lane + (2 * kTrip + j) * nthreads); }
            }
            compute(tail_tag, ch, cb + k * nthreads + lane);
        }
    };
    const int lim = ALIGNED ? c1 : min(c1, P >> 2);
    int cb = w0;
    for (; cb + (kTrip - 1) * nthreads + 64 <= lim; cb += nthreads * kTrip) trip(std::false_type{}, cb);
    if (cb < c1) trip(std::true_type{}, cb);
    // what is left: fewer than 128 entries.  (Entries beyond the fill are stale ring contents, always valid pixels: read by every
    // lane -- no predicated load -- and masked out of the result.)
    {
        const uint32_t q0 = ring_read((uint32_t)lane), q1 = ring_read(64u + (uint32_t)lane);
        const G3 e0 = gather(q0 & 0xffffffu), e1 = gather(q1 & 0xffffffu);
        const unsigned long long l0 = __builtin_amdgcn_ballot_w64((uint32_t)lane < rn), l1 = __builtin_amdgcn_ballot_w64(64u + (uint32_t)lane < rn);
        const M2 f0 = flags(e0), f1 = flags(e1);
        out.put2(f0.a & l0, f0.c & l0, q0, f1.a & l1, f1.c & l1, q1, lane);
        rn = 0;
    }
}

}  // namespace sl
