// slide_merged.hip -- the POOLED slide-level statistics in ONE full sweep (round 6; SURVEY 8e-2, BASELINE.json configs[4]).
//
// slide.hip pins the statistics of the concatenated slide (macenko_stain_extractor.py:18-44 and normalizer.py:36,45-47 on the tall
// image of every tile of every rank) with three full sweeps: moments, angular window, concentration window.  Here the moments sweep
// also collects the candidates of all four order statistics, the way the two-sweep schedule of the per-tile kernel does
// (stats_twosweep.hpp), with the estimate taken from a stratified pixel SAMPLE of the whole slide instead of one tile:
//   S1  sl_pool2_sample   one 64-pixel sub-row in 2^sample_log2: the sample's moment sums and the sampled pixels as a packed list
//       -> all-reduce (12 doubles) -> sl_pool2_begin: eigenvectors V~ of the sample, its plane's normal, the a-priori tilt bound
//   S2  sl_pool2_hist(sample, angle)  8192-bin histogram of the pseudo-angle under V~ (+ fourth moments for the tilt bound)
//       -> all-reduce -> sl_pool2_bands(angle): brackets of both angular percentiles, the two half-spaces of the plain cone,
//          the box of stain matrices (ts_box)
//   S3  sl_pool2_hist(sample, conc)   histogram of both concentrations under the box centre
//       -> all-reduce -> sl_pool2_bands(conc): concentration thresholds (ts_thresholds)
//   F   sl_pool2_sweep    THE full sweep: exact moment sums + every pixel that is not PROVEN plain (inside the angular cone and below
//       both concentration thresholds, whatever exact plane and stain matrix the bounds leave possible) appended raw to a block list
//       -> all-reduce (16 doubles) -> sl_pool2_exact: exact eigenvectors, the plane check (ts_verify's conditions), ranks
//   R   per key set, up to SL_POOL2_LEVELS times: sl_pool2_hist(candidates, window) -> all-reduce -> sl_pool2_step: a radix descent
//       in the ordered binary32 domain, one target per wanted rank (k and k + 1 of each order statistic share a window until they fall
//       into different bins) -> the exact order statistics of the binary32 keys, the stain matrix (merged_verify), maxC
// Non-candidates are proven to lie strictly inside (hi0, lo1) / below L_i under the EXACT basis, so the order statistics of the
// whole slide are those of the candidates at shifted ranks: results never depend on the sample, only whether this route succeeds
// does (state[SL_POOL_MISS] != 0: the caller takes the three-sweep chain of slide.hip).  Every decision is taken on the device from
// all-reduced data: every rank reaches the same state without a broadcast, and nothing is read back before the end.
#include <type_traits>
#include "stats_kernels.hpp"
#include "sl_host.hpp"
#include <cmath>
#include <cstring>

using namespace sl;

namespace {

constexpr int kP2ListThreads = 1024;     // list passes (64 KB of LDS bins per workgroup: two workgroups = 32 waves per CU; with 256 threads the passes ran at 2 waves per SIMD)
constexpr int kP2SampleBlkLog2 = 8;      // sample list: blocks of 256 entries (one wave iteration)
constexpr int kP2CandBlkLog2 = 10;       // candidate list: blocks of 1024 entries owned by one wave (each wave a private run of blocks, then a shared pool)
constexpr int kP2Stage = 224;            // per-wave LDS staging entries of the sweep (8 waves: 7 KB beside the 64 KB row table)
constexpr double kP2Deff = 16.0;         // design effect assumed for the clustered sample (64-pixel sub-rows) in rank and tilt statistics
constexpr double kP2Z = 6.0;             // bracket half-width in standard deviations of the sample rank
constexpr uint32_t kBitTissue = 1u << 24, kBitValid = 1u << 25, kBitAng = 1u << 26, kBitConc = 1u << 27;
constexpr int kP2TailSlots = SL_POOL2_TAIL_SLOTS;
constexpr int kP2GridBins = SL_POOL2_GRID_BINS;
constexpr int kP2WinBits = 11;
constexpr int kP2WinBins = SL_POOL2_WINDOW_BINS;   // bins per target of a window pass (11 key bits per level: two levels settle a bracket of up to 2^22 values)
constexpr int kP2KeyBins = SL_POOL2_KEY_BINS;      // single keys per target at the last level of a window descent
constexpr int kP2KeyBits = 12;
constexpr int kTailPer = 8;               // tail words per slot: [0..3] keys below the window of target u (grid passes: 0, 1), [4] listed entries or q1, [5] q2
constexpr int kTailWords = kTailPer * kP2TailSlots;
constexpr double kQScale = 1048576.0;    // fixed-point scale of the fourth-moment sums riding in the uint64 histogram buffer

// ---- the pool2 state (doubles; include/stainlib_hip.h SL_POOL2_*).  [0, 10) as SL_POOL_*.
enum {
    kM = SL_POOL_M, kMaxC = SL_POOL_MAXC, kStatus = SL_POOL_STATUS, kMiss = SL_POOL_MISS,
    kT = 10, kNpx = 11, kVd = 12, kVf = 18, kK = 24, kG = 26, kSub = 28,
    kTs = 30, kNs = 31, kSlog = 32, kWhy = SL_POOL2_WHY, kVhD = 34, kVhF = 40, kNh = 46, kTau = 49, kK1 = 50, kK2 = 51, kZref = 52,
    kMean = 53, kGap = 56, kLam = 58, kPct = 59,
    kBrk = 60,            // sample-space angular brackets lo0, hi0, lo1, hi1 (pseudo-angle under V~)
    kGH = 64, kGL = 67,   // half-space normals (binary64)
    kSw = 70,             // the sweep's binary32 constants: fgH[3] fgL[3] fn[3] fk1 W[6] kt[2] eps[2] zeta[2] thr[2] = 24
    kNa = 96, kNc = 97, kOvf = 98,
    kBr = 100,            // exact-space: lo end, hi0 (plain cone's lower edge), lo1 (its upper edge), hi end
    kGridLo = 104, kGridScale = 106, kRes = 110,
    kDiag = 116, kDone = 120, kLevel = 121,          // diagnostics: [0] cmax, [1] dn, [2..3] eps seen, ...
    kMk = 128,            // MergedConc (raw bytes)
    // the windows of the selection on the candidates, per TARGET u = 2 t + j: rank k (j = 0) or k + 1 (j = 1) of pair t (t = 0 / 1: lower / upper
    // percentile, or stain 1 / 2): first key (ordered-uint domain) and log2 of the keys per bin.  The two targets of a pair share a window until
    // the two ranks fall into different bins.
    kWinLo = 240, kSh = 244
};
static_assert(kMk * 8 + sizeof(MergedConc) <= kWinLo * 8 && kSh + 4 <= SL_POOL2_STATE_DOUBLES, "");
static_assert(kSw + 24 <= kNa, "");
static_assert(kP2WinBins == (1 << kP2WinBits) && kP2WinBins % 1024 == 0 && kP2WinBins <= kP2KeyBins, "");
static_assert(kP2KeyBins == (1 << kP2KeyBits) && kP2KeyBins % 1024 == 0 && 4 * kP2KeyBins == 2 * kP2GridBins, "");

// why a route was declined (state[SL_POOL2_WHY]; 0 = not declined)
enum { kWhyNoEstimate = 1, kWhyTilt = 2, kWhyBracket = 3, kWhyBox = 4, kWhyConc = 5 };
// bits of state[SL_POOL_MISS] this chain sets: 1 angle stage, 2 concentration stage, 4 plane check, 8 list overflow, 16 box check, 32 declined
enum { kMissAngle = 1, kMissConc = 2, kMissPlane = 4, kMissOverflow = 8, kMissBox = 16, kMissDeclined = 32 };

struct P2List {
    uint32_t* entries;
    uint32_t* counts;        // per block: entries it holds (0: unused)
    unsigned int* n_blocks;  // sample list: = cap_blocks; candidate list: blocks taken from the shared pool (may exceed it: overflow)
    uint32_t cap_blocks;
    int blk_log2;
    uint32_t priv, pool0;    // candidate list: private blocks per wave, first block of the shared pool
};

struct P2Layout {
    int parts, n_items, grid, bpi;
    uint32_t c_priv;         // candidate list: blocks each wave of the sweep owns before it turns to the shared pool
    uint32_t s_cap, c_cap;
    int hist_wgs;            // workgroups of a grid-mode list pass (its partial histograms live in the workspace)
    size_t hdr, partials, hpart, tpart, local, s_counts, s_entries, c_counts, c_entries, total;
};

P2Layout p2_layout(int n, int h, int w, int slog) {
    P2Layout L;
    const long P = (long)h * w;
    const int mg = max_resident_grid();
    L.parts = parts_for(P);
    const long want = (4L * mg + n - 1) / n;
    if (L.parts > want) L.parts = (int)(want < 1 ? 1 : want);
    L.n_items = n * L.parts;
    L.grid = L.n_items < mg ? L.n_items : mg;
    const long nch = (P + 3) >> 2;
    const long kAlign = (long)kSweepThreads * kPhaseTrip;
    const long span = (((nch + L.parts - 1) / L.parts) + kAlign - 1) / kAlign * kAlign;
    const long m_max = (((span + 15) >> 4) >> slog) + 2;
    L.bpi = (int)((m_max + 3) / 4);
    L.s_cap = (uint32_t)((long)L.n_items * L.bpi);
    const long total_px = (long)n * P;
    long cand = total_px / 8;
    const long small = total_px < (1L << 22) ? total_px : (1L << 22);
    if (cand < small) cand = small;
    const long waves = (long)L.grid * (kSweepThreads / 64);
    long blocks = (cand >> kP2CandBlkLog2) + 2 * waves + 16;
    L.c_priv = (uint32_t)(blocks / (2 * waves));                  // half of the capacity in private runs (>= 1 block per wave)
    L.c_cap = (uint32_t)blocks;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    L.hdr = 0;
    L.partials = 256;
    // workgroups of a list pass (each leaves a 64 KB partial histogram for k_p2_gsum): one per CU up to 2^30 pixels, two beyond -- measured,
    // chain alone, 256 against 512: 32 tiles 0.45 -> 0.41 ms, 128 tiles 0.62 -> 0.57, 512 tiles 1.20 -> 1.16, 1 024 tiles 1.99 -> 1.98,
    // 2 048 tiles 4.00 -> 4.05, 12 500 tiles 21.7 -> 22.7
    const int want_wgs = (long)n * P <= (1L << 30) ? 256 : 512;
    L.hist_wgs = mg < want_wgs ? mg : want_wgs;
    L.hpart = up(L.partials + sizeof(double) * 16 * (size_t)mg);
    L.tpart = up(L.hpart + 4 * (size_t)(2 * kP2GridBins) * L.hist_wgs);
    L.local = up(L.tpart + 8 * (size_t)kTailPer * L.hist_wgs);               // sl_pool2_local: two 16-double vectors and one histogram buffer
    L.s_counts = up(L.local + 256 + 8 * (size_t)SL_POOL2_HIST_WORDS);
    L.s_entries = up(L.s_counts + 4 * (size_t)L.s_cap);
    L.c_counts = up(L.s_entries + (4 * (size_t)L.s_cap << kP2SampleBlkLog2));
    L.c_entries = up(L.c_counts + 4 * (size_t)L.c_cap);
    L.total = up(L.c_entries + (4 * (size_t)L.c_cap << kP2CandBlkLog2));
    return L;
}

// ------------------------------------------------------------------------------------------
// S1: the sample
// ------------------------------------------------------------------------------------------
struct P2SampleArgs {
    const uint8_t* rgb;
    int P, parts, n_items, slog;
    int bpi;                 // block slots per item: slot item * bpi + i holds the item's i-th wave iteration (count 0: unused)
    float ylimf;
    P2List list;
    double* partials;        // [grid][16]
};

template <bool ALIGNED>
__global__ __launch_bounds__(kSweepThreads) void k_p2_sample(P2SampleArgs a) {
    __shared__ SmallTab s_tab;
    __shared__ double s_red[kSweepThreads / 64][12];
    s_tab.fill();
    __syncthreads();
    const TabView tab = view_of(s_tab);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const size_t nbytes = (size_t)a.P * 3;
    const uint32_t mask = (1u << a.slog) - 1u;
    Moments mo;
    unsigned long long n_tissue = 0, n_valid = 0;       // wave-uniform
    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int tile = item / a.parts, part = item % a.parts;
        const uint8_t* src = a.rgb + (size_t)tile * nbytes;
        int c0, c1;
        part_range((a.P + 3) >> 2, a.parts, part, c0, c1);
        // the sampled 16-chunk sub-rows of this part: r16 = first + j << slog, j = 0 .. m - 1
        const uint32_t r0 = (uint32_t)c0 >> 4, rend = ((uint32_t)c1 + 15u) >> 4;
        const uint32_t hsh = sample_hash((uint32_t)item) >> 9;
        uint32_t first = (r0 & ~mask) | (hsh & mask);
        if (first < r0) first += mask + 1u;
        const int m = (c0 < c1 && first < rend) ? (int)(((rend - 1u - first) >> a.slog) + 1u) : 0;
        // every slot of the item gets its count (an empty part: zeros): a same-address atomic per block would cost more than the whole pass
        for (int i = tid; i < a.bpi; i += kSweepThreads) a.list.counts[(size_t)item * a.bpi + i] = 4 * i < m ? 256u : 0u;
        for (int j0 = wave * 4; j0 < m; j0 += (kSweepThreads / 64) * 4) {       // wave-uniform
            const int j = j0 + grp;
            const bool live_row = j < m;
            const int cc = (int)((first + ((uint32_t)(live_row ? j : 0) << a.slog)) * 16u) + sub;
            const bool live = live_row & (cc < c1);
            const Chunk ch = load_chunk_clamped<ALIGNED>(src, nbytes, cc, c1);
            uint32_t e[4];
            BurstMoments bm;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const uint32_t p = chunk_pixel(ch, px) & 0xffffffu;
                const uint32_t r = p & 255u, g = (p >> 8) & 255u, b = p >> 16;
                const bool ok = live & (ALIGNED | ((size_t)cc * 4 + px < (size_t)a.P));
                const bool tc = ok & is_tissue_f(tab.gam(r), tab.gam(g), tab.gam(b), a.ylimf);
                if (tc) bm.add(tab.odf(r), tab.odf(g), tab.odf(b));
                n_tissue += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(tc));
                n_valid += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(ok));
                e[px] = p | (tc ? kBitTissue : 0u) | (ok ? kBitValid : 0u);
            }
            bm.flush(mo);
            const uint32_t blk = (uint32_t)item * (uint32_t)a.bpi + (uint32_t)(j0 >> 2);
            if ((j0 >> 2) < a.bpi) {
                uint4 v; v.x = e[0]; v.y = e[1]; v.z = e[2]; v.w = e[3];
                *reinterpret_cast<uint4*>(a.list.entries + ((size_t)blk << kP2SampleBlkLog2) + 4u * (uint32_t)lane) = v;
            }
        }
    }
    double v[12];
    mo.to_array(v, 0u, lane);
    v[0] = lane == 0 ? (double)n_tissue : 0.0;
    v[10] = lane == 0 ? (double)n_valid : 0.0;
    v[11] = 0.0;
#pragma unroll
    for (int i = 0; i < 11; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0)
        for (int i = 0; i < 12; ++i) s_red[wave][i] = v[i];
    __syncthreads();
    if (tid < 12) {
        double t = 0;
        for (int wv = 0; wv < kSweepThreads / 64; ++wv) t += s_red[wv][tid];
        a.partials[(size_t)blockIdx.x * 16 + tid] = t;
    }
    if (blockIdx.x == 0 && tid == 0) *a.list.n_blocks = a.list.cap_blocks;      // every slot carries a count
}

// rows of 16 doubles summed in a fixed order (wave c: column c; lane l adds rows l, l + 64, ..., then a butterfly);
// out[npx_slot] = npx; out[ovf_slot] = 1 when the list outgrew its capacity.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_p2_sum(const double* partials, int rows, int n_out, double* out, double npx, int npx_slot,
                                                 const unsigned int* n_blocks, uint32_t cap_blocks, int ovf_slot) {
    const int col = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double t = 0;
    if (col < n_out) for (int r = lane; r < rows; r += 64) t += partials[(size_t)r * 16 + col];
    t = wave_sum(t);
    if (col == npx_slot) t = npx;
    if (col == ovf_slot) t = (*n_blocks > cap_blocks) ? 1.0 : 0.0;
    if (lane == 0) out[col] = t;
}

// ------------------------------------------------------------------------------------------
// decision step after S1: the sample's eigenvectors, its plane and the Gaussian tilt bound (phase0_estimate of the per-tile kernel)
// ------------------------------------------------------------------------------------------
__global__ void k_p2_begin(const double* mom, double* st, double pct, double lam, int slog) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < SL_POOL2_STATE_DOUBLES; ++i) st[i] = 0.0;
    double Vd[6], wv[3];
    float Vf[6];
    const int status = eigvecs_from_moments(mom, Vd, Vf, wv);
    const double ns = mom[0];
    st[kTs] = ns; st[kNs] = mom[10]; st[kSlog] = (double)slog; st[kLam] = lam; st[kPct] = pct;
    const double l1 = wv[0], l2 = wv[1], l3 = wv[2] > 0.0 ? wv[2] : 0.0;
    int why = 0;
    if (!(status == SL_TILE_OK && ns >= (double)kTsMinTissue && l2 > 1e-9 * l1 && l2 - l3 > 0.05 * l2) || mom[11] > 0.0) why = kWhyNoEstimate;
    double tau = kTsMinTau;
    if (!why && slog > 0) {
        tau = kTiltZ * sqrt(kP2Deff / ns) * fmax(sqrt(l3 * l2) / (l2 - l3), sqrt(l3 * l1) / (l1 - l3));
        tau = fmax(tau, kTsMinTau);
        if (!(tau <= kTsMaxTau)) why = kWhyTilt;
    }
    double nd[3] = {Vd[2] * Vd[5] - Vd[4] * Vd[3], Vd[4] * Vd[1] - Vd[0] * Vd[5], Vd[0] * Vd[3] - Vd[2] * Vd[1]};
    const double nn = 1.0 / sqrt(nd[0] * nd[0] + nd[1] * nd[1] + nd[2] * nd[2]);
    for (int i = 0; i < 6; ++i) { st[kVhD + i] = Vd[i]; st[kVhF + i] = (double)Vf[i]; }
    for (int c = 0; c < 3; ++c) { st[kNh + c] = why ? 0.0 : nd[c] * nn; st[kMean + c] = ns > 0 ? mom[1 + c] / ns : 0.0; }
    st[kTau] = tau;
    st[kGap] = l1 - l3; st[kGap + 1] = l2 - l3;
    st[kZref] = 4.0 * sqrt(l3);
    st[kWhy] = (double)why;
    // the grid of the sample's pseudo-angle histogram: [-1, 1) in kP2GridBins bins
    st[kGridLo] = -1.0; st[kGridScale] = 0.5 * kP2GridBins;
    st[kGridLo + 1] = -1.0; st[kGridScale + 1] = 0.5 * kP2GridBins;
}

// ------------------------------------------------------------------------------------------
// list passes: a histogram of the keys of a block list -- on a uniform grid of kP2GridBins bins per order statistic (mode 0: the sample),
// or (mode 1: the candidates) per TARGET u = 2 t + j (rank k / k + 1 of order statistic t) over a window of the ordered binary32 values:
// kP2WinBins coarse bins of 2^sh keys, or kP2KeyBins single keys (sh = 0: the bins ARE keys); a target whose window is its pair's (the
// normal case: both ranks in one bin) has no histogram of its own.  A histogram buffer is [kTailWords tail words][2 x kP2GridBins bins =
// 4 x kP2KeyBins]; tail word i (0..3 keys below the window of target i, 4 listed entries or q1, 5 q2) is the sum of its kP2TailSlots
// copies at [kTailPer slot + i].
// The passes touch no global atomic in their loop: the bins are LDS-private per workgroup, written out as partial histograms and
// added up by k_p2_gsum.  Measured on the way: 8 M uint64 atomics on a 65536-bin global histogram took 0.25 - 1.9 ms by how hot the
// bins were (a slide's keys are heavily tied: 13 G pixels over at most 16.7 M colours), a 65536-key window with global atomics
// 1.4 - 5 ms per 12 500 tiles, and one same-address atomic per wave on the tail words 0.5 ms per pass.
// ------------------------------------------------------------------------------------------
struct P2HistArgs {
    P2List list;
    const double* state;
    uint32_t need;           // flag bits an entry must carry
    int basis;               // angle: state offset of the binary32 basis (kVhF / kVf); conc: 0 = the box centre's constants, 1 = the exact M
    int fourth;              // angle over the sample: also the fourth-moment sums of the tilt bound
    uint32_t* part;          // grid mode: [gridDim][2 x kP2GridBins] partial histograms
    unsigned long long* tpart;   // [gridDim][kTailPer] partial tail words
    unsigned long long* hist;    // the pass's output buffer: zeroed here by workgroup 0 (k_p2_gsum, the next kernel on the stream, adds into it)
};

template <int KEYSET, int MODE>
// (two workgroups per CU: 2 x 70 KB of LDS, 64 VGPRs)
__global__ __launch_bounds__(kP2ListThreads, 8) void k_p2_hist(P2HistArgs a) {
    constexpr int NT = kP2ListThreads;
    __shared__ SmallTab s_tab;
    __shared__ uint32_t s_h[2 * kP2GridBins];
    __shared__ unsigned long long s_tail[kTailPer];
    // (a memset node per pass was eight more launches per chain: ~30 us of a 512-tile slide's 2 ms)
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < SL_POOL2_HIST_WORDS; i += NT) a.hist[i] = 0ull;
    s_tab.fill();
    const double* st = a.state;
    if (MODE == 1 && ((int)st[kDone] & (1 << KEYSET))) return;        // uniform: the key set is settled (k_p2_gsum skips alike: hist stays zero)
    // histograms: mode 0 two grids of kP2GridBins; mode 1 four targets of kP2KeyBins slots (kP2WinBins used by a coarse window, none by a
    // target that shares its pair's window)
    constexpr int NU = MODE == 0 ? 2 : 4;
    constexpr uint32_t kStride = MODE == 0 ? (uint32_t)kP2GridBins : (uint32_t)kP2KeyBins;
    uint32_t nbt[NU];
    uint32_t wlo[NU], wsh[NU];
    for (int u = 0; u < NU; ++u) {
        wlo[u] = MODE == 0 ? 0u : (uint32_t)st[kWinLo + u]; wsh[u] = MODE == 0 ? 0u : (uint32_t)st[kSh + u];
        nbt[u] = MODE == 0 ? (uint32_t)kP2GridBins : (wsh[u] == 0 ? (uint32_t)kP2KeyBins : (uint32_t)kP2WinBins);
    }
    if (MODE == 1)
        for (int t = 0; t < 2; ++t)
            if (wlo[2 * t + 1] == wlo[2 * t] && wsh[2 * t + 1] == wsh[2 * t]) nbt[2 * t + 1] = 0;       // uniform
    for (int u = 0; u < NU; ++u)
        for (uint32_t i = threadIdx.x; i < nbt[u]; i += NT) s_h[u * kStride + i] = 0;
    if (threadIdx.x < kTailPer) s_tail[threadIdx.x] = 0;
    __syncthreads();
    const TabView tab = view_of(s_tab);
    const int tid = threadIdx.x, lane = tid & 63;
    float V[6] = {0, 0, 0, 0, 0, 0};
    LassoK L{};
    if (KEYSET == SL_KEYSET_ANGLE) {
        for (int i = 0; i < 6; ++i) V[i] = (float)st[a.basis + i];
    } else if (a.basis == 0) {
        L = reinterpret_cast<const MergedConc*>(st + kMk)->Lc;
    } else {
        lasso_consts(st + kM, st[kLam], L);
    }
    float glo[2], gsc[2];
    for (int t = 0; t < 2; ++t) { glo[t] = (float)st[kGridLo + t]; gsc[t] = (float)st[kGridScale + t]; }
    float nf[3] = {0, 0, 0}, mean[3] = {0, 0, 0};
    if (a.fourth) for (int c = 0; c < 3; ++c) { nf[c] = (float)st[kNh + c]; mean[c] = (float)st[kMean + c]; }
    uint32_t nbel[NU], nmatch = 0;                              // per lane: well below 2^32
    for (int u = 0; u < NU; ++u) nbel[u] = 0;
    float q1 = 0.0f, q2 = 0.0f;
    const uint32_t nblk = a.list.cap_blocks;                   // every block carries a count (0: unused)
    const uint32_t B = 1u << a.list.blk_log2;
    // SPLIT: some target has a window of its own (its pair's two ranks fell into different bins) -- rare; the usual pass is the
    // straight-line two-histogram code (a uniform branch per entry cost the eight entries in flight their overlap: +35 % per pass)
    auto one = [&](uint32_t e, auto split_tag) {
        constexpr bool SPLIT = decltype(split_tag)::value;
        if ((e & a.need) != a.need) return;
        ++nmatch;
        const float ox = tab.odf(e & 255u), oy = tab.odf((e >> 8) & 255u), oz = tab.odf((e >> 16) & 255u);
        float k0, k1;
        if (KEYSET == SL_KEYSET_ANGLE) {
            k0 = k1 = angle_key(V, ox, oy, oz);
            if (a.fourth) {
                const float dx = ox - mean[0], dy = oy - mean[1], dz = oz - mean[2];
                const float a1 = fmaf(V[4], dz, fmaf(V[2], dy, V[0] * dx)), a2 = fmaf(V[5], dz, fmaf(V[3], dy, V[1] * dx));
                const float a3 = fmaf(nf[2], dz, fmaf(nf[1], dy, nf[0] * dx));
                q1 = fmaf(a1 * a3, a1 * a3, q1);
                q2 = fmaf(a2 * a3, a2 * a3, q2);
            }
        } else {
            lasso2(L, ox, oy, oz, k0, k1);
        }
        const float k[2] = {k0, k1};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (KEYSET == SL_KEYSET_ANGLE && a.fourth && t == 1) continue;      // the sample's angle histogram serves both percentiles
            if (MODE == 0) {
                if (k[t] < glo[t]) ++nbel[t];
                else {
                    const uint32_t b = (uint32_t)((k[t] - glo[t]) * gsc[t]);
                    if (b < (uint32_t)kP2GridBins) atomicAdd(&s_h[t * kStride + b], 1u);
                }
            } else {
                const uint32_t o = f2ord(k[t]);
#pragma unroll
                for (int j = 0; j < (SPLIT ? 2 : 1); ++j) {
                    const int u = 2 * t + j;
                    if (SPLIT && nbt[u] == 0) continue;              // uniform: the pair's histogram serves this rank too
                    if (o < wlo[u]) ++nbel[u];
                    else if (((o - wlo[u]) >> wsh[u]) < nbt[u]) atomicAdd(&s_h[u * kStride + ((o - wlo[u]) >> wsh[u])], 1u);
                }
            }
        }
    };
    // the counts of this workgroup's blocks are fetched a batch at a time (one dependent load per block was a latency chain of
    // ~1.5 us x blocks: most of a pass over a list with many unused blocks)
    __shared__ uint32_t s_cnt[NT];
    auto blocks = [&](auto split_tag) {
    for (uint32_t b0 = blockIdx.x; b0 < nblk; b0 += gridDim.x * NT) {
        __syncthreads();
        {
            const unsigned long long bb = (unsigned long long)b0 + (unsigned long long)tid * gridDim.x;
            s_cnt[tid] = bb < nblk ? min(a.list.counts[bb], B) : 0u;
        }
        __syncthreads();
        for (int k = tid >> 6; k < NT; k += NT / 64) {            // a wave per block: 8 independent loads per lane in flight
            const uint32_t cnt = s_cnt[k];
            if (cnt == 0) continue;
            const uint32_t b = b0 + (uint32_t)k * gridDim.x;
            const uint32_t* src = a.list.entries + ((size_t)b << a.list.blk_log2);
            for (uint32_t i0 = lane; i0 < cnt; i0 += 8 * 64) {
                uint32_t e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const uint32_t i = i0 + u * 64; e[u] = i < cnt ? as_global(src)[i] : 0u; }
#pragma unroll
                for (int u = 0; u < 8; ++u) one(e[u], split_tag);
            }
        }
    }
    };
    if (MODE == 1 && (nbt[NU - 1] != 0 || nbt[1] != 0)) blocks(std::true_type{});      // uniform
    else blocks(std::false_type{});
    unsigned long long nbw[NU];
    for (int u = 0; u < NU; ++u) nbw[u] = wave_sum((unsigned long long)nbel[u]);
    unsigned long long w4 = wave_sum((unsigned long long)nmatch), w5 = 0;      // [4]: entries of this key set on the list
    if (a.fourth) {
        w4 = (unsigned long long)(wave_sum((double)q1) * kQScale);
        w5 = (unsigned long long)(wave_sum((double)q2) * kQScale);
    }
    if (lane == 0) {
        for (int u = 0; u < NU; ++u) if (nbw[u]) atomicAdd(&s_tail[u], nbw[u]);
        atomicAdd(&s_tail[4], w4); atomicAdd(&s_tail[5], w5);
    }
    __syncthreads();
    uint32_t* dst = a.part + (size_t)blockIdx.x * (2 * kP2GridBins);
    for (int u = 0; u < NU; ++u)
        for (uint32_t i = tid; i < nbt[u]; i += NT) dst[u * kStride + i] = s_h[u * kStride + i];
    if (tid < kTailPer) a.tpart[(size_t)blockIdx.x * kTailPer + tid] = s_tail[tid];
}

// the partial histograms of the nwg workgroups of a list pass added up into hist (zeroed by the entry point): block (x, y) adds slice y of
// the workgroups for 256 bins -- one thread walking all of them was 170 us of latency per pass
constexpr int kGsumSlices = 16;
__global__ __launch_bounds__(256) void k_p2_gsum(const uint32_t* part, const unsigned long long* tpart, int nwg, unsigned long long* hist,
                                                const double* st, int done_bit) {
    if (done_bit && ((int)st[kDone] & done_bit)) return;       // the pass did not run: hist stays zero
    const int i = blockIdx.x * 256 + threadIdx.x;               // one bin per thread (grid passes: statistic i / kP2GridBins; window passes: target i / kP2KeyBins)
    const int per = (nwg + kGsumSlices - 1) / kGsumSlices;
    const int g0 = blockIdx.y * per, g1 = min(nwg, g0 + per);
    if (blockIdx.x == 0 && threadIdx.x < kTailPer) {
        unsigned long long v = 0;
        for (int gg = g0; gg < g1; ++gg) v += tpart[(size_t)gg * kTailPer + threadIdx.x];
        if (v) atomicAdd(&hist[kTailPer * blockIdx.y + threadIdx.x], v);
    }
    // (block-uniform: 256 bins never straddle the bins a coarse window leaves unused, or two targets)
    if (done_bit) {
        const int u = i / kP2KeyBins;
        if ((int)st[kSh + u] != 0 && (i % kP2KeyBins) >= kP2WinBins) return;
        if ((u & 1) && st[kWinLo + u] == st[kWinLo + u - 1] && st[kSh + u] == st[kSh + u - 1]) return;     // shares its pair's histogram
    }
    unsigned long long t = 0;
    int g = g0;
    for (; g + 8 <= g1; g += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(g + u) * (2 * kP2GridBins) + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; g < g1; ++g) t += part[(size_t)g * (2 * kP2GridBins) + i];
    if (t) atomicAdd(&hist[kTailWords + i], t);
}

static_assert(kGsumSlices <= kP2TailSlots, "");
__device__ __forceinline__ unsigned long long p2_tail(const unsigned long long* hist, int i) {
    unsigned long long t = 0;
    for (int s = 0; s < kP2TailSlots; ++s) t += hist[kTailPer * s + i];
    return t;
}

// ---- a prefix structure over nbins (a multiple of 1024, at most 65536) bins for 1024 threads: which bin holds 0-based rank r (below = keys under the
// first bin)?  returns -1: r lies under the histogram; nbins: beyond it.  All threads call; both end with a barrier.
struct P2Scan {
    unsigned long long seg[1024], cum[1024];
    long long out;
    int which;
};
__device__ __forceinline__ void p2_scan_build(P2Scan& S, const unsigned long long* h, int nbins, int tid) {
    const int per = nbins >> 10;
    unsigned long long s = 0;
    if (per <= 8) {
        for (int j = 0; j < per; ++j) s += h[tid * per + j];
    } else {                                                 // per == 64: eight lanes add up a segment, 64 bytes each (a wave: 4 KB in a row)
        const int wave = tid >> 6, lane = tid & 63;
        for (int j0 = 0; j0 < 64; j0 += 8) {
            const unsigned long long* q = h + ((size_t)(wave * 64 + j0) * 64 + (size_t)lane * 8);
            unsigned long long t = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) t += q[u];
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
            const unsigned long long mine = __shfl(t, (lane & 7) * 8, 64);          // segment j0 + (lane & 7)
            if ((lane >> 3) == (j0 >> 3)) s = mine;                                   // lane j keeps segment j
        }
    }
    __syncthreads();
    S.seg[tid] = s;
    __syncthreads();
    if (tid < 64) {                                   // wave 0: exclusive scan of 1024 segment sums, 16 per lane
        unsigned long long loc = 0;
        for (int j = 0; j < 16; ++j) loc += S.seg[tid * 16 + j];
        unsigned long long inc = loc;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(inc, o, 64);
            if (tid >= o) inc += t;
        }
        unsigned long long acc = inc - loc;
        for (int j = 0; j < 16; ++j) { S.cum[tid * 16 + j] = acc; acc += S.seg[tid * 16 + j]; }
    }
    __syncthreads();
}
__device__ __forceinline__ long long p2_scan_locate(P2Scan& S, const unsigned long long* h, int nbins, long long r, unsigned long long below, int tid) {
    const int per = nbins >> 10;
    const unsigned long long inside = S.cum[1023] + S.seg[1023];
    __syncthreads();
    if (tid == 0) { S.out = r < (long long)below ? -1 : nbins; S.which = -1; }
    __syncthreads();
    const bool in = r >= (long long)below && (unsigned long long)r < below + inside;
    const unsigned long long want = in ? (unsigned long long)r - below : 0ull;
    if (in && want >= S.cum[tid] && want < S.cum[tid] + S.seg[tid]) S.which = tid;      // exactly one thread
    __syncthreads();
    const int sg = S.which;
    if (sg >= 0 && tid < 64) {                         // wave 0 walks the segment's bins side by side
        const unsigned long long v = tid < per ? h[(size_t)sg * per + tid] : 0ull;
        unsigned long long inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(inc, o, 64);
            if (tid >= o) inc += t;
        }
        const unsigned long long w = want - S.cum[sg];
        if (tid < per && w >= inc - v && w < inc) S.out = (long long)sg * per + tid;
    }
    __syncthreads();
    return S.out;
}

// ------------------------------------------------------------------------------------------
// decision steps after S2 / S3: brackets, half-spaces, the box, the thresholds
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_p2_bands(double* st, const unsigned long long* hist, int keyset) {
    __shared__ P2Scan S;
    __shared__ float s_e[8];
    __shared__ MergedConc s_mk;
    __shared__ int s_why;
    const int tid = threadIdx.x;
    if ((int)st[kWhy] != 0) return;                            // uniform: declined earlier
    if (tid == 0) s_why = 0;
    const double deff = (int)st[kSlog] > 0 ? kP2Deff : 0.0;
    if (keyset == SL_KEYSET_ANGLE) {
        const double n = st[kTs], pct = st[kPct];
        const unsigned long long* h0 = hist + kTailWords;
        p2_scan_build(S, h0, kP2GridBins, tid);
        const unsigned long long below = p2_tail(hist, 0);
        // ends 0..3: the brackets at kP2Z sigma of the sample rank; 4..7: the box of stain matrices at kBoxFrac of that (bin edges on
        // the safe side both times: with a sample that is the whole slide, sigma = 0, the box is the bracket)
        for (int i = 0; i < 8; ++i) {
            const int b = (i >> 1) & 1, upper = i & 1;
            const double z = i < 4 ? kP2Z : kBoxFrac * kP2Z;
            const double q = (b == 0 ? 100.0 - pct : pct) / 100.0;
            const double r = q * (n - 1.0), sd = sqrt(fmax(q * (1.0 - q) * n * deff, 0.0));
            const long long rk = upper ? (long long)ceil(r + z * sd) + 1 : (long long)floor(r - z * sd) - 1;
            float e;
            if (upper ? rk > (long long)n - 1 : rk < 0) e = upper ? INFINITY : -INFINITY;
            else {
                const long long bin = p2_scan_locate(S, h0, kP2GridBins, rk, below, tid);
                if (bin < 0 || bin >= kP2GridBins) e = upper ? INFINITY : -INFINITY;
                // bin edges on the safe side; the brackets one bin further out: the exact keys must clear the cone's edge by the
                // margins of the half-space test (with sigma = 0 a bracket is ONE bin, and a key in its last 8 % missed the window)
                else e = (float)(-1.0 + (double)(bin + upper + (i < 4 ? (upper ? 1 : -1) : 0)) / (0.5 * kP2GridBins));
            }
            if (tid == 0) s_e[i] = e;
        }
        __syncthreads();
        if (tid < 64) {
            const float lo0 = s_e[0], hi0 = s_e[1], lo1 = s_e[2], hi1 = s_e[3];
            // the inner ends must exist and leave a cone; the box needs all four
            const bool fin = (lo0 > -INFINITY) & (hi0 < INFINITY) & (lo1 > -INFINITY) & (hi1 < INFINITY);
            const bool cone = (hi0 < INFINITY) & (lo1 > -INFINITY) & (hi0 + 4.0f * kAngleMargin < lo1) & (hi0 > -0.999f) & (lo1 < 0.999f);
            double Vd[6], nd[3];
            for (int i = 0; i < 6; ++i) Vd[i] = st[kVhD + i];
            for (int c = 0; c < 3; ++c) nd[c] = st[kNh + c];
            // the tilt bound from the sample's own fourth moments (fused_phase0)
            double tau = st[kTau];
            if ((int)st[kSlog] > 0) {
                const double t1 = (double)p2_tail(hist, 4) / kQScale, t2 = (double)p2_tail(hist, 5) / kQScale;
                const double se = fmax(sqrt(t1) / (n * st[kGap]), sqrt(t2) / (n * st[kGap + 1]));
                tau = fmax(tau, kTiltZ4 * sqrt(kP2Deff) * se);
            }
            const double kappa1 = tau, kappa2 = tau * tau + 1e-7;
            float box[4] = {s_e[4], s_e[5], s_e[6], s_e[7]};
            if (!fin) { box[0] = -INFINITY; box[1] = INFINITY; box[2] = -INFINITY; box[3] = INFINITY; }
            ts_box(Vd, nd, tau, box, st[kLam], tid, s_mk);
            if (tid == 0) {
                int why = 0;
                if (!(tau <= kTsMaxTau)) why = kWhyTilt;
                else if (!cone || !fin) why = kWhyBracket;
                else if (!s_mk.ok) why = kWhyBox;
                const double aH = angle_of_pseudo((double)hi0), aL = angle_of_pseudo((double)lo1);
                double sH, cH, sL, cL;
                sincos(aH, &sH, &cH);
                sincos(aL, &sL, &cL);
                const double k2 = kappa2 + 6e-6;
                for (int c = 0; c < 3; ++c) {
                    const double gH = Vd[2 * c] * -sH + Vd[2 * c + 1] * cH, gL = Vd[2 * c] * sL + Vd[2 * c + 1] * -cL;
                    st[kGH + c] = gH; st[kGL + c] = gL;
                    st[kSw + c] = (double)(float)(gH - k2); st[kSw + 3 + c] = (double)(float)(gL - k2);
                    st[kSw + 6 + c] = (double)(float)nd[c];
                }
                st[kSw + 9] = (double)(float)(kappa1 * (1.0 + 1e-6));
                st[kTau] = tau; st[kK1] = kappa1; st[kK2] = kappa2;
                st[kBrk] = lo0; st[kBrk + 1] = hi0; st[kBrk + 2] = lo1; st[kBrk + 3] = hi1;
                // the grid of the sample's concentration histograms: [1e-30, 16) in kP2GridBins bins (keys of exactly 0 count as below)
                for (int t = 0; t < 2; ++t) { st[kGridLo + t] = 1e-30; st[kGridScale + t] = kP2GridBins / 16.0; }
                *reinterpret_cast<MergedConc*>(st + kMk) = s_mk;
                s_why = why;
            }
        }
        __syncthreads();
        if (tid == 0 && s_why) st[kWhy] = (double)s_why;
        return;
    }
    // ---- concentrations under the box centre: brackets of the 99th percentile of each column
    const double n = st[kNs];
    for (int t = 0; t < 2; ++t) {
        const unsigned long long* h = hist + kTailWords + kP2GridBins * t;
        p2_scan_build(S, h, kP2GridBins, tid);
        const unsigned long long below = p2_tail(hist, t);
        const double q = 0.99;
        const double r = q * (n - 1.0), sd = sqrt(fmax(q * (1.0 - q) * n * deff, 0.0));
        for (int upper = 0; upper < 2; ++upper) {
            const long long rk = upper ? (long long)ceil(r + kP2Z * sd) + 1 : (long long)floor(r - kP2Z * sd) - 1;
            float e;
            if (upper ? rk > (long long)n - 1 : rk < 0) e = upper ? INFINITY : -INFINITY;
            else {
                const long long bin = p2_scan_locate(S, h, kP2GridBins, rk, below, tid);
                if (bin < 0) e = -INFINITY;
                else if (bin >= kP2GridBins) e = INFINITY;
                else e = (float)((double)(bin + upper) / (kP2GridBins / 16.0));
            }
            if (tid == 0) s_e[2 * t + upper] = e;
        }
        __syncthreads();
    }
    if (tid == 0) {
        MergedConc mk = *reinterpret_cast<MergedConc*>(st + kMk);
        ts_thresholds(mk, s_e[0], s_e[2], s_e[1], s_e[3], (float)st[kZref]);
        if (!mk.ok) st[kWhy] = (double)kWhyConc;
        *reinterpret_cast<MergedConc*>(st + kMk) = mk;
        for (int i = 0; i < 2; ++i) {
            for (int c = 0; c < 3; ++c) st[kSw + 10 + 3 * i + c] = (double)(float)mk.C.W[i][c];
            st[kSw + 16 + i] = (double)mk.kt[i];
            st[kSw + 18 + i] = (double)mk.eps[i];
            st[kSw + 20 + i] = (double)(float)(mk.zeta[i] * (1.0 + 1e-6));
            st[kSw + 22 + i] = (double)mk.thr[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// F: the full sweep -- exact moment sums and the raw candidates
// ------------------------------------------------------------------------------------------
struct BlkPos { uint32_t blk, fill; };
// a wave's staged candidates to its current block of the list (a new block when that is full); out of line like raw_flush
__device__ __noinline__ BlkPos p2_flush(uint32_t buf_lds, uint32_t n_, uint32_t blk_, uint32_t fill_, uint32_t* entries_, uint32_t* counts_,
                                       unsigned int* n_blocks_, uint32_t cap_blocks_, uint32_t priv_end_, uint32_t pool0_) {
#if defined(__HIP_DEVICE_COMPILE__)
    SL_LDS const uint32_t* buf = (SL_LDS const uint32_t*)buf_lds;
#else
    const uint32_t* buf = nullptr;
#endif
    constexpr uint32_t B = 1u << kP2CandBlkLog2;
    const int lane = threadIdx.x & 63;
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_), cap_blocks = (uint32_t)__builtin_amdgcn_readfirstlane((int)cap_blocks_);
    const uint32_t priv_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)priv_end_), pool0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pool0_);
    uint32_t blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)blk_), fill = (uint32_t)__builtin_amdgcn_readfirstlane((int)fill_);
    uint32_t* entries = uni_ptr(entries_);
    uint32_t* counts = uni_ptr(counts_);
    unsigned int* n_blocks = uni_ptr(n_blocks_);
    uint32_t done = 0;
    while (done < n) {
        if (fill == B) {
            if (blk < cap_blocks && lane == 0) counts[blk] = B;
            if (blk + 1u < priv_end) ++blk;                       // the wave's own run of blocks (it starts on the first of them)
            else {
                uint32_t nb = 0;
                if (lane == 0) nb = atomicAdd(n_blocks, 1u);
                blk = pool0 + (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
                if (blk < pool0) blk = 0xfffffff0u;                // (wrapped)
            }
            fill = 0;
        }
        const uint32_t take = min(n - done, B - fill);
        if (blk < cap_blocks)
            for (uint32_t i = lane; i < take; i += 64) as_global(entries)[((size_t)blk << kP2CandBlkLog2) + fill + i] = buf[done + i];
        fill += take;
        done += take;
    }
    return BlkPos{blk, fill};
}

struct BlockSink {
    uint32_t buf;               // LDS byte address of this wave's kP2Stage staging entries
    uint32_t n;                 // wave-uniform fill of the staging buffer
    uint32_t blk, fill;         // the wave's current block of the list and its fill
    uint32_t* entries;
    uint32_t* counts;
    unsigned int* n_blocks;
    uint32_t cap_blocks, priv_end, pool0;
    __device__ __forceinline__ void flush() {
        if (n != 0) {
            const BlkPos p = p2_flush(buf, n, blk, fill, entries, counts, n_blocks, cap_blocks, priv_end, pool0);
            blk = p.blk; fill = p.fill;
        }
        n = 0;
    }
    __device__ __forceinline__ void close(int lane) {
        flush();
        if (blk < cap_blocks && lane == 0) counts[blk] = fill;
    }
    // RawSink::put_value: branch-free masked LDS write of the flagged lanes' values
    __device__ __forceinline__ void put_value(unsigned long long m, uint32_t value) {
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (__builtin_expect(n + cnt > (uint32_t)kP2Stage, 0)) flush();
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t sbase;
        asm("s_lshl2_add_u32 %0, %1, %2" : "=s"(sbase) : "s"(n), "s"(buf) : "scc");
        const uint32_t addr = sbase + 4u * rank;
        unsigned long long saved;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "s"(m), "v"(addr), "v"(value) : "memory");
#else
        (void)rank; (void)value;
#endif
        n += cnt;
    }
};

#ifndef P2_TRIP
#define P2_TRIP 2          // chunks per lane and trip: with 4 (as the other sweeps) the kernel spilled 250 VGPRs and 19 SGPRs into its loop
#endif
struct P2SweepK {            // VGPR-resident
    float gH[3], gL[3], n[3], k1;
    float W[2][3], kt[2], eps[2], zeta[2], thr[2];
    float ylimf;
};

template <bool ALIGNED, int kTrip, bool STREAM>
__device__ __forceinline__ void p2_sweep_part(const uint8_t* src, int P, int c0, int c1, int t, const TabReaderB& T, const P2SweepK& K,
                                              BlockSink& sink, Moments& mo, unsigned long long& n_tissue) {
    constexpr int nthreads = kSweepThreads;
    const size_t nbytes = (size_t)P * 3;
    const int lane = t & 63;
    const int w0 = __builtin_amdgcn_readfirstlane(c0 + (t & ~63));
    struct G { float2 v[12]; };
    auto fetch = [&](int cc) { return load_chunk_clamped<ALIGNED, STREAM>(src, nbytes, cc, c1); };
    auto gather = [&](const Chunk& ch) {
        G g;
#pragma unroll
        for (int i = 0; i < 12; ++i) g.v[i] = T.gam_odf(T.addr(ch, i));
        return g;
    };
    BurstMoments bm;
    uint32_t cnt_t = 0;                                        // wave-uniform, per part
    auto compute = [&](auto tail_tag, const Chunk& ch, const G& g, int cc) {
        constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const float2 er = g.v[3 * px], eg = g.v[3 * px + 1], eb = g.v[3 * px + 2];
            bool tc = is_tissue_f(er.x, eg.x, eb.x, K.ylimf);
            const float z = fmaf(K.n[2], eb.y, fmaf(K.n[1], eg.y, K.n[0] * er.y));
            const float tH = fmaf(K.gH[2], eb.y, fmaf(K.gH[1], eg.y, K.gH[0] * er.y));
            const float tL = fmaf(K.gL[2], eb.y, fmaf(K.gL[1], eg.y, K.gL[0] * er.y));
            const bool pp = fmaf(-K.k1, fabsf(z), fminf(tH, tL)) > 0.0f;
            const float a1 = fmaf(K.W[0][2], eb.y, fmaf(K.W[0][1], eg.y, fmaf(K.W[0][0], er.y, K.kt[0])));
            const float a2 = fmaf(K.W[1][2], eb.y, fmaf(K.W[1][1], eg.y, fmaf(K.W[1][0], er.y, K.kt[1])));
            const float sa = fabsf(a1) + fabsf(a2);
            const bool g1 = !(fmaf(K.zeta[0], fabsf(z), fmaf(K.eps[0], sa, a1)) < K.thr[0]), g2 = !(fmaf(K.zeta[1], fabsf(z), fmaf(K.eps[1], sa, a2)) < K.thr[1]);
            bool gc = g1 | g2;
            if (TAIL) {
                const bool inb = (cc < c1) & (ALIGNED | ((size_t)cc * 4 + px < (size_t)P));
                tc = tc & inb; gc = gc & inb;
            }
            const bool ga = tc & !pp;
            cnt_t += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(tc));
            if (tc) bm.add(er.y, eg.y, eb.y);
            const uint32_t value = (chunk_pixel(ch, px) & 0xffffffu) | (ga ? kBitAng : 0u) | (gc ? kBitConc : 0u);
            sink.put_value(__builtin_amdgcn_ballot_w64(ga | gc), value);
#ifdef P2_PXBAR
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    // Two chunks per lane and trip (with the four of the other sweeps this kernel -- moments, staged candidates, 25 constants --
    // spilled 250 VGPRs and 19 SGPRs into its loop); a burst of the binary32 moment sums still spans 16 pixels per lane, the same
    // 16 as in moments_sweep_b: it is flushed after every SECOND trip, so the sums are bit-identical to sl_tile_moments'.
    static_assert(kTrip == 2, "");
    Chunk cur[kTrip], nx[kTrip];
#pragma unroll
    for (int k = 0; k < kTrip; ++k) { cur[k] = fetch(w0 + lane + k * nthreads); nx[k] = fetch(w0 + lane + (kTrip + k) * nthreads); }
#ifdef P2_DBUF
    G g[2];
    g[0] = gather(cur[0]);
#endif
    auto trip = [&](auto tail_tag, int cb) {
#pragma unroll
        for (int k = 0; k < kTrip; ++k) {
            const Chunk ch = cur[k];
#ifdef P2_DBUF
            if (k + 1 < kTrip) {
                g[(k + 1) & 1] = gather(cur[k + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < kTrip; ++j) { cur[j] = nx[j]; nx[j] = fetch(cb + lane + (2 * kTrip + j) * nthreads); }
                g[0] = gather(cur[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            compute(tail_tag, ch, g[k & 1], cb + k * nthreads + lane);
#else
            const G g = gather(ch);
            if (k + 1 == kTrip) {
#pragma unroll
                for (int j = 0; j < kTrip; ++j) { cur[j] = nx[j]; nx[j] = fetch(cb + lane + (2 * kTrip + j) * nthreads); }
            }
            __builtin_amdgcn_sched_barrier(0);
            compute(tail_tag, ch, g, cb + k * nthreads + lane);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int lim = ALIGNED ? c1 : min(c1, P >> 2);
    int cb = w0;
    for (; cb + (2 * kTrip - 1) * nthreads + 64 <= lim; cb += 2 * nthreads * kTrip) {
#pragma nounroll
        for (int hh = 0; hh < 2; ++hh) trip(std::false_type{}, cb + hh * nthreads * kTrip);     // (ONE copy of the trip: two spill like kTrip = 4)
        bm.flush(mo);
    }
    if (cb < c1) {                                               // at most one ragged pair per wave
#pragma nounroll
        for (int hh = 0; hh < 2; ++hh)
            if (cb + hh * nthreads * kTrip < c1) trip(std::true_type{}, cb + hh * nthreads * kTrip);
        bm.flush(mo);
    }
    n_tissue += cnt_t;
}

struct P2SweepArgs {
    const uint8_t* rgb;
    int P, parts, n_items;
    float ylimf;
    const double* state;
    P2List list;
    double* partials;        // [grid][16]
};

template <bool ALIGNED>
__global__ __launch_bounds__(kSweepThreads, 4) void k_p2_sweep(P2SweepArgs a) {
    __shared__ RowTab s_tab;
    __shared__ uint32_t s_stage[kSweepThreads / 64][kP2Stage];
    __shared__ double s_red[kSweepThreads / 64][12];
    const double* st = a.state;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)st[kWhy] != 0) {                                   // declined: the chain ends in a miss, nothing to sweep for
        if (tid < 16) a.partials[(size_t)blockIdx.x * 16 + tid] = 0.0;
        return;
    }
    s_tab.fill_b();
    __syncthreads();
    const TabReaderB T = TabReaderB::make(s_tab);
    P2SweepK K;
    for (int c = 0; c < 3; ++c) { K.gH[c] = in_vgpr((float)st[kSw + c]); K.gL[c] = in_vgpr((float)st[kSw + 3 + c]); K.n[c] = in_vgpr((float)st[kSw + 6 + c]); }
    K.k1 = in_vgpr((float)st[kSw + 9]);
    for (int i = 0; i < 2; ++i) {
        for (int c = 0; c < 3; ++c) K.W[i][c] = in_vgpr((float)st[kSw + 10 + 3 * i + c]);
        K.kt[i] = in_vgpr((float)st[kSw + 16 + i]); K.eps[i] = in_vgpr((float)st[kSw + 18 + i]);
        K.zeta[i] = in_vgpr((float)st[kSw + 20 + i]); K.thr[i] = in_vgpr((float)st[kSw + 22 + i]);
    }
    K.ylimf = in_vgpr(a.ylimf);
    BlockSink sink;
    sink.buf = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(&s_stage[wave][0]));
    {
        const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kSweepThreads / 64) + wave));
        sink.n = 0; sink.blk = wid * a.list.priv; sink.fill = 0;
        sink.priv_end = (wid + 1u) * a.list.priv; sink.pool0 = a.list.pool0;
    }
    sink.entries = a.list.entries; sink.counts = a.list.counts; sink.n_blocks = a.list.n_blocks; sink.cap_blocks = a.list.cap_blocks;
    Moments mo;
    unsigned long long n_tissue = 0;
    const bool stream = (size_t)a.P * 3 >= kStreamBytes;
    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int tile = item / a.parts, part = item % a.parts;
        const uint8_t* src = a.rgb + (size_t)tile * a.P * 3;
        int c0, c1;
        part_range((a.P + 3) >> 2, a.parts, part, c0, c1);
        if (c0 >= c1) continue;
        if (stream) p2_sweep_part<ALIGNED, P2_TRIP, true>(src, a.P, c0, c1, tid, T, K, sink, mo, n_tissue);
        else p2_sweep_part<ALIGNED, P2_TRIP, false>(src, a.P, c0, c1, tid, T, K, sink, mo, n_tissue);
    }
    sink.close(lane);
    double v[12];
    mo.to_array(v, 0u, lane);
    v[0] = lane == 0 ? (double)n_tissue : 0.0;
    v[10] = v[11] = 0.0;
#pragma unroll
    for (int i = 0; i < 10; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0)
        for (int i = 0; i < 12; ++i) s_red[wave][i] = v[i];
    __syncthreads();
    if (tid < 12) {
        double t = 0;
        for (int wv = 0; wv < kSweepThreads / 64; ++wv) t += s_red[wv][tid];
        a.partials[(size_t)blockIdx.x * 16 + tid] = t;
    }
}

// ------------------------------------------------------------------------------------------
// decision step after F: the exact eigenvectors and the plane check (ts_verify), ranks and grids of the angular stage
// ------------------------------------------------------------------------------------------
// the first window of a target: the keys from lo to hi in kP2WinBins bins of 2^sh consecutive binary32 values
__device__ __forceinline__ void p2_set_window(double* st, int t, double lo, double hi) {
    const uint32_t olo = f2ord((float)lo);
    uint32_t ohi = f2ord((float)hi);
    if (ohi < olo) ohi = olo;
    const unsigned long long span = (unsigned long long)(ohi - olo) + 1ull;
    int sh = 0;                                                  // single keys when kP2KeyBins of them span the bracket, else kP2WinBins coarse bins
    if (span > (unsigned long long)kP2KeyBins)
        while (((span + (1ull << sh) - 1ull) >> sh) > (unsigned long long)kP2WinBins) ++sh;
    for (int j = 0; j < 2; ++j) {                                // both ranks of the pair start in the same window
        st[kWinLo + 2 * t + j] = (double)olo;
        st[kSh + 2 * t + j] = (double)sh;
    }
}

__global__ void k_p2_exact(const double* tot, double* st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double pct = st[kPct];
    double Vd[6] = {0, 0, 0, 0, 0, 0};
    float Vf[6] = {0, 0, 0, 0, 0, 0};
    // (a declined route swept nothing: its totals are zeros and say nothing about the slide -- a miss, not "no tissue")
    const bool declined = (int)st[kWhy] != 0;
    const int status = declined ? (int)SL_TILE_OK : eigvecs_from_moments(tot, Vd, Vf);
    st[kStatus] = (double)status;
    st[kT] = tot[0]; st[kNpx] = tot[12];
    st[kOvf] = tot[13];
    for (int i = 0; i < 6; ++i) { st[kVd + i] = Vd[i]; st[kVf + i] = (double)Vf[i]; }
    int miss = (int)st[kMiss];
    if ((int)st[kWhy] != 0) miss |= kMissDeclined;
    if (tot[13] > 0.0) miss |= kMissOverflow;
    long long k;
    double g;
    percentile_pos(tot[0], 100.0 - pct, k, g);
    st[kK] = (double)k; st[kG] = g;
    percentile_pos(tot[0], pct, k, g);
    st[kK + 1] = (double)k; st[kG + 1] = g;
    st[kSub] = st[kSub + 1] = 0.0;                   // set by sl_pool2_step from the list's entry count
    st[kDone] = 0.0; st[kLevel] = 0.0;
    if (status == SL_TILE_OK && !(miss & kMissDeclined)) {
        // ts_verify: the exact unit normal against the sample's, the lines of the exact projection
        double gH[3], gL[3], nh[3];
        for (int c = 0; c < 3; ++c) { gH[c] = st[kGH + c]; gL[c] = st[kGL + c]; nh[c] = st[kNh + c]; }
        double n[3] = {Vd[2] * Vd[5] - Vd[4] * Vd[3], Vd[4] * Vd[1] - Vd[0] * Vd[5], Vd[0] * Vd[3] - Vd[2] * Vd[1]};
        const double nn = 1.0 / sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        const double sgn = (n[0] * nh[0] + n[1] * nh[1] + n[2] * nh[2]) < 0.0 ? -nn : nn;
        double dn = 0.0;
        for (int c = 0; c < 3; ++c) { n[c] *= sgn; dn = fmax(dn, fabs(n[c] - nh[c])); }
        const double cH = fabs(n[0] * gH[0] + n[1] * gH[1] + n[2] * gH[2]), cL = fabs(n[0] * gL[0] + n[1] * gL[1] + n[2] * gL[2]);
        const double cmax = fmax(cH, cL);
        bool ok = (cmax <= st[kK1]) & (cmax * dn <= st[kK2]);
        const double qH[2] = {Vd[0] * gH[0] + Vd[2] * gH[1] + Vd[4] * gH[2], Vd[1] * gH[0] + Vd[3] * gH[1] + Vd[5] * gH[2]};
        const double qL[2] = {Vd[0] * gL[0] + Vd[2] * gL[1] + Vd[4] * gL[2], Vd[1] * gL[0] + Vd[3] * gL[1] + Vd[5] * gL[2]};
        const double rHx = qH[1], rHy = -qH[0], rLx = -qL[1], rLy = qL[0];
        ok = ok & (rHx > 1e-3) & (rLx > 1e-3);
        const double pH = rHy / (fabs(rHx) + fabs(rHy)), pL = rLy / (fabs(rLx) + fabs(rLy));
        ok = ok & (pH + 4.0 * (double)kAngleMargin < pL);
        st[kDiag] = cmax; st[kDiag + 1] = dn;
        if (!ok) miss |= kMissPlane;
        // the outer ends of the brackets, carried over by the rotation inside the plane and padded (they only size the grids)
        const double pad = 4.0 * st[kTau] + 1e-3;
        double ends[2] = {st[kBrk], st[kBrk + 3]};
        for (int e = 0; e < 2; ++e) {
            const double p = ends[e];
            if (p > -1.0 && p < 1.0) {
                const double dx = 1.0 - fabs(p), dy = p;
                double v3[3];
                for (int c = 0; c < 3; ++c) v3[c] = st[kVhD + 2 * c] * dx + st[kVhD + 2 * c + 1] * dy;
                const double x = Vd[0] * v3[0] + Vd[2] * v3[1] + Vd[4] * v3[2], y = Vd[1] * v3[0] + Vd[3] * v3[1] + Vd[5] * v3[2];
                ends[e] = x > 0.0 ? y / (x + fabs(y)) + (e == 0 ? -pad : pad) : (e == 0 ? -2.0 : 2.0);
            } else {
                ends[e] = e == 0 ? -2.0 : 2.0;
            }
        }
        const double br1 = (double)((float)pH - kAngleMargin), br2 = (double)((float)pL + kAngleMargin);
        if (!(ends[0] < br1)) ends[0] = -2.0;
        if (!(ends[1] > br2)) ends[1] = 2.0;
        st[kBr] = ends[0]; st[kBr + 1] = br1; st[kBr + 2] = br2; st[kBr + 3] = ends[1];
        p2_set_window(st, 0, ends[0], br1);
        p2_set_window(st, 1, br2, ends[1]);
    }
    st[kMiss] = (double)miss;
    if (miss || status != SL_TILE_OK) {                   // nothing to select: the chain ends unusable (NaN in M, maxC: the apply pass copies)
        st[kDone] = 3.0;
        const double nan = nan_d();
        for (int i = 0; i < 6; ++i) st[kM + i] = nan;
        st[kMaxC] = st[kMaxC + 1] = nan;
    }
}

// One level of the exact selection on the candidates: the all-reduced window histograms -> for every target (rank k and rank k + 1 of
// each of the two order statistics) the bin that holds its rank; a window of single keys (sh = 0) settles the target's key, a coarser one
// narrows its window to the bin (11 key bits per level, the last level kP2KeyBins single keys: three levels settle any bracket).  The two
// ranks of a pair share window and histogram until they fall into different bins -- from there each descends on its own, so sparse keys
// (a small slide) and heavily tied ones (few colours) are settled like dense ones.  Settling the angular stage: the stain matrix
// (macenko_stain_extractor.py:33-44), merged_verify, the windows of the concentration stage; settling that: maxC (normalizer.py:36,47).
// A settled key set ignores further calls (and sl_pool2_hist skips its pass).
__global__ __launch_bounds__(1024) void k_p2_step(double* st, const unsigned long long* hist, int keyset) {
    __shared__ P2Scan S;
    __shared__ float s_res[4];
    __shared__ double s_nwlo[4];
    __shared__ int s_nsh[4];
    __shared__ int s_miss;
    const int tid = threadIdx.x;
    const int bit = keyset == SL_KEYSET_ANGLE ? kMissAngle : kMissConc;
    if ((int)st[kDone] & (1 << keyset)) return;                 // uniform
    if (tid == 0) s_miss = 0;
    if (tid < 4) s_res[tid] = 0.0f;
    const double N = keyset == SL_KEYSET_ANGLE ? st[kT] : st[kNpx];
    // the pixels that are NOT on the list were proven plain: inside the cone (above the lower bracket, below the upper one) / below both
    // concentration thresholds.  Lower angular pair: ranks count from the list's start; every other pair: shifted by their number.
    const double n_listed = (double)p2_tail(hist, 4);
    const double sub[2] = {keyset == SL_KEYSET_ANGLE ? 0.0 : N - n_listed, N - n_listed};
    __syncthreads();
    bool exact = true;
    for (int t = 0; t < 2; ++t) {
        const double kd = st[kK + t];
        const long long kg = (long long)(kd < 0 ? 0 : (kd > N - 1.0 ? N - 1.0 : kd));
        const long long rank[2] = {kg, (double)(kg + 1) <= N - 1.0 ? kg + 1 : kg};
        const bool shared = st[kWinLo + 2 * t + 1] == st[kWinLo + 2 * t] && st[kSh + 2 * t + 1] == st[kSh + 2 * t];     // uniform
        for (int j = 0; j < 2; ++j) {
            const int u = 2 * t + j, uh = (shared && j == 1) ? u - 1 : u;       // uh: the target whose histogram holds this rank
            const unsigned long long* h = hist + kTailWords + kP2KeyBins * uh;
            const uint32_t lo = (uint32_t)st[kWinLo + u];
            const int sh = (int)st[kSh + u];
            const int nb = sh == 0 ? kP2KeyBins : kP2WinBins;
            if (uh == u) p2_scan_build(S, h, nb, tid);                        // (a shared histogram: the scan of j = 0 is still there)
            const unsigned long long below = p2_tail(hist, uh);
            const long long kc = rank[j] - (long long)sub[t];
            const long long bin = N >= 1.0 && kc >= 0 ? p2_scan_locate(S, h, nb, kc, below, tid) : -1;
            if (bin < 0 || bin >= nb) { if (tid == 0) s_miss = 1; continue; }      // uniform
            if (sh == 0) {
                if (tid == 0) { s_res[u] = ord2f(lo + (uint32_t)bin); s_nwlo[u] = (double)lo; s_nsh[u] = 0; }
            } else {
                exact = false;
                // the bin's 2^sh keys: kP2KeyBins single keys if they fit, else kP2WinBins bins of 2^(sh - 11)
                if (tid == 0) { s_nwlo[u] = (double)lo + (double)((unsigned long long)bin << sh); s_nsh[u] = sh > kP2KeyBits ? sh - kP2WinBits : 0; }
            }
        }
    }
    __syncthreads();
    auto poison = [&]() {
        const double nan = nan_d();
        for (int i = 0; i < 6; ++i) st[kM + i] = nan;
        st[kMaxC] = st[kMaxC + 1] = nan;
    };
    int miss = (int)st[kMiss];
    const int level = (int)st[kLevel];
    if (s_miss || (!exact && level >= 2)) {                       // a rank outside its window, or three levels were not enough
        if (tid == 0) {
            st[kMiss] = (double)(miss | bit);
            st[kDone] = 3.0;
            poison();
        }
        return;
    }
    if (!exact) {
        if (tid == 0) {
            for (int u = 0; u < 4; ++u) { st[kWinLo + u] = s_nwlo[u]; st[kSh + u] = (double)s_nsh[u]; }
            st[kLevel] = (double)(level + 1);
        }
        return;
    }
    if (keyset == SL_KEYSET_ANGLE) {
        // every tissue pixel that is not on the list has its key strictly inside (br1, br2): the lower pair must end at or below br1,
        // the upper pair must start at or above br2
        if (!((double)s_res[1] <= st[kBr + 1] && (double)s_res[2] >= st[kBr + 2])) miss |= bit;
        __shared__ double s_Vd[6], s_g[2];
        if (tid < 6) s_Vd[tid] = st[kVd + tid];
        if (tid < 2) s_g[tid] = st[kG + tid];
        __syncthreads();
        if (tid < 64) {
            double M[6];
            stain_matrix_from_angles(s_Vd, s_res, s_g, M, tid);
            if (tid == 0) {
                st[kNa] = n_listed;
                for (int i = 0; i < 4; ++i) st[kRes + i] = (double)s_res[i];
                for (int i = 0; i < 6; ++i) st[kM + i] = M[i];
                if (!(miss & bit) && stain_matrix_singular(M)) st[kStatus] = (double)SL_TILE_DEGENERATE_COV;
                const MergedConc mk = *reinterpret_cast<const MergedConc*>(st + kMk);
                if (!(miss & (bit | kMissDeclined)) && !merged_verify(mk, M, st[kLam])) miss |= kMissBox;
                long long k;
                double g;
                percentile_pos(st[kNpx], 99.0, k, g);                       // normalizer.py:36,47
                st[kK] = st[kK + 1] = (double)k;
                st[kG] = st[kG + 1] = g;
                for (int i = 0; i < 2; ++i) {
                    const double L = (double)mk.L[i];
                    const double H = mk.H[i] < INFINITY ? (double)mk.H[i] : 2.0 * L + 1.0;
                    p2_set_window(st, i, L, H);
                }
                st[kLevel] = 0.0;
                st[kMiss] = (double)miss;
                st[kDone] = miss ? 3.0 : 1.0;
                if (miss) poison();
            }
        }
    } else if (tid == 0) {
        const MergedConc mk = *reinterpret_cast<const MergedConc*>(st + kMk);
        // every pixel that is not on the list has both concentrations below L_i: the k-th key must not lie below it
        if (!(s_res[0] >= mk.L[0] && s_res[2] >= mk.L[1])) miss |= bit;
        st[kNc] = n_listed;
        for (int i = 0; i < 4; ++i) st[kRes + i] = (double)s_res[i];
        for (int t = 0; t < 2; ++t) st[kMaxC + t] = np_lerp((double)s_res[2 * t], (double)s_res[2 * t + 1], st[kG + t]);
        st[kMiss] = (double)miss;
        st[kDone] = 3.0;
        if (miss != 0 || (int)st[kStatus] != SL_TILE_OK) poison();
    }
}

int p2_check(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int slog, const void* ws, size_t ws_bytes, SlParams& p, P2Layout& L) {
    if (!rgb || n <= 0 || h <= 0 || w <= 0 || slog < 0 || slog > 12) return SL_ERR_BADARG;
    if ((long)h * w > (1L << 30) || (long)n * h * w > (1L << 40)) return SL_ERR_BADARG;
    sl_default_params(&p);
    if (!params_ok(params)) return SL_ERR_BADARG;
    if (params) p = *params;
    L = p2_layout(n, h, w, slog);
    if (!ws || ws_bytes < L.total || ((uintptr_t)ws & 255u)) return SL_ERR_WORKSPACE;
    return SL_OK;
}

}  // namespace

extern "C" size_t sl_pool2_workspace_bytes(int n, int h, int w, int sample_log2) {
    if (n <= 0 || h <= 0 || w <= 0 || sample_log2 < 0 || sample_log2 > 12 || (long)h * w > (1L << 30) || (long)n * h * w > (1L << 40)) return 0;
    return p2_layout(n, h, w, sample_log2).total;
}

extern "C" int sl_pool2_sample(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int sample_log2, void* workspace,
                               size_t workspace_bytes, double* moments16_out, void* stream) {
    SlParams p;
    P2Layout L;
    const int rc = p2_check(rgb, n, h, w, params, sample_log2, workspace, workspace_bytes, p, L);
    if (rc) return rc;
    if (!moments16_out) return SL_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    uint8_t* ws = (uint8_t*)workspace;
    zero_async(ws, 256, s);                                     // (a kernel: see common.hip)
    P2SampleArgs a;
    a.rgb = rgb; a.P = h * w; a.parts = L.parts; a.n_items = L.n_items; a.slog = sample_log2; a.bpi = L.bpi;
    a.ylimf = (float)y_limit_for_threshold(p.luminosity_threshold) - 2048.0f;
    a.list = P2List{(uint32_t*)(ws + L.s_entries), (uint32_t*)(ws + L.s_counts), (unsigned int*)(ws + L.hdr), L.s_cap, kP2SampleBlkLog2, 0u, 0u};
    a.partials = (double*)(ws + L.partials);
    const dim3 g((unsigned)L.grid), b(kSweepThreads);
    if (aligned4(rgb, (long)h * w)) hipLaunchKernelGGL((k_p2_sample<true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_p2_sample<false>), g, b, 0, s, a);
    hipLaunchKernelGGL(k_p2_sum, dim3(1), dim3(1024), 0, s, (const double*)a.partials, L.grid, 11, moments16_out, 0.0, -1,
                       (const unsigned int*)(ws + L.hdr), L.s_cap, 11);
    return launch_status();
}

extern "C" int sl_pool2_begin(const double* moments16_reduced, const SlParams* params, int sample_log2, double* state, void* stream) {
    if (!moments16_reduced || !state || sample_log2 < 0 || sample_log2 > 12) return SL_ERR_BADARG;
    SlParams p;
    sl_default_params(&p);
    if (!params_ok(params)) return SL_ERR_BADARG;
    if (params) p = *params;
    hipLaunchKernelGGL(k_p2_begin, dim3(1), dim3(64), 0, (hipStream_t)stream, moments16_reduced, state, p.angular_percentile, p.lasso_lambda,
                       sample_log2);
    return launch_status();
}

extern "C" int sl_pool2_hist(int which, int keyset, int mode, int n, int h, int w, const SlParams* params, int sample_log2, const double* state,
                             void* workspace, size_t workspace_bytes, unsigned long long* hist, void* stream) {
    SlParams p;
    P2Layout L;
    static const uint8_t dummy = 0;
    const int rc = p2_check(&dummy, n, h, w, params, sample_log2, workspace, workspace_bytes, p, L);
    if (rc) return rc;
    if (!state || !hist || (which != 0 && which != 1) || (mode != 0 && mode != 1) || (keyset != SL_KEYSET_ANGLE && keyset != SL_KEYSET_CONC))
        return SL_ERR_BADARG;
    if (mode != which) return SL_ERR_BADARG;                    // the sample is histogrammed on a grid, the candidates in windows
    uint8_t* ws = (uint8_t*)workspace;
    P2HistArgs a;
    a.state = state;
    long entries;
    if (which == 0) {
        a.list = P2List{(uint32_t*)(ws + L.s_entries), (uint32_t*)(ws + L.s_counts), (unsigned int*)(ws + L.hdr), L.s_cap, kP2SampleBlkLog2, 0u, 0u};
        a.need = keyset == SL_KEYSET_ANGLE ? (kBitTissue | kBitValid) : kBitValid;
        a.basis = keyset == SL_KEYSET_ANGLE ? kVhF : 0;
        a.fourth = keyset == SL_KEYSET_ANGLE ? 1 : 0;
        entries = (long)L.s_cap << kP2SampleBlkLog2;
    } else {
        a.list = P2List{(uint32_t*)(ws + L.c_entries), (uint32_t*)(ws + L.c_counts), (unsigned int*)(ws + L.hdr + 64), L.c_cap, kP2CandBlkLog2, 0u, 0u};
        a.need = keyset == SL_KEYSET_ANGLE ? kBitAng : kBitConc;
        a.basis = keyset == SL_KEYSET_ANGLE ? kVf : 1;
        a.fourth = 0;
        entries = (long)L.c_cap << kP2CandBlkLog2;
    }
    // grid mode: few, fat workgroups (each writes a 64 KB partial histogram); window mode: many
    long blocks = (entries + 8 * kP2ListThreads - 1) / (8 * kP2ListThreads);
    if (blocks > L.hist_wgs) blocks = L.hist_wgs;
    if (blocks < 1) blocks = 1;
    a.part = (uint32_t*)(ws + L.hpart);
    a.tpart = (unsigned long long*)(ws + L.tpart);
    hipStream_t s = (hipStream_t)stream;
    a.hist = hist;
    const dim3 g((unsigned)blocks), b(kP2ListThreads);
    if (keyset == SL_KEYSET_ANGLE) {
        if (mode == 0) hipLaunchKernelGGL((k_p2_hist<SL_KEYSET_ANGLE, 0>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_p2_hist<SL_KEYSET_ANGLE, 1>), g, b, 0, s, a);
    } else {
        if (mode == 0) hipLaunchKernelGGL((k_p2_hist<SL_KEYSET_CONC, 0>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_p2_hist<SL_KEYSET_CONC, 1>), g, b, 0, s, a);
    }
    hipLaunchKernelGGL(k_p2_gsum, dim3(2 * kP2GridBins / 256, kGsumSlices), dim3(256), 0, s, (const uint32_t*)a.part,
                       (const unsigned long long*)a.tpart, (int)blocks, hist, state, mode == 1 ? (1 << keyset) : 0);
    return launch_status();
}

extern "C" int sl_pool2_bands(double* state, int keyset, const unsigned long long* hist_reduced, void* stream) {
    if (!state || !hist_reduced || (keyset != SL_KEYSET_ANGLE && keyset != SL_KEYSET_CONC)) return SL_ERR_BADARG;
    hipLaunchKernelGGL(k_p2_bands, dim3(1), dim3(1024), 0, (hipStream_t)stream, state, hist_reduced, keyset);
    return launch_status();
}

extern "C" int sl_pool2_sweep(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int sample_log2, const double* state,
                              void* workspace, size_t workspace_bytes, double* totals16_out, void* stream) {
    SlParams p;
    P2Layout L;
    const int rc = p2_check(rgb, n, h, w, params, sample_log2, workspace, workspace_bytes, p, L);
    if (rc) return rc;
    if (!state || !totals16_out) return SL_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    uint8_t* ws = (uint8_t*)workspace;
    zero_async(ws + 64, 64, s);
    zero_async(ws + L.c_counts, 4 * (size_t)L.c_cap, s);
    P2SweepArgs a;
    a.rgb = rgb; a.P = h * w; a.parts = L.parts; a.n_items = L.n_items;
    a.ylimf = (float)y_limit_for_threshold(p.luminosity_threshold) - 2048.0f;
    a.state = state;
    const uint32_t pool0 = (uint32_t)L.grid * (kSweepThreads / 64) * L.c_priv;
    a.list = P2List{(uint32_t*)(ws + L.c_entries), (uint32_t*)(ws + L.c_counts), (unsigned int*)(ws + L.hdr + 64), L.c_cap, kP2CandBlkLog2, L.c_priv, pool0};
    a.partials = (double*)(ws + L.partials);
    const dim3 g((unsigned)L.grid), b(kSweepThreads);
    if (aligned4(rgb, (long)h * w)) hipLaunchKernelGGL((k_p2_sweep<true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_p2_sweep<false>), g, b, 0, s, a);
    hipLaunchKernelGGL(k_p2_sum, dim3(1), dim3(1024), 0, s, (const double*)a.partials, L.grid, 12, totals16_out, (double)n * h * w, 12,
                       (const unsigned int*)(ws + L.hdr + 64), L.c_cap - pool0, 13);
    return launch_status();
}

extern "C" int sl_pool2_exact(const double* totals16_reduced, double* state, void* stream) {
    if (!totals16_reduced || !state) return SL_ERR_BADARG;
    hipLaunchKernelGGL(k_p2_exact, dim3(1), dim3(64), 0, (hipStream_t)stream, totals16_reduced, state);
    return launch_status();
}

extern "C" int sl_pool2_step(double* state, int keyset, const unsigned long long* hist_reduced, void* stream) {
    if (!state || !hist_reduced || (keyset != SL_KEYSET_ANGLE && keyset != SL_KEYSET_CONC)) return SL_ERR_BADARG;
    hipLaunchKernelGGL(k_p2_step, dim3(1), dim3(1024), 0, (hipStream_t)stream, state, hist_reduced, keyset);
    return launch_status();
}

// The whole chain on ONE process (no collective between the steps): the ~45 launches enqueued by one call -- issued from Python one by one
// they take the host longer than a 512-tile slide takes the device.
extern "C" int sl_pool2_local(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int sample_log2, void* workspace,
                              size_t workspace_bytes, double* state, void* stream) {
    SlParams p;
    P2Layout L;
    int rc = p2_check(rgb, n, h, w, params, sample_log2, workspace, workspace_bytes, p, L);
    if (rc) return rc;
    if (!state) return SL_ERR_BADARG;
    uint8_t* ws = (uint8_t*)workspace;
    double* mom = (double*)(ws + L.local);
    double* tot = mom + 16;
    unsigned long long* hist = (unsigned long long*)(ws + L.local + 256);
    if ((rc = sl_pool2_sample(rgb, n, h, w, params, sample_log2, workspace, workspace_bytes, mom, stream))) return rc;
    if ((rc = sl_pool2_begin(mom, params, sample_log2, state, stream))) return rc;
    for (int keyset = 0; keyset < 2; ++keyset) {
        if ((rc = sl_pool2_hist(0, keyset, 0, n, h, w, params, sample_log2, state, workspace, workspace_bytes, hist, stream))) return rc;
        if ((rc = sl_pool2_bands(state, keyset, hist, stream))) return rc;
    }
    if ((rc = sl_pool2_sweep(rgb, n, h, w, params, sample_log2, state, workspace, workspace_bytes, tot, stream))) return rc;
    if ((rc = sl_pool2_exact(tot, state, stream))) return rc;
    for (int keyset = 0; keyset < 2; ++keyset)
        for (int level = 0; level < SL_POOL2_LEVELS; ++level) {
            if ((rc = sl_pool2_hist(1, keyset, 1, n, h, w, params, sample_log2, state, workspace, workspace_bytes, hist, stream))) return rc;
            if ((rc = sl_pool2_step(state, keyset, hist, stream))) return rc;
        }
    return SL_OK;
}
