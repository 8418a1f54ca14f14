/*
 * stainlib_hip.h -- C ABI of the MI355X-native H&E stain-normalization engine.
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json.  The
 * reference (sebastianffx/stainlib v0.6.1) has no FFI: its "operator API" is a
 * handful of Python classes whose arithmetic lives in numpy / OpenCV / spams /
 * scikit-image.  Each entry point below replaces one reference call chain
 * (cited as file:line, relative to the reference checkout) and is what a
 * ctypes binding inside stainlib would call (INTEGRATION.md shows the stubs).
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer (HBM), e.g. torch tensor.data_ptr();
 *     SlParams is the only host pointer.
 *   - images: n tiles, each h x w x 3 interleaved RGB uint8, tiles contiguous
 *     (NHWC).  Stain matrices: row-major 2x3 double, row 0 = haematoxylin.
 *   - the caller owns every buffer including the workspace; the library keeps
 *     no state, allocates nothing, and is re-entrant per stream.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream) and returns without synchronising the host.
 *   - return value: 0 on success, SL_ERR_* (<0) otherwise.  No C++ exception
 *     crosses this boundary.  Per-tile conditions are reported in status[i].
 */
#ifndef STAINLIB_HIP_H
#define STAINLIB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SL_VERSION 600 /* 0.6.0: sl_pool2_* (the pooled slide statistics in one full sweep), two_sweep validated; 0.5.0: SlParams starts with struct_size (checked by every entry point), two_sweep / twosweep_out; 0.4.0: prefilter, prefilter_out; 0.3.0: fused_min_tiles, sl_pool_*, resweeps_out carries reasons */

/* The library is built with -fvisibility=hidden: exactly the functions declared here are exported. */
#if defined(__GNUC__)
#define SL_API __attribute__((visibility("default")))
#else
#define SL_API
#endif

/* return codes */
#define SL_OK 0
#define SL_ERR_BADARG (-1)    /* null pointer, n/h/w <= 0, unknown op/mode */
#define SL_ERR_WORKSPACE (-2) /* workspace missing or smaller than sl_workspace_bytes() */
#define SL_ERR_NODEVICE (-3)  /* no HIP device / wrong architecture */
#define SL_ERR_HIP_BASE (-1000) /* HIP runtime error e is returned as SL_ERR_HIP_BASE - e */

/* per-tile status[i] */
#define SL_TILE_OK 0
#define SL_TILE_EMPTY_MASK 1     /* stain_utils.py:46-47 -> TissueMaskException */
#define SL_TILE_DEGENERATE_COV 2 /* fewer than 2 tissue pixels (np.cov is NaN in the reference), or tissue of a single colour:
                                    two parallel stain vectors, inf/NaN concentrations in the reference */
#define SL_TILE_ZERO_MAXC 3      /* 99th percentile of a concentration is 0: normalizer.py:48 divides by it (inf / NaN cast to uint8
                                    in the reference).  A tile with any non-zero status is passed through unchanged by the transforms. */

/* ops for sl_workspace_bytes */
#define SL_OP_MACENKO_FIT 1
#define SL_OP_MACENKO_TRANSFORM 2
#define SL_OP_VAHADANE_FIT 3
#define SL_OP_VAHADANE_TRANSFORM 4
#define SL_OP_HED_AUGMENT 5
#define SL_OP_STAIN_AUGMENT 6
#define SL_OP_TILE_MOMENTS 7
#define SL_OP_LAB_STATS 8        /* sl_reinhard_stats, sl_reinhard_transform, sl_luminosity_standardize, sl_standardize_brightness */

/* selection key sets of the pooled slide-level mode (sl_slide_key_*): each set carries TWO targets */
#define SL_KEYSET_ANGLE 0 /* both targets: pseudo-angle of the projected OD, tissue pixels only (macenko_stain_extractor.py:29-34) */
#define SL_KEYSET_CONC 1  /* target i: lasso concentration of stain i, all pixels (normalizer.py:36,47) */

/* skimage semantics selector for sl_hed_augment.  Only 0.18 is pinned by vectors from a real scikit-image
 * (0.18.3); the other three are restated from memory and checked against the CPU restatement only -- no source or wheel
 * of those releases exists in the build environment. */
#define SL_HED_SKIMAGE_018 0 /* ln(max(rgb,1e-6))/ln(1e-6) @ hed_from_rgb  (golden-pinned) */
#define SL_HED_SKIMAGE_019 1 /* as 0.18 + stains clamped at 0 after separation  (unpinned) */
#define SL_HED_SKIMAGE_017 2 /* presumed <= 0.17 (environment.yml:107 pins 0.17.2): -ln(rgb + 2) @ hed_from_rgb,
                                exp(-stains @ rgb_from_hed) - 2, clipped  (natural logarithm; unpinned) */
#define SL_HED_EXPERIMENTAL_LOG10 3 /* the same with a base-10 logarithm (round 1's reading of 0.17; kept as an experiment, unpinned) */

/* Optional kernel timing.  When SlParams.profile is non-NULL, sl_*_fit / sl_*_transform bracket every
 * launch of the selected kernel classes with two caller-created hipEvent_t from events[] (start, stop),
 * on the same stream as the kernels, and record which class each pair belongs to.  The library only
 * records; the caller synchronises and reads hipEventElapsedTime.  Launches beyond `capacity` are
 * simply not bracketed. */
#define SL_PROF_MOMENTS 1
#define SL_PROF_SELECT_ANGLE 2
#define SL_PROF_SELECT_CONC 4
#define SL_PROF_FINISH 8
#define SL_PROF_APPLY 16
#define SL_PROF_DICT 32
#define SL_PROF_FUSED_FIT 64       /* the persistent one-workgroup-per-tile kernel, fit only */
#define SL_PROF_FUSED_TRANSFORM 128 /* the same including the apply sweep */
typedef struct SlProfile {
    void** events;      /* capacity caller-owned hipEvent_t */
    int32_t* tags;      /* capacity/2 ints, out: SL_PROF_* class of pair i */
    int32_t* tiles;     /* capacity/2 ints, out: tiles covered by the launch of pair i */
    int32_t capacity;   /* number of events available (even) */
    int32_t used;       /* in/out: events consumed so far (start at 0) */
    int32_t mask;       /* OR of SL_PROF_* classes to bracket */
    int32_t reserved;
} SlProfile;

/* Extractor constants.  The reference never forwards these from fit/transform, so the
 * defaults are effectively constants (macenko_stain_extractor.py:7,
 * vahadane_stain_extractor.py:19, stain_utils.py:69). */
#define SL_RESWEEP_NO_BOX 1         /* the sample's brackets left no usable box of stain matrices (open, or too wide) */
#define SL_RESWEEP_OUTSIDE_BOX 2    /* the exact stain matrix fell outside the box the sweep assumed */
#define SL_RESWEEP_BRACKET_MISSED 3 /* a concentration bracket did not hold the wanted rank */
#define SL_RESWEEP_LIST_FULL 4      /* the candidate list overflowed */
/* what became of a tile's two-sweep attempt (SlParams.twosweep_out) */
#define SL_TWOSWEEP_DIRECT 1        /* the tile's stain matrix came out of two read sweeps (moments + candidates in one, then the apply pass) */
#define SL_TWOSWEEP_OFF 0           /* not attempted (SlParams.two_sweep == 1, a schedule without it, or the workgroup was backing off after a decline) */
#define SL_TWOSWEEP_NO_ESTIMATE (-1) /* the cluster sample gave no usable estimate (few tissue entries, an ill-defined plane, open brackets) */
#define SL_TWOSWEEP_SHARE (-2)      /* too many of the sample's pixels in colour-cube cells the mask could not prove plain: not worth it */
#define SL_TWOSWEEP_PLANE (-3)      /* the exact eigenvector plane left the tilt the sweep allowed for: three-sweep route from the exact moments */
#define SL_TWOSWEEP_BRACKET (-4)    /* an angular bracket carried over from the estimate missed its rank (or a list overflowed): likewise */
#define SL_TWOSWEEP_LISTS (-5)      /* the sample predicts more bracket members than the candidate lists hold: not attempted */
typedef struct SlParams {
    uint32_t struct_size;        /* sizeof(SlParams) of the header the CALLER was compiled against: set by sl_default_params, checked by every
                                    entry point that takes an SlParams (SL_ERR_BADARG on a mismatch) -- a caller built against another
                                    version of this struct is refused instead of being read past its end */
    uint32_t reserved0;
    double luminosity_threshold; /* 0.8  (binary64 like the Python float the reference compares with) */
    double angular_percentile;   /* 99   */
    double lasso_lambda;        /* 0.01 */
    double dl_lambda;           /* 0.1  (Vahadane) */
    int32_t dl_max_sweeps;      /* 200  (Vahadane; the reference is wall-clock budgeted) */
    int32_t schedule;           /* 0 = automatic; 1 = one launch per phase; 2 = persistent fused kernel (two 512-thread workgroups per CU);
                                   3 = Macenko only: the fused kernel with one 1024-thread workgroup per CU where the batch has no more tiles
                                   than the device has CUs (else as 2) */
    double dl_tol;              /* 1e-7 the dictionary iteration stops when a full sweep moves D by less than this (max-abs), or
                                        when the a-posteriori estimate of its distance to the fixed point (the step the next sweep would
                                        take, predicted from the last two) is below it */
    SlProfile* profile;         /* NULL (default): no timing events */
    int32_t* fallbacks_out;     /* NULL (default) or DEVICE pointer to n ints: per tile, how many of its four order statistics
                                   (two angular, two concentration percentiles) needed the slow exact selection over the whole
                                   tile because the sampled bracket missed or its candidate list overflowed (diagnostics;
                                   results never depend on it).  Written by sl_macenko_* / sl_vahadane_*. */
    int32_t* resweeps_out;      /* NULL (default) or DEVICE pointer to n ints: nonzero (an SL_RESWEEP_* reason, below) for a tile whose concentration percentiles needed a
                                   selection sweep of their own (the persistent Macenko kernel collects the angular and the
                                   concentration candidates in ONE sweep under a sample estimate of the stain matrix and repeats
                                   the concentration part when the exact matrix falls outside the assumed box; diagnostics).
                                   Written by the fused schedule of sl_macenko_*; left untouched otherwise. */
    int32_t fused_min_tiles;    /* schedule == 0 only: batches of at least this many tiles run the persistent fused kernel, smaller ones
                                   one launch per phase.  0 (default) = the library's measured crossovers on an MI355X at its 1400 W
                                   power state (tools/crossover.py, profiles/r04_crossover*.txt): Macenko 352 tiles (320 for tiles below 512 Ki
                                   pixels, 288 up to 256 Ki), Vahadane 640 (192 below 512 Ki pixels; 704 when the batch exceeds one resident grid).  A Macenko batch larger than the fused
                                   kernel's resident grid (2 x compute units) is split: whole fused rounds, and a remainder below this
                                   number of tiles one launch per phase (default: 208 / 224 / 192 tiles by the same tile sizes, never for
                                   tiles up to 64 Ki pixels).  Results do not depend on the schedule. */
    int32_t prefilter;          /* the colour-cube pre-filter of the fused Macenko kernel's selection sweep (a 32^3-cell mask of colours that are
                                   provably "plain", built per tile by finish 1; pixels of the other cells are re-tested exactly):
                                   0 (default) = per tile, wherever the tile's pixel sample says it pays; 1 = never (the per-pixel sweep);
                                   2 = wherever the mask can be built, whatever the sample says (tests).  Results do not depend on it. */
    int32_t* prefilter_out;     /* NULL (default) or DEVICE pointer to n ints: bit 0 set for a tile whose selection sweep ran behind the mask,
                                   bits 8.. the share (percent) of the tile's sample pixels that fell into cells the mask could not
                                   prove plain (0 when no mask was built; diagnostics).  Written by the fused schedule of sl_macenko_*;
                                   left untouched otherwise. */
    int32_t two_sweep;          /* the two-read-sweep schedule of the fused Macenko kernel (the candidates of all four order statistics are
                                   collected in the moments sweep, under eigenvectors estimated from a cluster sample gathered before it, and
                                   the finish verifies the estimate against the exact eigenvectors; a tile that fails takes the three-sweep
                                   route): 0 (default) = per tile, wherever the sample says it pays (a workgroup whose tile declined skips the attempt on its next
                                   three tiles); 1 = never; 2 = wherever an estimate
                                   exists (tests); 3 = as 2 with the verification forced to fail, 4 = as 2 with the sample's plane tilted
                                   (tests of the fallback).  Any other value: SL_ERR_BADARG.  The test modes 2-4 run the merged sweep behind
                                   the colour-cube mask even under prefilter == 1 (that sweep has no per-pixel form); in the automatic mode
                                   prefilter == 1 rules the route out.  Results do not depend on it.  On real tissue the automatic mode
                                   mostly declines after its sample's eigen-solve (+2-3 % against two_sweep = 1 on such batches, -6-10 % on
                                   synthetic ones: DESIGN 4.1). */
    int32_t reserved1;
    int32_t* twosweep_out;      /* NULL (default) or DEVICE pointer to n ints: SL_TWOSWEEP_* per tile (diagnostics).  Written by the fused
                                   schedule of sl_macenko_*; left untouched otherwise. */
} SlParams;

/* sl_version() == SL_VERSION must be checked BEFORE any other call: sl_default_params writes sizeof(SlParams) of THIS library's
 * header into the caller's struct -- a caller compiled against a header with a smaller SlParams would be written past its end before
 * any struct_size check could run (the ctypes binding refuses a library of another version: stainlib_amd/_ffi.py). */
SL_API int sl_version(void);
SL_API const char* sl_error_string(int code);
SL_API void sl_default_params(SlParams* p);

/* Bytes of device workspace an op needs for n tiles of h x w.  0 for ops that need none.
 * The layout depends on the number of compute units of the CURRENT HIP device (persistent grids are sized to it): call this
 * with the device current on which the operator will be launched (same partition mode); a workspace sized under another
 * device may be refused with SL_ERR_WORKSPACE, never overrun. */
SL_API size_t sl_workspace_bytes(int op, int n_tiles, int h, int w);
/* sl_workspace_bytes is the maximum over every SlParams (a workspace of that size serves any call of the op at this batch shape).
 * What ONE call needs -- the schedule its SlParams (NULL: the defaults) select: the one-launch-per-phase schedule has no angular
 * candidate list, 28 % less per tile -- is sl_workspace_bytes_for; the entry points check against this figure.  0: bad arguments. */
SL_API size_t sl_workspace_bytes_for(int op, int n_tiles, int h, int w, const SlParams* params);

/* MacenkoStainExtractor.get_stain_matrix (extraction/macenko_stain_extractor.py:7-44)
 * + get_concentrations (utils/stain_utils.py:69-78) + np.percentile(C, 99, axis=0)
 * (normalization/normalizer.py:34-36), for each of n tiles.
 *   M_out    n x 2 x 3 double   unit-norm rows, H first
 *   maxC_out n x 2 double
 *   status   n int32 */
SL_API int sl_macenko_fit(const uint8_t* rgb, int n, int h, int w, const SlParams* params,
                   double* M_out, double* maxC_out, int32_t* status,
                   void* workspace, size_t workspace_bytes, void* stream);

/* VahadaneStainExtractor.get_stain_matrix (extraction/vahadane_stain_extractor.py:19-43) with
 * spams.trainDL replaced by the converged optimum of the same objective, then as above.  "The" optimum of this
 * non-convex problem is the point plain full-batch block-coordinate descent reaches from Ruifrok's H and E vectors
 * (oracle/stain_oracle.py vahadane_dictionary); the device iteration is an acceleration of that scheme that takes a
 * step back whenever the objective rose or an atom lost all its pixels, so that it stays on the plain scheme's path.
 *   sweeps_out  n int32 (may be NULL): dictionary sweeps used per tile */
SL_API int sl_vahadane_fit(const uint8_t* rgb, int n, int h, int w, const SlParams* params,
                    double* M_out, double* maxC_out, int32_t* status, int32_t* sweeps_out,
                    void* workspace, size_t workspace_bytes, void* stream);

/* The OD + reconstruction pass: get_concentrations on every pixel with the tile's own
 * stain matrix, rescale by maxC_tgt / maxC_src, 255*exp(-C @ M_tgt), truncating uint8 cast
 * (utils/stain_utils.py:69-78,101-112 ; normalization/normalizer.py:46-50).
 *   M_src n x 2 x 3, maxC_src n x 2 (per tile); M_tgt 2 x 3, maxC_tgt 2 (shared)
 *   prequant  optional n x P x 3 float: the values before the uint8 cast (parity tests) */
SL_API int sl_normalize_apply(const uint8_t* rgb, uint8_t* out, int n, int h, int w,
                       const double* M_src, const double* maxC_src,
                       const double* M_tgt, const double* maxC_tgt,
                       double lasso_lambda, float* prequant, void* stream);

/* ExtractiveStainNormalizer('macenko').transform (normalization/normalizer.py:39-50) for a
 * batch: per-tile fit stages + the apply pass in one cache-friendly schedule.
 *   M_src_out n x 2 x 3 / maxC_src_out n x 2 may be NULL. */
SL_API int sl_macenko_transform(const uint8_t* rgb, uint8_t* out, int n, int h, int w,
                         const SlParams* params, const double* M_tgt, const double* maxC_tgt,
                         double* M_src_out, double* maxC_src_out, int32_t* status,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ExtractiveStainNormalizer('vahadane').transform, same shape. */
SL_API int sl_vahadane_transform(const uint8_t* rgb, uint8_t* out, int n, int h, int w,
                          const SlParams* params, const double* M_tgt, const double* maxC_tgt,
                          double* M_src_out, double* maxC_src_out, int32_t* status,
                          void* workspace, size_t workspace_bytes, void* stream);

/* HedColorAugmenter.transform (augmentation/augmenter.py:276-331) for uint8 tiles:
 * cutoff test on the tile mean, rgb2hed, per-channel x*(1+sigma)+bias, hed2rgb, clip, *255,
 * truncate.  sigma, bias: n x 3 double (H, E, D).  applied[i] (may be NULL) = 0 when tile i failed
 * cutoff_lo <= mean/255 <= cutoff_hi; such a tile is copied through unchanged.
 * workspace: sl_workspace_bytes(SL_OP_HED_AUGMENT, n, h, w) = 8 n bytes. */
SL_API int sl_hed_augment(const uint8_t* rgb, uint8_t* out, int n, int h, int w,
                   const double* sigma, const double* bias, double cutoff_lo, double cutoff_hi,
                   int skimage_mode, int32_t* applied,
                   void* workspace, size_t workspace_bytes, void* stream);

/* The float branch of HedColorAugmenter.transform (augmentation/augmenter.py:288-289, 319-320): patches of
 * binary64 values in [0,1], n x h x w x 3; cutoff on np.mean(patch); binary64 arithmetic; clipped output.
 * workspace: 8 n bytes. */
SL_API int sl_hed_augment_f64(const double* rgb, double* out, int n, int h, int w,
                       const double* sigma, const double* bias, double cutoff_lo, double cutoff_hi,
                       int skimage_mode, int32_t* applied,
                       void* workspace, size_t workspace_bytes, void* stream);

/* convert_RGB_to_OD (utils/stain_utils.py:101-112) materialised: od_out n x h x w x 3 double. */
SL_API int sl_rgb_to_od(const uint8_t* rgb, int n, int h, int w, double* od_out, void* stream);

/* convert_OD_to_RGB (utils/stain_utils.py:114-124): rgb_out[i] = uint8(255 * exp(-max(od[i], 1e-6))) for n_values doubles.
 * negative_flag (device int32, may be NULL) is set to 1 when any od[i] < 0 (the reference asserts "Negative optical density."). */
SL_API int sl_od_to_rgb(const double* od, size_t n_values, uint8_t* rgb_out, int32_t* negative_flag, void* stream);

/* StainAugmentor.pop (augmentation/augmenter.py:428-449): concentrations with the tile's
 * stain matrix M (n x 2 x 3), C[:,i] = C[:,i]*alpha_i + beta_i on tissue pixels (all pixels
 * when augment_background), 255*exp(-C @ M), clip to [0,255], truncate.
 *   alpha_beta n x 4 double: alpha0, beta0, alpha1, beta1 (the reference's draw order) */
SL_API int sl_stain_augment(const uint8_t* rgb, uint8_t* out, int n, int h, int w,
                     const double* M, const double* alpha_beta, int augment_background,
                     const SlParams* params, void* stream);

/* LuminosityThresholdTissueLocator.get_tissue_mask (utils/stain_utils.py:32-48):
 * mask_out n x P uint8 (0/1, may be NULL), counts n int64 (may be NULL). */
SL_API int sl_tissue_mask(const uint8_t* rgb, int n, int h, int w, double luminosity_threshold,
                   uint8_t* mask_out, int64_t* counts, void* stream);

/* get_concentrations (utils/stain_utils.py:69-78) materialised: C_out n x P x 2 float. */
SL_API int sl_concentrations(const uint8_t* rgb, int n, int h, int w, const double* M,
                      double lasso_lambda, float* C_out, void* stream);

/* GrayscaleAugmentor.pop (augmentation/augmenter.py:390-401): out = 3 x uint8(255 * clip(rgb2gray * alpha +
 * beta, 0, 1)), binary64 arithmetic like the reference.  alpha_beta: n x 2 doubles (device). */
SL_API int sl_grayscale_augment(const uint8_t* rgb, uint8_t* out, int n, int h, int w, const double* alpha_beta,
                         void* stream);

/* ---- OpenCV 8-bit Lab family: ReinhardStainNormalizer, LuminosityStandardizer, LAB helpers.
 * cv2.cvtColor on uint8 images is restated from OpenCV's published RGB2Lab_b / Lab2RGBinteger (integer tables); no cv2 exists in
 * the build environment, so this family is PARITY UNPINNED against OpenCV itself (tools/pin_cv2.py checks it wherever cv2 is
 * installed).  The reference's own arithmetic around it (percentiles, binary32/binary64 promotion, clip-then-truncate, masking)
 * is golden-pinned.  Workspace of the four entry points that take one: sl_workspace_bytes(SL_OP_LAB_STATS, n, h, w). */

/* cv2.cvtColor(I, COLOR_RGB2LAB), uint8 (utils/stain_utils.py:62,152): lab_out n x h x w x 3 = L*255/100, a+128, b+128. */
SL_API int sl_rgb_to_lab8(const uint8_t* rgb, uint8_t* lab_out, int n, int h, int w, void* stream);
/* cv2.cvtColor(LAB, COLOR_LAB2RGB), uint8 (utils/stain_utils.py:66,172). */
SL_API int sl_lab8_to_rgb(const uint8_t* lab, uint8_t* rgb_out, int n, int h, int w, void* stream);
/* lab_split (utils/stain_utils.py:146-158): three n x h x w binary32 planes L8/2.55, a8-128, b8-128. */
SL_API int sl_lab_split(const uint8_t* rgb, int n, int h, int w, float* I1, float* I2, float* I3, void* stream);
/* merge_back (utils/stain_utils.py:160-172): I1*2.55, I2+128, I3+128 in the planes' own precision (is_f64: binary64 planes,
 * else binary32), clip [0,255], truncate, LAB2RGB.  The planes are not modified (the reference scales them in place). */
SL_API int sl_lab_merge(const void* I1, const void* I2, const void* I3, int is_f64, int n, int h, int w, uint8_t* rgb_out, void* stream);

/* standardize_brightness (utils/stain_utils.py:188-194): p = 90th percentile (np.percentile, linear) of ALL byte values of the
 * tile, out = uint8(clip(I * 255.0 / p, 0, 255)).  p_out: n doubles (may be NULL). */
SL_API int sl_standardize_brightness(const uint8_t* rgb, uint8_t* out, int n, int h, int w, double* p_out,
                              void* workspace, size_t workspace_bytes, void* stream);

/* get_mean_std (utils/stain_utils.py:174-186; cv2.meanStdDev of the lab_split planes: population std), optionally of the
 * brightness-standardised tile (standardize != 0: what ReinhardStainNormalizer.fit / transform feed it, normalizer.py:65-66,78-80).
 *   stats_out n x 8 double: p90 (NaN when !standardize), mean L, a, b, std L, a, b, and the number of tissue pixels of the
 *   (standardised) tile at luminosity threshold 0.8 (what transform(mask_background=True) tests for emptiness) */
SL_API int sl_reinhard_stats(const uint8_t* rgb, int n, int h, int w, int standardize, double* stats_out,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ReinhardStainNormalizer.transform (normalization/normalizer.py:70-94): standardize_brightness, lab_split, per-channel
 * (x - mean) * (target_std / std) + target_mean in binary64, optional background masking with the luminosity test of the
 * standardised tile (background -> L 254, a = b = 0 before merge_back), merge_back.
 *   target_means / target_stds: 3 doubles each (DEVICE), shared by all tiles
 *   stats_out n x 8 (may be NULL): as sl_reinhard_stats(standardize = 1), last entry = tissue pixels of the standardised tile
 *                                  (0 with mask_background -> the reference raises TissueMaskException) */
SL_API int sl_reinhard_transform(const uint8_t* rgb, uint8_t* out, int n, int h, int w, const double* target_means,
                          const double* target_stds, int mask_background, double luminosity_threshold, double* stats_out,
                          void* workspace, size_t workspace_bytes, void* stream);

/* LuminosityStandardizer.standardize (utils/stain_utils.py:52-67): p = np.percentile(L8, percentile),
 * L8 <- uint8(clip(255 * L8 / p, 0, 255)), LAB2RGB.  p_out: n doubles (may be NULL). */
SL_API int sl_luminosity_standardize(const uint8_t* rgb, uint8_t* out, int n, int h, int w, double percentile, double* p_out,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ---- pooled slide-level mode (BASELINE.json configs[4]; an extension: the reference has no notion of a slide).
 * Every tile of a slide gets the statistics the reference would compute from the vertical concatenation of all
 * the tiles.  Each call reduces THIS process's tiles; the host sums / all-reduces the small results over ranks
 * (stainlib_amd/distributed.py: PooledSlideStatistics).
 *
 * sl_tile_moments: per tile {n, sum od[3], sum od od^T [xx,xy,xz,yy,yz,zz]} over the tissue pixels, binary64,
 * run-to-run identical.  workspace: sl_workspace_bytes(SL_OP_TILE_MOMENTS, n, h, w). */
SL_API int sl_tile_moments(const uint8_t* rgb, int n, int h, int w, const SlParams* params,
                    double* moments_out /* n x 10 */, void* workspace, size_t workspace_bytes, void* stream);

/* Keys are compared as order-preserving uint32 images of their binary32 value: ord(f) = bits(f) ^ 0x80000000 for
 * f >= 0, ~bits(f) for f < 0.  `basis` is a HOST pointer to 6 doubles: V (3x2, V[c*2+k]) for SL_KEYSET_ANGLE, the
 * stain matrix M (2x3) for SL_KEYSET_CONC.  Both targets of the key set are served by the same sweep: hist
 * (device, 2 x 256 uint64) is ACCUMULATED into, target t counting bin = the 8 key bits below the top `prefix_bits`
 * bits over the keys whose top `prefix_bits` (0, 8, 16 or 24) bits equal prefixes[t] (HOST pointer, 2 values). */
SL_API int sl_slide_key_histogram(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset,
                           const double* basis, const uint32_t* prefixes, int prefix_bits,
                           unsigned long long* hist, void* stream);
/* The last two rounds in one sweep: hist16 (device, 2 x 65536 uint64, ACCUMULATED into) counts the LOW 16 key bits
 * over the keys whose top 16 bits equal prefixes16[t] (HOST pointer, 2 values). */
SL_API int sl_slide_key_histogram16(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset,
                             const double* basis, const uint32_t* prefixes16, unsigned long long* hist16, void* stream);
/* sl_slide_key_histogram over a stratified pixel sample: one 64-chunk row in 2^sample_log2 (0 <= sample_log2 <= 12). */
SL_API int sl_slide_key_histogram_sampled(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset,
                                   const double* basis, const uint32_t* prefixes, int prefix_bits, int sample_log2,
                                   unsigned long long* hist, void* stream);
/* One sweep for an order statistic whose neighbourhood is known: hist_below (device, 2 x 65536 + 2 uint64, ACCUMULATED
 * into) = per target the histogram of key - window_lo[t] over the keys in [window_lo[t], window_lo[t] + 65536), followed
 * by the two counts of keys below window_lo[t] (HOST pointer, 2 ordered-uint32 values). */
SL_API int sl_slide_key_window(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset,
                        const double* basis, const uint32_t* window_lo, unsigned long long* hist_below, void* stream);
/* min_out[t] (device uint32 x 2, set to 0xffffffff by the caller) = min(min_out[t], smallest key of target t
 * above key_ords[t]) (key_ords: HOST pointer, 2 values). */
SL_API int sl_slide_key_next_above(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset,
                            const double* basis, const uint32_t* key_ords, uint32_t* min_out, void* stream);

/* ---- the same pooled statistics, DEVICE-DRIVEN: no host read-back between the steps ----------------------------
 * `state` is SL_POOL_STATE_DOUBLES doubles of device memory owned by the caller.  One computation is the chain
 *     [sum of sl_tile_moments over the local tiles, pixel count appended: 11 doubles]  -> all-reduce ->  sl_pool_begin
 *     for keyset in (SL_KEYSET_ANGLE, SL_KEYSET_CONC):
 *         for round in 0, 1, 2:  sl_pool_histogram (sampled, 2 x 256)  -> all-reduce ->  sl_pool_pick
 *         sl_pool_window (2 x 65536 + 2)  -> all-reduce ->  sl_pool_resolve
 * enqueued on one stream; every decision (eigenvectors, which bin holds the rank, where the window sits, whether it caught the
 * ranks, the stain matrix, maxC) is taken on the device from all-reduced data, so every rank reaches the same state.  Afterwards
 * state[SL_POOL_M .. +5] is the slide's stain matrix, state[SL_POOL_MAXC .. +1] its 99th-percentile concentrations,
 * state[SL_POOL_STATUS] a SL_TILE_* code and state[SL_POOL_MISS] != 0 says a window missed its rank (an estimate off by more than
 * 32768 consecutive binary32 values, or an empty sample): the caller then takes the host-driven radix rounds above.  ONE read-back,
 * at the end.  The sweeps are the kernels of sl_slide_key_histogram_sampled / sl_slide_key_window reading their basis, prefixes
 * and windows from `state`. */
#define SL_POOL_STATE_DOUBLES 64
#define SL_POOL_M 0
#define SL_POOL_MAXC 6
#define SL_POOL_STATUS 8
#define SL_POOL_MISS 9
SL_API int sl_pool_begin(const double* moments11, const SlParams* params, double* state, void* stream);
SL_API int sl_pool_histogram(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset, const double* state,
                      int round, int sample_log2, unsigned long long* hist, void* stream);
SL_API int sl_pool_pick(double* state, int keyset, int round, const unsigned long long* hist_reduced, void* stream);
SL_API int sl_pool_window(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int keyset, const double* state,
                   unsigned long long* hist_below, void* stream);
SL_API int sl_pool_resolve(double* state, int keyset, const unsigned long long* window_reduced, const SlParams* params, void* stream);

/* ---- the pooled statistics in ONE full sweep (round 6; stainlib_amd/csrc/slide_merged.hip) ---------------------------------------
 * The moments sweep also collects the raw candidates of all four order statistics, against an estimate of the eigenvectors, the angular
 * brackets and the stain matrix taken from a stratified pixel sample of the whole slide (one 64-pixel sub-row in 2^sample_log2; the same
 * sample_log2 on every rank); once the exact moments are all-reduced the estimate is CHECKED, and the exact order statistics of the
 * binary32 keys are those of the candidates at shifted ranks.  Results never depend on the sample: when a check fails
 * (state[SL_POOL_MISS] != 0 at the end) the caller takes the three-sweep chain above.  One computation is the chain
 *     sl_pool2_sample                          -> all-reduce moments16 (16 doubles, SUM)  -> sl_pool2_begin
 *     sl_pool2_hist(0, ANGLE, 0)               -> all-reduce hist (SL_POOL2_HIST_WORDS uint64)  -> sl_pool2_bands(ANGLE)
 *     sl_pool2_hist(0, CONC, 0)                -> all-reduce                              -> sl_pool2_bands(CONC)
 *     sl_pool2_sweep  (THE full sweep)         -> all-reduce totals16 (16 doubles)        -> sl_pool2_exact
 *     for keyset in (ANGLE, CONC):  SL_POOL2_LEVELS times:  sl_pool2_hist(1, keyset, 1) -> all-reduce -> sl_pool2_step
 *         (a radix descent in the ordered binary32 domain, one target per wanted RANK -- k and k + 1 of each of the two order statistics share
 *          a window until they fall into different bins: SL_POOL2_WINDOW_BINS coarse bins, 11 key bits, per level, the last level
 *          SL_POOL2_KEY_BINS single keys; two levels settle a bracket of up to 2^23 values, three every bracket; sparse and heavily tied keys
 *          are settled like dense ones; a settled key set turns the remaining passes and steps into no-ops)
 * on one stream; state[SL_POOL_M / _MAXC / _STATUS / _MISS] as for sl_pool_*;
 * state[SL_POOL2_WHY] != 0 says the sample gave no usable estimate (the sweep then returns at once and the chain ends in a miss).
 * `workspace` (sl_pool2_workspace_bytes, 256-byte aligned) carries the sample list, the candidate list (up to 1/8 of the pixels) and
 * the per-workgroup partial sums from sl_pool2_sample to the last sl_pool2_hist.  Every step consumes all-reduced data only: the ranks
 * reach the same state without a broadcast.  Reference: macenko_stain_extractor.py:18-44, normalizer.py:36,45-47 on the concatenation. */
#define SL_POOL2_STATE_DOUBLES 256
#define SL_POOL2_TAIL_SLOTS 32
#define SL_POOL2_GRID_BINS 8192
#define SL_POOL2_WINDOW_BINS 2048 /* coarse bins per target of a window pass (mode 1) */
#define SL_POOL2_KEY_BINS 4096    /* single keys per target at the last level of a window pass: 4 targets x SL_POOL2_KEY_BINS = 2 x SL_POOL2_GRID_BINS */
/* a histogram buffer: 8 x SL_POOL2_TAIL_SLOTS tail words, then 2 x SL_POOL2_GRID_BINS bins (mode 0: two grids; mode 1: four targets of
 * SL_POOL2_KEY_BINS); every pass WRITES it whole */
#define SL_POOL2_HIST_WORDS (8 * SL_POOL2_TAIL_SLOTS + 2 * SL_POOL2_GRID_BINS)
#define SL_POOL2_WHY 33
SL_API size_t sl_pool2_workspace_bytes(int n, int h, int w, int sample_log2);
SL_API int sl_pool2_sample(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int sample_log2, void* workspace,
                    size_t workspace_bytes, double* moments16_out, void* stream);
SL_API int sl_pool2_begin(const double* moments16_reduced, const SlParams* params, int sample_log2, double* state, void* stream);
/* which: 0 the sample list (mode 0: a uniform grid of SL_POOL2_GRID_BINS bins per order statistic), 1 the candidate list (mode 1: per target a
 * window of bins of 2^sh consecutive binary32 values, position and sh from `state`); hist (SL_POOL2_HIST_WORDS uint64) is written whole */
SL_API int sl_pool2_hist(int which, int keyset, int mode, int n, int h, int w, const SlParams* params, int sample_log2, const double* state,
                  void* workspace, size_t workspace_bytes, unsigned long long* hist, void* stream);
SL_API int sl_pool2_bands(double* state, int keyset, const unsigned long long* hist_reduced, void* stream);
SL_API int sl_pool2_sweep(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int sample_log2, const double* state,
                   void* workspace, size_t workspace_bytes, double* totals16_out, void* stream);
SL_API int sl_pool2_exact(const double* totals16_reduced, double* state, void* stream);
SL_API int sl_pool2_step(double* state, int keyset, const unsigned long long* hist_reduced, void* stream);
/* the whole chain above on ONE process (no all-reduce between the steps), enqueued by one call; state as above */
#define SL_POOL2_LEVELS 3
SL_API int sl_pool2_local(const uint8_t* rgb, int n, int h, int w, const SlParams* params, int sample_log2, void* workspace,
                   size_t workspace_bytes, double* state, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STAINLIB_HIP_H */
