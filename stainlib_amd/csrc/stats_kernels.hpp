// stats_kernels.hpp -- per-tile statistics of the Macenko path and the fused transform (gfx950).
//
// Reference chain (macenko_stain_extractor.py:7-44, normalizer.py:34-36,45-50):
//   mask -> OD -> cov(3x3) -> eigh -> project -> arctan2 -> percentile(1,99) -> M
//   -> lasso concentrations of ALL pixels -> percentile(99) per stain -> rescale -> 255 exp(-C M_t)
//
// Four dependent streaming sweeps over the uint8 tile, nothing per-pixel stored in between:
//   sweep 1  moments   tissue test, 9 moment sums (binary32 bursts of one trip per lane, binary64 totals), one
//                      stratified-random sample pixel per `stride` pixels
//   finish 1           cov -> Jacobi eigh -> V ; sample -> brackets [lo,hi] that contain the 1st/99th
//                      angular order statistics with overwhelming probability
//   sweep 2  select    exact counts below / inside each bracket, bracket members ("candidates", ~1 %
//                      of the pixels) collected
//   finish 2           EXACT order statistics k, k+1 among the candidates -> numpy-style linear
//                      interpolation -> stain matrix M ; sample -> brackets for the 99th percentile of
//                      both concentration columns
//   sweep 3  select    same skeleton on the lasso concentrations of all pixels
//   finish 3           exact 99th percentiles -> maxC, status
//   sweep 4  apply     (transform only) OD -> lasso -> rescale -> exp -> truncate -> store
//
// If a bracket misses (probability ~1e-9 per tile) or overflows (heavy ties) the finish step falls
// back to an exact radix selection over the whole tile: results never depend on the sampling, only
// the speed does.  Order statistics are exact on the binary32 keys the sweeps compute
// (pseudo-angle, concentrations); interpolation, trigonometry, eigen-decomposition and moment sums
// are binary64.
//
// Two schedules share every device function below:
//   * k_macenko_fused : persistent kernel, ONE 1024-thread workgroup owns a tile through all phases
//     (phase hand-offs are __syncthreads, the sample lives in LDS, no launch boundaries, no
//     inter-workgroup traffic).  Used for batches large enough to fill the chip.
//   * k_moments / k_finish_* / k_select : one launch per phase with each tile split over several
//     workgroups.  Used for small batches, where a tile per workgroup would leave most CUs idle.
#pragma once
#include "stats_common.hpp"        // constants, TileState / StatsArgs, table views, sample, keys
#include "stats_selection.hpp"     // workgroup-level exact selection
#include "stats_linalg.hpp"        // eigh, stain matrix, merged box
#include "stats_sweeps.hpp"        // moments_sweep_b, select_sweep
#include "stats_finish.hpp"        // refine / census / pick, RawSink, sample brackets
#include "stats_cube.hpp"          // colour-cube pre-filter of the merged sweep (mask build + sweep)
#include "stats_dict.hpp"          // Vahadane dictionary
#include "stats_phase_kernels.hpp" // one launch per phase
#include "stats_fused.hpp"         // fused_finish1/2, k_finish1m/2m, k_fused
