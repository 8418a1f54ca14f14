"""Multi-GPU sharding of the stain-normalization path: one process per GPU, torch.distributed.

Tiles are independent in the reference (ExtractiveStainNormalizer.transform re-estimates the stain
matrix and the 99th-percentile concentrations from each tile, normalizer.py:45-47), so the per-tile
mode shards contiguous tile ranges over ranks and needs NO data-path collective.

The only exchange is small and optional:
  * ``gather_tile_stats``  all-gather of the per-tile (M 2x3, maxC 2, status) = 9 numbers/tile, for QC
    and for
  * slide-level mode (BASELINE.json configs[4]): every tile of a slide is normalised with ONE stain
    matrix / ONE pair of 99th-percentile concentrations -- the per-slide median of the per-tile
    estimates over valid tiles.  Each rank fits its own tiles, the 9 numbers per tile are all-gathered
    (RCCL over xGMI on GPUs, gloo in the CPU tests; latency-bound: 36 B/tile), every rank reduces the
    same gathered table to the same slide statistics, and the apply pass runs locally.
This is an extension (the reference has no notion of a slide); its check is the same recipe run on one process.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of the n tiles owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (rank * n) // world, ((rank + 1) * n) // world


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def gather_tile_stats(M: torch.Tensor, maxC: torch.Tensor, status: torch.Tensor, group=None):
    """All ranks' per-tile statistics in global tile order: (M_all (N,2,3), maxC_all (N,2), status_all (N,)).

    Shards may have different sizes; they are padded to the largest for the all-gather and trimmed after."""
    rank, world = _world(group)
    n_local = M.shape[0]
    packed = torch.cat([M.reshape(n_local, 6).double(), maxC.reshape(n_local, 2).double(),
                        status.reshape(n_local, 1).double()], dim=1)
    if world == 1:
        return M.reshape(n_local, 2, 3), maxC.reshape(n_local, 2), status.reshape(n_local)
    counts = torch.zeros(world, dtype=torch.int64, device=packed.device)
    counts[rank] = n_local
    dist.all_reduce(counts, group=group)
    n_max = int(counts.max())
    pad = torch.zeros((n_max, 9), dtype=torch.float64, device=packed.device)
    pad[:n_local] = packed
    bucket = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bucket, pad, group=group)
    allp = torch.cat([bucket[r][: int(counts[r])] for r in range(world)], dim=0)
    return allp[:, :6].reshape(-1, 2, 3), allp[:, 6:8], allp[:, 8].to(torch.int32)


def slide_statistics(M_all: torch.Tensor, maxC_all: torch.Tensor, status_all: torch.Tensor):
    """Per-slide stain matrix (2,3; unit-norm rows) and maxC (2,): element-wise median over the tiles whose
    fit succeeded.  Deterministic, so every rank derives identical values from the same gathered table."""
    ok = status_all == 0
    if int(ok.sum()) == 0:
        raise ValueError("no tile of the slide has a valid stain estimate")
    M = torch.quantile(M_all[ok].double(), 0.5, dim=0)            # numpy-style median (mean of the middle two)
    M = M / M.norm(dim=1, keepdim=True)
    maxC = torch.quantile(maxC_all[ok].double(), 0.5, dim=0)
    return M, maxC


class SlideNormalizer:
    """Slide-level Macenko/Vahadane normalisation over a sharded set of tiles (see module docstring)."""

    def __init__(self, normalizer, group=None):
        self.normalizer = normalizer          # a fitted stainlib_amd ExtractiveStainNormalizer
        self.group = group

    def transform_shard(self, tiles_local: torch.Tensor, out: Optional[torch.Tensor] = None):
        """tiles_local: this rank's (n_local,H,W,3) uint8 device tensor.  Returns (out, M_slide, maxC_slide, status_local)."""
        from . import engine
        M, maxC, status = self.normalizer.fit_batch_targets(tiles_local)
        M_all, maxC_all, st_all = gather_tile_stats(M, maxC, status, self.group)
        M_s, maxC_s = slide_statistics(M_all, maxC_all, st_all)
        n = tiles_local.shape[0]
        out = engine.normalize_apply(tiles_local, M_s.expand(n, 2, 3).contiguous(), maxC_s.expand(n, 2).contiguous(),
                                     self.normalizer.stain_matrix_target, self.normalizer.maxC_target.reshape(2), out=out)
        return out, M_s, maxC_s, status
