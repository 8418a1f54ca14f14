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

  * POOLED slide-level mode (``SlideNormalizer(..., mode="pooled")``): the slide statistics are
    exactly those the reference computes from the vertical concatenation of ALL tiles as one tall image --
    covariance over every tissue pixel of the slide, 1st/99th angular percentiles over those pixels, 99th
    percentile of each concentration over every pixel.  Sums and order statistics decompose over tiles and
    ranks: per-tile moment sums are all-reduced (10 doubles), and each exact order statistic of the binary32
    key is pinned by a 4-round radix select whose 256-bin histograms are all-reduced (two order statistics per
    sweep, 4 KiB per round).  All
    collectives are tiny and latency-bound.  Oracle: the reference restatement on the concatenated image.
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


# Test hook: run the collectives even in a ONE-rank process group, so that a single GPU exercises the RCCL path end to end
# (bench.py sets it under SL_BENCH_FORCE_DIST=1; tests/test_gpu_rccl.py).  Off: a one-rank job pays for no collective.
COLLECTIVES_AT_WORLD_1 = False


def _coll(world: int, group=None) -> bool:
    """Whether the collectives of a step run: more than one rank, or the test hook above with a live process group."""
    if world > 1:
        return True
    return bool(COLLECTIVES_AT_WORLD_1 and group is not False and dist.is_available() and dist.is_initialized())


def _world(group=None):
    if group is False:           # "this process only": no collective even when a process group exists
        return 0, 1
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
    if not _coll(world, group):
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


# ---- exact order statistics of a key that is spread over tiles and ranks ------------------------------------------
def ord_to_float(o: int) -> float:
    """Inverse of the order-preserving uint32 image of a binary32 value (include/stainlib_hip.h)."""
    import struct
    bits = (o & 0x7fffffff) if (o & 0x80000000) else (~o & 0xffffffff)
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def percentile_position(n: int, pct: float):
    """numpy.percentile(method='linear'): 0-based rank k and interpolation weight g between ranks k and k+1."""
    import math
    vi = min(max((pct / 100.0) * (n - 1), 0.0), float(n - 1))
    k = math.floor(vi)
    return int(k), vi - k


def np_lerp(a: float, b: float, t: float) -> float:
    d = b - a
    return b - d * (1.0 - t) if t >= 0.5 else a + d * t


def exact_rank_pairs(hist_fn, next_above_fn, ks, group=None, hist16_fn=None, fractions=None, stop_bits=32):
    """For each of TWO targets t: the keys of ranks ks[t] and min(ks[t]+1, N_t-1) (0-based, ascending) of the union of
    every rank's keys of that target, as ordered uint32: [(key_k, key_k1, N_t), ...].

    hist_fn(prefixes, prefix_bits) -> int64 (2, 256) tensor: per target THIS rank's histogram of the next 8 key bits
    among its keys whose top prefix_bits bits equal prefixes[t]; next_above_fn(keys) -> per target this rank's smallest
    key above keys[t] (0xffffffff if none).  Both targets advance in lockstep, so the all-reduced histogram rounds that
    pin one order statistic exactly pin both (one sweep over the tiles per round).  With hist16_fn(prefixes16) ->
    int64 (2, 65536) (the low 16 bits of the keys under a 16-bit prefix) the last two rounds are one sweep: 8 + 8 + 16
    bits.  The k+1-th key is read off the last round's histogram (its bins are keys); the extra next_above sweep only
    runs when the k-th key is the largest of its window and unique."""
    import numpy as np
    _, world = _world(group)
    prefix, below, in_bin, total, k = [0, 0], [0, 0], [0, 0], [None, None], [int(ks[0]), int(ks[1])]
    succ = [None, None]          # the next larger key inside the last round's window, if there is one
    rounds = [(0, 8), (8, 8), (16, 16)] if hist16_fn is not None else [(0, 8), (8, 8), (16, 8), (24, 8)]
    for bits, width in rounds:
        if bits >= stop_bits:        # an estimate is enough: the remaining low bits are set to the middle of their range
            rest = 32 - bits
            return [((prefix[t] << rest) | (1 << (rest - 1)), (prefix[t] << rest) | (1 << (rest - 1)), total[t]) for t in range(2)]
        h = hist16_fn(prefix) if width == 16 else hist_fn(prefix, bits)
        if _coll(world, group):
            dist.all_reduce(h, group=group)
        hc = h.detach().cpu().numpy().astype(np.int64)
        last = bits + width == 32
        for t in range(2):
            if total[t] is None:
                total[t] = int(hc[t].sum())
                if total[t] == 0:
                    raise ValueError("no pixel carries this key")
                if fractions is not None:          # the rank as a fraction of however many keys there turn out to be
                    k[t] = int(fractions[t] * (total[t] - 1))
                k[t] = min(max(k[t], 0), total[t] - 1)
            cum = np.cumsum(hc[t])
            b = int(np.searchsorted(cum, k[t] - below[t], side="right"))     # first bin with below + cum[b] > k
            if last:                 # the bins of the last round ARE keys: the successor is the next non-empty bin
                nz = np.nonzero(hc[t][b + 1:])[0]
                succ[t] = None if len(nz) == 0 else ((prefix[t] << width) | (b + 1 + int(nz[0])))
            below[t] += int(cum[b] - hc[t][b])
            in_bin[t] = int(hc[t][b])
            prefix[t] = (prefix[t] << width) | b
    need = [not (k[t] + 1 < below[t] + in_bin[t] or k[t] + 1 >= total[t]) for t in range(2)]
    nxt = list(prefix)
    for t in range(2):           # the sweep for the next larger key is only needed when the window holds none
        if need[t] and succ[t] is not None:
            nxt[t], need[t] = succ[t], False
    if any(need) and next_above_fn is not None:
        got = next_above_fn(prefix)
        if _coll(world, group):
            tt = torch.tensor(got, dtype=torch.int64, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MIN, group=group)
            got = [int(x) for x in tt.tolist()]
        nxt = [(got[t] if (need[t] and got[t] != 0xffffffff) else nxt[t]) for t in range(2)]
    return [(prefix[t], nxt[t], total[t]) for t in range(2)]


def window_rank_pairs(sample_hist_fn, window_fn, ks, totals, group=None):
    """The same result as exact_rank_pairs in ONE sweep over the tiles (plus four over a 1/64 pixel sample), or None.

    sample_hist_fn(prefixes, prefix_bits) -> (2, 256) int64: this rank's histogram over its pixel SAMPLE; the exact
    order statistic of the union of the samples at the same fraction estimates the key.  window_fn(lo) -> (2 * 65536 + 2,)
    int64: per target this rank's histogram of key - lo[t] inside [lo[t], lo[t] + 65536) and its count of keys below
    lo[t].  The window is centred on the estimate; if the wanted ranks k and k + 1 (totals[t] keys in all) do not both
    fall inside it -- the estimate was off by more than 32768 consecutive binary32 values -- the answer is None and the
    caller takes the radix rounds.  Everything is all-reduced, so all ranks decide alike."""
    import numpy as np
    _, world = _world(group)
    fr = [(int(ks[t]) / (totals[t] - 1)) if totals[t] > 1 else 0.0 for t in range(2)]
    try:
        est = exact_rank_pairs(sample_hist_fn, None, (0, 0), group, fractions=[min(max(f, 0.0), 1.0) for f in fr], stop_bits=24)
    except ValueError:
        return None
    lo = [min(max(est[t][0] - 32768, 0), 0xffffffff - 65535) for t in range(2)]
    buf = window_fn(lo)
    if _coll(world, group):
        dist.all_reduce(buf, group=group)
    b = buf.detach().cpu().numpy().astype(np.int64)
    out = []
    for t in range(2):
        hist, below, n = b[t * 65536:(t + 1) * 65536], int(b[2 * 65536 + t]), int(totals[t])
        k = min(max(int(ks[t]), 0), n - 1)
        k1 = min(k + 1, n - 1)
        inside = int(hist.sum())
        if not (below <= k and k1 < below + inside):
            return None
        cum = np.cumsum(hist)
        i0 = int(np.searchsorted(cum, k - below, side="right"))
        i1 = int(np.searchsorted(cum, k1 - below, side="right"))
        out.append((lo[t] + i0, lo[t] + i1, n))
    return out


class PooledSlideStatistics:
    """Stain matrix and 99th-percentile concentrations of the tall image made of every tile on every rank."""

    def __init__(self, group=None, luminosity_threshold=0.8, angular_percentile=99.0, lasso_lambda=0.01):
        self.group = group
        self.thr, self.pct, self.lam = luminosity_threshold, angular_percentile, lasso_lambda
        self.last_path = []          # per stage of the last call: "merged" (the one-sweep chain), "window" (a sweep per stage) or "radix" (the fallback rounds)
        self.last_miss = 0           # state[POOL_MISS] of the last device-driven chain read back
        self.last_why = 0            # state[POOL2_WHY] of the last one-sweep chain (why the sample gave no estimate)
        self.sample_log2 = None      # None: the sample density follows the slide's pixel count (sample_log2_for); 0...12: one 64-pixel sub-row in 2^s
                                     # (tests and experiments; the same on every rank -- the RESULT does not depend on it, only which route settles it)

    def enqueue(self, tiles_local: torch.Tensor, ws=None, n_tiles_total: Optional[int] = None) -> torch.Tensor:
        """DEVICE-DRIVEN: enqueue the whole computation (4 full sweeps, 6 sampled passes, the all-reduces between them and the
        single-workgroup decision steps) on the current stream and return the pool state tensor (device float64,
        _ffi.POOL_STATE_DOUBLES): state[POOL_M:POOL_M+6] / state[POOL_MAXC:+2] are the slide's stain matrix and maxC once
        state[POOL_STATUS] == 0 and state[POOL_MISS] == 0 (``finish`` checks them with one read-back).  Every rank reaches the same
        state: each step consumes all-reduced data only.
        With ``n_tiles_total`` (the slide's tile count over ALL ranks -- every rank must pass the same value, or none of them may) nothing
        is read back and no extra collective runs; on one rank the chain is then graph-capturable.  WITHOUT it, on more than one rank,
        the sample density is agreed on with one small MAX all-reduce of the shard sizes whose result IS read back (a host
        synchronisation per call): pass the total where the call sits on a latency-critical path.  Ranks that disagree on whether they
        pass it issue different collective sequences and hang -- it is part of the call's collective contract."""
        import math
        from . import engine, _ffi
        params = engine.make_params(luminosity_threshold=self.thr, angular_percentile=self.pct, lasso_lambda=self.lam)
        _, world = _world(self.group)
        n_local, h, w, _ = tiles_local.shape
        dev = tiles_local.device
        # 10 moment sums + this rank's pixel count (torch.full: a fill kernel -- a scalar copied from the host could not be captured)
        mom = torch.cat([engine.tile_moments(tiles_local, params=params, ws=ws).sum(dim=0),
                         torch.full((1,), float(n_local * h * w), dtype=torch.float64, device=dev)])
        if _coll(world, self.group):
            dist.all_reduce(mom, group=self.group)
        state = engine.pool_begin(mom, params=params)
        # the sample: ~4 M pixels of the slide or more.  The density must be the SAME on every rank (the sampled histograms are
        # all-reduced), so it is derived from a rank-independent tile count: the caller's n_tiles_total, else the largest shard
        # (shard_range gives ceil(n / world) to some rank) agreed on with one tiny MAX all-reduce -- NOT from this rank's own
        # n_local, which differs by one tile across ranks on uneven shards and can sit on the other side of a power of two
        # (3 vs 4 tiles of 1024^2 on two ranks; round-3 advisor finding).  Results never depended on it, the window hit rate did.
        if n_tiles_total is not None:
            n_pixels = int(n_tiles_total) * h * w
        elif _coll(world, self.group) and world > 1:
            nl = torch.tensor([n_local], dtype=torch.int64, device=dev if dist.get_backend(self.group) == "nccl" else "cpu")
            dist.all_reduce(nl, op=dist.ReduceOp.MAX, group=self.group)
            n_pixels = world * int(nl.item()) * h * w
        else:
            n_pixels = world * n_local * h * w
        slog = min(6, max(0, int(math.floor(math.log2(max(n_pixels, 1) / 4.0e6))))) if n_pixels > 4.0e6 else 0
        hists = torch.zeros((2, 3, 2, 256), dtype=torch.int64, device=dev)
        wins = torch.zeros((2, 2 * 65536 + 2), dtype=torch.int64, device=dev)
        for si, keyset in enumerate((_ffi.KEYSET_ANGLE, _ffi.KEYSET_CONC)):
            for rnd in range(3):
                hb = engine.pool_histogram(tiles_local, keyset, state, rnd, slog, hists[si, rnd], params=params)
                if _coll(world, self.group):
                    dist.all_reduce(hb, group=self.group)
                engine.pool_pick(state, keyset, rnd, hb)
            wb = engine.pool_window(tiles_local, keyset, state, wins[si], params=params)
            if _coll(world, self.group):
                dist.all_reduce(wb, group=self.group)
            engine.pool_resolve(state, keyset, wb, params=params)
        return state

    MERGED_LEVELS = 3            # window levels enqueued per key set by the one-sweep chain (11 key bits each; SL_POOL2_LEVELS)
    one_call = True              # on one process the chain is enqueued by sl_pool2_local (False: step by step, as on several ranks)

    @staticmethod
    def sample_log2_for(n_pixels: int) -> int:
        """Density of the merged chain's pixel sample (one 64-pixel sub-row in 2**result): everything up to 4 Mpx, then the sample grows
        like the slide's size to the power 2/3 -- the candidate lists shrink like 1/sqrt(sample) while the sample passes grow with it."""
        import math
        if n_pixels <= (1 << 22):
            return 0
        return int(min(12, max(0, math.floor((math.log2(n_pixels) - 11.0) / 3.0))))

    def _agreed_pixels(self, tiles_local, n_tiles_total, world):
        """The slide's pixel count as every rank computes it alike (see ``enqueue``)."""
        n_local, h, w, _ = tiles_local.shape
        if n_tiles_total is not None:
            return int(n_tiles_total) * h * w
        if _coll(world, self.group) and world > 1:
            nl = torch.tensor([n_local], dtype=torch.int64, device=tiles_local.device if dist.get_backend(self.group) == "nccl" else "cpu")
            dist.all_reduce(nl, op=dist.ReduceOp.MAX, group=self.group)
            return world * int(nl.item()) * h * w
        return world * n_local * h * w

    def enqueue_merged(self, tiles_local: torch.Tensor, ws=None, n_tiles_total: Optional[int] = None) -> torch.Tensor:
        """DEVICE-DRIVEN, ONE full sweep (round 6; csrc/slide_merged.hip): a stratified pixel sample of the whole slide gives an estimate
        of the eigenvectors, the angular brackets and the stain matrix; the one sweep over the tiles computes the exact moment sums AND
        appends every pixel that is not proven plain under that estimate to a candidate list; the exact order statistics are then those
        of the candidates (four passes over the list).  Eight small all-reduces; returns the pool state (device float64,
        _ffi.POOL2_STATE_DOUBLES) with the layout of ``enqueue``'s in its first ten entries.  state[POOL_MISS] != 0 at the end: a check
        of the estimate failed (or a list overflowed) -- results never depend on the sample, the caller takes ``enqueue`` then.
        n_tiles_total: as for ``enqueue`` (the same on every rank, or on none).  ws: None, or a dict the caller keeps between calls (the
        chain's workspace -- sample list, candidate list of up to 1/8 of the pixels -- is then allocated once per shape)."""
        from . import engine, _ffi
        params = engine.make_params(luminosity_threshold=self.thr, angular_percentile=self.pct, lasso_lambda=self.lam)
        _, world = _world(self.group)
        n_local, h, w, _ = tiles_local.shape
        dev = tiles_local.device
        coll = _coll(world, self.group)
        slog = self.sample_log2_for(self._agreed_pixels(tiles_local, n_tiles_total, world)) if self.sample_log2 is None else int(self.sample_log2)
        if ws is None or ws.get("key") != (n_local, h, w, slog, dev):
            buf = engine.pool2_workspace(n_local, h, w, slog, dev)
            if ws is not None:               # a caller's cache (a dict): the buffer is reused by its next call with this shape
                ws.clear()
                ws.update(key=(n_local, h, w, slog, dev), buf=buf)
            ws = {"buf": buf}
        ws = ws["buf"]
        if not coll and self.one_call:       # one process: the chain enqueued by ONE call into the library (the same kernels in the same order)
            self._merged_ws = ws
            return engine.pool2_local(tiles_local, slog, ws, params=params)
        shape = (n_local, h, w)
        hists = torch.empty((2 + 2 * self.MERGED_LEVELS, _ffi.POOL2_HIST_WORDS), dtype=torch.int64, device=dev)   # every pass writes its buffer whole

        def reduced(t):
            if coll:
                dist.all_reduce(t, group=self.group)
            return t
        mom = reduced(engine.pool2_sample(tiles_local, slog, ws, params=params))
        state = engine.pool2_begin(mom, slog, params=params)
        for i, keyset in enumerate((_ffi.KEYSET_ANGLE, _ffi.KEYSET_CONC)):
            engine.pool2_bands(state, keyset, reduced(engine.pool2_hist(0, keyset, 0, shape, slog, state, ws, hists[i], params=params)))
        engine.pool2_exact(reduced(engine.pool2_sweep(tiles_local, slog, state, ws, params=params)), state)
        for i, keyset in enumerate((_ffi.KEYSET_ANGLE, _ffi.KEYSET_CONC)):
            for level in range(self.MERGED_LEVELS):      # (a settled key set turns its remaining passes and steps into no-ops)
                h = hists[2 + self.MERGED_LEVELS * i + level]
                engine.pool2_step(state, keyset, reduced(engine.pool2_hist(1, keyset, 1, shape, slog, state, ws, h, params=params)))
        self._merged_ws = ws             # (kept until the next call: the chain's kernels are still queued when this returns)
        return state

    def finish(self, state: torch.Tensor):
        """The one read-back of the device-driven path: (M, maxC) as numpy, or None when a window missed (the caller then runs the
        host-driven rounds).  Raises like the reference on an empty tissue mask."""
        import numpy as np
        from . import _ffi
        from .utils.excepts import TissueMaskException
        s = state.cpu().numpy()
        status, miss = int(s[_ffi.POOL_STATUS]), int(s[_ffi.POOL_MISS])
        if status == _ffi.TILE_EMPTY_MASK:
            raise TissueMaskException("Empty tissue mask computed")
        self.last_miss = miss
        if len(s) > _ffi.POOL2_WHY:
            self.last_why = int(s[_ffi.POOL2_WHY])
        if status != 0 or miss != 0:
            return None
        self.last_path = ["merged", "merged"] if len(s) == _ffi.POOL2_STATE_DOUBLES else ["window", "window"]
        return s[_ffi.POOL_M:_ffi.POOL_M + 6].reshape(2, 3).copy(), s[_ffi.POOL_MAXC:_ffi.POOL_MAXC + 2].copy()

    def __call__(self, tiles_local: torch.Tensor, device_driven: bool = True, n_tiles_total: Optional[int] = None, merged: bool = True):
        """(M, maxC) of the slide.  n_tiles_total: see ``enqueue`` (the same on every rank, or on none).  The one-sweep chain first
        (``merged``), the three-sweep chain if one of its checks fails, the host-driven radix rounds if a window misses: the same
        numbers whichever route settles them (up to the order of the moment sums)."""
        if device_driven:
            if merged:
                got = self.finish(self.enqueue_merged(tiles_local, n_tiles_total=n_tiles_total))
                if got is not None:
                    return got
            got = self.finish(self.enqueue(tiles_local, n_tiles_total=n_tiles_total))
            if got is not None:
                return got
        return self.host_driven(tiles_local)

    def host_driven(self, tiles_local: torch.Tensor):
        """The same statistics with the decisions on the host (a read-back per step): the radix fallback lives here."""
        import math
        import numpy as np
        from . import engine, _ffi
        from .utils.excepts import TissueMaskException
        params = engine.make_params(luminosity_threshold=self.thr, angular_percentile=self.pct, lasso_lambda=self.lam)
        self.last_path = []
        _, world = _world(self.group)
        n_local, h, w, _ = tiles_local.shape
        # ---- covariance of the optical density over every tissue pixel (macenko_stain_extractor.py:18-27)
        mom = engine.tile_moments(tiles_local, params=params).sum(dim=0)
        npx = torch.tensor([float(n_local * h * w)], dtype=torch.float64, device=mom.device)
        if _coll(world, self.group):
            dist.all_reduce(mom, group=self.group)
            dist.all_reduce(npx, group=self.group)
        m = mom.cpu().numpy()
        T, n_pixels = int(round(m[0])), int(round(float(npx.item())))
        # the sample: ~4 M pixels of the slide or more (one row in 2**slog), everything for small slides
        slog = min(6, max(0, int(math.floor(math.log2(max(n_pixels, 1) / 4.0e6))))) if n_pixels > 4.0e6 else 0
        if T < 1:
            raise TissueMaskException("Empty tissue mask computed")
        mean = m[1:4] / T
        S2 = np.array([[m[4], m[5], m[6]], [m[5], m[7], m[8]], [m[6], m[8], m[9]]])
        cov = (S2 - T * np.outer(mean, mean)) / (T - 1.0)
        _, V = np.linalg.eigh(cov)
        V = V[:, [2, 1]].copy()
        for i in range(2):
            if V[0, i] < 0:
                V[:, i] *= -1.0
        Vf = V.astype(np.float32).astype(np.float64)                # the keys are evaluated in binary32
        # ---- exact angular percentiles over those pixels (:29-34): both in the same four sweeps
        def pairs(keyset, basis, ks, totals):
            # one sweep, with the window centred on an estimate from a 1/64 pixel sample; the radix rounds if it missed
            res = window_rank_pairs(lambda pre, bits: engine.slide_key_histogram_sampled(tiles_local, keyset, basis, pre, bits, slog, params=params),
                                    lambda lo: engine.slide_key_window(tiles_local, keyset, basis, lo, params=params), ks, totals, self.group)
            self.last_path.append("window" if res is not None else "radix")
            if res is not None:
                return [(ord_to_float(a), ord_to_float(b)) for a, b, _ in res]
            res = exact_rank_pairs(lambda pre, bits: engine.slide_key_histogram(tiles_local, keyset, basis, pre, bits, params=params),
                                   lambda o: engine.slide_key_next_above(tiles_local, keyset, basis, o, params=params), ks, self.group,
                                   hist16_fn=lambda pre: engine.slide_key_histogram16(tiles_local, keyset, basis, pre, params=params))
            return [(ord_to_float(a), ord_to_float(b)) for a, b, _ in res]

        def angle_of_pseudo(p):
            if abs(p) <= 1.0:
                return math.atan2(p, 1.0 - abs(p))
            pp = 2.0 - p if p > 0 else -2.0 - p
            return math.atan2(pp, -(1.0 - abs(pp)))
        (k_lo, g_lo), (k_hi, g_hi) = percentile_position(T, 100.0 - self.pct), percentile_position(T, self.pct)
        (xa0, xb0), (xa1, xb1) = pairs(_ffi.KEYSET_ANGLE, Vf.reshape(6), (k_lo, k_hi), (T, T))
        phis = [np_lerp(angle_of_pseudo(xa0), angle_of_pseudo(xb0), g_lo), np_lerp(angle_of_pseudo(xa1), angle_of_pseudo(xb1), g_hi)]
        v1 = V @ np.array([math.cos(phis[0]), math.sin(phis[0])])          # :36-37
        v2 = V @ np.array([math.cos(phis[1]), math.sin(phis[1])])
        M = np.array([v1, v2]) if v1[0] > v2[0] else np.array([v2, v1])     # :40-43
        M = M / np.linalg.norm(M, axis=1, keepdims=True)                    # :44
        # ---- 99th percentile of each concentration over every pixel (normalizer.py:36,47): both columns per sweep
        k, g = percentile_position(n_pixels, 99.0)
        (ca0, cb0), (ca1, cb1) = pairs(_ffi.KEYSET_CONC, M.reshape(6), (k, k), (n_pixels, n_pixels))
        maxC = [np_lerp(float(ca0), float(cb0), g), np_lerp(float(ca1), float(cb1), g)]
        return M, np.asarray(maxC, dtype=np.float64)


class SlideNormalizer:
    """Slide-level Macenko/Vahadane normalisation over a sharded set of tiles (see module docstring).

    mode="median" (default): per-tile fits, all-gather, element-wise median.  mode="pooled": the exact statistics
    of the concatenated slide (Macenko only).

    graph=True (pooled mode on ONE process: no collective sits between the steps): the one-sweep chain and the apply pass behind it --
    some fifty launches -- are captured into a HIP graph the first time a (tiles buffer, out buffer) pair is seen and REPLAYED on every
    later call with the same buffers (a pipeline that refills fixed staging buffers): 512 tiles 1.92 -> 1.84 ms, 128 tiles 0.79 -> 0.77.
    The read-back after the replay is the same one; a replay that ends in a miss falls back exactly like the eager chain.  `out` is
    allocated once and reused when the caller passes none."""

    def __init__(self, normalizer, group=None, mode="median", merged=True, graph=False):
        if mode not in ("median", "pooled"):
            raise ValueError("mode must be 'median' or 'pooled'")
        self.normalizer = normalizer          # a fitted stainlib_amd ExtractiveStainNormalizer
        self.group = group
        self.mode = mode
        self.merged = merged                  # pooled mode: the one-sweep chain first (PooledSlideStatistics.enqueue_merged)
        self.graph = graph
        self._pool2_ws = {}                   # its workspace, kept between calls
        self._graphed = None                  # (key, engine.Graphed, pinned tensors) of the last captured chain

    def _targets(self, device):
        """The normalizer's 8 target doubles for the apply pass: its cached device tensors where it has them (uploaded once per fit; as
        numpy arguments they are two small pageable copies per call), else the public attributes as they are."""
        cached = getattr(self.normalizer, "_target_on", None)
        if cached is not None:
            return cached(device)
        return self.normalizer.stain_matrix_target, self.normalizer.maxC_target.reshape(2)

    def transform_shard(self, tiles_local: torch.Tensor, out: Optional[torch.Tensor] = None, n_tiles_total: Optional[int] = None):
        """tiles_local: this rank's (n_local,H,W,3) uint8 device tensor.  Returns (out, M_slide, maxC_slide, status_local).
        n_tiles_total (pooled mode, optional): the slide's tile count over all ranks; saves the one tiny all-reduce that otherwise
        agrees on the sample density.  On failure (TissueMaskException) `out` holds a copy of the input tiles."""
        from . import engine
        if self.mode == "pooled":
            from . import _ffi
            stats = PooledSlideStatistics(self.group)
            dev = tiles_local.device
            n = tiles_local.shape[0]
            # device-driven: the statistics AND the apply pass are enqueued before anything is read back; the one read-back
            # afterwards only confirms that both windows caught their ranks (else: the host-driven rounds, and the pass again).
            # When the chain ends in an unusable state (a window miss, an empty tissue mask, a degenerate covariance) its last step
            # leaves NaN in (M, maxC) and the enqueued apply pass COPIES the tiles through (k_apply's rule for unusable statistics):
            # `out` then holds the input, never exp(NaN) bytes, until the host-driven rounds below rewrite it -- or, on an empty
            # mask, when TissueMaskException leaves this function.
            Mt, mct = self._targets(dev)
            got = None
            chains = (stats.enqueue_merged, stats.enqueue) if self.merged else (stats.enqueue,)
            _, world = _world(self.group)
            if self.graph and self.merged and stats.one_call and not _coll(world, self.group):
                # the captured chain: same buffers as last time -> replay; else capture (one warm-up run, one run under capture)
                import numpy as np
                tgt_key = (np.asarray(self.normalizer.stain_matrix_target, dtype=np.float64).tobytes(),
                           np.asarray(self.normalizer.maxC_target, dtype=np.float64).tobytes())          # (host values: no device read-back for the key)
                if out is None:
                    out = self._graphed[2]["out"] if (self._graphed and self._graphed[2]["out"].shape == tiles_local.shape
                                                      and self._graphed[2]["out"].device == dev) else torch.empty_like(tiles_local)
                key = (tiles_local.data_ptr(), out.data_ptr(), tuple(tiles_local.shape), n_tiles_total, str(dev), stats.thr, stats.pct, stats.lam, tgt_key)
                if self._graphed is None or self._graphed[0] != key:
                    keep = {"out": out, "tiles": tiles_local,
                            "Mt": torch.as_tensor(Mt, dtype=torch.float64, device=dev).reshape(2, 3).clone(),
                            "mct": torch.as_tensor(mct, dtype=torch.float64, device=dev).reshape(2).clone()}

                    def captured():
                        st_ = stats.enqueue_merged(tiles_local, n_tiles_total=n_tiles_total, ws=self._pool2_ws)
                        M_ = st_[_ffi.POOL_M:_ffi.POOL_M + 6].reshape(2, 3)
                        mc_ = st_[_ffi.POOL_MAXC:_ffi.POOL_MAXC + 2]
                        engine.normalize_apply(tiles_local, M_.expand(n, 2, 3).contiguous(), mc_.expand(n, 2).contiguous(), keep["Mt"], keep["mct"], out=out)
                        return st_, M_, mc_
                    self._graphed = None                       # (the old graph goes before its buffers do)
                    self._graphed = (key, engine.Graphed(captured), keep)
                state, M_s, maxC_s = self._graphed[1].replay()
                got = stats.finish(state)
                chains = chains[1:]                            # a miss: the three-sweep chain, eagerly
            for chain in (chains if got is None else ()):
                def run(chain=chain):
                    st_ = chain(tiles_local, n_tiles_total=n_tiles_total, ws=self._pool2_ws if chain == stats.enqueue_merged else None)
                    M_ = st_[_ffi.POOL_M:_ffi.POOL_M + 6].reshape(2, 3)
                    mc_ = st_[_ffi.POOL_MAXC:_ffi.POOL_MAXC + 2]
                    o_ = engine.normalize_apply(tiles_local, M_.expand(n, 2, 3).contiguous(), mc_.expand(n, 2).contiguous(), Mt, mct, out=out)
                    return st_, M_, mc_, o_
                state, M_s, maxC_s, out = run()
                got = stats.finish(state)
                if got is not None:
                    break
            if got is None:
                M_np, maxC_np = stats.host_driven(tiles_local)
                M_s = torch.as_tensor(M_np, dtype=torch.float64, device=dev)
                maxC_s = torch.as_tensor(maxC_np, dtype=torch.float64, device=dev)
                out = engine.normalize_apply(tiles_local, M_s.expand(n, 2, 3).contiguous(), maxC_s.expand(n, 2).contiguous(), Mt, mct, out=out)
            else:
                M_s, maxC_s = M_s.clone(), maxC_s.clone()
            self.last_path = stats.last_path             # per stage: "merged" (one sweep for both), "window" (one each) or "radix"
            self.last_miss, self.last_why = stats.last_miss, stats.last_why
            return out, M_s, maxC_s, torch.zeros((n,), dtype=torch.int32, device=dev)
        M, maxC, status = self.normalizer.fit_batch_targets(tiles_local)
        M_all, maxC_all, st_all = gather_tile_stats(M, maxC, status, self.group)
        M_s, maxC_s = slide_statistics(M_all, maxC_all, st_all)
        n = tiles_local.shape[0]
        Mt, mct = self._targets(tiles_local.device)
        out = engine.normalize_apply(tiles_local, M_s.expand(n, 2, 3).contiguous(), maxC_s.expand(n, 2).contiguous(), Mt, mct, out=out)
        return out, M_s, maxC_s, status
