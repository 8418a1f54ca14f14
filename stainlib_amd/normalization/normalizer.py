"""ExtractiveStainNormalizer (stainlib/normalization/normalizer.py:16-50) on the HIP engine.

Drop-in contract: ``fit(target)`` / ``transform(I)`` take and return numpy uint8 HWC images and set
``stain_matrix_target`` (2,3), ``maxC_target`` (1,2) and ``target_concentrations`` (P,2) exactly
like the reference.  Extension: ``transform_batch`` takes an (N,H,W,3) uint8 device tensor and
keeps everything in HBM -- that is the path bench.py measures.
"""
from __future__ import annotations

import warnings

import numpy as np

from ..utils.stain_utils import _UINT8_MSG, _to_device, get_concentrations, is_uint8_image, raise_for_status

_METHODS = ("macenko", "vahadane")
# A single image of this many pixels or more is not handled as ONE tile (whose finish steps run on one workgroup and
# grow with the tile) but as the vertical concatenation of its row bands: the pooled slide statistics are, by
# definition, the reference's statistics of that concatenation, i.e. of the image itself -- computed with chip-wide
# sweeps only (8192 x 8192: 1.9 instead of 4.9 ms).  The two paths agree to the rounding of the binary32 burst sums of the
# moments (stain matrix ~1e-7, maxC ~1e-6 relative: tests/test_gpu_macenko.py::test_large_single_image_goes_through_the_pooled_statistics),
# so results step by that much at this size; both are within 2e-6 of the reference.  The pooled path is
# host-driven (~1.3 ms whatever the size), so it only pays from the measured crossover on (tools/big_image.py:
# 4096^2 1.27 vs 1.58 ms, 6144^2 3.4 vs 1.9 ms).
BIG_IMAGE_PIXELS = 5 << 22        # ~21 Mpx


def _row_bands(dev_img):
    """(1, H, W, 3) device image -> (H/r, r, W, 3) view, r = the largest divisor of H with r * W <= 2**20 (or 1)."""
    _, H, W, _ = dev_img.shape
    r = 1
    for cand in range(1, H + 1):
        if H % cand == 0 and cand * W <= (1 << 20):
            r = cand
    return dev_img.view(H // r, r, W, 3)


class ExtractiveStainNormalizer(object):
    def __init__(self, method):
        name = method.lower()
        if name not in _METHODS:
            raise Exception('Method not recognized.')                      # normalizer.py:25
        self.method = name
        if name == "macenko":
            from ..extraction.macenko_stain_extractor import MacenkoStainExtractor
            self.extractor = MacenkoStainExtractor
        else:
            from ..extraction.vahadane_stain_extractor import VahadaneStainExtractor
            self.extractor = VahadaneStainExtractor
        self._target = None
        self._target_concentrations = None

    # -- engine entry points of this method ------------------------------------------------------
    def _fit_tiles(self, tiles, ws=None):
        from .. import engine
        if self.method == "macenko":
            return engine.macenko_fit(tiles, ws=ws)
        M, maxC, status, _ = engine.vahadane_fit(tiles, ws=ws)
        return M, maxC, status

    def _target_on(self, device):
        """The 8 target doubles as device tensors, uploaded once per fit: as numpy arguments they are a small pageable
        copy per call, which queues behind any large upload in flight (see pipeline.py)."""
        import torch
        key = (str(device), np.asarray(self.stain_matrix_target, dtype=np.float64).tobytes(),
               np.asarray(self.maxC_target, dtype=np.float64).tobytes())        # by value: the attributes are public
        if getattr(self, "_target_dev_key", None) != key:
            self._target_dev = (torch.as_tensor(np.asarray(self.stain_matrix_target, dtype=np.float64), device=device).reshape(2, 3).contiguous(),
                                torch.as_tensor(np.asarray(self.maxC_target, dtype=np.float64), device=device).reshape(2).contiguous())
            self._target_dev_key = key
        return self._target_dev

    def _transform_tiles(self, tiles, out=None, ws=None):
        from .. import engine
        fn = engine.macenko_transform if self.method == "macenko" else engine.vahadane_transform
        M_t, c_t = self._target_on(tiles.device)
        return fn(tiles, M_t, c_t, out=out, ws=ws)

    def _big_image_statistics(self, dev):
        """(M (2,3), maxC (2,)) of one large image through the pooled statistics, or None when that path does not apply
        (Vahadane; degenerate images, which the per-tile path reports through its status codes)."""
        if self.method != "macenko" or dev.shape[1] * dev.shape[2] < BIG_IMAGE_PIXELS:
            return None
        from ..distributed import PooledSlideStatistics
        from ..utils.excepts import TissueMaskException
        try:
            M, maxC = PooledSlideStatistics(group=False)(_row_bands(dev))
        except (TissueMaskException, ValueError):
            return None
        if not (np.isfinite(M).all() and np.isfinite(maxC).all() and (maxC > 0).all()):
            return None
        G = M @ M.T
        if not (G[0, 0] * G[1, 1] - G[0, 1] ** 2 > 1e-8 * G[0, 0] * G[1, 1]):     # parallel stain vectors: the per-tile path reports it (status 2)
            return None
        return M, maxC

    # -- reference API -----------------------------------------------------------------------------
    def fit(self, target):
        """Fit to a target image (RGB uint8), normalizer.py:27-36."""
        assert is_uint8_image(target), _UINT8_MSG
        dev = _to_device(target)
        big = self._big_image_statistics(dev)
        if big is not None:
            self.stain_matrix_target, self.maxC_target = big[0], big[1].reshape((1, 2))
            self._target = target.copy()
            self._target_concentrations = None
            return
        M, maxC, status = self._fit_tiles(dev)
        raise_for_status(int(status[0]))
        self.stain_matrix_target = M[0].cpu().numpy()
        self.maxC_target = maxC[0].cpu().numpy().reshape((1, 2))
        self._target = target.copy()            # (the reference computes target_concentrations eagerly: later edits of the caller's array must not leak in)
        self._target_concentrations = None

    @property
    def target_concentrations(self):
        """(P, 2) concentrations of the target (normalizer.py:35); materialised on first use only --
        the reference stores this array but nothing reads it."""
        if self._target_concentrations is None and self._target is not None:
            self._target_concentrations = get_concentrations(self._target, self.stain_matrix_target)
        return self._target_concentrations

    def transform(self, I):
        """Transform an image (RGB uint8) to the fitted target's stain appearance, normalizer.py:39-50."""
        assert is_uint8_image(I), _UINT8_MSG
        dev = _to_device(I)
        big = self._big_image_statistics(dev)
        if big is not None:
            from .. import engine
            out = engine.normalize_apply(dev, big[0][None], big[1][None], self.stain_matrix_target, self.maxC_target.reshape(2))
            return out[0].cpu().numpy()
        out, _, _, status = self._transform_tiles(dev)
        st = int(status[0])
        raise_for_status(st)
        if st != 0:
            warnings.warn("99th-percentile concentration of the source is zero; the reference divides by it",
                          RuntimeWarning)
        return out[0].cpu().numpy()

    # -- batched extension -------------------------------------------------------------------------
    def fit_batch_targets(self, tiles, ws=None):
        """Per-tile (M, maxC, status) device tensors for a batch of candidate targets."""
        return self._fit_tiles(tiles, ws=ws)

    def transform_batch(self, tiles, out=None, ws=None):
        """(N,H,W,3) uint8 device tensor -> (out, M_src, maxC_src, status) device tensors.  A tile whose
        status is non-zero (1 = empty tissue mask, 2 = degenerate, 3 = a zero 99th-percentile concentration) is passed through unchanged.
        ``ws``: an ``engine.Workspace`` to reuse (ONE stream at a time); by default every call takes its scratch from
        torch's stream-ordered caching allocator, so concurrent streams / threads never share it."""
        return self._transform_tiles(tiles, out=out, ws=ws)

    def state_dict(self):
        return {"method": self.method, "stain_matrix_target": np.array(self.stain_matrix_target),
                "maxC_target": np.array(self.maxC_target)}

    def load_state_dict(self, d):
        assert d["method"] == self.method
        self.stain_matrix_target = np.array(d["stain_matrix_target"], dtype=np.float64).reshape(2, 3)
        self.maxC_target = np.array(d["maxC_target"], dtype=np.float64).reshape(1, 2)
        self._target = None                      # the state of an earlier fit() no longer describes this target
        self._target_concentrations = None


class MacenkoNormalizer(ExtractiveStainNormalizer):
    """StainTools-style alias named by BASELINE.json's north_star."""

    def __init__(self):
        super().__init__("macenko")


class VahadaneNormalizer(ExtractiveStainNormalizer):
    def __init__(self):
        super().__init__("vahadane")


class ReinhardStainNormalizer(object):
    """normalization/normalizer.py:54-94 (exported at stainlib/__init__.py:28): Reinhard colour transfer in cv2's 8-bit Lab.

    ``fit`` keeps ``target_means`` / ``target_stds`` as tuples of (1,1) float64 arrays like cv2.meanStdDev returns them.
    The whole transform is three sweeps of csrc/lab.hip (two histogram sweeps, one table-driven map); OpenCV's integer Lab
    conversions are restated there -- parity unpinned against cv2 itself, the reference's own arithmetic
    around them is golden-pinned."""

    def __init__(self, target_means=0, target_stds=0):
        self.target_means = target_means                                  # normalizer.py:61-62
        self.target_stds = target_stds

    def fit(self, target):
        from .. import engine
        st = engine.reinhard_stats(_to_device(target), standardize=True)[0].cpu().numpy()      # normalizer.py:65-66
        self.target_means = tuple(np.array([[st[1 + c]]]) for c in range(3))
        self.target_stds = tuple(np.array([[st[4 + c]]]) for c in range(3))

    def _targets(self):
        return ([float(np.asarray(m).reshape(-1)[0]) for m in self.target_means],
                [float(np.asarray(s).reshape(-1)[0]) for s in self.target_stds])

    def transform(self, I, mask_background=False, luminosity_threshold=0.8):
        """normalizer.py:70-94."""
        from .. import engine
        from ..utils.excepts import TissueMaskException
        tm, ts = self._targets()
        out, st = engine.reinhard_transform(_to_device(I), tm, ts, mask_background, luminosity_threshold)
        if mask_background and int(st[0, 7]) == 0:
            raise TissueMaskException("Empty tissue mask computed")       # stain_utils.py:46-47 via normalizer.py:86
        return out[0].cpu().numpy()

    def transform_batch(self, tiles, mask_background=False, luminosity_threshold=0.8, out=None, ws=None):
        """Batched extension: (N,H,W,3) uint8 device tensor -> (out, stats (N,8): p90, means, stds, tissue pixels)."""
        from .. import engine
        tm, ts = self._targets()
        return engine.reinhard_transform(tiles, tm, ts, mask_background, luminosity_threshold, out=out, ws=ws)
