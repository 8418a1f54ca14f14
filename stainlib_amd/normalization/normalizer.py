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
    def _fit_tiles(self, tiles):
        from .. import engine
        if self.method == "macenko":
            return engine.macenko_fit(tiles)
        M, maxC, status, _ = engine.vahadane_fit(tiles)
        return M, maxC, status

    def _transform_tiles(self, tiles, out=None):
        from .. import engine
        fn = engine.macenko_transform if self.method == "macenko" else engine.vahadane_transform
        return fn(tiles, self.stain_matrix_target, self.maxC_target.reshape(2), out=out)

    # -- reference API -----------------------------------------------------------------------------
    def fit(self, target):
        """Fit to a target image (RGB uint8), normalizer.py:27-36."""
        assert is_uint8_image(target), _UINT8_MSG
        M, maxC, status = self._fit_tiles(_to_device(target))
        raise_for_status(int(status[0]))
        self.stain_matrix_target = M[0].cpu().numpy()
        self.maxC_target = maxC[0].cpu().numpy().reshape((1, 2))
        self._target = target
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
        out, _, _, status = self._transform_tiles(_to_device(I))
        st = int(status[0])
        raise_for_status(st)
        if st != 0:
            warnings.warn("99th-percentile concentration of the source is zero; the reference divides by it",
                          RuntimeWarning)
        return out[0].cpu().numpy()

    # -- batched extension -------------------------------------------------------------------------
    def fit_batch_targets(self, tiles):
        """Per-tile (M, maxC, status) device tensors for a batch of candidate targets."""
        return self._fit_tiles(tiles)

    def transform_batch(self, tiles, out=None):
        """(N,H,W,3) uint8 device tensor -> (out, M_src, maxC_src, status) device tensors.  A tile whose
        status is non-zero (1 = empty tissue mask, 2 = degenerate) is passed through unchanged."""
        return self._transform_tiles(tiles, out=out)

    def state_dict(self):
        return {"method": self.method, "stain_matrix_target": np.array(self.stain_matrix_target),
                "maxC_target": np.array(self.maxC_target)}

    def load_state_dict(self, d):
        assert d["method"] == self.method
        self.stain_matrix_target = np.array(d["stain_matrix_target"], dtype=np.float64).reshape(2, 3)
        self.maxC_target = np.array(d["maxC_target"], dtype=np.float64).reshape(1, 2)


class MacenkoNormalizer(ExtractiveStainNormalizer):
    """StainTools-style alias named by BASELINE.json's north_star."""

    def __init__(self):
        super().__init__("macenko")


class VahadaneNormalizer(ExtractiveStainNormalizer):
    def __init__(self):
        super().__init__("vahadane")
