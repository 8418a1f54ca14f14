"""Host-side mirror of the pieces of stainlib/utils/stain_utils.py that sit on the hot path.

Everything numeric is a call into the HIP engine (stainlib_amd.engine); this module only adapts
numpy uint8 HWC images (the reference's contract) to batched device tensors and maps per-tile
status codes to the reference's exceptions.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from .excepts import TissueMaskException

_UINT8_MSG = "Image should be RGB uint8."          # stain_utils.py:40, macenko_stain_extractor.py:16


def is_image(I) -> bool:
    """stain_utils.py:126-134 -- any 3-D ndarray (channel count is not checked by the reference)."""
    return isinstance(I, np.ndarray) and I.ndim == 3


def is_uint8_image(I) -> bool:
    """stain_utils.py:136-144."""
    return is_image(I) and I.dtype == np.uint8


def _to_device(I: np.ndarray):
    """(H,W,3) uint8 ndarray -> (1,H,W,3) device tensor."""
    import torch
    if I.shape[2] != 3:
        raise ValueError(f"expected 3 channels, got shape {I.shape}")   # the reference fails later, inside cv2
    return torch.from_numpy(np.ascontiguousarray(I)[None]).cuda()


def raise_for_status(status: int) -> None:
    """Per-tile status of the engine -> what the reference does for a single image."""
    from .. import _ffi
    if status == _ffi.TILE_EMPTY_MASK:
        raise TissueMaskException("Empty tissue mask computed")          # stain_utils.py:47
    if status == _ffi.TILE_DEGENERATE_COV:
        # fewer than two tissue pixels: the reference computes np.cov of a single row (NaN) and fails inside eigh.
        # tissue of a single colour: its two stain vectors coincide and its concentrations are inf/NaN; reported too
        raise np.linalg.LinAlgError("degenerate tile: fewer than two tissue pixels, or tissue of a single colour (singular stain matrix)")


class ABCStainExtractor(ABC):
    """stain_utils.py:8-17."""

    @staticmethod
    @abstractmethod
    def get_stain_matrix(I):
        """Estimate the 2x3 stain matrix of an image."""


class ABCTissueLocator(ABC):
    """stain_utils.py:19-27."""

    @staticmethod
    @abstractmethod
    def get_tissue_mask(I):
        """Boolean tissue mask of an image."""


class LuminosityThresholdTissueLocator(ABCTissueLocator):
    """stain_utils.py:29-48; the OpenCV 8-bit Lab L test runs as an integer LUT kernel."""

    @staticmethod
    def get_tissue_mask(I, luminosity_threshold=0.8):
        assert is_uint8_image(I), _UINT8_MSG
        from .. import engine
        mask, counts = engine.tissue_mask(_to_device(I), luminosity_threshold)
        if int(counts[0]) == 0:
            raise TissueMaskException("Empty tissue mask computed")
        return mask[0].cpu().numpy().astype(bool)


class LuminosityStandardizer(object):
    """stain_utils.py:50-67 (exported at stainlib/__init__.py:30).  cv2's 8-bit Lab both ways is OpenCV's integer algorithm
    restated in csrc/lab.hip (parity unpinned against cv2 itself); the percentile and the rescaling of L are
    the reference's arithmetic, evaluated once per byte value."""

    @staticmethod
    def standardize(I, percentile=95):
        assert is_uint8_image(I), _UINT8_MSG
        from .. import engine
        out, _ = engine.luminosity_standardize(_to_device(I), percentile)
        return out[0].cpu().numpy()


def get_concentrations(I, stain_matrix, regularizer=0.01):
    """stain_utils.py:69-78 -> (P, 2) float64 (binary32 values from the device)."""
    assert is_uint8_image(I), _UINT8_MSG
    from .. import engine
    C = engine.concentrations(_to_device(I), np.asarray(stain_matrix, dtype=np.float64)[None], regularizer)
    return C[0].cpu().numpy().astype(np.float64)


def get_sign(x):
    """The sign of a scalar: +1, -1 or 0 (None for NaN, like the chain of comparisons of stain_utils.py:80-91)."""
    if x > 0:
        return +1
    if x < 0:
        return -1
    if x == 0:
        return 0
    return None


def normalize_matrix_rows(A):
    """stain_utils.py:93-99 (six numbers: host arithmetic)."""
    A = np.asarray(A, dtype=np.float64)
    return A / np.linalg.norm(A, axis=1)[:, None]


def convert_RGB_to_OD(I):
    """stain_utils.py:101-112: max(-ln(max(I,1)/255), 1e-6), float64, same shape (a 256-entry table on the device)."""
    assert is_uint8_image(I), _UINT8_MSG
    from .. import engine
    return engine.rgb_to_od(_to_device(I))[0].cpu().numpy()


def convert_OD_to_RGB(OD):
    """stain_utils.py:114-124: uint8(255 * exp(-max(OD, 1e-6))), same shape; asserts on negative optical densities."""
    import torch
    from .. import engine
    od = torch.from_numpy(np.ascontiguousarray(OD, dtype=np.float64)).cuda()
    out, neg = engine.od_to_rgb(od)
    assert int(neg[0]) == 0, "Negative optical density."                # stain_utils.py:122
    return out.cpu().numpy()


def lab_split(I):
    """stain_utils.py:146-158: (I1, I2, I3) float32 planes L8/2.55, a8-128, b8-128 of cv2's 8-bit Lab."""
    from .. import engine
    I1, I2, I3 = engine.lab_split(_to_device(I))
    return I1[0].cpu().numpy(), I2[0].cpu().numpy(), I3[0].cpu().numpy()


def merge_back(I1, I2, I3):
    """stain_utils.py:160-172: planes (float32 or float64, like numpy promotes them) -> RGB uint8.  The reference scales its
    arguments in place; this mirror leaves them untouched."""
    import torch
    from .. import engine
    dt = np.result_type(I1, I2, I3)
    dt = np.float64 if dt == np.float64 else np.float32
    planes = [torch.from_numpy(np.ascontiguousarray(np.asarray(p), dtype=dt)[None]).cuda() for p in (I1, I2, I3)]
    return engine.lab_merge(*planes)[0].cpu().numpy()


def get_mean_std(I):
    """stain_utils.py:174-186: ((m1, m2, m3), (sd1, sd2, sd3)), each a (1,1) float64 array like cv2.meanStdDev returns."""
    from .. import engine
    st = engine.reinhard_stats(_to_device(I), standardize=False)[0].cpu().numpy()
    return tuple(np.array([[st[1 + c]]]) for c in range(3)), tuple(np.array([[st[4 + c]]]) for c in range(3))


def standardize_brightness(I):
    """stain_utils.py:188-194: uint8(clip(I * 255.0 / percentile(I, 90), 0, 255))."""
    from .. import engine
    out, _ = engine.standardize_brightness(_to_device(I))
    return out[0].cpu().numpy()
