"""Colour augmenters of stainlib/augmentation/augmenter.py (lines 19-372, 403-449) on the HIP engine.

Public behaviour kept from the reference: class names, ``keyword``/``shapes``, the private attributes
users peek at (``_sigmas``, ``_biases``, ``_sigma_ranges``, ``_bias_ranges``, ``_cutoff_range``), the
range validation and its ``InvalidRangeError`` titles, the order in which ``randomize()`` / ``pop()``
consume the GLOBAL ``np.random`` stream, and the "un-randomized augmenter applies the lower bounds"
quirk (augmenter.py:194-198, 246-250).  The arithmetic is ``sl_hed_augment`` / ``sl_stain_augment``.
"""
from __future__ import annotations

import numpy as np

from ..utils.excepts import InvalidRangeError
from ..utils.stain_utils import (_UINT8_MSG, LuminosityThresholdTissueLocator, _to_device, get_concentrations,
                                 is_uint8_image)

_CHANNELS = ("Haematoxylin", "Eosin", "Dab")


class AugmenterBase(object):
    """augmenter.py:19-70."""

    def __init__(self, keyword):
        self._keyword = keyword

    @property
    def keyword(self):
        return self._keyword

    def shapes(self, target_shapes):
        """Output shapes equal input shapes (augmenter.py:41-53)."""
        return target_shapes

    def transform(self, patch):
        pass

    def randomize(self):
        pass


class ColorAugmenterBase(AugmenterBase):
    """augmenter.py:72-84."""


_CUTOFF_BAND = 5e-6      # relative distance to a cutoff bound below which the reference's float32 mean is re-evaluated on the host


def _checked(title, rng, lowest):
    """One interval of the constructor: None passes; otherwise a pair lo <= hi inside [lowest, 1]."""
    if rng is not None:
        bad = len(rng) != 2 or rng[1] < rng[0] or rng[0] < lowest or 1.0 < rng[1]
        if bad:
            raise InvalidRangeError(title, rng)
    return rng


class HedColorAugmenter(ColorAugmenterBase):
    """Colour perturbation in HED space: value * (1 + sigma) + bias per channel (augmenter.py:86-344).

    ``skimage_mode`` selects the rgb2hed/hed2rgb semantics.  Only "0.18" (default) is pinned by vectors from a real
    scikit-image; the others are restated from memory and checked against the CPU restatement only: "0.19" (stains clamped
    at zero after separation), "0.17" (presumed behaviour of the release the reference's environment.yml pins:
    -ln(rgb + 2) and exp(x) - 2, clipped) and "experimental_log10" (the same with base-10 logarithms)."""

    def __init__(self, haematoxylin_sigma_range, haematoxylin_bias_range, eosin_sigma_range, eosin_bias_range,
                 dab_sigma_range, dab_bias_range, cutoff_range, skimage_mode="0.18"):
        super().__init__(keyword="hed_color")
        sig = (haematoxylin_sigma_range, eosin_sigma_range, dab_sigma_range)
        bia = (haematoxylin_bias_range, eosin_bias_range, dab_bias_range)
        self._sigma_ranges = [_checked(ch + " Sigma", r, -1.0) for ch, r in zip(_CHANNELS, sig)]
        self._bias_ranges = [_checked(ch + " Bias", r, -1.0) for ch, r in zip(_CHANNELS, bia)]
        # until randomize() is called the lower bounds are applied (augmenter.py:194-198, 246-250)
        self._sigmas = [r[0] if r is not None else 0.0 for r in self._sigma_ranges]
        self._biases = [r[0] if r is not None else 0.0 for r in self._bias_ranges]
        cut = _checked("Cutoff", cutoff_range, 0.0)
        self._cutoff_range = cut if cut is not None else [0.0, 1.0]
        modes = {"0.18": 0, "0.19": 1, "0.17": 2, "experimental_log10": 3}
        if skimage_mode not in modes:
            raise ValueError("skimage_mode must be one of " + ", ".join(repr(m) for m in modes))
        self._skimage_mode = modes[skimage_mode]

    def randomize(self):
        """Six draws from the global numpy stream: sigma H, E, D then bias H, E, D (augmenter.py:333-344)."""
        self._sigmas = [np.random.uniform(low=r[0], high=r[1], size=None) if r is not None else 1.0
                        for r in self._sigma_ranges]
        self._biases = [np.random.uniform(low=r[0], high=r[1], size=None) if r is not None else 0.0
                        for r in self._bias_ranges]

    def transform(self, patch):
        """augmenter.py:276-331 for uint8 patches.  A patch whose mean is outside the cutoff interval is
        returned as the same object."""
        from .. import engine
        if isinstance(patch, np.ndarray) and patch.ndim == 3 and patch.dtype.kind == "f":
            # float branch (augmenter.py:288-289): values in [0,1]; the result is float64, clipped, not rescaled
            import torch
            dev = torch.from_numpy(np.ascontiguousarray(patch, dtype=np.float64)[None]).cuda()
            out, applied = engine.hed_augment_float(dev, [self._sigmas], [self._biases], cutoff=self._cutoff_range,
                                                    skimage_mode=self._skimage_mode)
            return out[0].cpu().numpy() if int(applied[0]) else patch
        if not is_uint8_image(patch):
            raise TypeError("HedColorAugmenter.transform expects a uint8 or float (H, W, 3) ndarray")
        dev = _to_device(patch)
        out, applied, sums = engine.hed_augment(dev, [self._sigmas], [self._biases], cutoff=self._cutoff_range,
                                                skimage_mode=self._skimage_mode, want_sums=True)
        ok = bool(int(applied[0]))
        # The device tests the EXACT mean (integer byte sum); the reference tests np.mean of the float32 image / 255
        # (augmenter.py:291-293), whose pairwise binary32 sum can be off by ~2e-6 relative on a large patch.  Within _CUTOFF_BAND of
        # a bound the reference's own value decides (one host mean, only then).
        exact = float(int(sums[0])) / patch.size / 255.0
        lo, hi = self._cutoff_range
        if min(abs(exact - lo), abs(exact - hi)) <= _CUTOFF_BAND * max(abs(lo), abs(hi), 1e-30):
            ref_mean = np.mean(a=patch.astype(dtype=np.float32)) / 255.0
            ref_ok = bool(lo <= ref_mean <= hi)
            if ref_ok and not ok:                                     # transform after all: no cutoff this time
                out, _ = engine.hed_augment(dev, [self._sigmas], [self._biases], cutoff=(-np.inf, np.inf),
                                            skimage_mode=self._skimage_mode)
            ok = ref_ok
        if not ok:
            return patch                                              # augmenter.py:331
        return out[0].cpu().numpy()

    def transform_batch(self, tiles, sigmas=None, biases=None, out=None):
        """Batched extension: (N,H,W,3) uint8 device tensor; per-tile (N,3) sigmas / biases (defaults: the
        augmenter's current ones for every tile).  Returns (out, applied)."""
        from .. import engine
        n = tiles.shape[0]
        sigmas = [self._sigmas] * n if sigmas is None else sigmas
        biases = [self._biases] * n if biases is None else biases
        out, applied, sums = engine.hed_augment(tiles, sigmas, biases, cutoff=self._cutoff_range,
                                                skimage_mode=self._skimage_mode, out=out, want_sums=True)
        # The same knife-edge rule as transform(): a tile whose EXACT mean lies within _CUTOFF_BAND of a cutoff bound is decided by
        # the reference's own expression (the float32 mean, augmenter.py:291-293) on the host, so that a tile gets the same answer
        # alone and in a batch.  Costs one 8-byte-per-tile read-back per call; tiles near a bound are rare.
        import torch
        lo, hi = self._cutoff_range
        exact = sums.to(torch.float64) / float(tiles.shape[1] * tiles.shape[2] * 3) / 255.0
        band = _CUTOFF_BAND * max(abs(lo), abs(hi), 1e-30)
        near = torch.nonzero(torch.minimum((exact - lo).abs(), (exact - hi).abs()) <= band).reshape(-1).tolist()
        for i in near:
            patch = tiles[i].cpu().numpy()
            ref_mean = np.mean(a=patch.astype(dtype=np.float32)) / 255.0
            ref_ok = bool(lo <= ref_mean <= hi)
            if ref_ok and not int(applied[i]):
                engine.hed_augment(tiles[i:i + 1], [sigmas[i]], [biases[i]], cutoff=(-np.inf, np.inf), skimage_mode=self._skimage_mode,
                                   out=out[i:i + 1])
                applied[i] = 1
            elif not ref_ok and int(applied[i]):
                out[i].copy_(tiles[i])                                    # augmenter.py:331: the patch comes back unchanged
                applied[i] = 0
        return out, applied

    def randomize_batch(self, n):
        """n successive randomize() calls (same global stream order) -> (n,3) sigmas, (n,3) biases."""
        s, b = [], []
        for _ in range(n):
            self.randomize()
            s.append(list(self._sigmas))
            b.append(list(self._biases))
        return np.array(s, dtype=np.float64), np.array(b, dtype=np.float64)


class HedColorAugmenter1(HedColorAugmenter):
    """Symmetric ranges (-t, t) for every sigma and bias, cutoff (0.05, 0.95) (augmenter.py:346-360)."""

    def __init__(self, thresh, skimage_mode="0.18"):
        r = (-thresh, thresh)
        super().__init__(r, r, r, r, r, r, (0.05, 0.95), skimage_mode=skimage_mode)


class HedLighterColorAugmenter(HedColorAugmenter1):
    def __init__(self, skimage_mode="0.18"):
        super().__init__(0.03, skimage_mode=skimage_mode)              # augmenter.py:362-364


class HedLightColorAugmenter(HedColorAugmenter1):
    def __init__(self, skimage_mode="0.18"):
        super().__init__(0.1, skimage_mode=skimage_mode)               # augmenter.py:366-368


class HedStrongColorAugmenter(HedColorAugmenter1):
    def __init__(self, skimage_mode="0.18"):
        super().__init__(1.0, skimage_mode=skimage_mode)               # augmenter.py:370-372


class StainAugmentor(object):
    """Stain-space augmentation of a fitted image (augmenter.py:403-449)."""

    def __init__(self, method, sigma1=0.2, sigma2=0.2, augment_background=False):
        name = method.lower()
        if name == 'macenko':
            from ..extraction.macenko_stain_extractor import MacenkoStainExtractor
            self.extractor = MacenkoStainExtractor
        elif name == 'vahadane':
            from ..extraction.vahadane_stain_extractor import VahadaneStainExtractor
            self.extractor = VahadaneStainExtractor
        else:
            raise Exception('Method not recognized.')                  # augmenter.py:411
        self.sigma1 = sigma1
        self.sigma2 = sigma2
        self.augment_background = augment_background
        self._image = None
        self._dev = None
        self._conc = None
        self._mask = None

    def fit(self, I):
        """augmenter.py:416-426: stain matrix of I; concentrations and tissue mask stay on the device
        (recomputed inside the pop kernel) and are only materialised if the attributes are read."""
        assert is_uint8_image(I), _UINT8_MSG
        self.image_shape = I.shape
        self.stain_matrix = self.extractor.get_stain_matrix(I)
        self.n_stains = 2
        self._image = I
        self._dev = _to_device(I)
        self._conc = None
        self._mask = None

    @property
    def source_concentrations(self):
        if self._conc is None and self._image is not None:
            self._conc = get_concentrations(self._image, self.stain_matrix)
        return self._conc

    @property
    def tissue_mask(self):
        if self._mask is None and self._image is not None:
            self._mask = LuminosityThresholdTissueLocator.get_tissue_mask(self._image).ravel()
        return self._mask

    def pop(self):
        """One augmented version of the fitted image; draws alpha0, beta0, alpha1, beta1 from the global
        numpy stream in the reference's order (augmenter.py:435-437)."""
        ab = []
        for _ in range(self.n_stains):
            ab.append(np.random.uniform(1 - self.sigma1, 1 + self.sigma1))
            ab.append(np.random.uniform(-self.sigma2, self.sigma2))
        return self.pop_with(ab)

    def pop_with(self, alpha_beta):
        from .. import engine
        out = engine.stain_augment(self._dev, self.stain_matrix[None], [alpha_beta], self.augment_background)
        return out[0].cpu().numpy()


class GrayscaleAugmentor(object):
    """Grayscale intensity augmentation of a fitted image (augmenter.py:374-401)."""

    def __init__(self, sigma1=0.2, sigma2=0.2, augment_background=False):
        self.sigma1 = sigma1
        self.sigma2 = sigma2
        self.augment_background = augment_background
        self._dev = None

    def fit(self, I):
        """augmenter.py:380-388 (the tissue mask is computed there too, so an all-background image raises here)."""
        self.image_shape = I.shape
        self.tissue_mask = LuminosityThresholdTissueLocator.get_tissue_mask(I).ravel()
        self.image = I
        self._dev = _to_device(I)

    def pop(self):
        """One augmented version; alpha ~ U(0.8, 1.2), beta ~ U(-0.2, 0.2) from the global numpy stream -- the reference
        uses the literal 0.2 here, not sigma1 / sigma2 (augmenter.py:394-395)."""
        alpha = np.random.uniform(1 - 0.2, 1 + 0.2)
        beta = np.random.uniform(-0.2, 0.2)
        return self.pop_with(alpha, beta)

    def pop_with(self, alpha, beta):
        from .. import engine
        return engine.grayscale_augment(self._dev, [[alpha, beta]])[0].cpu().numpy()

