"""stainlib_amd -- MI355X-native drop-in for the hot path of sebastianffx/stainlib.

The export list mirrors stainlib/__init__.py:19-30.  ReinhardStainNormalizer and LuminosityStandardizer (SURVEY
8f-3 / 8f-4) sit on OpenCV's 8-bit Lab conversions, restated in csrc/lab.hip: parity unpinned against cv2 itself.
Importing the package does not need a GPU; calling anything numeric does, and fails loudly without the
HIP library -- there is no CPU fallback.
"""
from . import _ffi  # noqa: F401
from .augmentation.augmenter import (HedLightColorAugmenter, HedLighterColorAugmenter,  # noqa: F401
                                     HedStrongColorAugmenter, StainAugmentor, GrayscaleAugmentor)
from .extraction.macenko_stain_extractor import MacenkoStainExtractor  # noqa: F401
from .extraction.vahadane_stain_extractor import VahadaneStainExtractor  # noqa: F401
from .normalization.normalizer import (ExtractiveStainNormalizer, MacenkoNormalizer,  # noqa: F401
                                       ReinhardStainNormalizer, VahadaneNormalizer)
from .utils.stain_utils import LuminosityStandardizer  # noqa: F401
from .utils.excepts import InvalidRangeError, TissueMaskException  # noqa: F401

__version__ = "0.6.0"
