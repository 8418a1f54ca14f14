"""stainlib_amd -- MI355X-native drop-in for the hot path of sebastianffx/stainlib.

The export list mirrors stainlib/__init__.py:19-30 for the classes on the path named by BASELINE.json
plus GrayscaleAugmentor; ReinhardStainNormalizer and LuminosityStandardizer need a bit-exact OpenCV
Lab<->RGB restatement in both directions that cannot be pinned here and are not offered.
Importing the package does not need a GPU; calling anything numeric does, and fails loudly without the
HIP library -- there is no CPU fallback.
"""
from . import _ffi  # noqa: F401
from .augmentation.augmenter import (HedLightColorAugmenter, HedLighterColorAugmenter,  # noqa: F401
                                     HedStrongColorAugmenter, StainAugmentor, GrayscaleAugmentor)
from .extraction.macenko_stain_extractor import MacenkoStainExtractor  # noqa: F401
from .extraction.vahadane_stain_extractor import VahadaneStainExtractor  # noqa: F401
from .normalization.normalizer import (ExtractiveStainNormalizer, MacenkoNormalizer,  # noqa: F401
                                       VahadaneNormalizer)
from .utils.excepts import InvalidRangeError, TissueMaskException  # noqa: F401

__version__ = "0.1.0"
