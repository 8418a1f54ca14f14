"""CPU-only: the host-side mirror of the reference interface and the C-ABI library's surface.
No compute call is made here (there is no GPU in this container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import stainlib_amd
from stainlib_amd import _ffi
from stainlib_amd.augmentation import augmenter as aug
from stainlib_amd.utils import excepts, stain_utils

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "stainlib_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|void|const char\*)\s+(sl_\w+)\(", hdr, flags=re.M))
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)
    lib = C.CDLL(_ffi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _ffi.lib().sl_version() == 100
    assert _ffi.lib().sl_error_string(-2).decode() == "workspace missing or too small"


def test_default_params_match_reference_constants():
    p = _ffi.default_params()
    # macenko_stain_extractor.py:7, stain_utils.py:69, vahadane_stain_extractor.py:19
    assert (p.luminosity_threshold, p.angular_percentile, p.lasso_lambda, p.dl_lambda) == (0.8, 99.0, 0.01, 0.1)
    assert not p.profile


def test_workspace_sizes_are_sane():
    lib = _ffi.lib()
    assert lib.sl_workspace_bytes(_ffi.OP_MACENKO_TRANSFORM, 512, 1024, 1024) > 0
    assert lib.sl_workspace_bytes(_ffi.OP_MACENKO_TRANSFORM, 512, 1024, 1024) < (1 << 30)
    assert lib.sl_workspace_bytes(_ffi.OP_HED_AUGMENT, 10, 512, 512) >= 80
    assert lib.sl_workspace_bytes(_ffi.OP_MACENKO_FIT, 0, 8, 8) == 0


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = _ffi.lib()
    assert lib.sl_normalize_apply(None, None, 1, 8, 8, None, None, None, None, 0.01, None, None) == -1
    assert lib.sl_macenko_fit(None, 1, 8, 8, None, None, None, None, None, 0, None) == -1
    assert lib.sl_hed_augment(None, None, 1, 8, 8, None, None, 0.05, 0.95, 0, None, None, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", "/nonexistent/libstainlib_hip.so")
    with pytest.raises(_ffi.StainlibHipError, match="no CPU fallback"):
        _ffi.lib()


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "stainlib_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(root, f)


def test_export_list_mirrors_reference():
    # stainlib/__init__.py:19-30 (classes on the hot path) + the StainTools-style aliases of the north star
    for name in ("MacenkoStainExtractor", "VahadaneStainExtractor", "HedLighterColorAugmenter",
                 "HedLightColorAugmenter", "HedStrongColorAugmenter", "ExtractiveStainNormalizer",
                 "MacenkoNormalizer", "VahadaneNormalizer", "StainAugmentor"):
        assert hasattr(stainlib_amd, name)


def test_error_contract_matches_reference_goldens():
    g = np.load(os.path.join(GOLDEN, "errors.npz"))
    with pytest.raises(Exception) as e:
        stainlib_amd.ExtractiveStainNormalizer("reinhard")
    assert str(e.value) == str(g["bad_method_msg"])
    with pytest.raises(Exception, match="Method not recognized."):
        stainlib_amd.StainAugmentor("reinhard")
    with pytest.raises(AssertionError) as e:
        stainlib_amd.MacenkoStainExtractor.get_stain_matrix(np.zeros((8, 8, 3), np.float32))
    assert str(e.value) == str(g["float_msg"])
    with pytest.raises(AssertionError, match="Image should be RGB uint8."):
        stainlib_amd.ExtractiveStainNormalizer("macenko").fit(np.zeros((8, 8), np.uint8))
    with pytest.raises(excepts.TissueMaskException) as e:
        stain_utils.raise_for_status(_ffi.TILE_EMPTY_MASK)
    assert str(e.value) == str(g["white_msg"])
    err = excepts.InvalidRangeError("Cutoff", (2, 1))
    assert str(err) == "Invalid range of Cutoff: (2, 1)" and err.title == "Cutoff" and err.range == (2, 1)
    assert issubclass(excepts.InvalidRangeError, excepts.DigitalPathologyAugmentationError)
    assert issubclass(excepts.DigitalPathologyAugmentationError, excepts.DigitalPathologyError)


@pytest.mark.parametrize("kw,title", [
    (dict(haematoxylin_sigma_range=(0.5, 0.1)), "Haematoxylin Sigma"),
    (dict(eosin_sigma_range=(-1.5, 0.1)), "Eosin Sigma"),
    (dict(dab_sigma_range=(0.0, 1.1)), "Dab Sigma"),
    (dict(haematoxylin_bias_range=(0.1,)), "Haematoxylin Bias"),
    (dict(eosin_bias_range=(0.3, 0.2)), "Eosin Bias"),
    (dict(dab_bias_range=(-2, 0)), "Dab Bias"),
    (dict(cutoff_range=(-0.1, 0.5)), "Cutoff"),
    (dict(cutoff_range=(0.2, 1.5)), "Cutoff"),
])
def test_range_validation(kw, title):
    base = dict(haematoxylin_sigma_range=None, haematoxylin_bias_range=None, eosin_sigma_range=None,
                eosin_bias_range=None, dab_sigma_range=None, dab_bias_range=None, cutoff_range=None)
    base.update(kw)
    with pytest.raises(excepts.InvalidRangeError) as e:
        aug.HedColorAugmenter(**base)
    assert e.value.title == title                       # augmenter.py:167-271


def test_hed_augmenter_state_and_random_stream():
    a = stainlib_amd.HedLighterColorAugmenter()
    assert a.keyword == "hed_color" and a.shapes({0: (4, 4)}) == {0: (4, 4)}
    assert a._sigmas == [-0.03] * 3 and a._biases == [-0.03] * 3          # augmenter.py:194-198,246-250
    assert a._sigma_ranges == [(-0.03, 0.03)] * 3 and tuple(a._cutoff_range) == (0.05, 0.95)
    for path, seed in (("hed_128_s2_np0.npz", 0), ("hed_128_s3_np123.npz", 123)):
        g = np.load(os.path.join(GOLDEN, path))
        np.random.seed(seed)
        a.randomize()
        np.testing.assert_array_equal(a._sigmas, g["sigmas"])               # same six draws, same order
        np.testing.assert_array_equal(a._biases, g["biases"])
    assert stainlib_amd.HedLightColorAugmenter()._sigma_ranges[0] == (-0.1, 0.1)
    assert stainlib_amd.HedStrongColorAugmenter()._bias_ranges[2] == (-1.0, 1.0)
    none = aug.HedColorAugmenter(None, None, None, None, None, None, None)
    none.randomize()
    assert none._sigmas == [1.0] * 3 and none._biases == [0.0] * 3          # augmenter.py:338 quirk
    assert none._cutoff_range == [0.0, 1.0]


def test_stain_augmentor_draw_order():
    g = np.load(os.path.join(GOLDEN, "stainaug_128_s2_np7.npz"))
    s = stainlib_amd.StainAugmentor("macenko")
    s.n_stains = 2
    seen = []
    s.pop_with = lambda ab: seen.append(list(ab))
    np.random.seed(int(g["npseed"]))
    s.pop()
    np.testing.assert_array_equal(seen[0], g["draws0"])                      # alpha0, beta0, alpha1, beta1
