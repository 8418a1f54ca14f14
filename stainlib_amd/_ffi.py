"""ctypes binding of the C ABI in include/stainlib_hip.h.

The HIP library is the product: if ``libstainlib_hip.so`` is missing or a call fails this
module raises -- there is no CPU fallback anywhere in ``stainlib_amd``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# STAINLIB_HIP_LIB: development override (tools/ load the -DSL_DEVTOOLS build, libstainlib_hip_dev.so, through it)
LIB_PATH = os.environ.get("STAINLIB_HIP_LIB") or os.path.join(_HERE, "csrc", "libstainlib_hip.so")


class SlProfile(C.Structure):
    _fields_ = [
        ("events", C.POINTER(C.c_void_p)),
        ("tags", C.POINTER(C.c_int32)),
        ("tiles", C.POINTER(C.c_int32)),
        ("capacity", C.c_int32),
        ("used", C.c_int32),
        ("mask", C.c_int32),
        ("reserved", C.c_int32),
    ]


PROF_MOMENTS, PROF_SELECT_ANGLE, PROF_SELECT_CONC, PROF_FINISH, PROF_APPLY, PROF_DICT = 1, 2, 4, 8, 16, 32
PROF_FUSED_FIT, PROF_FUSED_TRANSFORM = 64, 128
PROF_NAMES = {1: "k_moments", 2: "k_select<merged>", 4: "k_select<conc>", 8: "k_finish_*", 16: "k_apply", 32: "k_dict",
              64: "k_macenko_fused<fit>", 128: "k_macenko_fused<transform>"}


class SlParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("reserved0", C.c_uint32),
        ("luminosity_threshold", C.c_double),
        ("angular_percentile", C.c_double),
        ("lasso_lambda", C.c_double),
        ("dl_lambda", C.c_double),
        ("dl_max_sweeps", C.c_int32),
        ("schedule", C.c_int32),
        ("dl_tol", C.c_double),
        ("profile", C.POINTER(SlProfile)),
        ("fallbacks_out", C.c_void_p),
        ("resweeps_out", C.c_void_p),
        ("fused_min_tiles", C.c_int32),
        ("prefilter", C.c_int32),
        ("prefilter_out", C.c_void_p),
        ("two_sweep", C.c_int32),
        ("reserved1", C.c_int32),
        ("twosweep_out", C.c_void_p),
    ]


class StainlibHipError(RuntimeError):
    pass


# ops (sl_workspace_bytes)
(OP_MACENKO_FIT, OP_MACENKO_TRANSFORM, OP_VAHADANE_FIT, OP_VAHADANE_TRANSFORM, OP_HED_AUGMENT, OP_STAIN_AUGMENT, OP_TILE_MOMENTS,
 OP_LAB_STATS) = range(1, 9)
# skimage semantics of sl_hed_augment (only 0.18 is golden-pinned)
HED_SKIMAGE_018, HED_SKIMAGE_019, HED_SKIMAGE_017, HED_EXPERIMENTAL_LOG10 = range(4)
# selection key sets of the pooled slide-level mode (two targets each)
KEYSET_ANGLE, KEYSET_CONC = range(2)
# per-tile status
TILE_OK, TILE_EMPTY_MASK, TILE_DEGENERATE_COV, TILE_ZERO_MAXC = range(4)

_P = C.c_void_p
_SIGNATURES = {
    "sl_version": (C.c_int, []),
    "sl_error_string": (C.c_char_p, [C.c_int]),
    "sl_default_params": (None, [C.POINTER(SlParams)]),
    "sl_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "sl_workspace_bytes_for": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams)]),
    "sl_macenko_fit": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), _P, _P, _P, _P, C.c_size_t, _P]),
    "sl_vahadane_fit": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "sl_normalize_apply": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_double, _P, _P]),
    "sl_macenko_transform": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "sl_vahadane_transform": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "sl_hed_augment": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_double, C.c_double, C.c_int, _P, _P, C.c_size_t, _P]),
    "sl_hed_augment_f64": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_double, C.c_double, C.c_int, _P, _P, C.c_size_t, _P]),
    "sl_rgb_to_od": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sl_stain_augment": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.POINTER(SlParams), _P]),
    "sl_tissue_mask": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_double, _P, _P, _P]),
    "sl_concentrations": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_double, _P, _P]),
    "sl_grayscale_augment": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sl_od_to_rgb": (C.c_int, [_P, C.c_size_t, _P, _P, _P]),
    "sl_rgb_to_lab8": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "sl_lab8_to_rgb": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "sl_lab_split": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "sl_lab_merge": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sl_standardize_brightness": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_size_t, _P]),
    "sl_reinhard_stats": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_size_t, _P]),
    "sl_reinhard_transform": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_double, _P, _P, C.c_size_t, _P]),
    "sl_luminosity_standardize": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_double, _P, _P, C.c_size_t, _P]),
    "sl_tile_moments": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), _P, _P, C.c_size_t, _P]),
    "sl_slide_key_histogram": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), C.c_int, C.POINTER(C.c_double),
                                          C.POINTER(C.c_uint32), C.c_int, _P, _P]),
    "sl_slide_key_next_above": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), C.c_int, C.POINTER(C.c_double),
                                           C.POINTER(C.c_uint32), _P, _P]),
    "sl_slide_key_histogram16": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), C.c_int, C.POINTER(C.c_double),
                                            C.POINTER(C.c_uint32), _P, _P]),
    "sl_slide_key_histogram_sampled": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), C.c_int, C.POINTER(C.c_double),
                                                  C.POINTER(C.c_uint32), C.c_int, C.c_int, _P, _P]),
    "sl_slide_key_window": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), C.c_int, C.POINTER(C.c_double),
                                       C.POINTER(C.c_uint32), _P, _P]),
    # device-driven pooled statistics (state: SL_POOL_STATE_DOUBLES doubles on the device)
    "sl_pool_begin": (C.c_int, [_P, C.POINTER(SlParams), _P, _P]),
    "sl_pool_histogram": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    "sl_pool_pick": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "sl_pool_window": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), C.c_int, _P, _P, _P]),
    "sl_pool_resolve": (C.c_int, [_P, C.c_int, _P, C.POINTER(SlParams), _P]),
    # the pooled statistics in one full sweep (state: SL_POOL2_STATE_DOUBLES doubles; workspace: sl_pool2_workspace_bytes)
    "sl_pool2_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "sl_pool2_sample": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), C.c_int, _P, C.c_size_t, _P, _P]),
    "sl_pool2_begin": (C.c_int, [_P, C.POINTER(SlParams), C.c_int, _P, _P]),
    "sl_pool2_hist": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), C.c_int, _P, _P, C.c_size_t, _P, _P]),
    "sl_pool2_bands": (C.c_int, [_P, C.c_int, _P, _P]),
    "sl_pool2_sweep": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), C.c_int, _P, _P, C.c_size_t, _P, _P]),
    "sl_pool2_exact": (C.c_int, [_P, _P, _P]),
    "sl_pool2_step": (C.c_int, [_P, C.c_int, _P, _P]),
    "sl_pool2_local": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(SlParams), C.c_int, _P, C.c_size_t, _P, _P]),
}
POOL_STATE_DOUBLES, POOL_M, POOL_MAXC, POOL_STATUS, POOL_MISS = 64, 0, 6, 8, 9
POOL2_STATE_DOUBLES, POOL2_HIST_WORDS, POOL2_WHY = 256, 2 * 8192 + 8 * 32, 33
EXPECTED_VERSION = 600     # the SL_VERSION this binding (SlParams, signatures) was written for
EXPORTS = tuple(_SIGNATURES)

_lib = None


def lib() -> C.CDLL:
    """Load libstainlib_hip.so (once).  torch is imported first so that the library binds to
    the same HIP runtime (libamdhip64.so.7) torch already has in the process."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StainlibHipError(
                f"{LIB_PATH} not found: build it with `make -C stainlib_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        import torch  # noqa: F401  (loads the HIP runtime)
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(h, name)   # AttributeError here == a symbol the header declares is missing
            fn.restype = res
            fn.argtypes = args
        if h.sl_version() != EXPECTED_VERSION:      # a stale .so would read SlParams at other offsets (round-5 advisor finding)
            raise StainlibHipError(f"{LIB_PATH} is ABI {h.sl_version()}, this package binds ABI {EXPECTED_VERSION}: rebuild it "
                                   "(make -C stainlib_amd/csrc)")
        _lib = h
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise StainlibHipError(f"{what} failed: {lib().sl_error_string(code).decode()} (code {code})")


def default_params() -> SlParams:
    p = SlParams()
    lib().sl_default_params(C.byref(p))
    return p
