"""CPU oracle for the stain-normalization hot path -- TEST INFRASTRUCTURE ONLY.

This module is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  ``stainlib_amd`` never imports anything from ``oracle/``.

It restates, in float64 numpy and in the reference's own operation order, the
algorithms of sebastianffx/stainlib v0.6.1 for the path named by BASELINE.json.
Every function cites the reference file:line it follows (paths relative to the
reference checkout).

Pinning status (see DESIGN.md "Oracle"):
  * Reference-owned numpy arithmetic (OD, cov/eigh/percentiles, rescale,
    Beer-Lambert reconstruction, truncating cast, StainAugmentor, HED affine):
    PINNED by tests/golden/*.npz, produced by importing the reference itself
    (tests/golden/make_golden.py) in the build container.
  * skimage rgb2hed/hed2rgb: PINNED against real scikit-image 0.18.3.
  * spams.lasso (spams 2.6.2.5, absent): mathematically pinned -- unique
    optimum, KKT certificate + independent coordinate-descent solver
    (scikit-learn) used as the stand-in when generating the goldens.
  * cv2.cvtColor(RGB2LAB) L channel (opencv-python 4.4.0.46, absent):
    PARITY UNPINNED -- restated from OpenCV's published 8-bit fixed-point
    algorithm (modules/imgproc/src/color_lab.cpp, RGB2Lab_b); no vector from a
    real cv2 is available in this environment.
  * spams.trainDL (Vahadane): PARITY UNPINNED AND UNPINNABLE -- the reference
    call is wall-clock budgeted (iter=-1) and randomly initialised; the oracle
    defines the converged optimum of the same objective as the target.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# OpenCV 8-bit Lab (L channel) restatement
# third-party: opencv-python 4.4.0.46 (stainlib/utils/environment.yml:143),
# call site stainlib/utils/stain_utils.py:41
# --------------------------------------------------------------------------
_GAMMA_SHIFT = 3
_LAB_SHIFT = 12
_LAB_SHIFT2 = _LAB_SHIFT + _GAMMA_SHIFT
# round((1<<12) * sRGB->XYZ(D65) Y row): 0.212671, 0.715160, 0.072169
_CY = (871, 2929, 296)
_LSCALE = (116 * 255 + 50) // 100                               # 296
_LSHIFT = -((16 * 255 * (1 << _LAB_SHIFT2) + 50) // 100)        # -1336934


def _srgb_gamma_tab_b() -> np.ndarray:
    """OpenCV ``sRGBGammaTab_b``: round(2040 * inverse-sRGB-gamma(v/255)), u16[256].

    OpenCV evaluates v/255 in binary32, the power law in binary64, converts
    back to binary32, scales by 255*(1<<gamma_shift) in binary32 and rounds
    half-to-even (cvRound)."""
    x = (np.arange(256, dtype=np.float32) / np.float32(255.0)).astype(np.float64)
    lin = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)
    y = lin.astype(np.float32) * np.float32(255 * (1 << _GAMMA_SHIFT))
    return np.rint(y).astype(np.int64)


def _lab_cbrt_tab_b() -> np.ndarray:
    """OpenCV ``LabCbrtTab_b``: round(2^15 * f(i/2040)), u16[3072], f = Lab cube-root curve."""
    n = 256 * 3 // 2 * (1 << _GAMMA_SHIFT)
    x = (np.float32(1.0) / np.float32(255 * (1 << _GAMMA_SHIFT))
         * np.arange(n, dtype=np.float32)).astype(np.float64)
    lthresh = 216.0 / 24389.0
    lscale = 841.0 / 108.0
    lbias = 16.0 / 116.0
    f = np.where(x < lthresh, x * lscale + lbias, np.cbrt(x))
    return np.rint(f * (1 << _LAB_SHIFT2)).astype(np.int64)


SRGB_GAMMA_TAB = _srgb_gamma_tab_b()
LAB_CBRT_TAB = _lab_cbrt_tab_b()


def lab_y_index(I: np.ndarray) -> np.ndarray:
    """Index into LabCbrtTab_b for the Y (luminance) channel of each pixel."""
    g = SRGB_GAMMA_TAB
    R = g[I[..., 0]]
    G = g[I[..., 1]]
    B = g[I[..., 2]]
    return (R * _CY[0] + G * _CY[1] + B * _CY[2] + (1 << (_LAB_SHIFT - 1))) >> _LAB_SHIFT


def lab_l8(I: np.ndarray) -> np.ndarray:
    """8-bit L channel as ``cv2.cvtColor(I, COLOR_RGB2LAB)[:, :, 0]`` (stain_utils.py:41)."""
    fY = LAB_CBRT_TAB[lab_y_index(I)]
    L = (_LSCALE * fY + _LSHIFT + (1 << (_LAB_SHIFT2 - 1))) >> _LAB_SHIFT2
    return np.clip(L, 0, 255).astype(np.uint8)


def y_index_threshold(luminosity_threshold: float = 0.8) -> int:
    """Largest Y-table index whose L8 satisfies ``L8 / 255.0 < threshold``.

    Both tables are monotone, so ``mask <=> lab_y_index <= y_index_threshold``.
    Returns -1 when no index qualifies."""
    fY = LAB_CBRT_TAB
    L = np.clip((_LSCALE * fY + _LSHIFT + (1 << (_LAB_SHIFT2 - 1))) >> _LAB_SHIFT2, 0, 255)
    ok = (L / 255.0) < luminosity_threshold
    idx = np.nonzero(ok)[0]
    return int(idx.max()) if idx.size else -1


class TissueMaskException(Exception):
    """Oracle-side twin of stainlib/utils/excepts.py:22."""


def is_uint8_image(I) -> bool:
    """stain_utils.py:126-144 (ndarray, ndim == 3, uint8; channels unchecked)."""
    return isinstance(I, np.ndarray) and I.ndim == 3 and I.dtype == np.uint8


def tissue_mask(I: np.ndarray, luminosity_threshold: float = 0.8) -> np.ndarray:
    """LuminosityThresholdTissueLocator.get_tissue_mask, stain_utils.py:32-48."""
    assert is_uint8_image(I), "Image should be RGB uint8."
    L = lab_l8(I) / 255.0
    mask = L < luminosity_threshold
    if mask.sum() == 0:
        raise TissueMaskException("Empty tissue mask computed")
    return mask


# --------------------------------------------------------------------------
# Optical density, stain_utils.py:101-124
# --------------------------------------------------------------------------
def od_lut() -> np.ndarray:
    """The 256 distinct values convert_RGB_to_OD can produce (f64)."""
    v = np.arange(256, dtype=np.float64)
    v[0] = 1.0
    return np.maximum(-1.0 * np.log(v / 255.0), 1e-6)


def rgb_to_od(I: np.ndarray) -> np.ndarray:
    """convert_RGB_to_OD, stain_utils.py:101-112: 0 -> 1, max(-ln(I/255), 1e-6)."""
    J = I.copy()
    J[J == 0] = 1
    return np.maximum(-1.0 * np.log(J / 255), 1e-6)


def normalize_rows(A: np.ndarray) -> np.ndarray:
    """normalize_matrix_rows, stain_utils.py:93-99."""
    return A / np.linalg.norm(A, axis=1)[:, None]


# --------------------------------------------------------------------------
# spams.lasso(mode=2, pos=True) for two atoms, stain_utils.py:69-78
# third-party: spams 2.6.2.5 (environment.yml:148); restated as the unique
# optimum of  min_{a>=0} 1/2 ||x - M^T a||^2 + lam * sum(a)
# --------------------------------------------------------------------------
def lasso2_nonneg(OD: np.ndarray, M: np.ndarray, lam: float = 0.01) -> np.ndarray:
    """Exact non-negative lasso for K=2 atoms (rows of M), one problem per row of OD.

    Active-set enumeration of the KKT system; atoms need not be unit norm."""
    OD = np.asarray(OD, dtype=np.float64)
    M = np.asarray(M, dtype=np.float64)
    g11 = M[0] @ M[0]
    g22 = M[1] @ M[1]
    g12 = M[0] @ M[1]
    b1 = OD @ M[0] - lam
    b2 = OD @ M[1] - lam
    det = g11 * g22 - g12 * g12
    a1 = (g22 * b1 - g12 * b2) / det
    a2 = (g11 * b2 - g12 * b1) / det
    both = (a1 >= 0) & (a2 >= 0)
    s1 = b1 / g11                       # only atom 1 active
    s2 = b2 / g22                       # only atom 2 active
    only1 = ~both & (b1 > 0) & (b2 - g12 * s1 <= 0)
    only2 = ~both & ~only1 & (b2 > 0) & (b1 - g12 * s2 <= 0)
    C = np.zeros((OD.shape[0], 2), dtype=np.float64)
    C[both, 0] = a1[both]
    C[both, 1] = a2[both]
    C[only1, 0] = s1[only1]
    C[only2, 1] = s2[only2]
    return C


def lasso_kkt_violation(OD: np.ndarray, M: np.ndarray, C: np.ndarray, lam: float) -> float:
    """Largest KKT residual of C as a solution of the non-negative lasso (certificate)."""
    r = OD - C @ M
    grad = r @ M.T - lam                # must be == 0 where C > 0, <= 0 where C == 0
    act = C > 0
    v_act = np.abs(grad[act]).max() if act.any() else 0.0
    v_in = np.maximum(grad[~act], 0).max() if (~act).any() else 0.0
    v_neg = np.maximum(-C, 0).max()
    return float(max(v_act, v_in, v_neg))


def get_concentrations(I: np.ndarray, M: np.ndarray, regularizer: float = 0.01) -> np.ndarray:
    """get_concentrations, stain_utils.py:69-78 -> (P, 2) f64 (all pixels, incl. background)."""
    OD = rgb_to_od(I).reshape((-1, 3))
    return lasso2_nonneg(OD, M, regularizer)


# --------------------------------------------------------------------------
# Macenko, stainlib/extraction/macenko_stain_extractor.py:7-44
# --------------------------------------------------------------------------
def macenko_stain_matrix(I, luminosity_threshold=0.8, angular_percentile=99, details=None):
    """MacenkoStainExtractor.get_stain_matrix; optionally records intermediates."""
    assert is_uint8_image(I), "Image should be RGB uint8."                       # :16
    mask = tissue_mask(I, luminosity_threshold).reshape((-1,))                   # :18
    OD = rgb_to_od(I).reshape((-1, 3))                                           # :19
    OD = OD[mask]                                                                # :20
    cov = np.cov(OD, rowvar=False)
    _, V = np.linalg.eigh(cov)                                                   # :22
    V = V[:, [2, 1]]                                                             # :24
    if V[0, 0] < 0:
        V[:, 0] *= -1                                                            # :26
    if V[0, 1] < 0:
        V[:, 1] *= -1                                                            # :27
    That = np.dot(OD, V)                                                         # :29
    phi = np.arctan2(That[:, 1], That[:, 0])                                     # :31
    minPhi = np.percentile(phi, 100 - angular_percentile)                        # :33
    maxPhi = np.percentile(phi, angular_percentile)                              # :34
    v1 = np.dot(V, np.array([np.cos(minPhi), np.sin(minPhi)]))                   # :36
    v2 = np.dot(V, np.array([np.cos(maxPhi), np.sin(maxPhi)]))                   # :37
    HE = np.array([v1, v2]) if v1[0] > v2[0] else np.array([v2, v1])             # :40-43
    M = normalize_rows(HE)                                                       # :44
    if details is not None:
        details.update(n_tissue=int(mask.sum()), cov=cov, V=V.copy(), minPhi=float(minPhi),
                       maxPhi=float(maxPhi), mask=mask)
    return M


# --------------------------------------------------------------------------
# Vahadane, stainlib/extraction/vahadane_stain_extractor.py:19-43
# third-party: spams.trainDL (absent).  Objective (spams mode=2, modeD=0,
# posAlpha, posD, K=2):
#   min_{D>=0, ||d_k||<=1} (1/T) sum_i min_{a>=0} 1/2||x_i - D a||^2 + lam*||a||_1
# The reference run is time-budgeted/random; the oracle (and the GPU engine)
# define the target as the converged point of the deterministic full-batch
# block-coordinate scheme below (Mairal et al. 2010, Alg. 2 dictionary update).
# --------------------------------------------------------------------------
def vahadane_init(OD: np.ndarray) -> np.ndarray:
    """Deterministic initial dictionary: the Macenko-style extreme directions are not
    needed; two fixed Ruifrok H&E OD vectors (unit norm) make every run reproducible."""
    D0 = np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]], dtype=np.float64)
    return normalize_rows(D0)


def vahadane_dictionary(OD: np.ndarray, lam: float = 0.1, max_sweeps: int = 200, tol: float = 1e-9,
                        D0: np.ndarray | None = None, info: dict | None = None) -> np.ndarray:
    """Full-batch dictionary learning on tissue OD rows -> D (2,3), rows = atoms.

    One sweep = exact codes for every pixel (lasso2_nonneg) followed by one pass of
    projected block-coordinate descent over the two atoms using A = sum a a^T and
    B = sum x a^T."""
    D = vahadane_init(OD) if D0 is None else np.array(D0, dtype=np.float64)
    sweeps = 0
    for sweeps in range(1, max_sweeps + 1):
        Cc = lasso2_nonneg(OD, D, lam)
        A = Cc.T @ Cc                    # (2,2)
        B = OD.T @ Cc                    # (3,2)
        Dn = D.copy()
        for j in range(2):
            if A[j, j] > 1e-300:
                u = (B[:, j] - Dn.T @ A[:, j]) / A[j, j] + Dn[j]
                u = np.maximum(u, 0.0)
                Dn[j] = u / max(np.linalg.norm(u), 1.0)
        delta = np.abs(Dn - D).max()
        D = Dn
        if delta < tol:
            break
    if info is not None:
        Cc = lasso2_nonneg(OD, D, lam)
        r = OD - Cc @ D
        info.update(sweeps=sweeps,
                    objective=float((0.5 * (r * r).sum(1) + lam * Cc.sum(1)).mean()))
    return D


def vahadane_stain_matrix(I, luminosity_threshold=0.8, regularizer=0.1, max_sweeps=200, tol=1e-9,
                          info=None):
    """VahadaneStainExtractor.get_stain_matrix with trainDL replaced by its converged optimum."""
    assert is_uint8_image(I), "Image should be RGB uint8."                       # :28
    mask = tissue_mask(I, luminosity_threshold).reshape((-1,))                   # :30
    OD = rgb_to_od(I).reshape((-1, 3))[mask]                                     # :31-32
    D = vahadane_dictionary(OD, regularizer, max_sweeps, tol, info=info)         # :35-36
    if D[0, 0] < D[1, 0]:
        D = D[[1, 0], :]                                                         # :40-41
    return normalize_rows(D)                                                     # :43


_EXTRACTORS = {"macenko": macenko_stain_matrix, "vahadane": vahadane_stain_matrix}


# --------------------------------------------------------------------------
# ExtractiveStainNormalizer, stainlib/normalization/normalizer.py:16-50
# --------------------------------------------------------------------------
class ExtractiveStainNormalizer:
    def __init__(self, method):
        if method.lower() not in _EXTRACTORS:
            raise Exception("Method not recognized.")                            # :25
        self.extract = _EXTRACTORS[method.lower()]

    def fit(self, target):
        self.stain_matrix_target = self.extract(target)                          # :34
        self.target_concentrations = get_concentrations(target, self.stain_matrix_target)  # :35
        self.maxC_target = np.percentile(self.target_concentrations, 99, axis=0).reshape((1, 2))  # :36

    def transform(self, I, details=None):
        M_src = self.extract(I)                                                  # :45
        C = get_concentrations(I, M_src)                                         # :46
        maxC_src = np.percentile(C, 99, axis=0).reshape((1, 2))                  # :47
        C = C * (self.maxC_target / maxC_src)                                    # :48
        tmp = 255 * np.exp(-1 * np.dot(C, self.stain_matrix_target))             # :49
        if details is not None:
            details.update(M_src=M_src, maxC_src=maxC_src, prequant=tmp.reshape(I.shape))
        return truncate_u8(tmp).reshape(I.shape)                                 # :50


def truncate_u8(x: np.ndarray) -> np.ndarray:
    """``.astype(np.uint8)`` of normalizer.py:50 -- truncation toward zero, no clip.

    For in-range values this is exactly numpy's cast.  Out-of-range values are
    platform-dependent in numpy; the engine defines them as int32 truncation
    followed by wrap modulo 256, and so does the oracle."""
    return np.trunc(x).astype(np.int64).astype(np.uint8)


# --------------------------------------------------------------------------
# HED colour augmentation, stainlib/augmentation/augmenter.py:276-344
# third-party: scikit-image rgb2hed/hed2rgb; 0.18.3 semantics are pinned
# (colorconv.py:1448-1454 separate_stains, :1511-1518 combine_stains).
# --------------------------------------------------------------------------
RGB_FROM_HED = np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11], [0.27, 0.57, 0.78]])
HED_FROM_RGB = np.linalg.inv(RGB_FROM_HED)


def rgb2hed(rgb_u8_or_float: np.ndarray, mode: str = "0.18") -> np.ndarray:
    x = rgb_u8_or_float
    x = x.astype(np.float64) / 255.0 if x.dtype == np.uint8 else x.astype(np.float64)
    if mode == "0.17":                   # -log10(rgb + 2) @ hed_from_rgb (from memory; unpinned)
        return (-np.log10(x + 2.0)) @ HED_FROM_RGB
    x = np.maximum(x, 1e-6)
    st = (np.log(x) / np.log(1e-6)) @ HED_FROM_RGB
    if mode == "0.19":                   # >=0.19 clamps stains at 0 (from memory; unpinned)
        st = np.maximum(st, 0)
    return st


def hed2rgb(hed: np.ndarray, mode: str = "0.18") -> np.ndarray:
    if mode == "0.17":
        return np.clip(10.0 ** (-(hed @ RGB_FROM_HED)) - 2.0, 0, 1)
    log_rgb = -(hed * (-np.log(1e-6))) @ RGB_FROM_HED
    return np.clip(np.exp(log_rgb), 0, 1)


def hed_transform(patch: np.ndarray, sigmas, biases, cutoff=(0.05, 0.95), mode="0.18"):
    """HedColorAugmenter.transform, augmenter.py:276-331 (uint8 and float inputs)."""
    if patch.dtype.kind == "f":
        patch_mean = np.mean(patch)                                              # :289
    else:
        patch_mean = np.mean(patch.astype(np.float32)) / 255.0                   # :291
    if not (cutoff[0] <= patch_mean <= cutoff[1]):                               # :293
        return patch                                                             # :331 (same object)
    hed = rgb2hed(patch, mode)                                                   # :295
    for c in range(3):                                                           # :298-316
        if sigmas[c] != 0.0:
            hed[:, :, c] *= 1.0 + sigmas[c]
        if biases[c] != 0.0:
            hed[:, :, c] += biases[c]
    rgb = np.clip(hed2rgb(hed, mode), 0.0, 1.0)                                  # :319-320
    if patch.dtype.kind != "f":
        rgb = (rgb * 255.0).astype(np.uint8)                                     # :324-325
    return rgb


def hed_randomize(thresh: float):
    """HedColorAugmenter.randomize, augmenter.py:333-344: six global np.random draws."""
    s = [np.random.uniform(-thresh, thresh) for _ in range(3)]
    b = [np.random.uniform(-thresh, thresh) for _ in range(3)]
    return s, b


# --------------------------------------------------------------------------
# StainAugmentor, stainlib/augmentation/augmenter.py:403-449
# --------------------------------------------------------------------------
class StainAugmentor:
    def __init__(self, method, sigma1=0.2, sigma2=0.2, augment_background=False):
        if method.lower() not in _EXTRACTORS:
            raise Exception("Method not recognized.")                            # :411
        self.extract = _EXTRACTORS[method.lower()]
        self.sigma1, self.sigma2, self.augment_background = sigma1, sigma2, augment_background

    def fit(self, I):
        self.image_shape = I.shape                                               # :422
        self.stain_matrix = self.extract(I)                                      # :423
        self.source_concentrations = get_concentrations(I, self.stain_matrix)    # :424
        self.n_stains = 2
        self.tissue_mask = tissue_mask(I).ravel()                                # :426

    def pop_with(self, alphas, betas):
        C = self.source_concentrations.copy()                                    # :433
        for i in range(2):
            if self.augment_background:
                C[:, i] = C[:, i] * alphas[i] + betas[i]                         # :439-440
            else:
                C[self.tissue_mask, i] = C[self.tissue_mask, i] * alphas[i] + betas[i]  # :442-443
        out = 255 * np.exp(-1 * np.dot(C, self.stain_matrix))                    # :445
        out = out.reshape(self.image_shape)
        return np.clip(out, 0, 255).astype(np.uint8)                             # :447

    def pop(self):
        a, b = [], []
        for _ in range(2):                                                       # :435-437
            a.append(np.random.uniform(1 - self.sigma1, 1 + self.sigma1))
            b.append(np.random.uniform(-self.sigma2, self.sigma2))
        return self.pop_with(a, b)


# --------------------------------------------------------------------------
# GrayscaleAugmentor  (augmentation/augmenter.py:374-401)
# --------------------------------------------------------------------------
def rgb2gray(I: np.ndarray) -> np.ndarray:
    """skimage 0.18 color.rgb2gray on a uint8 image: img_as_float (x * (1/255), binary64) @ [0.2125, 0.7154, 0.0721]."""
    rgb = np.multiply(I[..., :3], 1.0 / 255, dtype=np.float64)
    return rgb @ np.array([0.2125, 0.7154, 0.0721], dtype=np.float64)


class GrayscaleAugmentor:
    def __init__(self, sigma1=0.2, sigma2=0.2, augment_background=False):
        self.sigma1, self.sigma2, self.augment_background = sigma1, sigma2, augment_background   # :376-378 (unused by pop)

    def fit(self, I):
        self.image_shape = I.shape                                               # :386
        self.tissue_mask = tissue_mask(I).ravel()                                # :387 (raises on an empty mask)
        self.image = I

    def pop_with(self, alpha, beta):
        g = np.clip(rgb2gray(self.image) * alpha + beta, 0, 1)                   # :396-397
        g3 = np.stack([g, g, g], axis=2)                                         # :398
        return np.clip(g3 * 255, 0, 255).astype(np.uint8)                        # :399

    def pop(self):
        alpha = np.random.uniform(1 - 0.2, 1 + 0.2)                              # :394 (literal 0.2, not sigma1)
        beta = np.random.uniform(-0.2, 0.2)                                      # :395
        return self.pop_with(alpha, beta)


# --------------------------------------------------------------------------
# Synthetic H&E tiles -- shared by tests and bench
# --------------------------------------------------------------------------
M_TRUE_SRC = np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]])
M_TRUE_TGT = np.array([[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])


def synth_tile(h: int, w: int, seed: int, M_true: np.ndarray = M_TRUE_SRC) -> np.ndarray:
    rng = np.random.RandomState(seed)
    M = normalize_rows(np.asarray(M_true, dtype=np.float64))
    P = h * w
    C = rng.gamma(2.0, 0.35, size=(P, 2))
    bg = rng.rand(P) < 0.2
    C[bg] *= 0.02
    OD = C @ M + rng.normal(0.0, 0.01, size=(P, 3))
    rgb = np.clip(255.0 * np.exp(-OD), 0, 255)
    return rgb.astype(np.uint8).reshape(h, w, 3)
