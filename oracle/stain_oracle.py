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


# --------------------------------------------------------------------------
# OpenCV 8-bit Lab, all three channels and the way back
# third-party: opencv-python 4.4.0.46; call sites stain_utils.py:62,66,152,183
# (cv.cvtColor COLOR_RGB2LAB / COLOR_LAB2RGB on uint8 images).  Restated from
# OpenCV's published modules/imgproc/src/color_lab.cpp: RGB2Lab_b (integer
# tables above) and Lab2RGBinteger (LabToYF_b, abToXZ_b, sRGBInvGammaTab_b).
# OpenCV builds these tables with its own binary32 soft-float (softfloat pow /
# cbrt); here the power laws are evaluated in binary64 and rounded to binary32,
# which can differ from OpenCV by one unit in isolated table entries.
# PARITY UNPINNED like the L channel: tools/pin_cv2.py checks every function
# below against a real cv2 over all 2^24 colours wherever OpenCV is installed.
# --------------------------------------------------------------------------
_f32 = np.float32
_D65 = (0.950456, 1.0, 1.088754)
_SRGB2XYZ = ((0.412453, 0.357580, 0.180423), (0.212671, 0.715160, 0.072169), (0.019334, 0.119193, 0.950227))
_XYZ2SRGB = ((3.240479, -1.53715, -0.498535), (-0.969256, 1.875991, 0.041556), (0.055648, -0.204043, 1.057311))
# RGB2Lab_b: coeffs[i][j] = cvRound((1 << lab_shift) * sRGB2XYZ[i][j] / whitePt[i])
LAB_FWD_COEFFS = np.array([[int(np.rint(4096.0 * c / _D65[i])) for c in row] for i, row in enumerate(_SRGB2XYZ)], dtype=np.int64)
# Lab2RGBinteger: coeffs[i][j] = cvRound((1 << lab_shift) * XYZ2sRGB[i][j] * whitePt[j])
LAB_INV_COEFFS = np.array([[int(np.rint(4096.0 * c * _D65[j])) for j, c in enumerate(row)] for row in _XYZ2SRGB], dtype=np.int64)
assert LAB_FWD_COEFFS.tolist() == [[1777, 1541, 778], [871, 2929, 296], [73, 448, 3575]]
assert LAB_INV_COEFFS.tolist() == [[12615, -6296, -2223], [-3773, 7684, 185], [217, -836, 4715]]
_LAB_BASE_SHIFT = 14
_LAB_BASE = 1 << _LAB_BASE_SHIFT
_INV_GAMMA_SHIFT = 12
_MIN_AB = -8145


def _lab_to_yf_tab_b() -> np.ndarray:
    """OpenCV ``LabToYF_b``: for L8 = 0..255 the pair (Y, f(Y)) scaled by 2^14, binary32 arithmetic."""
    out = np.zeros((256, 2), dtype=np.int64)
    base = _LAB_BASE
    for i in range(256):
        if i <= 20:                                                   # 8 * 255 / 100 == 20.4: the linear segment
            y = np.rint(_f32(i * base * 20 * 9) / _f32(17 * 24389))
            ify = np.rint(_f32(base) * (_f32(16) / _f32(116) + _f32(i * 5) / _f32(3 * 17 * 29)))
        else:
            fy = _f32(i * 100 * base) / _f32(255 * 116) + _f32(16 * base) / _f32(116)
            ify = np.rint(fy)
            y = np.rint(fy * fy * fy / _f32(base * base))
        out[i] = (int(y), int(ify))
    return out


def _ab_to_xz_tab_b() -> np.ndarray:
    """OpenCV ``abToXZ_b``: f^-1 on the 2^14 scale for arguments _MIN_AB .. _MIN_AB + 9 * 2^14 / 4 - 1, C integer arithmetic
    (division truncates toward zero)."""
    n = _LAB_BASE * 9 // 4
    i = np.arange(_MIN_AB, _MIN_AB + n, dtype=np.int64)
    tdiv = lambda a, b: np.sign(a) * (np.abs(a) // b)                 # noqa: E731  C-style truncation
    lin = tdiv(i * 108, 841) - (_LAB_BASE * 16 // 116 * 108 // 841)
    cub = (i * i // _LAB_BASE) * i // _LAB_BASE                       # only used for i > 3390 > 0
    return np.where(i <= 3390, lin, cub)


def _srgb_inv_gamma_tab_b() -> np.ndarray:
    """OpenCV ``sRGBInvGammaTab_b``: round(255 * sRGB-gamma(i / 4095)), 4096 entries."""
    x = _f32(1.0) * np.arange(1 << _INV_GAMMA_SHIFT, dtype=np.float32) / _f32((1 << _INV_GAMMA_SHIFT) - 1)
    thr = _f32(7827) / _f32(2500000)                                  # 0.0031308
    low = _f32(323) / _f32(25)                                        # 12.92
    xs = _f32(11) / _f32(200)                                         # 0.055
    p = (x.astype(np.float64) ** (1.0 / (_f32(12) / _f32(5)).astype(np.float64))).astype(np.float32)
    y = np.where(x <= thr, x * low, p * (_f32(1) + xs) - xs).astype(np.float32)
    return np.rint(_f32(255) * y).astype(np.int64)


LAB_TO_YF_TAB = _lab_to_yf_tab_b()
AB_TO_XZ_TAB = _ab_to_xz_tab_b()
SRGB_INV_GAMMA_TAB = _srgb_inv_gamma_tab_b()
assert AB_TO_XZ_TAB.min() == -1335 and AB_TO_XZ_TAB.max() == 88231     # the bounds OpenCV's source comments state


def rgb2lab_u8(I: np.ndarray) -> np.ndarray:
    """``cv2.cvtColor(I, cv2.COLOR_RGB2LAB)`` for uint8 RGB (OpenCV RGB2Lab_b): L*255/100, a+128, b+128, saturated."""
    g = SRGB_GAMMA_TAB
    R, G, B = g[I[..., 0]], g[I[..., 1]], g[I[..., 2]]
    half = 1 << (_LAB_SHIFT - 1)
    C = LAB_FWD_COEFFS
    fX = LAB_CBRT_TAB[(R * C[0, 0] + G * C[0, 1] + B * C[0, 2] + half) >> _LAB_SHIFT]
    fY = LAB_CBRT_TAB[(R * C[1, 0] + G * C[1, 1] + B * C[1, 2] + half) >> _LAB_SHIFT]
    fZ = LAB_CBRT_TAB[(R * C[2, 0] + G * C[2, 1] + B * C[2, 2] + half) >> _LAB_SHIFT]
    h2 = 1 << (_LAB_SHIFT2 - 1)
    L = (_LSCALE * fY + _LSHIFT + h2) >> _LAB_SHIFT2
    a = (500 * (fX - fY) + 128 * (1 << _LAB_SHIFT2) + h2) >> _LAB_SHIFT2
    b = (200 * (fY - fZ) + 128 * (1 << _LAB_SHIFT2) + h2) >> _LAB_SHIFT2
    return np.clip(np.stack([L, a, b], axis=-1), 0, 255).astype(np.uint8)


def lab2rgb_u8(LAB: np.ndarray) -> np.ndarray:
    """``cv2.cvtColor(LAB, cv2.COLOR_LAB2RGB)`` for uint8 Lab (OpenCV Lab2RGBinteger::process)."""
    LL = LAB[..., 0].astype(np.int64)
    aa = LAB[..., 1].astype(np.int64)
    bb = LAB[..., 2].astype(np.int64)
    y = LAB_TO_YF_TAB[LL, 0]
    ify = LAB_TO_YF_TAB[LL, 1]
    adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * _LAB_BASE // 500
    bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * _LAB_BASE // 200 + 1
    x = AB_TO_XZ_TAB[ify + adiv - _MIN_AB]
    z = AB_TO_XZ_TAB[ify - bdiv - _MIN_AB]
    shift = _LAB_SHIFT + (_LAB_BASE_SHIFT - _INV_GAMMA_SHIFT)
    C = LAB_INV_COEFFS
    top = (1 << _INV_GAMMA_SHIFT) - 1
    chans = []
    for r in range(3):
        v = (C[r, 0] * x + C[r, 1] * y + C[r, 2] * z + (1 << (shift - 1))) >> shift
        chans.append(SRGB_INV_GAMMA_TAB[np.clip(v, 0, top)])
    return np.clip(np.stack(chans, axis=-1), 0, 255).astype(np.uint8)


def mean_std_dev(x: np.ndarray):
    """``cv2.meanStdDev`` of a single-channel array: double sums, population variance clamped at 0 -> two (1,1) float64."""
    v = np.asarray(x, dtype=np.float64).ravel()
    n = v.size
    mean = v.sum() / n
    var = max((v * v).sum() / n - mean * mean, 0.0)
    return np.array([[mean]]), np.array([[np.sqrt(var)]])


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


# --------------------------------------------------------------------------
# LAB helpers, ReinhardStainNormalizer, LuminosityStandardizer
# stainlib/utils/stain_utils.py:50-67,114-124,146-194; normalization/normalizer.py:54-94
# All of these sit on cv2's 8-bit Lab (rgb2lab_u8 / lab2rgb_u8 above): PARITY UNPINNED.
# --------------------------------------------------------------------------
def od_to_rgb(OD: np.ndarray) -> np.ndarray:
    """convert_OD_to_RGB, stain_utils.py:114-124."""
    assert OD.min() >= 0, "Negative optical density."                            # :122
    OD = np.maximum(OD, 1e-6)                                                    # :123
    return (255 * np.exp(-1 * OD)).astype(np.uint8)                              # :124


def standardize_brightness(I: np.ndarray) -> np.ndarray:
    """standardize_brightness, stain_utils.py:188-194: divide by the 90th percentile of ALL byte values."""
    p = np.percentile(I, 90)                                                     # :193
    return np.clip(I * 255.0 / p, 0, 255).astype(np.uint8)                       # :194


def lab_split(I: np.ndarray):
    """lab_split, stain_utils.py:146-158: binary32 channels L/2.55, a-128, b-128."""
    lab = rgb2lab_u8(I).astype(np.float32)                                       # :152-153
    I1, I2, I3 = lab[..., 0].copy(), lab[..., 1].copy(), lab[..., 2].copy()      # :154 (cv.split)
    I1 /= 2.55                                                                   # :155
    I2 -= 128.0                                                                  # :156
    I3 -= 128.0                                                                  # :157
    return I1, I2, I3


def merge_back(I1, I2, I3) -> np.ndarray:
    """merge_back, stain_utils.py:160-172 (operates on copies; the reference scales its arguments in place)."""
    I1 = I1 * 2.55                                                               # :168
    I2 = I2 + 128.0                                                              # :169
    I3 = I3 + 128.0                                                              # :170
    lab = np.clip(np.stack([I1, I2, I3], axis=-1), 0, 255).astype(np.uint8)      # :171 (cv.merge, clip, truncate)
    return lab2rgb_u8(lab)                                                       # :172


def get_mean_std(I: np.ndarray):
    """get_mean_std, stain_utils.py:174-186: cv.meanStdDev per Lab channel -> ((m1,m2,m3), (sd1,sd2,sd3)), (1,1) float64 each."""
    chans = lab_split(I)
    ms = [mean_std_dev(c) for c in chans]
    return tuple(m for m, _ in ms), tuple(sd for _, sd in ms)


class ReinhardStainNormalizer:
    """normalization/normalizer.py:54-94."""

    def __init__(self, target_means=0, target_stds=0):
        self.target_means, self.target_stds = target_means, target_stds          # :61-62

    def fit(self, target):
        target = standardize_brightness(target)                                  # :65
        self.target_means, self.target_stds = get_mean_std(target)               # :66-68

    def transform(self, I, mask_background=False, luminosity_threshold=0.8):
        I = standardize_brightness(I)                                            # :78
        I1, I2, I3 = lab_split(I)                                                # :79
        means, stds = get_mean_std(I)                                            # :80
        norm1 = ((I1 - means[0]) * (self.target_stds[0] / stds[0])) + self.target_means[0]   # :81 (binary64 from here)
        norm2 = ((I2 - means[1]) * (self.target_stds[1] / stds[1])) + self.target_means[1]   # :82
        norm3 = ((I3 - means[2]) * (self.target_stds[2] / stds[2])) + self.target_means[2]   # :83
        if mask_background:
            mask = tissue_mask(I, luminosity_threshold)                          # :86-87
            background = np.array(~mask * 254).astype(np.uint8)                  # :88
            norm1, norm2, norm3 = np.multiply(mask, norm1), np.multiply(mask, norm2), np.multiply(mask, norm3)  # :89
            return merge_back(background + norm1, norm2, norm3)                  # :90
        return merge_back(norm1, norm2, norm3)                                   # :92


def luminosity_standardize(I: np.ndarray, percentile=95) -> np.ndarray:
    """LuminosityStandardizer.standardize, stain_utils.py:52-67."""
    assert is_uint8_image(I), "Image should be RGB uint8."                       # :61
    lab = rgb2lab_u8(I)                                                          # :62
    L_float = lab[:, :, 0].astype(float)                                         # :63
    p = np.percentile(L_float, percentile)                                       # :64
    lab[:, :, 0] = np.clip(255 * L_float / p, 0, 255).astype(np.uint8)           # :65
    return lab2rgb_u8(lab)                                                       # :66


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
    if mode == "0.17":                   # presumed <= 0.17: -ln(rgb + 2) @ hed_from_rgb (from memory; unpinned)
        return (-np.log(x + 2.0)) @ HED_FROM_RGB
    if mode == "experimental_log10":     # the same with base-10 logarithms (round 1's reading; unpinned)
        return (-np.log10(x + 2.0)) @ HED_FROM_RGB
    x = np.maximum(x, 1e-6)
    st = (np.log(x) / np.log(1e-6)) @ HED_FROM_RGB
    if mode == "0.19":                   # >=0.19 clamps stains at 0 (from memory; unpinned)
        st = np.maximum(st, 0)
    return st


def hed2rgb(hed: np.ndarray, mode: str = "0.18") -> np.ndarray:
    if mode == "0.17":                   # exp(.) - 2, rescale_intensity(in_range=(-1, 1)) = clip to [-1, 1]; the caller clips to [0, 1]
        return np.clip(np.exp(-(hed @ RGB_FROM_HED)) - 2.0, 0, 1)
    if mode == "experimental_log10":
        return np.clip(10.0 ** (-(hed @ RGB_FROM_HED)) - 2.0, 0, 1)
    log_rgb = -(hed * (-np.log(1e-6))) @ RGB_FROM_HED
    return np.clip(np.exp(log_rgb), 0, 1)


def hed_transform(patch: np.ndarray, sigmas, biases, cutoff=(0.05, 0.95), mode="0.18", details=None):
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
        if details is not None:
            details.update(prequant=rgb * 255.0)                                 # (tests: the values the cast truncates)
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

    def pop_with(self, alphas, betas, details=None):
        C = self.source_concentrations.copy()                                    # :433
        for i in range(2):
            if self.augment_background:
                C[:, i] = C[:, i] * alphas[i] + betas[i]                         # :439-440
            else:
                C[self.tissue_mask, i] = C[self.tissue_mask, i] * alphas[i] + betas[i]  # :442-443
        out = 255 * np.exp(-1 * np.dot(C, self.stain_matrix))                    # :445
        out = out.reshape(self.image_shape)
        if details is not None:
            details.update(prequant=np.clip(out, 0, 255))                        # (tests: the values the cast truncates)
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


def structured_tile(kind: str, h: int, w: int, seed: int) -> np.ndarray:
    """Synthetic tiles WITH spatial structure and colour ties (the i.i.d. generator above has neither):
      white_bg   saturated (255,255,255) background: a band on the left and a disc, ~35 % of the pixels
      palette12  11 tissue colours + white in 4x4 blocks (heavy ties in every order statistic)
      quantized  colours snapped to multiples of 4 plus 2 (JPEG-like ties), white band on top
      blobs      smooth spatial structure: haematoxylin-rich nuclei (soft-edged discs) on an eosin background that varies
                 slowly across the tile, a white lumen, little pixel noise -- neighbouring pixels are strongly correlated,
                 as in a real slide (what a stratified pixel sample has to cope with)"""
    if kind == "blobs":
        rng = np.random.RandomState(seed + 104729)
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        s = float(min(h, w))
        cH = np.full((h, w), 0.08)
        for _ in range(60):                                            # nuclei
            cy, cx, r, a = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(0.015, 0.045) * s, rng.uniform(0.9, 2.2)
            cH += a / (1.0 + np.exp((np.hypot(yy - cy, xx - cx) - r) / (0.12 * r + 0.5)))
        cE = 0.55 + 0.35 * np.sin(2 * np.pi * (yy / h + 0.3)) * np.cos(2 * np.pi * (0.7 * xx / w + 0.1)) + 0.25 * (xx / w)
        cE = np.clip(cE, 0.05, None) * (1.0 - 0.5 * np.clip(cH / 2.0, 0, 1))
        lumen = ((yy - 0.3 * h) / (0.18 * h)) ** 2 + ((xx - 0.7 * w) / (0.12 * w)) ** 2 < 1.0
        C = np.stack([cH, cE], axis=-1).reshape(-1, 2)
        C[lumen.ravel()] *= 0.01
        OD = C @ normalize_rows(np.asarray(M_TRUE_SRC, dtype=np.float64)) + rng.normal(0.0, 0.004, size=(h * w, 3))
        return np.clip(255.0 * np.exp(-OD), 0, 255).astype(np.uint8).reshape(h, w, 3)
    base = synth_tile(h, w, seed)
    rng = np.random.RandomState(seed + 7919)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "white_bg":
        out = base.copy()
        hole = (xx < w // 5) | ((yy - 0.6 * h) ** 2 + (xx - 0.65 * w) ** 2 < (0.22 * min(h, w)) ** 2)
        out[hole] = 255
        return out
    if kind == "palette12":
        flat = base.reshape(-1, 3)
        tissue = flat[tissue_mask(base).ravel()]
        pal = np.concatenate([tissue[:: max(1, len(tissue) // 11)][:11], np.array([[255, 255, 255]], np.uint8)])
        idx = rng.choice(len(pal), size=((h + 3) // 4, (w + 3) // 4), p=np.r_[np.full(len(pal) - 1, 0.8 / (len(pal) - 1)), 0.2])
        return pal[np.kron(idx, np.ones((4, 4), dtype=np.int64))[:h, :w]].astype(np.uint8)
    if kind == "quantized":
        out = (base & 0xFC) | 2
        out[: h // 8] = 255
        return out.astype(np.uint8)
    raise ValueError(kind)
