"""The mathematics of the colour-cube pre-filter (stainlib_amd/csrc/stats_cube.hpp), restated in numpy and checked on the CPU: for
every cell of the 32^3 colour cube the separable bounds must be CONSERVATIVE -- a cell called plain may not hold a single colour the
per-pixel test of the merged selection sweep would flag -- whatever the signs of the bracket bounds (the cone test is piecewise
linear: its lower / upper bounds switch between min and max of the two half-plane functionals with the sign of the bound) and of the
eigenvector components.  All 2^24 colours are enumerated.  (The HIP code itself is held to byte identity with the per-pixel sweep by
tests/test_gpu_macenko.py::test_prefilter_*; this test is about the bounds' derivation.)"""
import numpy as np
import pytest

from oracle import stain_oracle as so

OD = so.rgb_to_od(np.arange(256, dtype=np.uint8).reshape(1, 256, 1).repeat(3, axis=2))[0, :, 0]
GAM = so.SRGB_GAMMA_TAB.astype(np.float64)
YLIMF = ((so.y_index_threshold(0.8) + 1) << 12) - 2048
MARGIN = 2e-5


def _cells(V, hi0, lo1, u, kt, eps, thr):
    """plain[r5, g5, b5] from the per-channel tables, as cube_tables / cube_mask form them (binary64 here: no rounding allowance)."""
    k = np.arange(32)
    od_lo, od_hi, gam_lo = OD[8 * k + 7], OD[8 * k], GAM[8 * k]

    def box(fn, coef):
        t = [fn(coef[c] * od_lo, coef[c] * od_hi) for c in range(3)]
        return t[0][:, None, None] + t[1][None, :, None] + t[2][None, None, :]
    hi0m, lo1m = hi0 + 2 * MARGIN, lo1 - 2 * MARGIN
    v0, v1 = V[:, 0], V[:, 1]
    lum = 871 * gam_lo[:, None, None] + 2929 * gam_lo[None, :, None] + 296 * gam_lo[None, None, :]
    xmin = box(np.minimum, v0)
    Lp, Lm = (1 - hi0m) * v1 - hi0m * v0, (1 + hi0m) * v1 - hi0m * v0
    Up, Um = (1 - lo1m) * v1 - lo1m * v0, (1 + lo1m) * v1 - lo1m * v0
    t0 = (np.minimum if hi0m >= 0 else np.maximum)(box(np.minimum, Lp), box(np.minimum, Lm))
    t1 = (np.minimum if lo1m >= 0 else np.maximum)(box(np.maximum, Up), box(np.maximum, Um))
    cone = (xmin > 0) & (t0 > 0) & (t1 < 0)
    a = [u[i][0] * v0 + u[i][1] * v1 for i in range(2)]
    amin = [box(np.minimum, a[i]) + kt[i] for i in range(2)]
    amax = [box(np.maximum, a[i]) + kt[i] for i in range(2)]
    sa = sum(np.maximum(np.abs(amin[i]), np.abs(amax[i])) for i in range(2))
    conc = (amax[0] + eps[0] * sa < thr[0]) & (amax[1] + eps[1] * sa < thr[1])
    return ((lum >= YLIMF) | cone) & conc


def _flagged_all_colours(V, hi0, lo1, u, kt, eps, thr):
    """The per-pixel test of select_sweep<kStageMerged> (the variant with the tissue test) for all 2^24 colours, by blocks of red."""
    g, b = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    og, ob = OD[g], OD[b]
    lum_gb = 2929 * GAM[g] + 296 * GAM[b]
    out = np.zeros((256, 256, 256), bool)
    for r in range(256):
        x = V[0, 0] * OD[r] + V[1, 0] * og + V[2, 0] * ob
        y = V[0, 1] * OD[r] + V[1, 1] * og + V[2, 1] * ob
        d = x + np.abs(y)
        pp = (x > 0) & (y - (hi0 + MARGIN) * d > 0) & (y - (lo1 - MARGIN) * d < 0)
        tissue = 871 * GAM[r] + lum_gb < YLIMF
        a1 = u[0][0] * x + u[0][1] * y + kt[0]
        a2 = u[1][0] * x + u[1][1] * y + kt[1]
        s = np.abs(a1) + np.abs(a2)
        out[r] = (tissue & ~pp) | (a1 + eps[0] * s >= thr[0]) | (a2 + eps[1] * s >= thr[1])
    return out


def _setup(I):
    det = {}
    M = so.macenko_stain_matrix(I, details=det)
    V = det["V"]
    Gi = np.linalg.inv(M @ M.T)
    W, k = Gi @ M, -0.01 * Gi.sum(axis=1)
    # a = W od + k written on the projections t = V^T od (W's rows lie in the plane of V)
    u = [np.linalg.lstsq(V, W[i], rcond=None)[0] for i in range(2)]
    return V, u, k


@pytest.mark.parametrize("case", ["typical", "both_bounds_positive", "both_bounds_negative", "wide_open", "tight_conc"])
def test_a_plain_cell_holds_no_colour_the_per_pixel_test_flags(case):
    V, u, kt = _setup(so.synth_tile(128, 128, 7))
    hi0, lo1 = {"typical": (-0.35, 0.40), "both_bounds_positive": (0.05, 0.45), "both_bounds_negative": (-0.5, -0.04),
                "wide_open": (-0.9, 0.9), "tight_conc": (-0.35, 0.40)}[case]
    eps = (0.02, 0.03)
    thr = (0.6, 0.5) if case == "tight_conc" else (2.2, 1.9)
    plain = _cells(V, hi0, lo1, u, kt, eps, thr)
    flagged = _flagged_all_colours(V, hi0, lo1, u, kt, eps, thr)
    per_cell = flagged.reshape(32, 8, 32, 8, 32, 8).any(axis=(1, 3, 5))
    assert not (plain & per_cell).any(), f"{int((plain & per_cell).sum())} cells called plain hold a flagged colour"
    # ... and the bounds are not vacuous: a fair share of the cells without any flagged colour is recognised
    clean = ~per_cell
    assert plain.sum() >= 0.5 * clean.sum(), (int(plain.sum()), int(clean.sum()))


def test_with_a_negative_eigenvector_component_and_real_tissue_statistics():
    I = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "tissue_ihc_512.npz"))["input"][:256, :256]
    V, u, kt = _setup(I)
    V2 = V.copy()
    V2[2, 0] = -abs(V2[2, 0]) - 0.05                     # force a negative component of the first eigenvector (x can be <= 0)
    for Vx in (V, V2):
        plain = _cells(Vx, -0.3, 0.35, u, kt, (0.03, 0.03), (1.5, 1.2))
        per_cell = _flagged_all_colours(Vx, -0.3, 0.35, u, kt, (0.03, 0.03), (1.5, 1.2)).reshape(32, 8, 32, 8, 32, 8).any(axis=(1, 3, 5))
        assert not (plain & per_cell).any()
