"""How much of the tissue mask (SURVEY 8 row a-B, stain_utils.py:41-44: cv2.cvtColor(I, COLOR_RGB2LAB)[:, :, 0] / 255.0 < threshold)
hangs on the part of OpenCV that cannot be checked here (no cv2 on either box): the two tables of its integer RGB2Lab_b path are
built once with OpenCV's own binary32 soft-float (sRGBGammaTab_b[256], LabCbrtTab_b[3072]) and rounded to integers; the oracle
builds them through binary64.  Everything after the tables is integer arithmetic and identical by construction.  So the mask can
only differ where a table ENTRY rounds the other way, i.e. where its pre-rounding value lies within the construction error of a
rounding boundary -- and, for the cube-root table, only if that entry also straddles the L8 value at which `L8 / 255.0 < threshold`
flips.  This test computes those margins (round-3 review, item 6).  Error model: OpenCV's entry is the correctly rounded
value up to ERR_ULPS binary32 units in the last place of the pre-rounding value (one for x = i * fl(1/255) against fl(i / 255), the
power law in binary64, its conversion to binary32, one multiplication: 2-4 in all; 6 is generous).

Result: at thresholds 0.6 / 0.8 / 0.9 the decisive cube-root entries sit 0.96 ... 11 table units from the rounding boundary that
matters (6 ulps are 0.012 units there), and every gamma entry is more than 11 ulps from its boundary -- the default mask does not
depend on OpenCV's soft-float ulps.  Least margin: sRGBGammaTab_b[217] (1415.4986, 0.0014 = 11 ulps from 1415.5), then [249]
(44 ulps); entry 12 is closest in absolute terms (7.6e-5) but that is 159 ulps of its value 7.5.  The FULL Lab conversions (rows f-3 /
f-4) are another matter: the cube-root table as a whole does change with the way x is formed (checked below), so those stay
"parity unpinned" until tools/pin_cv2.py has run against a real cv2."""
from fractions import Fraction

import numpy as np

from oracle import stain_oracle as so

ERR_ULPS = 6.0


def _gamma_pre_rounding(variant):
    i = np.arange(256, dtype=np.float32)
    x = (i / np.float32(255)) if variant == 0 else (i * (np.float32(1) / np.float32(255)))
    x = x.astype(np.float64)
    return np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4) * 2040.0


def _cbrt_pre_rounding(variant):
    i = np.arange(3072, dtype=np.float32)
    x = (i * (np.float32(1) / np.float32(2040))) if variant == 0 else (i / np.float32(2040))
    x = x.astype(np.float64)
    return np.where(x < 216.0 / 24389.0, x * (841.0 / 108.0) + 16.0 / 116.0, np.cbrt(x)) * 32768.0


def test_every_gamma_entry_is_many_ulps_from_its_rounding_boundary():
    worst = []
    for variant in (0, 1):
        g = _gamma_pre_rounding(variant)
        assert np.array_equal(np.rint(g).astype(np.int64), so.SRGB_GAMMA_TAB)            # both ways of forming x give the oracle's table
        dist = np.abs(g - np.floor(g) - 0.5)                                             # to the nearest x.5
        ulp = np.spacing(np.maximum(g, 1e-3).astype(np.float32)).astype(np.float64)
        d_ulps = dist / ulp
        k = int(np.argmin(d_ulps[1:])) + 1                                               # entry 0 is exactly 0
        worst.append((k, float(d_ulps[k])))
        assert d_ulps[1:].min() > ERR_ULPS, (k, g[k], d_ulps[k])
    assert [k for k, _ in worst] == [217, 217] and all(10.0 < d < 14.0 for _, d in worst), worst   # the entry the docs name


def test_the_mask_decision_does_not_sit_on_a_cube_root_rounding_boundary():
    margins = {}
    for thr in (0.6, 0.8, 0.9):
        L = np.arange(256)
        l_max = int(L[(L / 255.0) < thr].max())                                          # largest L8 that is tissue
        # L8 = (296 fY - 1336934 + 16384) >> 15: the smallest REAL fY that gives l_max + 1, and the integer entry that reaches it
        decisive = Fraction((l_max + 1) * 32768 + 1336934 - 16384, 296)
        flip_at = int(np.ceil(float(decisive)))
        boundary = flip_at - 0.5                                                         # pre-rounding values above it round to >= flip_at
        for variant in (0, 1):
            f = _cbrt_pre_rounding(variant)
            tab = np.rint(f).astype(np.int64)
            idx = int(np.nonzero(tab < flip_at)[0].max())                                # last tissue index: entries are monotone
            assert idx == so.y_index_threshold(thr)
            assert tab[idx] < flip_at <= tab[idx + 1]
            below, above = boundary - f[idx], f[idx + 1] - boundary
            err = ERR_ULPS * float(np.spacing(np.float32(f[idx + 1])))
            assert below > 20 * err and above > 20 * err, (thr, variant, below, above, err)
            margins[(thr, variant)] = (round(float(below), 2), round(float(above), 2))
    assert margins[(0.8, 0)] == (6.9, 0.96), margins                                      # the numbers quoted in DESIGN section 2
    # ... whereas the table as a whole is NOT indifferent to how x is formed (rows f-3 / f-4 stay unpinned):
    assert not np.array_equal(np.rint(_cbrt_pre_rounding(0)), np.rint(_cbrt_pre_rounding(1)))
    assert np.array_equal(np.rint(_cbrt_pre_rounding(0)).astype(np.int64), so.LAB_CBRT_TAB)


def test_a_one_unit_change_of_the_weakest_gamma_entry_moves_few_colours():
    """What would be at stake if sRGBGammaTab_b[217] did come out as 1416: the colours with a byte 217 whose luminance index sits
    on the threshold -- counted exactly over all (g, b) for r = 217 and the two other placements: 159 of the 16.7 M colours."""
    thr_idx = so.y_index_threshold(0.8)
    g0 = so.SRGB_GAMMA_TAB.copy()
    g1 = g0.copy()
    g1[217] += 1
    v = np.arange(256)
    flips = 0
    for ch in range(3):
        w = so._CY
        others = [c for c in range(3) if c != ch]
        A, B = np.meshgrid(v, v, indexing="ij")
        def idx(g):
            s = g[217] * w[ch] + g[A] * w[others[0]] + g[B] * w[others[1]] + 2048
            return s >> 12
        flips += int(((idx(g0) <= thr_idx) != (idx(g1) <= thr_idx)).sum())
    assert 0 < flips < 400, flips                                  # of 3 x 65 536 colours with one byte at 217 (16.7 M colours in all)
