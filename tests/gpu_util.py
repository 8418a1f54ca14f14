"""Helpers for the -m gpu parity tests (oracle = checker, HIP engine = thing under test)."""
import numpy as np
import torch

from oracle import stain_oracle as so


def to_dev(tiles):
    """list/array of HxWx3 uint8 -> (N,H,W,3) cuda tensor"""
    a = np.stack(tiles) if isinstance(tiles, (list, tuple)) else tiles
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def u8_parity(got: np.ndarray, want: np.ndarray, max_flips=None, label="", src=None, prequant=None):
    """The stated uint8 bar (hard part 2; north_star: 1e-4 on reconstructed RGB): every byte within 1 of the
    reference's, and at most max(4, 1e-4 N + 3 sqrt(1e-4 N)) of the N bytes different at all -- a COUNT, so that small tiles are not
    judged by a rate one byte already exceeds: the ~1e-7 error of a tile's (M, maxC) moves ALL its pixels together, so the
    flips of a small tile are a handful or none (measured, tools/small_tile_errors.py over 300 random small tiles: at most 6, none above
    this bar; 18 of 3.1 M bytes at 1024^2).  Bit identity is impossible in binary32: the reference truncates.
    With ``src`` (the input tile) AND content of few distinct colours (JPEG-like quantisation, palettes, real tissue: at least two pixels per
    distinct colour on average) all pixels of a colour move together, so ONE colour whose exact value sits within 1e-6 of an integer
    flips hundreds of bytes at once; there, and only there, the bar is on the distinct input colours among the flipped pixels (the
    same 1e-4, at least 4) instead of on the bytes.
    With ``prequant`` (the oracle's values BEFORE its truncating cast) a count above the bar is tolerated only where it can be a matter
    of the truncation alone: every differing byte's oracle value must lie within 1e-5 relative of the integer the two truncations
    disagree about (the kernel delivers 1.5e-6 end to end; the north star's own figure, 1e-4, would pass a hundredfold regression), AND
    the count stays below ten times the bar (round-4 advisor: a systematic bias must not hide behind the tolerance).
    Prints what was measured; returns the mismatch rate."""
    d = got.astype(np.int16) - want.astype(np.int16)
    # a truncating cast may also wrap 255<->0 only if values exceed 255, which H&E never does
    assert np.abs(d).max() <= 1, f"max |delta| = {np.abs(d).max()}"
    flips, n = int((d != 0).sum()), d.size
    # a COUNT consistent with the rate 1e-4: its expectation plus three standard deviations of a Poisson count, at least 4
    bound = max(4, int(1e-4 * n + 3.0 * (1e-4 * n) ** 0.5)) if max_flips is None else max_flips
    print(f"u8 parity {label}: {flips} of {n} bytes differ (rate {flips / n:.2e}, bound {bound})")
    if flips > bound and prequant is not None:
        v = np.asarray(prequant, dtype=np.float64).reshape(d.shape)[d != 0]
        dist = np.abs(v - np.rint(v))
        worst = float((dist / np.maximum(np.abs(v), 1.0)).max())
        print(f"          above the count bar: every differing byte within {worst:.1e} relative of an integer boundary (allowed: 1e-5)")
        assert worst <= 1e-5, f"{flips} of {n} bytes differ (> {bound}), one of them {worst:.2e} relative from the integer boundary"
        assert flips <= 10 * bound, f"{flips} of {n} bytes differ: more than ten times the bar ({bound})"
        return flips / n
    if flips > bound and src is not None:
        px = (d.reshape(-1, 3) != 0).any(axis=1)
        key = src.reshape(-1, 3).astype(np.int64) @ np.array([65536, 256, 1])
        flipped, total = len(np.unique(key[px])), len(np.unique(key))
        # (round-3 review: the colour-based bar is for content whose colours REPEAT only -- at least two pixels per distinct colour on
        # average: quantised tiles, palettes, real tissue (the ihc fixture: ~3); i.i.d. synthetic tiles (~1.4 at 1024^2) and anything
        # else are held to the byte count)
        assert 2 * total <= key.size, f"{flips} of {n} bytes differ (> {bound}) on content with {total} distinct colours in {key.size} pixels"
        cbound = max(4, int(1e-4 * total))
        print(f"          correlated flips: {flipped} of {total} distinct input colours (bound {cbound})")
        assert flipped <= cbound, f"{flips} of {n} bytes differ, from {flipped} of {total} distinct colours (> {cbound})"
        return flips / n
    assert flips <= bound, f"{flips} of {n} bytes differ (> {bound})"
    return flips / n


def oracle_fit_tile(I):
    M = so.macenko_stain_matrix(I)
    C = so.get_concentrations(I, M)
    return M, np.percentile(C, 99, axis=0)
