"""Helpers for the -m gpu parity tests (oracle = checker, HIP engine = thing under test)."""
import numpy as np
import torch

from oracle import stain_oracle as so


def to_dev(tiles):
    """list/array of HxWx3 uint8 -> (N,H,W,3) cuda tensor"""
    a = np.stack(tiles) if isinstance(tiles, (list, tuple)) else tiles
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def u8_parity(got: np.ndarray, want: np.ndarray, max_rate=1e-4):
    """The stated uint8 bar (hard part 2): |delta| <= 1 and mismatch rate <= 1e-4."""
    d = got.astype(np.int16) - want.astype(np.int16)
    # a truncating cast may also wrap 255<->0 only if values exceed 255, which H&E never does
    assert np.abs(d).max() <= 1, f"max |delta| = {np.abs(d).max()}"
    rate = float((d != 0).mean())
    assert rate <= max_rate, f"uint8 mismatch rate {rate:.3e} > {max_rate}"
    return rate


def oracle_fit_tile(I):
    M = so.macenko_stain_matrix(I)
    C = so.get_concentrations(I, M)
    return M, np.percentile(C, 99, axis=0)
