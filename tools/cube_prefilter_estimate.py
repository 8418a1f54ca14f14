"""CPU estimate for DESIGN section 9's colour-cube pre-filter: which share of a tile's pixels falls into cells of a 2^(3*bits) colour
cube that hold at least one colour the merged selection sweep could NOT prove plain (inside or next to an angular bracket, or above the
lower end of a concentration bracket)?  Brackets as finish 1 makes them: sample ranks -/+ 6 sigma of a 16 Ki-pixel sample, taken here
from the exact keys of all pixels.  Best case for the idea: every colour of a cell is enumerated (no interval arithmetic), exact M
instead of the box.    python tools/cube_prefilter_estimate.py [kind] [size] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import stain_oracle as so  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "iid"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    if kind == "iid":
        I = so.synth_tile(size, size, seed)
    elif kind == "ihc":
        ihc = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tissue_ihc_512.npz"))["input"]
        big = np.concatenate([ihc, ihc[:, ::-1]], axis=1)
        I = np.concatenate([big, big[::-1]], axis=0)[:size, :size].copy()
    else:
        I = so.structured_tile(kind, size, size, seed)
    det = {}
    M = so.macenko_stain_matrix(I, details=det)
    V, mask = det["V"], det["mask"]
    n_tis = int(mask.sum())
    P = I.shape[0] * I.shape[1]
    n_s = 16384 * n_tis / P                                  # tissue entries of a 16 Ki sample
    # all 2^24 colours: tissue?, angle, concentrations
    c = np.arange(1 << 24, dtype=np.uint32)
    cube = np.stack([(c >> 16) & 255, (c >> 8) & 255, c & 255], axis=1).astype(np.uint8).reshape(4096, 4096, 3)
    tis = so.tissue_mask_unchecked(cube, 0.8).reshape(-1) if hasattr(so, "tissue_mask_unchecked") else (so.rgb2lab_u8(cube)[..., 0].reshape(-1) / 255.0 < 0.8)
    od = so.rgb_to_od(cube).reshape(-1, 3)
    That = od @ V
    phi = np.arctan2(That[:, 1], That[:, 0])
    C = so.get_concentrations(cube, M)
    # brackets from the tile's own pixels
    key = (I[..., 0].astype(np.uint32) << 16 | I[..., 1].astype(np.uint32) << 8 | I[..., 2]).reshape(-1)
    phi_t = np.sort(phi[key[mask]])
    amb = np.zeros(1 << 24, bool)
    for p in (0.01, 0.99):
        sd = np.sqrt(n_s * p * (1 - p))
        lo = phi_t[int(np.clip((n_s * p - 6 * sd) / n_s * n_tis, 0, n_tis - 1))]
        hi = phi_t[int(np.clip((n_s * p + 6 * sd) / n_s * n_tis, 0, n_tis - 1))]
        amb |= tis & (phi >= lo) & (phi <= hi)
    for col in range(2):
        ct = np.sort(C[key, col])
        p, n_all = 0.99, 16384.0
        sd = np.sqrt(n_all * p * (1 - p))
        lo = ct[int(np.clip((n_all * p - 6 * sd) / n_all * P, 0, P - 1))]
        amb |= C[:, col] >= lo
    print(f"{kind} {size}^2 seed {seed}: tissue {n_tis / P:.3f}; pixels the exact sweep collects {amb[key].mean():.4f}")
    for bits in (4, 5, 6):
        sh = 8 - bits
        cell = ((c >> 16 & 255) >> sh) << (2 * bits) | ((c >> 8 & 255) >> sh) << bits | ((c & 255) >> sh)
        cell_amb = np.zeros(1 << (3 * bits), bool)
        np.logical_or.at(cell_amb, cell, amb)
        frac = cell_amb[cell[key]].mean()
        print(f"   {1 << bits}^3 cells ({(1 << (3 * bits)) // 8} B bitmask): ambiguous cells {cell_amb.mean():.3f}, pixels in them {frac:.4f}")


if __name__ == "__main__":
    main()
