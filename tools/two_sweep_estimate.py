"""What a schedule with ONE read sweep less would have to collect: the eigenvectors of a tile's 16 Ki-pixel sample against
the exact ones, and the share of tissue pixels whose angular key could leave the plain zone under any eigenvector pair within k times that
distance -- the candidates a sweep that does not yet know the exact eigenvectors must keep.  CPU only (numpy, the oracle's tiles).
    python tools/two_sweep_estimate.py [k=3]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import stain_oracle as so  # noqa: E402

K = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


def planes(OD):
    _, V = np.linalg.eigh(np.cov(OD, rowvar=False))
    V = V[:, [2, 1]]
    V[:, 0] *= np.sign(V[0, 0]) or 1.0
    V[:, 1] *= np.sign(V[0, 1]) or 1.0
    return V


def report(name, I):
    OD = so.rgb_to_od(I).reshape(-1, 3)
    OD = OD[so.tissue_mask(I).ravel()]
    V = planes(OD)
    rng = np.random.RandomState(1)
    S = OD[rng.choice(len(OD), size=min(len(OD), 16384 * len(OD) // (I.shape[0] * I.shape[1]) or 1), replace=False)]
    Vs = planes(S)
    dist = float(np.abs(V - Vs).max())
    phi = np.arctan2(OD @ V[:, 1], OD @ V[:, 0])
    lo, hi = np.percentile(phi, 1), np.percentile(phi, 99)
    # brackets as the sample leaves them (+- 6 sigma of the rank) and the plain zone between them
    phis = np.arctan2(S @ Vs[:, 1], S @ Vs[:, 0])
    n = len(phis)
    sd = 6.0 * np.sqrt(0.01 * 0.99 * n)
    b_lo = np.sort(phis)[[max(0, int(0.01 * n - sd)), min(n - 1, int(0.01 * n + sd))]]
    b_hi = np.sort(phis)[[max(0, int(0.99 * n - sd)), min(n - 1, int(0.99 * n + sd))]]
    today = float(((phi <= b_lo[1]) | (phi >= b_hi[0])).mean())
    # under every V within K * dist (max-abs) of the sample's: a pixel's projections move by at most |od|_1 * K * dist each, its angle by
    # at most atan of that over the projection's length
    r = np.hypot(OD @ Vs[:, 0], OD @ Vs[:, 1])
    slack = np.arctan2(np.abs(OD).sum(1) * K * dist * np.sqrt(2.0), np.maximum(r, 1e-12))
    phi_s = np.arctan2(OD @ Vs[:, 1], OD @ Vs[:, 0])
    wide = float(((phi_s - slack <= b_lo[1]) | (phi_s + slack >= b_hi[0])).mean())
    print(f"{name:10s} tissue {len(OD):8d}  |V - V_sample| {dist:.2e}  angular candidates today {100 * today:5.2f} %  "
          f"under every V within {K:g} x that: {100 * wide:5.2f} %   (percentile angles {lo:.3f}, {hi:.3f})")


if __name__ == "__main__":
    report("iid", so.synth_tile(1024, 1024, 7))
    for kind in ("white_bg", "quantized", "blobs"):
        report(kind, so.structured_tile(kind, 1024, 1024, 21))
    ihc = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tissue_ihc_512.npz"))["input"]
    row = np.concatenate([ihc, ihc[:, ::-1]], axis=1)
    report("ihc", np.ascontiguousarray(np.concatenate([row, row[::-1]], axis=0)))
