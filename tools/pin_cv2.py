#!/usr/bin/env python3
"""Pin the oracle's OpenCV restatement against a REAL cv2 -- for whoever has opencv-python installed (no GPU needed).

The build container has no cv2 (and no network), so three pieces of oracle/stain_oracle.py are "parity unpinned":

  * lab_l8 / tissue_mask        cv2.cvtColor(I, COLOR_RGB2LAB)[..., 0]         stainlib/utils/stain_utils.py:41
  * rgb2lab_u8                  cv2.cvtColor(I, COLOR_RGB2LAB), all channels   stain_utils.py:62,152
  * lab2rgb_u8                  cv2.cvtColor(LAB, COLOR_LAB2RGB)               stain_utils.py:66,172

They are restated from OpenCV's published integer algorithm (modules/imgproc/src/color_lab.cpp: RGB2Lab_b,
Lab2RGBinteger).  This script runs all 2^24 uint8 triples through cv2 and through the oracle and reports every
difference.  With --write it also stores what cv2 returned as fixtures:

    tests/golden/cv2_mask_bits.npz       the tissue mask (L8/255.0 < thr) of every colour at 0.6 / 0.8 / 0.9, packed bits (2 MB each)
    tests/golden/cv2_lab_sample.npz      cv2's Lab / RGB for every 251st triple (both directions) + SHA-256 of the full arrays

Commit those and tests/test_oracle_golden.py::test_cv2_pins (skipped while the files are absent) turns the three rows
from "unpinned" to "pinned by vectors of opencv-python <version>".

    python tools/pin_cv2.py            # compare, exit status 1 on any difference
    python tools/pin_cv2.py --write    # compare and write the fixtures
"""
import argparse
import hashlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import stain_oracle as so  # noqa: E402


def all_triples():
    v = np.arange(1 << 24, dtype=np.uint32)
    return np.stack([v & 255, (v >> 8) & 255, v >> 16], axis=-1).astype(np.uint8).reshape(4096, 4096, 3)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="store cv2's answers under tests/golden/")
    a = ap.parse_args()
    try:
        import cv2
    except ImportError:
        print("cv2 is not installed here: nothing can be pinned (pip install opencv-python==4.4.0.46 is the reference's pin)")
        return 2
    print("opencv-python", cv2.__version__, "(the reference's environment.yml:143 pins 4.4.0.46)")
    I = all_triples()
    bad = 0
    lab_cv = cv2.cvtColor(I, cv2.COLOR_RGB2LAB)
    lab_or = so.rgb2lab_u8(I)
    for c, name in enumerate("Lab"):
        d = lab_cv[..., c].astype(np.int16) - lab_or[..., c].astype(np.int16)
        n = int((d != 0).sum())
        bad += n
        print(f"RGB2LAB channel {name}: {n} of {d.size} colours differ (max |delta| {int(np.abs(d).max())})")
        if n:
            idx = np.argwhere(d != 0)[:5]
            for y, x in idx:
                print("   rgb", I[y, x].tolist(), "cv2", lab_cv[y, x].tolist(), "oracle", lab_or[y, x].tolist())
    masks = {}
    for thr in (0.6, 0.8, 0.9):
        m_cv = (lab_cv[..., 0] / 255.0) < thr
        m_or = (so.lab_l8(I) / 255.0) < thr
        n = int((m_cv != m_or).sum())
        bad += n
        masks[f"thr_{thr}"] = np.packbits(m_cv.ravel())
        print(f"tissue mask at {thr}: {n} of {m_cv.size} colours differ; oracle index threshold {so.y_index_threshold(thr)}")
    rgb_cv = cv2.cvtColor(I, cv2.COLOR_LAB2RGB)          # the same 2^24 triples read as (L8, a8, b8)
    rgb_or = so.lab2rgb_u8(I)
    d = rgb_cv.astype(np.int16) - rgb_or.astype(np.int16)
    n = int((d != 0).any(axis=-1).sum())
    bad += n
    print(f"LAB2RGB: {n} of {d.shape[0] * d.shape[1]} triples differ (max |delta| {int(np.abs(d).max())})")
    if n:
        for y, x in np.argwhere((d != 0).any(axis=-1))[:5]:
            print("   lab", I[y, x].tolist(), "cv2", rgb_cv[y, x].tolist(), "oracle", rgb_or[y, x].tolist())
    m1, s1 = cv2.meanStdDev(lab_cv[:64, :64, 0].astype(np.float32) / np.float32(2.55))
    m2, s2 = so.mean_std_dev(lab_cv[:64, :64, 0].astype(np.float32) / np.float32(2.55))
    print(f"meanStdDev: mean {float(m1[0, 0])!r} vs {float(m2[0, 0])!r}, std {float(s1[0, 0])!r} vs {float(s2[0, 0])!r}")
    if a.write:
        g = os.path.join(REPO, "tests", "golden")
        np.savez_compressed(os.path.join(g, "cv2_mask_bits.npz"), cv2_version=cv2.__version__, **masks)
        np.savez_compressed(os.path.join(g, "cv2_lab_sample.npz"), cv2_version=cv2.__version__, stride=251,
                            rgb2lab=lab_cv.reshape(-1, 3)[::251], lab2rgb=rgb_cv.reshape(-1, 3)[::251],
                            rgb2lab_sha=sha(lab_cv), lab2rgb_sha=sha(rgb_cv))
        print("wrote tests/golden/cv2_mask_bits.npz and tests/golden/cv2_lab_sample.npz -- commit them")
    print("PINNED: the restatement reproduces this cv2 on every input" if bad == 0 else f"NOT PINNED: {bad} differences")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
