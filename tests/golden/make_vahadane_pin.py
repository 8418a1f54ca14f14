#!/usr/bin/env python3
"""Pin the oracle's Vahadane dictionary with an INDEPENDENT solver: scikit-learn's positive DictionaryLearning.

    python3 tests/golden/make_vahadane_pin.py        # the default python (3.10) of the build container: scikit-learn 1.7.2

The reference call (stainlib/extraction/vahadane_stain_extractor.py:35-36) is spams.trainDL(K=2, lambda1=0.1, mode=2,
modeD=0, posAlpha=True, posD=True): wall-clock budgeted and randomly initialised, i.e. not reproducible against
itself, and spams exists nowhere in this image.  What IS well defined is the optimum of its objective

    min_{D >= 0, |d_k| <= 1}  (1/T) sum_i  min_{a >= 0}  1/2 |x_i - D^T a|^2 + lambda |a|_1 ,

which scikit-learn's DictionaryLearning(fit_algorithm='cd', positive_code=True, positive_dict=True) minimises too
(same data term, same l1 weight, atoms projected onto the unit ball), with its own coordinate-descent lasso and its own
block update -- none of the oracle's code.  SURVEY 8c names it as the secondary oracle.  This script writes, for three
seeded synthetic tiles, the tissue optical densities' hash, scikit-learn's dictionary and its objective value;
tests/test_oracle_golden.py demands that oracle.vahadane_dictionary lands within 1e-5 of it with the same objective.
Seed 3 starts scikit-learn from a different (perturbed, swapped) dictionary: the optimum does not depend on the start.

Only arrays are written; the oracle module is imported for the tile generator, the mask and the OD table only."""
import hashlib
import os
import sys

import numpy as np
import sklearn
from sklearn.decomposition import DictionaryLearning

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import stain_oracle as so  # noqa: E402

LAM = 0.1


def objective(OD, D):
    """The objective above, with the codes solved by scikit-learn's coordinate-descent Lasso run to 1e-14
    (sklearn minimises 1/(2 n) |y - Xw|^2 + alpha |w|_1 with n = 3 rows: alpha = lambda / 3)."""
    from sklearn.linear_model import Lasso
    est = Lasso(alpha=LAM / 3, fit_intercept=False, positive=True, tol=1e-14, max_iter=1000000, precompute=False)
    est.fit(D.T, OD.T)
    A = est.coef_
    r = OD - A @ D
    return float((0.5 * (r * r).sum(1) + LAM * A.sum(1)).mean())


def case(size, seed, perturb):
    I = so.synth_tile(size, size, seed)
    mask = so.tissue_mask(I).ravel()
    OD = so.rgb_to_od(I).reshape(-1, 3)[mask]
    D0 = so.normalize_rows(np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]]))
    if perturb:
        D0 = so.normalize_rows(D0[::-1] * np.array([[1.3, 0.8, 1.1], [0.7, 1.0, 1.6]]))
    dl = DictionaryLearning(n_components=2, alpha=LAM, fit_algorithm="cd", transform_algorithm="lasso_cd",
                            positive_code=True, positive_dict=True, max_iter=5000, tol=1e-14, dict_init=D0,
                            random_state=0, transform_max_iter=100000)
    dl.fit(OD)
    D = dl.components_
    if D[0, 0] < D[1, 0]:
        D = D[::-1]                          # H row first, as vahadane_stain_extractor.py:40-41 orders them
    return {"size": size, "seed": seed, "input_sha": hashlib.sha256(I.tobytes()).hexdigest(),
            "od_sha": hashlib.sha256(np.ascontiguousarray(OD).tobytes()).hexdigest(), "n_tissue": int(mask.sum()),
            "D": np.array(D), "objective": objective(OD, D), "n_iter": int(dl.n_iter_), "lambda": LAM,
            "perturbed_start": bool(perturb), "sklearn_version": sklearn.__version__}


def main():
    for seed, perturb in ((1, False), (2, False), (3, True)):
        rec = case(96, seed, perturb)
        path = os.path.join(HERE, "vahadane_pin_96_s%d.npz" % seed)
        np.savez_compressed(path, **rec)
        print("wrote", path, rec["D"], rec["objective"], rec["n_iter"])


if __name__ == "__main__":
    main()
