"""One Vahadane fit from the inside: the dictionary after k full sweeps (dl_max_sweeps = 1, 2, ...), both schedules, its
objective next to the oracle's converged one.  Arguments: the label test_gpu_stress prints on a failure, e.g.
    python tools/vahadane_case.py 79 32      (SL_FUZZ_SEED, index of the case in the random test's sequence)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import stain_oracle as so  # noqa: E402
from stainlib_amd import engine  # noqa: E402
from tests.gpu_util import to_dev  # noqa: E402
from tests.test_gpu_stress import vahadane_cases  # noqa: E402


def main():
    seed, index = int(sys.argv[1]), int(sys.argv[2])
    for i, (label, I, thr, lam, sched) in enumerate(vahadane_cases(seed)):
        if i == index:
            break
    print(label)
    OD = so.rgb_to_od(I).reshape(-1, 3)[so.tissue_mask(I, thr).ravel()]

    def obj(D):
        Cc = so.lasso2_nonneg(OD, D, lam)
        r = OD - Cc @ D
        return (0.5 * (r * r).sum(1) + lam * Cc.sum(1)).mean()
    info = {}
    Mo = so.vahadane_stain_matrix(I, thr, lam, 2000, 1e-12, info=info)
    print("oracle", info, Mo.round(5).tolist(), "start", obj(so.vahadane_init(OD)))
    for s in (1, 2):
        for k in (1, 2, 3, 4, 6, 8, 12, 20, 50, 400):
            p = engine.make_params(luminosity_threshold=thr, dl_lambda=lam, dl_tol=1e-9, dl_max_sweeps=k, schedule=s)
            M, mc, st, sweeps = engine.vahadane_fit(to_dev([I]), params=p)
            M = M.cpu().numpy()[0]
            print(f"schedule {s} max_sweeps {k:3d}: used {int(sweeps[0]):3d} status {int(st[0])} objective {obj(M):.12f} "
                  f"|M - Mo| {np.abs(M - Mo).max():.2e} {M.round(5).tolist()}")


if __name__ == "__main__":
    main()
