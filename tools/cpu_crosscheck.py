#!/opt/conda/bin/python3.9
"""The honesty check of BASELINE.md section 4 -- runs only in the BUILD container (needs /root/reference):

    /opt/conda/bin/python3.9 tools/cpu_crosscheck.py        # writes profiles/r02_cpu_crosscheck.json

bench.py's cpu_baseline times the repo's numpy oracle, because the reference cannot travel to the GPU box.  This script
times the REFERENCE ITSELF (stainlib imported from /root/reference, ExtractiveStainNormalizer('macenko').transform on a
1024 x 1024 tile) next to the oracle on the same tile, same process, one thread.  The reference's two absent third-party
calls get the SAME stand-ins the oracle's own code uses (cv2.cvtColor -> the integer Lab restatement; spams.lasso -> the
closed-form two-atom solve, returned as the scipy sparse matrix the reference expects), so the comparison is between the
reference's numpy code and the oracle's restatement of it.  The oracle must not be faster than the reference by more
than 20 % (it would flatter the GPU/CPU ratio) -- the result is recorded, and bench.py copies it into its JSON line."""
import json
import os
import sys
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[v] = "1"
from oracle import stain_oracle as so  # noqa: E402
import scipy.sparse  # noqa: E402

cv2 = types.ModuleType("cv2")
cv2.COLOR_RGB2LAB, cv2.COLOR_LAB2RGB = 45, 57
cv2.cvtColor = lambda I, code: so.rgb2lab_u8(I) if code == 45 else so.lab2rgb_u8(I)
sys.modules["cv2"] = cv2
spams = types.ModuleType("spams")
spams.lasso = lambda X, D, mode, lambda1, pos: scipy.sparse.csc_matrix(so.lasso2_nonneg(np.asarray(X).T, np.asarray(D).T, lambda1).T)
sys.modules["spams"] = spams
sys.path.insert(0, "/root/reference")
from stainlib.normalization.normalizer import ExtractiveStainNormalizer  # noqa: E402

size = 1024
tgt = so.synth_tile(size, size, 1, so.M_TRUE_TGT)
tiles = [so.synth_tile(size, size, 100 + i) for i in range(3)]
ref = ExtractiveStainNormalizer("macenko")
ref.fit(tgt)
orc = so.ExtractiveStainNormalizer("macenko")
orc.fit(tgt)


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        for t in tiles:
            t0 = time.perf_counter()
            fn(t)
            ts.append(time.perf_counter() - t0)
    return float(np.min(ts)), float(np.median(ts))


ref.transform(tiles[0]); orc.transform(tiles[0])
r_min, r_med = best(ref.transform)
o_min, o_med = best(orc.transform)
same = all(np.array_equal(ref.transform(t), orc.transform(t)) for t in tiles)
rec = {"where": "build container (the reference cannot travel to the GPU box)", "host": os.uname().nodename, "cpus": os.cpu_count(),
       "tile": [size, size, 3], "threads": 1, "reference_with_standins_s_per_tile": {"min": round(r_min, 4), "median": round(r_med, 4)},
       "oracle_s_per_tile": {"min": round(o_min, 4), "median": round(o_med, 4)}, "oracle_over_reference_time": round(o_med / r_med, 3),
       "within_20_percent": bool(abs(o_med / r_med - 1.0) <= 0.2), "not_sandbagged": bool(o_med / r_med <= 1.2), "outputs_identical": bool(same),
       "numpy": np.__version__, "note": "ratio < 1: the oracle is FASTER than the reference's own code run with stand-ins (it skips the reference's deepcopy in "
                                         "convert_RGB_to_OD and the scipy-sparse round trip around spams.lasso, part of which is the stand-in's own cost): the "
                                         "CPU baseline bench.py reports is therefore optimistic for the CPU -- the GPU/CPU ratio is understated, never inflated. "
                                         "A ratio above 1.2 would mean a sandbagged baseline; that is what BASELINE.md section 4 guards against"}
path = os.path.join(REPO, "profiles", "r02_cpu_crosscheck.json")
json.dump(rec, open(path, "w"), indent=1)
print(json.dumps(rec, indent=1))
