"""One large image: as ONE tile (per-phase schedule, finish steps on one workgroup) vs as the pooled statistics of its row
bands + apply.  Device-resident, wall clock per call; where BIG_IMAGE_PIXELS (normalizer.py) should sit."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from stainlib_amd import engine  # noqa: E402
from stainlib_amd.distributed import PooledSlideStatistics  # noqa: E402
from stainlib_amd.normalization.normalizer import _row_bands  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402


def per_call(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, _ = engine.macenko_fit(tgt)
for size in (1024, 2048, 3072, 4096, 6144, 8192, 12288):
    img = synth_tiles(1, size, size, seed=5)
    out = torch.empty_like(img)
    ws = engine.Workspace()
    t_tile = per_call(lambda: engine.macenko_transform(img, Mt[0], mct[0], out=out, ws=ws))

    def pooled():
        M, mc = PooledSlideStatistics(group=False)(_row_bands(img))
        engine.normalize_apply(img, M[None], mc[None], Mt[0], mct[0], out=out)
    t_pool = per_call(pooled)
    print(f"{size:6d}^2 ({size * size / 2**20:6.1f} Mpx): one tile {t_tile:7.3f} ms   row bands, pooled {t_pool:7.3f} ms", flush=True)
    del img, out
