"""What a drop-in user sees: MacenkoNormalizer / VahadaneNormalizer / Reinhard .transform(numpy image) per call, host array in,
host array out (H2D + kernels + D2H + Python), by tile size.  Next to it the device-resident engine call alone."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import stainlib_amd as sl  # noqa: E402
from stainlib_amd import engine  # noqa: E402
from oracle import stain_oracle as so  # noqa: E402


def per_call(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6


tgt = so.synth_tile(256, 256, 1001, so.M_TRUE_TGT)
for size in (256, 512, 1024, 2048):
    I = so.synth_tile(size, size, 7)
    row = [f"{size:5d}^2"]
    for name, cls in (("macenko", sl.MacenkoNormalizer), ("vahadane", sl.VahadaneNormalizer), ("reinhard", sl.ReinhardStainNormalizer)):
        n = cls()
        n.fit(tgt)
        row.append(f"{name} {per_call(lambda: n.transform(I)):8.0f} us")
    d = torch.as_tensor(I[None], device="cuda")
    o = torch.empty_like(d)
    nm = sl.MacenkoNormalizer()
    nm.fit(tgt)
    Mt = torch.as_tensor(nm.stain_matrix_target, device="cuda")
    mct = torch.as_tensor(nm.maxC_target.reshape(2), device="cuda")
    ws = engine.Workspace()
    row.append(f"| device-resident macenko_transform {per_call(lambda: engine.macenko_transform(d, Mt, mct, out=o, ws=ws)):8.0f} us")
    print("  ".join(row), flush=True)
