"""The 1024-thread fused kernel (schedule 3: one workgroup per CU, batches of up to #CU tiles) with three sweeps against the two-sweep route
forced (SlParams.two_sweep = 2), interleaved; outputs must be identical.
    python tools/wide_two_sweep.py [counts=128,192,224,256] [size=1024] [kind=iid|ihc]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stainlib_amd import engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "128,192,224,256").split(",")]
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kind = sys.argv[3] if len(sys.argv) > 3 else "iid"
tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, _ = engine.macenko_fit(tgt)
ws = engine.Workspace()
for n in counts:
    rgb = synth_tiles(n, size, size, seed=7)
    if kind == "ihc":
        I = np.load(os.path.join("tests", "golden", "tissue_ihc_512.npz"))["input"]
        row = np.concatenate([I, I[:, ::-1]], axis=1)
        T = torch.as_tensor(np.ascontiguousarray(np.concatenate([row, row[::-1]], axis=0))[:size, :size], device="cuda")
        rgb[:] = T
    runs = [dict(p=engine.make_params(schedule=3, two_sweep=m), out=torch.empty_like(rgb)) for m in (1, 2, 0)]
    fns = [(lambda r=r: engine.macenko_transform(rgb, Mt[0], mct[0], params=r["p"], out=r["out"], ws=ws)) for r in runs]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while True:
        for _ in range(8):
            fns[0]()
        e1.record(); torch.cuda.synchronize()
        if e0.elapsed_time(e1) >= 150.0:
            break
    ts = [[] for _ in fns]
    for _ in range(8):
        for i, fn in enumerate(fns):
            fn(); fn()
            e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
            ts[i].append(e0.elapsed_time(e1) / 3.0)
    res = [fn() for fn in fns]
    torch.cuda.synchronize()
    same = all(torch.equal(res[0][0], r[0]) and torch.equal(torch.nan_to_num(res[0][1]), torch.nan_to_num(r[1])) for r in res[1:])
    print(f"{kind} n {n:4d}: three sweeps {np.median(ts[0]):.4f} ms  two-sweep forced {np.median(ts[1]):.4f} ms  automatic {np.median(ts[2]):.4f} ms   identical {same}", flush=True)
