"""Fused Macenko transform, default parameters, on a few batch shapes: median / min ms after a spin-up (one process = one library setting).
    python tools/time_batches.py [kinds=iid,mixed] [shapes=512x1024,1024x1024,2048x512,700x1024]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stainlib_amd import engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

kinds = (sys.argv[1] if len(sys.argv) > 1 else "iid,mixed").split(",")
shapes = [tuple(int(v) for v in x.split("x")) for x in (sys.argv[2] if len(sys.argv) > 2 else "512x1024,1024x1024,2048x512,700x1024").split(",")]
tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, _ = engine.macenko_fit(tgt)
ws = engine.Workspace()
for n, size in shapes:
    for kind in kinds:
        rgb = synth_tiles(n, size, size, seed=7)
        if kind == "mixed":
            rgb[3::4] = 255
        out = torch.empty_like(rgb)
        p = engine.make_params(schedule=2)
        fn = lambda: engine.macenko_transform(rgb, Mt[0], mct[0], params=p, out=out, ws=ws)  # noqa: E731
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while True:
            for _ in range(8):
                fn()
            e1.record(); torch.cuda.synchronize()
            if e0.elapsed_time(e1) >= 150.0:
                break
        ts = []
        for _ in range(10):
            e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3.0)
        print(f"{n:5d} x {size}^2 {kind:6s}: {np.median(ts):7.4f} ms (min {min(ts):7.4f})", flush=True)
