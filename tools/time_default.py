"""ms per launch of the fused Macenko transform with DEFAULT parameters (schedule 2) on one batch kind, after a spin-up; one line per kind.
    python tools/time_default.py [kinds=iid] [n=512] [size=1024] [two_sweep=0]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import stain_oracle as so  # noqa: E402
from stainlib_amd import engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
SIZE = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda", 0)


def batch(kind):
    if kind == "iid":
        return synth_tiles(N, SIZE, SIZE, seed=7, device=dev)
    if kind == "ihc":
        I = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tissue_ihc_512.npz"))["input"]
        row = np.concatenate([I, I[:, ::-1]], axis=1)
        T = np.ascontiguousarray(np.concatenate([row, row[::-1]], axis=0))[:SIZE, :SIZE]
        four = np.stack([T, np.roll(T, 301, axis=0), np.roll(T, 517, axis=1), np.ascontiguousarray(T.transpose(1, 0, 2))])
    else:
        four = np.stack([so.structured_tile(kind, SIZE, SIZE, 20 + s) for s in range(4)])
    return torch.as_tensor(np.ascontiguousarray(four), device=dev)[torch.arange(N, device=dev) % 4].contiguous()


tgt = synth_tiles(1, 1024, 1024, seed=1, device=dev, M_true=so.M_TRUE_TGT.tolist())
Mt, mct, _ = engine.macenko_fit(tgt)

kinds = (sys.argv[1] if len(sys.argv) > 1 else "iid").split(",")
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ws = engine.Workspace()
for kind in kinds:
    rgb = batch(kind)
    out = torch.empty_like(rgb)
    p = engine.make_params(schedule=2, two_sweep=mode)
    fn = lambda: engine.macenko_transform(rgb, Mt[0], mct[0], params=p, out=out, ws=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while True:
        for _ in range(8):
            fn()
        e1.record(); torch.cuda.synchronize()
        if e0.elapsed_time(e1) >= 200.0:
            break
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 5.0)
    print(f"{kind} {np.median(ts):.4f} ms (min {min(ts):.4f})", flush=True)
