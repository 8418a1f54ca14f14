"""A/B of the colour-cube pre-filter of the fused Macenko kernel's selection sweep (SlParams.prefilter: 1 = off, 0 = where the
sample says it pays, 2 = wherever the mask can be built): ms per 512-tile launch, tiles that swept behind the mask, resweeps and
exact fallbacks, and byte / statistics identity of the three (results must not depend on the pre-filter).
    python tools/cube_ab.py [kinds...]      kinds: iid blobs white_bg quantized grey_bg ihc palette"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import stain_oracle as so  # noqa: E402
from stainlib_amd import engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

dev = torch.device("cuda", 0)
N = int(os.environ.get("SL_AB_TILES", "512"))
SIZE = int(os.environ.get("SL_AB_SIZE", "1024"))


def batch(kind):
    if kind == "iid":
        return synth_tiles(N, SIZE, SIZE, seed=7, device=dev)
    if kind == "ihc":
        I = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tissue_ihc_512.npz"))["input"]
        row = np.concatenate([I, I[:, ::-1]], axis=1)
        T = np.ascontiguousarray(np.concatenate([row, row[::-1]], axis=0))[:SIZE, :SIZE]
        four = np.stack([T, np.roll(T, 301, axis=0), np.roll(T, 517, axis=1), np.ascontiguousarray(T.transpose(1, 0, 2))])
    elif kind == "grey_bg":
        rng = np.random.RandomState(8)
        four = []
        for s in range(4):
            I = so.synth_tile(SIZE, SIZE, 40 + s).copy()
            I[rng.rand(SIZE, SIZE) < 0.6] = 245
            four.append(I)
        four = np.stack(four)
    elif kind == "palette":
        four = np.stack([so.structured_tile("palette12", SIZE, SIZE, 20 + s) for s in range(4)])
    else:
        four = np.stack([so.structured_tile(kind, SIZE, SIZE, 20 + s) for s in range(4)])
    return torch.as_tensor(np.ascontiguousarray(four), device=dev)[torch.arange(N, device=dev) % 4].contiguous()


def timed(fn, reps=10):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.12:
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tgt = synth_tiles(1, 1024, 1024, seed=1, device=dev, M_true=so.M_TRUE_TGT.tolist())
Mt, mct, _ = engine.macenko_fit(tgt)
Mt, mct = Mt[0].contiguous(), mct[0].contiguous()
ws = engine.Workspace()
for kind in sys.argv[1:] or ["iid", "white_bg", "quantized", "ihc", "grey_bg", "blobs"]:
    rgb = batch(kind)
    outs = {}
    line = f"{kind:10s}"
    for pf in (1, 0, 2, 1, 0):
        out = torch.empty_like(rgb)
        p = engine.make_params(schedule=2, prefilter=pf)
        fb = engine.attach_fallbacks(p, N, device=dev)
        rsw = torch.zeros((N,), dtype=torch.int32, device=dev)
        cub = torch.zeros((N,), dtype=torch.int32, device=dev)
        p.resweeps_out, p.prefilter_out = rsw.data_ptr(), cub.data_ptr()
        ms = timed(lambda: engine.macenko_transform(rgb, Mt, mct, params=p, out=out, ws=ws))
        o, M, mc, st = engine.macenko_transform(rgb, Mt, mct, params=p, out=out, ws=ws)
        torch.cuda.synchronize()
        line += f" | pf{pf} {ms:6.3f} ms cube {int((cub & 1).sum()):3d} ({float((cub >> 8).float().mean()):.0f} %) rsw {int((rsw != 0).sum()):3d} fb {int(fb.sum()):3d} bad {int((st != 0).sum())}"
        if pf not in outs:
            outs[pf] = (o.clone(), M.clone(), mc.clone(), st.clone())
    same = all(torch.equal(outs[1][0], outs[k][0]) and torch.equal(outs[1][1], outs[k][1]) and torch.equal(outs[1][2], outs[k][2]) and
               torch.equal(outs[1][3], outs[k][3]) for k in (0, 2))
    print(line + f" | identical {same}", flush=True)
    del rgb, outs
