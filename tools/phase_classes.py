"""Per-kernel-class time of the one-launch-per-phase Macenko transform by batch size (development aid).

Tells how fast each sweep kernel runs when the batch fits the Infinity Cache (<= ~64 tiles of 1024^2) and when it
does not: the price of the HBM re-reads per sweep.  usage: python tools/phase_classes.py [size]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from bench import HipEvents  # noqa: E402
from stainlib_amd import _ffi, engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tgt = synth_tiles(1, size, size, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, st = engine.macenko_fit(tgt)
ev = HipEvents(4096)
ns = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 32, 48, 64, 96, 128, 256, 512]
for n in ns:
    rgb = synth_tiles(n, size, size, seed=3)
    out = torch.empty_like(rgb)
    p = engine.make_params(schedule=1)
    for _ in range(3):
        engine.macenko_transform(rgb, Mt[0], mct[0], params=p, out=out)
    torch.cuda.synchronize()
    acc = {}
    reps = 5
    for _ in range(reps):
        prof = ev.profile(255)
        p.profile = C.pointer(prof)
        engine.macenko_transform(rgb, Mt[0], mct[0], params=p, out=out)
        torch.cuda.synchronize()
        seq = ev.pairs(prof)
        for i, (tag, tiles, ms) in enumerate(seq):
            key = f"{i}:{_ffi.PROF_NAMES[tag]}"
            acc[key] = acc.get(key, 0.0) + ms / reps
    p.profile = None
    tot = sum(acc.values())
    print(f"size {size} n {n:4d}: total {tot * 1e3:8.1f} us  " + "  ".join(f"{k} {v * 1e3:.1f}" for k, v in acc.items()))
