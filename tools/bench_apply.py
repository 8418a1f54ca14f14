"""Micro-benchmark of the apply kernel alone (development aid; bench.py is the contract)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from stainlib_amd import engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
h = w = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rgb = synth_tiles(n, h, w, seed=1)
M = torch.tensor([[0.626, 0.727, 0.283], [0.106, 0.987, 0.122]], dtype=torch.float64, device="cuda")
M = (M / M.norm(dim=1, keepdim=True)).expand(n, 2, 3).contiguous()
mc = torch.tensor([1.9, 1.5], dtype=torch.float64, device="cuda").expand(n, 2).contiguous()
Mt = torch.tensor([[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]], dtype=torch.float64, device="cuda")
mct = torch.tensor([2.0, 1.4], dtype=torch.float64, device="cuda")
out = torch.empty_like(rgb)
for _ in range(3):
    engine.normalize_apply(rgb, M, mc, Mt, mct, out=out)
torch.cuda.synchronize()
reps = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    engine.normalize_apply(rgb, M, mc, Mt, mct, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
gb = n * h * w * 6 / 1e9
print(f"apply: n={n} {h}x{w}  {ms:.3f} ms/launch  {gb / ms * 1e3:.1f} GB/s  {n / ms * 1e3:.0f} tiles/s  frac(8TB/s)={gb / ms * 1e3 / 8000:.3f}")
# copy baseline
t = torch.empty_like(rgb)
for _ in range(3):
    t.copy_(rgb)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    t.copy_(rgb)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"torch copy: {gb / ms * 1e3:.1f} GB/s")
