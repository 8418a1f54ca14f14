"""Sub-step times of the fused kernel's finish steps (development aid).  Needs a -DSL_DEVTOOLS -DSL_DEBUG_SUBCLK build:
    make -C stainlib_amd/csrc variant V=sub VFLAGS="-DSL_DEVTOOLS -DSL_DEBUG_SUBCLK"
    STAINLIB_HIP_LIB=$PWD/stainlib_amd/csrc/libstainlib_hip_sub.so python tools/sub_times.py"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stainlib_amd import _ffi, engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rgb = synth_tiles(n, 1024, 1024, seed=3)
tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, st = engine.macenko_fit(tgt)
buf = torch.zeros((n * 24,), dtype=torch.int64, device="cuda")
lib = _ffi.lib()
lib.sl_debug_set_phase_clock.argtypes = [C.c_void_p]
lib.sl_debug_bclk.argtypes = [C.c_void_p, C.c_int]
out = torch.empty_like(rgb)
engine.macenko_transform(rgb, Mt[0], mct[0], out=out)
torch.cuda.synchronize()
z = (C.c_ulonglong * 16)()
lib.sl_debug_bclk(z, 1)
lib.sl_debug_set_phase_clock(C.c_void_p(buf.data_ptr()))
engine.macenko_transform(rgb, Mt[0], mct[0], out=out)
torch.cuda.synchronize()
lib.sl_debug_set_phase_clock(C.c_void_p(0))
lib.sl_debug_bclk(z, 1)
t = buf.cpu().numpy().astype(np.float64) * 0.01
ph = t[: n * 8].reshape(n, 8)
sub = t[n * 8:].reshape(n, 16)
d = np.diff(ph, axis=1).mean(0)
print("phases (us):", " ".join(f"{x:.1f}" for x in d))
names = ["F1 eig (0->1)", None, None, "F2 refine (2->3)", "F2 ostat0 (3->4)", "F2 ostat1 (4->5)", "F2 trig (5->6)", None,
         None, "F3 refine (8->9)", "F3 ostat0 (9->10)", "F3 ostat1 (10->11)"]
for j in range(1, 12):
    if names[j]:
        print(f"  {names[j]:22s} {(sub[:, j] - sub[:, j - 1]).mean():8.1f}")
print(f"  F1 brackets (sub1->phase2) {(ph[:, 2] - sub[:, 1]).mean():8.1f}")
print(f"  F2 pre (phase3->sub2)      {(sub[:, 2] - ph[:, 3]).mean():8.1f}")
print(f"  F2 lasso consts (6->7)     {(sub[:, 7] - sub[:, 6]).mean():8.1f}")
print(f"  F2 conc brackets (7->ph4)  {(ph[:, 4] - sub[:, 7]).mean():8.1f}")
print(f"  F3 pre (phase5->sub8)      {(sub[:, 8] - ph[:, 5]).mean():8.1f}")
print(f"  F3 tail (sub11->phase6)    {(ph[:, 6] - sub[:, 11]).mean():8.1f}")
b = np.array(list(z), dtype=np.float64) * 0.01 / n
print("bracket steps per tile (us):", " ".join(f"[{i}] {x:.1f}" for i, x in enumerate(b[:8])))
