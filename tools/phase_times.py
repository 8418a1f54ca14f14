"""Per-phase time of the fused kernel (development aid).  Needs the -DSL_DEVTOOLS build of the library:
    make -C stainlib_amd/csrc dev && STAINLIB_HIP_LIB=$PWD/stainlib_amd/csrc/libstainlib_hip_dev.so python tools/phase_times.py"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stainlib_amd import _ffi, engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rgb = synth_tiles(n, size, size, seed=3)
tgt = synth_tiles(1, size, size, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, st = engine.macenko_fit(tgt)
buf = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
lib = _ffi.lib()
lib.sl_debug_set_phase_clock.argtypes = [C.c_void_p]
out = torch.empty_like(rgb)
engine.macenko_transform(rgb, Mt[0], mct[0], out=out)
torch.cuda.synchronize()
lib.sl_debug_set_phase_clock(C.c_void_p(buf.data_ptr()))
engine.macenko_transform(rgb, Mt[0], mct[0], out=out)
torch.cuda.synchronize()
lib.sl_debug_set_phase_clock(C.c_void_p(0))
t = buf.cpu().numpy().astype(np.float64) * 0.01      # 100 MHz -> us
d = np.diff(t, axis=1)
names = ["sweep1 moments", "finish1 eig+brackets", "sweep2 angle", "finish2 M + C brackets", "sweep3 conc", "finish3 maxC", "sweep4 apply"]
print("per-tile phase time (us): mean / median / max over", n, "tiles")
for i, nm in enumerate(names):
    print(f"  {nm:24s} {d[:, i].mean():9.1f} {np.median(d[:, i]):9.1f} {d[:, i].max():9.1f}")
print("  total per tile           %9.1f" % (t[:, 7] - t[:, 0]).mean(), " kernel span %.1f us" % (t[:, 7].max() - t[:, 0].min()))
