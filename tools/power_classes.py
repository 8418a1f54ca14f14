"""Which kernels run into the package power limit: rocm-smi (sclk, package power) while one kernel class loops (development aid).
The HBM-heavy kernels leave less power for the cores: their sclk is the lowest."""
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from stainlib_amd import _ffi, engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

n = 2048
rgb = synth_tiles(n, 1024, 1024, seed=7)
out = torch.empty_like(rgb)
tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, _ = engine.macenko_fit(tgt)
ws = engine.Workspace()
M, mc, _ = engine.macenko_fit(rgb[:8])
Mn, mcn = M[0].cpu().numpy(), mc[0].cpu().numpy()
Ms, mcs = M[:1].expand(n, 2, 3).contiguous(), mc[:1].expand(n, 2).contiguous()
f2o = lambda f: (lambda u: (~u & 0xffffffff) if u & 0x80000000 else (u | 0x80000000))(int(np.float32(f).view(np.uint32)))
Vb = np.array([0.55, 0.2, 0.7, -0.6, 0.45, 0.75])
sg = torch.zeros((n, 3), dtype=torch.float64, device="cuda") + 0.01
classes = [
    ("fused transform (512 tiles per launch)", lambda: [engine.macenko_transform(rgb[i:i + 512], Mt[0], mct[0], out=out[i:i + 512], ws=ws) for i in range(0, n, 512)]),
    ("moments sweep (read only, 20 instr/px)", lambda: engine.tile_moments(rgb)),
    ("angle window sweep (read only)", lambda: engine.slide_key_window(rgb, _ffi.KEYSET_ANGLE, Vb, (f2o(-0.35) - 32768, f2o(0.62) - 32768))),
    ("apply (read + write)", lambda: engine.normalize_apply(rgb, Ms, mcs, Mt[0], mct[0], out=out)),
    ("HED (read + write)", lambda: engine.hed_augment(rgb, sg, sg, out=out, ws=ws)),
    ("device copy (torch)", lambda: out.copy_(rgb)),
]


def smi():
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = [l for l in r.splitlines() if "sclk" in l]
    pw = [l for l in r.splitlines() if "Power (W)" in l]
    return (sclk[0].split("(")[-1].split("Mhz")[0] if sclk else "?"), (pw[0].split(":")[-1].strip() if pw else "?")


for name, fn in classes:
    stop = False

    def loop():
        while not stop:
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
    t = threading.Thread(target=loop)
    t.start()
    time.sleep(1.5)
    a = smi()
    time.sleep(1.0)
    b = smi()
    stop = True
    t.join()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    print(f"{name:42s} sclk {a[0]:>5s} / {b[0]:>5s} MHz   package power {a[1]:>7s} / {b[1]:>7s} W   {e0.elapsed_time(e1) / n * 1e3:6.3f} us per tile", flush=True)
