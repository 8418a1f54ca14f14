"""Phase times of the fused Macenko kernel per tile with the development build (make -C stainlib_amd/csrc dev;
STAINLIB_HIP_LIB=.../libstainlib_hip_dev.so): two-sweep schedule (phase 0, merged sweep, finish, apply) beside the three-sweep one.
    python tools/ts_phases.py [tiles=512] [size=1024] [kind=iid]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stainlib_amd import _ffi, engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kind = sys.argv[3] if len(sys.argv) > 3 else "iid"
if kind == "iid":
    rgb = synth_tiles(n, size, size, seed=3)
else:
    from oracle import stain_oracle as so
    import os
    if kind == "ihc":
        I = np.load(os.path.join("tests", "golden", "tissue_ihc_512.npz"))["input"]
        row = np.concatenate([I, I[:, ::-1]], axis=1)
        T = np.ascontiguousarray(np.concatenate([row, row[::-1]], axis=0))
        base = np.stack([T, np.roll(T, 301, axis=0), np.roll(T, 517, axis=1), T.transpose(1, 0, 2)])[:, :size, :size]
    else:
        base = np.stack([so.structured_tile(kind, size, size, 20 + s) for s in range(4)])
    rgb = torch.as_tensor(np.ascontiguousarray(base), device="cuda")[torch.arange(n, device="cuda") % 4].contiguous()
tgt = synth_tiles(1, size, size, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, st = engine.macenko_fit(tgt)
lib = _ffi.lib()
assert hasattr(lib, "sl_debug_set_phase_clock"), "needs the development build"
lib.sl_debug_set_phase_clock.argtypes = [C.c_void_p]
out = torch.empty_like(rgb)
for mode, name in ((1, "three-sweep"), (0, "two-sweep")):
    p = engine.make_params(schedule=2, two_sweep=mode)
    tso = torch.zeros((n,), dtype=torch.int32, device="cuda")
    p.twosweep_out = tso.data_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while True:                                                  # the clocks take ~25 ms of load to come up after the host-side setup
        for _ in range(8):
            engine.macenko_transform(rgb, Mt[0], mct[0], out=out, params=p)
        e1.record(); torch.cuda.synchronize()
        if e0.elapsed_time(e1) >= 200.0:
            break
    buf = torch.zeros((n * 40,), dtype=torch.int64, device="cuda")
    lib.sl_debug_set_phase_clock(C.c_void_p(buf.data_ptr()))
    lib.sl_debug_set_stop(-7)                                  # phase 0 writes its sub-step clocks into the third region
    engine.macenko_transform(rgb, Mt[0], mct[0], out=out, params=p)
    torch.cuda.synchronize()
    lib.sl_debug_set_stop(0)
    lib.sl_debug_set_phase_clock(C.c_void_p(0))
    t = buf.cpu().numpy().astype(np.float64)[: n * 8].reshape(n, 8) * 0.01
    t0 = t[:, 0].min()
    direct = (tso.cpu().numpy() == 1)
    print(f"{name}: direct {int(direct.sum())} of {n}; kernel span {t[:, 7].max() - t0:.1f} us")
    if mode == 0:
        t4 = np.where(t[:, 4] > 0, t[:, 4], t[:, 0])
        cols = [("phase 0", t4 - t[:, 0]), ("sweep 1 (moments + candidates)", t[:, 1] - t4), ("finish", t[:, 6] - t[:, 1]), ("apply", t[:, 7] - t[:, 6])]
    else:
        cols = [("sweep 1", t[:, 1] - t[:, 0]), ("finish 1", t[:, 2] - t[:, 1]), ("sweep 2", t[:, 3] - t[:, 2]), ("finish 2", t[:, 6] - t[:, 3]), ("apply", t[:, 7] - t[:, 6])]
    h = n // 2
    for nm, v in cols:
        print(f"   {nm:32s} mean {v.mean():7.1f}  median {np.median(v):7.1f}  max {v.max():7.1f}   first half {v[:h].mean():7.1f} second half {v[h:].mean():7.1f}")
    if mode == 0:
        sub = buf.cpu().numpy().astype(np.float64)[n * 24:].reshape(n, 16) * 0.01
        names = ["gather", "sample moments + eig", "fourth moments + angle brackets", "half-spaces + box", "conc brackets + thresholds", "cube tables, share, mask"]
        print("   phase 0: " + "  ".join(f"{nm} {(sub[direct, i + 1] - sub[direct, i]).mean():.1f}" for i, nm in enumerate(names[:6])) + f"  tail {(t[direct, 4] - sub[direct, 6]).mean():.1f}   (tiles on the direct route; of the conc step, the candidate prediction pass: {(sub[direct, 5] - sub[direct, 7]).mean():.1f})")
    sb = buf.cpu().numpy().astype(np.float64)[n * 8: n * 24].reshape(n, 16) * 0.01
    sel = direct if mode == 0 else np.ones(n, bool)
    if sel.any():
        f = [("sums+eig", sb[:, 1] - t[:, 1]), ("verify / finish 1 + sweep 2", sb[:, 2] - sb[:, 1]), ("refine angle", sb[:, 3] - sb[:, 2]), ("pick angle", sb[:, 5] - sb[:, 3]),
             ("M + verify", sb[:, 6] - sb[:, 5]), ("refine conc", sb[:, 14] - sb[:, 6]), ("pick conc", sb[:, 15] - sb[:, 14]), ("tail -> apply", t[:, 6] - sb[:, 15])]
        print("   finish steps (tiles on this route): " + "  ".join(f"{nm} {v[sel].mean():.1f}" for nm, v in f))
    cnt = buf.cpu().numpy().astype(np.float64)[n * 8: n * 24].reshape(n, 16)[:, 8:12] * 0.01
    print("   lists per tile (mean / max): angular candidates %.0f / %.0f, concentration candidates %.0f / %.0f, angle members %.0f / %.0f, concentration members %.0f / %.0f" % (
        cnt[:, 0].mean(), cnt[:, 0].max(), cnt[:, 1].mean(), cnt[:, 1].max(), cnt[:, 2].mean(), cnt[:, 2].max(), cnt[:, 3].mean(), cnt[:, 3].max()))
    tot = t[:, 7] - t[:, 0]
    print(f"   total per tile {tot.mean():.1f}; ends pct 0/50/100: {np.percentile(t[:, 7] - t0, [0, 50, 100]).round(1)}")
