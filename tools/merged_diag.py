"""The fused Macenko kernel from the inside (development aid): how often the merged selection sweep settles the concentration
percentiles (resweeps = tiles that needed sweep 3), slow exact fallbacks, and -- with the development build
(make -C stainlib_amd/csrc dev; STAINLIB_HIP_LIB=.../libstainlib_hip_dev.so) -- per-tile phase times, the sub-steps of both finish
steps, list sizes and the bracket step timers.    python tools/merged_diag.py [tiles] [size]"""
import ctypes as C
import os
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
p = engine.make_params(schedule=2, prefilter=int(os.environ.get("SL_PREFILTER", "0")), two_sweep=1)      # SL_PREFILTER=1: the per-pixel selection sweep; the THREE-sweep route for every tile (tools/ts_phases.py times the two-sweep one)
fb = engine.attach_fallbacks(p, n)
rs = torch.full((n,), -1, dtype=torch.int32, device="cuda")
p.resweeps_out = rs.data_ptr()
out = torch.empty_like(rgb)
o, M, mc, status = engine.macenko_transform(rgb, Mt[0], mct[0], out=out, params=p)
torch.cuda.synchronize()
print(f"tiles {n} x {size}^2: resweeps {int((rs > 0).sum())} (reasons 1 no box / 2 outside box / 3 bracket missed / 4 list full: {[int((rs == k).sum()) for k in (1, 2, 3, 4)]})  untouched {int((rs == -1).sum())}  exact fallbacks {int(fb.sum())}  bad status {int((status != 0).sum())}")
def med(fn, reps=15):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


ws = engine.Workspace()
p1 = engine.make_params(schedule=1)
print("fused  ms (median, min):", med(lambda: engine.macenko_transform(rgb, Mt[0], mct[0], out=out, params=p, ws=ws)))
print("phases ms (median, min):", med(lambda: engine.macenko_transform(rgb, Mt[0], mct[0], out=out, params=p1, ws=ws)))
lib = _ffi.lib()
if hasattr(lib, "sl_debug_set_phase_clock"):
    sub_build = hasattr(lib, "sl_debug_bclk")
    buf = torch.zeros((n * 24,), dtype=torch.int64, device="cuda")
    lib.sl_debug_set_phase_clock.argtypes = [C.c_void_p]
    z = (C.c_ulonglong * 16)()
    if sub_build:
        lib.sl_debug_bclk.argtypes = [C.c_void_p, C.c_int]
        lib.sl_debug_bclk(z, 1)                                   # reset the bracket step timers
    lib.sl_debug_set_phase_clock(C.c_void_p(buf.data_ptr()))
    engine.macenko_transform(rgb, Mt[0], mct[0], out=out, params=p)
    torch.cuda.synchronize()
    lib.sl_debug_set_phase_clock(C.c_void_p(0))
    if sub_build:
        lib.sl_debug_bclk(z, 0)
    tall = buf.cpu().numpy().astype(np.float64) * 0.01
    t = tall[: n * 8].reshape(n, 8)
    if sub_build:
        sub = tall[n * 8:].reshape(n, 16)
        ph = t
        steps = [("F1 eig", sub[:, 1] - sub[:, 0]), ("F1 angle brackets", sub[:, 12] - sub[:, 1]), ("F1 box", sub[:, 13] - sub[:, 12]),
                 ("F1 conc brackets", sub[:, 4] - sub[:, 13]), ("F1 cube mask + share", sub[:, 7] - sub[:, 4]), ("F1 bound guard", ph[:, 2] - sub[:, 7]),
                 ("F2 pre", sub[:, 2] - ph[:, 3]), ("F2 refine angle", sub[:, 3] - sub[:, 2]),
                 ("F2 pick angle", sub[:, 5] - sub[:, 3]), ("F2 M + verify", sub[:, 6] - sub[:, 5]), ("F2b refine conc", sub[:, 14] - sub[:, 6]),
                 ("F2b pick conc", sub[:, 15] - sub[:, 14]), ("F2b tail -> apply", ph[:, 6] - sub[:, 15])]
        cnt = tall[n * 8:].reshape(n, 16)[:, 8:12]       # slots 8..11: list sizes x 100 written by the kernel (not clocks)
        print("    per tile: raw entries with an angle key %.0f, raw entries %.0f, angle members %.0f, concentration members %.0f" % tuple(cnt.mean(0)))
        if True:
            bz = np.array(list(z), dtype=np.float64) * 0.01 / n
            print("    bracket steps per tile (us, summed over the two calls): " + " ".join(f"[{i}] {x:.1f}" for i, x in enumerate(bz[:8])),
                  " ([5]/[6] key evaluation angle/conc, [0] min-max, [1] coarse histogram, [2] locate, [3] refinement histogram, [4] locate)")
        h2 = n // 2
        for nm, v in steps:
            print(f"    {nm:20s} {v.mean():8.1f}   first half {v[:h2].mean():8.1f}  second half {v[h2:].mean():8.1f}")
    t = t.copy()
    no_resweep = t[:, 4] == 0                           # markers 4 and 5 are only written on the resweep path
    t[no_resweep, 4] = t[no_resweep, 6]
    t[no_resweep, 5] = t[no_resweep, 6]
    d = np.diff(t, axis=1)
    names = ["sweep1 moments", "finish1 eig+brackets+box", "sweep2 merged", "finish2 M (+maxC)", "sweep3 conc (resweep)", "finish3 maxC", "sweep4 apply"]
    for i, nm in enumerate(names):
        print(f"  {nm:26s} {d[:, i].mean():9.1f} {np.median(d[:, i]):9.1f} {d[:, i].max():9.1f}")
    print("  total per tile           %9.1f" % (t[:, 7] - t[:, 0]).mean(), " kernel span %.1f us" % (t[:, 7].max() - t[:, 0].min()))
    t0 = t[:, 0] - t[:, 0].min()
    t7 = t[:, 7] - t[:, 0].min()
    print("  start of tiles: pct 0/10/50/90/100", np.percentile(t0, [0, 10, 50, 90, 100]).round(1), " end:", np.percentile(t7, [0, 10, 50, 90, 100]).round(1))
    tot = t[:, 7] - t[:, 0]
    print("  total by XCD (tile % 8): mean", " ".join(f"{tot[x::8].mean():.0f}" for x in range(8)), " max", " ".join(f"{tot[x::8].max():.0f}" for x in range(8)))
    sw = d[:, [0, 2, 6]]
    print("  correlation of the sweep times across tiles (s1-s2, s1-s4, s2-s4):", np.corrcoef(sw.T)[np.triu_indices(3, 1)].round(2))
    fin = tot - sw.sum(1)
    print("  finish total: mean %.1f min %.1f max %.1f ; corr(finish, sweeps) %.2f" % (fin.mean(), fin.min(), fin.max(), np.corrcoef(fin, sw.sum(1))[0, 1]))
    order = np.argsort(tot)
    print("  fastest 8 tiles", order[:8].tolist(), "slowest 8", order[-8:].tolist())
    h = n // 2
    print("  first half of the grid vs second half: total %.0f / %.0f   sweep1 %.0f / %.0f   sweep2 %.0f / %.0f   apply %.0f / %.0f   finish %.0f / %.0f" % (
        tot[:h].mean(), tot[h:].mean(), d[:h, 0].mean(), d[h:, 0].mean(), d[:h, 2].mean(), d[h:, 2].mean(), d[:h, 6].mean(), d[h:, 6].mean(), fin[:h].mean(), fin[h:].mean()))
