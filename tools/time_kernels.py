"""Quick timings of the main kernels for A/B runs (development aid):
    for lib in A B A B; do STAINLIB_HIP_LIB=$PWD/stainlib_amd/csrc/libstainlib_hip_$lib.so python tools/time_kernels.py; done
prints one line: fused Macenko transform (512 x 1024^2), k_apply, per-phase transform at 128 tiles, StainAugmentor.pop and
HED (1250 x 512^2), Vahadane transform (128 and 512 x 1024^2), and with `lab` the Lab family on 1250 x 512^2 (Reinhard
transform, its statistics sweeps alone, LuminosityStandardizer, both plain conversions); milliseconds, median of `reps` launches."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from stainlib_amd import _ffi, engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

what = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fused", "apply", "phase128", "aug", "hed", "vah128", "vah512"]
reps = 15


def med(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, _ = engine.macenko_fit(tgt)
rgb = synth_tiles(512, 1024, 1024, seed=7)
out = torch.empty_like(rgb)
ws = engine.Workspace()
res = {}
o, M, mc, st = engine.macenko_transform(rgb, Mt[0], mct[0], out=out, ws=ws)
if "fused" in what:
    res["fused512"] = med(lambda: engine.macenko_transform(rgb, Mt[0], mct[0], out=out, ws=ws))
if "apply" in what:
    res["apply512"] = med(lambda: engine.normalize_apply(rgb, M, mc, Mt[0], mct[0], out=out))
if "phase128" in what:
    res["phase128"] = med(lambda: engine.macenko_transform(rgb[:128], Mt[0], mct[0], out=out[:128], ws=ws))
    res["phase256"] = med(lambda: engine.macenko_transform(rgb[:256], Mt[0], mct[0], out=out[:256], ws=ws))
if "vah128" in what:
    p = engine.make_params(dl_tol=1e-6, dl_max_sweeps=100)
    res["vah128"] = med(lambda: engine.vahadane_transform(rgb[:128], Mt[0], mct[0], params=p, out=out[:128], ws=ws))
if "vah512" in what:
    p = engine.make_params(dl_tol=1e-6, dl_max_sweeps=100)
    res["vah512"] = med(lambda: engine.vahadane_transform(rgb, Mt[0], mct[0], params=p, out=out, ws=ws))
if "aug" in what or "hed" in what:
    t5 = rgb.view(-1, 512, 512, 3)[:1250]
    o5 = out.view(-1, 512, 512, 3)[:1250]
    M5, _, _ = engine.macenko_fit(t5, ws=ws)
    ab = np.tile(np.array([[1.1, 0.05, 0.9, -0.05]]), (1250, 1))
    sg = np.tile(np.array([[0.01, -0.02, 0.015]]), (1250, 1))
    if "aug" in what:
        res["aug1250"] = med(lambda: engine.stain_augment(t5, M5, ab, out=o5))
    if "hed" in what:
        res["hed1250"] = med(lambda: engine.hed_augment(t5, sg, sg, out=o5, ws=ws))
if "lab" in what:
    t5 = rgb.view(-1, 512, 512, 3)[:1250]
    o5 = out.view(-1, 512, 512, 3)[:1250]
    tm, ts = np.array([60.0, 12.0, -8.0]), np.array([18.0, 6.0, 5.0])
    res["reinhard1250"] = med(lambda: engine.reinhard_transform(t5, tm, ts, out=o5, ws=ws))
    res["reinhard_stats1250"] = med(lambda: engine.reinhard_stats(t5, ws=ws))
    res["luminosity1250"] = med(lambda: engine.luminosity_standardize(t5, 95, out=o5, ws=ws))
    res["rgb2lab1250"] = med(lambda: engine.rgb_to_lab8(t5))
    res["lab2rgb1250"] = med(lambda: engine.lab8_to_rgb(t5))
print(os.path.basename(_ffi.LIB_PATH), " ".join(f"{k} {v:.3f}" for k, v in res.items()))
