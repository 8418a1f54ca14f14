"""A few launches of the Macenko transform / apply / copy on n tiles: rocprofv3 target.
    python tools/run_fused_once.py [n] [schedule] [two_sweep]     schedule 0 automatic (default), 1 one launch per phase, 2 fused;
    two_sweep: SlParams.two_sweep (0 automatic, 1 off = three sweeps, 2 every tile tries); a fourth argument "fit" adds three launches of
    the fused FIT kernel (k_fused<.., false, ..>: no output -- what it writes is lists, sample and scratch)"""
import sys
import torch
sys.path.insert(0, ".")
from stainlib_amd import engine
from tools.synth import synth_tiles
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sched = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ts = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rgb = synth_tiles(n, 1024, 1024, seed=3)
tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, st = engine.macenko_fit(tgt)
out = torch.empty_like(rgb)
for _ in range(3):
    o, M, mc, s = engine.macenko_transform(rgb, Mt[0], mct[0], out=out, params=engine.make_params(schedule=sched, two_sweep=ts))
if len(sys.argv) > 4 and sys.argv[4] == "fit":
    for _ in range(3):
        engine.macenko_fit(rgb, params=engine.make_params(schedule=2, two_sweep=ts))
for _ in range(3):
    engine.normalize_apply(rgb, M, mc, Mt[0], mct[0], out=out)
for _ in range(3):
    out.copy_(rgb)          # calibration: a plain 1.6 GB read + 1.6 GB write
torch.cuda.synchronize()
