import sys, torch
sys.path.insert(0, ".")
from stainlib_amd import engine
from tools.synth import synth_tiles
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rgb = synth_tiles(n, 1024, 1024, seed=3)
tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, st = engine.macenko_fit(tgt)
out = torch.empty_like(rgb)
p = engine.make_params(schedule=1)
for _ in range(30):
    engine.macenko_transform(rgb, Mt[0], mct[0], out=out, params=p)
torch.cuda.synchronize()
