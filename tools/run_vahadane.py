"""A few Vahadane transforms of n tiles 1024^2 (bench parameters): rocprofv3 target.   python tools/run_vahadane.py [n=128]"""
import sys
import torch
sys.path.insert(0, ".")
from stainlib_amd import engine
from tools.synth import synth_tiles
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rgb = synth_tiles(n, 1024, 1024, seed=5)
tgt = synth_tiles(1, 1024, 1024, seed=1001)
out = torch.empty_like(rgb)
p = engine.make_params(dl_tol=1e-6, dl_max_sweeps=100)
Mt, mct, _, _ = engine.vahadane_fit(tgt, params=p)
for _ in range(10):
    engine.vahadane_transform(rgb, Mt[0], mct[0], params=p, out=out)
torch.cuda.synchronize()
