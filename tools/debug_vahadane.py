import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from stainlib_amd import engine
from tools.synth import synth_tiles
rgb = synth_tiles(128, 1024, 1024, seed=5)
p = engine.make_params(dl_tol=1e-6, dl_max_sweeps=100)
M, mc, st, sw = engine.vahadane_fit(rgb, params=p)
sw = sw.cpu().numpy()
print("sweeps histogram:", np.bincount(sw)[:40], "n at cap:", (sw >= 100).sum())
bad = int(np.argmax(sw))
t = rgb[bad:bad + 1].contiguous()
prev = None
for k in list(range(4, 24)) + [40, 41, 42, 43, 99, 100]:
    Mk, _, _, s = engine.vahadane_fit(t, params=engine.make_params(dl_tol=0.0, dl_max_sweeps=k))
    Mk = Mk.cpu().numpy()[0]
    if prev is not None:
        print(k, "sweeps", int(s[0]), "delta vs prev k: %.3e" % np.abs(Mk - prev).max())
    prev = Mk
