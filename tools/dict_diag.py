"""Development aid (library built with -DSL_EXP_DICT_DIAG: make -C stainlib_amd/csrc variant V=diag VFLAGS=-DSL_EXP_DICT_DIAG):
per-tile full sweeps, sample iterations and rejected steps of the Vahadane dictionary on the bench's synthetic tiles."""
import sys
import numpy as np
sys.path.insert(0, ".")
from stainlib_amd import engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402
rgb = synth_tiles(128, 1024, 1024, seed=7)
p = engine.make_params(dl_tol=1e-6, dl_max_sweeps=100, schedule=1)
M, mc, st, sw = engine.vahadane_fit(rgb, params=p)
sw = sw.cpu().numpy()
print("full sweeps", np.bincount(sw % 100), "sample iterations", np.bincount((sw // 100) % 100), "rejected", np.bincount(sw // 10000))
