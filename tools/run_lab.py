"""Reinhard / LuminosityStandardizer on 1250 x 512^2 tiles a few times: rocprofv3 target."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from stainlib_amd import engine
from tools.synth import synth_tiles
t5 = synth_tiles(1250, 512, 512, seed=7)
o5 = torch.empty_like(t5)
tm, ts = np.array([60.0, 10.0, -5.0]), np.array([15.0, 6.0, 5.0])
for _ in range(3):
    engine.reinhard_transform(t5, tm, ts, out=o5)
    engine.luminosity_standardize(t5, out=o5)
torch.cuda.synchronize()
