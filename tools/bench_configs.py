"""Secondary BASELINE.json configs on one GPU (development / reporting aid; bench.py is the contract):
  cfg3  128 tiles 1024x1024, Vahadane transform
  cfg4  HED-lighter + StainAugmentor.pop over 512x512 tiles (1250 per GPU)
  cfg5  slide-level Macenko (per-tile fit, stats gather (world 1), shared apply)
Prints one JSON object."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import stainlib_amd as sl  # noqa: E402
from stainlib_amd import engine  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402
from stainlib_amd.distributed import SlideNormalizer  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


res = {}
tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])

# cfg3 ------------------------------------------------------------------------------------------
rgb = synth_tiles(128, 1024, 1024, seed=5)
out = torch.empty_like(rgb)
Mt, mct, st, sw = engine.vahadane_fit(tgt, params=engine.make_params(dl_tol=1e-6, dl_max_sweeps=100))
p = engine.make_params(dl_tol=1e-6, dl_max_sweeps=100)
t = timeit(lambda: engine.vahadane_transform(rgb, Mt[0], mct[0], params=p, out=out))
_, _, _, sweeps = engine.vahadane_fit(rgb, params=p)
res["cfg3_vahadane_128x1024"] = {"tiles_per_s": 128 / t, "ms": 1e3 * t, "mean_dictionary_sweeps": float(sweeps.float().mean()),
                                  "max_dictionary_sweeps": int(sweeps.max()), "tol": 1e-6}
rgb512 = synth_tiles(512, 1024, 1024, seed=6)
out512 = torch.empty_like(rgb512)
t = timeit(lambda: engine.vahadane_transform(rgb512, Mt[0], mct[0], params=p, out=out512), reps=3, warm=1)
res["vahadane_512x1024"] = {"tiles_per_s": 512 / t, "ms": 1e3 * t}

# cfg5 (one rank's share, scaled down) ------------------------------------------------------------
n = sl.MacenkoNormalizer()
n.fit(tgt[0].cpu().numpy())
sn = SlideNormalizer(n)
t = timeit(lambda: sn.transform_shard(rgb512, out=out512), reps=5, warm=2)
res["cfg5_slide_mode_512x1024"] = {"tiles_per_s": 512 / t, "ms": 1e3 * t}
snp = SlideNormalizer(n, mode="pooled")
t = timeit(lambda: snp.transform_shard(rgb512, out=out512), reps=3, warm=1)
res["cfg5_slide_mode_pooled_512x1024"] = {"tiles_per_s": 512 / t, "ms": 1e3 * t,
                                          "note": "exact statistics of the concatenated slide: 1 moment sweep + 2 window sweeps (+ 6 over a 1/64 sample) + apply"}
del rgb512, out512, rgb, out

# cfg4 ------------------------------------------------------------------------------------------
N = 1250
t5 = synth_tiles(N, 512, 512, seed=7)
o5 = torch.empty_like(t5)
a = sl.HedLighterColorAugmenter()
np.random.seed(0)
sig, bia = a.randomize_batch(N)
sig_d = torch.as_tensor(sig, device="cuda")
bia_d = torch.as_tensor(bia, device="cuda")
t = timeit(lambda: engine.hed_augment(t5, sig_d, bia_d, out=o5), reps=10)
gbs = N * 512 * 512 * 6 / t / 1e9
res["cfg4_hed_lighter_1250x512"] = {"tiles_per_s": N / t, "ms": 1e3 * t, "GBps_6Bpx": gbs, "frac_hbm_8TBs": gbs / 8000}
M, mc, st = engine.macenko_fit(t5)
ab = torch.rand((N, 4), device="cuda") * torch.tensor([0.4, 0.4, 0.4, 0.4], device="cuda") + torch.tensor([0.8, -0.2, 0.8, -0.2], device="cuda")
t = timeit(lambda: engine.stain_augment(t5, M, ab, out=o5), reps=10)
gbs = N * 512 * 512 * 6 / t / 1e9
res["cfg4_stain_augmentor_pop_1250x512"] = {"tiles_per_s": N / t, "ms": 1e3 * t, "GBps_6Bpx": gbs, "frac_hbm_8TBs": gbs / 8000}
t = timeit(lambda: engine.macenko_fit(t5), reps=5)
res["macenko_fit_1250x512"] = {"tiles_per_s": N / t, "ms": 1e3 * t}
print(json.dumps(res))
