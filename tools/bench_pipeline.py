"""PCIe-inclusive rate of the normalizer (host uint8 in -> host uint8 out), SURVEY 8f-1."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import stainlib_amd as sl
from stainlib_amd import engine
from tools.synth import synth_tiles
from stainlib_amd.pipeline import normalizer_pipeline
B, nb = 128, 12
n = sl.MacenkoNormalizer()
n.fit(synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])[0].cpu().numpy())
host = synth_tiles(B, 1024, 1024, seed=9).cpu().numpy()
pipe = normalizer_pipeline(n, (B, 1024, 1024, 3))
res = {}
for name, src in (("pageable_numpy_in", host), ("pinned_in", torch.from_numpy(host).pin_memory())):
    for _ in pipe.run([src] * 3):
        pass
    t0 = time.perf_counter()
    for _ in pipe.run([src] * nb):
        pass
    t = time.perf_counter() - t0
    gb = nb * B * 1024 * 1024 * 3 / 1e9
    res[name] = {"pcie_inclusive_tiles_per_s": nb * B / t, "GBps_each_direction": gb / t}
res.update(batch=B, batches=nb)
print(json.dumps(res))
