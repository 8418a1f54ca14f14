"""The headline transform on batches that are NOT i.i.d. pixels: 512 tiles of 1024^2 cycled from four oracle.structured_tile
tiles of each kind (spatially smooth 'blobs', 'white_bg', 'quantized', 'palette12'), with the count of order statistics that
needed the exact fallback.  The i.i.d. bench batch is the best case for the sample brackets; this is the other side."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from stainlib_amd import engine  # noqa: E402
from oracle import stain_oracle as so  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402


def med(fn, reps=8):
    """ms per launch, timed like bench.py's timed region: `reps` launches back to back between ONE pair of events on the launch stream,
    after a spin-up (one launch per host synchronisation, as this tool did until round 5, measures the cold-clock latency of a launch:
    1.50 ms where the bench reads 1.37)."""
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts))


tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, _ = engine.macenko_fit(tgt)
n = 512
ws = engine.Workspace()
iid = synth_tiles(n, 1024, 1024, seed=7)
out = torch.empty_like(iid)
print(f"i.i.d.      {med(lambda: engine.macenko_transform(iid, Mt[0], mct[0], out=out, ws=ws)):.3f} ms per {n} tiles")
def ihc_four():
    import os
    I = np.load(os.path.join("tests", "golden", "tissue_ihc_512.npz"))["input"]
    row = np.concatenate([I, I[:, ::-1]], axis=1)
    T = np.ascontiguousarray(np.concatenate([row, row[::-1]], axis=0))
    return np.stack([T, np.ascontiguousarray(np.roll(T, 301, axis=0)), np.ascontiguousarray(np.roll(T, 517, axis=1)), np.ascontiguousarray(T.transpose(1, 0, 2))])


for kind in ("blobs", "white_bg", "quantized", "palette12", "real_tissue_ihc"):
    base = torch.as_tensor(ihc_four() if kind == "real_tissue_ihc" else np.stack([so.structured_tile(kind, 1024, 1024, 20 + s) for s in range(4)]), device="cuda")
    rgb = base[torch.arange(n, device="cuda") % 4].contiguous()
    p = engine.make_params()
    fb = engine.attach_fallbacks(p, n)
    rs = torch.zeros((n,), dtype=torch.int32, device="cuda")
    p.resweeps_out = rs.data_ptr()
    t = med(lambda: engine.macenko_transform(rgb, Mt[0], mct[0], params=p, out=out, ws=ws))
    o, M, mc, st = engine.macenko_transform(rgb, Mt[0], mct[0], params=p, out=out, ws=ws)
    torch.cuda.synchronize()
    Mo = so.macenko_stain_matrix(base[0].cpu().numpy())
    print(f"{kind:10s}  {t:.3f} ms per {n} tiles = {n / t:.1f} k tiles/s   failed tiles {int((st != 0).sum())}   exact fallbacks {int(fb.sum())} of {4 * n}   resweeps {int((rs != 0).sum())} (reasons 1..4: {[int((rs == k).sum()) for k in (1, 2, 3, 4)]})"
          f"   |M - oracle| tile 0 {np.abs(M[0].cpu().numpy() - Mo).max():.1e}", flush=True)
