"""The two-sweep schedule of the fused Macenko kernel against the three-sweep one, on the GPU: bytes, statistics and status must be identical
for every SlParams.two_sweep mode (0 automatic, 2 forced, 3 forced + failing plane check, 4 forced + tilted sample plane); prints what became
of each tile's attempt (SL_TWOSWEEP_*), resweeps, fallbacks, and the time per batch.
    python tools/ts_check.py [n_tiles=512] [size=1024] [kinds=iid,white_bg,quantized,blobs,ihc,palette12]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stainlib_amd import engine  # noqa: E402
from oracle import stain_oracle as so  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kinds = (sys.argv[3] if len(sys.argv) > 3 else "iid,white_bg,quantized,blobs,ihc,palette12").split(",")
REPS = int(os.environ.get("TS_REPS", "6"))


def spin(fn, ms=150.0):
    """sustained load until the clocks have ramped (~25 ms after an idle period)"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while True:
        for _ in range(8):
            fn()
        e1.record(); torch.cuda.synchronize()
        if e0.elapsed_time(e1) >= ms:
            return


def med_interleaved(fns, reps=REPS):
    """median ms of each fn, measured in alternation (A B C A B C ...) after one common spin-up: same clocks, same box state"""
    spin(fns[0])
    ts = [[] for _ in fns]
    for _ in range(reps):
        for i, fn in enumerate(fns):
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
            ts[i].append(e0.elapsed_time(e1) / 3.0)
    return [float(np.median(t)) for t in ts]


def ihc_four(size):
    I = np.load(os.path.join("tests", "golden", "tissue_ihc_512.npz"))["input"]
    row = np.concatenate([I, I[:, ::-1]], axis=1)
    T = np.ascontiguousarray(np.concatenate([row, row[::-1]], axis=0))
    four = [T, np.roll(T, 301, axis=0), np.roll(T, 517, axis=1), T.transpose(1, 0, 2)]
    return np.stack([np.ascontiguousarray(x[:size, :size]) for x in four])


def batch(kind):
    if kind == "iid":
        return synth_tiles(n, size, size, seed=7)
    base = torch.as_tensor(ihc_four(size) if kind == "ihc" else np.stack([so.structured_tile(kind, size, size, 20 + s) for s in range(4)]), device="cuda")
    return base[torch.arange(n, device="cuda") % 4].contiguous()


tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, _ = engine.macenko_fit(tgt)
ws = engine.Workspace()
bad = 0
for kind in kinds:
    rgb = batch(kind)
    ref = None
    modes = ((1, "three-sweep"), (0, "automatic"), (2, "forced"), (3, "forced, check fails"), (4, "forced, tilted plane"))
    runs = []
    for mode, name in modes:
        p = engine.make_params(schedule=2, two_sweep=mode)
        fb = engine.attach_fallbacks(p, n)
        rs = torch.zeros((n,), dtype=torch.int32, device="cuda")
        tso = torch.full((n,), 99, dtype=torch.int32, device="cuda")
        pf = torch.zeros((n,), dtype=torch.int32, device="cuda")
        p.resweeps_out = rs.data_ptr()
        p.twosweep_out = tso.data_ptr()
        p.prefilter_out = pf.data_ptr()
        runs.append(dict(p=p, fb=fb, rs=rs, tso=tso, pf=pf, out=torch.empty_like(rgb)))
    times = med_interleaved([(lambda r=r: engine.macenko_transform(rgb, Mt[0], mct[0], params=r["p"], out=r["out"], ws=ws)) for r in runs])
    for (mode, name), r, t in zip(modes, runs, times):
        fb, rs, tso, pf = r["fb"], r["rs"], r["tso"], r["pf"]
        o, M, mc, st = engine.macenko_transform(rgb, Mt[0], mct[0], params=r["p"], out=r["out"], ws=ws)
        torch.cuda.synchronize()
        got = (o.clone(), M.clone(), mc.clone(), st.clone())
        same = ""
        if ref is None:
            ref = got
        else:
            eq = [bool(torch.equal(a, b)) for a, b in zip(got[:1] + got[3:], ref[:1] + ref[3:])]
            eqM = bool(torch.equal(torch.nan_to_num(got[1], nan=-7.0), torch.nan_to_num(ref[1], nan=-7.0)))
            eqC = bool(torch.equal(torch.nan_to_num(got[2], nan=-7.0), torch.nan_to_num(ref[2], nan=-7.0)))
            ok = all(eq) and eqM and eqC
            if not ok:
                bad += 1
                nb = int((got[0] != ref[0]).sum())
                dM = float(torch.nan_to_num(got[1] - ref[1]).abs().max())
                dC = float(torch.nan_to_num(got[2] - ref[2]).abs().max())
                same = f"  *** DIFFERS: bytes {nb}, |dM| {dM:.2e}, |dmaxC| {dC:.2e}, status equal {eq[1]}"
            else:
                same = "  identical"
        hist = {int(k): int((tso == k).sum()) for k in torch.unique(tso).tolist()}
        share = (pf >> 8).float().mean().item()
        why = {int(k): int((rs == k).sum()) for k in torch.unique(rs).tolist() if k != 0}
        print(f"{kind:10s} {name:22s} {t:7.3f} ms per {n} tiles = {n / t:6.1f} k tiles/s  twosweep {hist}  resweeps {why}  fallbacks {int(fb.sum())}"
              f"  failed {int((st != 0).sum())}  cube share {share:4.1f} %{same}", flush=True)
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
