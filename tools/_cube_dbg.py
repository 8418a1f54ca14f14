import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from oracle import stain_oracle as so
from stainlib_amd import engine
from tools.synth import synth_tiles
dev = torch.device("cuda", 0)
tgt = synth_tiles(1, 1024, 1024, seed=1, device=dev, M_true=so.M_TRUE_TGT.tolist())
Mt, mct, _ = engine.macenko_fit(tgt)
rgb = synth_tiles(8, 1024, 1024, seed=7, device=dev)
for pf in (2,):
    p = engine.make_params(schedule=2, prefilter=pf)
    rsw = torch.zeros((8,), dtype=torch.int32, device=dev); cub = torch.zeros((8,), dtype=torch.int32, device=dev)
    p.resweeps_out, p.prefilter_out = rsw.data_ptr(), cub.data_ptr()
    o, M, mc, st = engine.macenko_transform(rgb, Mt[0], mct[0], params=p)
    print(pf, rsw.tolist(), [(c & 1, c >> 8) for c in cub.tolist()], st.tolist())

for kind in ("ihc","blobs","white_bg"):
    import numpy as np
    if kind=="ihc":
        I = np.load('/root/repo/tests/golden/tissue_ihc_512.npz')["input"]
        row = np.concatenate([I, I[:, ::-1]], axis=1); T = np.ascontiguousarray(np.concatenate([row, row[::-1]], axis=0))
        t = torch.as_tensor(T[None], device=dev)
    else:
        t = torch.as_tensor(so.structured_tile(kind,1024,1024,20)[None], device=dev)
    p = engine.make_params(schedule=2, prefilter=2)
    cub = torch.zeros((1,), dtype=torch.int32, device=dev); p.prefilter_out = cub.data_ptr()
    engine.macenko_transform(t, Mt[0], mct[0], params=p)
    print(kind, [(c & 1, c >> 8) for c in cub.tolist()])
