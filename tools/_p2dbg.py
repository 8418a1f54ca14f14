import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import stain_oracle as so
from tests.gpu_util import to_dev
from stainlib_amd import engine, _ffi
np.set_printoptions(precision=7, linewidth=220)
def ord2f(o):
    o = np.uint32(o)
    u = np.where(o & np.uint32(0x80000000), o ^ np.uint32(0x80000000), ~o).astype(np.uint32)
    return u.view(np.float32)
def run(label, tiles):
    dev = to_dev(tiles)
    n, h, w, _ = dev.shape
    params = engine.make_params()
    slog = 0
    ws = engine.pool2_workspace(n, h, w, slog, dev.device)
    shape = (n, h, w)
    hists = torch.empty((8, _ffi.POOL2_HIST_WORDS), dtype=torch.int64, device="cuda")
    mom = engine.pool2_sample(dev, slog, ws, params=params)
    state = engine.pool2_begin(mom, slog, params=params)
    for i, ks in enumerate((0, 1)):
        engine.pool2_bands(state, ks, engine.pool2_hist(0, ks, 0, shape, slog, state, ws, hists[i], params=params))
    tot = engine.pool2_sweep(dev, slog, state, ws, params=params)
    print("==", label, "totals", tot.cpu().numpy()[[0, 12, 13]])
    engine.pool2_exact(tot, state)
    s = state.cpu().numpy()
    print("  after exact: miss", s[9], "K", s[24:26], "Br", s[100:104], "winlo", s[240:244], ord2f(np.array(s[240:244], dtype=np.uint64).astype(np.uint32)), "sh", s[244:248], "brk", s[60:64])
    for ks in (0, 1):
        for level in range(3):
            hh = engine.pool2_hist(1, ks, 1, shape, slog, state, ws, hists[2 + 3 * ks + level], params=params)
            H = hh.cpu().numpy()
            tails = H[:256].reshape(32, 8).sum(axis=0)
            b0, b1 = H[256:256 + 8192], H[256 + 8192:]
            nz0, nz1 = np.nonzero(b0)[0], np.nonzero(b1)[0]
            print(f"  keyset {ks} level {level}: tails {tails} bins0 sum {b0.sum()} range {nz0[[0,-1]] if len(nz0) else None} bins1 sum {b1.sum()} range {nz1[[0,-1]] if len(nz1) else None}")
            engine.pool2_step(state, ks, hh)
            s = state.cpu().numpy()
            print("     -> miss", s[9], "done", s[120], "level", s[121], "winlo", s[240:244], "sh", s[244:248], "res", s[110:114], "sub", s[28:30], "K", s[24:26], "na nc", s[96:98])
run("iid 1x48x128", [so.synth_tile(48, 128, 5)])
run("iid 1x128x128", [so.synth_tile(128, 128, 5)])
run("iid 1x200x200", [so.synth_tile(200, 200, 5)])
