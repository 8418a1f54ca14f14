"""The one-sweep pooled chain (sl_pool2_*) against the three-sweep chain on slides of several compositions (development aid).
    python tools/pool2_check.py [n_tiles [side]]
Prints, per slide: which route settled the statistics, |dM|, |dmaxC| between the two chains, the state's diagnostics (sample
density, tilt bound, candidate share, why / miss codes) and event times of the chain's steps."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stainlib_amd import _ffi, engine  # noqa: E402
from stainlib_amd.distributed import PooledSlideStatistics  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
side = int(sys.argv[2]) if len(sys.argv) > 2 else 1024


def tissue_tiles(n, side):
    """real stained tissue (the ihc fixture of tests/golden) mirror-tiled to side x side, shifted per tile"""
    img = np.load("tests/golden/tissue_ihc_512.npz")["input"]
    img = np.concatenate([img, img[:, ::-1]], axis=1)
    img = np.concatenate([img, img[::-1]], axis=0)
    reps = (side + img.shape[0] - 1) // img.shape[0] + 1
    big = np.tile(img, (reps, reps, 1))
    t = torch.from_numpy(big).cuda()
    out = torch.empty((n, side, side, 3), dtype=torch.uint8, device="cuda")
    for i in range(n):
        oy, ox = (37 * i) % img.shape[0], (101 * i) % img.shape[1]
        out[i] = t[oy:oy + side, ox:ox + side]
    return out


def smooth_tiles(n, side, seed=5):
    """spatially smooth tiles: low-pass filtered concentration fields (what the 'blobs' fixture stands for)"""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    M = torch.tensor([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]], device="cuda")
    M = M / M.norm(dim=1, keepdim=True)
    out = torch.empty((n, side, side, 3), dtype=torch.uint8, device="cuda")
    for i in range(n):
        c = torch.rand((1, 2, side // 16 + 1, side // 16 + 1), generator=g, device="cuda")
        c = torch.nn.functional.interpolate(c, size=(side, side), mode="bicubic", align_corners=False)[0].clamp_min(0)
        c = (c ** 2) * 2.2 + 0.02 * torch.rand((2, side, side), generator=g, device="cuda")
        od = torch.einsum("khw,kc->hwc", c, M) + 0.004 * torch.randn((side, side, 3), generator=g, device="cuda")
        out[i] = (255.0 * torch.exp(-od)).clamp_(0, 255).to(torch.uint8)
    return out


def ev(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


slides = {"iid": lambda: synth_tiles(n, side, side, seed=9)}
try:
    np.load("tests/golden/tissue_ihc_512.npz")
    slides["tissue"] = lambda: tissue_tiles(n, side)
except Exception as e:  # noqa: BLE001
    print("no tissue fixture:", e)
slides["smooth"] = lambda: smooth_tiles(n, side)


def structured(kind):
    """four structured tiles of the oracle's generator (heavy ties, saturated background), cycled"""
    from oracle import stain_oracle as so
    four = np.stack([so.structured_tile(kind, side, side, 20 + s) for s in range(4)])
    return torch.as_tensor(four, device="cuda")[torch.arange(n, device="cuda") % 4].contiguous()


for kind in ("quantized", "palette12", "white_bg"):
    slides[kind] = lambda kind=kind: structured(kind)

for name, make in slides.items():
    rgb = make()
    st = PooledSlideStatistics(group=False)
    s_old = st.enqueue(rgb)
    got_old = st.finish(s_old)
    path_old = list(st.last_path)
    s_new = st.enqueue_merged(rgb)
    got_new = st.finish(s_new)
    s = s_new.cpu().numpy()
    npx = n * side * side
    print(f"== {name}: {n} tiles of {side}^2, slog {int(s[32])}  sample {int(s[31])} px ({int(s[30])} tissue)  tau {s[49]:.2e}  why {int(s[33])} "
          f"miss {int(s[9])} status {int(s[8])}")
    print(f"   listed: angle {int(s[96])} ({100 * s[96] / max(s[10], 1):.2f} % of tissue)  conc {int(s[97])} ({100 * s[97] / npx:.2f} % of px)  "
          f"overflow {int(s[98])}  plane cmax {s[116]:.2e} (k1 {s[50]:.2e}) dn {s[117]:.2e}")
    print(f"   brackets(sample) {s[60:64]}  exact {s[100:104]}  res {s[110:114]}")
    if got_old is not None and got_new is not None:
        print(f"   old path {path_old}  new path {st.last_path}   |dM| {np.abs(got_old[0] - got_new[0]).max():.2e}  |dmaxC|/maxC "
              f"{(np.abs(got_old[1] - got_new[1]) / got_old[1]).max():.2e}")
    else:
        print(f"   old {'ok' if got_old is not None else 'MISS'} new {'ok' if got_new is not None else 'MISS'}")
    print(f"   M {None if got_new is None else got_new[0].round(5).tolist()} maxC {None if got_new is None else got_new[1].round(5).tolist()}")
    t_old = ev(lambda: st.enqueue(rgb))
    t_new = ev(lambda: st.enqueue_merged(rgb))
    print(f"   chain (event ms): three-sweep {t_old:.3f}   one-sweep {t_new:.3f}")
    # the steps of the new chain
    params = engine.make_params()
    slog = int(s[32])
    ws = engine.pool2_workspace(n, side, side, slog, rgb.device)
    hist = torch.zeros((_ffi.POOL2_HIST_WORDS,), dtype=torch.int64, device="cuda")
    shape = (n, side, side)
    t_s1 = ev(lambda: engine.pool2_sample(rgb, slog, ws, params=params))
    t_h0 = ev(lambda: engine.pool2_hist(0, 0, 0, shape, slog, s_new, ws, hist, params=params))
    t_h1 = ev(lambda: engine.pool2_hist(0, 1, 0, shape, slog, s_new, ws, hist, params=params))
    t_sw = ev(lambda: engine.pool2_sweep(rgb, slog, s_new, ws, params=params))
    s_tmp = s_new.clone(); s_tmp[120] = 0.0       # (not settled: the passes run)
    t_c = [ev(lambda k=k: engine.pool2_hist(1, k, 1, shape, slog, s_tmp, ws, hist, params=params)) for k in (0, 1)]
    t_mom = ev(lambda: engine.tile_moments(rgb))
    print(f"   steps: sample {t_s1:.3f}  sample hists {t_h0:.3f} {t_h1:.3f}  SWEEP {t_sw:.3f} (moments sweep alone {t_mom:.3f})  "
          f"candidate passes {' '.join(f'{x:.3f}' for x in t_c)}   workspace {ws.numel() / 2**20:.0f} MiB")
    del rgb
