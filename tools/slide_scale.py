"""configs[4] at the size of one GPU's shard: pooled slide statistics + apply by slide size (development aid).
    python tools/slide_scale.py [n,n,...]      (default 512,2048,8192,12500 tiles of 1024^2 = up to 39 GB in + 39 GB out)"""
import sys
import time

import torch

sys.path.insert(0, ".")
import stainlib_amd as sl  # noqa: E402
from stainlib_amd import engine  # noqa: E402
from stainlib_amd.distributed import PooledSlideStatistics, SlideNormalizer  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [512, 2048, 8192, 12500]
tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, _ = engine.macenko_fit(tgt)
nrm = sl.MacenkoNormalizer()
nrm.stain_matrix_target, nrm.maxC_target = Mt[0].cpu().numpy(), mct[0].cpu().numpy().reshape(1, 2)
nmax = max(sizes)
rgb_all = synth_tiles(nmax, 1024, 1024, seed=9)
out_all = torch.empty_like(rgb_all)
for n in sizes:
    rgb, out = rgb_all[:n], out_all[:n]
    sn = SlideNormalizer(nrm, group=False, mode="pooled")
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.25:             # spin-up: the clocks ramp for ~25 ms, and a 512-tile slide is 2 ms
        sn.transform_shard(rgb, out=out)
    torch.cuda.synchronize()
    reps = 20 if n <= 512 else (5 if n <= 2048 else 2)
    t0 = time.perf_counter()
    for _ in range(reps):
        sn.transform_shard(rgb, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    ms_g = None
    if n <= 2048:                                           # the same from a HIP graph (captured once per buffer pair)
        sng = SlideNormalizer(nrm, group=False, mode="pooled", graph=True)
        sng.transform_shard(rgb, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            sng.transform_shard(rgb, out=out)
        torch.cuda.synchronize()
        ms_g = (time.perf_counter() - t0) / reps * 1e3
        del sng
    # stage split: statistics alone, apply alone
    st = PooledSlideStatistics(group=False)
    st(rgb)                                                 # (untimed: the allocator may have to fetch the chain's workspace for this shape afresh)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    M, mc = st(rgb)
    torch.cuda.synchronize()
    ms_stats = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    engine.tile_moments(rgb)
    torch.cuda.synchronize()
    ms_mom = (time.perf_counter() - t0) * 1e3
    # the individual sweeps (event time): window sweeps at the windows the statistics ended on, one sampled pass
    import numpy as np
    from stainlib_amd import _ffi
    def ev(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize()
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    Vb = np.array([0.55, 0.2, 0.7, -0.6, 0.45, 0.75])
    f2o = lambda f: (lambda u: (~u & 0xffffffff) if u & 0x80000000 else (u | 0x80000000))(int(np.float32(f).view(np.uint32)))
    t_wa = ev(lambda: engine.slide_key_window(rgb, _ffi.KEYSET_ANGLE, Vb, (f2o(-0.35) - 32768, f2o(0.62) - 32768)))
    t_wc = ev(lambda: engine.slide_key_window(rgb, _ffi.KEYSET_CONC, M.reshape(6), (f2o(mc[0]) - 32768, f2o(mc[1]) - 32768)))
    t_sa = ev(lambda: engine.slide_key_histogram_sampled(rgb, _ffi.KEYSET_ANGLE, Vb, (0, 0), 0, 6))
    t_ap = ev(lambda: engine.normalize_apply(rgb, torch.as_tensor(M, device="cuda").expand(n, 2, 3).contiguous(),
                                             torch.as_tensor(mc, device="cuda").expand(n, 2).contiguous(), Mt[0], mct[0], out=out))
    print(f"      sweeps (event ms): moments {ev(lambda: engine.tile_moments(rgb)):.2f}  angle window {t_wa:.2f}  conc window {t_wc:.2f}  "
          f"one sampled pass {t_sa:.2f}  apply {t_ap:.2f}", flush=True)
    print(f"pooled slide of {n:6d} tiles: {ms:9.2f} ms -> {n / ms:8.1f} k tiles/s   statistics {ms_stats:8.2f} ms (moments sweep {ms_mom:7.2f}) "
          f"apply {ms - ms_stats:8.2f} ms   paths {sn.last_path}", flush=True)
    if ms_g is not None:
        print(f"      replayed from a HIP graph (SlideNormalizer(graph=True)): {ms_g:9.2f} ms -> {n / ms_g:8.1f} k tiles/s", flush=True)
    # the same slide through the three-sweep chain of rounds 3-5 (the fallback), and the steps of the one-sweep chain (event ms)
    sn3 = SlideNormalizer(nrm, group=False, mode="pooled", merged=False)
    sn3.transform_shard(rgb, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sn3.transform_shard(rgb, out=out)
    torch.cuda.synchronize()
    ms3 = (time.perf_counter() - t0) / reps * 1e3
    print(f"      three-sweep chain (merged=False): {ms3:9.2f} ms -> {n / ms3:8.1f} k tiles/s   paths {sn3.last_path}", flush=True)
    pst = PooledSlideStatistics(group=False)
    s_new = pst.enqueue_merged(rgb)
    slog = int(s_new[32].item())
    ws2 = engine.pool2_workspace(n, 1024, 1024, slog, rgb.device)
    hist = torch.zeros((_ffi.POOL2_HIST_WORDS,), dtype=torch.int64, device="cuda")
    shape = (n, 1024, 1024)
    prm = engine.make_params()
    s_tmp = s_new.clone()
    s_tmp[120] = 0.0
    t_s1 = ev(lambda: engine.pool2_sample(rgb, slog, ws2, params=prm))
    t_h = [ev(lambda k=k: engine.pool2_hist(0, k, 0, shape, slog, s_new, ws2, hist, params=prm)) for k in (0, 1)]
    t_sw = ev(lambda: engine.pool2_sweep(rgb, slog, s_new, ws2, params=prm))
    t_c = [ev(lambda k=k: engine.pool2_hist(1, k, 1, shape, slog, s_tmp, ws2, hist, params=prm)) for k in (0, 1)]
    t_chain = ev(lambda: pst.enqueue_merged(rgb))
    sv = s_new.cpu().numpy()
    print(f"      one-sweep chain (event ms): whole {t_chain:.3f} = sample {t_s1:.3f} + sample histograms {t_h[0]:.3f} {t_h[1]:.3f} + SWEEP {t_sw:.3f} + "
          f"candidate passes {t_c[0]:.3f} {t_c[1]:.3f} per level (two levels each) + decision steps;  sample 1 in {1 << slog} sub-rows, listed "
          f"{100 * sv[96] / max(sv[10], 1):.2f} % of the tissue (angle) {100 * sv[97] / (n * 1048576):.2f} % of the pixels (concentrations), workspace {ws2.numel() / 2**20:.0f} MiB", flush=True)
    del ws2
    # per-tile mode on the same tiles, for comparison (chunks of 512 through the fused kernel)
    if n >= 512:
        ws = engine.Workspace()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(0, n - n % 512, 512):
            engine.macenko_transform(rgb[i:i + 512], Mt[0], mct[0], out=out[i:i + 512], ws=ws)
        torch.cuda.synchronize()
        msf = (time.perf_counter() - t0) * 1e3
        print(f"      per-tile mode, chunks of 512: {msf:9.2f} ms -> {(n - n % 512) / msf:8.1f} k tiles/s", flush=True)
