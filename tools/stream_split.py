"""Does a mid-size batch gain from running its one-launch-per-phase chain as K sub-batches on K streams (their latency-bound finish
launches overlap each other and the other sub-batches' sweeps)?  Emulated from Python with torch streams; development aid."""
import sys, time, torch
sys.path.insert(0, ".")
from stainlib_amd import engine
from tools.synth import synth_tiles
tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, st = engine.macenko_fit(tgt)
Mt, mct = Mt[0].contiguous(), mct[0].contiguous()
for n in (32, 64, 96, 128, 192, 256, 320):
    rgb = synth_tiles(n, 1024, 1024, seed=3)
    out = torch.empty_like(rgb)
    res = []
    for K in (1, 2, 3, 4, 6):
        if n % K:
            res.append(float("nan")); continue
        streams = [torch.cuda.Stream() for _ in range(K)]
        wss = [engine.Workspace() for _ in range(K)]
        p = engine.make_params(schedule=1)
        m = n // K
        def go():
            cur = torch.cuda.current_stream()
            for k, s in enumerate(streams):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    engine.macenko_transform(rgb[k * m:(k + 1) * m], Mt, mct, params=p, out=out[k * m:(k + 1) * m], ws=wss[k])
            for s in streams:
                cur.wait_stream(s)
        for _ in range(3):
            go()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            go()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"n {n:4d}: " + "  ".join(f"K={K} {r:.3f} ms" for K, r in zip((1, 2, 3, 4, 6), res)), flush=True)
    del rgb, out
