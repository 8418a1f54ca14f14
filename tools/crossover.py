import sys, time, torch
sys.path.insert(0, ".")
from stainlib_amd import engine
from tools.synth import synth_tiles
tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, st = engine.macenko_fit(tgt)
for size in (1024, 256):
    for n in (16, 32, 64, 128, 256, 512):
        rgb = synth_tiles(n, size, size, seed=3)
        out = torch.empty_like(rgb)
        r = []
        for sched in (1, 2):
            p = engine.make_params(schedule=sched)
            for _ in range(3):
                engine.macenko_transform(rgb, Mt[0], mct[0], params=p, out=out)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                engine.macenko_transform(rgb, Mt[0], mct[0], params=p, out=out)
            torch.cuda.synchronize(); r.append((time.perf_counter() - t0) / 10 * 1e3)
        print(f"size {size} n {n:4d}: per-phase {r[0]:.3f} ms  fused {r[1]:.3f} ms")
