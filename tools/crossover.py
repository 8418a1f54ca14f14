"""Fused vs one-launch-per-phase schedule by batch and tile size, Macenko and Vahadane, with the automatic schedule beside them (where the
switch and the split of a batch beyond the resident grid should sit).  python tools/crossover.py [methods] [sizes] [batch sizes]"""
import sys, time, torch
sys.path.insert(0, ".")
from stainlib_amd import engine
from tools.synth import synth_tiles
tgt = synth_tiles(1, 1024, 1024, seed=1, M_true=[[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]])
Mt, mct, st = engine.macenko_fit(tgt)
methods = sys.argv[1].split(",") if len(sys.argv) > 1 else ["macenko", "vahadane"]
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1024, 256]
for method in methods:
    fn = engine.macenko_transform if method == "macenko" else engine.vahadane_transform
    for size in sizes:
        counts = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 448, 512, 640, 768, 1024)
        for n in counts:
            rgb = synth_tiles(n, size, size, seed=3)
            out = torch.empty_like(rgb)
            r = []
            scheds = (1, 2, 0) + ((3,) if method == "macenko" and n <= 256 else ())
            for sched in scheds:
                p = engine.make_params(schedule=sched, dl_tol=1e-6, dl_max_sweeps=100)
                for _ in range(3):
                    fn(rgb, Mt[0], mct[0], params=p, out=out)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10):
                    fn(rgb, Mt[0], mct[0], params=p, out=out)
                torch.cuda.synchronize(); r.append((time.perf_counter() - t0) / 10 * 1e3)
            wide = f"  fused-1024 {r[3]:.3f} ms" if len(r) > 3 else ""
            print(f"{method} size {size} n {n:4d}: per-phase {r[0]:.3f} ms  fused {r[1]:.3f} ms{wide}  automatic {r[2]:.3f} ms -> {n / r[2]:.1f} k tiles/s")
            del rgb, out
