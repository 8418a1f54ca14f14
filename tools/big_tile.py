import sys, time, torch
sys.path.insert(0, ".")
from stainlib_amd import engine
from tools.synth import synth_tiles
big = synth_tiles(128, 2048, 2048, seed=4)
for _ in range(2):
    M2, mc2, st2 = engine.macenko_fit(big)
torch.cuda.synchronize(); t0 = time.time()
M2, mc2, st2 = engine.macenko_fit(big); torch.cuda.synchronize(); dt = time.time() - t0
print("fused fit 128 tiles 2048^2: %.2f ms -> %.0f Mpx/s" % (dt * 1e3, 128 * 4.19 / dt), int(st2.sum()))
