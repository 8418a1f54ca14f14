"""Run the one-sweep pooled chain a few times (for rocprofv3 --kernel-trace --stats).  python tools/pool2_chain.py [n_tiles [reps [kind]]]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from stainlib_amd.distributed import PooledSlideStatistics  # noqa: E402
from tools.synth import synth_tiles  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rgb = synth_tiles(n, 1024, 1024, seed=9)
st = PooledSlideStatistics(group=False)
for _ in range(2):
    s = st.enqueue_merged(rgb, n_tiles_total=n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    s = st.enqueue_merged(rgb, n_tiles_total=n)
torch.cuda.synchronize()
print(f"one-sweep chain, {n} tiles: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call; miss {int(s[9])} why {int(s[33])}")
