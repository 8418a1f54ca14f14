"""-m gpu: what ONE GPU can verify of the N > 1 path -- bench.py under the driver's launcher with a real RCCL process group of
one rank (SL_BENCH_FORCE_DIST=1): the process group bound to the device, device barriers, the max / gather collectives of the
bench line, the per-rank core slice, and the pooled slide statistics whose every stage all-reduces device tensors through RCCL.
More than one rank stays with the gloo tests (tests/test_distributed_gloo.py) and the driver's 8-GPU run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_the_launcher_with_a_one_rank_rccl_group():
    env = dict(os.environ, SL_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29611", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--tiles", "48", "--size", "256",
           "--no-cpu-baseline", "--no-secondary", "--slide-pooled"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    d = line["distributed"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and len(d["per_rank_tiles_per_s"]) == 1
    assert d["cpu_affinity_of_rank0"] is not None
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["failed_tiles"] == 0
    assert line["parity"]["all_tile_status_ok"] and line["parity"]["fused_equals_per_phase_schedule_on_the_whole_batch"]
    sp = d["slide_pooled"]
    assert sp["tiles"] == 48 and sp["ranks_agree_bitwise"] and len(sp["selection_paths"]) == 2
    # the same slide without any process group: the collectives of a one-rank group change nothing
    import torch
    from stainlib_amd.distributed import PooledSlideStatistics
    from tools.synth import synth_tiles
    rgb = synth_tiles(48, 256, 256, seed=7, device=torch.device("cuda", 0))      # bench.py: seed = 1000 * rank + 7
    M, maxC = PooledSlideStatistics()(rgb)
    np.testing.assert_allclose(np.asarray(sp["M_slide"]), M.reshape(-1), rtol=0, atol=2e-6)
