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


def _two_rank_worker(rank, world, port, n_tiles, q):
    """one of two processes that SHARE the GPU: its contiguous shard of the slide through the product's one-sweep chain, with gloo
    carrying the all-reduces of device tensors between the steps"""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stainlib_amd.distributed import PooledSlideStatistics, SlideNormalizer, shard_range
        from tools.synth import synth_tiles
        import stainlib_amd as sl
        dev = torch.device("cuda", 0)
        slide = synth_tiles(n_tiles, 256, 320, seed=21, device=dev)
        lo, hi = shard_range(n_tiles, rank, world)
        st = PooledSlideStatistics()
        got = st.finish(st.enqueue_merged(slide[lo:hi], n_tiles_total=n_tiles))
        miss = st.last_miss
        nrm = sl.MacenkoNormalizer()
        nrm.stain_matrix_target = np.array([[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]]) / np.linalg.norm([[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]], axis=1, keepdims=True)
        nrm.maxC_target = np.array([[1.9, 1.1]])
        sn = SlideNormalizer(nrm, mode="pooled")
        out, M_s, mc_s, _ = sn.transform_shard(slide[lo:hi], n_tiles_total=n_tiles)
        q.put((rank, None if got is None else (got[0], got[1]), miss, list(sn.last_path), M_s.cpu().numpy(), out.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_sharing_the_gpu_run_the_one_sweep_chain_over_gloo():
    """What one GPU can verify of the pooled mode on MORE than one rank with the REAL kernels: two processes on cuda:0, each with
    a contiguous shard of a 24-tile slide (12 + 12) and of a 23-tile one (12 + 11), every all-reduce of the chain carried by gloo.  Both ranks must reach the single-process statistics of the
    whole slide bit for bit (sums of integers and of binary64 partials in a fixed order per rank; the all-reduce adds two numbers),
    and the bytes of their shards must be those of the single-process transform."""
    import torch
    import torch.multiprocessing as mp
    import stainlib_amd as sl
    from stainlib_amd.distributed import PooledSlideStatistics, SlideNormalizer
    from tools.synth import synth_tiles
    ctx = mp.get_context("spawn")
    for n_tiles, port in ((24, 29631), (23, 29632)):
        q = ctx.Queue()
        procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, n_tiles, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        slide = synth_tiles(n_tiles, 256, 320, seed=21, device=torch.device("cuda", 0))
        st = PooledSlideStatistics(group=False)
        want = st.finish(st.enqueue_merged(slide))
        assert want is not None
        nrm = sl.MacenkoNormalizer()
        nrm.stain_matrix_target = np.array([[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]]) / np.linalg.norm([[0.55, 0.75, 0.35], [0.10, 0.95, 0.20]], axis=1, keepdims=True)
        nrm.maxC_target = np.array([[1.9, 1.1]])
        out_all = SlideNormalizer(nrm, group=False, mode="pooled").transform_shard(slide)[0].cpu().numpy()
        lo = 0
        for rank, got, miss, path, M_s, out in res:
            assert got is not None and miss == 0 and path == ["merged", "merged"], (rank, miss, path)
            # the moment sums of two shards are added in another order than one process adds its workgroups' partials: last bits only
            np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-12)
            np.testing.assert_allclose(got[1], want[1], rtol=1e-12)
            assert np.array_equal(M_s, res[0][4])                      # the ranks agree to the bit
            hi = lo + out.shape[0]
            assert np.mean(out != out_all[lo:hi]) < 1e-6                # (a last-bit difference of M may flip a byte; none seen)
            lo = hi
        assert lo == n_tiles
