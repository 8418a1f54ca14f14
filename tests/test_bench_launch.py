"""bench.py --gpus N must produce an N-rank run by itself or fail (round-3 review, item 2): without a launcher in the environment it
re-executes under torch.distributed.run; with fewer than N GPUs on the box it exits with status 2.  The CPU tests drive the launch,
rendezvous, placement and collective logic with SL_BENCH_DRY=1 (a sleeping stand-in for the step: no kernel, no product import);
the GPU tests run the real step on two gloo ranks sharing the box's one GPU, and the refusal on a 1-GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    env.update(extra)
    return env


def _line(stdout):
    return json.loads([ln for ln in stdout.strip().splitlines() if ln.startswith("{")][-1])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=ROOT,
                       env=_clean_env(SL_BENCH_DRY="1", SL_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = _line(r.stdout)
    assert line["dry_run"] is True and line["n_gpus"] == 2
    d = line["distributed"]
    assert d["backend"] == "gloo" and d["world_size"] == 2
    assert len(d["per_rank_tiles_per_s"]) == 2 and all(x > 0 for x in d["per_rank_tiles_per_s"])
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and [x["device"] for x in d["ranks"]] == [0, 1]
    c0, c1 = d["ranks"][0]["cores"], d["ranks"][1]["cores"]
    if c0 is not None and c1 is not None and len(os.sched_getaffinity(0)) >= 2:
        assert c0[1] < c1[0] or c1[1] < c0[0], (c0, c1)            # disjoint core slices


def test_world_8_dry_run_of_the_pooled_slide_mode():
    """Round-5 review, item 7: the first real 8-GPU run should not be the first time eight ranks meet.  `bench.py --gpus 8
    --slide-pooled` on the CPU (SL_BENCH_DRY=1, gloo): launcher, rendezvous of eight ranks, eight disjoint core slices, the shard
    arithmetic of the 100 000-tile slide, and the product's one-sweep pooled chain with its collectives on eight ranks (device steps:
    the numpy stand-ins of tests/pool2_standins.py) -- the ranks agree to the bit and match the reference on the concatenated slide."""
    import numpy as np
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--slide-pooled"], cwd=ROOT,
                       env=_clean_env(SL_BENCH_DRY="1", SL_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = _line(r.stdout)
    d = line["distributed"]
    assert line["dry_run"] is True and line["n_gpus"] == 8 and d["backend"] == "gloo" and d["world_size"] == 8
    assert [x["rank"] for x in d["ranks"]] == list(range(8)) and [x["device"] for x in d["ranks"]] == list(range(8))
    cores = [x["cores"] for x in d["ranks"]]
    if all(c is not None for c in cores) and len(os.sched_getaffinity(0)) >= 8:
        spans = sorted(cores)
        assert all(spans[i][1] < spans[i + 1][0] for i in range(7)), spans              # eight disjoint core slices
    sp = d["slide_pooled"]
    assert sp["shards_contiguous"] and sum(sp["shard_sizes_of_100000_tiles"]) == 100000 and set(sp["shard_sizes_of_100000_tiles"]) == {12500}
    assert sp["settled"] and sp["ranks_agree_bitwise"] and sp["selection_paths"] == ["merged", "merged"] and sp["tiles"] == 48
    sys.path.insert(0, ROOT)
    from oracle import stain_oracle as so
    tiles = [so.synth_tile(96, 128, 700 + s) for s in range(46)] + [np.full((96, 128, 3), 255, np.uint8)] * 2
    tall = np.concatenate(tiles, axis=0)
    M_ref = so.macenko_stain_matrix(tall)
    np.testing.assert_allclose(np.asarray(sp["M_slide"]).reshape(2, 3), M_ref, rtol=0, atol=2e-6)
    np.testing.assert_allclose(sp["maxC_slide"], np.percentile(so.get_concentrations(tall, M_ref), 99, axis=0), rtol=2e-6)


def test_more_gpus_than_the_box_has_is_refused():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "64", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=_clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert "refusing to measure fewer GPUs" in r.stderr and not r.stdout.strip()


def test_a_launcher_with_another_world_size_is_refused():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29641", "bench.py", "--gpus", "4", "--steps", "1", "--warmup", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(SL_BENCH_DRY="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "launcher started 2 rank(s)" in r.stderr, (r.returncode, r.stderr[-1500:])


def test_core_slices_follow_the_gpus_numa_nodes(monkeypatch, tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    allowed = list(range(16))
    # no NUMA information: an even split
    s = bench.core_slices(allowed, 4, [None] * 4)
    assert s == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    # two nodes of 8 cores, GPUs 0-1 on node 1, GPUs 2-3 on node 0
    real_open = open

    def fake_open(path, *a, **k):
        if str(path).endswith("node0/cpulist"):
            return real_open(tmp_path / "n0", *a, **k)
        if str(path).endswith("node1/cpulist"):
            return real_open(tmp_path / "n1", *a, **k)
        return real_open(path, *a, **k)
    (tmp_path / "n0").write_text("0-7\n")
    (tmp_path / "n1").write_text("8-15\n")
    monkeypatch.setattr("builtins.open", fake_open)
    s = bench.core_slices(allowed, 4, [1, 1, 0, 0])
    assert s == [[8, 9, 10, 11], [12, 13, 14, 15], [0, 1, 2, 3], [4, 5, 6, 7]]
    # a GPU without a node shares what the others left
    s = bench.core_slices(allowed, 3, [1, None, 1])
    assert s[0] == [8, 9, 10, 11] and s[2] == [12, 13, 14, 15] and s[1] == list(range(8))


@pytest.mark.gpu
def test_gpus_2_on_one_gpu_with_gloo_runs_the_real_step_on_two_ranks():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--tiles", "48", "--size", "256", "--sustain-s", "0.2",
                        "--no-cpu-baseline", "--no-secondary"], cwd=ROOT,
                       env=_clean_env(SL_BENCH_BACKEND="gloo", SL_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["failed_tiles"] == 0 and "dry_run" not in line
    d = line["distributed"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and len(d["per_rank_tiles_per_s"]) == 2
    assert line["sustained"]["steps"] >= 16 and line["roofline"]["bytes_per_pixel"] == 6.0


@pytest.mark.gpu
def test_gpus_8_on_a_smaller_box_is_refused():
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box has 8 GPUs")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8"], cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "refusing to measure fewer GPUs" in r.stderr, (r.returncode, r.stderr[-1500:])
