#!/usr/bin/env python3
"""bench.py -- 1024x1024 H&E tiles/s normalised (Macenko) on N MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (ExtractiveStainNormalizer('macenko').transform) over one
batch of synthetic 1024x1024x3 uint8 tiles already resident in HBM (BASELINE.json configs[1]).
Tiles are independent, so ranks shard the batch with NO data-path collective ("weak" scaling:
every rank processes --tiles tiles per step); the only RCCL traffic is a QC all-gather of the
per-tile (M, maxC, status) after the timed region.

    python bench.py                                  # N=1, prints ONE JSON line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Extra objects in the JSON line: "roofline" (the OD+reconstruction kernel k_apply, timed with HIP
events on its own stream inside the timed region), "kernels" (per-kernel-class ms per step from
an untimed instrumented pass), "end_to_end" (compulsory / sweep-model HBM fractions),
"cpu_baseline" (the numpy oracle on the box's host cores, rank 0 at N=1 only), "parity".
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


# ----------------------------------------------------------------------------------------------
# CPU baseline (oracle = "port"), run BEFORE the GPU is touched so that fork() is safe
# ----------------------------------------------------------------------------------------------
def _cpu_worker(args):
    seed, n_tiles, size, Mt, mct = args
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:  # noqa: BLE001
        pass
    import numpy as np  # noqa: F401
    from oracle import stain_oracle as so
    nrm = so.ExtractiveStainNormalizer("macenko")
    nrm.stain_matrix_target, nrm.maxC_target = Mt, mct.reshape(1, 2)
    tiles = [so.synth_tile(size, size, seed * 100 + i) for i in range(n_tiles + 1)]
    nrm.transform(tiles[0])                      # warm-up
    t0 = time.perf_counter()
    for t in tiles[1:]:
        nrm.transform(t)
    return time.perf_counter() - t0


def cpu_baseline(size: int, tiles_per_core: int = 2):
    import multiprocessing as mp

    import numpy as np
    from oracle import stain_oracle as so
    tgt = so.synth_tile(size, size, 1, so.M_TRUE_TGT)
    n = so.ExtractiveStainNormalizer("macenko")
    n.fit(tgt)
    Mt, mct = n.stain_matrix_target, n.maxC_target.reshape(2)
    t_single = _cpu_worker((0, 3, size, Mt, mct)) / 3.0
    cores = max(1, min(os.cpu_count() or 1, 64))
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        el = pool.map(_cpu_worker, [(k + 1, tiles_per_core, size, Mt, mct) for k in range(cores)])
    value = cores * tiles_per_core / max(el)
    return {
        "value": round(value, 3), "unit": "tiles/s", "cores": cores, "kind": "port",
        "sample": f"{cores} processes x {tiles_per_core} tiles of {size}x{size} (numpy oracle, 1 thread each, "
                  f"transform only); single process: {1.0 / t_single:.3f} tiles/s; host has {os.cpu_count()} logical cores",
        "single_core_tiles_per_s": round(1.0 / t_single, 3),
        "_Mt": Mt, "_mct": mct,
    }, np


# ----------------------------------------------------------------------------------------------
# HIP events through the runtime torch already loaded (timing on the kernels' own stream)
# ----------------------------------------------------------------------------------------------
class HipEvents:
    def __init__(self, n):
        self.hip = C.CDLL("libamdhip64.so.7")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventDestroy.argtypes = [C.c_void_p]
        self.n = n
        self.ev = (C.c_void_p * n)()
        for i in range(n):
            e = C.c_void_p()
            assert self.hip.hipEventCreate(C.byref(e)) == 0
            self.ev[i] = e
        self.tags = (C.c_int32 * (n // 2))()
        self.tiles = (C.c_int32 * (n // 2))()

    def profile(self, mask):
        from stainlib_amd import _ffi
        p = _ffi.SlProfile()
        p.events = C.cast(self.ev, C.POINTER(C.c_void_p))
        p.tags = C.cast(self.tags, C.POINTER(C.c_int32))
        p.tiles = C.cast(self.tiles, C.POINTER(C.c_int32))
        p.capacity, p.used, p.mask = self.n, 0, mask
        return p

    def pairs(self, prof):
        out = []
        for i in range(prof.used // 2):
            ms = C.c_float()
            rc = self.hip.hipEventElapsedTime(C.byref(ms), self.ev[2 * i], self.ev[2 * i + 1])
            if rc == 0:
                out.append((int(self.tags[i]), int(self.tiles[i]), float(ms.value)))
        return out

    def close(self):
        for i in range(self.n):
            self.hip.hipEventDestroy(self.ev[i])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tiles", type=int, default=512, help="tiles per GPU per step (BASELINE configs[1]: 512)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        a.gpus = world

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu, _ = cpu_baseline(a.size)

    import numpy as np
    import torch
    import torch.distributed as dist

    from oracle import stain_oracle as so
    from stainlib_amd import _ffi, engine
    from tools.synth import synth_tiles

    backend = os.environ.get("SL_BENCH_BACKEND", "nccl")      # "gloo" only for single-GPU dry runs of the N>1 logic
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()

    h = w = a.size
    P = h * w
    B = a.tiles
    rgb = synth_tiles(B, h, w, seed=1000 * rank + 7, device=dev)
    out = torch.empty_like(rgb)
    # fit once, outside the timed region: target tile with the target stain matrix
    tgt = synth_tiles(1, h, w, seed=1, device=dev, M_true=so.M_TRUE_TGT.tolist())
    Mt, mct, st = engine.macenko_fit(tgt)
    assert int(st[0]) == 0
    Mt, mct = Mt[0].contiguous(), mct[0].contiguous()
    ws = engine.Workspace()
    n_groups_max = 4 * B + 64
    ev = HipEvents(2 * n_groups_max * max(a.steps, 1))
    params = _ffi.default_params()

    def step(p):
        return engine.macenko_transform(rgb, Mt, mct, params=p, out=out, ws=ws)

    for _ in range(a.warmup):
        step(params)
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps; the dominant kernel (the fused persistent transform, or k_apply in
    # the one-launch-per-phase schedule) is bracketed by HIP events on the stream it is launched on
    prof = ev.profile(_ffi.PROF_APPLY | _ffi.PROF_FUSED_TRANSFORM)
    params.profile = C.pointer(prof)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step(params)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    timed_pairs = ev.pairs(prof)
    status = res[3]
    n_bad = int((status != 0).sum())

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])
        # QC gather of per-tile stats (36 B/tile) -- the only collective, outside the timed region
        stats = torch.cat([res[1].reshape(B, 6).float(), res[2].float(), status.float().reshape(B, 1)], dim=1)
        gathered = [torch.empty_like(stats) for _ in range(world)]
        dist.all_gather(gathered, stats)
        n_bad = int(sum(int((g[:, 8] != 0).sum()) for g in gathered))

    line = None
    if rank == 0:
        # ---- untimed instrumented pass: every kernel class
        prof_all = ev.profile(255)
        params.profile = C.pointer(prof_all)
        step(params)
        torch.cuda.synchronize()
        per = {}
        for tag, tiles, ms in ev.pairs(prof_all):
            per[_ffi.PROF_NAMES[tag]] = per.get(_ffi.PROF_NAMES[tag], 0.0) + ms
        params.profile = None

        # dominant kernel: whole fused transform (15 B/px sweep model: 4 dependent read sweeps + 1 write,
        # SURVEY 8d) -- or, for the per-phase schedule, its k_apply launches (6 B/px)
        fused = [(t, ms) for tag, t, ms in timed_pairs if tag == _ffi.PROF_FUSED_TRANSFORM]
        if fused:
            dom_name, bpp = "k_macenko_fused<transform> (mask+moments, angle select, conc select, apply: 4 sweeps + 1 write)", 15.0
            dom = fused
        else:
            dom_name, bpp = "k_apply (OD + reconstruction pass)", 6.0
            dom = [(t, ms) for tag, t, ms in timed_pairs if tag == _ffi.PROF_APPLY]
        dom_ms = sum(ms for _, ms in dom) / max(len(dom), 1)
        dom_bytes = bpp * P * (sum(t for t, _ in dom) / max(len(dom), 1))
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0

        # the graded OD + reconstruction pass on its own (sl_normalize_apply over the same batch, same stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        engine.normalize_apply(rgb, res[1], res[2], Mt, mct, out=out)
        reps = 10
        e0.record()
        for _ in range(reps):
            engine.normalize_apply(rgb, res[1], res[2], Mt, mct, out=out)
        e1.record()
        torch.cuda.synchronize()
        ap_ms = e0.elapsed_time(e1) / reps
        ap_bytes = 6.0 * P * B
        ap_gbs = ap_bytes / (ap_ms * 1e-3) / 1e9
        tiles_per_s = world * B * a.steps / elapsed
        per_gpu = tiles_per_s / world

        # HBM traffic from PMC counters: collected in separate rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json
        # documents command, units and the gfx950 FETCH_SIZE correction), scaled to this launch's pixel count
        traffic_dom = traffic_ap = None
        try:
            pmc = json.load(open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")))["kernels"]
            if fused:
                traffic_dom = pmc["k_fused<macenko,transform>"]["bytes_per_pixel"] * P * (sum(t for t, _ in dom) / len(dom))
            traffic_ap = pmc["k_apply"]["bytes_per_pixel"] * P * B
            if not fused:
                traffic_dom = pmc["k_apply"]["bytes_per_pixel"] * P * (sum(t for t, _ in dom) / max(len(dom), 1))
        except Exception:  # noqa: BLE001
            pass

        parity = None
        if world == 1:
            I = rgb[0].cpu().numpy()
            nrm = so.ExtractiveStainNormalizer("macenko")
            nrm.stain_matrix_target, nrm.maxC_target = Mt.cpu().numpy(), mct.cpu().numpy().reshape(1, 2)
            want = nrm.transform(I)
            got = out[0].cpu().numpy()
            d = np.abs(got.astype(np.int16) - want.astype(np.int16))
            parity = {"tile": 0, "u8_mismatch_rate": float((d != 0).mean()), "u8_max_abs_diff": int(d.max()),
                      "M_src_max_abs_err": float(np.abs(res[1][0].cpu().numpy() - so.macenko_stain_matrix(I)).max())}

        line = {
            "metric": "1024x1024 H&E tiles/sec normalized (Macenko)",
            "value": round(tiles_per_s, 1),
            "unit": "tiles/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: batch of {B} tiles {h}x{w}x3 uint8 per GPU, Macenko transform "
                                   "(fit once outside the timed region), tiles resident in HBM",
                       "tiles_per_gpu": B, "tile": [h, w, 3], "sharding": f"independent tiles x{world}, no data-path collective",
                       "failed_tiles": n_bad},
            "roofline": {"kernel": dom_name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic_dom,
                         "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                         "bytes_per_launch": dom_bytes, "bytes_per_pixel_model": bpp,
                         "avg_launch_ms": round(dom_ms, 5), "launches_timed": len(dom),
                         "frac_compulsory_6Bpx": round(achieved * 6.0 / bpp / HBM_PEAK_GBS, 4)},
            "roofline_apply": {"kernel": "k_apply (OD + reconstruction pass, sl_normalize_apply)", "bound": "hbm",
                               "achieved": round(ap_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ap_gbs / HBM_PEAK_GBS, 4), "traffic": traffic_ap, "bytes_per_launch": ap_bytes,
                               "avg_launch_ms": round(ap_ms, 5), "launches_timed": reps},
            "end_to_end": {"per_gpu_tiles_per_s": round(per_gpu, 1),
                           "frac_hbm_compulsory_6Bpx": round(per_gpu * 6.0 * P / 1e9 / HBM_PEAK_GBS, 4),
                           "frac_hbm_sweep_model_15Bpx": round(per_gpu * 15.0 * P / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(per.items())},
            "parity": parity,
        }
        if cpu is not None:
            line["cpu_baseline"] = {k: v for k, v in cpu.items() if not k.startswith("_")}
            line["gpu_over_cpu"] = round(tiles_per_s / cpu["value"], 1)
    ev.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
