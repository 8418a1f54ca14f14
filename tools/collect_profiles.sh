#!/bin/bash
# Collects every measurement the docs cite into gpurun_out/ (run on the GPU box from the repo root):
#   make -C tools && /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/collect_profiles.sh r04'
# then copy gpurun_out/r04_* to profiles/ (tracked); bench.py reads <tag>_pmc_traffic.json from there.
tag=${1:-r06}
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
dev=$PWD/stainlib_amd/csrc/libstainlib_hip_dev.so
python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > "$out/${tag}_bench_n1.json"
# the N > 1 path of bench.py on a real RCCL process group of one rank (what one GPU can verify of it)
SL_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --slide-pooled 2>/dev/null | grep '^{' | tail -1 > "$out/${tag}_bench_rccl_world1.json"
python tools/crossover.py 2>/dev/null | grep "size" > "$out/${tag}_crossover.txt"
python tools/crossover.py macenko 700,512,384 2>/dev/null | grep "size" > "$out/${tag}_crossover_small.txt"
python tools/phase_classes.py 1024 2>/dev/null | grep "^size" > "$out/${tag}_phase_classes.txt"
[ -f "$dev" ] && STAINLIB_HIP_LIB=$dev python tools/merged_diag.py 512 1024 2>/dev/null | grep -v amdgpu > "$out/${tag}_phase_times.txt"
python tools/bench_pipeline.py 2>/dev/null | tail -3 > "$out/${tag}_pipeline.txt"
# rocprofv3 kernel trace of the same bench command (no CPU baseline / secondary: the trace is about the headline kernels)
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/rocpd_stats.py "$(ls /tmp/kt/*/*.db /tmp/kt/*.db 2>/dev/null | head -1)" --last 20 "k_fused<0, true" > "$out/${tag}_kernel_stats.md" 2>&1
# Vahadane, 128 tiles (BASELINE configs[2]) and 512 tiles: per-kernel times
cat > /tmp/vah.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from stainlib_amd import engine
from tools.synth import synth_tiles
n = int(sys.argv[1])
rgb = synth_tiles(n, 1024, 1024, seed=5)
tgt = synth_tiles(1, 1024, 1024, seed=1001)
out = torch.empty_like(rgb)
p = engine.make_params(dl_tol=1e-6, dl_max_sweeps=100)
Mt, mct, _, _ = engine.vahadane_fit(tgt, params=p)
for _ in range(10):
    engine.vahadane_transform(rgb, Mt[0], mct[0], params=p, out=out)
torch.cuda.synchronize()
PY
for n in 128 512; do
  rm -rf /tmp/ktv; timeout 600 rocprofv3 --kernel-trace -d /tmp/ktv -o p -- python /tmp/vah.py $n > /dev/null 2>&1
  python tools/rocpd_stats.py "$(ls /tmp/ktv/*/*.db /tmp/ktv/*.db 2>/dev/null | head -1)" 2>&1 | grep -v "at::native\|rocclr\|Cijk" > "$out/${tag}_kernel_stats_vahadane$n.md"
done
# PMC passes (each in its own run, --pmc only): HBM traffic, SQ activity, instruction mix
for pass in "f:FETCH_SIZE" "w:WRITE_SIZE" "s1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "s2:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  t=${pass%%:*}; c=${pass#*:}
  rm -rf /tmp/pmc_$t; timeout 400 rocprofv3 --pmc $c -d /tmp/pmc_$t -o p -- python tools/run_fused_once.py 512 > /dev/null 2>&1
  python tools/pmc_summary.py "$(ls /tmp/pmc_$t/*/*.db /tmp/pmc_$t/*.db 2>/dev/null | head -1)" > "$out/${tag}_pmc_$t.txt" 2>&1
done
# round 5: the same two traffic counters with the two-sweep route off (three sweeps for every tile) and forced (every tile tries)
for ts in 1 2; do
  for pass in "f:FETCH_SIZE" "w:WRITE_SIZE"; do
    t=${pass%%:*}; c=${pass#*:}
    rm -rf /tmp/pmc_${t}_ts$ts; timeout 400 rocprofv3 --pmc $c -d /tmp/pmc_${t}_ts$ts -o p -- python tools/run_fused_once.py 512 0 $ts > /dev/null 2>&1
    python tools/pmc_summary.py "$(ls /tmp/pmc_${t}_ts$ts/*/*.db /tmp/pmc_${t}_ts$ts/*.db 2>/dev/null | head -1)" > "$out/${tag}_pmc_${t}_ts$ts.txt" 2>&1
  done
done
# the same two traffic counters on the one-launch-per-phase schedule at 64 tiles (192 MB: fits the Infinity Cache) and 512
for n in 64 512; do
  rm -rf /tmp/pm_$n; timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pm_$n -o p -- python tools/run_fused_once.py $n 1 > /dev/null 2>&1
  python tools/pmc_summary.py "$(ls /tmp/pm_$n/*/*.db /tmp/pm_$n/*.db 2>/dev/null | head -1)" > "$out/${tag}_pmc_phase_fetch_n$n.txt" 2>&1
done
# round 5: what the fused FIT kernel writes (no output: lists, sample and scratch) and reads beyond the tile
for pass in "f:FETCH_SIZE" "w:WRITE_SIZE"; do
  t=${pass%%:*}; c=${pass#*:}
  rm -rf /tmp/pmc_fit_$t; timeout 400 rocprofv3 --pmc $c -d /tmp/pmc_fit_$t -o p -- python tools/run_fused_once.py 512 2 0 fit > /dev/null 2>&1
  python tools/pmc_summary.py "$(ls /tmp/pmc_fit_$t/*/*.db /tmp/pmc_fit_$t/*.db 2>/dev/null | head -1)" 2>&1 | grep -A2 "k_fused" > "$out/${tag}_pmc_fit_$t.txt"
done
# ... and what bounds the Lab sweeps: instruction counts and LDS conflicts
for pass in "v:SQ_INSTS_VALU SQ_INSTS_LDS" "l:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "b:SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  t=${pass%%:*}; c=${pass#*:}
  rm -rf /tmp/pmc_lab_$t; timeout 400 rocprofv3 --pmc $c -d /tmp/pmc_lab_$t -o p -- python tools/run_lab.py > /dev/null 2>&1
  python tools/pmc_summary.py "$(ls /tmp/pmc_lab_$t/*/*.db /tmp/pmc_lab_$t/*.db 2>/dev/null | head -1)" 2>&1 | grep -A3 "k_lab\|k_byte" >> "$out/${tag}_pmc_lab.txt"
done
python tools/make_pmc_traffic.py "$tag" "$out" > "$out/${tag}_pmc_traffic.json" 2> /dev/null
python tools/time_batches.py 2>/dev/null | grep " ms" > "$out/${tag}_batches_mixed.txt"
# configs[4] at one GPU's shard size; round 6: the kernels of the one-sweep pooled chain (512 tiles and one rank's 12 500), its steps by
# content class, what binary64 apply arithmetic would cost
python tools/slide_scale.py 512,2048,12500 2>/dev/null | grep -v amdgpu > "$out/${tag}_slide_scale.txt"
for n in 512 12500; do
  rm -rf /tmp/kp; timeout 600 rocprofv3 --kernel-trace -d /tmp/kp -o p -- python tools/pool2_chain.py $n 5 > /dev/null 2>&1
  python tools/rocpd_stats.py "$(ls /tmp/kp/*/*.db /tmp/kp/*.db 2>/dev/null | head -1)" 2>&1 | grep -v "at::native\|rocclr\|Cijk" > "$out/${tag}_kernel_stats_pooled$n.md"
done
(python tools/pool2_check.py 512 1024; python tools/pool2_check.py 12500 1024) 2>/dev/null | grep -v amdgpu > "$out/${tag}_pooled_classes.txt"
[ -x tools/bin/kbench_apply_f64 ] && timeout 120 tools/bin/kbench_apply_f64 > "$out/${tag}_apply_f64.txt" 2>&1
timeout 200 python tools/power_classes.py 2>/dev/null | grep -v amdgpu > "$out/${tag}_power_classes.txt"
python tools/structured_rate.py 2>/dev/null | grep -v amdgpu > "$out/${tag}_structured_rate.txt"
python tools/cube_ab.py iid white_bg quantized ihc grey_bg blobs 2>/dev/null | grep -v amdgpu > "$out/${tag}_cube_prefilter_ab.txt"
python tools/ts_check.py 512 1024 iid,white_bg,quantized,blobs,ihc,palette12 2>/dev/null | grep -v amdgpu > "$out/${tag}_two_sweep_ab.txt"
(python tools/wide_two_sweep.py 160,192,224,256; python tools/wide_two_sweep.py 192,256 1024 ihc) 2>/dev/null | grep " n " > "$out/${tag}_wide_two_sweep.txt"
[ -f "$dev" ] && STAINLIB_HIP_LIB=$dev python tools/ts_phases.py 512 1024 iid 2>/dev/null | grep -v amdgpu > "$out/${tag}_two_sweep_phases.txt"
rm -rf /tmp/kl; timeout 300 rocprofv3 --kernel-trace -d /tmp/kl -o p -- python tools/run_lab.py > /dev/null 2>&1
python tools/rocpd_stats.py "$(ls /tmp/kl/*/*.db /tmp/kl/*.db 2>/dev/null | head -1)" 2>&1 | grep -v "at::native\|rocclr\|Cijk" > "$out/${tag}_kernel_stats_lab.md"
[ -x tools/bin/ubench_ops ] && timeout 120 tools/bin/ubench_ops > "$out/${tag}_ubench_ops.txt" 2>&1
[ -x tools/bin/ubench_issue ] && timeout 120 tools/bin/ubench_issue > "$out/${tag}_ubench_issue.txt" 2>&1
[ -x tools/bin/kbench_stream ] && timeout 120 tools/bin/kbench_stream > "$out/${tag}_kbench_stream.txt" 2>&1
ls -la "$out" | tail -30
