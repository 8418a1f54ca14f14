#!/bin/bash
# round 3: baseline of the three-sweep fused kernel (sub-step clocks, instruction mix)
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp
L=$PWD/stainlib_amd/csrc
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > "$out/r03_base_gputests.txt"
STAINLIB_HIP_LIB=$L/libstainlib_hip_sub.so python tools/merged_diag.py 512 1024 2>&1 | grep -v amdgpu > "$out/r03_base_diag.txt"
for pass in "s2:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "s1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  t=${pass%%:*}; c=${pass#*:}
  rm -rf /tmp/pmc_$t; timeout 400 rocprofv3 --pmc $c -d /tmp/pmc_$t -o p -- python tools/run_fused_once.py 512 > /dev/null 2>&1
  python tools/pmc_summary.py "$(ls /tmp/pmc_$t/*/*.db /tmp/pmc_$t/*.db 2>/dev/null | head -1)" > "$out/r03_base_pmc_$t.txt" 2>&1
done
cat "$out/r03_base_gputests.txt" "$out/r03_base_diag.txt" "$out/r03_base_pmc_s2.txt" "$out/r03_base_pmc_s1.txt"
