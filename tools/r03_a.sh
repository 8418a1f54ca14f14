#!/bin/bash
# round 3: staged refine (finish 2) -- correctness on the Macenko / tissue GPU tests, then sub-step clocks
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp
L=$PWD/stainlib_amd/csrc
tag=${1:-a}
timeout 900 python -m pytest tests/test_gpu_macenko.py tests/test_gpu_tissue.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -15 > "$out/r03_${tag}_gputests.txt"
STAINLIB_HIP_LIB=$L/libstainlib_hip_sub.so timeout 300 python tools/merged_diag.py 512 1024 2>&1 | grep -v amdgpu > "$out/r03_${tag}_diag.txt"
for rep in 1 2 3; do
for v in "" _h0 _p1; do
  [ -f $L/libstainlib_hip$v.so ] && STAINLIB_HIP_LIB=$L/libstainlib_hip$v.so timeout 120 python tools/time_kernels.py fused 2>/dev/null | tail -1
done; done > "$out/r03_${tag}_times.txt"
cat "$out/r03_${tag}_gputests.txt" "$out/r03_${tag}_diag.txt" "$out/r03_${tag}_times.txt"
