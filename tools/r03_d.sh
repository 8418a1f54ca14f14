#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp
L=$PWD/stainlib_amd/csrc
tag=${1:-d}; shift
timeout 600 python -m pytest tests/test_gpu_macenko.py tests/test_gpu_tissue.py -m gpu -x -q 2>&1 | tail -3 > "$out/r03_${tag}_gputests.txt"
for rep in 1 2 3; do
for v in "$@"; do
  f=$L/libstainlib_hip$v.so
  [ -f $f ] && STAINLIB_HIP_LIB=$f timeout 120 python tools/time_kernels.py fused 2>/dev/null | tail -1
done; done > "$out/r03_${tag}_times.txt"
cat "$out/r03_${tag}_gputests.txt" "$out/r03_${tag}_times.txt"
