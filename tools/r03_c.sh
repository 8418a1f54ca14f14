#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp
L=$PWD/stainlib_amd/csrc
for rep in 1 2; do
for v in dev0 dev1; do
  echo "== $v"; STAINLIB_HIP_LIB=$L/libstainlib_hip_$v.so timeout 300 python tools/merged_diag.py 512 1024 2>&1 | grep -E "fused|phases|sweep|finish|total|first half"
done; done > "$out/r03_c_diag.txt"
for rep in 1 2 3; do
for v in "" _h0; do
  STAINLIB_HIP_LIB=$L/libstainlib_hip$v.so timeout 120 python tools/time_kernels.py fused 2>/dev/null | tail -1
done; done > "$out/r03_c_times.txt"
cat "$out/r03_c_diag.txt" "$out/r03_c_times.txt"
