#!/bin/bash
# round 3, first GPU call: what the box's image holds (cv2 / skimage / spams / sklearn), the baseline, and the
# aliased-tiles experiment (512-item grid over 64 / 8 / 1 distinct input tiles).
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out; mkdir -p "$out"
{
  for m in cv2 skimage spams sklearn scipy numpy torch PIL; do
    python3 -c "import $m; print('$m', getattr($m,'__version__','?'))" 2>&1 | tail -1
  done
  for py in /opt/conda/bin/python3.9 /opt/conda/bin/python; do
    [ -x $py ] && for m in cv2 skimage spams; do $py -c "import $m; print('$py $m', $m.__version__)" 2>&1 | tail -1; done
  done
  ls /opt/conda/bin 2>&1 | head -5
  find / -name "cv2*" -not -path "/proc/*" 2>/dev/null | head
  find / -iname "*opencv*" -not -path "/proc/*" 2>/dev/null | head
} > "$out/r03_probe_imports.txt" 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > "$out/r03_first_gputests.txt"
L=$PWD/stainlib_amd/csrc
for rep in 1 2; do
for v in "" _same63 _same7 _same0; do
  STAINLIB_HIP_LIB=$L/libstainlib_hip$v.so python tools/time_kernels.py fused,apply 2>/dev/null | tail -1
done; done > "$out/r03_aliased_tiles.txt"
cat "$out/r03_probe_imports.txt" "$out/r03_first_gputests.txt" "$out/r03_aliased_tiles.txt"
