cd "$GRAFT_REPO_ROOT"; L=$PWD/stainlib_amd/csrc
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for rep in 1 2 3; do for v in _base ""; do STAINLIB_HIP_LIB=$L/libstainlib_hip$v.so timeout 120 python tools/time_kernels.py fused,apply,aug 2>/dev/null | tail -1; done; done
