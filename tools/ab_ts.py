"""Interleaved A/B of library variants on one box with tools/time_default.py (default parameters, fused schedule):
    python tools/ab_ts.py base,<variant>,... [kinds=iid] [n] [size] [two_sweep]      variant = V of `make -C stainlib_amd/csrc variant V=...`"""
import os
import subprocess
import sys

libs = sys.argv[1].split(",")
for rep in range(int(os.environ.get("AB_REPS", "3"))):
    for v in libs:
        env = dict(os.environ)
        if v != "base":
            env["STAINLIB_HIP_LIB"] = os.path.abspath(f"stainlib_amd/csrc/libstainlib_hip_{v}.so")
        r = subprocess.run([sys.executable, "tools/time_default.py"] + sys.argv[2:], env=env, capture_output=True, text=True)
        print(f"{v:8s}", " | ".join(ln for ln in r.stdout.splitlines() if " ms " in ln) or r.stderr[-300:], flush=True)
