"""Interleaved A/B of library variants on one box: runs tools/cube_ab.py under each library twice, alternating (box-to-box and
minute-to-minute drift is +-3 %, so only same-call interleaved pairs mean anything).
    python tools/ab_libs.py base,<variant>,... [kinds...]      variant = V of `make -C stainlib_amd/csrc variant V=... VFLAGS=...`"""
import os
import subprocess
import sys

libs = sys.argv[1].split(",")
for rep in range(2):
    for v in libs:
        env = dict(os.environ)
        if v != "base":
            env["STAINLIB_HIP_LIB"] = os.path.abspath(f"stainlib_amd/csrc/libstainlib_hip_{v}.so")
        r = subprocess.run([sys.executable, "tools/cube_ab.py"] + sys.argv[2:], env=env, capture_output=True, text=True)
        for ln in r.stdout.splitlines():
            if "|" in ln:
                print(f"{v:8s}", ln[:11], " ".join(f"{seg.split()[0]}={seg.split()[1]}" for seg in ln.split("|")[1:6]), ln.split("|")[-1], flush=True)
