#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace into the classic --stats table.

    python tools/rocpd_stats.py gpurun_out/prof/xyz_results.db > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
    "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
    "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch B | grid_x | wg_x |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    name = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
    print(f"| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | {100 * r[2] / tot:.1f} | "
          f"{r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} | {r[12]} |")

# optional: --last N substring  -> the mean over the last N launches of the kernels whose name contains `substring`
# (bench.py's timed region is its last K launches: the launches before them bring the clocks up and warm up)
if "--last" in sys.argv:
    i = sys.argv.index("--last")
    n_last, sub = int(sys.argv[i + 1]), sys.argv[i + 2]
    d = [r[0] for r in cur.execute("select duration from kernels where name like ? order by start", (f"%{sub}%",)).fetchall()]
    if d:
        tail = d[-n_last:]
        print(f"\nlast {len(tail)} of {len(d)} launches of `*{sub}*`: avg {sum(tail) / len(tail) / 1e3:.2f} us, min {min(tail) / 1e3:.2f}, max {max(tail) / 1e3:.2f} "
              f"(the {len(d) - len(tail)} before them: avg {sum(d[:-n_last]) / max(1, len(d) - len(tail)) / 1e3:.2f} us -- clock spin-up and warm-up launches)")
