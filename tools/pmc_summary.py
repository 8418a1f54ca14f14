"""Summarise rocprofv3 --pmc output (rocpd sqlite) per kernel: mean counter values per dispatch."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if 'counters_collection' in tabs:
    cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    out = {}
    for k, c, v, n in rows:
        out.setdefault(k, {})[c] = (v, n)
    for k, d in out.items():
        if 'sl::' not in k:
            continue
        print(k[:90])
        for c, (v, n) in sorted(d.items()):
            print(f"    {c:28s} {v:16.1f}  (n={n})")
else:
    print(tabs)
