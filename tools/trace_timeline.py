"""Start / end of the last launches in a rocprofv3 kernel trace (rocpd SQLite), relative to the first of them: who ran beside whom.
    python tools/trace_timeline.py trace.db [last=16]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute(f"select name, start, end, {q} from kernels order by start").fetchall()[-n:]
t0 = rows[0][1]
for name, s, e, qid in rows:
    short = name.split("(")[0].replace("void sl::", "").replace("sl::", "")[:34]
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  q{qid}  {short}")
