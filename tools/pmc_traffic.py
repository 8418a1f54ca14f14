import sqlite3, sys
for tag in ("f", "w"):
    db = sqlite3.connect(f"gpurun_out/pmc_{tag}/p_results.db"); cur = db.cursor()
    for k, c, v, n in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        if "sl::" in k or "elementwise" in k.lower() or "copy" in k.lower():
            print(tag, k[:80], c, v, n)
