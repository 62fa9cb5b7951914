"""Timeline of the last N kernel dispatches of a rocprofv3 rocpd database: start offset, duration and the idle gap
before each kernel (us).  Usage: python scripts/rocpd_timeline.py results.db [N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(cur.execute(f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id "
                        f"order by d.start"))
rows = rows[-n:]
t0 = rows[0][1]
prev_end = None
for name, st, en in rows:
    gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(st - t0) / 1e3:10.1f}  dur {(en - st) / 1e3:8.1f}  gap {gap:8.1f}  {name.split('(')[0][:70]}")
    prev_end = max(prev_end or en, en)
