"""Per-kernel summary (calls, total, average, min, max in microseconds) of a rocprofv3 rocpd SQLite database,
equivalent to the --stats kernel table.  Usage: python scripts/rocpd_summary.py results.db [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
q = (f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
     f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
rows = list(cur.execute(q))
total = sum(r[2] for r in rows) or 1
out = [("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
for name, n, tot, mn, mx in rows:
    out.append((name.split("(")[0][:90], n, round(tot / 1e3, 2), round(tot / n / 1e3, 2), round(mn / 1e3, 2),
                round(mx / 1e3, 2), round(100.0 * tot / total, 2)))
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerows(out)
if len(sys.argv) > 2:
    for r in out[:25]:
        print(*r, sep=" | ")
