"""Per-kernel averages of every PMC counter in a rocprofv3 rocpd database.  Usage: pmc_table.py results.db [...]"""
import sqlite3, sys, collections
tab = collections.defaultdict(dict)
for db in sys.argv[1:]:
    c = sqlite3.connect(db); cur = c.cursor()
    t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [x for x in t if x.startswith("rocpd_pmc_event")][0]; info = [x for x in t if x.startswith("rocpd_info_pmc")][0]
    disp = [x for x in t if x.startswith("rocpd_kernel_dispatch")][0]; sym = [x for x in t if x.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, i.name, count(*), sum(p.value) from {pmc} p join {info} i on p.pmc_id = i.id "
         f"join {disp} d on p.event_id = d.event_id join {sym} s on d.kernel_id = s.id group by s.kernel_name, i.name")
    for k, n, cnt, tot in cur.execute(q):
        if "agh" in k:
            short = k[k.index("agh") + 3:].lstrip("0123456789")
            name = short.split("EP")[0].split("ENS")[0].split("EEv")[0]
            tab[name][n] = tot / cnt
for k in sorted(tab, key=lambda k: -tab[k].get("SQ_WAVE_CYCLES", tab[k].get("SQ_BUSY_CYCLES", 0))):
    print(k)
    for n, v in sorted(tab[k].items()):
        print("    %-28s %14.0f" % (n, v))
