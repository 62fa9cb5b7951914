"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE needs 3 of the 4 TCC slots).  Per-launch HBM-side bytes
per kernel = 2 x FETCH_SIZE KiB (gfx950 tallies the 128-B requests of wide 16 B/lane reads at 64 B -- the guide's
correction; our gathers are float4 loads) + WRITE_SIZE KiB (uncalibrated, taken as reported).
Usage: python scripts/pmc_traffic.py fetch.db write.db KEY [out.json]"""
import json
import os
import sqlite3
import sys


def per_kernel(db):
    c = sqlite3.connect(db)
    cur = c.cursor()
    t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [x for x in t if x.startswith("rocpd_pmc_event")][0]
    disp = [x for x in t if x.startswith("rocpd_kernel_dispatch")][0]
    sym = [x for x in t if x.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), sum(p.value) from {pmc} p join {disp} d on p.event_id = d.event_id "
         f"join {sym} s on d.kernel_id = s.id group by s.kernel_name")
    return {r[0]: (r[1], r[2]) for r in cur.execute(q)}


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
key = sys.argv[3]
out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
res = {}
if os.path.exists(out_path):
    res = json.load(open(out_path))
entry = {"kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if "agh" not in k:
        continue
    short = k.split("agh")[1].lstrip("0123456789").split("E")[0] if "_ZN3agh" in k else k
    short = k[k.index("agh") + 3:]
    short = short.lstrip("0123456789")
    name = short.split("ILi")[0].split("ILb")[0].split("EP")[0].split("ENS")[0]
    cap = ("<" + short.split("ILi")[1].split("E")[0] + ">") if "ILi" in short else ""
    if "ILb" in short:  # bool template argument (k_taubin_eigen<latency build?>)
        cap = "<latency>" if short.split("ILb")[1].startswith("1") else "<plain>"
    f = fetch.get(k, (1, 0.0))
    w = write.get(k, (1, 0.0))
    entry["kernels"][name + cap] = {"fetch_kib_per_launch": f[1] / f[0], "write_kib_per_launch": w[1] / w[0],
                                    "hbm_bytes_per_launch": (2.0 * f[1] / f[0] + w[1] / w[0]) * 1024.0}
hs = [v for k, v in entry["kernels"].items() if k.startswith("k_hand_sweep")]
entry["hand_sweep_bytes_per_launch"] = hs[0]["hbm_bytes_per_launch"] if hs else None
entry["note"] = "2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; separate --pmc passes; L2 / Infinity-Cache hits never reach these counters"
res[key] = entry
json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(entry, indent=1)[:1500])
