"""Summarise a rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv run of scripts/host_timeline.py:
the trace is cut at the marker calls (hipDriverGetVersion) into call windows; per window kind (position modulo `phases`) it reports the
wall time of the window, the time inside HIP API calls by function, the device's busy time (kernels, copies) and what is left
(host-side work outside HIP + idle waits), then prints the event list of the LAST window of each kind.
    python scripts/trace_summary.py DIR phases name1,name2,... > summary.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

d, phases = sys.argv[1], int(sys.argv[2])
names = sys.argv[3].split(",")


def load(pattern):
    f = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    if not f:
        return []
    return list(csv.DictReader(open(f[0])))


api = load("*hip_api_trace.csv")
ker = load("*kernel_trace.csv")
cpy = load("*memory_copy_trace.csv")
ev = []
for r in api:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api", r["Function"], r.get("Correlation_Id", "")))
for r in ker:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", r["Kernel_Name"].split("(")[0][:60], r.get("Correlation_Id", "")))
copy_corr = {}
for r in cpy:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "copy"), r.get("Correlation_Id", "")))
    copy_corr[r.get("Correlation_Id", "")] = r
ev.sort()
mk = [e for e in ev if e[2] == "api" and e[3] == "hipDriverGetVersion"]
if len(mk) < 2:
    print("no marker windows found; api functions seen:", sorted({e[3] for e in ev if e[2] == 'api'}))
    sys.exit(0)
# the timed windows are the last ones: (count - 1) windows, phases kinds
wins = [(mk[i][1], mk[i + 1][0]) for i in range(len(mk) - 1)]
n_use = (len(wins) // phases) * phases
wins = wins[len(wins) - n_use:]
agg = [defaultdict(float) for _ in range(phases)]
cnt = [0] * phases
for wi, (a, b) in enumerate(wins):
    k = wi % phases
    cnt[k] += 1
    g = agg[k]
    g["_wall"] += b - a
    busy = []
    for s, e, kind, name, _ in ev:
        if e <= a or s >= b:
            continue
        s2, e2 = max(s, a), min(e, b)
        if kind == "api":
            g["api:" + name] += e2 - s2
            g["_api_calls:" + name] += 1
            g["_api"] += e2 - s2
        else:
            g[kind + ":" + name] += e2 - s2
            busy.append((s2, e2))
    busy.sort()
    tot, cur_s, cur_e = 0, None, None
    for s, e in busy:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    g["_device_busy"] += tot
for k in range(phases):
    g, c = agg[k], max(cnt[k], 1)
    print(f"== {names[k] if k < len(names) else k}: {c} windows, wall {g['_wall'] / c / 1e3:.1f} us per call; inside HIP API calls "
          f"{g['_api'] / c / 1e3:.1f} us; device busy (kernels + copies, union) {g['_device_busy'] / c / 1e3:.1f} us")
    for key in sorted((x for x in g if x.startswith("api:")), key=lambda x: -g[x]):
        print(f"   {key:42s} {g[key] / c / 1e3:9.1f} us  x{g['_api_calls:' + key[4:]] / c:.1f}")
    for key in sorted((x for x in g if x.startswith("kernel:") or x.startswith("copy:")), key=lambda x: -g[x]):
        print(f"   {key:70s} {g[key] / c / 1e3:9.1f} us")
for k in range(phases):
    idx = [i for i in range(len(wins)) if i % phases == k]
    a, b = wins[idx[-1]]
    print(f"\n-- last window of {names[k] if k < len(names) else k}: events (start offset us, duration us)")
    for s, e, kind, name, _ in ev:
        if e <= a or s >= b:
            continue
        print(f"   {(s - a) / 1e3:9.1f} {(e - s) / 1e3:9.1f}  {kind:6s} {name}")
