#!/bin/bash
# rocprofv3 kernel trace of agh_localize in a loop (scripts/localize_loop.py); per-call kernel budget.
#   gpurun -- 'bash scripts/localize_trace.sh [tag]'
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-localize}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/lt -o kt -- python $R/scripts/localize_loop.py 50 > /tmp/lt.log 2>&1
grep "ms per call" /tmp/lt.log
db=$(find /tmp/lt -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $db $R/gpurun_out/${TAG}_kernel_trace_stats.csv > /dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/${TAG}_kernel_trace_stats.csv")))
calls = max(int(r["calls"]) for r in rows if "hand_sweep" in r["kernel"])
tot = 0.0
for r in rows:
    per = float(r["total_us"]) / calls
    tot += per
    print(f"{r['kernel'].replace('_ZN3agh', '')[:44]:46s} launches {int(r['calls']):5d} avg {float(r['avg_us']):8.2f} us  per call {per:8.2f}")
print("kernels per call, us:", round(tot, 1))
PY
