#!/bin/bash
# A/B of library builds on ONE box: every ab/lib*.so in turn replaces the in-tree library; bench.py (HIP events) for C2 (+ the
# batch of eight) and C4, two rounds; with TRACE=1 also one rocprofv3 kernel trace each.
#   gpurun -- 'bash scripts/ab_bench.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp agile_grasp_amd/lib/libagile_grasp_hip.so /tmp/lib_keep.so
show='
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(" ", d["config"]["workload"][:12], round(d["ms_per_step"],4), {k:round(v*1000,1) for k,v in d["kernel_ms_per_step"].items()})
b=d.get("batched")
if b: print("  batch", round(b["ms_per_cloud"],4), {k:round(v*1000) for k,v in b["kernel_ms_per_batch"].items()})'
for round in 1 2; do
for L in ab/lib*.so; do
  cp $L agile_grasp_amd/lib/libagile_grasp_hip.so
  echo "== $L round $round"
  timeout 300 python bench.py --config C2 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$show"
  timeout 300 python bench.py --config C4 --steps 20 --warmup 5 --no-cpu-baseline --batch-clouds 0 2>/dev/null | python -c "$show"
  if [ "${AB_RAND:-0}" = 1 ]; then
    timeout 300 python bench.py --config C2 --normals rand50 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$show"
    timeout 300 python bench.py --config C4 --normals rand50 --steps 20 --warmup 5 --no-cpu-baseline --batch-clouds 0 2>/dev/null | python -c "$show"
  fi
  if [ "${AB_C3:-0}" = 1 ]; then
    timeout 300 python bench.py --config C3 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$show"
  fi
  if [ "${TRACE:-0}" = 1 ] && [ $round = 1 ]; then
    bash scripts/quick_trace.sh C2 > /dev/null 2>&1
    python - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/qt_C2.csv')):
    if float(r['calls']) > 100: print('   ', r['kernel'][7:40].ljust(34), r['avg_us'])
PY
  fi
done
done
cp /tmp/lib_keep.so agile_grasp_amd/lib/libagile_grasp_hip.so
