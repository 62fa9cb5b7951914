#!/bin/bash
# One rocprofv3 kernel-trace of bench.py for a config; prints the per-kernel summary.
#   gpurun -- 'bash scripts/quick_trace.sh C2'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cfg=${1:-C2}; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qt_$cfg
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/qt_$cfg -o kt -- python $R/bench.py --config $cfg --steps 50 --warmup 5 \
  --no-events --no-cpu-baseline --batch-clouds 0 "$@" > /tmp/qt_$cfg.log 2>&1
db=$(find /tmp/qt_$cfg -name "*.db" | head -1)
[ -n "$db" ] && python $R/scripts/rocpd_summary.py $db $R/gpurun_out/qt_${cfg}.csv > /dev/null
cut -c1-60,100- $R/gpurun_out/qt_${cfg}.csv | head -16
