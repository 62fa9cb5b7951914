#!/bin/bash
# gpurun -- 'bash scripts/host_timeline.sh TAG'  ->  gpurun_out/TAG_host_timeline_{api,pipeline}.txt (+ the workloads' JSON lines)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r04}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in api pipeline; do
  rm -rf /tmp/ht_$mode
  timeout 600 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ht_$mode -o ht -- \
    python $R/scripts/host_timeline.py $mode 20 > /tmp/ht_$mode.log 2>&1
  grep '^{' /tmp/ht_$mode.log > $R/gpurun_out/${tag}_host_timeline_$mode.json
  if [ $mode = api ]; then
    python $R/scripts/trace_summary.py /tmp/ht_$mode 1 set_cloud+find_hands > $R/gpurun_out/${tag}_host_timeline_$mode.txt 2>&1
  else
    python $R/scripts/trace_summary.py /tmp/ht_$mode 4 preprocess,find_hands,classify,find_handles > $R/gpurun_out/${tag}_host_timeline_$mode.txt 2>&1
  fi
  tail -5 /tmp/ht_$mode.log
  ls /tmp/ht_$mode/* | head
done
