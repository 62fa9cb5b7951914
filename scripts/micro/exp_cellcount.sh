#!/bin/bash
# k_cell_count in a batch of clouds: total work-groups (AGH_CELLCOUNT_WGS, experiment hook) against the batch's grid build
cd ${GRAFT_REPO_ROOT:-.}
for w in 0 2048 4096 8192 0 4096; do
  AGH_CELLCOUNT_WGS=$w python scripts/batch_bench.py --clouds 8 --steps 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cellcount wgs $w: batch %.4f ms  grid_build %.1f us' % (d['ms_per_batch'], d['kernel_ms_per_batch']['grid_build'] * 1e3))"
done
