#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
AGH_DEBUG_BUILD=1 python -c "from agile_grasp_amd import build; build.build(force=True)" > /dev/null 2>&1
for k in 2 3 4 0; do
  AGH_DEBUG_STOP_MOMENTS=$k timeout 200 python bench.py --config C2 --steps 30 --warmup 5 --no-cpu-baseline --batch-clouds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('stop $k moments us', round(d['kernel_ms_per_step']['taubin_moments']*1000,1))"
done
