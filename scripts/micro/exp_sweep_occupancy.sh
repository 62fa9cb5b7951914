#!/bin/bash
# k_hand_sweep: work-groups per CU against the LDS tile (built on the box; the product build is restored at the end):
#   gpurun -- 'bash scripts/micro/exp_sweep_occupancy.sh'
set -u
cd ${GRAFT_REPO_ROOT:-.}
for cfg in ${CONFIGS:-"4 1408" "3 2176" "4 1408" "3 2176"}; do
  set -- $cfg
  AGH_EXTRA_FLAGS="-DAGH_SWEEP_WGS=$1 -DAGH_SWEEP_TILE=$2" python -c "from agile_grasp_amd import build; build.build(force=True)" > /dev/null 2>&1
  echo "=== WGS=$1 TILE=$2"
  timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_path" 2>&1 | grep -E "passed|failed"
  timeout 300 python scripts/micro/quick_bench.py C2 C4 2>&1 | grep "step"
done
python -c "from agile_grasp_amd import build; build.build(force=True)" > /dev/null 2>&1
