#!/bin/bash
# k_hand_sweep with 256-thread (3 per CU) against 512-thread (2 per CU, 3712-point tile) work-groups, one box
cd $GRAFT_REPO_ROOT
for nt in 256 512 256 512; do
  AGH_EXTRA_FLAGS="-DAGH_SWEEP_NT=$nt" python -c "from agile_grasp_amd import build; build.build(force=True)" > /dev/null 2>&1
  if [ "$nt" = "512" ] && [ -z "${DONE_PARITY:-}" ]; then
    timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed"; DONE_PARITY=1
  fi
  for cfg in C2 C4; do
    python bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-extras --batch-clouds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('NT=$nt $cfg step %.4f ms  sweep %.2f us' % (d['ms_per_step'], d['kernel_ms_per_step']['hand_sweep']*1e3))"
  done
done
