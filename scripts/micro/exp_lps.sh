#!/bin/bash
# k_taubin_eigen: eight lanes per sample up to how many samples?  (C4: 8000, the batch of eight: 16 000)
cd $GRAFT_REPO_ROOT
for mx in 4096 8192 16384; do
  AGH_EXTRA_FLAGS="-DAGH_LPS8_MAX=$mx" python -c "from agile_grasp_amd import build; build.build(force=True)" > /dev/null 2>&1
  python bench.py --config C4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --batch-clouds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('LPS8_MAX=$mx C4 step %.4f ms eigen %.1f us' % (d['ms_per_step'], d['kernel_ms_per_step']['taubin_eigen']*1e3))"
  python scripts/batch_bench.py --clouds 8 --steps 20 2>/dev/null | tail -1 | cut -c1-300
done
