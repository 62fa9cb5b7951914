#!/bin/bash
# k_hand_sweep's split threshold (AGH_SWEEP_FORK_THRESH: slab candidates above which a sample is searched by two work-groups;
# 0 = never) against the step and the kernel's own time, C2 and C4:  gpurun -- 'bash scripts/micro/exp_sweep_fork.sh'
cd ${GRAFT_REPO_ROOT:-.}
for cfg in C2 C4; do
  for t in ${THRESHOLDS:-0 3000 4000 5000 6000 0 4000}; do
    AGH_SWEEP_FORK_THRESH=$t timeout 300 python bench.py --config $cfg --steps 50 --warmup 5 --no-extras --batch-clouds 0 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg thresh $t: step %.4f ms (median %.4f)  sweep %.2f us  frac %.3f  hyp %d' % (d['ms_per_step'], d['ms_per_step_spread']['median_ms'], d['kernel_ms_per_step']['hand_sweep'] * 1e3, d['roofline']['frac'], d['config']['hypotheses']))"
  done
done
