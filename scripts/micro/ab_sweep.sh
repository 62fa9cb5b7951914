#!/bin/bash
# A/B of two versions of hand_sweep.hip on ONE box: scripts/micro/hand_sweep_{A,B}.hip.txt (not tracked); alternates builds.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in A B; do
  cp scripts/micro/hand_sweep_$v.hip.txt agile_grasp_amd/csrc/hand_sweep.hip
  python -c "from agile_grasp_amd import build; build.build(force=True)" > /dev/null 2>&1
  for cfg in C2 C4; do
    python bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-extras --batch-clouds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v $cfg step %.4f ms  sweep %.2f us' % (d['ms_per_step'], d['kernel_ms_per_step']['hand_sweep']*1e3))"
  done
done
done
