#!/bin/bash
# Instruction counts of k_hand_sweep up to each of its debug stops (SQ counters, one rocprofv3 --pmc pass per stop):
# where the instructions of the kernel are, not only where its time goes.  Needs the phase-timing build of the library:
#   AGH_DEBUG_BUILD=1 python -c "from agile_grasp_amd import build; build.build(force=True)"
#   gpurun -- 'bash scripts/moments_instr_diag.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for k in 1 2 3 4 0; do
  rm -rf /tmp/si_$k
  AGH_DEBUG_STOP_SWEEP=$k timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS -d /tmp/si_$k -o si -- python $R/bench.py --config C2 --steps 10 --warmup 2 --no-events \
    --no-cpu-baseline --no-extras --batch-clouds 0 --spin-seconds 0 > /tmp/si_$k.log 2>&1
  q=$(find /tmp/si_$k -name "*.db" | head -1)
  echo "== stop $k"; [ -n "$q" ] && python $R/scripts/pmc_table.py $q | grep -A9 "k_hand_sweep"
done
