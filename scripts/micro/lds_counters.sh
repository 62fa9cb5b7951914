#!/bin/bash
# Which unit does k_hand_sweep wait for?  LDS / VMEM / SALU issue and bank-conflict counters of one C2 bench (separate --pmc passes).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | grep -E "LDS|VMEM|SALU|SMEM|FLAT|INST_LEVEL|WAIT" | tr '\n' ' '; echo
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmc_x
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_x -o x -- python $R/bench.py --config C2 --steps 10 --warmup 2 --no-events \
    --no-cpu-baseline --no-extras --batch-clouds 0 --spin-seconds 0 > /tmp/pmc_x.log 2>&1
  q=$(find /tmp/pmc_x -name "*.db" | head -1)
  [ -n "$q" ] && python $R/scripts/pmc_table.py $q | grep -A12 "k_hand_sweep" | head -14 || tail -3 /tmp/pmc_x.log
done
