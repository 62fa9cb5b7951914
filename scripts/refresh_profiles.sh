#!/bin/bash
# Regenerates everything under profiles/ on the GPU box (one gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/refresh_profiles.sh'
# then copy gpurun_out/profiles/* into profiles/.  Every step runs under `timeout`; rocprofv3 runs use --no-events
# (HIP events inside a profiled process have hung before) and PMC counters are collected in their own passes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TAG=${TAG:-r06}
for cfg in C2 C3 C4; do
  rm -rf /tmp/kt_$cfg
  # (--batch-clouds 0: every k_hand_sweep launch of the trace is the single-cloud launch the bench line's roofline describes)
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$cfg -o kt -- python $R/bench.py --config $cfg --steps 50 --warmup 5 \
    --no-events --no-cpu-baseline --no-extras --batch-clouds 0 > /tmp/kt_$cfg.log 2>&1
  db=$(find /tmp/kt_$cfg -name "*.db" | head -1)
  [ -n "$db" ] && python $R/scripts/rocpd_summary.py $db $OUT/${TAG}_$(echo $cfg | tr A-Z a-z)_kernel_trace_stats.csv > /dev/null
done
# the batch of 8 clouds in one context (what bench.py reports under "batched")
rm -rf /tmp/kt_batch
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_batch -o kt -- python $R/scripts/batch_bench.py --clouds 8 --steps 30 --no-events \
  > /tmp/kt_batch.log 2>&1
db=$(find /tmp/kt_batch -name "*.db" | head -1)
[ -n "$db" ] && python $R/scripts/rocpd_summary.py $db $OUT/${TAG}_batch8_kernel_trace_stats.csv > /dev/null
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$ctr -o pmc -- python $R/bench.py --config C2 --steps 10 --warmup 2 \
    --no-events --no-cpu-baseline --no-extras --batch-clouds 0 --spin-seconds 0 > /tmp/pmc_$ctr.log 2>&1
done
rm -rf /tmp/pmc_sq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
  SQ_INSTS_VALU SQ_WAVES -d /tmp/pmc_sq -o sq -- python $R/bench.py --config C2 --steps 10 --warmup 2 --no-events --no-cpu-baseline --no-extras \
  --batch-clouds 0 --spin-seconds 0 > /tmp/pmc_sq.log 2>&1
q=$(find /tmp/pmc_sq -name "*.db" | head -1); [ -n "$q" ] && python $R/scripts/pmc_table.py $q > $OUT/${TAG}_c2_sq_counters.txt
f=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); w=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
[ -n "$f" ] && [ -n "$w" ] && python $R/scripts/pmc_traffic.py $f $w C2:det $OUT/${TAG}_pmc_traffic.json > /dev/null
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json 2>/dev/null   # bench.py reads roofline.traffic from here
cp $OUT/${TAG}_c2_sq_counters.txt $R/profiles/${TAG}_c2_sq_counters.txt 2>/dev/null   # ... and roofline.issue from here
cd $R
timeout 400 python bench.py --config C2 --steps 20 --warmup 5 > $OUT/${TAG}_bench_c2.json 2> /dev/null   # the driver's own command line
# the multi-GPU line as far as one GPU can show it: the sharded entry points on an RCCL communicator of ONE rank (--dist), first
# the default series (C5, one cloud per GPU; c2_ / c4_sample_sharded ride along), then sample sharding as the headline
timeout 400 python bench.py --gpus 1 --dist --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | grep '^{' > $OUT/${TAG}_bench_c5_cloud_per_gpu_x1.json
timeout 300 python bench.py --gpus 1 --dist --shard samples --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> /dev/null | grep '^{' > $OUT/${TAG}_bench_c2_sharded_x1.json
# what one rank of an N-GPU sample-sharded run executes before the exchange
timeout 300 python scripts/slice_timing.py 2> /dev/null > $OUT/${TAG}_slice_timing.txt
# the host-buffer entry points from plain C++, and their timeline
timeout 200 bash scripts/host_api_c.sh C2 50 2> /dev/null | grep '^{' > $OUT/${TAG}_host_api_c.jsonl
timeout 400 bash scripts/host_timeline.sh ${TAG} > /dev/null 2>&1
cp $R/gpurun_out/${TAG}_host_timeline_api.txt $R/gpurun_out/${TAG}_host_timeline_pipeline.txt $R/gpurun_out/${TAG}_host_timeline_api.json \
  $R/gpurun_out/${TAG}_host_timeline_pipeline.json $OUT/ 2> /dev/null
timeout 300 python bench.py --config C3 > $OUT/${TAG}_bench_c3.json 2> /dev/null
timeout 300 python bench.py --config C2 --normals rand50 --no-cpu-baseline > $OUT/${TAG}_bench_c2_rand50.json 2> /dev/null
timeout 300 python bench.py --config C4 --steps 20 --no-cpu-baseline > $OUT/${TAG}_bench_c4.json 2> /dev/null
timeout 300 python scripts/preprocess_bench.py 2> /dev/null | grep raw_points > $OUT/${TAG}_preprocess.jsonl
timeout 300 python scripts/handles_bench.py 2> /dev/null | grep '"hands"' > $OUT/${TAG}_handles.jsonl
timeout 300 python scripts/pipeline_bench.py 2> /dev/null | tail -1 > $OUT/${TAG}_pipeline.json
# a stream of captures: agh_localize against agh_localize_begin / _stage / _end (the next upload under this capture's kernels)
timeout 300 python scripts/micro/pipeline_overlap.py 40 2> /dev/null | grep '^{' > $OUT/${TAG}_pipeline_overlap.json
# the axis-aligned scene of SURVEY 8d (K1c's exhaustive argmax) under the kernel trace
rm -rf /tmp/kt_C2u
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_C2u -o kt -- python $R/scripts/micro/quick_bench.py C2u --steps 50 > /tmp/kt_C2u.log 2>&1)
db=$(find /tmp/kt_C2u -name "*.db" | head -1)
[ -n "$db" ] && python $R/scripts/rocpd_summary.py $db $OUT/${TAG}_c2u_kernel_trace_stats.csv > /dev/null
# the one-call chain's kernel budget (agh_localize in a loop under rocprofv3 --kernel-trace)
timeout 400 bash scripts/localize_trace.sh ${TAG}_localize > $OUT/${TAG}_localize_kernels_per_call.txt 2>&1
cp $R/gpurun_out/${TAG}_localize_kernel_trace_stats.csv $OUT/ 2> /dev/null
cd $R
timeout 600 python scripts/train_bench.py 2> /dev/null | grep '"instances"' > $OUT/${TAG}_training.jsonl
rm -rf /tmp/kt_train
(cd /tmp && TRAIN_BENCH_CPU_N=300 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_train -o kt -- python $R/scripts/train_bench.py > /tmp/kt_train.log 2>&1)
db=$(find /tmp/kt_train -name "*.db" | head -1)
[ -n "$db" ] && python $R/scripts/rocpd_summary.py $db $OUT/${TAG}_training_kernel_trace_stats.csv > /dev/null
ls -la $OUT
head -c 600 $OUT/${TAG}_bench_c2.json; echo
head -8 $OUT/${TAG}_c2_kernel_trace_stats.csv
