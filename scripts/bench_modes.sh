set -x
python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench_n1.json; cat gpurun_out/bench_n1.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -3
AGH_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -3
AGH_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --shard clouds --config C3 2>&1 | tail -3
