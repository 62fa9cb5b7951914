AGH_BENCH_FORCE_DIST=1 AGH_BENCH_FORCE_SECONDARY=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['scaling']); print(d.get('cloud_per_gpu'))"
python bench.py --steps 20 --warmup 5 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['batched']['ms_per_cloud'], d['batched']['value'])"
