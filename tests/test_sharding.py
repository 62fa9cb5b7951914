"""Multi-process tests of the N > 1 path on CPU (gloo, world_size 2): slice bookkeeping and the one-collective
exchange of fixed-slot hypothesis records."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_slices_partition_the_samples():
    from agile_grasp_amd.sharding import shard_slice

    for n in (0, 1, 7, 2000, 2001, 8000):
        for world in (1, 2, 3, 8):
            sl = [shard_slice(n, r, world) for r in range(world)]
            assert sl[0].start == 0 and sl[-1].stop == n
            for a, b in zip(sl, sl[1:]):
                assert a.stop == b.start
            sizes = [s.stop - s.start for s in sl]
            assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from agile_grasp_amd import sharding, binding
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
S = 5
rng = np.random.default_rng(100 + rank)
n = 3 + 4 * rank
recs = np.zeros(n, binding.HYP_DTYPE)
recs["sample"] = np.sort(rng.integers(0, S, n))
recs["width"] = rng.random(n)
recs["valid"] = 1
buf = np.zeros(sharding.buffer_bytes(S), np.uint8)
buf[:8] = np.frombuffer(np.int64(n).tobytes(), np.uint8)
buf[160:160 + n * 160] = np.frombuffer(recs.tobytes(), np.uint8)
buf[160 + n * 160:] = 0xAB  # stale bytes behind the valid records must be ignored
local = torch.from_numpy(buf)
gathered = torch.zeros(world * buf.size, dtype=torch.uint8)
sharding.all_gather_records(local, gathered)
parts = sharding.unpack_gathered(gathered.numpy(), world, binding.HYP_DTYPE)
ok = True
for g in range(world):
    r2 = np.random.default_rng(100 + g)
    n2 = 3 + 4 * g
    s2 = np.sort(r2.integers(0, S, n2)); w2 = r2.random(n2)
    ok &= len(parts[g]) == n2 and np.array_equal(parts[g]["sample"], s2) and np.array_equal(parts[g]["width"], w2)
# compact exchange: only the header + 7 slots travel; rank 1 (7 records) just fits, 6 slots must be refused
nb = sharding.buffer_bytes_records(7)
g2 = torch.zeros(world * nb, dtype=torch.uint8)
sharding.all_gather_records(local[:nb], g2)
p2 = sharding.unpack_gathered(g2.numpy(), world, binding.HYP_DTYPE)
ok &= all(np.array_equal(a, b) for a, b in zip(p2, parts))
nb = sharding.buffer_bytes_records(6)
g3 = torch.zeros(world * nb, dtype=torch.uint8)
sharding.all_gather_records(local[:nb], g3)
try:
    sharding.unpack_gathered(g3.numpy(), world, binding.HYP_DTYPE)
    ok = False
except OverflowError:
    pass
slices = [sharding.shard_slice(world * S, g, world) for g in range(world)]
merged = sharding.merge_sample_sharded(parts, slices)
ok &= len(merged) == sum(3 + 4 * g for g in range(world)) and bool((np.diff(merged["sample"]) >= 0).all())
dist.barrier()
dist.destroy_process_group()
print("RANK", rank, "OK" if ok else "FAIL")
sys.exit(0 if ok else 1)
'''


def test_two_rank_all_gather_of_fixed_slot_records(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   AGH_NO_TORCH="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "OK" in o
