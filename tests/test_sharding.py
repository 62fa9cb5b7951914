"""The N > 1 path on CPU: slice bookkeeping (the C ABI's agh_shard_slice), and a world_size-2 gloo run in which every
rank searches ITS slice of one cloud's samples (with the oracle -- there is no GPU here), contributes one fixed-size
segment to one all-gather, and the merged list equals the single-rank list byte for byte.  The HIP / RCCL implementation
of the same schedule (csrc/shard.hip) is tested on the GPU in test_gpu_sharding.py."""
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


def test_a_batch_of_equal_clouds_shards_in_cloud_order():
    """BASELINE config C5 (bench.py --gpus N): G clouds of S samples each, the concatenated list sharded G ways -- rank r's
    slice is exactly cloud r's samples, so "samples sharded across the GPUs" and "one cloud per GPU" are the same schedule."""
    from agile_grasp_amd.sharding import shard_slice

    for S in (1, 64, 2000, 8000):
        for G in (1, 2, 4, 8):
            for r in range(G):
                sl = shard_slice(G * S, r, G)
                assert (sl.start, sl.stop) == (r * S, (r + 1) * S)


def test_host_side_slices_equal_the_c_abi():
    """sharding.shard_slice restates agh_shard_slice (so that the bookkeeping needs no built library); the two must agree."""
    from agile_grasp_amd import binding
    from agile_grasp_amd.sharding import shard_slice

    for n in (0, 1, 7, 2000, 2001, 8000, 300_000):
        for world in (1, 2, 3, 5, 8, 64):
            for r in range(world):
                sl = shard_slice(n, r, world)
                assert (sl.start, sl.stop) == binding.shard_slice(n, r, world)


def test_segment_sizes():
    from agile_grasp_amd import sharding

    assert sharding.segment_records(2000, 8) == 1024  # 250 samples per rank: the 1024-record floor
    assert sharding.segment_records(8000, 8) == 2000  # 2 per sample
    assert sharding.segment_records(64, 2) == 256     # never more than 8 per sample
    assert sharding.segment_records(2000, 8, full=True) == 2000
    assert sharding.segment_bytes(1024) == 160 + 1024 * 160


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from agile_grasp_amd import sharding, synthetic
from oracle import oracle_py as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
sc = synthetic.config("tiny")
p = O.default_params(sc.cam_origins)
S = sc.samples.size
sl = sharding.shard_slice(S, rank, world)
mine = O.find_hands(p, sc.xyz, sc.cam, sc.samples[sl])["hyps"]           # this rank's slice of the SAME cloud
full = O.find_hands(p, sc.xyz, sc.cam, sc.samples)["hyps"]                 # what one rank alone finds
ok = True
for seg in (sharding.segment_records(S, world), 4):                       # the default segment, and one that overflows
    local = torch.from_numpy(sharding.pack_segment(mine, seg))
    gathered = torch.zeros(world * local.numel(), dtype=torch.uint8)
    sharding.all_gather_records(local, gathered)
    try:
        merged = sharding.merge_segments(gathered.numpy(), world, S, seg, O.HYP_DTYPE)
        ok &= seg != 4 and merged.tobytes() == full.tobytes() and len(full) > 8
    except OverflowError:
        ok &= seg == 4                                                     # every rank sees the same headers
dist.barrier()
dist.destroy_process_group()
print("RANK", rank, "OK" if ok else "FAIL", len(mine), len(full))
sys.exit(0 if ok else 1)
'''


def test_two_rank_sample_sharded_search_equals_single_rank(tmp_path):
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
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "OK" in o
