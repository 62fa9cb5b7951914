"""Multi-GPU sharding of the grasp search (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

The path shards over independent units and needs no data-path collective:
  * samples of one cloud are independent work items (reference OMP loops A/B, hand_search.cpp:77-80,135-138):
    ``shard_slice`` hands rank g the contiguous slice g of the sample list, so concatenating the ranks' result
    lists in rank order reproduces the single-GPU (= reference) order;
  * clouds of a batch are independent: ``bench.py --gpus N`` gives every rank one cloud.
The only exchange is the optional hand-over of the results: ONE all-gather per step of a fixed-size buffer
``[header | 8*S records of 160 B]`` whose header carries the rank's record count (no variable-count exchange;
xGMI ring all-gathers of a few MB are latency bound, so one large fixed collective beats several small ones).
"""
from __future__ import annotations

import numpy as np

RECORD_BYTES = 160


def shard_slice(n_items: int, rank: int, world: int) -> slice:
    """Contiguous, balanced slice of rank ``rank`` (first ``n_items % world`` ranks get one extra item)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return slice(lo, lo + q + (1 if rank < r else 0))


def buffer_bytes(n_samples: int) -> int:
    """Bytes of one rank's exchange buffer: a 160-byte header (int64 count first) + 8 slots per sample."""
    return (8 * n_samples + 1) * RECORD_BYTES


def buffer_bytes_records(n_records: int) -> int:
    """Bytes of the buffer prefix holding the header and the first ``n_records`` record slots.  Exchanging a prefix
    is valid because records are compacted to the front; receivers compare the header count with the slots sent."""
    return (n_records + 1) * RECORD_BYTES


def all_gather_records(local_t, gather_t):
    """One collective: every rank contributes its whole fixed-size buffer; gather_t is world * len(local_t)."""
    import torch.distributed as dist

    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(gather_t, local_t)
    else:  # gloo (CPU tests)
        world = dist.get_world_size()
        parts = list(gather_t.view(world, -1).unbind(0))
        dist.all_gather(parts, local_t)
        for i, p in enumerate(parts):
            gather_t.view(world, -1)[i].copy_(p)
    return gather_t


def unpack_gathered(gathered: np.ndarray, world: int, dtype: np.dtype):
    """Split an all-gathered byte buffer into the per-rank record arrays (rank order = reference order)."""
    per = gathered.size // world
    out = []
    for g in range(world):
        blob = gathered[g * per:(g + 1) * per]
        n = int(np.frombuffer(blob[:8].tobytes(), np.int64)[0])
        if (n + 1) * RECORD_BYTES > per:
            raise OverflowError(f"rank {g} produced {n} records but only {per // RECORD_BYTES - 1} slots were exchanged")
        recs = np.frombuffer(blob[RECORD_BYTES:RECORD_BYTES + n * RECORD_BYTES].tobytes(), dtype)
        out.append(recs)
    return out


def merge_sample_sharded(per_rank_records, slices):
    """Concatenate sample-sharded results, re-basing the per-rank sample positions to the global sample list."""
    merged = []
    for recs, sl in zip(per_rank_records, slices):
        r = recs.copy()
        r["sample"] += sl.start
        merged.append(r)
    return np.concatenate(merged) if merged else np.zeros(0)
