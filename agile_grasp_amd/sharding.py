"""Bookkeeping of the sample-sharded search (one process per GPU), mirrored from csrc/shard.hip for the host side of
bench.py and for the CPU (gloo) tests.  The data path itself is C++ + RCCL: agh_find_hands_sharded_device.

Rank g of G takes the contiguous slice [g*S/G, (g+1)*S/G) of the sample list (agh_shard_slice) -- the reference's OpenMP
loops run over independent samples (hand_search.cpp:77-80, 135-138), so the ranks' lists concatenated in rank order ARE
the reference's sample-major list.  Each rank contributes one fixed-size segment
``[160-byte header: int64 count, int64 flags | seg_records records of 160 B]`` to ONE in-place all-gather (flags bit 0: the
rank met a Taubin neighbourhood beyond the capacity classes it had launched -- every rank then reports AGH_ERR_RETRY
together; this host-side mirror leaves it zero).
"""
from __future__ import annotations

import numpy as np

RECORD_BYTES = 160
HEADER_BYTES = 160


def shard_slice(n_items: int, rank: int, world: int) -> slice:
    """The slice rank ``rank`` of ``world`` takes: [n r / G, n (r + 1) / G) in integer arithmetic -- agh_shard_slice of the C ABI
    (csrc/shard.hip: shard_lo), restated here so that the host-side bookkeeping needs no built library;
    tests/test_sharding.py checks the two against each other."""
    world = max(int(world), 1)
    return slice((n_items * rank) // world, (n_items * (rank + 1)) // world)


def segment_records(n_samples: int, world: int, full: bool = False) -> int:
    """Record slots of one rank's segment: max(2 ceil(S/G), 1024), never more than 8 ceil(S/G) (`full`: 8 ceil(S/G))."""
    smax = -(-n_samples // world)
    return 8 * smax if full else min(8 * smax, max(2 * smax, 1024))


def segment_bytes(seg_records: int) -> int:
    return HEADER_BYTES + seg_records * RECORD_BYTES


def pack_segment(records: np.ndarray, seg_records: int) -> np.ndarray:
    """A rank's compacted records as its exchange segment.  The header carries the TRUE count; records beyond the
    segment are dropped, which the receivers detect (count > seg_records)."""
    buf = np.zeros(segment_bytes(seg_records), np.uint8)
    buf[:8] = np.frombuffer(np.int64(len(records)).tobytes(), np.uint8)
    k = min(len(records), seg_records)
    buf[HEADER_BYTES:HEADER_BYTES + k * RECORD_BYTES] = np.frombuffer(records[:k].tobytes(), np.uint8)
    return buf


def all_gather_records(local_t, gather_t):
    """One collective through torch.distributed (the CPU tests' gloo, or RCCL when bench.py falls back to torch)."""
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


def merge_segments(gathered: np.ndarray, world: int, n_samples: int, seg_records: int, dtype: np.dtype) -> np.ndarray:
    """k_shard_merge on the host: concatenate the ranks' records in rank order, sample positions re-based to the full
    list.  Raises OverflowError if a rank found more than its segment holds (the caller repeats with full segments)."""
    per = segment_bytes(seg_records)
    assert gathered.size == world * per
    out = []
    for g in range(world):
        blob = gathered[g * per:(g + 1) * per]
        n = int(np.frombuffer(blob[:8].tobytes(), np.int64)[0])
        if n > seg_records:
            raise OverflowError(f"rank {g} found {n} hypotheses, its segment holds {seg_records}")
        recs = np.frombuffer(blob[HEADER_BYTES:HEADER_BYTES + n * RECORD_BYTES].tobytes(), dtype).copy()
        recs["sample"] += shard_slice(n_samples, g, world).start
        out.append(recs)
    return np.concatenate(out) if out else np.zeros(0, dtype)


# ---- round-1 measurement scripts (scripts/host_overhead.py, xstream_experiment*.py) size their buffers with these ----
def buffer_bytes(n_samples: int) -> int:
    """A segment with all 8 slots per sample."""
    return segment_bytes(8 * n_samples)


def buffer_bytes_records(n_records: int) -> int:
    return segment_bytes(n_records)
