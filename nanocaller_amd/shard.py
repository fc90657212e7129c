"""Multi-GPU sharding of the path: one process per GPU, chunks shard embarrassingly.

The reference runs `cpu` worker processes that pull chunks from a queue and use files on disk as the gather
medium (snpCaller.py:238-241, 278-280).  Here every rank owns a contiguous, depth-weighted block of the chunk
list, writes its own per-rank VCF, and rank 0 merges -- host scatter / gather only, no collective on the data
path (per-site results are ~80 B; there is nothing to exchange between regions).  torch.distributed is used for
rendezvous, the timing barrier and two scalar reductions.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def chunk_weight(c):
    # scanned columns of a chunk incl. its two 50 kb flanks (generate_SNP_pileups.py:156)
    return c['end'] - c['start'] + 1 + 100_000


def shard_chunks(chunks, rank, world):
    """Contiguous block partition balanced by scanned columns.  Chunk boundaries are never moved (they define the
    coverage constant E2 and the duplicated boundary record E3)."""
    if world <= 1:
        return list(chunks)
    w = [chunk_weight(c) for c in chunks]
    total = float(sum(w))
    out, acc, r = [[] for _ in range(world)], 0.0, 0
    for c, wi in zip(chunks, w):
        # advance to the rank whose [r, r+1) * total/world band contains this chunk's midpoint
        mid = acc + wi / 2.0
        r = min(world - 1, int(mid * world / total))
        out[r].append(c)
        acc += wi
    return out[rank]


def shard_range(n_items, rank, world):
    """contiguous, near-equal block of range(n_items) owned by `rank` (bench.py: the job's contig list over the GPUs)"""
    return range(rank * n_items // world, (rank + 1) * n_items // world)


def _dev(device):
    return device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")


def dist_max(x: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=_dev(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def dist_sum(x: int, device=None) -> int:
    if not (dist.is_available() and dist.is_initialized()):
        return int(x)
    t = torch.tensor([x], dtype=torch.int64, device=_dev(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
