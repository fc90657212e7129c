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


def shard_plan(chunks, world, snap=0.10):
    """-> list (per rank) of chunk lists: CONTIGUOUS blocks of the chunk list balanced by scanned columns, with every cut that lies
    within `snap` x (a rank's share) of a contig boundary moved onto it.  A rank decodes and uploads only what its chunks need
    (snpCaller.call_chunks: whole contigs, or -- for the at most two contigs a block shares with its neighbours -- the span of its
    chunks +- the 50 kb scan flank), so no part of a contig is decoded by two ranks beyond that flank.  Chunk boundaries are never
    moved (they define the coverage constant E2 and the duplicated boundary record E3)."""
    chunks = list(chunks)
    if world <= 1:
        return [chunks]
    w = [chunk_weight(c) for c in chunks]
    total = float(sum(w)) or 1.0
    share = total / world
    pre = [0.0]
    for wi in w:
        pre.append(pre[-1] + wi)
    # first chunk of every block: the chunk whose midpoint crosses k * share
    cuts = []
    for k in range(1, world):
        i = 0
        while i < len(chunks) and pre[i] + w[i] / 2.0 < k * share:
            i += 1
        cuts.append(i)
    bounds = [i for i in range(1, len(chunks)) if chunks[i]['chrom'] != chunks[i - 1]['chrom']]
    out_cuts = []
    for i in cuts:
        best = i
        near = [b for b in bounds if abs(pre[b] - pre[i]) <= snap * share]
        if near:
            best = min(near, key=lambda b: abs(pre[b] - pre[i]))
        out_cuts.append(best)
    out_cuts = sorted(out_cuts)
    edges = [0] + out_cuts + [len(chunks)]
    return [chunks[edges[r]:edges[r + 1]] for r in range(world)]


def shard_chunks(chunks, rank, world):
    """this rank's chunks of shard_plan()"""
    return shard_plan(chunks, world)[rank] if world > 1 else list(chunks)


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
