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
    # scanned columns of a chunk incl. its two 50 kb flanks (generate_SNP_pileups.py:156): the weight when nothing is known about the depth
    return c['end'] - c['start'] + 1 + 100_000


def bai_linear_index(bai_path):
    """{reference index: [file offset (bytes) of the first BGZF block holding an alignment that overlaps each 16 kb window]} of a BAI file
    (SAM specification 5.2: magic, n_ref, per reference the bins with their chunks, then n_intv 64-bit virtual offsets).  Windows no
    alignment overlaps carry the offset of the next one that does."""
    import struct
    out = {}
    with open(bai_path, "rb") as f:
        buf = f.read()
    if buf[:4] != b"BAI\1":
        raise ValueError("%s is not a BAI file" % bai_path)
    n_ref, = struct.unpack_from("<i", buf, 4)
    o = 8
    for r in range(n_ref):
        n_bin, = struct.unpack_from("<i", buf, o)
        o += 4
        for _ in range(n_bin):
            _, n_chunk = struct.unpack_from("<Ii", buf, o)
            o += 8 + 16 * n_chunk
        n_intv, = struct.unpack_from("<i", buf, o)
        o += 4
        iv = [v >> 16 for v in struct.unpack_from("<%dQ" % n_intv, buf, o)]
        o += 8 * n_intv
        nxt = 0
        for k in range(n_intv - 1, -1, -1):                         # empty windows (0): the next window's offset
            if iv[k] == 0:
                iv[k] = nxt
            nxt = iv[k]
        out[r] = iv
    return out


def depth_weights(sam_path, chunks, ref_names=None):
    """shards are balanced by the alignments they hold (sum of depth), not by their length.  -> one weight per chunk = the
    compressed BAM bytes between the first alignment overlapping the chunk's scan span (chunk +- 50 kb) and the first one behind it, read off
    the BAI linear index (16 kb windows) -- proportional to sum(depth) x bytes per base, no decode; scaled so that the mean weight is the
    mean scanned length (a contig's last windows, which have no successor offset, fall back to that density).  None when there is no BAI
    beside the BAM (CSI-indexed or in-memory inputs: the column count is used).  Every rank computes the same numbers from the same file."""
    import os
    if not isinstance(sam_path, str):
        return None
    bai = sam_path + ".bai" if os.path.exists(sam_path + ".bai") else os.path.splitext(sam_path)[0] + ".bai"
    if not os.path.exists(bai):
        return None
    try:
        lin = bai_linear_index(bai)
        if ref_names is None:
            from .bam import BamFile
            bf = BamFile(sam_path)
            try:
                ref_names = list(bf.references)
            finally:
                bf.close()
    except Exception:
        return None
    tid = {n: i for i, n in enumerate(ref_names)}
    raw = []
    for c in chunks:
        iv = lin.get(tid.get(c['chrom'], -1))
        lo, hi = max(0, c['start'] - 50_000) >> 14, (c['end'] + 50_000) >> 14
        if not iv or lo >= len(iv):
            raw.append(None)
            continue
        a = iv[lo]
        b = iv[hi + 1] if hi + 1 < len(iv) else None
        raw.append(max(0, b - a) if b is not None and a else None)
    known = [(w, chunk_weight(c)) for w, c in zip(raw, chunks) if w is not None and w > 0]
    if not known:
        return None
    per_col = sum(w for w, _ in known) / float(sum(n for _, n in known))           # bytes per scanned column
    return [(w if w is not None else per_col * chunk_weight(c)) / per_col for w, c in zip(raw, chunks)]


def shard_plan(chunks, world, snap=0.10, weights=None):
    """-> list (per rank) of chunk lists: CONTIGUOUS blocks of the chunk list balanced by `weights` (depth_weights: the alignments a chunk
    holds; default: its scanned columns), with every cut that lies
    within `snap` x (a rank's share) of a contig boundary moved onto it.  A rank decodes and uploads only what its chunks need
    (snpCaller.call_chunks: whole contigs, or -- for the at most two contigs a block shares with its neighbours -- the span of its
    chunks +- the 50 kb scan flank), so no part of a contig is decoded by two ranks beyond that flank.  Chunk boundaries are never
    moved (they define the coverage constant E2 and the duplicated boundary record E3)."""
    chunks = list(chunks)
    if world <= 1:
        return [chunks]
    w = [float(x) for x in weights] if weights is not None and len(weights) == len(chunks) else [chunk_weight(c) for c in chunks]
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


def shard_chunks(chunks, rank, world, weights=None):
    """this rank's chunks of shard_plan()"""
    return shard_plan(chunks, world, weights=weights)[rank] if world > 1 else list(chunks)


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
