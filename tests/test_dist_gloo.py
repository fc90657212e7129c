"""CPU-only, world_size 2 over gloo: chunk sharding, scalar reductions and the file-based VCF gather used by
the multi-GPU path.  The per-rank compute is stood in by the oracle (there is no GPU here); what is under test
is the host logic that N>1 runs add."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from nanocaller_amd.shard import shard_chunks
from nanocaller_amd.utils import get_chunks


def test_shards_partition_the_chunk_list():
    chunks = get_chunks([("chr1", 1, 248_956_422, "diploid"), ("chr2", 1, 242_193_529, "diploid"),
                         ("chrM", 1, 16_569, "haploid")], cpu=16)
    for world in (1, 2, 3, 8):
        parts = [shard_chunks(chunks, r, world) for r in range(world)]
        flat = [c for p in parts for c in p]
        assert flat == chunks                                       # disjoint, complete, order-preserving, contiguous
        if world > 1:
            sizes = [sum(c['end'] - c['start'] for c in p) for p in parts]
            assert max(sizes) < 1.25 * (sum(sizes) / world)         # balanced


def test_shard_cuts_snap_to_contig_boundaries_and_split_contigs_are_decoded_by_span():
    """a 24-contig genome over 8 ranks: blocks stay contiguous and balanced, cuts near a contig boundary sit ON it, and a contig shared
    by two ranks is decoded by each only over its own span +- the scan flank (generate_SNP_pileups.contig_span)"""
    from nanocaller_amd.shard import chunk_weight, shard_plan
    lens = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57]
    chunks = get_chunks([("chr%d" % (k + 1), 1, L * 1_000_000, "diploid") for k, L in enumerate(lens)], cpu=16)
    plan = shard_plan(chunks, 8)
    assert [c for p in plan for c in p] == chunks
    wts = [sum(chunk_weight(c) for c in p) for p in plan]
    assert max(wts) < 1.15 * (sum(wts) / 8)
    owners = {}
    for r, p in enumerate(plan):
        for c in p:
            owners.setdefault(c["chrom"], set()).add(r)
    assert sum(len(v) > 1 for v in owners.values()) <= 7 and max(len(v) for v in owners.values()) <= 2
    whole = sum(len(v) == 1 for v in owners.values())
    assert whole >= 17                                              # most contigs are one rank's
    # spans of a shared contig do not overlap beyond the flanks
    for name, rs in owners.items():
        if len(rs) == 2:
            a, b = sorted(rs)
            hi_a = max(c["end"] for c in plan[a] if c["chrom"] == name)
            lo_b = min(c["start"] for c in plan[b] if c["chrom"] == name)
            assert lo_b >= hi_a                                     # (chunks share one boundary position, utils.get_chunks)


def test_shard_range_partitions_a_contig_list():
    from nanocaller_amd.shard import shard_range
    for n in (8, 9, 24, 5):
        for world in (1, 2, 4, 8):
            if world > n:
                continue
            parts = [list(shard_range(n, r, world)) for r in range(world)]
            assert [i for p in parts for i in p] == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    import torch.distributed as dist

    from nanocaller_amd import snpCaller
    from nanocaller_amd.shard import barrier, dist_max, dist_sum
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle
    from tests.util import load_world
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    world_ = load_world("ont")
    chunks = get_chunks([(world_.chrom, 20_000, 120_000, "diploid")], cpu=5)
    mine = shard_chunks(chunks, rank, world)
    path, cov = get_SNP_model("ONT-HG002")
    w = Weights(path)
    dct = dict(threshold=[0.4, 0.6], mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, seq="ont")
    lines = []
    for c in mine:
        pos, ref, mat, dp, freq, depth, fwd, rev = oracle.get_snp_testing_candidates(world_, dct, c)
        rc = np.argmax(ref, 1).astype(np.int32)
        probs, _ = oracle.snp_forward(w.flat, mat[:40], rc[:40], cov / depth)      # a slice keeps the CPU test fast
        lines += snpCaller.snp_vcf_lines(world_.chrom, pos[:40], rc[:40], probs, dp[:40], freq[:40], fwd[:40], rev[:40])
    with open(os.path.join(tmpdir, "t.%d.snps.vcf" % (rank + 1)), "w") as f:
        f.writelines(lines)
    tmax = dist_max(1.0 + rank)
    total = dist_sum(len(lines))
    barrier()
    if rank == 0:
        with open(os.path.join(tmpdir, "result.txt"), "w") as f:
            f.write("%g %d %d" % (tmax, total, len(chunks)))
    dist.destroy_process_group()


def test_two_rank_gloo_run_matches_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    tmax, total, nchunks = open(os.path.join(str(tmp_path), "result.txt")).read().split()
    assert float(tmax) == 2.0 and int(nchunks) >= 4
    merged = []
    for r in range(world):
        merged += open(os.path.join(str(tmp_path), "t.%d.snps.vcf" % (r + 1))).readlines()
    assert int(total) == len(merged) > 50
    # single-process run of the same thing
    port2 = _free_port()
    single = tmp_path / "single"
    single.mkdir()
    mp.spawn(_worker, args=(1, port2, str(single)), nprocs=1, join=True)
    ref = open(os.path.join(str(single), "t.1.snps.vcf")).readlines()
    assert merged == ref                                             # contiguous shards => same order, same records


def _decode_worker(rank, world, port, bam, fa, tmpdir):
    """what a rank's ingest does for its shard: the contig-aware plan, then one decode per contig (or per span of a shared one)"""
    import json

    import torch.distributed as dist

    from nanocaller_amd import generate_SNP_pileups as gsp
    from nanocaller_amd.shard import barrier, shard_plan
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    chunks = get_chunks([("cA", 1, 900_000, "diploid"), ("cB", 1, 600_000, "diploid"), ("cC", 1, 300_000, "diploid")], cpu=4)
    mine = shard_plan(chunks, world)[rank]
    by_contig = {}
    for c in mine:
        by_contig.setdefault(c["chrom"], []).append(c)
    del gsp.DECODES[:]
    for name, grp in by_contig.items():
        span = gsp.contig_span(bam, name, grp)
        gsp._resolve(bam, name, fa, span)
        gsp._resolve(bam, name, fa, span)                            # a second request is served from the cache
    with open(os.path.join(tmpdir, "decodes.%d.json" % rank), "w") as f:
        json.dump([[d[1], list(d[2]) if d[2] else None] for d in gsp.DECODES], f)
    barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_ingest_decodes_no_contig_twice(tmp_path):
    """under the contig-aware shard plan every contig is decoded ONCE across the job; a contig shared by two ranks is decoded by
    each over its own span (+- the 50 kb scan flank) only -- the real decode path (native BAM reader) of two gloo ranks on one file"""
    import json

    from tests import bamio
    rng = np.random.default_rng(3)
    lens = dict(cA=900_000, cB=600_000, cC=300_000)
    recs = []
    for tid, (name, L) in enumerate(lens.items()):
        for p in range(1_000, L - 6_000, 25_000):                    # a sparse BAM: the test is about WHO decodes WHAT
            recs.append(dict(name="%s_%d" % (name, p), flag=0, pos0=p, tid=tid, cigar=[("M", 5_000)], seq="".join("ACGT"[b] for b in rng.integers(0, 4, 5_000))))
    bam, fa = str(tmp_path / "g.bam"), str(tmp_path / "g.fa")
    bamio.write_bam(bam, "cA", lens["cA"], recs, other_refs=[("cB", lens["cB"]), ("cC", lens["cC"])], level=1)
    bamio.write_fasta(fa, "cA", "A" * lens["cA"], extra=[("cB", "C" * lens["cB"]), ("cC", "G" * lens["cC"])])
    port = _free_port()
    mp.spawn(_decode_worker, args=(2, port, bam, fa, str(tmp_path)), nprocs=2, join=True)
    dec = [json.load(open(os.path.join(str(tmp_path), "decodes.%d.json" % r))) for r in range(2)]
    assert all(len(d) == len({(n, tuple(s) if s else None) for n, s in d}) for d in dec)          # the cache: one decode per (contig, span) and rank
    seen = {}
    for r, d in enumerate(dec):
        for name, span in d:
            seen.setdefault(name, []).append((r, span))
    assert set(seen) == set(lens)
    for name, who in seen.items():
        if len(who) == 1:
            continue                                                  # a whole contig (or most of one) is one rank's
        assert len(who) == 2 and who[0][0] != who[1][0]
        (ra, sa), (rb, sb) = sorted(who, key=lambda t: t[1][0] if t[1] else 0)
        assert sa is not None and sb is not None                      # neither rank decoded the whole contig
        assert sb[0] >= sa[1] - 2 * 50_000 - 1                        # the spans overlap by the two flanks at most
    assert sum(len(w) for w in seen.values()) <= len(lens) + 1        # at most one contig is shared between the two ranks
