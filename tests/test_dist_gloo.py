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


def test_depth_weights_follow_the_alignments_not_the_lengths(tmp_path):
    """shards are balanced by sum(depth).  A BAM whose second half carries four times the reads of its first half: the weights read off
    the BAI linear index (no decode) say so, and the two-rank cut moves from the middle of the contig towards the dense half"""
    from nanocaller_amd.shard import bai_linear_index, depth_weights, shard_plan
    from tests import bamio
    rng = np.random.default_rng(8)
    L = 2_000_000
    recs = []
    for p in range(1_000, L - 3_000, 2_000):
        for k in range(1 if p < L // 2 else 4):
            recs.append(dict(name="r%d_%d" % (p, k), flag=0, pos0=p + 17 * k, cigar=[("M", 2_000)], seq="".join("ACGT"[b] for b in rng.integers(0, 4, 2_000))))
    recs.sort(key=lambda r: r["pos0"])
    bam = str(tmp_path / "w.bam")
    bamio.write_bam(bam, "c1", L, recs, level=1)
    lin = bai_linear_index(bam + ".bai")
    assert list(lin) == [0] and len(lin[0]) >= (L - 3_000) >> 14 and all(b >= a for a, b in zip(lin[0], lin[0][1:]))
    chunks = get_chunks([("c1", 1, L, "diploid")], cpu=40)
    w = depth_weights(bam, chunks)
    assert w is not None and len(w) == len(chunks)
    first = np.mean([x for x, c in zip(w, chunks) if c["end"] < L // 2 - 100_000])
    second = np.mean([x for x, c in zip(w, chunks) if c["start"] > L // 2 + 100_000 and c["end"] < L - 200_000])
    assert 2.5 < second / first < 6.0
    by_len = shard_plan(chunks, 2)
    by_depth = shard_plan(chunks, 2, weights=w)
    assert [c for p in by_depth for c in p] == chunks
    assert abs(by_len[0][-1]["end"] - L // 2) < 150_000 and by_depth[0][-1]["end"] > L // 2 + 250_000
    assert depth_weights(str(tmp_path / "none.bam"), chunks) is None and depth_weights(object(), chunks) is None


def test_numa_binding_plan():
    """ranks split the CPUs of the NUMA node their GPUs hang off; unknown topology = no binding"""
    from nanocaller_amd import numa
    cpus = {0: list(range(0, 64)), 1: list(range(64, 128))}
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]                                # an 8-GPU node, four GPUs per socket
    allowed = set(range(128))
    seen = []
    for lr in range(8):
        got, note = numa.plan_binding(nodes, lr, allowed, lambda n: cpus[n])
        assert len(got) == 16 and set(got) <= set(cpus[nodes[lr]]), note
        seen += got
    assert sorted(seen) == list(range(128))                         # disjoint and complete
    got, _ = numa.plan_binding(nodes, 5, set(range(64, 72)), lambda n: cpus[n])      # a cgroup that allows 8 CPUs of socket 1
    assert got == [66, 67]
    assert numa.plan_binding([-1] * 8, 3, allowed, lambda n: cpus[n])[0] is None
    assert numa.plan_binding(nodes, 2, set(range(64, 128)), lambda n: cpus[n])[0] is None   # no allowed CPU next to the GPU: leave the mask alone
    assert numa._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert numa.bind_rank(0)["bound"] is False                      # no GPU here: nothing is bound, nothing fails


def _wgs_worker(rank, world, port, tmpdir):
    """one rank of an 8-rank job over a WGS-shaped contig list: indelCaller.call_manager with the per-chunk work stood in"""
    import torch
    import torch.distributed as dist

    from nanocaller_amd import indelCaller
    from tests.test_boundary import _fake_snp_vcf
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: 8
    seen = []

    def fake_indel_run(params, indel_dict, job_Q, counter_Q, files, device=0, worker_id=1, aligner=None):
        path = os.path.join(params["intermediate_indel_files_dir"], "%s.%d.indel.vcf" % (params["prefix"], worker_id))
        files.append(path)
        with open(path, "a") as f:
            while not job_Q.empty():
                kind, chunk = job_Q.get()
                seen.append((device, worker_id, chunk["chrom"], chunk["start"]))
                f.write("%s\t%d\t.\tAT\tA\t12.00\tPASS\t.\tGT:GQ\t0|1:3.00\n" % (chunk["chrom"], chunk["start"] + 7))
    indelCaller.indel_run = fake_indel_run
    indelCaller._whatshap_available = lambda: False
    regions = [("chr%d" % (k + 1), 1, n * 10_000, "diploid") for k, n in enumerate(WGS_LENS)]
    snp_vcf = os.path.join(tmpdir, "t.snps.vcf.gz")
    if rank == 0:
        _fake_snp_vcf(snp_vcf, [r[0] for r in regions])
    dist.barrier()
    params = dict(chunks_list=get_chunks(regions, 16, max_chunk_size=10_000), mode="all", snp_vcf=snp_vcf, regions_list=regions, sam_path="in.bam",
                  fasta_path="x.fa", vcf_path=tmpdir, prefix="t", sample="S", phase_qual_score=10, suppress_progress=True, verbose=False,
                  enable_whatshap=False, cpu=2)
    out = indelCaller.call_manager(params)
    with open(os.path.join(tmpdir, "wlog.%d" % rank), "w") as f:
        f.write(repr((out, seen)))
    dist.destroy_process_group()


WGS_LENS = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57]


def test_eight_rank_gloo_wgs_plan_through_the_indel_call_manager(tmp_path):
    """BASELINE.json configs[3]'s shape on CPU: 24 contigs (GRCh38's proportions), 3,078 indel chunks, eight ranks over gloo through
    indelCaller.call_manager (rank 0 phases every contig, every rank releases ITS block of chunks to its own GPU's worker, rank 0 merges): each chunk
    is worked on exactly once, in contiguous contig-aware blocks of near-equal size, and the merged outputs are sorted"""
    from tests.test_boundary import _records
    world = 8
    mp.spawn(_wgs_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    logs = [eval(open(os.path.join(str(tmp_path), "wlog.%d" % r)).read()) for r in range(world)]
    regions = [("chr%d" % (k + 1), 1, n * 10_000, "diploid") for k, n in enumerate(WGS_LENS)]
    chunks = get_chunks(regions, 16, max_chunk_size=10_000)
    assert len(chunks) > 3000
    got = [(c, s) for lg in logs for (_, _, c, s) in lg[1]]
    assert len(got) == len(chunks) and sorted(got) == sorted((c["chrom"], c["start"]) for c in chunks)        # every chunk once
    order = {(c["chrom"], c["start"]): i for i, c in enumerate(chunks)}
    for r, lg in enumerate(logs):
        assert {d for (d, _, _, _) in lg[1]} == {r} and {w for (_, w, _, _) in lg[1]} == {r + 1}              # rank -> its GPU, its worker file
        idx = sorted(order[(c, s)] for (_, _, c, s) in lg[1])
        assert idx == list(range(idx[0], idx[-1] + 1))                                                        # one contiguous block
        assert 0.8 * len(chunks) / world < len(idx) < 1.2 * len(chunks) / world
    assert all(lg[0] == logs[0][0] for lg in logs)
    ind = _records(logs[0][0]["indels"])
    assert len(ind) == len(chunks)
    keys = [(int(r.split("\t")[0][3:]), int(r.split("\t")[1])) for r in ind]
    assert keys == sorted(keys)


def _merge_worker(rank, world, port, tmpdir):
    """one rank of the indel half over a small 5-contig list: every chunk yields two records (the second BEFORE the first in position, as two overlapping
    chunks' records can be) so that the merge has to order what the workers wrote"""
    import torch
    import torch.distributed as dist

    from nanocaller_amd import indelCaller
    from tests.test_boundary import _fake_snp_vcf
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: 8

    def fake_indel_run(params, indel_dict, job_Q, counter_Q, files, device=0, worker_id=1, aligner=None):
        path = os.path.join(params["intermediate_indel_files_dir"], "%s.%d.indel.vcf" % (params["prefix"], worker_id))
        files.append(path)
        with open(path, "a") as f:
            while not job_Q.empty():
                kind, chunk = job_Q.get()
                for off, alt in ((900, "ATT"), (40, "A")):
                    f.write("%s\t%d\t.\tAT\t%s\t%d.00\tPASS\t.\tGT:GQ\t0|1:3.00\n" % (chunk["chrom"], chunk["start"] + off, alt, 10 + chunk["start"] % 7))
    indelCaller.indel_run = fake_indel_run
    indelCaller._whatshap_available = lambda: False
    regions = [("chr%d" % (k + 1), 1, n * 1_000, "diploid") for k, n in enumerate([61, 17, 33, 9, 48])]
    snp_vcf = os.path.join(tmpdir, "t.snps.vcf.gz")
    if rank == 0:
        _fake_snp_vcf(snp_vcf, [r[0] for r in regions])
    dist.barrier()
    params = dict(chunks_list=get_chunks(regions, 16, max_chunk_size=3_000), mode="all", snp_vcf=snp_vcf, regions_list=regions, sam_path="in.bam",
                  fasta_path="x.fa", vcf_path=tmpdir, prefix="t", sample="S", phase_qual_score=10, suppress_progress=True, verbose=False,
                  enable_whatshap=False, cpu=2)
    out = indelCaller.call_manager(params)
    with open(os.path.join(tmpdir, "out.%d" % rank), "w") as f:
        f.write(repr(out))
    dist.destroy_process_group()


def test_two_rank_indel_merge_equals_the_single_process_file(tmp_path):
    """the indel half's gather (indelCaller.py:290-353: per-worker files, merged and sorted by rank 0): two ranks' per-rank files merge into the SAME records
    in the SAME order as one process writes -- headers aside, the .indels.vcf.gz and the final .vcf.gz are identical"""
    from tests.test_boundary import _records
    outs = {}
    for world in (1, 2):
        d = tmp_path / ("w%d" % world)
        d.mkdir()
        mp.spawn(_merge_worker, args=(world, _free_port(), str(d)), nprocs=world, join=True)
        outs[world] = eval(open(os.path.join(str(d), "out.0")).read())
    one, two = _records(outs[1]["indels"]), _records(outs[2]["indels"])
    assert len(one) > 100 and one == two
    keys = [(int(r.split("\t")[0][3:]), int(r.split("\t")[1])) for r in two]
    assert keys == sorted(keys)
    assert _records(outs[1]["final"]) == _records(outs[2]["final"])
