"""the drop-in boundary.  CPU-only: the reference's module / function / class names under `nanocaller_src`,
`utils.get_regions_list`, the job structure and merges of `indelCaller.call_manager`, the reference's own `run(args)`
(NanoCaller:12-56, executed from /root/reference when present -- build container only) against this package, the
contig-keyed pack cache, and the rank -> GPU mapping of `snpCaller.call_manager` under a world_size-2 gloo run."""
import gzip
import os
import queue
import socket
import sys
import types

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests import bamio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


def test_reference_module_names_resolve_to_this_package():
    import nanocaller_src
    import nanocaller_src.generate_indel_pileups_haploid as gh
    import nanocaller_src.model_architect_indel as mi
    import nanocaller_src.model_architect_indels_haploid as mih
    import nanocaller_src.model_architect_SNP_haploid as msh
    from nanocaller_src import indelCaller, snpCaller
    from nanocaller_src.generate_indel_pileups import get_indel_testing_candidates      # noqa: F401
    from nanocaller_src.generate_SNP_pileups import get_snp_testing_candidates          # noqa: F401
    from nanocaller_src.model_architect import SNP_model                                # noqa: F401
    import nanocaller_amd.snpCaller
    assert snpCaller is nanocaller_amd.snpCaller
    assert callable(gh.get_indel_testing_candidates_haploid) and mi.Indel_model and mih.haploid_Indel_model and msh.haploid_SNP_model
    # the names snpCaller.py / indelCaller.py define and NanoCaller calls
    for name in ("call_manager", "caller", "get_SNP_model"):
        assert callable(getattr(snpCaller, name))
    for name in ("call_manager", "caller", "indel_run", "phase_run", "get_indel_model"):
        assert callable(getattr(indelCaller, name))
    assert set(nanocaller_src.MODULES) == {"utils", "snpCaller", "indelCaller", "generate_SNP_pileups", "generate_indel_pileups",
                                           "generate_indel_pileups_haploid", "model_architect", "model_architect_SNP_haploid",
                                           "model_architect_indel", "model_architect_indels_haploid"}


def test_model_classes_accept_the_reference_weight_paths():
    """load_weights() gets a TF checkpoint prefix / .h5 in the reference (snpCaller.py:71,78, indelCaller.py:52,57)"""
    from nanocaller_amd import weights as W
    p = W.resolve_weight_file("/x/nanocaller_src/release_data/ONT_models/SNPs/HG002_guppy4.2.2_giab-4.2.1/model-100", W.KIND_SNP)
    assert p == W.get_SNP_model("ONT-HG002")[0]
    p = W.resolve_weight_file("release_data/hifi_models/indels/HG002_giab-4.2.1/model-100", W.KIND_INDEL)
    assert p == W.get_indel_model("CCS-HG002")
    assert W.resolve_weight_file("a/b/haploid_models/SNPs/CHM13/model.24-0.9985.h5", W.KIND_SNP_HAP) == W.get_SNP_model("haploid")[0]
    assert W.resolve_weight_file("a/b/haploid_models/indels/CHM13/model.19-0.9811.h5", W.KIND_INDEL_HAP) == W.get_indel_model("haploid")
    with pytest.raises(FileNotFoundError):
        W.resolve_weight_file("/nowhere/ONT_models/SNPs/unknown/model-1", W.KIND_SNP)


@pytest.fixture(scope="module")
def two_contig_files(tmp_path_factory):
    """one BAM with two contigs (different reads on each) + its FASTA"""
    d = tmp_path_factory.mktemp("two")
    w1 = bamio.make_pass2_world(seed=3, length=30_000, depth=16)
    w2 = bamio.make_pass2_world(seed=4, length=22_000, depth=20)
    w1.chrom, w2.chrom = "chr1", "chrX"
    recs = []
    for tid, w in enumerate((w1, w2)):
        for r in bamio.world_to_records(w, None):
            r["tid"] = tid
            r["name"] = "%s_%s" % (w.chrom, r["name"])
            recs.append(r)
    bam, fa = str(d / "t.bam"), str(d / "t.fa")
    bamio.write_bam(bam, w1.chrom, w1.length, recs, other_refs=[(w2.chrom, w2.length)])
    bamio.write_fasta(fa, w1.chrom, w1.ref, extra=[(w2.chrom, w2.ref)])
    return w1, w2, bam, fa


class _Args(types.SimpleNamespace):
    pass


def _args(bam, fa, out, **kw):
    a = _Args(bam=bam, ref=fa, wgs_contigs=None, regions=None, bed=None, haploid_genome=False, haploid_X=False, mode="snps",
              cpu=2, neighbor_threshold="0.4,0.6", exclude_bed=None, mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1,
              snp_model="ONT-HG002", indel_model="ONT-HG002", output=out, prefix="t", sample="SAMPLE", sequencing="ont",
              supplementary=False, suppress_progress_bar=True, phase_qual_score=10, verbose=False,
              disable_coverage_normalization=False, phase=False, del_threshold=0.6, ins_threshold=0.4, impute_indel_phase=False,
              win_size=40, small_win_size=4, enable_whatshap=False)
    a.__dict__.update(kw)
    return a


def test_get_regions_list_from_the_bam_header(two_contig_files, tmp_path):
    from nanocaller_amd.utils import get_regions_list
    w1, w2, bam, fa = two_contig_files
    a = _args(bam, fa, str(tmp_path))
    assert get_regions_list(a) == (("chr1", 1, w1.length, "diploid"), ("chrX", 1, w2.length, "diploid"))
    a.haploid_X = True
    assert get_regions_list(a)[1] == ("chrX", 1, w2.length, "haploid")
    a.regions = ["chr1:100-2000", "chrX", "chr7", "chr1:5", "a:b:c"]
    assert get_regions_list(a) == (("chr1", 100, 2000, "diploid"), ("chrX", 1, w2.length, "haploid"))
    a.regions, a.wgs_contigs, a.haploid_genome = None, "chr1-22XY", True
    assert get_regions_list(a) == (("chr1", 1, w1.length, "haploid"), ("chrX", 1, w2.length, "haploid"))
    bed = tmp_path / "r.bed"
    bed.write_text("chr1\t10\t500\nchr9\t1\t2\nchrX\t7\t9000\n")
    a.wgs_contigs, a.bed, a.haploid_genome = None, str(bed), False
    assert get_regions_list(a) == (("chr1", 10, 500, "diploid"), ("chrX", 7, 9000, "haploid"))
    a.bed, a.regions = None, ["chrNope"]
    with pytest.raises(SystemExit):
        get_regions_list(a)


def test_pack_cache_is_keyed_by_contig_and_bounded(two_contig_files, monkeypatch):
    """the HBM pack cache must not hand the first contig's pack to the second; and it must not grow with
    the number of contigs.  (engine stubbed: the upload is the identity.)"""
    from nanocaller_amd import generate_SNP_pileups as g
    w1, w2, bam, fa = two_contig_files

    from nanocaller_amd import wire
    monkeypatch.setattr(g, "get_engine", lambda device=0: None)
    monkeypatch.setattr(wire, "upload_wire", lambda eng, wp: wp)            # the "device pack" is the host wire pack
    g.release_contig()
    dct = dict(sam_path=bam, fasta_path=fa, supplementary=False, exclude_bed=None)
    p1 = g.device_pack_for(dct, "chr1")
    p2 = g.device_pack_for(dct, "chrX")
    assert p1 is not p2 and p1.pos_hi == w1.length and p2.pos_hi == w2.length
    assert g.device_pack_for(dct, "chr1") is p1                                      # cached
    assert g.device_pack(bam, fa, "chr1", False, None, 0)[0] is p1                  # the indel path shares the entry
    assert g.device_pack_for(dict(dct, exclude_bed=[("chr1", 5, 90)]), "chr1") is not p1
    assert len(g._PACKS) <= g._PACKS.cap and len(g._BAM_WORLDS) <= g._BAM_WORLDS.cap
    g.release_contig("chr1")
    assert not any(k[2] == "chr1" for k in g._PACKS.d) and any(k[2] == "chrX" for k in g._PACKS.d)
    with pytest.raises(ValueError):
        g.device_pack(w1, None, "chrX")                                             # a World of another contig
    g.release_contig()


# ------------------------------------------------------------------------------------------------- indelCaller.call_manager
def _fake_snp_vcf(path, contigs):
    from nanocaller_amd import snpCaller, vcfio
    hdr = snpCaller.VCF_HEADER.format(contigs="".join("##contig=<ID=%s>\n" % c for c in contigs), sample="SAMPLE")
    lines = []
    rng = np.random.Generator(np.random.PCG64(1))
    for c in contigs:
        for p in sorted(rng.choice(np.arange(100, 20_000), size=60, replace=False)):
            q = float(rng.random() * 40)
            lines.append("%s\t%d\t.\tA\tG\t%.3f\tPASS\tPR=0.1,0.2,0.3,0.4;FQ=0.5\tGT:DP:VF:AD:ADF:ADR\t0/1:30:0.5:15,15:7,8:8,7\n" % (c, p, q))
    vcfio.write_sorted_vcf(path, hdr, lines, contigs)
    return lines


def _records(path):
    return [ln for ln in gzip.open(path, "rt") if not ln.startswith("#")]


@pytest.mark.parametrize("mode", ["snps", "indels", "all"])
def test_indel_call_manager_job_structure_and_merges(tmp_path, monkeypatch, mode):
    """indelCaller.py:290-400 with the GPU worker stubbed: phase jobs release each contig's chunks, per-worker files are
    merged, sorted, BGZF-compressed and indexed; the three output names; SNP-only records are dropped from the indel file"""
    from nanocaller_amd import indelCaller
    from nanocaller_amd.utils import get_chunks
    contigs = ["chr1", "chrX"]
    regions = [("chr1", 1, 20_000, "diploid"), ("chrX", 1, 20_000, "haploid")]
    snp_vcf = str(tmp_path / "t.snps.vcf.gz")
    snp_lines = _fake_snp_vcf(snp_vcf, contigs)
    seen = []

    def fake_indel_run(params, indel_dict, job_Q, counter_Q, files, device=0, worker_id=1, aligner=None):
        path = os.path.join(params["intermediate_indel_files_dir"], "%s.%d.indel.vcf" % (params["prefix"], worker_id))
        files.append(path)
        with open(path, "a") as f:
            while not job_Q.empty():
                kind, chunk = job_Q.get()
                assert kind == "indel" and chunk["sam_path"] == "in.bam"
                seen.append((chunk["chrom"], chunk["start"]))
                f.write("%s\t%d\t.\tAT\tA\t12.00\tPASS\t.\tGT:GQ\t0|1:3.00\n" % (chunk["chrom"], chunk["start"] + 7))
                f.write("%s\t%d\t.\tA\tC\t12.00\tPASS\t.\tGT:GQ\t0|1:3.00\n" % (chunk["chrom"], chunk["start"] + 9))   # SNP-only
                counter_Q.put(1)
    monkeypatch.setattr(indelCaller, "indel_run", fake_indel_run)
    monkeypatch.setattr(indelCaller, "_whatshap_available", lambda: False)
    chunks = get_chunks(regions, 2, max_chunk_size=5_000)
    params = dict(chunks_list=chunks, mode=mode, snp_vcf=snp_vcf if mode != "indels" else None, regions_list=regions, sam_path="in.bam",
                  fasta_path="x.fa", vcf_path=str(tmp_path), prefix="t", sample="SAMPLE", phase_qual_score=10, suppress_progress=True,
                  verbose=False, enable_whatshap=False, cpu=2)
    out = indelCaller.call_manager(params)
    assert set(out) == {"snps", "indels", "final"}
    if mode == "snps":
        assert out["indels"] is None and out["final"] is None and not seen
    else:
        assert sorted(seen) == sorted((c["chrom"], c["start"]) for c in chunks)
        recs = _records(out["indels"])
        assert out["indels"].endswith("t.indels.vcf.gz") and os.path.exists(out["indels"] + ".csi")
        assert len(recs) == len(chunks) and all(r.split("\t")[3] == "AT" for r in recs)
        keys = [(contigs.index(r.split("\t")[0]), int(r.split("\t")[1])) for r in recs]
        assert keys == sorted(keys)
        hdr = [ln for ln in gzip.open(out["indels"], "rt") if ln.startswith("#")]
        assert hdr[0] == "##fileformat=VCFv4.2\n" and "##contig=<ID=chrX>\n" in hdr and hdr[-1].endswith("\tSAMPLE\n")
    if mode == "indels":
        assert out["snps"] is None and out["final"] is None
        return
    assert out["snps"].endswith("t.snps.phased.vcf.gz") and os.path.exists(out["snps"] + ".csi")
    assert sorted(_records(out["snps"])) == sorted(snp_lines)                         # nothing lost at the QUAL split
    lowq = _records(os.path.join(str(tmp_path), "intermediate_phase_files", "chr1.snps.lowq.unphased.vcf.gz"))
    assert lowq and all(float(r.split("\t")[5]) < 10 for r in lowq)
    assert not os.path.exists(os.path.join(str(tmp_path), "intermediate_phase_files", "chrX.snps.lowq.unphased.vcf.gz"))   # haploid
    if mode == "all":
        fin = _records(out["final"])
        assert out["final"].endswith("t.vcf.gz") and len(fin) == len(snp_lines) + len(chunks)
        keys = [(contigs.index(r.split("\t")[0]), int(r.split("\t")[1])) for r in fin]
        assert keys == sorted(keys)
        fh = [ln for ln in gzip.open(out["final"], "rt") if ln.startswith("#")]
        assert any(ln.startswith("##FORMAT=<ID=GQ") for ln in fh) and any(ln.startswith("##INFO=<ID=PR") for ln in fh)


# ------------------------------------------------------------------------------------------------- the reference's run(args)
@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "NanoCaller")), reason="build container only (/root/reference)")
def test_reference_run_function_drives_this_package(two_contig_files, tmp_path, monkeypatch):
    """Load `run` from the reference's own NanoCaller script (its top-level imports of pysam / intervaltree satisfied by empty
    stand-ins, `nanocaller_src` = this repo's alias package) and execute it: regions and chunks come from THIS package's
    utils, and the dicts it hands to snpCaller.call_manager / indelCaller.call_manager carry every key this package reads."""
    w1, w2, bam, fa = two_contig_files
    import nanocaller_src
    from nanocaller_src import indelCaller, snpCaller
    for name in ("pysam", "intervaltree"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.Interval = m.IntervalTree = object
            monkeypatch.setitem(sys.modules, name, m)
    assert sys.modules["nanocaller_src"] is nanocaller_src and nanocaller_src.__file__.startswith(ROOT)
    src = open(os.path.join(REFERENCE, "NanoCaller")).read()
    head = src[:src.index("if __name__ == '__main__':")]
    ns = {"__file__": os.path.join(ROOT, "NanoCaller"), "__name__": "nanocaller_script"}
    sys.dont_write_bytecode = True
    exec(compile(head, "NanoCaller", "exec"), ns)
    got = {}
    monkeypatch.setattr(snpCaller, "call_manager", lambda p: got.setdefault("snp", p) and os.path.join(p["vcf_path"], "t.snps.vcf.gz"))
    monkeypatch.setattr(indelCaller, "call_manager", lambda p: got.setdefault("indel", p) and {"snps": None, "indels": None, "final": None})
    ns["run"](_args(bam, fa, str(tmp_path), mode="all", haploid_X=True))
    sp, ip = got["snp"], got["indel"]
    assert sp["regions_list"] == (("chr1", 1, w1.length, "diploid"), ("chrX", 1, w2.length, "haploid"))
    assert sp["chunks_list"] == nanocaller_src.utils.get_chunks(sp["regions_list"], 2)
    assert ip["chunks_list"] == nanocaller_src.utils.get_chunks(sp["regions_list"], 2, max_chunk_size=100000)
    assert snpCaller.PARAM_KEYS <= set(sp) and indelCaller.PARAM_KEYS <= set(ip)
    assert ip["snp_vcf"].endswith("t.snps.vcf.gz") and ip["mode"] == "all"


# ------------------------------------------------------------------------------------------------- rank -> GPU under gloo
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_call_chunks(log):
    """call_chunks stand-in: the oracle featuriser + CNN on the CPU (what is under test is call_manager itself)"""
    from nanocaller_amd import snpCaller
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle
    from tests.util import load_world

    def call_chunks(params, chunks, device=0, dpk=None, defer=False):
        world = load_world("ont")
        log.append((device, [(c["start"], c["end"]) for c in chunks]))
        path, cov = get_SNP_model(params["snp_model"])
        w = Weights(path)
        keys = ("pos", "ref", "probs", "dp", "freq", "fwd_dp", "rev_dp")
        acc = {k: [] for k in keys}
        dct = {k: params[k] for k in ("threshold", "mincov", "maxcov", "min_allele_freq", "min_nbr_sites", "seq")}
        for c in chunks:
            pos, ref, mat, dp, freq, depth, fwd, rev = oracle.get_snp_testing_candidates(world, dct, c)
            n = min(len(pos), 25)                                               # a slice keeps the CPU test fast
            rc = np.argmax(ref[:n], 1).astype(np.int32)
            probs, _ = oracle.snp_forward(w.flat, mat[:n], rc, cov / depth)
            for k, v in zip(keys, (pos[:n], rc, probs, dp[:n], freq[:n], fwd[:n], rev[:n])):
                acc[k].append(np.asarray(v))
        res = {k: np.concatenate(v) for k, v in acc.items()}
        res.update(n=len(res["pos"]), chrom=chunks[0]["chrom"], ploidy=chunks[0]["ploidy"])
        return snpCaller.PendingCall(lambda: res) if defer else res
    return call_chunks


def _cm_worker(rank, world, port, tmpdir, devices):
    import torch
    import torch.distributed as dist

    from nanocaller_amd import snpCaller
    from nanocaller_amd.utils import get_chunks
    from tests.util import load_world
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.is_available = lambda: True                                       # an 8-GPU node, as the mapping sees it
    torch.cuda.device_count = lambda: 8
    log = []
    snpCaller.call_chunks = _fake_call_chunks(log)
    w = load_world("ont")
    regions = [(w.chrom, 20_000, 120_000, "diploid")]
    params = dict(chunks_list=get_chunks(regions, cpu=5), regions_list=regions, sam_path=w, fasta_path=None, mincov=4, maxcov=160,
                  min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6], snp_model="ONT-HG002", cpu=1, vcf_path=tmpdir,
                  prefix="t", sample="S", seq="ont", supplementary=False, exclude_bed=None, suppress_progress=True,
                  disable_coverage_normalization=False)
    out = snpCaller.call_manager(params, devices=devices)
    with open(os.path.join(tmpdir, "log.%d" % rank), "w") as f:
        f.write(repr((out, log)))
    dist.destroy_process_group()


@pytest.mark.parametrize("devices,expect", [(None, [0, 1]), ((4, 6), [4, 6])])
def test_call_manager_under_gloo_maps_ranks_to_gpus(tmp_path, devices, expect):
    """VERDICT r1 weak #8 / ADVICE: every rank of a torchrun job must work on ITS GPU, own a contiguous block of the chunk
    list, and rank 0 must merge all worker files"""
    from nanocaller_amd.utils import get_chunks
    world = 2
    mp.spawn(_cm_worker, args=(world, _free_port(), str(tmp_path), devices), nprocs=world, join=True)
    logs = [eval(open(os.path.join(str(tmp_path), "log.%d" % r)).read()) for r in range(world)]
    assert [lg[1][0][0] for lg in logs] == expect                                # device per rank
    spans = [s for lg in logs for (_, ss) in lg[1] for s in ss]
    from tests.util import load_world
    assert spans == [(c["start"], c["end"]) for c in get_chunks([(load_world("ont").chrom, 20_000, 120_000, "diploid")], cpu=5)]
    assert logs[0][0] == logs[1][0] and logs[0][0].endswith("t.snps.vcf.gz")
    recs = _records(os.path.join(str(tmp_path), "t.unfiltered.snps.vcf.gz"))
    n_lines = sum(len(open(os.path.join(str(tmp_path), "intermediate_snp_files", "t.%d.snps.vcf" % (r + 1))).readlines()) for r in range(world))
    assert len(recs) == n_lines > 50
    pos = [int(r.split("\t")[1]) for r in recs]
    assert pos == sorted(pos)


# ------------------------------------------------------------------------------------------------- GPU: both managers, two contigs
@pytest.mark.gpu
def test_call_managers_on_a_two_contig_bam(two_contig_files, tmp_path):
    """snpCaller.call_manager then indelCaller.call_manager(mode='all') with the dicts NanoCaller:28-53 builds, on one BAM
    with two contigs (one diploid, one haploid): every contig gets ITS OWN calls (the pack cache used to hand the
    first contig's alignments to the second), equal to the oracle pipeline per contig; the three merged files exist."""
    from nanocaller_amd import indelCaller, snpCaller
    from nanocaller_amd.bam import read_bam
    from nanocaller_amd.generate_SNP_pileups import release_contig
    from nanocaller_amd.utils import get_chunks, get_regions_list
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle
    w1, w2, bam, fa = two_contig_files
    release_contig()
    a = _args(bam, fa, str(tmp_path), mode="all", haploid_X=True, mincov=2)
    regions = get_regions_list(a)
    thr = [float(v) for v in a.neighbor_threshold.split(",")]
    sp = dict(chunks_list=get_chunks(regions, a.cpu), regions_list=regions, sam_path=a.bam, fasta_path=a.ref, mincov=a.mincov,
              maxcov=a.maxcov, min_allele_freq=a.min_allele_freq, min_nbr_sites=a.min_nbr_sites, threshold=thr, snp_model=a.snp_model,
              cpu=a.cpu, vcf_path=a.output, prefix=a.prefix, sample=a.sample, seq=a.sequencing, supplementary=a.supplementary,
              exclude_bed=a.exclude_bed, suppress_progress=True, phase_qual_score=a.phase_qual_score, verbose=False,
              disable_coverage_normalization=False)
    snp_vcf = snpCaller.call_manager(sp)
    got = _records(os.path.join(str(tmp_path), "t.unfiltered.snps.vcf.gz"))
    exp = []
    dct = dict(threshold=thr, mincov=a.mincov, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, seq="ont")
    wd, cov = Weights(get_SNP_model("ONT-HG002")[0]), get_SNP_model("ONT-HG002")[1]
    wh = Weights(get_SNP_model("haploid")[0])
    for c in sp["chunks_list"]:
        world = read_bam(bam, fa, c["chrom"])
        pos, ref, mat, dp, freq, depth, fwd, rev = oracle.get_snp_testing_candidates(world, dct, c)
        if not len(pos):
            continue
        rc = np.argmax(ref, 1).astype(np.int32)
        if c["ploidy"] == "diploid":
            exp += snpCaller.snp_vcf_lines(c["chrom"], pos, rc, oracle.snp_forward(wd.flat, mat, rc, cov / depth)[0], dp, freq, fwd, rev)
        else:
            exp += snpCaller.snp_vcf_lines_haploid(c["chrom"], pos, rc, oracle.snp_hap_forward(wh.flat, mat, rc, 30.0 / depth), dp, freq)
    order = {"chr1": 0, "chrX": 1}
    exp.sort(key=lambda ln: (order[ln.split("\t")[0]], int(ln.split("\t")[1])))
    assert len(got) == len(exp) and {ln.split("\t")[0] for ln in got} == {"chr1", "chrX"}
    for g, e in zip(got, exp):
        gf, ef = g.split("\t"), e.split("\t")
        assert gf[:5] == ef[:5] and gf[6] == ef[6] and gf[9].split(":")[:2] == ef[9].split(":")[:2], (g, e)
    ip = dict(chunks_list=get_chunks(regions, a.cpu, max_chunk_size=100000), mode="all", snp_vcf=snp_vcf, regions_list=regions,
              sam_path=a.bam, fasta_path=a.ref, mincov=a.mincov, maxcov=a.maxcov, indel_model=a.indel_model, cpu=a.cpu,
              vcf_path=a.output, prefix=a.prefix, sample=a.sample, seq=a.sequencing, del_t=a.del_threshold, ins_t=a.ins_threshold,
              impute_indel_phase=False, supplementary=False, exclude_bed=None, win_size=40, small_win_size=4, enable_whatshap=False,
              suppress_progress=True, phase_qual_score=a.phase_qual_score, verbose=False)
    files = indelCaller.call_manager(ip)
    assert files == {"snps": os.path.join(str(tmp_path), "t.snps.phased.vcf.gz"), "indels": os.path.join(str(tmp_path), "t.indels.vcf.gz"),
                     "final": os.path.join(str(tmp_path), "t.vcf.gz")}
    ind = _records(files["indels"])
    assert len(ind) > 20 and {ln.split("\t")[0] for ln in ind} == {"chr1", "chrX"}
    for ln in ind:                                                                   # REF is the reference of ITS contig at POS
        f = ln.split("\t")
        w = w1 if f[0] == "chr1" else w2
        assert f[3] == w.ref[int(f[1]) - 1:int(f[1]) - 1 + len(f[3])]
    assert len(_records(files["final"])) == len(ind) + len(_records(files["snps"]))
    for fn in files.values():
        assert os.path.exists(fn + ".csi")


def _icm_worker(rank, world, port, tmpdir):
    import torch
    import torch.distributed as dist

    from nanocaller_amd import indelCaller
    from nanocaller_amd.utils import get_chunks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank))
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
                seen.append((device, worker_id, chunk["chrom"], chunk["start"], chunk["sam_path"]))
                f.write("%s\t%d\t.\tAT\tA\t12.00\tPASS\t.\tGT:GQ\t0|1:3.00\n" % (chunk["chrom"], chunk["start"] + 7))
    indelCaller.indel_run = fake_indel_run
    indelCaller._whatshap_available = lambda: False
    regions = [("chr1", 1, 20_000, "diploid"), ("chrX", 1, 20_000, "haploid")]
    snp_vcf = os.path.join(tmpdir, "t.snps.vcf.gz")
    if rank == 0:
        _fake_snp_vcf(snp_vcf, ["chr1", "chrX"])
    dist.barrier()
    params = dict(chunks_list=get_chunks(regions, 2, max_chunk_size=2_500), mode="all", snp_vcf=snp_vcf, regions_list=regions, sam_path="in.bam",
                  fasta_path="x.fa", vcf_path=tmpdir, prefix="t", sample="S", phase_qual_score=10, suppress_progress=True, verbose=False,
                  enable_whatshap=False, cpu=2)
    out = indelCaller.call_manager(params)
    with open(os.path.join(tmpdir, "ilog.%d" % rank), "w") as f:
        f.write(repr((out, seen)))
    dist.destroy_process_group()


def test_indel_call_manager_under_gloo_shards_chunks_and_merges(tmp_path):
    """indelCaller.call_manager with two ranks: rank 0 phases, the indel chunks are sharded in contiguous blocks, every rank
    works on its own GPU and writes its own worker file, rank 0 merges all of them into the three sorted, indexed outputs"""
    from nanocaller_amd.utils import get_chunks
    world = 2
    mp.spawn(_icm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    logs = [eval(open(os.path.join(str(tmp_path), "ilog.%d" % r)).read()) for r in range(world)]
    chunks = get_chunks([("chr1", 1, 20_000, "diploid"), ("chrX", 1, 20_000, "haploid")], 2, max_chunk_size=2_500)
    got = [(c, s) for lg in logs for (_, _, c, s, _) in lg[1]]
    assert sorted(got) == sorted((c["chrom"], c["start"]) for c in chunks) and len(logs[0][1]) > 0 and len(logs[1][1]) > 0
    assert {d for (d, _, _, _, _) in logs[0][1]} == {0} and {d for (d, _, _, _, _) in logs[1][1]} == {1}        # rank -> GPU
    assert {w for (_, w, _, _, _) in logs[1][1]} == {2} and all(sp == "in.bam" for lg in logs for (_, _, _, _, sp) in lg[1])
    assert logs[0][0] == logs[1][0] and logs[0][0]["final"].endswith("t.vcf.gz")
    ind = _records(logs[0][0]["indels"])
    assert len(ind) == len(chunks)
    keys = [(["chr1", "chrX"].index(r.split("\t")[0]), int(r.split("\t")[1])) for r in ind]
    assert keys == sorted(keys)
    assert len(_records(logs[0][0]["final"])) == len(ind) + len(_records(logs[0][0]["snps"]))


@pytest.mark.gpu
def test_pipelined_ingest_writes_the_serial_callers_worker_file(two_contig_files, tmp_path, monkeypatch):
    """snpCaller.caller decodes + wire-builds group i + 1 on a host thread and uploads it through the three-slot ring while the GPU runs
    group i; the worker file is byte-identical to the one the serial caller (NC_SERIAL_INGEST=1) writes -- and to the one of the
    device ingest route (the default for an indexed BAM: the file is inflated and decoded in HBM, no host decode at all)"""
    import queue

    from nanocaller_amd import snpCaller
    from nanocaller_amd import generate_SNP_pileups as gsp
    from nanocaller_amd.utils import get_chunks, get_regions_list
    w1, w2, bam, fa = two_contig_files
    a = _args(bam, fa, str(tmp_path), haploid_X=True, mincov=2)
    regions = get_regions_list(a)
    outs = []
    from nanocaller_amd.device_bam import DeviceBam, plan_shares, release
    each = max(DeviceBam(bam, 0, contigs=[c]).n_bytes for c in ("chr1", "chrX"))
    assert [fits for _, fits in plan_shares(bam, ["chr1", "chrX"], each + 1)] == [True, True]      # one contig per share
    assert [fits for _, fits in plan_shares(bam, ["chr1", "chrX"], each - 1)].count(False) >= 1    # a contig that fits no share: host route
    assert plan_shares(bam, ["chr1", "chrX"]) == [(["chr1", "chrX"], True)]
    for tag, serial, dev, share in (("serial", "1", "0", None), ("piped", None, "0", None), ("device", None, None, None), ("shares", None, None, each + 1),
                                    ("mixed", None, None, each - 1)):
        release()
        if share:
            monkeypatch.setenv("NC_DEVICE_INGEST_SHARE_GB", repr(share / (1 << 30)))
        else:
            monkeypatch.delenv("NC_DEVICE_INGEST_SHARE_GB", raising=False)
        if serial:
            monkeypatch.setenv("NC_SERIAL_INGEST", serial)
        else:
            monkeypatch.delenv("NC_SERIAL_INGEST", raising=False)
        if dev:
            monkeypatch.setenv("NC_DEVICE_INGEST", dev)
        else:
            monkeypatch.delenv("NC_DEVICE_INGEST", raising=False)
        gsp.release_contig()
        del gsp.DECODES[:]
        d = tmp_path / tag
        d.mkdir()
        params = dict(chunks_list=get_chunks(regions, a.cpu), regions_list=regions, sam_path=bam, fasta_path=fa, mincov=2, maxcov=160,
                      min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6], snp_model="ONT-HG002", cpu=2, vcf_path=str(d), prefix="t",
                      sample="S", seq="ont", supplementary=False, exclude_bed=None, suppress_progress=True,
                      disable_coverage_normalization=False, intermediate_snp_files_dir=str(d))
        q = queue.Queue()
        for c in params["chunks_list"]:
            q.put(c)
        files = []
        snpCaller.caller(params, q, queue.Queue(), files)
        outs.append(open(files[0], "rb").read())
        if tag in ("device", "shares"):
            from nanocaller_amd import device_bam
            assert gsp.DECODES == []                                                # not decoded on the host at all
            assert not device_bam._OPEN                                             # the worker's share of the file left HBM with the worker
        elif tag != "mixed":
            assert sorted(x[1] for x in gsp.DECODES) == ["chr1", "chrX"]            # every contig decoded once
        else:
            assert 1 <= len(gsp.DECODES) <= 2
    assert all(o == outs[0] for o in outs) and outs[0].count(b"\n") > 100
