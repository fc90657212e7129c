"""CPU-only: the native BGZF/BAM(+BAI)/FASTA ingest round-trips synthetic worlds written to real files."""
import gzip
import os

import numpy as np
import pytest

from nanocaller_amd.bam import BamFile, read_bam, read_fasta
from tests import bamio


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("bam")
    w = bamio.make_bam_world()
    rng = np.random.Generator(np.random.PCG64(1))
    recs = bamio.world_to_records(w, rng)
    # an unmapped read and a read on another contig must be skipped
    recs_all = recs + [dict(name="other", flag=0, pos0=5, cigar=[("M", 20)], seq="A" * 20, tags={})]
    bam = str(d / "w.bam")
    bamio.write_bam(bam, w.chrom, w.length, recs, other_refs=[("chrOther", 1000)])
    fa = str(d / "ref.fa")
    bamio.write_fasta(fa, w.chrom, w.ref, extra=[("chrOther", "ACGT" * 250)])
    return w, bam, fa, recs_all


def test_written_bam_is_valid_bgzf(files):
    _, bam, _, _ = files
    raw = gzip.open(bam, "rb").read()
    assert raw[:4] == b"BAM\x01"


def _check(world, got, sel):
    assert np.array_equal(got.read_start, world.read_start[sel]) and np.array_equal(got.read_end, world.read_end[sel])
    assert np.array_equal(got.read_flag, world.read_flag[sel])
    assert got.names == [world.names[i] for i in sel]
    ev_off, ev_pos, ev_len = world.meta["events"]
    g_off, g_pos, g_len = got.meta["events"]
    for k, i in enumerate(sel):
        assert np.array_equal(got.read_codes(k), world.read_codes(i)), i
        assert np.array_equal(g_pos[g_off[k]:g_off[k + 1]], ev_pos[ev_off[i]:ev_off[i + 1]]), i
        assert np.array_equal(g_len[g_off[k]:g_off[k + 1]], ev_len[ev_off[i]:ev_off[i + 1]]), i
    assert np.array_equal(got.meta["hap"], world.meta["hap"][sel])
    assert np.array_equal(got.meta["ps"], np.where(world.meta["hap"][sel] > 0, world.meta["ps"][sel], 0))


def test_whole_contig_roundtrip(files):
    world, bam, fa, _ = files
    bf = BamFile(bam)
    assert bf.references == [world.chrom, "chrOther"] and bf.get_reference_length(world.chrom) == world.length and bf.has_index
    got = read_bam(bam, fa, world.chrom, keep_seq=True)
    mapped = np.nonzero((world.read_flag & 4) == 0)[0]
    assert len(mapped) < world.n_reads                      # the world contains unmapped-flag reads
    _check(world, got, mapped)
    assert got.ref == world.ref and read_fasta(fa, "chrOther") == "ACGT" * 250
    assert got.meta["seq_off"][-1] == got.meta["seq"].size > got.codes.size * 0.9
    assert sum(int((got.meta["events"][2] > 0).sum()) for _ in [0]) > 20 and int((got.meta["events"][2] < 0).sum()) > 20


def test_table_form_of_the_sequence_decode_equals_the_shuffle_form(files):
    """nc_bam_decode turns a record's 4-bit SEQ into codes with two byte shuffles per 32 bases where the CPU has SSSE3 and through a
    256-entry pair table otherwise; the library picks once at load time, so the other form runs in a second interpreter"""
    import hashlib
    import subprocess
    import sys
    world, bam, fa, _ = files
    prog = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from nanocaller_amd.bam import read_bam\n"
            "g = read_bam(%r, %r, %r, keep_seq=True)\n"
            "print(hashlib.md5(g.codes.tobytes() + g.meta['seq'].tobytes()).hexdigest())\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), bam, fa, world.chrom)
    out = subprocess.run([sys.executable, "-c", prog], env=dict(os.environ, NC_BAM_PLAIN_SEQ="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = read_bam(bam, fa, world.chrom, keep_seq=True)
    assert out.stdout.strip().splitlines()[-1] == hashlib.md5(got.codes.tobytes() + got.meta["seq"].tobytes()).hexdigest()


@pytest.mark.parametrize("use_index", [True, False])
def test_region_queries(files, tmp_path, use_index):
    world, bam, fa, _ = files
    path = bam
    if not use_index:
        path = str(tmp_path / "noidx.bam")
        open(path, "wb").write(open(bam, "rb").read())
    bf = BamFile(path)
    assert bf.has_index == use_index
    for (a, b) in [(1, 500), (9_000, 9_001), (16_300, 16_500), (20_000, 30_000), (29_990, 40_000)]:
        d = bf.decode(world.chrom, a, b)
        exp = np.nonzero(((world.read_flag & 4) == 0) & (world.read_start <= min(b, world.length)) & (world.read_end > a))[0]
        assert np.array_equal(d["read_start"], world.read_start[exp]), (a, b)
        assert d["names"] == [world.names[i] for i in exp]
    # no .fai: the plain scanner gives the same sequence
    fa2 = str(tmp_path / "r2.fa")
    bamio.write_fasta(fa2, world.chrom, world.ref, width=71, with_fai=False)
    assert read_fasta(fa2, world.chrom) == world.ref


def test_cigar_edge_cases(tmp_path):
    recs = [
        dict(name="a", flag=16, pos0=99, cigar=[("H", 5), ("S", 2), ("M", 4), ("I", 3), ("M", 2), ("D", 2), ("M", 3), ("N", 4), ("=", 2), ("X", 1), ("S", 1)],
             seq="TTACGTAAAGGCATACGA", tags={"HP": 2, "PS": 70000, "XX": "str"}),
        dict(name="b", flag=0x800, pos0=100, cigar=[("I", 2), ("M", 5)], seq="GGACGTN", tags={}),
    ]
    bam = str(tmp_path / "e.bam")
    bamio.write_bam(bam, "c", 1000, recs)
    bf = BamFile(bam)
    d = bf.decode("c", 1, 1000, keep_seq=True)
    assert d["read_start"].tolist() == [100, 101] and d["read_end"].tolist() == [100 + 4 + 2 + 2 + 3 + 4 + 3, 106]
    c0 = d["codes"][d["read_off"][0]:d["read_off"][1]].tolist()
    #      A C G T | G G | D D | C A T | N N N N | A C | G      (codes A0 G1 T2 C3, del/skip 4)
    assert c0 == [0, 3, 1, 2, 1, 1, 4, 4, 3, 0, 2, 4, 4, 4, 4, 0, 3, 1]
    assert d["ev_pos"][d["ev_off"][0]:d["ev_off"][1]].tolist() == [103, 105] and d["ev_len"][:2].tolist() == [3, -2]
    assert d["hap"].tolist() == [2, 0] and d["ps"].tolist() == [70000, 0]
    assert d["read_flag"].tolist() == [16 | 0x10000, 0x800]                    # bit 16 (NC_FLAG_REFSKIP): the CIGAR holds a reference skip
    assert d["ev_off"].tolist() == [0, 2, 2]                # a leading insertion has no previous column: no marker
    assert d["codes"][d["read_off"][1]:].tolist() == [0, 3, 1, 2, 4] and d["names"] == ["a", "b"]


def test_parallel_region_decode_equals_sequential(files):
    """decode_parallel (one handle + BAI seek per host thread, reads assigned to the region they start in) returns exactly
    the sequential decode, including offsets, events, tags, names and query sequences"""
    from nanocaller_amd.bam import decode_parallel
    _, bam, _, _ = files
    bf = BamFile(bam)
    chrom = bf.references[0]
    L = bf.lengths[0]
    seq = bf.decode(chrom, 1, L, keep_seq=True)
    bf.close()
    for threads, min_region in ((4, 3_000), (7, 1_000), (3, 10_000_000)):
        par = decode_parallel(bam, chrom, 1, L, keep_seq=True, threads=threads, min_region=min_region)
        assert par["names"] == seq["names"]
        keys = ("read_start", "read_end", "read_flag", "read_off", "codes", "ev_off", "ev_pos", "ev_len", "hap", "ps", "seq_off", "seq",
                "qstart")
        for k in keys:
            assert np.array_equal(par[k], seq[k]) and par[k].dtype == seq[k].dtype, k
        # sub-intervals, more regions than reads start in, and the Python-threads statement of the same merge
        from nanocaller_amd.bam import _decode_parallel_py
        for (a, b) in ((1, L), (L // 3, 2 * L // 3), (L - 50, L)):
            sub = BamFile(bam)
            want = sub.decode(chrom, a, b, keep_seq=True)
            sub.close()
            got = decode_parallel(bam, chrom, a, b, keep_seq=True, threads=threads, min_region=min(min_region, 500))
            py = _decode_parallel_py(bam, chrom, a, b, True, threads, min(threads, 5))
            assert got["names"] == want["names"] == py["names"]
            for k in keys:
                assert np.array_equal(got[k], want[k]) and np.array_equal(py[k], want[k]), (k, a, b)
    # the views stay valid after every other reference to the native result is gone
    import gc
    codes = decode_parallel(bam, chrom, 1, L, threads=4, min_region=2_000)["codes"]
    ref_sum = int(seq["codes"].astype(np.int64).sum())
    del par, got
    gc.collect()
    assert int(codes.astype(np.int64).sum()) == ref_sum and codes.flags.writeable


def test_records_without_bases_and_long_cigars_in_the_cg_tag(tmp_path):
    """(a) SEQ '*' (l_seq 0, minimap2's secondary alignments) with a long CIGAR must not be read past the record: aligned
    positions decode as 'N' (code 4) and the pileup flag filter drops the read later; (b) SAMv1 4.2.2: a CIGAR of > 65535
    operations lives in the CG:B,I tag behind the placeholder <l_seq>S<ref_len>N -- the real one must be decoded"""
    ops = "MIDNSHP=X"
    ref = "ACGT" * 500
    seq = "ACGTACGTAC" + "GG" + "TACGTACG"                    # 10M 2I 3D 8M over ref positions 101..121
    real = [("M", 10), ("I", 2), ("D", 3), ("M", 8)]
    cg = [(ln << 4) | ops.index(op) for op, ln in real]
    recs = [dict(name="noseq", flag=0x100, pos0=40, cigar=[("M", 300), ("D", 4), ("M", 200)], seq="", tags={}),
            dict(name="short", flag=0, pos0=60, cigar=[("M", 30)], seq="ACGTA", tags={}),
            dict(name="longcig", flag=0, pos0=100, cigar=[("S", len(seq)), ("N", 21)], seq=seq, tags={"HP": 2, "CG": cg, "PS": 77}),
            dict(name="plain", flag=16, pos0=100, cigar=real, seq=seq, tags={})]
    bam, fa = str(tmp_path / "c.bam"), str(tmp_path / "c.fa")
    bamio.write_bam(bam, "chrT", len(ref), recs)
    bamio.write_fasta(fa, "chrT", ref)
    w = read_bam(bam, fa, "chrT", keep_seq=True)
    assert w.names == ["noseq", "short", "longcig", "plain"]
    assert (w.read_start[0], w.read_end[0]) == (41, 41 + 504) and set(w.read_codes(0).tolist()) == {4}
    assert set(w.read_codes(1).tolist()) == {4}
    # the CG record decodes exactly like the same alignment with an inline CIGAR
    assert (w.read_start[2], w.read_end[2]) == (w.read_start[3], w.read_end[3]) == (101, 122)
    assert np.array_equal(w.read_codes(2), w.read_codes(3))
    ev_off, ev_pos, ev_len = w.meta["events"]
    assert ev_pos[ev_off[2]:ev_off[3]].tolist() == ev_pos[ev_off[3]:ev_off[4]].tolist() == [110, 110]
    assert ev_len[ev_off[2]:ev_off[3]].tolist() == [2, -3]
    assert w.meta["hap"].tolist() == [0, 0, 2, 0] and w.meta["ps"][2] == 77
    # pass-2 windows over the record without bases are empty strings, never out-of-range reads
    bf = BamFile(bam)
    d = bf.decode("chrT", 1, 600, anchors=[105, 300], window_before=0, window_after=160, keep_mask=0x4)
    bf.close()
    win = {d["names"][r]: t for r, t in d["windows"][0]}
    assert win["noseq"] == "" and win["longcig"] == win["plain"] and len(win["plain"]) > 5


def test_csi_index_gives_the_same_regions_as_bai(tmp_path):
    """`samtools index -c` writes a CSI index (the only kind for contigs longer than 2^29): region decodes through it equal
    those through the BAI linear index and those of a sequential scan"""
    import shutil
    w = bamio.make_bam_world(seed=9, length=90_000, depth=10)
    recs = bamio.world_to_records(w, np.random.Generator(np.random.PCG64(3)))
    bai, csi, plain = str(tmp_path / "a.bam"), str(tmp_path / "c.bam"), str(tmp_path / "p.bam")
    bamio.write_bam(bai, w.chrom, w.length, recs, write_bai=True)
    bamio.write_bam(csi, w.chrom, w.length, recs, write_bai=False, write_csi=True)
    bamio.write_bam(plain, w.chrom, w.length, recs, write_bai=False)
    fa = BamFile(bai), BamFile(csi), BamFile(plain)
    assert fa[0].has_index and fa[1].has_index and not fa[2].has_index
    for (a, b) in [(1, 90_000), (40_000, 41_000), (16_384, 16_385), (70_001, 90_000), (89_990, 90_000), (32_768, 49_152), (5, 5)]:
        ds = [f.decode(w.chrom, a, b, keep_seq=True) for f in fa]
        for d in ds[1:]:
            assert d["names"] == ds[0]["names"], (a, b)
            for k in ("read_start", "read_end", "read_flag", "codes", "ev_pos", "ev_len", "hap", "ps", "seq"):
                assert np.array_equal(d[k], ds[0][k]), (k, a, b)
        assert len(ds[0]["names"]) == sum(1 for r in recs if not r["flag"] & 4 and r["pos0"] < b and r["pos0"] + sum(n for o, n in r["cigar"] if o in "MDN=X") >= a)
    for f in fa:
        f.close()
    # a whole-contig parallel decode through the CSI
    from nanocaller_amd.bam import decode_parallel
    d1 = decode_parallel(csi, w.chrom, threads=4, min_region=10_000)
    d2 = decode_parallel(bai, w.chrom, threads=4, min_region=10_000)
    assert d1["names"] == d2["names"] and np.array_equal(d1["codes"], d2["codes"])
