"""GPU: the device ingest (nanocaller_amd/device_bam.py: inflate + record walk + decode in HBM) builds the SAME read pack as the host route
(nc_bam_decode -> wire -> nc_wire_expand), byte for byte, on every kind of test BAM; CPU: the BGZF member scan."""
import os
import zlib

import numpy as np
import pytest

from nanocaller_amd import _lib
from tests import bamio

G = os.path.join(os.path.dirname(__file__), "golden")


def test_member_scan_of_the_spec_fixture():
    raw = open(os.path.join(G, "spec.bam"), "rb").read()
    L = _lib.lib()
    import ctypes as C
    data = np.frombuffer(raw, np.uint8)
    cap = 4096
    coff, clen, isize = np.empty(cap, np.int64), np.empty(cap, np.int32), np.empty(cap, np.int32)
    n = C.c_int64()
    assert L.nc_bgzf_members(_lib.npp(data), len(raw), cap, _lib.npp(coff), _lib.npp(clen), _lib.npp(isize), C.byref(n)) == _lib.NC_OK
    k = int(n.value)
    assert k > 30
    o = 0
    for i in range(k):                                                  # against a walk of the same bytes in Python
        bsize = int.from_bytes(raw[o + 16:o + 18], "little") + 1
        assert coff[i] == o + 18 and clen[i] == bsize - 26
        assert len(zlib.decompress(raw[coff[i]:coff[i] + clen[i]], -15)) == isize[i]
        o += bsize
    assert o == len(raw)
    assert L.nc_bgzf_members(_lib.npp(data), len(raw), 3, _lib.npp(coff), _lib.npp(clen), _lib.npp(isize), C.byref(n)) == _lib.NC_ERR_CAPACITY and n.value == k
    bad = data.copy()
    bad[0] = 0
    assert L.nc_bgzf_members(_lib.npp(bad), len(raw), cap, _lib.npp(coff), _lib.npp(clen), _lib.npp(isize), C.byref(n)) == -1
    assert L.nc_bgzf_members(_lib.npp(data), len(raw) - 5, cap, _lib.npp(coff), _lib.npp(clen), _lib.npp(isize), C.byref(n)) == -1


def _same_pack(bam, fa, chrom, supplementary=False, span=None, exclude=None, contigs=None):
    import torch
    from nanocaller_amd.bam import read_bam, read_fasta
    from nanocaller_amd.device_bam import DeviceBam
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.wire import build_wire_from_world, upload_wire
    eng = get_engine(0)
    from nanocaller_amd.generate_SNP_pileups import _check_supported
    world = read_bam(bam, fa, chrom) if span is None else read_bam(bam, fa, chrom, span[0], span[1])
    _check_supported(world, bam, chrom, supplementary)
    kw = dict(pos_lo=span[0], pos_hi=span[1]) if span else {}
    want = upload_wire(eng, build_wire_from_world(world, supplementary=supplementary, exclude=exclude, **kw))
    db = DeviceBam(bam, 0, contigs=contigs).load()
    prep = db.prepare(chrom, read_fasta(fa, chrom), supplementary=supplementary, span=span, exclude=exclude)
    got = db.pack(prep)
    torch.cuda.synchronize()
    for f in ("tile_size", "tile_pos0", "n_tiles", "n_entries", "pos_lo", "pos_hi"):
        assert getattr(got, f) == getattr(want, f), f
    assert np.array_equal(got.tile_off.cpu().numpy(), want.tile_off.cpu().numpy())
    assert np.array_equal(got.tile_ent.cpu().numpy().view(np.uint8).reshape(-1), want.tile_ent.cpu().numpy().view(np.uint8).reshape(-1)[:got.tile_ent.numel()])
    assert np.array_equal(got.ref_code.cpu().numpy(), want.ref_code.cpu().numpy())
    a, b = got.codes.cpu().numpy(), want.codes.cpu().numpy()
    assert a.shape == b.shape
    assert np.array_equal(a, b), np.flatnonzero(a != b)[:10]
    assert prep["n_reads"] == world.n_reads and np.array_equal(prep["read_start"], world.read_start) and np.array_equal(prep["read_flag"], world.read_flag)
    return got, prep, world


@pytest.mark.gpu
def test_synthetic_world_bam_gives_the_host_routes_pack(tmp_path):
    w = bamio.make_bam_world()
    recs = bamio.world_to_records(w, np.random.Generator(np.random.PCG64(1)))
    bam, fa = str(tmp_path / "w.bam"), str(tmp_path / "w.fa")
    bamio.write_bam(bam, w.chrom, w.length, recs, other_refs=[("chrOther", 1000)])
    bamio.write_fasta(fa, w.chrom, w.ref, extra=[("chrOther", "ACGT" * 250)])
    got, prep, world = _same_pack(bam, fa, w.chrom)
    assert prep["n_kept"] > 100
    lo, hi = w.length // 3, 2 * w.length // 3
    _same_pack(bam, fa, w.chrom, span=(lo, hi))
    _same_pack(bam, fa, w.chrom, supplementary=True)
    _same_pack(bam, fa, w.chrom, exclude=((0, 40), (w.length // 2, w.length // 2 + 700), (w.length - 5, w.length + 90)))


@pytest.mark.gpu
def test_cigar_edge_cases_soft_clips_cg_tags_and_records_without_bases(tmp_path):
    ops = "MIDNSHP=X"
    ref = "ACGT" * 500
    seq = "ACGTACGTAC" + "GG" + "TACGTACG"
    real = [("M", 10), ("I", 2), ("D", 3), ("M", 8)]
    cg = [(ln << 4) | ops.index(op) for op, ln in real]
    long_m = "ACGTTGCA" * 40                                              # a run longer than the wave-cooperative threshold
    recs = [dict(name="noseq", flag=0x100, pos0=40, cigar=[("M", 300), ("D", 4), ("M", 200)], seq="", tags={}),
            dict(name="short", flag=0, pos0=60, cigar=[("M", 30)], seq="ACGTA", tags={}),
            dict(name="clip", flag=16, pos0=99, cigar=[("H", 5), ("S", 2), ("M", 4), ("I", 3), ("M", 2), ("D", 2), ("M", 3), ("=", 2), ("X", 1), ("S", 1)],
                 seq="TTACGTAAAGGCATACGA", tags={"HP": 2, "PS": 70000, "XX": "str"}),
            dict(name="longcig", flag=0, pos0=100, cigar=[("S", len(seq)), ("N", 21)], seq=seq, tags={"HP": 2, "CG": cg, "PS": 77}),
            dict(name="plain", flag=16, pos0=100, cigar=real, seq=seq, tags={"HP": 1}),
            dict(name="ins1st", flag=0, pos0=120, cigar=[("I", 2), ("M", 5)], seq="GGACGTN", tags={}),
            dict(name="hifi", flag=0, pos0=300, cigar=[("M", 200), ("D", 70), ("M", 120)], seq=long_m, tags={"HP": 1}),
            dict(name="unmapped", flag=4, pos0=310, cigar=[], seq="ACGT", tags={}),
            dict(name="dup", flag=0x400, pos0=400, cigar=[("M", 20)], seq="A" * 20, tags={})]
    bam, fa = str(tmp_path / "c.bam"), str(tmp_path / "c.fa")
    bamio.write_bam(bam, "chrT", len(ref), recs)
    bamio.write_fasta(fa, "chrT", ref)
    got, prep, world = _same_pack(bam, fa, "chrT")
    assert prep["n_reads"] == 8 and prep["n_kept"] == 6


@pytest.mark.gpu
def test_reference_skips_and_same_name_overlaps_are_refused_like_the_host_route(tmp_path):
    from nanocaller_amd.bam import read_fasta
    from nanocaller_amd.device_bam import DeviceBam
    ref = "ACGT" * 500
    recs = [dict(name="a", flag=0, pos0=10, cigar=[("M", 30), ("N", 40), ("M", 30)], seq="ACGT" * 15, tags={}),
            dict(name="b", flag=0, pos0=50, cigar=[("M", 40)], seq="ACGT" * 10, tags={})]
    bam, fa = str(tmp_path / "n.bam"), str(tmp_path / "n.fa")
    bamio.write_bam(bam, "chrT", len(ref), recs)
    bamio.write_fasta(fa, "chrT", ref)
    db = DeviceBam(bam, 0).load()
    with pytest.raises(_lib.NanoCallerHipError) as e:
        db.prepare("chrT", read_fasta(fa, "chrT"))
    assert e.value.status == _lib.NC_ERR_UNSUPPORTED and "reference skip" in str(e.value)
    recs = [dict(name="x", flag=0, pos0=10, cigar=[("M", 60)], seq="ACGT" * 15, tags={}),
            dict(name="x", flag=0x800, pos0=50, cigar=[("M", 40)], seq="ACGT" * 10, tags={}),
            dict(name="y", flag=0, pos0=55, cigar=[("M", 40)], seq="ACGT" * 10, tags={})]
    bam2 = str(tmp_path / "d.bam")
    bamio.write_bam(bam2, "chrT", len(ref), recs)
    db = DeviceBam(bam2, 0).load()
    assert db.prepare("chrT", read_fasta(fa, "chrT"))["n_kept"] == 2          # default filter: the supplementary record is dropped
    with pytest.raises(_lib.NanoCallerHipError) as e:
        db.prepare("chrT", read_fasta(fa, "chrT"), supplementary=True)
    assert e.value.status == _lib.NC_ERR_UNSUPPORTED and "same read name" in str(e.value)


@pytest.mark.gpu
def test_spec_fixture_every_contig(tmp_path):
    """the BAM assembled from the SAM specification (not written by tests/bamio.py): stored / fixed / dynamic members, records straddling
    members, every aux type"""
    from nanocaller_amd.device_bam import DeviceBam
    bam = os.path.join(G, "spec.bam")
    db = DeviceBam(bam, 0).load()
    assert db.n_rec > 0 and len(db.ref_names) > 0
    rng = np.random.default_rng(5)
    for chrom, length in zip(db.ref_names, db.ref_lengths):
        fa = str(tmp_path / ("%s.fa" % chrom))
        bamio.write_fasta(fa, chrom, "".join("ACGT"[i] for i in rng.integers(0, 4, length)))
        tid = db.ref_names.index(chrom)
        if tid not in db.tid_range:
            continue
        try:
            _same_pack(bam, fa, chrom)
        except _lib.NanoCallerHipError as e:                             # an unsupported input is refused by BOTH routes, with the same words
            from nanocaller_amd.bam import read_fasta
            assert getattr(e, "status", None) == _lib.NC_ERR_UNSUPPORTED
            with pytest.raises(_lib.NanoCallerHipError) as e2:
                db.prepare(chrom, read_fasta(fa, chrom))
            assert str(e2.value).split(": ", 1)[1] == str(e).split(": ", 1)[1]


@pytest.mark.gpu
def test_a_share_of_the_contigs_loads_only_its_part_of_the_file(tmp_path):
    """a rank that owns some contigs of a genome reads and inflates the members between their first record and the next contig's: same packs"""
    from nanocaller_amd.device_bam import DeviceBam, open_device_bam, release
    rng = np.random.default_rng(11)
    refs = [("c%d" % k, 60_000) for k in range(4)]
    seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, ln)) for _, ln in refs]
    recs = []
    for tid in (0, 1, 3):                                                  # c2 has no alignments
        for r in range(700):
            p0 = int(r * 80 + rng.integers(0, 40))
            ln = int(rng.integers(300, 1500))
            ln = min(ln, refs[tid][1] - p0)
            recs.append(dict(tid=tid, name="r%d_%d" % (tid, r), flag=16 * int(rng.integers(0, 2)), pos0=p0, cigar=[("M", ln)], seq=seqs[tid][p0:p0 + ln], tags={}))
    bam, fa = str(tmp_path / "m.bam"), str(tmp_path / "m.fa")
    bamio.write_bam(bam, refs[0][0], refs[0][1], recs, other_refs=refs[1:])
    bamio.write_fasta(fa, refs[0][0], seqs[0], extra=[(n, s_) for (n, _), s_ in zip(refs[1:], seqs[1:])])
    whole = DeviceBam(bam, 0)
    part = DeviceBam(bam, 0, contigs=["c1"])
    last = DeviceBam(bam, 0, contigs=["c3", "c2"])
    assert 0 < part.n_bytes < whole.n_bytes and part.B0 > 0 and last.B1 == whole.file_bytes and last.n_bytes < whole.n_bytes
    for contigs, chrom in ((["c1"], "c1"), (["c3", "c2"], "c3"), (["c0", "c1"], "c0"), (["c0", "c1"], "c1"), (None, "c3")):
        _, prep, _ = _same_pack(bam, fa, chrom, contigs=contigs)
        assert prep["n_kept"] == 700
    empty = DeviceBam(bam, 0, contigs=["c2"]).load()
    assert empty.n_rec == 0 and empty.prepare("c2", seqs[2])["n_kept"] == 0
    release()
    a = open_device_bam(bam, 0, contigs=["c1", "c3"])
    assert open_device_bam(bam, 0, contigs=["c3"]) is a                     # a loaded share that holds the wanted contigs is reused
    assert open_device_bam(bam, 0) is not a
    release()


def _four_contig_bam(tmp_path):
    rng = np.random.default_rng(11)
    refs = [("c%d" % k, 60_000) for k in range(4)]
    seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, ln)) for _, ln in refs]
    recs = []
    for tid in (0, 1, 3):
        for r in range(700):
            p0 = int(r * 80 + rng.integers(0, 40))
            ln = min(int(rng.integers(300, 1500)), refs[tid][1] - p0)
            recs.append(dict(tid=tid, name="r%d_%d" % (tid, r), flag=0, pos0=p0, cigar=[("M", ln)], seq=seqs[tid][p0:p0 + ln], tags={}))
    bam = str(tmp_path / "m.bam")
    bamio.write_bam(bam, refs[0][0], refs[0][1], recs, other_refs=refs[1:])
    return bam


def test_share_plan_of_a_file_that_does_not_fit_at_once(tmp_path):
    """CPU: device_bam.plan_shares cuts the contigs of a call into runs whose part of the file stays under the limit (from the .bai alone)"""
    from nanocaller_amd.device_bam import DeviceIngestUnavailable, contig_spans, plan_shares
    bam = _four_contig_bam(tmp_path)
    size = os.path.getsize(bam)
    names = ["c0", "c1", "c2", "c3"]
    spans, refs = contig_spans(bam)
    assert refs == names and sorted(spans) == ["c0", "c1", "c3"] and spans["c3"][1] == size and spans["c0"][0] < spans["c1"][0] < spans["c3"][0]
    assert plan_shares(bam, names) == [(names, True)]
    assert plan_shares(bam, names, size + 1) == [(names, True)]
    two = spans["c1"][1] - spans["c0"][0]                                 # c0 + c1 fit, c3 does not fit beside them
    assert plan_shares(bam, names, two) == [(["c0", "c1", "c2"], True), (["c3"], True)]
    one = max(hi - lo for lo, hi in spans.values())
    assert plan_shares(bam, names, one) == [(["c0"], True), (["c1", "c2"], True), (["c3"], True)]
    tiny = plan_shares(bam, names, 1000)
    assert tiny == [(["c0"], False), (["c1"], False), (["c2"], True), (["c3"], False)]
    assert plan_shares(bam, ["c3", "c0"], one) == [(["c3"], True), (["c0"], True)]      # the caller's order is kept
    with pytest.raises(ValueError):
        plan_shares(bam, ["nope"])
    os.remove(bam + ".bai")
    with pytest.raises(DeviceIngestUnavailable):
        plan_shares(bam, names)


def _same_indel_sections(bam, fa, chrom, supplementary=False):
    """the indel path's per-read sections made on the device == nc_bam_decode's events + nc_indel_pack_build's arrays (host route)"""
    import ctypes as C
    import torch
    from nanocaller_amd import generate_indel_pileups as gip
    from nanocaller_amd.bam import read_fasta
    from nanocaller_amd.device_bam import DeviceBam
    from nanocaller_amd.pack import pileup_depth_cap
    gip._CONTIGS.clear()
    ctg = gip.decoded_contig(bam, chrom, fa)
    dec = ctg["dec"]
    flag = 0x4 | 0x100 | 0x200 | 0x400 | (0 if supplementary else 0x800)
    keep = pileup_depth_cap(dec["read_start"], dec["read_end"], np.ascontiguousarray((dec["read_flag"] & flag) == 0, np.uint8))
    L = _lib.lib()
    h = C.c_void_p()
    assert L.nc_indel_pack_build(ctg["handle"], _lib.npp(keep), gip.TAIL_CAP, C.byref(h)) == _lib.NC_OK
    v = _lib.IndelPackArraysC()
    L.nc_indel_pack_view(h, C.byref(v))

    def arr(ptr, cnt, dt):
        return np.frombuffer((C.c_char * (int(cnt) * np.dtype(dt).itemsize)).from_address(ptr), dt).copy() if cnt and ptr else np.zeros(0, dt)
    want = dict(ev_off=arr(v.ev_off, v.n_reads + 1, np.int32), ev_pos=arr(v.ev_pos, v.n_events, np.int32), ev_len=arr(v.ev_len, v.n_events, np.int32),
                ins_off=arr(v.ins_off, v.n_events + 1, np.int32), ins_bases=arr(v.ins_bases, v.n_ins_bases, np.uint8),
                tail_off=arr(v.tail_off, v.n_reads + 1, np.int32), tail_bases=arr(v.tail_bases, v.n_tail_bases, np.uint8),
                read_ps=arr(v.read_ps, v.n_reads, np.int32), read_hap=arr(v.read_hap, v.n_reads, np.uint8), read_flag=arr(v.read_flag, v.n_reads, np.uint8))
    K = int(v.n_reads)
    L.nc_indel_pack_free(h)
    db = DeviceBam(bam, 0).load()
    dp = db.pack(db.prepare(chrom, read_fasta(fa, chrom), supplementary=supplementary), indel=True, tail_cap=gip.TAIL_CAP)
    torch.cuda.synchronize()
    assert dp.events["n_reads"] == dp.reads["n_reads"] == K
    got = dict(ev_off=dp.events["ev_off"], ev_pos=dp.events["ev_pos"], ev_len=dp.events["ev_len"], read_hap=dp.events["read_hap"], **dp.indel)
    n_ev = int(want["ev_off"][-1]) if K else 0
    for k, w in want.items():
        g = got[k].cpu().numpy()
        n = {"ev_pos": n_ev, "ev_len": n_ev, "ins_bases": want["ins_bases"].size, "tail_bases": want["tail_bases"].size, "read_ps": K, "read_hap": K, "read_flag": K}.get(k, w.size)
        assert np.array_equal(g[:n].astype(w.dtype), w[:n]), (k, g[:10], w[:10])
    kk = np.flatnonzero(keep)
    assert np.array_equal(dp.reads["rd_start"].cpu().numpy()[:K], dec["read_start"][kk]) and np.array_equal(dp.reads["rd_end"].cpu().numpy()[:K], dec["read_end"][kk])
    gip._CONTIGS.clear()
    return n_ev, want


@pytest.mark.gpu
def test_indel_sections_from_the_record_stream_equal_the_host_routes(tmp_path):
    w = bamio.make_bam_world()
    recs = bamio.world_to_records(w, np.random.Generator(np.random.PCG64(1)))
    bam, fa = str(tmp_path / "w.bam"), str(tmp_path / "w.fa")
    bamio.write_bam(bam, w.chrom, w.length, recs)
    bamio.write_fasta(fa, w.chrom, w.ref)
    n_ev, want = _same_indel_sections(bam, fa, w.chrom)
    assert n_ev > 200 and want["ins_bases"].size > 50
    ops = "MIDNSHP=X"
    ref = "ACGT" * 500
    seq = "ACGTACGTAC" + "GG" + "TACGTACG"
    real = [("M", 10), ("I", 2), ("D", 3), ("M", 8)]
    cg = [(ln << 4) | ops.index(op) for op, ln in real]
    recs = [dict(name="noseq", flag=0, pos0=40, cigar=[("M", 300), ("D", 4), ("M", 200)], seq="", tags={}),
            dict(name="short", flag=0, pos0=60, cigar=[("M", 10), ("I", 4), ("M", 20)], seq="ACGTACGTACGG", tags={"PS": 5}),
            dict(name="clip", flag=16, pos0=99, cigar=[("H", 5), ("S", 2), ("M", 4), ("I", 3), ("M", 2), ("D", 2), ("M", 3), ("=", 2), ("X", 1), ("I", 2), ("S", 4)],
                 seq="TTACGTAAAGGCATACGAGGTTTT", tags={"HP": 2, "PS": 70000}),
            dict(name="longcig", flag=0, pos0=100, cigar=[("S", len(seq)), ("N", 21)], seq=seq, tags={"HP": 2, "CG": cg, "PS": 77}),
            dict(name="ins1st", flag=0, pos0=120, cigar=[("S", 3), ("I", 2), ("M", 5), ("D", 1), ("M", 2)], seq="TTTGGACGTNAC" + "A" * 300, tags={}),
            dict(name="longtail", flag=0, pos0=300, cigar=[("M", 20), ("S", 400)], seq="ACGT" * 105, tags={"HP": 1})]
    bam2, fa2 = str(tmp_path / "c.bam"), str(tmp_path / "c.fa")
    bamio.write_bam(bam2, "chrT", len(ref), recs)
    bamio.write_fasta(fa2, "chrT", ref)
    n_ev, want = _same_indel_sections(bam2, fa2, "chrT")
    assert want["read_flag"].tolist() == [1, 0, 0, 0, 0, 0] and want["tail_off"][-1] > 272


@pytest.mark.gpu
def test_csi_indexed_bam_takes_the_device_route_too(tmp_path):
    """no .bai: the chain starts come from the .csi's bin offsets and chunk begins; same packs, whole file and a share of the contigs"""
    from nanocaller_amd.device_bam import contig_spans
    w = bamio.make_bam_world()
    recs = bamio.world_to_records(w, np.random.Generator(np.random.PCG64(1)))
    extra = [dict(tid=1, name="o%d" % k, flag=0, pos0=10 * k, cigar=[("M", 50)], seq="ACGTA" * 10, tags={}) for k in range(40)]
    bam, fa = str(tmp_path / "w.bam"), str(tmp_path / "w.fa")
    bamio.write_bam(bam, w.chrom, w.length, recs + extra, other_refs=[("chrOther", 1000)], write_bai=False, write_csi=True)
    bamio.write_fasta(fa, w.chrom, w.ref, extra=[("chrOther", "ACGT" * 250)])
    assert os.path.exists(bam + ".csi") and not os.path.exists(bam + ".bai")
    spans, names = contig_spans(bam)
    assert names == [w.chrom, "chrOther"] and sorted(spans) == sorted(names)
    _, prep, _ = _same_pack(bam, fa, w.chrom)
    assert prep["n_kept"] > 100
    _, prep, _ = _same_pack(bam, fa, "chrOther", contigs=["chrOther"])
    assert prep["n_kept"] == 40
    _same_indel_sections(bam, fa, w.chrom)


@pytest.mark.gpu
def test_a_member_with_a_damaged_crc_is_reported_not_called_from(tmp_path, monkeypatch):
    """a BGZF member whose bytes still inflate to the announced length but whose CRC-32 does not match (one stored trailer bit flipped: the cheapest
    way to make one) -- htslib refuses the block; the device route reports it too (nc_bgzf_crc_device), and NC_BGZF_CRC=0 restores round 4's
    lengths-only check"""
    from nanocaller_amd import device_bam
    w = bamio.make_bam_world()
    recs = bamio.world_to_records(w, np.random.Generator(np.random.PCG64(2)))
    bam = str(tmp_path / "w.bam")
    bamio.write_bam(bam, w.chrom, w.length, recs)
    raw = bytearray(open(bam, "rb").read())
    bsize = int.from_bytes(raw[16:18], "little") + 1
    second = bsize                                                                                # damage the SECOND member (the first holds the header the host reads)
    b2 = int.from_bytes(raw[second + 16:second + 18], "little") + 1
    raw[second + b2 - 8] ^= 0x04                                                                  # lowest byte of its CRC-32
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(raw))
    import shutil
    shutil.copy(bam + ".bai", bad + ".bai")
    device_bam.release()
    with pytest.raises(Exception, match="CRC-32"):
        device_bam.open_device_bam(bad, 0)
    device_bam.release()
    monkeypatch.setattr(device_bam, "CHECK_CRC", False)
    db = device_bam.open_device_bam(bad, 0)                                                       # (the bytes are intact: only the stored CRC is not)
    assert db.n_rec > 100
    device_bam.release()


@pytest.mark.gpu
def test_a_device_without_room_sends_the_file_to_the_host_route(tmp_path, monkeypatch):
    """the ingest is sized from what the device has free (resident_limit: half of free + pooled memory), and a failed allocation inside the loader is
    DeviceIngestUnavailable -- the caller's host route -- not a crash of the worker"""
    import torch
    from nanocaller_amd import device_bam
    w = bamio.make_bam_world()
    recs = bamio.world_to_records(w, np.random.Generator(np.random.PCG64(3)))
    bam = str(tmp_path / "w.bam")
    bamio.write_bam(bam, w.chrom, w.length, recs)
    device_bam.release(buffers=True)
    lim = device_bam.resident_limit(0)
    free, total = torch.cuda.mem_get_info(0)
    free += torch.cuda.memory_reserved(0) - torch.cuda.memory_allocated(0)                          # torch's cached blocks are served to the loader too
    assert (1 << 30) <= lim <= min(device_bam.MAX_RESIDENT, free // 2 + (1 << 20))
    monkeypatch.setattr(device_bam, "resident_limit", lambda device=0: 1024)
    with pytest.raises(device_bam.DeviceIngestUnavailable):
        device_bam.open_device_bam(bam, 0)
    monkeypatch.undo()

    def oom(*a, **k):
        raise torch.cuda.OutOfMemoryError("HIP out of memory (simulated)")
    monkeypatch.setattr(device_bam, "_work_buffer", oom)
    device_bam.release(buffers=True)
    with pytest.raises(device_bam.DeviceIngestUnavailable):
        device_bam.open_device_bam(bam, 0)
    monkeypatch.undo()
    device_bam.release(buffers=True)
    assert device_bam.open_device_bam(bam, 0).n_rec > 100
    device_bam.release(buffers=True)


@pytest.mark.gpu
def test_a_released_share_in_torchs_cache_does_not_shrink_the_next_share(monkeypatch):
    """a share whose stream buffer exceeds POOL_MAX goes back to torch's caching allocator on release; mem_get_info no longer shows it
    as free, torch.empty still gets it.  resident_limit() counts it, so share 2..N of a genome-sized file keep the device route; plan_shares takes
    the worker's device."""
    import torch
    from nanocaller_amd import device_bam
    device_bam.release(buffers=True)
    torch.cuda.empty_cache()
    monkeypatch.setattr(device_bam, "MAX_RESIDENT", 1 << 42)
    lim0 = device_bam.resident_limit(0)
    free0, _ = torch.cuda.mem_get_info(0)
    t = torch.empty(device_bam.POOL_MAX + (4 << 30), dtype=torch.uint8, device="cuda:0")         # a stream buffer too large for the module's pool
    del t                                                                                         # -> torch's cache, not the driver
    free1, _ = torch.cuda.mem_get_info(0)
    assert free1 < free0 - device_bam.POOL_MAX                                                    # invisible as 'free' ...
    lim1 = device_bam.resident_limit(0)
    assert lim1 >= lim0 - (256 << 20), (lim0, lim1)                                               # ... and still counted
    again = torch.empty(device_bam.POOL_MAX + (4 << 30), dtype=torch.uint8, device="cuda:0")      # served from the cache
    assert torch.cuda.mem_get_info(0)[0] >= free1 - (64 << 20)
    del again
    torch.cuda.empty_cache()
    import inspect
    assert "device" in inspect.signature(device_bam.plan_shares).parameters


@pytest.mark.gpu
def test_the_ont_like_bench_bam_decodes_to_the_workload_it_was_written_from(tmp_path):
    """tools/ont_like_bam.py (the from-BAM leg's file: qualities, ~700 CIGAR operations per read, soft clips, NM / MD / HP / PS tags): the device route's
    pack of it equals the host route's byte for byte, and both equal the synthetic workload the records were cut from (same aligned bases, deleted
    positions as code 4; insertions and clips have no column) -- so the bench's file is a valid BAM of the headline workload"""
    import sys
    import zlib
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ont_like_bam
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import make_device_workload
    eng = get_engine(0)
    L = 400_000
    bam, refs, fasta, st = ont_like_bam.make_files(eng, str(tmp_path), 2, L, depth=20.0, seed0=4100, level=1)
    fa = str(tmp_path / "b.fa")
    bamio.write_fasta(fa, fasta[0][0], fasta[0][1], extra=fasta[1:])
    assert st["cigar_ops"] > 300 * st["reads"] and st["bam_bytes"] > 5_000_000
    raw = open(bam, "rb").read()
    o, total = 0, 0
    while o < len(raw):                                                                           # every member: a valid gzip member with the right CRC and size
        bsize = int.from_bytes(raw[o + 16:o + 18], "little") + 1
        d = zlib.decompress(raw[o + 18:o + bsize - 8], -15)
        assert zlib.crc32(d) == int.from_bytes(raw[o + bsize - 8:o + bsize - 4], "little") and len(d) == int.from_bytes(raw[o + bsize - 4:o + bsize], "little")
        total += len(d)
        o += bsize
    assert o == len(raw)
    for k, (name, _) in enumerate(refs):
        got, prep, world = _same_pack(bam, fa, name)
        pack, info = make_device_workload(eng, L, depth=20.0, tech="ont", seed=4100 + k)
        assert prep["n_reads"] == info["n_reads"] and np.array_equal(prep["read_start"], info["read_start"])
        a, b = got.codes.cpu().numpy(), pack.codes.cpu().numpy()
        assert a.shape == b.shape and np.array_equal(a, b), (k, int((a != b).sum()))
        assert int((world.read_hp > 0).sum()) > 0.5 * info["n_reads"] if hasattr(world, "read_hp") else True
