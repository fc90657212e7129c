"""The native BGZF / BAM / BAI reader (csrc/nc_bam.cpp) on a file it has no shared author with at the byte level:
tests/golden/spec.bam(.bai) was assembled from the SAM/BAM specification by oracle/tools/make_spec_bam.py with struct + zlib only -- not by
tests/bamio.py, the writer behind every other ingest test.  spec_expected.json holds that script's INPUT record table; this file walks the
CIGARs itself (its own statement of pileup columns and '+n' / '-n' markers, generate_SNP_pileups.py:104,162 / generate_indel_pileups.py:216-231)
and compares with what the reader returns: whole contigs, the second reference, index-driven region queries, the parallel region decode."""
import json
import os
import zlib

import numpy as np
import pytest

from nanocaller_amd.bam import BamFile, decode_parallel

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BAM = os.path.join(G, "spec.bam")
CODE = {"A": 0, "G": 1, "T": 2, "C": 3}


@pytest.fixture(scope="module")
def expected():
    return json.load(open(os.path.join(G, "spec_expected.json")))


def _walk(rec):
    """-> (start, end, codes per spanned position, [(column, +ins / -del)], query codes): the reader's contract, restated from the CIGAR"""
    pos0, seq = rec["pos0"], rec["seq"]
    qcodes = [CODE.get(c, 4) for c in seq]
    qlen = sum(n for op, n in rec["cigar"] if op in "MIS=X")
    has_seq = qlen <= len(seq)
    codes, events, rp, qp = [], [], 0, 0
    for op, n in rec["cigar"]:
        if op in "M=X":
            codes += qcodes[qp:qp + n] if has_seq else [4] * n
            rp += n
            qp += n
        elif op == "I":
            if rp > 0:
                events.append((pos0 + rp, n))
            qp += n
        elif op == "D":
            if rp > 0:
                events.append((pos0 + rp, -n))
            codes += [4] * n
            rp += n
        elif op == "S":
            qp += n
    return pos0 + 1, pos0 + 1 + rp, codes, events, qcodes


def _compare(d, recs, with_seq=False):
    assert len(d["read_start"]) == len(recs)
    for k, r in enumerate(recs):
        s, e, codes, events, qcodes = _walk(r)
        assert (int(d["read_start"][k]), int(d["read_end"][k])) == (s, e), r["name"]
        assert d["names"][k] == r["name"] and int(d["read_flag"][k]) & 0xFFFF == r["flag"]
        o = int(d["read_off"][k])
        assert d["codes"][o:o + e - s].tolist() == codes, r["name"]
        e0, e1 = int(d["ev_off"][k]), int(d["ev_off"][k + 1])
        assert list(zip(d["ev_pos"][e0:e1].tolist(), d["ev_len"][e0:e1].tolist())) == events, r["name"]
        hp = r["hp"] if r["hp"] in (1, 2) else 0
        assert int(d["hap"][k]) == hp and int(d["ps"][k]) == (r["ps"] if hp else 0), r["name"]
        if with_seq:
            q0, q1 = int(d["seq_off"][k]), int(d["seq_off"][k + 1])
            assert d["seq"][q0:q1].tolist() == qcodes, r["name"]


def test_the_fixture_is_what_the_specification_says(expected):
    """independent of the reader: gzip members with the BC subfield whose BSIZE chains to the next member, CRC32 and ISIZE right, the EOF marker
    last, all three deflate block types present"""
    raw = open(BAM, "rb").read()
    o, n, kinds, out = 0, 0, set(), b""
    while o < len(raw):
        assert raw[o:o + 4] == b"\x1f\x8b\x08\x04" and raw[o + 12:o + 14] == b"BC"
        bsize = int.from_bytes(raw[o + 16:o + 18], "little") + 1
        payload = raw[o + 18:o + bsize - 8]
        data = zlib.decompress(payload, -15)
        assert zlib.crc32(data) == int.from_bytes(raw[o + bsize - 8:o + bsize - 4], "little") and len(data) == int.from_bytes(raw[o + bsize - 4:o + bsize], "little")
        kinds.add((payload[0] >> 1) & 3)
        out += data
        o += bsize
        n += 1
    assert n == expected["members"] + 1 and len(data) == 0 and kinds == {0, 1, 2}
    assert out[:4] == b"BAM\x01"


@pytest.mark.parametrize("keep_seq", [False, True])
def test_whole_contigs(expected, keep_seq):
    bf = BamFile(BAM)
    assert bf.references == [r[0] for r in expected["refs"]] and bf.lengths == [r[1] for r in expected["refs"]] and bf.has_index
    for tid, (name, _) in enumerate(expected["refs"]):
        recs = [r for r in expected["records"] if r["ref"] == tid and not (r["flag"] & 4) and sum(n for op, n in r["cigar"] if op in "MD=X") > 0]
        d = decode_parallel(BAM, name, keep_seq=keep_seq, threads=1)
        _compare(d, recs, keep_seq)
    assert len([r for r in expected["records"] if r["ref"] == 0]) > 100


def test_region_queries_through_the_index(expected):
    """the BAI of the fixture (bins by reg2bin, chunks as virtual offsets into members the records straddle, linear index) drives the seek"""
    recs = [r for r in expected["records"] if r["ref"] == 0 and not (r["flag"] & 4) and sum(n for op, n in r["cigar"] if op in "MD=X") > 0]
    spans = [_walk(r)[:2] for r in recs]
    for a, b in ((1, 400), (100, 100), (16_000, 17_000), (16_384, 16_385), (30_000, 50_000), (69_000, 70_000), (1, 70_000)):
        want = [r for r, (s, e) in zip(recs, spans) if s <= b and e > a]
        d = decode_parallel(BAM, "ctgA", a, b, threads=1)
        _compare(d, want)
    assert len(decode_parallel(BAM, "ctgB", 8_000, 9_000, threads=1)["read_start"]) == 0


def test_parallel_regions_equal_the_sequential_decode(expected):
    a = decode_parallel(BAM, "ctgA", keep_seq=True, threads=1)
    b = decode_parallel(BAM, "ctgA", keep_seq=True, threads=5, min_region=9_000)
    for k in ("read_start", "read_end", "read_flag", "read_off", "codes", "ev_off", "ev_pos", "ev_len", "hap", "ps", "seq_off", "seq"):
        assert np.array_equal(a[k], b[k]), k
    assert a["names"] == b["names"]
