"""BGZF + CSI writer: the files are read back with an independent reader that follows the CSIv1 / tabix /
BGZF specifications; region queries through the index must return exactly the overlapping records.  htslib / tabix /
bcftools are absent from this image; what htslib itself wrote and the image holds (the reference's bgzipped, tabix-indexed
BED files) pins the BGZF reader and the binning scheme -- see the last test."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from nanocaller_amd import _lib, vcfio


def _read_block(raw, coff):
    """-> (payload bytes, next compressed offset) of the BGZF block at `coff`"""
    assert raw[coff:coff + 4] == b"\x1f\x8b\x08\x04"
    xlen = struct.unpack_from("<H", raw, coff + 10)[0]
    extra = raw[coff + 12:coff + 12 + xlen]
    bsize, p = None, 0
    while p < xlen:
        si1, si2, slen = extra[p], extra[p + 1], struct.unpack_from("<H", extra, p + 2)[0]
        if (si1, si2, slen) == (66, 67, 2):
            bsize = struct.unpack_from("<H", extra, p + 4)[0] + 1
        p += 4 + slen
    cdata = raw[coff + 12 + xlen:coff + bsize - 8]
    crc, isize = struct.unpack_from("<II", raw, coff + bsize - 8)
    data = zlib.decompress(cdata, -15)
    assert len(data) == isize and (zlib.crc32(data) & 0xffffffff) == crc
    return data, coff + bsize


def _read_range(raw, vbeg, vend):
    """bytes between two virtual offsets"""
    out, coff, off = [], vbeg >> 16, vbeg & 0xffff
    while True:
        data, nxt = _read_block(raw, coff)
        if coff == vend >> 16:
            out.append(data[off:vend & 0xffff])
            break
        out.append(data[off:])
        coff, off = nxt, 0
    return b"".join(out)


def _parse_csi(blob):
    assert blob[:4] == b"CSI\1"
    min_shift, depth, l_aux = struct.unpack_from("<3i", blob, 4)
    aux = blob[16:16 + l_aux]
    fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<7i", aux, 0)
    names = aux[28:28 + l_nm].split(b"\0")[:-1]
    p = 16 + l_aux
    n_ref = struct.unpack_from("<i", blob, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", blob, p)[0]
        p += 4
        bins = {}
        for _ in range(n_bin):
            b, loff, n_chunk = struct.unpack_from("<IQi", blob, p)
            p += 16
            chunks = [struct.unpack_from("<QQ", blob, p + 16 * k) for k in range(n_chunk)]
            p += 16 * n_chunk
            bins[b] = (loff, chunks)
        refs.append(bins)
    n_no_coor = struct.unpack_from("<Q", blob, p)[0]
    assert p + 8 == len(blob)
    return dict(min_shift=min_shift, depth=depth, conf=(fmt, col_seq, col_beg, col_end, meta, skip), names=[n.decode() for n in names],
                refs=refs, n_no_coor=n_no_coor)


def _reg2bins(beg, end, min_shift, depth):
    """bins that may hold records overlapping [beg, end) (hts_reg2bins)"""
    out, s, t = [], min_shift + depth * 3, 0
    end -= 1
    for lv in range(depth + 1):
        out += list(range(t + (beg >> s), t + (end >> s) + 1))
        t += 1 << (lv * 3)
        s -= 3
    return out


@pytest.mark.parametrize("seed,n", [(1, 40_000), (2, 7)])
def test_bgzf_csi_roundtrip_and_region_queries(tmp_path, seed, n):
    rng = np.random.Generator(np.random.PCG64(seed))
    contigs = ["chr20", "chrX", "scaffold_7"]
    tid = np.sort(rng.integers(0, 3, size=n))
    pos = np.zeros(n, np.int64)
    for r in range(3):
        m = tid == r
        pos[m] = np.sort(rng.integers(1, 90_000_000 if r == 0 else 400_000, size=int(m.sum())))
    ref_len = np.where(rng.random(n) < 0.8, 1, rng.integers(2, 40, size=n))
    lines = [("%s\t%d\t.\t%s\tA\t%.3f\tPASS\tX=%d\tGT\t0/1\n" % (contigs[t], p, "C" * rl, rng.random() * 99, i)).encode()
             for i, (t, p, rl) in enumerate(zip(tid, pos, ref_len))]
    header = "##fileformat=VCFv4.2\n" + "".join("##contig=<ID=%s>\n" % c for c in contigs) + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"
    path = str(tmp_path / "t.vcf.gz")
    vcfio.write_vcf_gz_with_csi(path, header, b"".join(lines), contigs, tid, pos, ref_len, np.array([len(x) for x in lines]))
    raw = open(path, "rb").read()
    assert gzip.decompress(raw) == header.encode() + b"".join(lines)          # a valid multi-member gzip stream
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    idx = _parse_csi(gzip.decompress(open(path + ".csi", "rb").read()))
    assert (idx["min_shift"], idx["depth"], idx["conf"], idx["names"], idx["n_no_coor"]) == (14, 5, (2, 1, 2, 0, 35, 0), contigs, 0)
    meta_bin = ((1 << 18) - 1) // 7 + 1
    for r in range(3):
        assert idx["refs"][r][meta_bin][1][1] == (int((tid == r).sum()), 0)     # mapped / unmapped counts of the pseudo-bin
    queries = [(0, 0, 1 << 29), (1, 0, 1 << 29), (2, 100, 101)]
    for _ in range(60):
        r = int(rng.integers(0, 3))
        a = int(rng.integers(0, 90_000_000 if r == 0 else 400_000))
        queries.append((r, a, a + int(rng.choice([1, 50, 20_000, 3_000_000]))))
    for (r, a, b) in queries:
        want = [lines[i] for i in np.flatnonzero((tid == r) & (pos - 1 < b) & (pos - 1 + ref_len > a))]
        bins = idx["refs"][r]
        got = []
        seen = set()
        lin_min = min([bins[x][0] for x in _reg2bins(a, b, 14, 5) if x in bins] or [0])
        for x in _reg2bins(a, b, 14, 5):
            if x not in bins:
                continue
            for (cb, ce) in bins[x][1]:
                if ce <= lin_min:
                    continue
                for rec in _read_range(raw, cb, ce).splitlines(keepends=True):
                    f = rec.split(b"\t", 4)
                    p0 = int(f[1]) - 1
                    if f[0].decode() == contigs[r] and p0 < b and p0 + len(f[3]) > a and rec not in seen:
                        seen.add(rec)
                        got.append((p0, rec))
        got = [rec for _, rec in sorted(got, key=lambda t: t[0])]
        assert sorted(got) == sorted(want), (r, a, b, len(got), len(want))


def test_bgzf_empty_and_block_boundaries():
    for n in (0, 1, vcfio.BGZF_BLOCK - 1, vcfio.BGZF_BLOCK, vcfio.BGZF_BLOCK + 1, 3 * vcfio.BGZF_BLOCK):
        data = bytes(np.random.Generator(np.random.PCG64(n)).integers(0, 256, size=n, dtype=np.uint8))
        comp, coff = vcfio.bgzf_compress(data)
        assert gzip.decompress(comp.tobytes()) == data
        assert len(coff) == (n + vcfio.BGZF_BLOCK - 1) // vcfio.BGZF_BLOCK + 1 and coff[-1] == len(comp) - 28


# ---------------------------------------------------------------------------------------------------------------------------
# Files written by htslib itself (bgzip + tabix): the four centromere / telomere BED files that ship with the reference
# (nanocaller_src/release_data/bed_files, data the reference's `--exclude_bed hg38` resolves to, NanoCaller:21-22) are the only
# htslib output in this image.  They pin the BGZF layer of the BAM reader and the binning scheme of the index writer.
def _tbi(raw):
    """tabix index (SAM spec section 5 / tabix.pdf) -> (names, [per reference {bin: [(beg, end)]}], [linear index])"""
    assert raw[:4] == b"TBI\x01"
    n_ref, fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<8i", raw, 4)
    names = raw[36:36 + l_nm].split(b"\0")[:n_ref]
    p = 36 + l_nm
    bins, lin = [], []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", raw, p); p += 4
        d = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", raw, p); p += 8
            d[b] = [struct.unpack_from("<QQ", raw, p + 16 * c) for c in range(n_chunk)]
            p += 16 * n_chunk
        bins.append(d)
        (n_intv,) = struct.unpack_from("<i", raw, p); p += 4
        lin.append(struct.unpack_from("<%dQ" % n_intv, raw, p)); p += 8 * n_intv
    return dict(fmt=fmt, cols=(col_seq, col_beg, col_end)), [n.decode() for n in names], bins, lin


@pytest.mark.parametrize("genome", ["hg38", "hg19", "mm10", "mm39"])
def test_htslib_written_bgzf_files_and_their_tabix_bins(genome):
    from nanocaller_amd import generate_SNP_pileups as gsp
    path = os.path.join(gsp.RELEASE_BED_DIR, "%s_centro_telo.bed.gz" % genome)
    text = vcfio.bgzf_read(path)
    assert text == gzip.decompress(open(path, "rb").read()) and text.count(b"\n") > 40
    with pytest.raises(_lib.NanoCallerHipError):
        vcfio.bgzf_read(__file__)                                                   # not BGZF: refused, not garbage
    hdr, names, bins, lin = _tbi(vcfio.bgzf_read(path + ".tbi"))
    assert hdr["cols"] == (1, 2, 0) and hdr["fmt"] == 2          # indexed with tabix's VCF preset: a row is the position in column 2
    # every row sits in the bin reg2bin gives it, inside one of that bin's chunks (virtual offset = block offset << 16 | offset
    # in the block; these files are one block), and not before its 16 kb window's linear-index entry
    off, n_rows = 0, 0
    r_tid, r_beg, r_off, r_end = [], [], [], []
    for line in text.split(b"\n")[:-1]:
        c, a, b = line.split(b"\t")[:3]
        tid = names.index(c.decode())
        r_tid.append(tid); r_beg.append(int(a) - 1); r_off.append(off); r_end.append(off + len(line) + 1)
        beg0 = int(a) - 1                                                           # 1-based POS -> [POS - 1, POS)
        bn = int(vcfio.reg2bin(np.array([beg0]), np.array([beg0 + 1]))[0])
        assert bn in bins[tid], (line, bn)
        assert any(lo <= off < hi for lo, hi in bins[tid][bn]), (line, bins[tid][bn])
        assert lin[tid][beg0 >> 14] <= off
        off += len(line) + 1
        n_rows += 1
    assert sum(len(ch) for d in bins for k, ch in d.items() if k != 37450) >= len(names) and n_rows > 40
    # the index writer of the VCF path (vcfio.csi_bytes), fed the same rows and offsets, lists the bins, chunks, pseudo-bin
    # statistics and aux block that tabix wrote for this file
    r_end[-1] = (os.path.getsize(path) - 28) << 16       # the data block ends with the last row: "tell" = start of the next (EOF) block
    ours = _parse_csi(vcfio.csi_bytes(names, r_tid, r_beg, np.array(r_beg) + 1, r_off, r_end))
    assert ours["names"] == names and ours["conf"] == (2, 1, 2, 0, ord("#"), 0)
    for tid in range(len(names)):
        assert {k: [tuple(c) for c in v[1]] for k, v in ours["refs"][tid].items()} == {k: [tuple(c) for c in v] for k, v in bins[tid].items()}
        for k, (loff, _) in ours["refs"][tid].items():
            if k != 37450:
                assert loff == lin[tid][min(k - 4681, len(lin[tid]) - 1)]           # level-5 bins: loffset = the window's entry
    # and the exclusion rows the featuriser derives from it, by the path and by the reference CLI's short name
    rows = gsp._exclude_rows({"exclude_bed": genome}, "chr1")
    assert rows == gsp._exclude_rows({"exclude_bed": path}, "chr1") and len(rows) >= 2
    exp = tuple((int(t[1]), int(t[2])) for t in (ln.split() for ln in text.decode().splitlines()) if t[0] == "chr1")
    assert rows == exp
