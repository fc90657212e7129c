#!/usr/bin/env python3
"""A BAM + BAI pair assembled BYTE BY BYTE from the SAM/BAM specification (SAMv1.pdf sections 4.1 BGZF, 4.2 BAM, 5.2 BAI) with nothing
but struct and zlib -- no htslib here, and deliberately NOT through tests/bamio.py (the writer every other ingest test uses): this breaks
the loop of the repository's reader only ever meeting the repository's writer.

    python oracle/tools/make_spec_bam.py  ->  tests/golden/spec.bam, spec.bam.bai, spec_expected.json

The expected values in the JSON are the INPUTS of this script (the record table below), not something decoded back.  What the file
exercises, each item cited to the specification:
  * BGZF members with the BC extra subfield, several per file, alignment records that straddle member boundaries (4.1: "a BAM record may
    span two or more blocks"), one member stored uncompressed (deflate BTYPE 00), one made with fixed Huffman codes (Z_FIXED), the rest
    dynamic, and the 28-byte EOF marker;
  * two reference sequences; records of the second, and unmapped / secondary / qc-fail / duplicate / supplementary flags;
  * CIGAR operations M I D S H = X with lengths on both sides of 16, hard and soft clips on either end, an insertion next to a deletion;
  * SEQ with every one of the 16 four-bit codes (4.2.3 '=ACMGRSVTWYHKDBN'), odd lengths (the last nibble is padding), QUAL 0xFF ("absent");
  * auxiliary fields of every fixed-size type before and after the ones the reader wants (A c C s S i I f, Z, H, B arrays), HP as type C
    and as type i, PS as type I and as type s -- whatever width a writer happened to choose (4.2.4);
  * bin numbers by reg2bin (5.3), the 16 kb linear index, chunk lists with virtual file offsets (coffset << 16 | uoffset), the
    pseudo-bin 37450 with the mapped / unmapped counts, and n_no_coor."""
import json
import os
import struct
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden")
SEQ_CODES = "=ACMGRSVTWYHKDBN"
CIG_OPS = "MIDNSHP=X"


def reg2bin(beg, end):
    """SAMv1 5.3 (0-based, end exclusive)"""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def ref_span(cigar):
    return sum(n for op, n in cigar if op in "MDN=X")


def aux(tag, typ, val):
    t = tag.encode()
    if typ == "A":
        return t + b"A" + val.encode()
    if typ in "cCsSiIf":
        return t + typ.encode() + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[typ], val)
    if typ == "Z":
        return t + b"Z" + val.encode() + b"\0"
    if typ == "H":
        return t + b"H" + val.encode() + b"\0"
    if typ == "B":
        sub, arr = val
        return t + b"B" + sub.encode() + struct.pack("<I", len(arr)) + b"".join(
            struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub], v) for v in arr)
    raise ValueError(typ)


def record(rec, tid):
    name = rec["name"].encode() + b"\0"
    cig = rec["cigar"]
    seq = rec["seq"]
    pos = rec["pos0"]
    end = pos + max(1, ref_span(cig))
    packed = bytearray()
    for i in range(0, len(seq), 2):
        hi = SEQ_CODES.index(seq[i])
        lo = SEQ_CODES.index(seq[i + 1]) if i + 1 < len(seq) else 0
        packed.append(hi << 4 | lo)
    qual = bytes([0xFF] * len(seq)) if rec.get("qual") is None else bytes(rec["qual"])
    body = struct.pack("<iiBBHHHIiii", tid, pos, len(name), rec.get("mapq", 60), reg2bin(pos, end), len(cig), rec["flag"], len(seq), -1, -1, 0)
    body += name + b"".join(struct.pack("<I", n << 4 | CIG_OPS.index(op)) for op, n in cig) + bytes(packed) + qual
    body += b"".join(aux(*a) for a in rec.get("aux", ()))
    return struct.pack("<i", len(body)) + body


def bgzf_member(data, how):
    """one BGZF block (4.1): gzip header with FEXTRA / BC subfield, raw deflate payload, CRC32, ISIZE"""
    if how == "stored":
        comp = zlib.compressobj(0, zlib.DEFLATED, -15)
    elif how == "fixed":
        comp = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    else:
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = comp.compress(data) + comp.flush()
    bsize = 12 + 6 + len(payload) + 8 - 1
    assert bsize < 65536
    head = struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize)
    return head + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def make():
    import random
    rng = random.Random(20260929)
    refs = [("ctgA", 70_000), ("ctgB", 9_000)]

    def rand_seq(n, alphabet="ACGT"):
        return "".join(rng.choice(alphabet) for _ in range(n))
    recs = []
    # hand-made records: every CIGAR operation the pileup path meets, clips on both ends, all 16 SEQ codes, odd lengths, every aux type
    recs.append(dict(ref=0, name="allcodes", flag=0, pos0=100, cigar=[("M", 16)], seq=SEQ_CODES,
                     aux=[("XA", "A", "q"), ("Xc", "c", -5), ("HP", "C", 1), ("Xs", "s", -300), ("PS", "I", 101), ("XS", "S", 65000), ("Xf", "f", 1.5)]))
    recs.append(dict(ref=0, name="clips", flag=16, pos0=120, cigar=[("H", 7), ("S", 3), ("M", 20), ("I", 2), ("M", 5), ("D", 17), ("M", 9), ("S", 4), ("H", 2)],
                     seq=rand_seq(3 + 20 + 2 + 5 + 9 + 4), aux=[("XZ", "Z", "text with spaces"), ("XH", "H", "1AE301"), ("HP", "i", 2), ("PS", "s", 101),
                                                                 ("XB", "B", ("S", [1, 2, 65535])), ("Xi", "i", -70000)]))
    recs.append(dict(ref=0, name="eqx_odd", flag=0, pos0=130, cigar=[("=", 10), ("X", 1), ("=", 6), ("I", 16), ("D", 1), ("=", 20)], seq=rand_seq(53),
                     aux=[("XB", "B", ("c", [-1, 2])), ("XF", "B", ("f", [0.5])), ("PS", "I", 4_000_000_000 % (1 << 31)), ("HP", "C", 2)]))
    recs.append(dict(ref=0, name="ins_del_adjacent", flag=0, pos0=200, cigar=[("M", 30), ("I", 1), ("D", 2), ("M", 30), ("D", 16), ("I", 15), ("M", 12)],
                     seq=rand_seq(30 + 1 + 30 + 15 + 12), aux=[]))
    recs.append(dict(ref=0, name="secondary", flag=256, pos0=210, cigar=[("M", 40)], seq=rand_seq(40), aux=[("HP", "C", 1)]))
    recs.append(dict(ref=0, name="qcfail", flag=512, pos0=215, cigar=[("M", 40)], seq=rand_seq(40), aux=[]))
    recs.append(dict(ref=0, name="duplicate", flag=1024 | 16, pos0=220, cigar=[("M", 40)], seq=rand_seq(40), aux=[]))
    recs.append(dict(ref=0, name="suppl", flag=2048, pos0=225, cigar=[("H", 100), ("M", 35)], seq=rand_seq(35), aux=[("HP", "C", 2), ("PS", "I", 225)]))
    recs.append(dict(ref=0, name="unmapped_placed", flag=4, pos0=230, cigar=[], seq=rand_seq(21), aux=[]))
    recs.append(dict(ref=0, name="no_seq", flag=0, pos0=240, cigar=[("M", 25)], seq="", aux=[]))             # SEQ '*': l_seq = 0
    # long reads over several 16 kb windows and index bins, noise-like CIGARs, straddling BGZF members
    pos = 1_000
    k = 0
    while pos < 64_000:
        n_ops = rng.randint(3, 40)
        cig, qlen = [], 0
        for j in range(n_ops):
            m = rng.randint(1, 400)
            cig.append(("M", m))
            qlen += m
            if j + 1 < n_ops:
                if rng.random() < 0.5:
                    d = rng.choice([1, 1, 2, 3, 15, 16, 17, 40])
                    cig.append(("D", d))
                else:
                    i_ = rng.choice([1, 1, 2, 5, 16, 31])
                    cig.append(("I", i_))
                    qlen += i_
        if rng.random() < 0.3:
            cig = [("S", rng.randint(1, 30))] + cig
            qlen += cig[0][1]
        if rng.random() < 0.3:
            cig.append(("S", rng.randint(1, 30)))
            qlen += cig[-1][1]
        if pos + ref_span(cig) >= refs[0][1]:
            break
        a = []
        if k % 3:
            a += [("HP", "C" if k % 2 else "i", 1 + k % 2), ("PS", "I" if k % 4 else "i", 1000 * (1 + k // 8))]
        a.append(("NM", "C", k % 200))
        recs.append(dict(ref=0, name="long%d" % k, flag=16 if k % 5 == 0 else 0, pos0=pos, cigar=cig, seq=rand_seq(qlen, "ACGTN" if k % 7 == 0 else "ACGT"), aux=a))
        pos += rng.randint(50, 1_200)
        k += 1
    recs_a = sorted([r for r in recs], key=lambda r: r["pos0"])
    recs_b = [dict(ref=1, name="b%d" % i, flag=0, pos0=50 + 700 * i, cigar=[("M", 300), ("D", 3), ("M", 300)], seq=rand_seq(600), aux=[("HP", "C", 1 + i % 2), ("PS", "I", 7)])
              for i in range(10)]
    unplaced = [dict(ref=-1, name="nocoor%d" % i, flag=4, pos0=-1, cigar=[], seq=rand_seq(30), aux=[]) for i in range(3)]
    ordered = recs_a + recs_b + unplaced
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs) + "@PG\tID:make_spec_bam\n"
    header = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for name, ln in refs:
        header += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    # the uncompressed stream, cut into members at sizes that do NOT respect record boundaries
    stream = bytearray(header)
    starts = []                                        # uncompressed offset of every record
    for r in ordered:
        starts.append(len(stream))
        stream += record(r, r["ref"])
    cuts, o, hows = [0], 0, []
    sizes = [len(header) + 5, 777, 65_280, 1, 12_345, 40_000, 65_280, 3]
    si = 0
    while o < len(stream):
        o = min(len(stream), o + sizes[si % len(sizes)])
        cuts.append(o)
        hows.append(("stored", "fixed", "dynamic")[si % 3] if si < 6 else "dynamic")
        si += 1
    members, coff = [], [0]
    for a, b, how in zip(cuts[:-1], cuts[1:], hows):
        m = bgzf_member(bytes(stream[a:b]), how)
        members.append(m)
        coff.append(coff[-1] + len(m))
    with open(os.path.join(OUT, "spec.bam"), "wb") as f:
        f.write(b"".join(members) + EOF_BLOCK)

    def voffset(u):
        """virtual file offset (4.1.1) of uncompressed offset u: coffset of its member << 16 | offset inside the member"""
        import bisect
        if u >= cuts[-1]:
            return coff[-1] << 16                      # the end of the data: the EOF member
        k_ = bisect.bisect_right(cuts, u) - 1
        return coff[k_] << 16 | (u - cuts[k_])
    # ---- BAI (5.2)
    ends = starts[1:] + [len(stream)]
    bai = b"BAI\x01" + struct.pack("<i", len(refs))
    for tid, (name, ln) in enumerate(refs):
        bins, lin, n_map, n_unmap, v_lo, v_hi = {}, {}, 0, 0, None, None
        for r, s, e in zip(ordered, starts, ends):
            if r["ref"] != tid:
                continue
            beg = r["pos0"]
            end = beg + max(1, ref_span(r["cigar"]))
            vs, ve = voffset(s), voffset(e)
            bins.setdefault(reg2bin(beg, end), []).append([vs, ve])
            if r["flag"] & 4:
                n_unmap += 1
            else:
                n_map += 1
            for w in range(beg >> 14, ((end - 1) >> 14) + 1):
                lin[w] = min(lin.get(w, vs), vs)
            v_lo = vs if v_lo is None else min(v_lo, vs)
            v_hi = ve if v_hi is None else max(v_hi, ve)
        out = b""
        n_bin = 0
        for b_ in sorted(bins):
            merged = []
            for c in bins[b_]:                          # adjacent chunks of a bin are merged, as indexers do
                if merged and merged[-1][1] == c[0]:
                    merged[-1][1] = c[1]
                else:
                    merged.append(list(c))
            out += struct.pack("<Ii", b_, len(merged)) + b"".join(struct.pack("<QQ", *c) for c in merged)
            n_bin += 1
        if v_lo is not None:
            out += struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", v_lo, v_hi, n_map, n_unmap)
            n_bin += 1
        n_intv = (max(lin) + 1) if lin else 0
        lin_arr = [lin.get(w, 0) for w in range(n_intv)]
        bai += struct.pack("<i", n_bin) + out + struct.pack("<i", n_intv) + b"".join(struct.pack("<Q", v) for v in lin_arr)
    bai += struct.pack("<Q", len(unplaced))
    with open(os.path.join(OUT, "spec.bam.bai"), "wb") as f:
        f.write(bai)
    exp = dict(refs=refs, members=len(members), member_kinds=hows,
               records=[dict(ref=r["ref"], name=r["name"], flag=r["flag"], pos0=r["pos0"], cigar=r["cigar"], seq=r["seq"],
                             hp=next((a[2] for a in r.get("aux", ()) if a[0] == "HP"), 0), ps=next((a[2] for a in r.get("aux", ()) if a[0] == "PS"), 0))
                        for r in ordered])
    with open(os.path.join(OUT, "spec_expected.json"), "w") as f:
        json.dump(exp, f)
    print("spec.bam: %d bytes in %d BGZF members (%s), %d records" % (coff[-1] + len(EOF_BLOCK), len(members), ", ".join(sorted(set(hows))), len(ordered)))


if __name__ == "__main__":
    make()
