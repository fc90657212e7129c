"""Test tooling: a minimal BAM / BAI / FASTA WRITER following the SAM/BAM specification (SAMv1 sections 4.2, 5.1.3,
5.2), used to turn synthetic worlds into real files for the native reader (nanocaller_amd/csrc/nc_bam.cpp)."""
import struct
import zlib

import numpy as np

_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
_NT16_LUT = np.full(256, 15, np.uint8)
for _c, _i in _NT16.items():
    _NT16_LUT[ord(_c)] = _i


def reg2bin(beg, end):          # SAMv1 5.3, 0-based half-open
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


class BgzfWriter:
    def __init__(self, path, block=0xff00, level=6):
        self.level = level
        self.f = open(path, "wb")
        self.buf = bytearray()
        self.coff = 0
        self.block = block

    def tell(self):
        return (self.coff << 16) | len(self.buf)

    def write(self, data):
        i = 0
        while i < len(data):
            k = min(len(data) - i, self.block - len(self.buf))
            self.buf += data[i:i + k]
            i += k
            if len(self.buf) >= self.block:
                self.flush()

    def flush(self):
        if not self.buf:
            return
        raw = bytes(self.buf)
        co = zlib.compressobj(self.level, zlib.DEFLATED, -15)
        comp = co.compress(raw) + co.flush()
        blk = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, len(comp) + 25) + comp + \
            struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw))
        self.f.write(blk)
        self.coff += len(blk)
        self.buf = bytearray()

    def close(self):
        self.flush()
        self.f.write(_BGZF_EOF)
        self.f.close()


def write_bam(path, chrom, length, records, other_refs=(), write_bai=True, write_csi=False, level=6):
    """records: list of dict(name, flag, pos0, cigar=[(op,len)...] with op in 'MIDNSHP=X', seq (str), tags={'HP':1,...},
    optional tid = index into [(chrom, length)] + other_refs, default 0; optional qual = bytes of len(seq), default 0xff: absent) in coordinate order."""
    refs = [(chrom, length)] + list(other_refs)
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    w = BgzfWriter(path, level=level)
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for n, ln in refs:
        hdr += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", ln)
    w.write(hdr)
    w.flush()
    lins = [dict() for _ in refs]
    spans = [[] for _ in refs]                       # (beg0, end0, voff_beg, voff_end) of every mapped record, for the CSI
    ops = "MIDNSHP=X"
    for r in records:
        lin = lins[r.get("tid", 0)]
        cig = r["cigar"]
        rlen = sum(ln for op, ln in cig if op in "MDN=X")
        name = r["name"].encode() + b"\0"
        seq = r["seq"]
        nib = _NT16_LUT[np.frombuffer(seq.encode("ascii"), np.uint8)]
        if nib.size & 1:
            nib = np.append(nib, np.uint8(0))
        packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8).tobytes()
        tags = b""
        for k, v in r.get("tags", {}).items():
            if isinstance(v, (list, tuple)):                 # B,I array (e.g. the CG tag)
                tags += k.encode() + b"BI" + struct.pack("<I", len(v)) + b"".join(struct.pack("<I", x) for x in v)
            elif isinstance(v, str):
                tags += k.encode() + b"Z" + v.encode() + b"\0"
            elif 0 <= v < 256:
                tags += k.encode() + b"C" + struct.pack("<B", v)
            else:
                tags += k.encode() + b"i" + struct.pack("<i", v)
        body = struct.pack("<iiBBHHHiiii", r.get("tid", 0), r["pos0"], len(name), 60, reg2bin(r["pos0"], r["pos0"] + max(1, rlen)), len(cig),
                           r["flag"], len(seq), -1, -1, 0) + name + b"".join(struct.pack("<I", (ln << 4) | ops.index(op)) for op, ln in cig) + \
            bytes(packed) + (r["qual"] if r.get("qual") is not None else b"\xff" * len(seq)) + tags
        voff = w.tell()
        if not (r["flag"] & 4):
            for win in range(r["pos0"] >> 14, ((r["pos0"] + max(1, rlen) - 1) >> 14) + 1):
                lin.setdefault(win, voff)
        w.write(struct.pack("<i", len(body)) + body)
        if not (r["flag"] & 4):
            spans[r.get("tid", 0)].append((r["pos0"], r["pos0"] + max(1, rlen), voff, w.tell()))
    w.close()
    if write_csi:
        write_bam_csi(path + ".csi", spans)
    if write_bai:
        with open(path + ".bai", "wb") as f:
            f.write(b"BAI\1" + struct.pack("<i", len(refs)))
            for lin in lins:
                n_intv = (max(lin) + 1) if lin else 0
                arr = [0] * n_intv
                last = 0
                for k in range(n_intv):           # samtools fills empty windows with the previous offset
                    if k in lin:
                        last = lin[k]
                    arr[k] = last
                f.write(struct.pack("<i", 0) + struct.pack("<i", n_intv) + b"".join(struct.pack("<Q", v) for v in arr))


def write_bam_csi(path, spans, min_shift=14, depth=5):
    """CSIv1 index of a BAM (hts-specs CSIv1: no auxiliary data; per reference the bins with their chunks and `loffset` =
    linear-index value of the bin's first window, empty windows back-filled from the next one as htslib does), BGZF-compressed
    like the files `samtools index -c` writes"""
    def reg2bin(beg, end):
        end -= 1
        s, t = min_shift, ((1 << (depth * 3)) - 1) // 7
        for lv in range(depth, 0, -1):
            if beg >> s == end >> s:
                return t + (beg >> s)
            s += 3
            t -= 1 << ((lv - 1) * 3)
        return 0

    def bin_start_window(b):
        lv, t = 0, 0
        while b >= t + (1 << (3 * lv)):
            t += 1 << (3 * lv)
            lv += 1
        return ((b - t) << (3 * (depth - lv)))           # index of the bin's first 2^min_shift window
    out = [b"CSI\1", struct.pack("<3i", min_shift, depth, 0), struct.pack("<i", len(spans))]
    for recs in spans:
        if not recs:
            out.append(struct.pack("<i", 0))
            continue
        n_win = ((max(e for _, e, _, _ in recs) - 1) >> min_shift) + 1
        lin = [None] * (n_win + 1)
        bins = {}
        for b0, e0, vb, ve in recs:
            for wdx in range(b0 >> min_shift, ((e0 - 1) >> min_shift) + 1):
                if lin[wdx] is None or vb < lin[wdx]:
                    lin[wdx] = vb
            ch = bins.setdefault(reg2bin(b0, e0), [])
            if ch and ch[-1][1] == vb:
                ch[-1][1] = ve                               # adjacent records form one chunk
            else:
                ch.append([vb, ve])
        for k in range(n_win - 1, -1, -1):
            if lin[k] is None:
                lin[k] = lin[k + 1]
        blob = [struct.pack("<i", len(bins))]
        for b in sorted(bins):
            wdx = bin_start_window(b)
            loff = lin[wdx] if wdx < n_win and lin[wdx] is not None else 0
            blob.append(struct.pack("<IQi", b, loff, len(bins[b])) + b"".join(struct.pack("<QQ", u, v) for u, v in bins[b]))
        out.append(b"".join(blob))
    out.append(struct.pack("<Q", 0))
    w = BgzfWriter(path)
    w.write(b"".join(out))
    w.close()


def write_fasta(path, chrom, seq, width=60, with_fai=True, extra=()):
    off = {}
    with open(path, "w") as f:
        for name, s in [(chrom, seq)] + list(extra):
            f.write(">%s some description\n" % name)
            off[name] = (f.tell(), len(s))
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")
    if with_fai:
        with open(path + ".fai", "w") as f:
            for name, (o, ln) in off.items():
                f.write("%s\t%d\t%d\t%d\t%d\n" % (name, ln, o, width, width + 1))


def world_to_records(world, rng):
    """Alignment records equivalent to a world: explicit insertion / deletion events become I / D operations (deleted
    positions must already be code 4 in the world), other code-4 positions are written as base N, reverse-strand /
    filtered reads keep their flags, a soft clip is added at both ends."""
    lut = np.frombuffer(b"AGTCNNNN", np.uint8)
    ev_off, ev_pos, ev_len = world.meta["events"]
    ins_off = ins_bases = None
    if "ev_ins" in world.meta:                                                   # inserted bases given by the world
        ins_off, ins_bases = world.meta["ev_ins"]
    recs = []
    for r in range(world.n_reads):
        s, e = int(world.read_start[r]), int(world.read_end[r])
        letters = lut[world.read_codes(r)].tobytes().decode("ascii")
        cig, seq = [("S", 3)], ["ACG"]
        p = s                                                                    # next reference position to emit
        for k in range(int(ev_off[r]), int(ev_off[r + 1])):
            ep, ev = int(ev_pos[k]), int(ev_len[k])
            if ep < p or ep >= e:
                continue
            cig.append(("M", ep - p + 1))
            seq.append(letters[p - s:ep - s + 1])
            p = ep + 1
            if ev > 0:
                cig.append(("I", ev))
                if ins_bases is not None:
                    seq.append(bytes(ins_bases[ins_off[k]:ins_off[k + 1]]).decode())
                else:
                    seq.append("".join("AGTC"[i] for i in rng.integers(0, 4, size=ev)))
            else:
                cig.append(("D", -ev))
                p += -ev
        if p < e:
            cig.append(("M", e - p))
            seq.append(letters[p - s:e - s])
        cig.append(("S", 2))
        seq.append("TT")
        tags = {}
        if world.meta["hap"][r]:
            tags = {"HP": int(world.meta["hap"][r]), "PS": int(world.meta["ps"][r])}
        recs.append(dict(name=world.names[r], flag=int(world.read_flag[r]), pos0=s - 1, cigar=cig, seq="".join(seq), tags=tags))
    return recs


def make_bam_world(seed=5, length=30_000, depth=12):
    """A world whose codes are consistent with its indel events (deleted positions are code 4; events never sit on the
    last positions of a read, never overlap, never start inside a deletion)."""
    from nanocaller_amd.synth import add_indels, make_world
    w = add_indels(make_world(seed=seed, length=length, depth=depth, read_len_scale=0.15, odd_flag_frac=0.05), seed=seed,
                   het_rate=1 / 400.0, noise_rate=0.003)
    ev_off, ev_pos, ev_len = w.meta["events"]
    new_off, new_pos, new_len = [0], [], []
    codes = w.codes.copy()
    for r in range(w.n_reads):
        s, e = int(w.read_start[r]), int(w.read_end[r])
        busy_until = s
        for k in range(ev_off[r], ev_off[r + 1]):
            p, ln = int(ev_pos[k]), int(ev_len[k])
            need = -ln if ln < 0 else 0
            if p < busy_until or p + need + 2 >= e or p <= s:
                continue
            if ln < 0:
                codes[w.read_off[r] + (p + 1 - s):w.read_off[r] + (p + 1 - s) + need] = 4
            new_pos.append(p)
            new_len.append(ln)
            busy_until = p + need + 1
        new_off.append(len(new_pos))
    w.codes = codes
    w.meta["events"] = (np.array(new_off, np.int32), np.array(new_pos, np.int32), np.array(new_len, np.int32))
    return w


def clean_noise_deletions(w):
    """The synthetic worlds code per-base deletion noise (and read 'N') as 4 without an event; written as 'N' bases they make
    msa() raise KeyError exactly as the reference does (generate_indel_pileups.py:56).  Give those positions the reference
    base instead (in place), so that every code 4 left belongs to a deletion event."""
    ev_off, ev_pos, ev_len = w.meta["events"]
    refc = np.array(["AGTC".find(c) for c in w.ref], np.int64)
    for r in range(w.n_reads):
        s0, o0 = int(w.read_start[r]), int(w.read_off[r])
        span = w.codes[o0:int(w.read_off[r + 1])]
        deleted = np.zeros(len(span), bool)
        for k in range(ev_off[r], ev_off[r + 1]):
            if ev_len[k] < 0:
                deleted[ev_pos[k] + 1 - s0:ev_pos[k] + 1 - s0 - ev_len[k]] = True
        fix = (span == 4) & ~deleted
        rc = refc[s0 - 1:s0 - 1 + len(span)][fix]
        span[fix] = np.where(rc >= 0, rc, 0)
    return w


def make_pass2_world(seed, length, depth, blocks=(), drop=0.85, alt_base_frac=0.3):
    """make_bam_world + clean_noise_deletions + per-event inserted bases (and, with `blocks`, unphased stretches):
    the input of the reference-executed pass-2 goldens (oracle/tools/make_goldens.py pass2) and of their tests"""
    from nanocaller_amd.synth import unphase_blocks
    w = clean_noise_deletions(make_bam_world(seed=seed, length=length, depth=depth))
    return unphase_blocks(w, list(blocks), seed=seed, drop=drop if blocks else 0.0, alt_base_frac=alt_base_frac)


def world_arrays(w, prefix):
    """the arrays that define a pass-2 world, for np.savez"""
    ev_off, ev_pos, ev_len = w.meta["events"]
    ins_off, ins_bases = w.meta["ev_ins"]
    return {prefix + k: v for k, v in dict(
        chrom=np.array(w.chrom), ref=np.frombuffer(w.ref.encode(), dtype=np.uint8), read_start=w.read_start, read_end=w.read_end,
        read_flag=w.read_flag, read_off=w.read_off, codes=w.codes, ev_off=ev_off, ev_pos=ev_pos, ev_len=ev_len,
        hap=np.asarray(w.meta["hap"], np.uint8), ps=np.asarray(w.meta["ps"], np.int32), ins_off=ins_off, ins_bases=ins_bases).items()}


def world_from_arrays(z, prefix):
    from nanocaller_amd.synth import World, apply_impute_inputs
    g = lambda k: z[prefix + k]
    R = g("read_start").shape[0]
    w = World(chrom=str(g("chrom")), ref=g("ref").tobytes().decode(), read_start=g("read_start"), read_end=g("read_end"),
              read_flag=g("read_flag"), read_off=g("read_off"), codes=g("codes"), names=["r%07d" % i for i in range(R)])
    w.meta.update(events=(g("ev_off"), g("ev_pos"), g("ev_len")), hap=g("hap"), ps=g("ps"))
    return apply_impute_inputs(w, g("hap"), g("ins_off"), g("ins_bases"))
