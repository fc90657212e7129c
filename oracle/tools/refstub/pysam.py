"""Stub `pysam` used ONLY in the build container to import and run the reference featurisers
(/root/reference/nanocaller_src/generate_SNP_pileups.py etc.) on synthetic worlds and capture
golden vectors.  It is test tooling: nothing here is shipped or imported
by the product path, and it never travels with reference code.

Semantics mirrored from pysam/htslib documentation [pysam-doc]: pileup() yields only columns
with depth > 0, applies `flag_filter`, a deletion is '*', the first character of each
get_query_sequences(add_indels=True) string is the base (upper = forward, lower = reverse
strand) and an indel that follows is appended as +<n><bases> / -<n><N...>.
"""
import numpy as np

WORLDS = {}          # path -> nanocaller_amd.synth.World
CAPTURE_INDEL = False    # True: the SECOND pileup() of a Samfile captures the caller's `variants` dict and yields nothing
CAPTURED = {}


def register(path, world):
    WORLDS[path] = world


_LET = "AGTC*"


class _Aln:
    def __init__(self, world, i):
        self.qname = world.names[i]
        self.query_name = self.qname
        self.flag = int(world.read_flag[i])
        self._tags = world.meta.get("tags", {}).get(i, {})

    def has_tag(self, t):
        return t in self._tags

    def get_tag(self, t):
        return self._tags[t]


class _Col:
    def __init__(self, pos0, names, seqs):
        self.pos = pos0
        self.reference_pos = pos0
        self._names = names
        self._seqs = seqs

    def get_query_sequences(self, mark_matches=False, mark_ends=False, add_indels=False):
        return list(self._seqs)

    def get_query_names(self):
        return list(self._names)

    def get_num_aligned(self):
        return len(self._names)


class RecordWorld:
    """SAM-like records (dict name, flag, pos0, cigar [(op, len)], seq, tags) in coordinate order: the second kind of
    world the stub serves.  Here the stub restates htslib's pileup engine from the SAM spec + pysam documentation
    [pysam-doc]: a column lists the reads whose alignment spans it in file order; inside M/=/X the entry is the read
    base (lower case on the reverse strand), inside D it is '*' with is_del set; the LAST aligned base before an I / D
    operation carries the indel (`+<n><bases>` / `-<n><N...>` with add_indels=True); soft clips stay in query_sequence and
    count in query positions; query_position_or_next of a deleted position is the index of the next aligned query base."""

    def __init__(self, chrom, length, ref, records):
        self.chrom, self.length, self.ref, self.records = chrom, length, ref, records
        self.names = [r["name"] for r in records]
        self.span = []
        for r in records:
            rlen = sum(ln for op, ln in r["cigar"] if op in "MDN=X")
            self.span.append((r["pos0"], r["pos0"] + rlen))
        self._exp = {}

    def expand(self, k):
        """per reference position of record k: (is_del, query index or next, pileup string)"""
        if k in self._exp:
            return self._exp[k]
        r = self.records[k]
        rev = bool(r["flag"] & 16)
        seq = r["seq"]
        out = []
        qp = 0
        cig = r["cigar"]
        for ci, (op, ln) in enumerate(cig):
            if op in "M=X":
                for t in range(ln):
                    ch = seq[qp].lower() if rev else seq[qp].upper()
                    suffix = ""
                    if t == ln - 1 and ci + 1 < len(cig):
                        nop, nln = cig[ci + 1]
                        if nop == "I":
                            ins = seq[qp + 1:qp + 1 + nln]
                            suffix = "+%d%s" % (nln, ins.lower() if rev else ins.upper())
                        elif nop == "D":
                            suffix = "-%d%s" % (nln, ("n" if rev else "N") * nln)
                    out.append((False, qp, ch + suffix))
                    qp += 1
            elif op in "IS":
                qp += ln
            elif op == "D":
                for _ in range(ln):
                    out.append((True, qp, "*"))
            elif op == "N":
                for _ in range(ln):
                    out.append((True, qp, "<" if rev else ">"))
            elif op in "HP":
                pass
            else:
                raise ValueError(op)
        self._exp[k] = out
        return out


def register_records(path, chrom, length, ref, records):
    WORLDS[path] = RecordWorld(chrom, length, ref, records)


class _RecAlignment:
    def __init__(self, rec):
        self.qname = self.query_name = rec["name"]
        self.flag = rec["flag"]
        self.query_sequence = rec["seq"]
        self._tags = rec.get("tags", {})

    def has_tag(self, t):
        return t in self._tags

    def get_tag(self, t):
        return self._tags[t]


class _PileupRead:
    def __init__(self, aln, is_del, qpos):
        self.alignment = aln
        self.is_del = int(is_del)
        self.is_refskip = 0
        self.query_position = None if is_del else qpos
        self.query_position_or_next = qpos


class _RecCol(_Col):
    def __init__(self, pos0, names, seqs, pileups):
        super().__init__(pos0, names, seqs)
        self.pileups = pileups


class Samfile:
    def __new__(cls, path, *a, **k):
        if isinstance(WORLDS.get(path), RecordWorld) and cls is Samfile:
            return object.__new__(_RecSamfile)
        return object.__new__(cls)

    def __init__(self, path, *a, reference_filename=None, **k):
        self.world = WORLDS[path]

    AlignmentFile = None

    def is_valid_reference_name(self, c):
        return c == self.world.chrom

    def get_reference_length(self, c):
        return self.world.length

    @property
    def references(self):
        return [self.world.chrom]

    def fetch(self, chrom, a, b, multiple_iterators=False):
        w = self.world
        # 0-based half-open [a,b); read covers 0-based [start-1, end-1)
        idx = np.nonzero((w.read_start - 1 < b) & (w.read_end - 1 > a))[0]
        for i in idx:
            yield _Aln(w, int(i))

    def pileup(self, chrom, a, b, min_base_quality=0, flag_filter=0, truncate=True,
               multiple_iterators=False, **k):
        self._n_pileup = getattr(self, "_n_pileup", 0) + 1
        if CAPTURE_INDEL and self._n_pileup == 2:
            import sys as _sys
            fl = _sys._getframe(1).f_locals
            CAPTURED["variants"] = dict(fl["variants"])
            CAPTURED["extra_variants"] = dict(fl["extra_variants"])
            return iter(())
        return self._pileup(chrom, a, b, flag_filter)

    def _pileup(self, chrom, a, b, flag_filter):
        w = self.world
        a = max(0, a)
        b = min(b, w.length)
        ok = (w.read_flag & flag_filter) == 0
        idx = np.nonzero(ok & (w.read_start - 1 < b) & (w.read_end - 1 > a))[0]
        starts = w.read_start[idx].astype(np.int64) - 1
        ends = w.read_end[idx].astype(np.int64) - 1
        rev = (w.read_flag[idx] & 16) != 0
        deco = w.meta.get("deco", {})
        for p0 in range(a, b):
            m = (starts <= p0) & (ends > p0)
            if not m.any():
                continue
            sel = idx[m]
            names, seqs = [], []
            for i, s0, rv in zip(sel, starts[m], rev[m]):
                c = int(w.codes[w.read_off[i] + (p0 - s0)])
                if "letters" in w.meta and (int(i), p0) in w.meta["letters"]:
                    ch = w.meta["letters"][(int(i), p0)]
                else:
                    ch = _LET[c]
                if rv and ch != "*":
                    ch = ch.lower()
                d = deco.get((int(i), p0))
                if d:
                    ch = ch + (d.lower() if rv else d)
                names.append(w.names[i])
                seqs.append(ch)
            yield _Col(p0, names, seqs)


class _RecSamfile(Samfile):
    def fetch(self, chrom, a, b, multiple_iterators=False):
        w = self.world
        for k, (s0, e0) in enumerate(w.span):
            if s0 < b and e0 > a:
                yield _RecAlignment(w.records[k])

    def _pileup(self, chrom, a, b, flag_filter):
        w = self.world
        a = max(0, a)
        b = min(b, w.length)
        live = [k for k, (s0, e0) in enumerate(w.span) if s0 < b and e0 > a and not (w.records[k]["flag"] & flag_filter)]
        alns = {k: _RecAlignment(w.records[k]) for k in live}
        for p0 in range(a, b):
            names, seqs, pil = [], [], []
            for k in live:
                s0, e0 = w.span[k]
                if s0 <= p0 < e0:
                    is_del, qp, text = w.expand(k)[p0 - s0]
                    names.append(w.names[k])
                    seqs.append(text)
                    pil.append(_PileupRead(alns[k], is_del, qp))
            if names:
                yield _RecCol(p0, names, seqs, pil)


AlignmentFile = Samfile


class FastaFile:
    def __init__(self, path, *a, **k):
        self.world = WORLDS[path]          # World or RecordWorld: both have .ref / .length

    def fetch(self, chrom, a=None, b=None):
        return self.world.ref[a:b]

    def get_reference_length(self, chrom):
        return self.world.length


class TabixFile:
    def __init__(self, path, *a, **k):
        self.rows = WORLDS[path]     # list of (chrom, start, end)

    def fetch(self, chrom, parser=None):
        rows = [r for r in self.rows if r[0] == chrom]
        if not rows:
            raise ValueError("could not create iterator for region")
        return iter(rows)


def asBed():
    return None


class VariantFile:
    def __init__(self, *a, **k):
        pass
