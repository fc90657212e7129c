"""BAM / FASTA ingest for the hot path: decoded alignments without pysam.

`read_bam(path, chrom, start, end)` uses the library's native BGZF/BAM reader (nc_bam_*) and returns a
`synth.World` -- the same decoded-alignment object the packer, the kernels and the tests use -- so
`dct['sam_path']` may be a BAM path wherever a World is accepted.  `read_fasta` is a plain (.fai-aware) reader.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .synth import World


def usable_cpus():
    """CPUs this process may really use: os.cpu_count() capped by the affinity mask and the cgroup CPU quota (a container
    that sees 256 CPUs under a quota of 16 runs 32 busy threads slower than 16)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def rank_threads():
    """host threads one rank of a multi-rank job should start: the CPUs it may use (numa.bind_rank narrows the affinity mask to the rank's
    share of its GPU's NUMA node), and no more than its share of a cgroup quota the ranks of the node draw from together"""
    n = usable_cpus()
    try:
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    except ValueError:
        lw = 1
    if lw > 1:
        quota = None
        try:
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = max(1, -(-int(q) // int(p)))
        except (OSError, ValueError):
            pass
        if quota is not None:
            n = min(n, max(1, quota // lw))
    return max(1, n)


class _Owner:
    """keeps a native result alive for as long as a numpy view of its arrays exists (views are zero-copy: a fresh copy of
    a gigabyte array costs more in page faults than the decode itself)"""

    def __init__(self, lib, handle, free):
        self.lib, self.handle, self.free = lib, handle, free

    def __del__(self):
        try:
            getattr(self.lib, self.free)(self.handle)
        except Exception:
            pass


def _arr(ptr, n, dtype, owner=None):
    """numpy array over native memory: a view that keeps `owner` alive, or (owner None) a copy"""
    if not n or not ptr:
        return np.zeros(0, dtype)
    buf = (C.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(ptr)
    if owner is None:
        return np.frombuffer(buf, dtype=dtype).copy()
    buf._nc_owner = owner                                # the ctypes array is the numpy array's base
    return np.frombuffer(buf, dtype=dtype)


class BamFile:
    def __init__(self, path):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        if self.L.nc_bam_open(os.fsencode(path), C.byref(self.h)) != _lib.NC_OK:
            raise IOError("cannot open %s as a BAM file" % path)
        n, idx = C.c_int32(), C.c_int32()
        self.L.nc_bam_n_refs(self.h, C.byref(n), C.byref(idx))
        self.has_index = bool(idx.value)
        self.references, self.lengths = [], []
        for i in range(n.value):
            nm, ln = C.c_char_p(), C.c_int32()
            self.L.nc_bam_ref(self.h, i, C.byref(nm), C.byref(ln))
            self.references.append(nm.value.decode())
            self.lengths.append(ln.value)

    def set_threads(self, n):
        """host threads inflating BGZF blocks for this handle (0 = all cores up to 32)"""
        self.L.nc_bam_set_threads(self.h, int(n))

    def close(self):
        if self.h:
            self.L.nc_bam_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # pysam.Samfile look-alikes used by utils.get_regions_list (utils.py:9-48)
    def is_valid_reference_name(self, c):
        return c in self.references

    def get_reference_length(self, c):
        return self.lengths[self.references.index(c)]

    def decode(self, chrom, start=1, end=None, keep_seq=False, anchors=None, window_before=0, window_after=160, keep_mask=None):
        """-> dict of numpy arrays for the mapped alignments overlapping [start, end] (1-based inclusive).  With `anchors`
        (ascending 1-based positions; implies keep_seq) also 'windows': the pass-2 read windows at those columns, reads whose
        flag has a bit of `keep_mask` set being left out (the pileup flag filter)."""
        if anchors is not None:
            keep_seq = True
        tid = self.references.index(chrom)
        end = self.lengths[tid] if end is None else min(int(end), self.lengths[tid])
        d = C.c_void_p()
        rc = self.L.nc_bam_decode(self.h, tid, max(1, int(start)), max(int(start), end), 1 if keep_seq else 0, C.byref(d))
        if rc != _lib.NC_OK:
            raise IOError("BAM decode failed: %s" % self.L.nc_bam_error(self.h).decode())
        out = _decoded_dict(self.L, d)
        if anchors is not None:
            out["windows"] = self._windows(d, out, anchors, window_before, window_after, keep_mask)
        return out

    def _windows(self, d, dec, anchors, window_before, window_after, keep_mask):
        """a11 (generate_indel_pileups.py:329-338): per anchor column the list [(read index, window string)] of the reads in
        the pileup there, in coordinate order, through nc_indel_slices."""
        anchors = np.ascontiguousarray(anchors, np.int32)
        keep = None
        if keep_mask is not None:
            keep = np.ascontiguousarray((dec["read_flag"] & int(keep_mask)) == 0, np.uint8)
        sl = C.c_void_p()
        rc = self.L.nc_indel_slices(d, len(anchors), _lib.npp(anchors), int(window_before), int(window_after), _lib.npp(keep), C.byref(sl))
        if rc != _lib.NC_OK:
            raise IOError("nc_indel_slices failed (%d)" % rc)
        v = _lib.SlicesArraysC()
        self.L.nc_slices_view(sl, C.byref(v))
        a_off = _arr(v.anchor_off, v.n_anchor + 1, np.int32)
        ridx = _arr(v.read_idx, v.n_slices, np.int32)
        s_off = _arr(v.seq_off, v.n_slices + 1, np.int64)
        text = _arr(v.seq, int(s_off[-1]) if v.n_slices else 0, np.uint8)
        self.L.nc_slices_free(sl)
        letters = np.frombuffer(b"AGTCN", np.uint8)[text].tobytes().decode("ascii")
        out = []
        for a in range(len(anchors)):
            out.append([(int(ridx[k]), letters[s_off[k]:s_off[k + 1]]) for k in range(a_off[a], a_off[a + 1])])
        return out


class _Names:
    """the read names of a decoded set as a read-only sequence of str, decoded when asked for (a chromosome has ~10^5-10^6 alignments and
    most runs never look at a name: building the list cost 2 ms per 3 Mb contig on the ingest thread, with the GIL held)"""

    def __init__(self, raw, off):
        self.raw, self.off = raw, off

    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self.raw[self.off[i]:self.off[i + 1] - 1].decode()

    def __iter__(self):
        return (self[k] for k in range(len(self)))

    def __eq__(self, other):
        return list(self) == list(other)

    def index(self, name):
        for k in range(len(self)):
            if self[k] == name:
                return k
        raise ValueError(name)

    def __contains__(self, name):
        return any(self[k] == name for k in range(len(self)))


def _decoded_dict(L, d):
    """dict of zero-copy numpy views over a native nc_decoded (freed when the last view goes)"""
    own = _Owner(L, d, "nc_decoded_free")
    v = _lib.DecodedArraysC()
    L.nc_decoded_view(d, C.byref(v))
    n = v.n_reads
    a = lambda ptr, cnt, dt: _arr(ptr, cnt, dt, own)     # noqa: E731
    out = dict(read_start=a(v.start, n, np.int32), read_end=a(v.end, n, np.int32), read_flag=a(v.flag, n, np.int32),
               read_off=a(v.off, n + 1, np.int64), codes=a(v.codes, v.n_codes, np.uint8),
               ev_off=a(v.ev_off, n + 1, np.int32), ev_pos=a(v.ev_pos, v.n_events, np.int32),
               ev_len=a(v.ev_len, v.n_events, np.int32), hap=a(v.hap, n, np.uint8), ps=a(v.ps, n, np.int32),
               seq_off=a(v.seq_off, n + 1, np.int64), seq=a(v.seq, v.n_seq, np.uint8),
               qstart=a(v.qstart, n, np.int32))
    if n == 0:                                           # empty arrays are copies: keep the (n + 1)-offset convention
        for k in ("read_off", "seq_off"):
            out[k] = np.zeros(1, np.int64)
        out["ev_off"] = np.zeros(1, np.int32)
    name_off = _arr(v.name_off, n + 1, np.int32)
    out["names"] = _Names(_arr(v.names, int(name_off[-1]) if n else 0, np.uint8).tobytes(), np.array(name_off, np.int64))
    out["_owner"] = own
    return out


def read_fasta(path, chrom):
    """Whole contig as a string (case preserved: soft-masked bases matter, quirk E4).  Uses <path>.fai if present."""
    return read_fasta_bytes(path, chrom).decode("ascii")


def read_fasta_bytes(path, chrom):
    """read_fasta without the decode: the contig's letters as bytes"""
    fai = path + ".fai"
    if os.path.exists(fai):
        for line in open(fai):
            f = line.rstrip("\n").split("\t")
            if f[0] == chrom:
                length, offset, lb, lw = int(f[1]), int(f[2]), int(f[3]), int(f[4])
                with open(path, "rb") as fh:
                    fh.seek(offset)
                    raw = fh.read(length + (length // lb + 1) * (lw - lb))
                return raw.replace(b"\n", b"").replace(b"\r", b"")[:length]
        raise KeyError(chrom)
    seq, on = [], False
    for line in open(path):
        if line.startswith(">"):
            if on:
                break
            on = line[1:].split()[0] == chrom
        elif on:
            seq.append(line.strip())
    if not seq:
        raise KeyError(chrom)
    return "".join(seq).encode("ascii")


def decode_parallel(bam_path, chrom, start=1, end=None, keep_seq=False, threads=None, min_region=125_000):
    """BamFile.decode of a long interval as parallel regions (nc_bam_decode_regions): each host thread opens its own handle,
    seeks through the .bai linear index and decodes the alignments that START in its region (the first region also takes
    those that merely overlap its left edge); inflate, CIGAR walk, tag parsing and the merge into one set of arrays all run
    on native threads.  Same result as one sequential decode; the arrays are zero-copy views of the native result."""
    bf = BamFile(bam_path)
    tid_len = bf.get_reference_length(chrom)
    has_index = bf.has_index
    bf.close()
    end = tid_len if end is None else min(int(end), tid_len)
    start = max(1, int(start))
    threads = threads or min(64, rank_threads())
    min_region = int(os.environ.get("NANOCALLER_DECODE_MIN_REGION", min_region))
    n_reg = min(threads, max(1, (end - start + 1) // min_region))
    if n_reg <= 1 or not has_index:
        bf = BamFile(bam_path)
        d = bf.decode(chrom, start, end, keep_seq)
        bf.close()
        return d
    L = _lib.lib()
    bf = BamFile(bam_path)
    tid = bf.references.index(chrom)
    bf.close()
    d = C.c_void_p()
    rc = L.nc_bam_decode_regions(os.fsencode(bam_path), tid, start, end, 1 if keep_seq else 0, n_reg, C.byref(d))
    if rc != _lib.NC_OK:
        raise IOError("BAM decode failed (%d): %s" % (rc, bam_path))
    return _decoded_dict(L, d)


def _decode_parallel_py(bam_path, chrom, start, end, keep_seq, threads, n_reg):
    """the same with Python threads around nc_bam_decode (kept as the independent statement the native merge is tested against)"""
    from concurrent.futures import ThreadPoolExecutor
    edges = [start + (end - start + 1) * k // n_reg for k in range(n_reg)] + [end + 1]

    def one(k):
        b = BamFile(bam_path)
        b.set_threads(max(1, threads // n_reg))              # the regions are the parallelism: do not oversubscribe
        d = b.decode(chrom, edges[k], edges[k + 1] - 1, keep_seq)
        b.close()
        if k == 0:
            return d
        cut = int(np.searchsorted(d["read_start"], edges[k], side="left"))       # reads starting before the region belong to an earlier one
        out = {}
        for key in ("read_start", "read_end", "read_flag", "hap", "ps", "qstart"):
            out[key] = d[key][cut:]
        for off_key, data_key in (("read_off", "codes"), ("ev_off", None), ("seq_off", "seq")):
            o = d[off_key]
            out[off_key] = o[cut:] - o[cut]
            if data_key:
                out[data_key] = d[data_key][o[cut]:]
        e0 = d["ev_off"][cut]
        out["ev_pos"], out["ev_len"] = d["ev_pos"][e0:], d["ev_len"][e0:]
        out["names"] = d["names"][cut:]
        return out

    with ThreadPoolExecutor(n_reg) as ex:
        parts = list(ex.map(one, range(n_reg)))
    out = {}
    for key in ("read_start", "read_end", "read_flag", "hap", "ps", "qstart", "codes", "ev_pos", "ev_len", "seq"):
        out[key] = np.concatenate([p[key] for p in parts])
    for off_key, data_key in (("read_off", "codes"), ("ev_off", "ev_pos"), ("seq_off", "seq")):
        offs, base = [np.zeros(1, parts[0][off_key].dtype)], 0
        for p in parts:
            offs.append(p[off_key][1:] + base)
            base += int(p[off_key][-1])
        out[off_key] = np.concatenate(offs).astype(parts[0][off_key].dtype)
    out["names"] = [n for p in parts for n in p["names"]]
    return out


def read_bam(bam_path, fasta_path, chrom, start=1, end=None, keep_seq=False, threads=None) -> World:
    """Decoded alignments of `chrom` overlapping [start, end] + the contig's reference sequence (long intervals are
    decoded as parallel regions, see decode_parallel)."""
    d = decode_parallel(bam_path, chrom, start, end, keep_seq, threads)
    ref = read_fasta(fasta_path, chrom)
    w = World(chrom=chrom, ref=ref, read_start=d["read_start"], read_end=d["read_end"], read_flag=d["read_flag"],
              read_off=d["read_off"], codes=d["codes"], names=d["names"])
    w.meta.update(events=(d["ev_off"], d["ev_pos"], d["ev_len"]), hap=d["hap"], ps=d["ps"])
    if keep_seq:
        w.meta.update(seq_off=d["seq_off"], seq=d["seq"])
    # inputs the library does not reproduce (nc_decoded_check): counted here, under either flag filter; the pack builders refuse them
    w.meta["unsupported"] = unsupported_counts(d)
    w.meta["name_gid"] = name_gids(d)
    return w


def unsupported_counts(d):
    """{supplementary flag: (kept alignments with a reference skip, pairs of kept alignments with one read name that overlap on the reference)}
    of a decode, under either flag filter: nc_decoded_check on the decode's native handle, or -- a decode assembled from several regions has
    none -- the same two counts from the arrays"""
    counts = {}
    owner = d.get("_owner")
    for supp in (False, True):
        keep = np.ascontiguousarray(((d["read_flag"] & (0x704 if supp else 0xF04)) == 0).astype(np.uint8))
        if owner is not None:
            a, b = C.c_int64(), C.c_int64()
            _lib.lib().nc_decoded_check(owner.handle, _lib.npp(keep), C.byref(a), C.byref(b))
            counts[supp] = (int(a.value), int(b.value))
        else:
            n_skip = int(np.count_nonzero(keep.astype(bool) & ((d["read_flag"] & _lib.FLAG_REFSKIP) != 0)))
            counts[supp] = (n_skip, same_name_overlaps(d["names"], d["read_start"], d["read_end"], keep))
    return counts


def name_gids(d):
    """{supplementary flag: gid int32 [n] or None}: per alignment the first kept alignment of its read name when the name is shared among the
    kept ones, else -1 (nc_decoded_name_groups; None = no shared name, or a decode without a native handle: pack.name_groups then walks the
    names).  What keys a split read's records by NAME as the reference's pileup dicts do (generate_SNP_pileups.py:175,185)."""
    owner = d.get("_owner")
    out = {}
    for supp in (False, True):
        if owner is None:
            out[supp] = None
            continue
        keep = np.ascontiguousarray(((d["read_flag"] & (0x704 if supp else 0xF04)) == 0).astype(np.uint8))
        gid = np.empty(max(1, keep.size), np.int32)
        ns = C.c_int64()
        rc = _lib.lib().nc_decoded_name_groups(owner.handle, _lib.npp(keep), _lib.npp(gid), C.byref(ns))
        if rc != _lib.NC_OK:
            raise _lib.NanoCallerHipError("nc_decoded_name_groups failed (%d)" % rc)
        out[supp] = gid[:keep.size] if ns.value else np.full(keep.size, -1, np.int32)
    return out


def same_name_overlaps(names, start, end, keep):
    """pairs of kept alignments that carry one read name and overlap on the reference (the reference's per-column dicts are keyed by name and
    hold one of them, generate_SNP_pileups.py:175,185,208): only supplementary alignments kept by --supplementary can produce them"""
    last, n = {}, 0
    for r in np.nonzero(np.asarray(keep))[0].tolist():             # coordinate order: starts ascend
        nm = names[r]
        e = last.get(nm)
        if e is not None and int(start[r]) < e:
            n += 1
        last[nm] = max(int(end[r]), e or 0)
    return n
