"""BGZF + CSI output without bgzip / tabix / bcftools (replaces the shell pipelines of snpCaller.py:284-285).

* `bgzf_write(path, data)`: BGZF file through the library's multi-threaded compressor (nc_bgzf_compress); returns the
  compressed offset of every block so that virtual file offsets can be formed.
* `write_csi(path, ...)`: coordinate-sorted index in the CSI v1 format with the tabix auxiliary block for VCF
  (`tabix -p vcf --csi`, min_shift 14, depth 5), following the CSIv1 / tabix specifications (hts-specs).  htslib itself is
  not in this image: the index is pinned by the reader in tests/ (region queries return exactly the overlapping records).
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import _lib

BGZF_BLOCK = 0xff00


def bgzf_compress(data, level=6):
    """-> (compressed bytes as uint8 array, block_coff int64 [n_blocks + 1])"""
    L = _lib.lib()
    buf = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
    n = buf.size
    nb = (n + BGZF_BLOCK - 1) // BGZF_BLOCK
    cap = n + n // 100 + 64 * nb + 128
    out = np.empty(cap, np.uint8)
    coff = np.empty(nb + 1, np.int64)
    n_out, n_blk = C.c_int64(), C.c_int64()
    rc = L.nc_bgzf_compress(_lib.npp(buf) if n else None, n, int(level), _lib.npp(out), cap, C.byref(n_out), _lib.npp(coff), nb + 1,
                            C.byref(n_blk))
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("nc_bgzf_compress failed (%d)" % rc)
    return out[:n_out.value], coff


def bgzf_write(path, data, level=6):
    """Write `data` as a BGZF file; -> block_coff (see bgzf_compress)."""
    comp, coff = bgzf_compress(data, level)
    with open(path, "wb") as f:
        f.write(comp)
    return coff


def bgzf_read(path):
    """The inflated bytes of a BGZF file (bgzip output), every block CRC-checked by the library's BAM-side BGZF reader."""
    L = _lib.lib()
    L.nc_bgzf_read_file.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    n = C.c_int64()
    rc = L.nc_bgzf_read_file(str(path).encode(), None, 0, C.byref(n))
    if rc not in (_lib.NC_OK, _lib.NC_ERR_CAPACITY):
        raise _lib.NanoCallerHipError("%s: not a readable BGZF file (%d)" % (path, rc))
    out = np.empty(max(n.value, 1), np.uint8)
    rc = L.nc_bgzf_read_file(str(path).encode(), _lib.npp(out), n.value, C.byref(n))
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("%s: not a readable BGZF file (%d)" % (path, rc))
    return out[:n.value].tobytes()


def virtual_offsets(block_coff, uoff):
    """virtual file offsets of uncompressed offsets `uoff` in a stream written by bgzf_write"""
    uoff = np.asarray(uoff, np.int64)
    return (block_coff[uoff // BGZF_BLOCK].astype(np.uint64) << np.uint64(16)) | (uoff % BGZF_BLOCK).astype(np.uint64)


def reg2bin(beg, end, min_shift=14, depth=5):
    """hts_reg2bin for arrays of 0-based half-open intervals"""
    beg = np.asarray(beg, np.int64)
    end = np.asarray(end, np.int64) - 1
    out = np.zeros(beg.shape, np.int64)
    done = np.zeros(beg.shape, bool)
    s, t = min_shift, ((1 << (depth * 3)) - 1) // 7
    for lv in range(depth, 0, -1):
        hit = ~done & ((beg >> s) == (end >> s))
        out[hit] = t + (beg[hit] >> s)
        done |= hit
        s += 3
        t -= 1 << ((lv - 1) * 3)
    return out


def csi_bytes(names, tid, beg0, end0, voff_beg, voff_end, min_shift=14, depth=5):
    """Uncompressed CSI index of records sorted by (tid, beg0): 0-based half-open [beg0, end0), virtual offsets of each
    record's first byte and of the byte after it.  tabix aux: format VCF (2), columns 1 / 2 / 0, meta '#', skip 0."""
    tid = np.asarray(tid, np.int64)
    beg0 = np.asarray(beg0, np.int64)
    end0 = np.asarray(end0, np.int64)
    vb = np.asarray(voff_beg, np.uint64)
    ve = np.asarray(voff_end, np.uint64)
    nm = b"".join(n.encode() + b"\0" for n in names)
    aux = struct.pack("<7i", 2, 1, 2, 0, ord("#"), 0, len(nm)) + nm
    out = [b"CSI\1", struct.pack("<3i", min_shift, depth, len(aux)), aux, struct.pack("<i", len(names))]
    meta_bin = ((1 << (depth * 3 + 3)) - 1) // 7 + 1
    bins = reg2bin(beg0, end0, min_shift, depth)
    for r in range(len(names)):
        sel = np.flatnonzero(tid == r)
        if sel.size == 0:
            out.append(struct.pack("<i", 0))
            continue
        b, e, ob, oe, bn = beg0[sel], end0[sel], vb[sel], ve[sel], bins[sel]
        # linear index over 2^min_shift windows: smallest offset of a record overlapping the window, back-filled
        n_lin = int((e.max() - 1) >> min_shift) + 1
        lin = np.full(n_lin + 1, np.iinfo(np.uint64).max, np.uint64)
        w0, w1 = b >> min_shift, (e - 1) >> min_shift
        for k in range(int((w1 - w0).max()) + 1):
            m = w0 + k <= w1
            np.minimum.at(lin, (w0 + k)[m], ob[m])
        for k in range(n_lin - 1, -1, -1):
            if lin[k] == np.iinfo(np.uint64).max:
                lin[k] = lin[k + 1]
        # bins: records of a bin in file order; adjacent records form one chunk
        order = np.argsort(bn, kind="stable")
        ub, first = np.unique(bn[order], return_index=True)
        blocks = [struct.pack("<i", len(ub) + 1)]
        for i, bid in enumerate(ub):
            idx = order[first[i]:first[i + 1] if i + 1 < len(ub) else len(order)]
            cb, ce = ob[idx], oe[idx]
            brk = np.flatnonzero(cb[1:] != ce[:-1]) + 1
            starts = np.concatenate([[0], brk])
            ends = np.concatenate([brk, [len(idx)]])
            # first position covered by this bin -> loffset from the linear index
            lv, t, bb = depth, ((1 << (depth * 3)) - 1) // 7, int(bid)
            while bb < t:
                lv -= 1
                t -= 1 << (lv * 3)
            win = ((bb - t) << (min_shift + 3 * (depth - lv))) >> min_shift
            loff = int(lin[win]) if win < n_lin else 0
            blocks.append(struct.pack("<IQi", bb, loff, len(starts)))
            blocks.append(b"".join(struct.pack("<QQ", int(cb[s]), int(ce[x - 1])) for s, x in zip(starts, ends)))
        blocks.append(struct.pack("<IQi", meta_bin, 0, 2) + struct.pack("<QQQQ", int(ob.min()), int(oe.max()), len(sel), 0))
        out.append(b"".join(blocks))
    out.append(struct.pack("<Q", 0))                          # n_no_coor
    return b"".join(out)


def write_vcf_gz_with_csi(path, header, lines_bytes, contigs, tid, pos1, ref_len, line_len):
    """Write header + records (one bytes object of all record lines, sorted by contig order then position) as BGZF and
    its .csi next to it.  tid / pos1 (1-based POS) / ref_len (len(REF)) / line_len (bytes incl. newline) per record."""
    hdr = header.encode() if isinstance(header, str) else header
    data = hdr + lines_bytes
    coff = bgzf_write(path, data)
    line_len = np.asarray(line_len, np.int64)
    ustart = len(hdr) + np.concatenate([[0], np.cumsum(line_len[:-1])]) if line_len.size else np.zeros(0, np.int64)
    vb = virtual_offsets(coff, ustart)
    # offset of the byte after a record = start of the next one; for the last record the end of the data, which may sit
    # exactly on a block boundary (= the EOF block's offset)
    uend = ustart + line_len
    ve = np.where(uend < len(data), virtual_offsets(coff, np.minimum(uend, max(len(data) - 1, 0))), np.uint64(int(coff[-1]) << 16))
    pos1 = np.asarray(pos1, np.int64)
    idx = csi_bytes(contigs, tid, pos1 - 1, pos1 - 1 + np.asarray(ref_len, np.int64), vb, ve)
    bgzf_write(path + ".csi", idx)
    return coff


def read_vcf_gz(path):
    """-> (header lines, record lines) of a (BGZF-)gzipped or plain VCF; lines keep their newline"""
    import gzip
    op = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    hdr, recs = [], []
    with op(path, "rt") as f:
        for ln in f:
            (hdr if ln.startswith("#") else recs).append(ln)
    return hdr, recs


def write_sorted_vcf(path, header, lines, contigs):
    """`bcftools sort | bgzip && tabix -p vcf --csi` of text records: stable sort by (contig order, POS), BGZF + .csi.
    -> number of records.  Contigs not in `contigs` are appended in order of first appearance."""
    contigs = list(contigs)
    order = {c: i for i, c in enumerate(contigs)}
    keyed = []
    for ln in lines:
        f = ln.split("\t", 4)
        if f[0] not in order:
            order[f[0]] = len(contigs)
            contigs.append(f[0])
        keyed.append((order[f[0]], int(f[1]), len(f[3]), ln))
    keyed.sort(key=lambda k: (k[0], k[1]))
    body = "".join(k[3] for k in keyed).encode()
    write_vcf_gz_with_csi(path, header, body, contigs, np.array([k[0] for k in keyed], np.int64), np.array([k[1] for k in keyed], np.int64),
                          np.array([k[2] for k in keyed], np.int64), np.array([len(k[3].encode()) for k in keyed], np.int64))
    return len(keyed)
