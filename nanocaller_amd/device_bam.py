"""BAM -> read pack on the device (csrc/nc_inflate.hip, csrc/nc_ingest.hip).

The reference reads alignments through pysam / htslib (generate_SNP_pileups.py:134-164).  The host route of this package inflates and
decodes them on worker threads (bam.py, nc_bam.cpp), diffs them against the reference and sends the difference over PCIe (wire.py).  Here
the FILE crosses PCIe as it is -- a BAM is already a compressed form of the pack -- and everything else happens in HBM:

  file image (page-locked) --H2D--> BGZF payloads --nc_inflate_device--> record stream --nc_bam_walk (.bai entries as chain starts)-->
  record offsets --nc_bam_meta--> per-record fields --D2H (48 B per record)--> [host: which reads are kept, tile index] --H2D-->
  nc_bam_codes --> the position-addressed codes of the pack

Which reads the pileup keeps (flag filter, htslib's depth cap, the unsupported-input checks) and the tile index are decided on the host
from the per-record fields with the same functions the host route uses, so a `DevicePack` made here is byte for byte the one
`wire.upload_wire(build_wire_from_world(bam.read_bam(...)))` makes (tests/test_device_bam.py); `pack(indel=True)` adds the indel path's sections.
Needs an index beside the file: a .bai's linear index, or a .csi's bin offsets and chunk begins, cut the record chain into independent walks.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib
from .engine import DevicePack, get_engine
from .pack import pileup_depth_cap
from .synth import FLAG_FILTER_DEFAULT, FLAG_FILTER_SUPPL

META_COLS = 12
(M_REFID, M_POS, M_FLAG, M_RLEN, M_LSEQ, M_HASSEQ, M_HAP, M_PS, M_HASH_LO, M_HASH_HI, M_NCIG, M_CIGD) = range(META_COLS)
CHECK_CRC = os.environ.get("NC_BGZF_CRC", "1") != "0"     # CRC-32 of every inflated member on the device (NC_BGZF_CRC=0: lengths only, as in round 4)
TRACE_LOAD = os.environ.get("NC_LOAD_TRACE") == "1"
LZ_SERIAL = os.environ.get("NC_LZ_SERIAL", "0") == "1"     # the match resolution behind the Huffman kernel on ONE stream instead of beside the next batch's
INFLATE_BATCH = int(os.environ.get("NC_INFLATE_BATCH", 28672))   # members per nc_inflate_device call: one full round of k_huff (256 CUs x 7 workgroups x 16 lanes); 7.5 GB of token workspace each, two in flight
POOL_MAX = 16 << 30              # buffers up to this size stay allocated between loads (larger ones go back to the allocator after use)
MAX_RESIDENT = 96 << 30          # upper bound of the inflated bytes kept in HBM at once; resident_limit() lowers it to what the device has free


def resident_limit(device=0):
    """inflated bytes one share of a file may take on `device`: the loader holds the stream (sized 8 x + 12.5 % of the compressed bytes), the file
    image (1/8 of the stream) and <= 8 GB of tokens; the read packs, the indel pipeline's workspaces and the CNN's need room beside them (two token workspaces of 7.5 GB at the default batch) -- half of
    what is free now (plus what this module's own pools already hold), at most MAX_RESIDENT"""
    try:
        free, _total = torch.cuda.mem_get_info(device)
        # blocks torch's caching allocator holds but has handed back (a released share larger than POOL_MAX lands there) are invisible to
        # mem_get_info and are served to the next torch.empty all the same: they count as free
        free += max(0, torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
    except Exception:
        return MAX_RESIDENT
    pooled = sum(t[0].numel() for k, t in _RAW_POOL.items() if k == device) + sum(t.numel() * t.element_size() for k, t in _WORK_POOL.items() if k[0] == device)
    # (the pools' tensors are torch allocations: part of memory_allocated, so they are added back here as the loader re-uses them)
    return int(max(1 << 30, min(MAX_RESIDENT, (free + pooled) // 2)))


LAST_LOAD = {}                   # seconds per stage of the most recent DeviceBam construction + load() (bench.py reports them)


class DeviceIngestUnavailable(RuntimeError):
    """the input cannot take the device route (no .bai, too large): the caller falls back to the host route"""


def _bai_path(path):
    """the index beside the BAM: .bai, else .csi (None: the device route is not available)"""
    for ext in (".bai", ".csi"):
        for p in (path + ext, os.path.splitext(path)[0] + ext):
            if os.path.exists(p):
                return p
    return None


_BAI = {}


def bai_linear_voffsets(bai_path):
    """{reference index: uint64 array of the non-zero virtual offsets of its 16 kb windows} (SAM specification 5.2); parsed once per file"""
    st = os.stat(bai_path)
    key = (os.path.abspath(bai_path), st.st_size, st.st_mtime_ns)
    if key not in _BAI:
        _BAI.clear()
        _BAI[key] = _csi_record_starts(bai_path) if bai_path.endswith(".csi") else _bai_linear_voffsets(bai_path)
    return _BAI[key]


def _csi_record_starts(csi_path):
    """the same from a CSI index (hts-specs CSIv1; the file is BGZF-compressed), which has no linear index: every bin's `loffset` and every
    chunk's begin are virtual offsets of record starts of that reference -- the leaf bins (16 kb with the default min_shift) make them as
    dense as a .bai's windows"""
    import gzip
    with open(csi_path, "rb") as f:
        buf = gzip.decompress(f.read())
    if buf[:4] != b"CSI\1":
        raise ValueError("%s is not a CSI file" % csi_path)
    _, depth, l_aux = struct.unpack_from("<3i", buf, 4)
    o = 16 + l_aux
    n_ref, = struct.unpack_from("<i", buf, o)
    o += 4
    meta_bin = ((1 << (depth * 3 + 3)) - 1) // 7 + 1                   # the pseudo-bin with the mapped / unmapped counts
    out = {}
    for r in range(n_ref):
        n_bin, = struct.unpack_from("<i", buf, o)
        o += 4
        starts = []
        for _ in range(n_bin):
            b, loff, n_chunk = struct.unpack_from("<IQi", buf, o)
            o += 16
            if b != meta_bin:
                starts.append(np.array([loff], np.uint64))
                starts.append(np.frombuffer(buf, np.uint64, 2 * n_chunk, o)[0::2])
            o += 16 * n_chunk
        v = np.concatenate(starts) if starts else np.zeros(0, np.uint64)
        out[r] = np.unique(v[v != 0])
    return out


def _bai_linear_voffsets(bai_path):
    with open(bai_path, "rb") as f:
        buf = f.read()
    if buf[:4] != b"BAI\1":
        raise ValueError("%s is not a BAI file" % bai_path)
    n_ref, = struct.unpack_from("<i", buf, 4)
    o, out = 8, {}
    for r in range(n_ref):
        n_bin, = struct.unpack_from("<i", buf, o)
        o += 4
        for _ in range(n_bin):
            _, n_chunk = struct.unpack_from("<Ii", buf, o)
            o += 8 + 16 * n_chunk
        n_intv, = struct.unpack_from("<i", buf, o)
        o += 4
        iv = np.frombuffer(buf, np.uint64, n_intv, o)
        o += 8 * n_intv
        out[r] = np.unique(iv[iv != 0])
    return out


class DeviceBam:
    """One BAM file: inflated and indexed in HBM by `load()`, then `prepare()` (host half, thread-safe) + `pack()` (device half) per contig."""

    def __init__(self, path, device=0, threads=None, contigs=None):
        """contigs: only these are wanted (a rank's share of a genome): the part of the file from the first record of the first of them to the
        first record behind the last is read and inflated, nothing else"""
        self.path, self.device = path, device
        self.eng = get_engine(device)
        bai = _bai_path(path)
        if bai is None:
            raise DeviceIngestUnavailable("%s: no .bai / .csi beside it" % path)
        self.lin = bai_linear_voffsets(bai)
        from .bam import rank_threads
        self.threads = threads or rank_threads()
        self.file_bytes = os.path.getsize(path)
        self._read_header()
        self.B0, self.B1, self.tids = 0, self.file_bytes, None
        if contigs is not None:
            unknown = [c for c in contigs if c not in self.ref_names]
            if unknown:
                raise ValueError("%s has no contig %r" % (path, unknown[0]))
            self.tids = sorted({self.ref_names.index(c) for c in contigs})
            have = [t for t in self.tids if t in self.lin and self.lin[t].size]
            if have:
                self.B0 = int(self.lin[have[0]][0] >> np.uint64(16))
                later = [t for t in sorted(self.lin) if t > have[-1] and self.lin[t].size]
                if later:                                                # through the member that holds the next contig's first record
                    self.B1 = min(self.file_bytes, int(self.lin[later[0]][0] >> np.uint64(16)) + 65536 + 1024)
            else:
                self.B1 = 0                                              # none of them has an alignment
        self.n_bytes = self.B1 - self.B0
        if self.n_bytes * 8 > resident_limit(device):                    # (the loader reserves eight times the compressed size for the inflated stream)
            raise DeviceIngestUnavailable("%s (%.1f GB to load): more than is kept in HBM at once" % (path, self.n_bytes / 1e9))
        self.loaded = False

    def _read_header(self):
        """reference names / lengths: the head of the file, inflated with zlib"""
        L, want = _lib.lib(), 1 << 20
        while True:
            with open(self.path, "rb") as f:
                head = np.frombuffer(f.read(min(want, self.file_bytes)), np.uint8)
            cap = head.size // 28 + 16
            coff, clen, isize = np.empty(cap, np.int64), np.empty(cap, np.int32), np.empty(cap, np.int32)
            k, nxt = C.c_int64(), C.c_int64()
            rc = L.nc_bgzf_scan(_lib.npp(head), head.size, 0, cap, _lib.npp(coff), _lib.npp(clen), _lib.npp(isize), C.byref(k), C.byref(nxt))
            if rc != _lib.NC_OK:
                raise _lib.NanoCallerHipError("%s is not a BGZF file (nc_bgzf_scan: %d)" % (self.path, rc))
            if self._header(head, coff[:int(k.value)], clen[:int(k.value)]):
                return
            if head.size >= self.file_bytes:
                raise _lib.NanoCallerHipError("%s: truncated BAM header" % self.path)
            want *= 8

    def _header(self, data, coff, clen):
        """reference names / lengths from the leading members (inflated with zlib: a few kilobytes).  -> False when the members seen so far
        do not hold the whole header yet"""
        got, k = b"", 0

        class Short(Exception):
            pass

        def need(n):
            nonlocal got, k
            while len(got) < n:
                if k >= len(coff):
                    raise Short()
                c = int(coff[k])
                got += zlib.decompress(data[c:c + int(clen[k])].tobytes(), -15)
                k += 1
        try:
            return self._header_fields(need, lambda: got)
        except Short:
            return False

    def _header_fields(self, need, buf):
        need(12)
        if buf()[:4] != b"BAM\1":
            raise _lib.NanoCallerHipError("%s is not a BAM file" % self.path)
        l_text, = struct.unpack_from("<i", buf(), 4)
        need(12 + l_text)
        n_ref, = struct.unpack_from("<i", buf(), 8 + l_text)
        o = 12 + l_text
        names, lengths = [], []
        for _ in range(n_ref):
            need(o + 4)
            l_name, = struct.unpack_from("<i", buf(), o)
            need(o + 8 + l_name)
            names.append(buf()[o + 4:o + 4 + l_name - 1].decode("ascii"))
            lengths.append(struct.unpack_from("<i", buf(), o + 4 + l_name)[0])
            o += 8 + l_name
        self.ref_names, self.ref_lengths = names, lengths
        return True

    def voffset_to_stream(self, voff):
        """virtual offsets (coffset << 16 | uoffset) -> offsets into the inflated stream"""
        voff = np.asarray(voff, np.uint64)
        co = (voff >> np.uint64(16)).astype(np.int64) - self.B0          # (member offsets are relative to the part of the file that was read)
        m = np.searchsorted(self.mstart, co)
        if m.size and (m.max() >= len(self.mstart) or not np.array_equal(self.mstart[m], co)):
            raise _lib.NanoCallerHipError("%s: an index entry does not point at a BGZF member" % self.path)
        return self.ooff[m] + (voff & np.uint64(0xffff)).astype(np.int64)

    # ------------------------------------------------------------------ the file, once
    def load(self):
        """The file -> HBM, inflated: worker threads read it into page-locked memory piece by piece; as soon as the pieces read so far hold
        INFLATE_BATCH more whole members their bytes are copied (copy stream) and inflated (compute stream) -- the GPU works while the rest
        of the file is still being read.  Then the record walk and the per-record fields."""
        if self.loaded:
            return self
        import time
        eng, L, dev = self.eng, _lib.lib(), self.eng.device
        eng.use_torch_stream()
        vp = lambda t, byte_off=0: C.c_void_p(t.data_ptr() + byte_off)   # noqa: E731
        t_start = time.perf_counter()
        n = self.n_bytes
        self.host_buf = torch.empty(n + 64, dtype=torch.uint8, pin_memory=True)
        t_pin = time.perf_counter() - t_start
        data = self.host_buf.numpy()
        data[n:] = 0
        view = memoryview(data)
        piece = max(32 << 20, -(-n // 64))
        fd = os.open(self.path, os.O_RDONLY)

        def read_piece(a):                                               # (buffer offsets; the file's are B0 higher)
            b_, o = min(n, a + piece), a
            while o < b_:
                got = os.preadv(fd, [view[o:b_]], self.B0 + o)
                if got <= 0:
                    raise IOError("short read of %s" % self.path)
                o += got
            return b_
        # half the rank's thread budget, at most eight: with the whole budget reading, the caller's other threads (reference letters, the launching
        # thread) push the process over its CPU quota and the scheduler throttles all of them for the rest of the period (measured: 22 -> 270 ms)
        pool = ThreadPoolExecutor(max_workers=max(2, min(self.threads // 2, 8)))
        futures = [pool.submit(read_piece, a) for a in range(0, n, piece)]
        raw_cap = n * 8 + (64 << 20)                                     # the inflated size is known only at the end: room for eight-fold
        # the stream's buffer is kept from load to load (share after share of a genome-sized file, run after run of a bench): a fresh multi-gigabyte
        # request that misses the allocator's cache makes it release and re-acquire device memory, which was seen to take > 100 ms
        import weakref
        pooled = _RAW_POOL.get(dev.index)
        if pooled is not None and pooled[0].numel() >= raw_cap and pooled[1]() in (None, self):
            self.raw = pooled[0]
        else:
            self.raw = torch.empty(raw_cap + raw_cap // 8, dtype=torch.uint8, device=dev)
        if self.raw.numel() <= POOL_MAX:
            _RAW_POOL[dev.index] = (self.raw, weakref.ref(self))
        else:
            _RAW_POOL.pop(dev.index, None)
        d_file = _work_buffer(dev, "file", n + 64, torch.uint8)
        cap = n // 1024 + 4096
        stage64 = torch.empty(2 * cap, dtype=torch.int64, pin_memory=True)    # [coff | ooff] of every member, page-locked: the batches' uploads are async
        stage32 = torch.empty(2 * cap, dtype=torch.int32, pin_memory=True)    # [clen | isize]
        coff, ooff = stage64.numpy()[:cap], stage64.numpy()[cap:]
        clen, isize = stage32.numpy()[:cap], stage32.numpy()[cap:]
        # two token workspaces: the Huffman kernel of batch i + 1 (compute stream) runs beside the match resolution of batch i (its own stream)
        # (256 KB of tokens per member: a small file gets small workspaces -- and, should its members be unusually short, more batches)
        batch = min(INFLATE_BATCH, max(64, (n // 4096 + 63) // 64 * 64))
        toks = [(_work_buffer(dev, "tok%d" % k_, ((batch + 63) // 64) << 22, torch.int32), torch.zeros(batch, dtype=torch.int32, device=dev), [None])
                for k_ in range(2)]
        t_alloc = time.perf_counter() - t_start
        copy_stream, lz_stream = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        compute = torch.cuda.current_stream(dev)
        lz_stream.wait_stream(compute)                                   # (the output buffer's allocation)
        n_batches = [0]
        n_mem, m0, scan_pos, total, statuses, keep = 0, 0, 0, 0, [], []
        traces = []

        pending = []                                                     # the batch whose bytes are on their way and whose kernels are not enqueued yet

        def kernels():
            """the inflate of the pending batch.  It is enqueued AFTER the next batch's copy: a host-to-device copy submitted while k_huff runs was
            seen to start only when that kernel ended (batch i + 1's 20 ms of PCIe behind batch i's 27 ms of Huffman decoding instead of under them)"""
            if not pending:
                return
            k, d64, d32, ev, tr, t_host, nbytes = pending.pop()

            def mark(stream, what):
                if tr is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(stream)
                    tr.append((what, e))
            compute.wait_event(ev)
            st = torch.zeros(k, dtype=torch.int32, device=dev)
            d_tok, d_ntok, lz_done = toks[n_batches[0] & 1]
            n_batches[0] += 1
            if lz_done[0] is not None:
                compute.wait_event(lz_done[0])                           # the workspace's previous tokens have been resolved
            args = (k, vp(d_file), vp(d64[0]), vp(d32[0]), vp(self.raw), vp(d64[1]), vp(d32[1]), vp(st), vp(d_tok), vp(d_ntok))
            mark(compute, "huff0")
            eng._check(L.nc_inflate_device_phase(eng.ctx, 1, *args), "nc_inflate_device_phase")
            mark(compute, "huff1")
            huffed = torch.cuda.Event()
            huffed.record(compute)
            with torch.cuda.stream(compute if LZ_SERIAL else lz_stream):
                (compute if LZ_SERIAL else lz_stream).wait_event(huffed)
                eng.use_torch_stream()
                mark(compute if LZ_SERIAL else lz_stream, "lz0")
                eng._check(L.nc_inflate_device_phase(eng.ctx, 2, *args), "nc_inflate_device_phase")
                mark(compute if LZ_SERIAL else lz_stream, "lz1")
                if CHECK_CRC:                                            # what htslib does for every block it inflates: the member's CRC-32 against its trailer
                    eng._check(L.nc_bgzf_crc_device(eng.ctx, *args[:8]), "nc_bgzf_crc_device")
                mark(compute if LZ_SERIAL else lz_stream, "crc1")
                lz_done[0] = torch.cuda.Event()
                lz_done[0].record(compute if LZ_SERIAL else lz_stream)
            if tr is not None:
                traces.append((t_host, k, nbytes, tr))
            eng.use_torch_stream()
            statuses.append(st)
            keep.append((d64, d32))                                      # (allocated on the copy stream, read on the compute stream: alive until the sync)

        def launch(m1):
            nonlocal m0
            a_byte = int(coff[m0 - 1]) + int(clen[m0 - 1]) + 8 if m0 else 0   # (from the batch's first member; the headers ride along)
            b_byte = int(coff[m1 - 1]) + int(clen[m1 - 1]) + 8
            k = m1 - m0
            tr = [] if TRACE_LOAD else None
            with torch.cuda.stream(copy_stream):
                if tr is not None:                                       # (NC_LOAD_TRACE=1: where a load's time goes, by events)
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(copy_stream)
                    tr.append(("copy0", e))
                d_file[a_byte:b_byte].copy_(self.host_buf[a_byte:b_byte], non_blocking=True)
                if tr is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(copy_stream)
                    tr.append(("copy1", e))
                d64 = stage64[m0:m1].to(dev, non_blocking=True), stage64[cap + m0:cap + m1].to(dev, non_blocking=True)
                d32 = stage32[m0:m1].to(dev, non_blocking=True), stage32[cap + m0:cap + m1].to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            kernels()                                                    # the batch before this one
            pending.append((k, d64, d32, ev, tr, time.perf_counter() - t_start, b_byte - a_byte))
            m0 = m1
        try:
            t_wait = t_launch = 0.0
            for fut in futures:
                tw = time.perf_counter()
                avail = fut.result()
                t_wait += time.perf_counter() - tw
                while True:
                    k, nxt = C.c_int64(), C.c_int64()
                    rc = L.nc_bgzf_scan(_lib.npp(data), avail, scan_pos, cap - n_mem, vp(stage64, 8 * n_mem), vp(stage32, 4 * n_mem), vp(stage32, 4 * (cap + n_mem)),
                                        C.byref(k), C.byref(nxt))
                    if rc != _lib.NC_OK:
                        raise _lib.NanoCallerHipError("%s is not a BGZF file (nc_bgzf_scan: %d at byte %d)" % (self.path, rc, scan_pos))
                    k = int(k.value)
                    if k:
                        ooff[n_mem] = total
                        if k > 1:
                            np.cumsum(isize[n_mem:n_mem + k - 1], out=ooff[n_mem + 1:n_mem + k])
                            ooff[n_mem + 1:n_mem + k] += total
                        total = int(ooff[n_mem + k - 1]) + int(isize[n_mem + k - 1])
                    n_mem += k
                    scan_pos = int(nxt.value)
                    if n_mem < cap or scan_pos >= avail:
                        break
                    raise DeviceIngestUnavailable("%s: more BGZF members than planned for" % self.path)
                if total + 64 > raw_cap:
                    raise DeviceIngestUnavailable("%s inflates more than eight-fold" % self.path)
                tw = time.perf_counter()
                while n_mem - m0 >= batch:
                    launch(m0 + batch)
                t_launch += time.perf_counter() - tw
            if scan_pos != n and self.B1 == self.file_bytes:
                raise _lib.NanoCallerHipError("%s does not end with a whole BGZF member" % self.path)
            while n_mem > m0:
                launch(min(n_mem, m0 + batch))
            kernels()
        finally:
            pool.shutdown(wait=True)
            os.close(fd)
        LAST_LOAD.clear()
        LAST_LOAD["read_scan_enqueue"] = time.perf_counter() - t_start
        LAST_LOAD["of_which_page_locked_alloc"] = t_pin
        LAST_LOAD["of_which_allocations"], LAST_LOAD["of_which_waiting_for_readers"], LAST_LOAD["of_which_enqueue"] = t_alloc, t_wait, t_launch
        t0 = time.perf_counter()
        self.coff, self.clen, self.isize = coff[:n_mem].copy(), clen[:n_mem].copy(), isize[:n_mem].copy()
        self.ooff = np.concatenate([ooff[:n_mem], [total]]).astype(np.int64)
        self.mstart = np.zeros(n_mem, np.int64)                                               # file offset of every member
        self.mstart[1:] = self.coff[:-1] + self.clen[:-1] + 8
        lz_stream.synchronize()
        bad = sum(int(st.count_nonzero().item()) for st in statuses)     # (also: the inflate is done)
        if bad:
            crc = sum(int((st == 7).sum().item()) for st in statuses)
            raise _lib.NanoCallerHipError("%s: %d BGZF members are not valid deflate streams of their announced size%s"
                                          % (self.path, bad - crc, (", %d fail their CRC-32" % crc) if crc else ""))
        compute.wait_stream(lz_stream)
        if traces:
            e0 = traces[0][3][0][1]
            for t_host, k_, nb_, tr_ in traces:
                print("  batch of %d members, %.0f MB, enqueued at %.1f ms: " % (k_, nb_ / 1e6, t_host * 1e3) + ", ".join("%s %.1f" % (w, e0.elapsed_time(e)) for w, e in tr_), flush=True)
        del toks, d_file, keep, statuses
        self.host_buf = None
        LAST_LOAD["inflate_wait"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        self.n_rec = 0
        self.meta = np.zeros((META_COLS, 0), np.int32)
        self.rec_off = np.zeros(0, np.int64)
        seeds, tids = [], []
        # chain starts: the linear index entries of every contig
        for tid in (sorted(self.lin) if self.tids is None else self.tids):
            v = self.lin.get(tid, np.zeros(0, np.uint64))
            if v.size:
                seeds.append(self.voffset_to_stream(v))
                tids.append(np.full(v.size, tid, np.int32))
        if seeds:
            seed, tid = np.concatenate(seeds), np.concatenate(tids)
            if np.any(np.diff(seed) <= 0):
                raise _lib.NanoCallerHipError("%s: the index entries are not in file order" % self.path)
            d_seed, d_tid = torch.from_numpy(seed).to(dev), torch.from_numpy(tid).to(dev)
            d_cnt = torch.zeros(seed.size, dtype=torch.int64, device=dev)
            d_status = torch.zeros(1, dtype=torch.int32, device=dev)
            eng._check(L.nc_bam_walk(eng.ctx, vp(self.raw), total, seed.size, vp(d_seed), vp(d_tid), None, vp(d_cnt), vp(d_status)), "nc_bam_walk")
            d_first = torch.cumsum(d_cnt, 0) - d_cnt
            self.n_rec = int(d_cnt.sum().item())
            self.d_rec_off = torch.empty(max(1, self.n_rec), dtype=torch.int64, device=dev)
            eng._check(L.nc_bam_walk(eng.ctx, vp(self.raw), total, seed.size, vp(d_seed), vp(d_tid), vp(d_first), vp(self.d_rec_off), vp(d_status)), "nc_bam_walk")
            d_meta = torch.empty((META_COLS, max(1, self.n_rec)), dtype=torch.int32, device=dev)
            eng._check(L.nc_bam_meta(eng.ctx, vp(self.raw), self.n_rec, vp(self.d_rec_off), vp(d_meta), vp(d_status)), "nc_bam_meta")
            self.meta = d_meta.cpu().numpy()[:, :self.n_rec]
            self.rec_off = self.d_rec_off.cpu().numpy()[:self.n_rec]
            st = int(d_status.item())
            if st:
                raise _lib.NanoCallerHipError("%s: corrupt BAM records or index (status %d)" % (self.path, st))
        # the records of a contig are one run of the record list (coordinate-sorted file)
        refid = self.meta[M_REFID]
        self.tid_range = {}
        if self.n_rec:
            cut = np.flatnonzero(np.diff(refid)) + 1
            lo = np.concatenate([[0], cut])
            hi = np.concatenate([cut, [self.n_rec]])
            for a, b in zip(lo.tolist(), hi.tolist()):
                t = int(refid[a])
                if t in self.tid_range:
                    raise _lib.NanoCallerHipError("%s is not coordinate-sorted (contig %d appears twice)" % (self.path, t))
                self.tid_range[t] = (a, b)
        LAST_LOAD["walk_meta"] = time.perf_counter() - t0
        LAST_LOAD["members"], LAST_LOAD["records"], LAST_LOAD["inflated_bytes"] = n_mem, self.n_rec, total
        self.loaded = True
        return self

    def fetch_names(self, recs):
        """read names of the records `recs` (indices into the record list): a D2H of a few bytes each (the duplicate-name check's slow path)"""
        out = []
        for r in np.asarray(recs).tolist():
            o = int(self.rec_off[r])
            head = self.raw[o + 4:o + 4 + 32].cpu().numpy()
            l_name = int(head[8])
            out.append(self.raw[o + 36:o + 36 + l_name - 1].cpu().numpy().tobytes().decode("ascii", "replace"))
        return out

    # ------------------------------------------------------------------ one contig: the host's decisions
    def prepare(self, chrom, ref, supplementary=False, exclude=None, span=None, tile_size=2048):
        """host half (numpy + native, no GPU call: may run on a worker thread).  ref: the contig's sequence.  -> dict for pack()"""
        if not self.loaded:
            raise RuntimeError("DeviceBam.load() first")
        if chrom not in self.ref_names:
            raise ValueError("%s has no contig %r" % (self.path, chrom))
        tid = self.ref_names.index(chrom)
        length = self.ref_lengths[tid]
        beg1, end1 = (1, length) if span is None else (max(1, int(span[0])), int(span[1]))
        a, b = self.tid_range.get(tid, (0, 0))
        m = self.meta[:, a:b]
        pos, rlen, flag = m[M_POS], m[M_RLEN], m[M_FLAG]
        # the alignments nc_bam_decode(tid, beg1, end1) returns
        sel = (pos < end1) & ((flag & 0x4) == 0) & (rlen > 0) & (pos.astype(np.int64) + rlen > beg1 - 1)
        idx = np.flatnonzero(sel)
        start = (pos[idx] + 1).astype(np.int32)
        end = (start + rlen[idx]).astype(np.int32)
        flag = flag[idx]
        # inputs the library does not reproduce (generate_SNP_pileups._check_supported)
        mask = 0x704 if supplementary else 0xF04
        kept_for_check = (flag & mask) == 0
        n_skip = int(np.count_nonzero(kept_for_check & ((flag & _lib.FLAG_REFSKIP) != 0)))
        h = (m[M_HASH_LO][idx].astype(np.uint32).astype(np.uint64) | (m[M_HASH_HI][idx].astype(np.uint32).astype(np.uint64) << np.uint64(32)))
        n_dup = self._same_name_overlaps(h, start, end, kept_for_check, a + idx, any_pair=bool(supplementary))
        if n_skip or n_dup:
            what = []
            if n_skip:
                what.append("%d alignments with a reference skip (CIGAR N) would enter the pileup; the reference's code table has no entry for "
                            "their '>' / '<' symbols (generate_SNP_pileups.py:104)" % n_skip)
            if n_dup:
                what.append("%d pairs of kept alignments carry the same read name%s; the reference's per-column dicts hold one entry per name "
                            "(generate_SNP_pileups.py:175,185,208): the host route (NC_DEVICE_INGEST=0) keys them by name" % (n_dup, "" if supplementary else " and overlap on the reference"))
            err = _lib.NanoCallerHipError("%s, contig %s: %s -- NC_ERR_UNSUPPORTED" % (self.path, chrom, "; ".join(what)))
            err.status = _lib.NC_ERR_UNSUPPORTED
            raise err
        filt = FLAG_FILTER_SUPPL if supplementary else FLAG_FILTER_DEFAULT
        keep = pileup_depth_cap(start, end, np.ascontiguousarray((flag & filt) == 0, np.uint8))
        strand = np.ascontiguousarray(((flag & 0x10) != 0).astype(np.uint8) | ((m[M_HAP][idx].astype(np.uint8) & 3) << 1))
        # tile index + slot layout (nc_pack_plan / nc_pack_fill, index only: what wire.build_wire does)
        L = _lib.lib()
        ref_bytes = np.frombuffer(ref.encode("ascii") if isinstance(ref, str) else ref, np.uint8)
        Lref = int(ref_bytes.shape[0])
        pos_lo = 1 if span is None else max(1, int(span[0]))
        pos_hi = max(pos_lo, Lref if span is None else min(Lref, int(span[1])))
        codes_len, n_ent = C.c_int64(), C.c_int64()
        tile_pos0, n_tiles = C.c_int32(), C.c_int32()
        n = int(idx.size)
        rc = L.nc_pack_plan(n, _lib.npp(start), _lib.npp(end), _lib.npp(keep), tile_size, pos_lo, pos_hi, C.byref(codes_len), C.byref(tile_pos0),
                            C.byref(n_tiles), C.byref(n_ent))
        if rc != _lib.NC_OK:
            raise _lib.NanoCallerHipError("nc_pack_plan failed (%d)" % rc)
        tile_off = np.empty(n_tiles.value + 1, np.int32)
        tile_ent = np.empty(max(1, n_ent.value), _lib.TILE_ENTRY_DTYPE)
        rc = L.nc_pack_fill(n, _lib.npp(start), _lib.npp(end), None, None, _lib.npp(strand), _lib.npp(keep), tile_size, tile_pos0.value,
                            n_tiles.value, None, codes_len.value, _lib.npp(tile_off), _lib.npp(tile_ent), n_ent.value)
        if rc != _lib.NC_OK:
            raise _lib.NanoCallerHipError("nc_pack_fill (index) failed (%d)" % rc)
        kk = np.flatnonzero(keep)
        ks, ke = start[kk], end[kk]
        size = ((ke.astype(np.int64) + 15) & ~15) - (ks.astype(np.int64) & ~15)
        slot = np.zeros(kk.size, np.int64)
        if kk.size:
            np.cumsum(size[:-1], out=slot[1:])
        mk = m[:, idx[kk]]
        ncig = mk[M_NCIG].astype(np.int64) | np.where(mk[M_HASSEQ] != 0, 0, 1 << 31)
        # everything the device half uploads, back to back in ONE page-locked buffer (a pageable source makes the copy wait for the GPU to
        # finish what is queued before it -- the previous contig's CNN -- and the launching thread with it)
        Lref_b = int(ref_bytes.shape[0])
        ga, gb = max(1, tile_pos0.value), min(Lref_b, tile_pos0.value + n_tiles.value * tile_size - 1)
        parts = dict(rec=np.ascontiguousarray(self.rec_off[a + idx[kk]]), slot=slot, cigd=np.ascontiguousarray(mk[M_CIGD]),
                     ncig=ncig.astype(np.uint32).view(np.int32), start=np.ascontiguousarray(ks), rd_end=np.ascontiguousarray(ke),
                     slot_off=np.concatenate([slot, [int(size.sum())]]).astype(np.int64), read_hap=np.ascontiguousarray(mk[M_HAP].astype(np.uint8)),
                     read_ps=np.ascontiguousarray(mk[M_PS]), read_flag=np.ascontiguousarray((mk[M_LSEQ] == 0).astype(np.uint8)), tile_off=tile_off,
                     tile_ent=tile_ent.view(np.uint8).reshape(-1), ref_letters=ref_bytes[ga - 1:gb] if gb >= ga else ref_bytes[:0])
        sections, total_b = {}, 0
        for k, v in parts.items():
            sections[k] = (total_b, v.dtype, int(v.size))
            total_b += (v.nbytes + 255) & ~255
        staged = torch.empty(max(256, total_b), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        host = staged.numpy()
        for k, v in parts.items():
            o = sections[k][0]
            host[o:o + v.nbytes] = v.view(np.uint8).reshape(-1)
        return dict(chrom=chrom, n_kept=int(kk.size), staged=staged, sections=sections, ref_span=(ga, gb), exclude=list(exclude or ()),
                    codes_len=int(codes_len.value), tile_size=tile_size,
                    tile_pos0=int(tile_pos0.value), n_tiles=int(n_tiles.value), n_entries=int(n_ent.value), pos_lo=pos_lo, pos_hi=pos_hi,
                    n_reads=n, read_start=start, read_end=end, read_flag=flag, keep=keep)

    def _same_name_overlaps(self, h, start, end, keep, recs, any_pair=False):
        """bam.same_name_overlaps on name hashes; hash-equal pairs are confirmed on the names themselves.  any_pair: count every pair of kept
        alignments that share a name, overlapping or not (under dct['supplementary'] the reference looks neighbour columns up by NAME: the host
        route keys such records by name, pack.name_groups; this route does not)"""
        k = np.flatnonzero(keep)
        if k.size < 2:
            return 0
        hk = h[k]
        u, inv, cnt = np.unique(hk, return_inverse=True, return_counts=True)
        cand = k[cnt[inv] > 1]                                            # kept alignments whose name (hash) occurs more than once
        if cand.size == 0:
            return 0
        names = self.fetch_names(recs[cand])
        last, n = {}, 0
        for r, nm in zip(cand.tolist(), names):
            e = last.get(nm)
            if e is not None and (any_pair or int(start[r]) < e):
                n += 1
            last[nm] = max(int(end[r]), e or 0)
        return n

    def _ref_lut(self):
        if getattr(self, "_lut", None) is None:
            lut = np.full(256, 4, np.uint8)                              # upper-case AGTC are scanned; soft-masked / other letters are not (quirk E4)
            for i, ch in enumerate("AGTC"):
                lut[ord(ch)] = i
            self._lut = torch.from_numpy(lut).to(self.eng.device)
        return self._lut

    # ------------------------------------------------------------------ one contig: the slots, in HBM
    def pack(self, prep, codes=None, indel=False, tail_cap=272) -> DevicePack:
        """device half: uploads the (small) arrays of prepare() and decodes the kept reads into `codes` (allocated when None).  indel=True: the
        pack also carries the indel path's sections (events, inserted bases, tails: what wire.build_wire(events=..., indel_extra=...) uploads on
        the host route), made from the record stream by nc_bam_indel_reads"""
        eng, L, dev = self.eng, _lib.lib(), self.eng.device
        eng.use_torch_stream()
        vp = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        dbuf = prep["staged"].to(dev, non_blocking=True)
        tdt = {np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32, np.dtype(np.uint8): torch.uint8}

        def sec(k):
            o, dt, cnt = prep["sections"][k]
            return dbuf[o:o + cnt * np.dtype(dt).itemsize].view(tdt[np.dtype(dt)])
        n = prep["codes_len"]
        if codes is None:
            codes = torch.empty(n, dtype=torch.uint8, device=dev)
        codes = codes[:n]
        codes.fill_(7)                                                   # NC_CODE_ABSENT
        if prep["n_kept"]:
            eng._check(L.nc_bam_codes(eng.ctx, vp(self.raw), prep["n_kept"], vp(sec("rec")), vp(sec("slot")), vp(sec("cigd")), vp(sec("ncig")), vp(sec("start")),
                                      vp(codes)), "nc_bam_codes")
        # the scan's reference codes on the tile grid, from the contig's letters (wire.ref_wire_from_string + nc_wire_expand's rule, in HBM: the
        # 9 MB table passes of a contig kept a worker thread -- and the interpreter lock the launching thread needs -- busy for milliseconds)
        ref_len, t0 = prep["n_tiles"] * prep["tile_size"], prep["tile_pos0"]
        ref_code = torch.full((ref_len,), 4, dtype=torch.uint8, device=dev)
        ga, gb = prep["ref_span"]
        if gb >= ga:
            ref_code[ga - t0:gb - t0 + 1] = self._ref_lut()[sec("ref_letters").to(torch.int32)]
        for (a, b) in prep["exclude"]:                                   # tree.overlaps(pos): a <= pos < b (generate_SNP_pileups.py:116-119,161)
            lo, hi = max(1, int(a)) - t0, max(1, int(b)) - t0
            if hi > max(lo, 0):
                ref_code[max(lo, 0):min(hi, ref_len)] = 4
        dp = DevicePack(codes=codes, tile_off=sec("tile_off"), tile_ent=sec("tile_ent"), ref_code=ref_code, tile_size=prep["tile_size"],
                        tile_pos0=prep["tile_pos0"], n_tiles=prep["n_tiles"], n_entries=prep["n_entries"], pos_lo=prep["pos_lo"], pos_hi=prep["pos_hi"])
        if indel:
            K = prep["n_kept"]
            i32 = lambda n_: torch.zeros(max(int(n_), 4), dtype=torch.int32, device=dev)   # noqa: E731
            u8 = lambda n_: torch.zeros(max(int(n_), 4), dtype=torch.uint8, device=dev)    # noqa: E731
            counts = torch.zeros((3, max(K, 1)), dtype=torch.int32, device=dev)
            args = (vp(self.raw), K, vp(sec("rec")), vp(sec("cigd")), vp(sec("ncig")), int(tail_cap))
            eng._check(L.nc_bam_indel_reads(eng.ctx, *args, vp(counts), None, None, None, None, None, None, None, None), "nc_bam_indel_reads")
            offs = torch.zeros((3, K + 1), dtype=torch.int64, device=dev)
            if K:
                torch.cumsum(counts[:, :K], 1, out=offs[:, 1:])
            n_ev, n_ins, n_tail = (int(x) for x in offs[:, K].tolist())
            if max(n_ev, n_ins, n_tail) > 2**31 - 2:
                raise _lib.NanoCallerHipError("more than 2^31 indel events / inserted bases in one contig")
            o32 = offs.to(torch.int32)
            ev_pos, ev_len, ins_off, ins_bases, tail_bases = i32(n_ev), i32(n_ev), i32(n_ev + 1), u8(n_ins), u8(n_tail)
            if K:
                eng._check(L.nc_bam_indel_reads(eng.ctx, *args, None, vp(o32[0]), vp(o32[1]), vp(o32[2]), vp(ev_pos), vp(ev_len), vp(ins_off), vp(ins_bases),
                                                vp(tail_bases)), "nc_bam_indel_reads")
            ins_off[n_ev] = n_ins
            z = lambda t: t if t.numel() else torch.zeros(4, dtype=t.dtype, device=dev)   # noqa: E731
            dp.events = dict(n_reads=K, ev_off=o32[0].contiguous(), ev_pos=ev_pos[:max(n_ev, 1)], ev_len=ev_len[:max(n_ev, 1)], read_hap=z(sec("read_hap")))
            dp.reads = dict(n_reads=K, rd_start=z(sec("start")), rd_end=z(sec("rd_end")), slot_off=sec("slot_off"))
            dp.indel = dict(ins_off=ins_off[:n_ev + 1], ins_bases=ins_bases, tail_off=o32[2].contiguous(), tail_bases=tail_bases, read_ps=z(sec("read_ps")),
                            read_flag=z(sec("read_flag")))
        return dp


def contig_spans(path):
    """{contig: (lo, hi)} = the bytes of the BAM file that hold its records, from the .bai alone: the member of its first record to the member of
    the next contig's first record (inclusive: a record straddles members); contigs without alignments are absent.  Raises
    DeviceIngestUnavailable when there is no .bai."""
    probe = DeviceBam.__new__(DeviceBam)
    bai = _bai_path(path)
    if bai is None:
        raise DeviceIngestUnavailable("%s: no .bai / .csi beside it" % path)
    probe.path, probe.lin, probe.file_bytes = path, bai_linear_voffsets(bai), os.path.getsize(path)
    probe._read_header()
    with_reads = [t for t in sorted(probe.lin) if probe.lin[t].size and t < len(probe.ref_names)]
    out = {}
    for k, t in enumerate(with_reads):
        hi = min(probe.file_bytes, int(probe.lin[with_reads[k + 1]][0] >> np.uint64(16)) + 65536 + 1024) if k + 1 < len(with_reads) else probe.file_bytes
        out[probe.ref_names[t]] = (int(probe.lin[t][0] >> np.uint64(16)), hi)
    return out, list(probe.ref_names)


def plan_shares(path, contigs, limit_bytes=None, device=None):
    """`contigs` (in the order they will be called) cut into runs whose part of the file -- first record of the run's first contig to the first
    record behind its last -- stays under `limit_bytes` of compressed BAM (default: what DeviceBam accepts), so that a genome-sized file
    passes through HBM share by share.  -> list of (contigs of the share, fits): fits False = that contig alone is too large (host route).
    Raises DeviceIngestUnavailable when there is no .bai."""
    spans, names = contig_spans(path)
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    limit = (resident_limit(device) // 8) if limit_bytes is None else int(limit_bytes)
    shares, cur, lo, hi = [], [], None, None
    for c in contigs:
        if c not in names:
            raise ValueError("%s has no contig %r" % (path, c))
        sp = spans.get(c)
        if sp is None:                                                   # no alignments: rides along
            cur.append(c)
            continue
        if sp[1] - sp[0] > limit:
            if cur:
                shares.append((cur, True))
            shares.append(([c], False))
            cur, lo, hi = [], None, None
            continue
        nlo, nhi = (sp[0], sp[1]) if lo is None else (min(lo, sp[0]), max(hi, sp[1]))
        if cur and lo is not None and nhi - nlo > limit:
            shares.append((cur, True))
            cur, nlo, nhi = [], sp[0], sp[1]
        cur.append(c)
        lo, hi = nlo, nhi
    if cur:
        shares.append((cur, True))
    return shares


_OPEN = {}
_RAW_POOL = {}               # device index -> (buffer of the inflated stream, weak reference to the DeviceBam that is using it)
_WORK_POOL = {}              # (device index, name) -> the loader's work buffers (file image in HBM, token workspaces), kept from load to load


def _work_buffer(dev, name, numel, dtype):
    """a buffer of the loader that lives only during load(): the same allocation every time (grown when a larger file comes).  Handing
    these back to the framework's allocator let it cut them up for other requests in between; the next load then had to get ~9 GB of fresh
    device memory, which was measured at 240 ms"""
    t = _WORK_POOL.get((dev.index, name))
    if t is None or t.numel() < numel or t.dtype != dtype:
        _WORK_POOL.pop((dev.index, name), None)
        t = torch.empty(numel, dtype=dtype, device=dev)
        if numel * t.element_size() <= POOL_MAX:
            _WORK_POOL[(dev.index, name)] = t
    return t[:numel]


def open_device_bam(path, device=0, contigs=None) -> DeviceBam:
    """the loaded DeviceBam of (path, device) holding at least `contigs` (None: the whole file), cached by path + size + mtime"""
    st = os.stat(path)
    want = None if contigs is None else frozenset(contigs)
    ident = (os.path.abspath(path), st.st_size, st.st_mtime_ns, device)
    for k, db in list(_OPEN.items()):
        if k[:4] == ident and (k[4] is None or (want is not None and want <= k[4])):
            return db.load()
        if k[0] == ident[0] and k[3] == device:
            del _OPEN[k]                                                # another version of the file, or another share of it
    db = _OPEN[ident + (want,)] = DeviceBam(path, device, contigs=None if want is None else sorted(want))
    try:
        return db.load()
    except (torch.cuda.OutOfMemoryError, MemoryError) as e:             # device or page-locked memory: the host route needs neither
        _OPEN.pop(ident + (want,), None)
        release(path, buffers=True)
        raise DeviceIngestUnavailable("%s: %s" % (path, str(e).splitlines()[0] if str(e) else type(e).__name__))
    except RuntimeError as e:
        if "out of memory" in str(e).lower() or "hipErrorOutOfMemory" in str(e):
            _OPEN.pop(ident + (want,), None)
            release(path, buffers=True)
            raise DeviceIngestUnavailable("%s: %s" % (path, str(e).splitlines()[0]))
        raise


def release(path=None, buffers=False):
    """forget the loaded files (of `path`, or all); buffers=True also gives the pooled stream buffers back"""
    for k in [k for k in _OPEN if path is None or k[0] == os.path.abspath(path)]:
        del _OPEN[k]
    if buffers:
        _RAW_POOL.clear()
        _WORK_POOL.clear()
