"""Device engine: drives libnanocaller_hip.so from Python.

PyTorch is used only as plumbing -- device memory (tensors), the current HIP stream and, for multi-GPU
runs, torch.distributed rendezvous.  All arithmetic of the hot path runs in the library's HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .pack import HostPack
from .weights import Weights


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@dataclass
class DevicePack:
    codes: torch.Tensor
    tile_off: torch.Tensor
    tile_ent: torch.Tensor      # uint8 view of the 16-byte entries
    ref_code: torch.Tensor
    tile_size: int
    tile_pos0: int
    n_tiles: int
    n_entries: int
    pos_lo: int
    pos_hi: int
    events: dict | None = None    # device tensors ev_off / ev_pos / ev_len / read_hap (indel scan inputs)
    reads: dict | None = None     # device tensors rd_start / rd_end / slot_off of the kept reads (device pass 2: tile entry -> read)
    indel: dict | None = None     # device tensors ins_off / ins_bases / tail_off / tail_bases / read_ps / read_flag (device pass 2)
    mates: tuple | None = None    # (key int64 [M], rec int32 [M, 4]) device tensors: alignments that share read names (nc_snp_set_mates)

    def c_struct(self) -> _lib.ReadPackC:
        return _lib.ReadPackC(codes_len=self.codes.numel(), codes=self.codes.data_ptr(), tile_size=self.tile_size,
                              tile_pos0=self.tile_pos0, n_tiles=self.n_tiles, tile_off=self.tile_off.data_ptr(),
                              tile_ent=self.tile_ent.data_ptr(), n_entries=self.n_entries)

    @property
    def nbytes(self):
        return self.codes.numel() + self.tile_off.numel() * 4 + self.tile_ent.numel() + self.ref_code.numel()


@dataclass
class SnpSites:
    """Results of scan + featurize for a batch of chunks (device tensors unless noted)."""
    n_sites: int
    n_nbr: int
    pos: np.ndarray            # host int32 [N]
    chunk: np.ndarray          # host int32 [N]
    dp: np.ndarray             # host int32 [N]   pileup entries incl. deletions
    alt: np.ndarray            # host int32 [N]   max non-reference base count
    x: torch.Tensor | None = None          # f32 [N,5,41,5]
    ref_code: torch.Tensor | None = None   # i32 [N]
    fwd_dp: torch.Tensor | None = None     # i32 [N,4]
    rev_dp: torch.Tensor | None = None
    depth: torch.Tensor | None = None      # i32 [N]
    valid: torch.Tensor | None = None      # u8 [N]


class _PinnedPool:
    """Page-locked host blocks for result arrays.  hipHostMalloc costs milliseconds, so blocks are reused -- but only
    when no numpy view handed out from a block is alive any more (tracked with a weak reference), so results of
    earlier calls stay valid for as long as the caller keeps them."""

    def __init__(self):
        self.blocks = []                                  # [uint8 pinned tensor, weakref to the numpy array handed out]

    def get(self, shape, dtype):
        """-> (torch tensor view, numpy array) over the same pinned memory"""
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * torch.empty(0, dtype=dtype).element_size()
        best = None
        for b in self.blocks:
            if (b[1] is None or b[1]() is None) and b[0].numel() >= nbytes and (best is None or b[0].numel() < best[0].numel()):
                best = b
        if best is None:
            cap = max(4096, nbytes + nbytes // 4)
            best = [torch.empty(cap, dtype=torch.uint8, pin_memory=True), None]
            self.blocks.append(best)
        t = best[0][:nbytes].view(dtype).view(shape)
        arr = t.numpy()
        best[1] = weakref.ref(arr)
        return t, arr


class Engine:
    """One engine per GPU (one process per GPU in multi-GPU runs).  Not thread-safe."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise _lib.NanoCallerHipError("no ROCm device visible: the NanoCaller HIP path has no CPU fallback")
        self.L = _lib.lib()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        ctx = C.c_void_p()
        rc = self.L.nc_ctx_create(device, C.byref(ctx))
        if rc != _lib.NC_OK:
            raise _lib.NanoCallerHipError("nc_ctx_create(device=%d) failed with status %d" % (device, rc))
        self.ctx = ctx
        self._loaded = {}
        self._pinned = _PinnedPool()
        self.use_torch_stream()

    def close(self):
        if getattr(self, "ctx", None):
            self.L.nc_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != _lib.NC_OK:
            msg = self.L.nc_last_error(self.ctx)
            raise _lib.NanoCallerHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def use_torch_stream(self):
        """The library's kernels go to torch's current stream of this thread.  Where that is the legacy default stream, the thread is first moved to
        this engine's own torch stream: torch's kernels (fills, reductions, copies) and the library's then share ONE hardware queue.  On the default
        stream the library used a stream of its own beside it, and every torch kernel in a pass cost two cross-queue hand-overs (~0.4 ms of idle GPU
        per 11 ms contig pass: bench.py 56.0 -> 58.2 M sites/s; `rocprofv3 --kernel-trace` timelines by tools/one_pass.py).  Work the caller has
        already enqueued on the default stream is waited for.  NC_ONE_QUEUE=0: the old arrangement."""
        import os
        s = torch.cuda.current_stream(self.device)
        if s.cuda_stream == 0 and os.environ.get("NC_ONE_QUEUE", "1") != "0":
            if getattr(self, "_tstream", None) is None:
                self._tstream = torch.cuda.Stream(device=self.device)
            self._tstream.wait_stream(s)
            torch.cuda.set_stream(self._tstream)
            s = self._tstream
        self._check(self.L.nc_ctx_set_stream(self.ctx, C.c_void_p(s.cuda_stream)), "nc_ctx_set_stream")

    def _adopt_thread(self):
        """get_engine() from a thread that meets this engine for the first time: a thread on the legacy default stream is moved to the engine's own torch
        stream (one hardware queue); the CONTEXT's launch stream is left alone (get_engine used to call use_torch_stream(), which rewrites
        the shared ctx->stream -- a worker thread could retarget another thread's launches between its `with torch.cuda.stream(...)` and its kernel).
        Only an explicit use_torch_stream() retargets the context."""
        import os
        s = torch.cuda.current_stream(self.device)
        if s.cuda_stream == 0 and os.environ.get("NC_ONE_QUEUE", "1") != "0" and getattr(self, "_tstream", None) is not None:
            self._tstream.wait_stream(s)
            torch.cuda.set_stream(self._tstream)

    def set_cnn_precision(self, exact_fp32: bool):
        """False (default): fp16x3 split-precision trunk; True: exact fp32 MFMA trunk."""
        self._check(self.L.nc_set_cnn_precision(self.ctx, 1 if exact_fp32 else 0), "nc_set_cnn_precision")
        self.exact_fp32 = bool(exact_fp32)

    def set_tensor_format(self, int16: bool):
        """SNP tensors between snp_featurize and snp_forward as int16 (exact: every entry is a small integer; half the
        bytes) instead of the reference's float32.  Read by the split-precision trunk only."""
        self._check(self.L.nc_set_tensor_format(self.ctx, 1 if int16 else 0), "nc_set_tensor_format")
        self.x_int16 = bool(int16)

    TRUNK_KERNELS = {0: "k4_conv12", 1: "k5_trunk_h3", 2: "k5_trunk_p3", 3: "k5_trunk_lin"}

    def trunk_info(self):
        """-> (v_mfma_f32_16x16x32_f16 instructions per site, kernel name) of the SNP trunk the next snp_forward of this context launches (it depends
        on the tensor format and the precision mode): k5_trunk_p3 / _h3 13 x 24 + 10 x 27 + 8 x 18 = 726, k5_trunk_lin (int16 tensors) 613"""
        n, k = C.c_int32(), C.c_int32()
        self._check(self.L.nc_snp_trunk_info(self.ctx, C.byref(n), C.byref(k)), "nc_snp_trunk_info")
        return int(n.value), self.TRUNK_KERNELS[int(k.value)]

    def trunk_mfma_per_site(self) -> int:
        return self.trunk_info()[0]

    def enable_timing(self, on=True, trunk_only=False):
        """HIP-event timers: all stages, or (trunk_only) just the trunk kernel's launches, whose events ride on the
        kernel's dispatch packets and leave the stream undisturbed"""
        self._check(self.L.nc_enable_timing(self.ctx, (2 if trunk_only else 1) if on else 0), "nc_enable_timing")

    def last_ms(self, which):
        ms = C.c_float()
        self._check(self.L.nc_last_kernel_ms(self.ctx, which, C.byref(ms)), "nc_last_kernel_ms")
        return float(ms.value)

    def timing_sums(self):
        """-> (sum_ms float64 [6], count int64 [6]) over the calls since enable_timing(True): 0 scan, 1 featurize, 2 CNN
        stage, 3 indel, 4 trunk kernel launches (summed), 5 number of trunk launches."""
        sums, cnt = np.zeros(6, np.float64), np.zeros(6, np.int64)
        self._check(self.L.nc_timing_sums(self.ctx, _lib.npp(sums), _lib.npp(cnt)), "nc_timing_sums")
        return sums, cnt

    # ------------------------------------------------------------------ data movement
    def upload(self, hp: HostPack) -> DevicePack:
        dev = self.device
        ent_bytes = np.frombuffer(hp.tile_ent.tobytes(), np.uint8) if hp.tile_ent.size else np.zeros(16, np.uint8)
        dp = DevicePack(codes=torch.from_numpy(hp.codes).to(dev), tile_off=torch.from_numpy(hp.tile_off).to(dev),
                        tile_ent=torch.from_numpy(ent_bytes.copy()).to(dev), ref_code=torch.from_numpy(hp.ref_code).to(dev),
                        tile_size=hp.tile_size, tile_pos0=hp.tile_pos0, n_tiles=hp.n_tiles,
                        n_entries=int(hp.tile_ent.shape[0]), pos_lo=hp.pos_lo, pos_hi=hp.pos_hi)
        if hp.mates is not None:
            dp.mates = (torch.from_numpy(hp.mates[0]).to(dev), torch.from_numpy(hp.mates[1]).to(dev))
        if hp.ev_off is not None:
            z = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt) if a.size else np.zeros(1, dt)).to(dev)   # noqa: E731
            dp.events = dict(n_reads=int(hp.read_hap.shape[0]), ev_off=z(hp.ev_off, np.int32), ev_pos=z(hp.ev_pos, np.int32),
                             ev_len=z(hp.ev_len, np.int32), read_hap=z(hp.read_hap, np.uint8))
        return dp

    def load_weights(self, kind: int, w: Weights):
        if self._loaded.get(kind) == w.path:
            return
        if w.kind != kind:
            raise ValueError("weights %s are of kind %d, expected %d" % (w.path, w.kind, kind))
        flat = np.ascontiguousarray(w.flat, np.float32)
        self._check(self.L.nc_load_weights(self.ctx, kind, _lib.npp(flat), flat.size), "nc_load_weights")
        self._loaded[kind] = w.path

    # ------------------------------------------------------------------ K1
    def snp_scan(self, dp: DevicePack, chunks, *, mincov, min_allele_freq, threshold, haploid=False, async_fetch=False, between=None) -> SnpSites:
        """chunks: list of (start, end) inclusive, same contig, ascending.  `between` (callable): run after the scan's kernels are enqueued and before
        the host waits for the scan's totals (nc_snp_scan_begin / _end): what it enqueues keeps the GPU busy during that round trip."""
        cs = np.ascontiguousarray([c[0] for c in chunks], np.int32)
        ce = np.ascontiguousarray([c[1] for c in chunks], np.int32)
        scan_lo = max(1, int(cs.min()) - _lib.FLANK)
        scan_hi = int(ce.max()) + _lib.FLANK
        if scan_lo < dp.pos_lo or min(scan_hi, dp.pos_hi) < scan_lo:
            pass  # the pack simply has no data there
        params = _lib.ScanParamsC(mincov=int(mincov), min_allele_freq=float(min_allele_freq), nbr_t0=float(threshold[0]),
                                  nbr_t1=float(threshold[1]), haploid=1 if haploid else 0)
        pc = dp.c_struct()
        self._check(self.L.nc_snp_scan_begin(self.ctx, C.byref(pc), _ptr(dp.ref_code), dp.tile_pos0, dp.ref_code.numel(), scan_lo,
                                             scan_hi, C.byref(params), len(chunks), _lib.npp(cs), _lib.npp(ce)), "nc_snp_scan_begin")
        if between is not None:
            between()                                                # enqueued behind the scan's kernels, ahead of the host's wait for its totals
        n_nbr, n_cand, n_sites = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.L.nc_snp_scan_end(self.ctx, C.byref(n_nbr), C.byref(n_cand), C.byref(n_sites)), "nc_snp_scan_end")
        N = n_sites.value
        # pinned host buffers (cached by torch's host allocator): the four copies run at PCIe speed, one sync
        hb, h = self._pinned.get((4, max(N, 1)), torch.int32)
        ptr = [C.c_void_p(hb[i].data_ptr()) for i in range(4)]
        if async_fetch:
            # copies go to the copy stream NOW: they run in the host's turn-around before the featuriser is launched (issued behind
            # that launch they collide with the featuriser and spill into the kernels after it: measured, worse)
            self._check(self.L.nc_snp_scan_fetch_async(self.ctx, self._copy_stream_ptr(), None, ptr[0], ptr[1], ptr[2], ptr[3]),
                        "nc_snp_scan_fetch_async")
        else:
            self._check(self.L.nc_snp_scan_fetch(self.ctx, None, ptr[0], ptr[1], ptr[2], ptr[3]), "nc_snp_scan_fetch")
        return SnpSites(n_sites=N, n_nbr=n_nbr.value, pos=h[0, :N], chunk=h[1, :N], dp=h[2, :N], alt=h[3, :N])

    def fetch_nbr_sites(self, n_nbr) -> np.ndarray:
        out = np.empty(n_nbr, np.int32)
        self._check(self.L.nc_snp_scan_fetch(self.ctx, _lib.npp(out), None, None, None, None), "nc_snp_scan_fetch")
        return out

    # ------------------------------------------------------------------ K2-K4
    def snp_featurize(self, dp: DevicePack, sites: SnpSites, *, seq, maxcov, min_nbr_sites=1) -> SnpSites:
        N = sites.n_sites
        dev = self.device
        if dp.mates is not None and not getattr(self, "x_int16", False):
            # alignments that share read names are keyed by name in the int16-tensor kernel only: run it, hand the tensors on as float32 (exact)
            self.set_tensor_format(int16=True)
            try:
                self.snp_featurize(dp, sites, seq=seq, maxcov=maxcov, min_nbr_sites=min_nbr_sites)
            finally:
                self.set_tensor_format(int16=False)
            sites.x = sites.x.to(torch.float32)
            return sites
        # one allocation carved into the six outputs (this sits between the scan's totals and the featuriser's launch: the GPU idles)
        i16 = getattr(self, "x_int16", False)
        xb = N * 1025 * (2 if i16 else 4)
        a16 = lambda v: (v + 15) & ~15     # noqa: E731
        o_ref = a16(xb); o_fwd = a16(o_ref + 4 * N); o_rev = a16(o_fwd + 16 * N); o_dep = a16(o_rev + 16 * N); o_val = a16(o_dep + 4 * N)
        buf = torch.empty(o_val + N + 16, dtype=torch.uint8, device=dev)
        sites.x = buf[:xb].view(torch.int16 if i16 else torch.float32).view(N, 5, 41, 5)
        sites.ref_code = buf[o_ref:o_ref + 4 * N].view(torch.int32)
        sites.fwd_dp = buf[o_fwd:o_fwd + 16 * N].view(torch.int32).view(N, 4)
        sites.rev_dp = buf[o_rev:o_rev + 16 * N].view(torch.int32).view(N, 4)
        sites.depth = buf[o_dep:o_dep + 4 * N].view(torch.int32)
        sites.valid = buf[o_val:o_val + N]
        pc = dp.c_struct()
        if dp.mates is not None:                                     # alignments that share read names: the featuriser keys them by name
            self._check(self.L.nc_snp_set_mates(self.ctx, int(dp.mates[0].numel()), _ptr(dp.mates[0]), _ptr(dp.mates[1])), "nc_snp_set_mates")
        try:
            self._check(self.L.nc_snp_featurize(self.ctx, C.byref(pc), _ptr(dp.ref_code), dp.tile_pos0, dp.ref_code.numel(),
                                                _lib.SEQ_MODES[seq], int(maxcov), int(min_nbr_sites), _ptr(sites.x),
                                                _ptr(sites.ref_code), _ptr(sites.fwd_dp), _ptr(sites.rev_dp), _ptr(sites.depth),
                                                _ptr(sites.valid)), "nc_snp_featurize")
        finally:
            if dp.mates is not None:
                self.L.nc_snp_set_mates(self.ctx, 0, None, None)
        return sites

    def snp_scale(self, sites: SnpSites, n_chunks, train_coverage, per_site=False, async_fetch=False):
        """-> (scale f64 [N] device tensor, chunk_depth float64 host [n_chunks]).  async_fetch: the chunk depths arrive in
        pinned memory through the copy stream (valid after wait_copies() / a later copy_event()), and the compute stream
        is not synchronised between the featuriser and the CNN."""
        scale = torch.empty(sites.n_sites, dtype=torch.float64, device=self.device)
        if not async_fetch:
            cd = np.zeros(n_chunks, np.float64)
            self._check(self.L.nc_snp_scale(self.ctx, _ptr(sites.depth), _ptr(sites.valid), float(train_coverage),
                                            1 if per_site else 0, _ptr(scale), _lib.npp(cd)), "nc_snp_scale")
            return scale, cd
        hb, cd = self._pinned.get((max(int(n_chunks), 1),), torch.float64)
        self._check(self.L.nc_snp_scale(self.ctx, _ptr(sites.depth), _ptr(sites.valid), float(train_coverage),
                                        1 if per_site else 0, _ptr(scale), None), "nc_snp_scale")
        self._check(self.L.nc_snp_chunk_depth_async(self.ctx, self._copy_stream_ptr(), C.c_void_p(hb.data_ptr())), "nc_snp_chunk_depth_async")
        return scale, cd[:int(n_chunks)]

    # ------------------------------------------------------------------ K5 / K9
    def x_limit(self, kind) -> float:
        """largest |scaled input value| for which the loaded model's weights prove that the split-precision trunk stays inside the
        fp16 range (nc_cnn_x_limit)"""
        v = C.c_float()
        self._check(self.L.nc_cnn_x_limit(self.ctx, kind, C.byref(v)), "nc_cnn_x_limit")
        return float(v.value)

    def snp_forward_guarded(self, kind, x, ref_code, scale, scale_mode=0):
        """snp_forward on the split-precision trunk with its range guard closed: sites whose scaled tensor exceeds x_limit are re-run
        on the exact fp32 trunk.  -> (probs, gt, number of re-run sites); synchronous (tests, small batches; call_chunks does the same
        asynchronously)."""
        n = int(x.shape[0])
        flags = torch.zeros(max(n, 1), dtype=torch.uint8, device=self.device)
        probs, gt = self.snp_forward(kind, x, ref_code, scale, scale_mode, range_flags=flags)
        idx = torch.nonzero(flags[:n]).squeeze(1)
        if idx.numel():
            self.range_rerun(kind, x, ref_code, scale, scale_mode, idx, probs, gt)
        return probs, gt, int(idx.numel())

    def range_rerun(self, kind, x, ref_code, scale, scale_mode, idx, probs, gt):
        """the sites `idx` (device int64) once more on the exact fp32 MFMA trunk; their rows of probs / gt (device tensors) are replaced"""
        was_exact, was_i16 = getattr(self, "exact_fp32", False), getattr(self, "x_int16", False)
        xs = x[idx].to(torch.float32).contiguous()                   # the exact trunk reads the reference's float32 layout
        self.set_tensor_format(int16=False)
        self.set_cnn_precision(exact_fp32=True)
        try:
            p2, g2 = self.snp_forward(kind, xs, ref_code[idx].contiguous(), scale[idx].contiguous() if scale is not None else None, scale_mode,
                                      want_gt=gt is not None)
        finally:
            self.set_cnn_precision(exact_fp32=was_exact)
            self.set_tensor_format(int16=was_i16)
        probs[idx] = p2
        if gt is not None and g2 is not None:
            gt[idx] = g2

    def snp_forward(self, kind, x, ref_code, scale, scale_mode=0, want_gt=True, drain=False, range_flags=None):
        """-> (probs, gt) device tensors; with drain=True also (probs_host, gt_host) numpy arrays in pinned memory that
        the copy stream fills batch by batch while the next batch computes (wait_copies() before reading them).
        range_flags (uint8 device tensor [n], zeroed): the split-precision trunk marks the sites whose scaled tensor exceeds the
        model's x_limit (nc_cnn_range_watch)."""
        n = int(x.shape[0])
        if range_flags is not None:
            self._check(self.L.nc_cnn_range_watch(self.ctx, C.c_void_p(range_flags.data_ptr())), "nc_cnn_range_watch")
        try:
            return self._snp_forward(kind, x, ref_code, scale, scale_mode, want_gt, drain, n)
        finally:
            if range_flags is not None:
                self.L.nc_cnn_range_watch(self.ctx, None)

    def _snp_forward(self, kind, x, ref_code, scale, scale_mode, want_gt, drain, n):
        probs = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        gt = torch.empty((n, 2), dtype=torch.float32, device=self.device) if (want_gt and kind == _lib.MODEL_SNP) else None
        if not drain:
            self._check(self.L.nc_snp_forward(self.ctx, kind, n, _ptr(x), _ptr(ref_code), _ptr(scale), int(scale_mode),
                                              _ptr(probs), _ptr(gt)), "nc_snp_forward")
            return probs, gt
        hp, hp_np = self._pinned.get((n, 4), torch.float32)
        hg, hg_np = self._pinned.get((n, 2), torch.float32) if gt is not None else (None, None)
        self._check(self.L.nc_snp_forward_drain(self.ctx, kind, n, _ptr(x), _ptr(ref_code), _ptr(scale), int(scale_mode),
                                                _ptr(probs), _ptr(gt), self._copy_stream_ptr(), C.c_void_p(hp.data_ptr()),
                                                C.c_void_p(hg.data_ptr()) if hg is not None else None), "nc_snp_forward_drain")
        return probs, gt, hp_np, hg_np

    def indel_forward(self, kind, x):
        n = int(x.shape[0])
        nout = 4 if kind == _lib.MODEL_INDEL else 1
        probs = torch.empty((n, nout), dtype=torch.float32, device=self.device)
        self._check(self.L.nc_indel_forward(self.ctx, kind, n, _ptr(x), _ptr(probs)), "nc_indel_forward")
        return probs

    def indel_scan(self, dp: DevicePack, start, end, *, mincov, win_size, small_win_size, ins_t, del_t, excl=None, haploid=False, impute=False):
        """K7 -> int8 [end-start+1] per-column decision (-1 none, 0 long-window rule, 1 small-window rule, 2 impute_indel_phase
        candidate when impute=True)."""
        if dp.events is None:
            raise ValueError("this read pack carries no indel events / haplotype tags")
        ev = dp.events
        evc = _lib.IndelEventsC(n_reads=ev["n_reads"], ev_off=ev["ev_off"].data_ptr(), ev_pos=ev["ev_pos"].data_ptr(),
                                ev_len=ev["ev_len"].data_ptr(), read_hap=ev["read_hap"].data_ptr())
        prm = _lib.IndelScanParamsC(mincov=int(mincov), win_size=int(win_size), small_win_size=int(small_win_size),
                                    ins_t=float(ins_t), del_t=float(del_t), haploid=1 if haploid else 0, impute=1 if impute else 0)
        lo = max(1, int(start))
        out = np.empty(int(end) - lo + 1, np.int8)
        pc = dp.c_struct()
        self._check(self.L.nc_indel_scan(self.ctx, C.byref(pc), C.byref(evc), _ptr(excl), lo, int(end), C.byref(prm),
                                         _lib.npp(out)), "nc_indel_scan")
        return out

    def indel_scan_batch(self, dp: DevicePack, chunks, *, mincov, win_size, small_win_size, ins_t, del_t, excl=None, haploid=False, impute=False):
        """K7 for a list of (start, end) chunks of one contig -> list of int8 arrays (one per chunk); ascending chunk lists run in
        the same kernel launches (chunk = a grid dimension)."""
        if dp.events is None:
            raise ValueError("this read pack carries no indel events / haplotype tags")
        ev = dp.events
        evc = _lib.IndelEventsC(n_reads=ev["n_reads"], ev_off=ev["ev_off"].data_ptr(), ev_pos=ev["ev_pos"].data_ptr(),
                                ev_len=ev["ev_len"].data_ptr(), read_hap=ev["read_hap"].data_ptr())
        prm = _lib.IndelScanParamsC(mincov=int(mincov), win_size=int(win_size), small_win_size=int(small_win_size),
                                    ins_t=float(ins_t), del_t=float(del_t), haploid=1 if haploid else 0, impute=1 if impute else 0)
        starts = np.ascontiguousarray([c[0] for c in chunks], np.int32)
        ends = np.ascontiguousarray([c[1] for c in chunks], np.int32)
        ncol = ends.astype(np.int64) - np.maximum(starts, 1) + 1
        off = np.zeros(len(chunks) + 1, np.int64)
        np.cumsum(ncol, out=off[1:])
        out = np.empty(int(off[-1]), np.int8)
        pc = dp.c_struct()
        self._check(self.L.nc_indel_scan_batch(self.ctx, C.byref(pc), C.byref(evc), _ptr(excl), len(chunks), _lib.npp(starts),
                                               _lib.npp(ends), C.byref(prm), _lib.npp(out), _lib.npp(off)), "nc_indel_scan_batch")
        return [out[off[k]:off[k + 1]] for k in range(len(chunks))]

    def star_msa_tensor(self, read_sets, refs, *, open_=None, extend=None, match=None, mismatch=None, max_cols=None, want_rows=False, cns_as_str=False):
        """Device star alignment (nc_star_msa_tensor) of many read sets at once + the rows -> tensor kernel.
        read_sets[s]: list of read strings, refs[s]: reference window string.
        -> (x f32 [S,5,128,2] device, cns list of uint8 arrays (or, cns_as_str, AGTC strings) with gaps removed, n_cols int32 [S]
            [, rows list of uint8 [n_reads, n_cols], ref_rows list of uint8 [n_cols]])"""
        S = len(read_sets)
        if S == 0:
            x = torch.empty((0, 5, 128, 2), dtype=torch.float32, device=self.device)
            return (x, [], np.zeros(0, np.int32)) + (([], []) if want_rows else ())
        n_reads = np.fromiter(map(len, read_sets), np.int64, S)
        set0 = np.zeros(S + 1, np.int32)
        np.cumsum(n_reads, out=set0[1:])
        flat = [q for rs in read_sets for q in rs]
        rlen = np.fromiter(map(len, flat), np.int64, len(flat))
        read_off = np.zeros(len(flat) + 1, np.int32)
        np.cumsum(rlen, out=read_off[1:])
        reflen = np.fromiter(map(len, refs), np.int64, S)
        ref_off = np.zeros(S + 1, np.int32)
        np.cumsum(reflen, out=ref_off[1:])
        raw_reads, raw_refs = "".join(flat).encode(), "".join(refs).encode()
        # every read base can at most add one column to its set
        per_set = np.zeros(S, np.int64)
        nz = n_reads > 0
        if len(flat):
            per_set[nz] = np.add.reduceat(rlen, set0[:-1][nz])
        cap = reflen + per_set
        mc = int(max_cols) if max_cols else int(cap.max())
        return self.star_msa_tensor_flat(S, raw_reads, read_off, set0, raw_refs, ref_off, mc, open_=open_, extend=extend, match=match,
                                         mismatch=mismatch, want_rows=want_rows, cns_as_str=cns_as_str, _n_reads=n_reads, _cap=cap)

    def star_msa_tensor_flat(self, S, reads, read_off, set_read0, refs, ref_off, max_cols, *, open_=None, extend=None, match=None, mismatch=None,
                             want_rows=False, cns_as_str=False, _n_reads=None, _cap=None, al_dup=None):
        """The same on flat host buffers (what nc_indel_pass2_sets produces): `reads` / `refs` bytes objects or raw pointers
        (ints / c_void_p), read_off / set_read0 / ref_off int32 arrays or pointers.  al_dup (int32 array or pointer, optional):
        nc_pass2_arrays.al_dup -- alignments that repeat an earlier one are not computed again."""
        open_, extend, match, mismatch = [d if v is None else v for v, d in zip((open_, extend, match, mismatch), _lib.STAR_SCORING)]
        x = torch.empty((S, 5, 128, 2), dtype=torch.float32, device=self.device)
        mc = int(max_cols)
        cns = np.empty((S, mc), np.uint8)
        ncols = np.empty(S, np.int32)
        ptr = lambda a: _lib.npp(a) if isinstance(a, np.ndarray) else a      # noqa: E731
        rows = rr = roff = rroff = None
        if want_rows:
            roff = np.zeros(S + 1, np.int64)
            np.cumsum(_n_reads * _cap, out=roff[1:])
            rroff = np.zeros(S + 1, np.int64)
            np.cumsum(_cap, out=rroff[1:])
            rows, rr = np.empty(max(int(roff[-1]), 1), np.uint8), np.empty(max(int(rroff[-1]), 1), np.uint8)
        self._check(self.L.nc_star_msa_tensor_dup(self.ctx, S, reads, ptr(read_off), ptr(set_read0), refs, ptr(ref_off),
                                                  int(open_), int(extend), int(match), int(mismatch), mc, _ptr(x), _lib.npp(cns), _lib.npp(ncols),
                                                  _lib.npp(rows), _lib.npp(roff), _lib.npp(rr), _lib.npp(rroff), ptr(al_dup)), "nc_star_msa_tensor")
        if int(ncols.max()) > mc and not want_rows:
            # the caller's column bound was an estimate (consensus columns beyond it are cut): once more with the real maximum
            return self.star_msa_tensor_flat(S, reads, read_off, set_read0, refs, ref_off, int(ncols.max()), open_=open_, extend=extend,
                                             match=match, mismatch=mismatch, cns_as_str=cns_as_str, al_dup=al_dup)
        # consensus with the gap symbols removed
        if cns_as_str:
            # natively (worker threads): the [S, mc] matrix of a contig arm is tens of megabytes, numpy masks over it cost more than the alignments
            cut = np.empty(S + 1, np.int64)
            flat = np.empty(max(S * mc, 1), np.uint8)
            self._check(self.L.nc_consensus_strings(_lib.npp(cns), S, mc, _lib.npp(ncols), _lib.npp(flat), _lib.npp(cut)), "nc_consensus_strings")
            cl = cut.tolist()
            big = flat[:cl[-1]].tobytes().decode("ascii")
            out_cns = [big[a:b] for a, b in zip(cl, cl[1:])]
        else:
            keep = (np.arange(mc, dtype=np.int32)[None, :] < np.minimum(ncols, mc)[:, None]) & (cns != 4)
            cnt = keep.sum(1)
            cut = np.zeros(S + 1, np.int64)
            np.cumsum(cnt, out=cut[1:])
            flat_cns = cns[keep]
            out_cns = [flat_cns[cut[k]:cut[k + 1]] for k in range(S)]
        if not want_rows:
            return x, out_cns, ncols
        out_rows = [rows[roff[s]:roff[s] + int(_n_reads[s]) * int(ncols[s])].reshape(int(_n_reads[s]), int(ncols[s])) for s in range(S)]
        out_rr = [rr[rroff[s]:rroff[s] + int(ncols[s])] for s in range(S)]
        return x, out_cns, ncols, out_rows, out_rr

    def indel_tensor(self, rows_list, ref_rows_list):
        """rows_list[s]: uint8 [n_rows, n_cols] aligned symbols 0..4; ref_rows_list[s]: uint8 [n_cols].
        -> (x f32 [S,5,128,2] device, cns list of uint8 arrays with gaps removed)"""
        S = len(rows_list)
        dev = self.device
        n_rows = np.array([r.shape[0] for r in rows_list], np.int32)
        n_cols = np.array([r.shape[1] for r in rows_list], np.int32)
        row_off = np.zeros(S, np.int64)
        ref_off = np.zeros(S, np.int64)
        if S:
            np.cumsum((n_rows.astype(np.int64) * n_cols)[:-1], out=row_off[1:])
            np.cumsum(n_cols.astype(np.int64)[:-1], out=ref_off[1:])
        max_cols = max(128, int(n_cols.max()) if S else 128)
        rows = np.concatenate([np.ascontiguousarray(r, np.uint8).ravel() for r in rows_list]) if S else np.zeros(1, np.uint8)
        refs = np.concatenate([np.ascontiguousarray(r, np.uint8).ravel() for r in ref_rows_list]) if S else np.zeros(1, np.uint8)
        t = lambda a: torch.from_numpy(a).to(dev)   # noqa: E731
        d_rows, d_refs, d_ro, d_fo, d_nr, d_nc = t(rows), t(refs), t(row_off), t(ref_off), t(n_rows), t(n_cols)
        x = torch.empty((S, 5, 128, 2), dtype=torch.float32, device=dev)
        cns = torch.empty((S, max_cols), dtype=torch.uint8, device=dev)
        self._check(self.L.nc_indel_tensor(self.ctx, S, _ptr(d_rows), _ptr(d_ro), _ptr(d_nr), _ptr(d_nc), _ptr(d_refs),
                                           _ptr(d_fo), max_cols, _ptr(x), _ptr(cns)), "nc_indel_tensor")
        cns_h = cns.cpu().numpy()
        out = []
        for s in range(S):
            row = cns_h[s, :n_cols[s]]
            out.append(row[row != 4])
        return x, out

    def to_host(self, tensors):
        """Copy device tensors to (cached) pinned host memory with ONE synchronisation; -> list of numpy views
        that own their pinned storage through the returned arrays' base tensors."""
        outs = []
        for t in tensors:
            if t is None:
                outs.append(None)
                continue
            h, h_np = self._pinned.get(t.shape, t.dtype)
            h.copy_(t, non_blocking=True)
            outs.append(h_np)
        torch.cuda.current_stream(self.device).synchronize()
        return outs

    # ------------------------------------------------------------------ copy stream (results drain overlapping compute)
    def _copy_stream(self):
        if getattr(self, "_cstream", None) is None:
            self._cstream = torch.cuda.Stream(device=self.device)
        return self._cstream

    def _copy_stream_ptr(self):
        return C.c_void_p(self._copy_stream().cuda_stream)

    def to_host_async(self, tensors):
        """Enqueue D2H copies of device tensors on the copy stream, ordered after everything already enqueued on the
        compute stream; -> list of numpy views of pinned memory, valid after wait_copies()."""
        cs = self._copy_stream()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        cs.wait_event(ev)
        outs = []
        with torch.cuda.stream(cs):
            for t in tensors:
                if t is None:
                    outs.append(None)
                    continue
                h, h_np = self._pinned.get(t.shape, t.dtype)
                t = t.contiguous()
                # small blocks go by copy kernel, bulk by the copy engine (see nc_d2h_async)
                self._check(self.L.nc_d2h_async(self.ctx, C.c_void_p(cs.cuda_stream), C.c_void_p(h.data_ptr()), C.c_void_p(t.data_ptr()),
                                                t.numel() * t.element_size()), "nc_d2h_async")
                t.record_stream(cs)
                outs.append(h_np)
        return outs

    def copy_event(self):
        """event that completes when the copies enqueued so far on the copy stream are done"""
        ev = torch.cuda.Event()
        ev.record(self._copy_stream())
        return ev

    def wait_copies(self):
        self._copy_stream().synchronize()

    def sync(self):
        torch.cuda.synchronize(self.device)


_engines = {}


def get_engine(device: int = 0) -> Engine:
    if device not in _engines:
        _engines[device] = Engine(device)
    else:
        _engines[device]._adopt_thread()                 # (a thread that meets the engine for the first time is moved to its stream; ctx->stream untouched)
    return _engines[device]


def local_device(devices=None, rank=0) -> int:
    """GPU of this process: one process per GPU.  `devices` (a sequence) is indexed by the local rank; by default the local
    rank itself (LOCAL_RANK from torchrun, else the global rank) modulo the number of visible devices."""
    import os
    local = int(os.environ.get("LOCAL_RANK", rank))
    if devices:
        return int(devices[local % len(devices)])
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return local % n if n > 0 else 0
