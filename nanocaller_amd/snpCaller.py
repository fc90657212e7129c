"""Host-side mirror of the reference's SNP caller (nanocaller_src/snpCaller.py).

Same entry points, `params` keys, model names, VCF text and output file names; the featurisation and
the CNN run on the MI355X through libnanocaller_hip.so.  Differences by design:

* one process per GPU instead of `cpu` worker processes; a worker takes ALL chunks of a contig as one
  device batch (the contig's alignments stay resident in HBM, scanned once);
* genotype rules (snpCaller.py:113-198) stay on the host: `snp_vcf_lines` / `snp_vcf_lines_haploid`;
* bcftools/bgzip are optional: records are written position-sorted and BGZF-compressed natively.
"""
from __future__ import annotations

import datetime
import os
import queue
import shutil
import struct
import sys
import zlib

import numpy as np
import torch

from . import _lib
from .engine import get_engine
from .generate_SNP_pileups import device_pack_for, release_contig
from .weights import Weights, get_SNP_model  # noqa: F401  (re-exported, same name as the reference)

num_to_base_map = {0: 'A', 1: 'G', 2: 'T', 3: 'C'}                 # snpCaller.py:14
_B = "AGTC"


def _qual(p, cap=99.0, mult=-10.0):
    """min(cap, mult*log10(1e-10 + 1 - p)) evaluated in float64 (numpy<2 scalar semantics)."""
    return min(cap, mult * np.log10(1e-10 + 1 - float(p)))


def snp_vcf_lines(chrom, pos, ref_idx, probs, dp, freq, fwd_dp, rev_dp):
    """Diploid genotype rules + VCF text (snpCaller.py:113-163).
    probs float32 [N,4] = class-1 probability of the A,G,T,C heads."""
    probs = np.asarray(probs, np.float32)
    order = np.argsort(probs, axis=1)                               # :118 (same call => same tie behaviour)
    k = np.sum(probs >= 0.5, axis=1)                                # :122
    out = []
    for j in range(len(pos)):
        pr = probs[j]
        info = 'PR=' + ','.join("{:.4f}".format(x) for x in pr[[0, 3, 1, 2]]) + ";FQ={:.4f}".format(freq[j])   # :127
        r = int(ref_idx[j])
        d = int(dp[j])
        ref_dp = (fwd_dp[j][r], rev_dp[j][r])
        p1, p2 = int(order[j, -1]), int(order[j, -2])
        head = '%s\t%d\t.\t%s\t' % (chrom, pos[j], _B[r])
        fmt = 'GT:DP:VF:AD:ADF:ADR'
        if k[j] >= 2:
            if p1 == r:                                             # :132
                a = (fwd_dp[j][p2], rev_dp[j][p2])
                out.append(head + '%s\t%.3f\t%s\t%s\t%s\t%s:%d:%.4f:%d,%d:%d,%d:%d,%d\n' % (
                    _B[p2], _qual(pr[p2]), 'PASS', info, fmt, '0/1', d, sum(a) / d, sum(ref_dp), sum(a), ref_dp[0], a[0],
                    ref_dp[1], a[1]))
            elif p2 == r and pr[p2] >= 0.5:                         # :138
                a = (fwd_dp[j][p1], rev_dp[j][p1])
                out.append(head + '%s\t%.3f\t%s\t%s\t%s\t%s:%d:%.4f:%d,%d:%d,%d:%d,%d\n' % (
                    _B[p1], _qual(pr[p2]), 'PASS', info, fmt, '0/1', d, sum(a) / d, sum(ref_dp), sum(a), ref_dp[0], a[0],
                    ref_dp[1], a[1]))
            elif p2 != r and p1 != r and pr[p2] >= 0.5:             # :143
                a1 = (fwd_dp[j][p1], rev_dp[j][p1])
                a2 = (fwd_dp[j][p2], rev_dp[j][p2])
                out.append(head + '%s,%s\t%.3f\t%s\t%s\t%s\t%s:%d:%.4f,%.4f:%d,%d,%d:%d,%d,%d:%d,%d,%d\n' % (
                    _B[p1], _B[p2], _qual(pr[p2]), 'PASS', info, fmt, '1/2', d, sum(a1) / d, sum(a2) / d, sum(ref_dp),
                    sum(a1), sum(a2), ref_dp[0], a1[0], a2[0], ref_dp[1], a1[1], a2[1]))
            # else: the reference writes nothing (k>=2 but the second allele is below 0.5 cannot happen)
        elif k[j] == 1 and r != p1 and pr[p1] >= 0.5:               # :150
            a = (fwd_dp[j][p1], rev_dp[j][p1])
            out.append(head + '%s\t%.3f\t%s\t%s\t%s\t%s:%d:%.4f:%d,%d:%d,%d:%d,%d\n' % (
                _B[p1], _qual(pr[p1]), 'PASS', info, fmt, '1/1', d, sum(a) / d, sum(ref_dp), sum(a), ref_dp[0], a[0],
                ref_dp[1], a[1]))
        elif k[j] == 1 and r == p1:                                 # :157
            out.append(head + '%s\t%.3f\t%s\t%s\t%s\t%s:%d:.:.:.:.\n' % ('.', _qual(pr[p1]), 'REF', info, fmt, './.', d))
        else:                                                       # :161
            out.append(head + '%s\t%.3f\t%s\t%s\t%s\t%s:%d:.:.:.:.\n' % ('.', 0, 'LOW', info, fmt, './.', d))
    return out


def snp_vcf_lines_haploid(chrom, pos, ref_idx, probs, dp, freq):
    """Haploid rules (snpCaller.py:184-198): arg-max of the 4-way softmax, PASS if it differs from ref."""
    probs = np.asarray(probs, np.float32)
    pred = np.argmax(probs, 1)
    out = []
    for j in range(len(pos)):
        pr = probs[j]
        p = int(pred[j])
        r = int(ref_idx[j])
        info = 'PR=' + ','.join("{:.4f}".format(x) for x in pr[[0, 3, 1, 2]]) + ";FQ={:.4f}".format(freq[j])
        out.append('%s\t%d\t.\t%s\t%s\t%.3f\t%s\t%s\tGT:DP:VF:AD:ADF:ADR\t%s:%d:%.4f:.:.:.\n' % (
            chrom, pos[j], _B[r], _B[p], _qual(pr[p], 999.0, -100.0), 'PASS' if p != r else 'REF', info, '1/1', int(dp[j]),
            freq[j]))
    return out


def argsort4(probs):
    """np.argsort(probs, axis=1) for [n, 4] float32 rows: the library sorts (threaded), numpy re-sorts only the rows that
    contain ties, so the result equals numpy's on this machine element for element (quirk E15)."""
    import ctypes as C
    L = _lib.lib()
    probs = np.ascontiguousarray(probs, np.float32)
    n = probs.shape[0]
    order = np.empty((n, 4), np.int32)
    ties = np.empty(max(n, 1), np.int64)
    nt = C.c_int64()
    rc = L.nc_argsort4(_lib.npp(probs), n, _lib.npp(order), C.byref(nt), _lib.npp(ties), ties.size)
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("nc_argsort4 failed (%d)" % rc)
    if nt.value:
        t = ties[:nt.value]
        order[t] = np.argsort(probs[t], axis=1)
    return order


def snp_vcf_text(chrom, pos, ref_idx, probs, dp, freq, fwd_dp=None, rev_dp=None, haploid=False, as_array=False, out=None):
    """Same records as snp_vcf_lines / snp_vcf_lines_haploid, formatted by the library's native formatter
    (nc_snp_vcf_format); ties in the allele order resolve exactly as in the Python path (argsort4, E15).
    -> bytes, or with as_array=True a uint8 array (buffer protocol: file.write() takes it without another copy).
    `out`: optional uint8 scratch array to format into (reused by callers that write the text out immediately: saves the
    page faults of a fresh 400 B/record buffer per call)."""
    import ctypes as C
    L = _lib.lib()
    n = len(pos)
    i32 = lambda a: np.ascontiguousarray(a, np.int32)          # noqa: E731
    probs = np.ascontiguousarray(probs, np.float32)
    order = None if haploid else argsort4(probs)
    pos_, ref_, dp_ = i32(pos), i32(ref_idx), i32(dp)
    freq_ = np.ascontiguousarray(freq, np.float64)
    fwd_ = None if haploid else i32(fwd_dp)
    rev_ = None if haploid else i32(rev_dp)
    cap = (400 + len(chrom)) * max(n, 1) + 1024
    if out is None or out.size < cap:
        out = np.empty(cap, np.uint8)
    cap = out.size
    nb = C.c_int64()
    rc = L.nc_snp_vcf_format(chrom.encode(), n, _lib.npp(pos_), _lib.npp(ref_), _lib.npp(probs), _lib.npp(order),
                             _lib.npp(dp_), _lib.npp(freq_), _lib.npp(fwd_), _lib.npp(rev_), 1 if haploid else 0,
                             _lib.npp(out), cap, C.byref(nb))
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("nc_snp_vcf_format failed (%d)" % rc)
    return out[:nb.value] if as_array else out[:nb.value].tobytes()


VCF_HEADER = (                                                      # snpCaller.py:259-276
    '##fileformat=VCFv4.2\n'
    '##FILTER=<ID=PASS,Description="All filters passed">\n'
    '##FILTER=<ID=LOW,Description="All alleles have probability less than 50%.">\n'
    '##FILTER=<ID=REF,Description="Homozygous Reference. Only reference allele has greater than 50% probability. '
    'All alternative alleles having probability less than 50%.">\n'
    '{contigs}'
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
    '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Depth">\n'
    '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="Allelic depths for the ref and alt alleles in the order listed">\n'
    '##FORMAT=<ID=ADF,Number=R,Type=Integer,Description="Allelic depths on forward strand for the ref and alt alleles in the order listed">\n'
    '##FORMAT=<ID=ADR,Number=R,Type=Integer,Description="Allelic depths on reverse strand for the ref and alt alleles in the order listed">\n'
    '##FORMAT=<ID=VF,Number=A,Type=Float,Description="Alternative allele frequency in the order listed">\n'
    '##INFO=<ID=PR,Number=4,Type=Float,Description="Probability of presence of alleles A, C, G and T, in the given order. '
    'Probability of each base is out of 1, independent of each other.">\n'
    '##INFO=<ID=FQ,Number=1,Type=Float,Description="Maximum frequency of non-reference base.">\n'
    '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t{sample}\n')


# keys of the `params` dict this module reads (all of them are in the dict NanoCaller:28-35 builds)
PARAM_KEYS = frozenset(['chunks_list', 'regions_list', 'sam_path', 'fasta_path', 'mincov', 'maxcov', 'min_allele_freq', 'min_nbr_sites',
                        'threshold', 'snp_model', 'vcf_path', 'prefix', 'sample', 'seq', 'supplementary', 'exclude_bed',
                        'suppress_progress', 'disable_coverage_normalization'])

_WEIGHTS = {}


def _weights(path):
    """parsed .ncw files, cached per path"""
    if path not in _WEIGHTS:
        _WEIGHTS[path] = Weights(path)
    return _WEIGHTS[path]


class PendingCall:
    """call_chunks(defer=True): everything is enqueued, the result copies may still be in flight; result() waits for them."""

    def __init__(self, finish):
        self._finish, self._res = finish, None

    def result(self):
        if self._finish is not None:
            self._res, self._finish = self._finish(), None
        return self._res


_PENDING_CNN = {}                  # device -> the CNN launch of the previous deferred call_chunks (enqueued inside the next call's scan, or on demand)


def flush_pending_cnn(device=0):
    """enqueue the CNN a pipelined call_chunks left pending on `device` (no-op when there is none)"""
    f = _PENDING_CNN.pop(device, None)
    if f is not None:
        f()


def call_chunks(params, chunks, device=0, dpk=None, defer=False, pipeline=None):
    """Run pileup featurisation + CNN for a list of chunks of ONE contig and ploidy on the GPU.
    `dpk`: alignments already resident in HBM (engine.DevicePack); default: packed from params['sam_path'].
    -> dict of host arrays (pos, chunk, ref, probs, gt, dp, freq, fwd_dp, rev_dp, chunk_depth).
    defer=True -> PendingCall: returns once the group's work is handed over; the caller starts the next group and collects result() afterwards.
    pipeline (NC_PIPE_CNN=1; OFF by default): this group's CNN is not enqueued by this call but INSIDE the next call's scan (nc_snp_scan_begin ->
    the pending CNN -> nc_snp_scan_end), or by result() / flush_pending_cnn() if no call follows: the host's round trip for the next scan's totals
    (~0.2 ms) then runs under 8 ms of CNN instead of beside an idle GPU.  Device order scan(k+1), CNN(k), tensors(k+1),
    scan(k+2), CNN(k+1) ...; same results.  Measured: 10.55 -> 11.2 ms per contig (the scan's and the featuriser's result copies -- blit kernels on this
    platform -- then collide with the trunk instead of running in the gap they filled): not the default (profiles/README.md, round 6)."""
    if pipeline is None:
        pipeline = bool(defer) and os.environ.get('NC_PIPE_CNN', '0') == '1'
    if not pipeline:
        flush_pending_cnn(device)                                    # keep the device order of calls that do not take part
    chrom = chunks[0]['chrom']
    ploidy = chunks[0]['ploidy']
    assert all(c['chrom'] == chrom and c['ploidy'] == ploidy for c in chunks)
    eng = get_engine(device)
    eng.use_torch_stream()
    if ploidy == 'diploid':
        path, train_cov = get_SNP_model(params['snp_model'])
        if path is None:
            print('Invalid SNP model name or path', flush=True)     # snpCaller.py:66-68
            sys.exit(1)
        kind = _lib.MODEL_SNP
    else:
        path, _ = get_SNP_model('haploid')
        train_cov = 30                                              # hap_train_coverage, snpCaller.py:73
        kind = _lib.MODEL_SNP_HAP
    eng.load_weights(kind, _weights(path))
    if dpk is None:
        # a rank that owns only part of a contig (shard.shard_plan) decodes and uploads only its span +- the scan flank
        from .generate_SNP_pileups import contig_span
        dpk = device_pack_for(params, chrom, device, span=contig_span(params['sam_path'], chrom, chunks))
    # Results drain on a second stream while the GPU keeps computing: the candidate arrays right after the scan, the
    # featuriser's per-site arrays during the CNN, and every CNN batch's probabilities while the next batch runs.
    sites = eng.snp_scan(dpk, [(c['start'], c['end']) for c in chunks], mincov=params['mincov'],
                         min_allele_freq=params['min_allele_freq'], threshold=params['threshold'],
                         haploid=(ploidy == 'haploid'), async_fetch=True, between=(lambda: flush_pending_cnn(device)) if pipeline else None)
    res = dict(chrom=chrom, ploidy=ploidy, n=0)
    if sites.n_sites == 0:
        eng.wait_copies()
        return PendingCall(lambda: res) if defer else res
    # the tensors stay on the device: int16 between featuriser and CNN (exact, half the HBM traffic) unless the exact-fp32
    # trunk, which reads the reference's float32 layout, has been selected
    eng.set_tensor_format(int16=not getattr(eng, "exact_fp32", False))
    eng.snp_featurize(dpk, sites, seq=params['seq'], maxcov=params['maxcov'], min_nbr_sites=params['min_nbr_sites'])
    all_valid = bool(sites.valid.all().item()) if params['min_nbr_sites'] > 1 else True   # default 1 never filters (:244)
    per_site = bool(params.get('disable_coverage_normalization'))
    scale, chunk_depth = eng.snp_scale(sites, len(chunks), train_cov, per_site=per_site, async_fetch=True)
    scan_copied = eng.copy_event()                                 # candidate arrays + chunk depths are on the host behind this
    # enqueued after the (short, latency-sensitive) scale kernel so that these copies run under the CNN, not beside it
    h_ref, h_fwd, h_rev, h_valid = eng.to_host_async([sites.ref_code, sites.fwd_dp, sites.rev_dp, None if all_valid else sites.valid])
    # range guard of the split-precision trunk (nc_cnn_range_watch): sites whose scaled tensor exceeds what the model's weights prove
    # safe for the fp16 range are marked and, in finish(), computed once more by the exact fp32 trunk
    guard = not getattr(eng, "exact_fp32", False)
    i16 = not getattr(eng, "exact_fp32", False)
    eng.set_tensor_format(int16=False)                             # the direct API keeps the reference's float32 default
    cnn = {}                                                       # filled when the CNN is enqueued: now, or inside the next call's scan

    def enqueue_cnn():
        eng.use_torch_stream()
        eng.set_tensor_format(int16=i16)
        try:
            flags_ = torch.zeros(sites.n_sites, dtype=torch.uint8, device=eng.device) if guard else None
            cnn['d_probs'], cnn['d_gt'], cnn['h_probs'], cnn['h_gt'] = eng.snp_forward(kind, sites.x, sites.ref_code, scale, scale_mode=1 if per_site else 0,
                                                                                   drain=True, range_flags=flags_)
            cnn['flags'] = flags_
            cnn['h_nflag'] = eng.to_host_async([flags_.sum(dtype=torch.int32).reshape(1)])[0] if guard else None
        finally:
            eng.set_tensor_format(int16=False)
        cnn['drained'] = eng.copy_event()                          # completes with this call's last result copy
    if pipeline:
        _PENDING_CNN[device] = enqueue_cnn
    else:
        enqueue_cnn()
    # host work that only needs the scan results runs under the CNN: freq = alt / n in float64 (:166)
    scan_copied.synchronize()
    freq = sites.alt.astype(np.float64) / sites.dp.astype(np.float64)
    keep_alive = (sites, scale, cnn)                               # device buffers the pending copies read from

    def finish():
        nonlocal keep_alive
        if 'drained' not in cnn:                                   # nobody enqueued this group's CNN yet (no call followed): now
            if _PENDING_CNN.get(device) is enqueue_cnn:
                flush_pending_cnn(device)
            else:
                enqueue_cnn()
        drained, h_nflag, flags, d_probs, d_gt, h_probs, h_gt = (cnn[k] for k in ('drained', 'h_nflag', 'flags', 'd_probs', 'd_gt', 'h_probs', 'h_gt'))
        drained.synchronize()
        n_rerun = int(h_nflag[0]) if h_nflag is not None else 0
        if n_rerun:
            idx = torch.nonzero(flags).squeeze(1)
            eng.use_torch_stream()
            eng.range_rerun(kind, sites.x, sites.ref_code, scale, 1 if per_site else 0, idx, d_probs, d_gt)
            ih = idx.cpu().numpy()
            h_probs[ih] = d_probs[idx].cpu().numpy()
            if h_gt is not None and d_gt is not None:
                h_gt[ih] = d_gt[idx].cpu().numpy()
        res['range_reruns'] = n_rerun
        keep_alive = None
        out = dict(pos=sites.pos, chunk=sites.chunk, ref=h_ref, probs=h_probs, gt=h_gt, dp=sites.dp, alt=sites.alt,
                   fwd_dp=h_fwd, rev_dp=h_rev, freq=freq)
        if not all_valid:
            m = h_valid.astype(bool)
            out = {k: (v[m] if v is not None else None) for k, v in out.items()}
        # host arrays keep the device dtypes (int32 / float32)
        res.update(out, n=int(out['pos'].shape[0]), chunk_depth=chunk_depth)
        return res
    return PendingCall(finish) if defer else finish()


def _prepare_wire(params, chrom, grp):
    """host half of a group's ingest, on a worker thread: decode the contig (or this rank's span of it) and build its wire pack
    in page-locked memory (what device_pack() does synchronously)"""
    from .generate_SNP_pileups import _check_supported, _exclude_rows, _resolve, contig_span
    from .wire import build_wire_from_world
    span = contig_span(params['sam_path'], chrom, grp)
    world = _resolve(params['sam_path'], chrom, params.get('fasta_path'), span)
    _check_supported(world, params['sam_path'], chrom, bool(params.get('supplementary')), by_name=True)
    kw = dict(pos_lo=span[0], pos_hi=span[1]) if span else {}
    return build_wire_from_world(world, supplementary=bool(params.get('supplementary')), exclude=_exclude_rows(params, chrom), **kw)


def _prepare_device(dbam, params, chrom, grp, ref=None):
    """host half of a group's ingest on the device route (device_bam.py), on a worker thread: the contig's reference sequence, which of its
    alignments the pileup keeps, the tile index -- from the per-record fields the device extracted when the file was loaded"""
    from .bam import read_fasta_bytes
    from .generate_SNP_pileups import _exclude_rows
    # (contig_span's rule, on the lengths the loader already holds: no second open of the BAM)
    length = dbam.ref_lengths[dbam.ref_names.index(chrom)] if chrom in dbam.ref_names else 0
    lo = max(1, min(c['start'] for c in grp) - _lib.FLANK)
    hi = min(length, max(c['end'] for c in grp) + _lib.FLANK)
    span = None if (hi - lo + 1) >= 0.9 * length else (lo, hi)
    ref = ref.result() if ref is not None else read_fasta_bytes(params['fasta_path'], chrom)
    return dbam.prepare(chrom, ref, supplementary=bool(params.get('supplementary')), exclude=_exclude_rows(params, chrom), span=span)


def caller(params, chunks_Q, counter_Q, snp_files, device=0, worker_id=1):
    """Worker with the reference's signature (snpCaller.py:57): drains `chunks_Q`, writes
    <intermediate_snp_files_dir>/<prefix>.<worker>.snps.vcf.  Chunks are grouped per (contig, ploidy)
    and each group is one device batch."""
    curr_vcf_path = os.path.join(params['intermediate_snp_files_dir'], '%s.%d.snps.vcf' % (params['prefix'], worker_id))
    snp_files.append(curr_vcf_path)
    chunks = []
    while True:
        try:
            chunks.append(chunks_Q.get(block=False))
        except queue.Empty:
            break
        except Exception:
            if chunks_Q.empty():
                break
            raise
    groups = {}
    for c in chunks:
        groups.setdefault((c['chrom'], c['ploidy']), []).append(c)
    # Host pipeline: the genotype rules + record text of one (contig, ploidy) group are produced and written by a worker
    # thread (the formatter is native code: the GIL is released) while the GPU already works on the next group.
    from concurrent.futures import ThreadPoolExecutor

    scratch = [None]

    def emit(f, chrom, ploidy, r, grp):
        if r['n']:
            need = (400 + len(chrom)) * r['n'] + 1024
            if scratch[0] is None or scratch[0].size < need:
                scratch[0] = np.empty(need + need // 4, np.uint8)
            f.write(snp_vcf_text(chrom, r['pos'], r['ref'], r['probs'], r['dp'], r['freq'], r['fwd_dp'], r['rev_dp'],
                                 haploid=(ploidy != 'diploid'), as_array=True, out=scratch[0]))
        f.flush()
        os.fsync(f.fileno())
        for _ in grp:
            counter_Q.put(1)

    with open(curr_vcf_path, 'wb') as f, ThreadPoolExecutor(max_workers=1) as pool:
        pending, in_flight = None, []                                # groups whose kernels are queued and whose results have not been collected

        def collect():
            nonlocal pending
            chrom, ploidy, call, grp, gi = in_flight.pop(0)
            r = call.result()
            if pending is not None:
                pending.result()                                    # keeps the records in group order; re-raises errors
            pending = pool.submit(emit, f, chrom, ploidy, r, grp)
            if last_use[chrom] == gi:
                # a genome is walked contig by contig: drop the decoded alignments and the HBM pack of the one just finished (on an
                # ingest thread when there is one: unmapping ~100 MB takes 8 ms this thread would not be launching kernels)
                if piped:
                    prep_pool.submit(release_contig, chrom)
                else:
                    release_contig(chrom)
        keys = list(groups)
        last_use = {chrom: i for i, (chrom, _) in enumerate(keys)}         # last group of every contig
        for k in keys:
            groups[k].sort(key=lambda c: c['start'])
        # Ingest pipeline (what bench.py's timed region does with its uploads): while the GPU runs group i, a host thread decodes the
        # BAM records of group i + 1's contig and puts them into the reference-difference wire form (native code, the GIL is released);
        # that pack then crosses PCIe on the upload stream through a ring of three device slots, under group i's kernels.  BAM inputs
        # only; NC_SERIAL_INGEST=1 keeps the round-2 behaviour (decode + upload inside call_chunks, the GPU idle meanwhile).
        piped = isinstance(params['sam_path'], str) and os.path.exists(params['sam_path']) and not os.environ.get('NC_SERIAL_INGEST')
        uploader = None
        preps = []                                                   # the next groups' packs, being prepared
        # Device route (device_bam.py; NC_DEVICE_INGEST=0 or params['device_ingest'] = False keeps the host threads' decode): the FILE crosses
        # PCIe, is inflated and cut into records in HBM, and every group's pack is decoded there from the record stream; the worker threads only
        # decide which alignments are kept and build the tile index.  Needs the .bai.  A file larger than HBM takes passes share by share: runs of
        # contigs whose part of the file fits (device_bam.plan_shares; NC_DEVICE_INGEST_SHARE_GB of compressed BAM, default 12); a contig
        # that alone does not fit is decoded by the host threads.
        dev_codes, refs = [None], {}
        run_end, run_dev = [len(keys)] * len(keys), [None] * len(keys)       # per group: where its run of groups ends; the run's contigs (None: host route)
        # (dct['supplementary']: a split read's records are keyed by NAME, pack.name_groups -- the host route's builders do that; the CLI never sets it)
        if piped and keys and params.get('device_ingest', os.environ.get('NC_DEVICE_INGEST', '1') != '0') and params.get('fasta_path') and not params.get('supplementary'):
            from .bam import read_fasta_bytes
            from .device_bam import DeviceIngestUnavailable, open_device_bam, plan_shares
            try:
                lim = os.environ.get('NC_DEVICE_INGEST_SHARE_GB')
                order = list(dict.fromkeys(k[0] for k in keys))
                share_of = {}
                for n_sh, (cs, fits) in enumerate(plan_shares(params['sam_path'], order, None if lim is None else int(float(lim) * (1 << 30)), device=device)):
                    for c in cs:
                        share_of[c] = (n_sh, tuple(cs) if fits else None)
                a0 = 0
                for j in range(1, len(keys) + 1):
                    if j == len(keys) or share_of[keys[j][0]][0] != share_of[keys[a0][0]][0]:
                        for q in range(a0, j):
                            run_end[q], run_dev[q] = j, share_of[keys[a0][0]][1]
                        a0 = j
            except DeviceIngestUnavailable:
                pass
        dbam, prepare, nxt = None, None, 0
        depth = max(1, int(os.environ.get('NC_CALLER_DEPTH', 2 if piped else 1)))
        if piped and keys:
            ahead = int(os.environ.get('NC_INGEST_AHEAD', 2))
            # two groups ahead: the Python half of one pack's preparation (header parsing, the wire's index arrays, freeing the decoded
            # arrays: ~35 of ~58 ms per 3 Mb contig) runs beside the native decode of the next one, which releases the GIL
            prep_pool = ThreadPoolExecutor(max_workers=max(1, ahead))

        def enter_run(i):
            """group i opens a run: the device route loads the run's share of the file (the previous share is dropped), the host route needs
            nothing; then the first preparations of the run are started"""
            nonlocal dbam, prepare, uploader, nxt
            # the previous share goes first: its DeviceBam is still referenced by the old `prepare` closure, and while it lives its stream buffer
            # cannot be re-used -- a second one next to it is ~2 x 108 GB for a genome-sized file (more than the card holds)
            dbam = None
            prepare = None
            if i > 0 and run_dev[i - 1] is not None:
                from .device_bam import release as _release_device_bam
                get_engine(device).sync()
                _release_device_bam(params['sam_path'])
            if run_dev[i] is not None:
                # the first contigs' reference letters are read while the file is loaded (the loader mostly waits: for its reader threads, for the GPU)
                ref_pool = ThreadPoolExecutor(max_workers=2)
                for k in keys[i:min(run_end[i], i + 3)]:
                    refs.setdefault(k[0], ref_pool.submit(read_fasta_bytes, params['fasta_path'], k[0]))
                try:
                    dbam = open_device_bam(params['sam_path'], device, contigs=list(run_dev[i]))
                except DeviceIngestUnavailable:
                    dbam = None                                           # (includes: not enough free device / page-locked memory -- the host route needs neither)
                ref_pool.shutdown(wait=False)
            if dbam is not None:
                db = dbam
                prepare = lambda chrom, grp: _prepare_device(db, params, chrom, grp, refs.pop(chrom, None))   # noqa: E731
            else:
                prepare = lambda chrom, grp: _prepare_wire(params, chrom, grp)   # noqa: E731
                if uploader is None:
                    from .wire import WireUploader
                    uploader = WireUploader(get_engine(device))
            nxt = i
            while nxt < run_end[i] and len(preps) < max(1, ahead):
                preps.append(prep_pool.submit(prepare, keys[nxt][0], groups[keys[nxt]]))
                nxt += 1
        for i, (chrom, ploidy) in enumerate(keys):
            grp = groups[(chrom, ploidy)]
            if piped:
                if i == 0 or run_end[i - 1] == i:
                    enter_run(i)
                wp = preps.pop(0).result()
                if nxt < run_end[i]:
                    preps.append(prep_pool.submit(prepare, keys[nxt][0], groups[keys[nxt]]))
                    nxt += 1
                if dbam is not None:
                    # one codes buffer for every group: all steps run on one stream, so group i + 1's decode is ordered behind group i's last reader
                    if dev_codes[0] is None or dev_codes[0].numel() < wp['codes_len']:
                        dev_codes[0] = torch.empty(wp['codes_len'] + wp['codes_len'] // 8, dtype=torch.uint8, device=get_engine(device).device)
                    call = call_chunks(params, grp, device, dpk=dbam.pack(wp, codes=dev_codes[0]), defer=True)
                else:
                    tk = uploader.submit(wp)
                    call = call_chunks(params, grp, device, dpk=uploader.expand(tk), defer=True)
                    uploader.release(tk)
            else:
                call = call_chunks(params, grp, device, defer=True)     # enqueued behind the previous group's CNN
            in_flight.append((chrom, ploidy, call, grp, i))
            # results are collected `depth` groups late: the copies that bring a group's results back run as kernels, and those wait for the NEXT
            # group's CNN (its persistent workgroups hold every CU) -- collecting group i - 1 right after enqueuing group i made the launching
            # thread wait out that CNN and only then prepare the next launch (6.3 ms a group for 4 ms of GPU work)
            while len(in_flight) > depth:
                collect()
        while in_flight:
            collect()
        if pending is not None:
            pending.result()
        if piped and keys:
            prep_pool.shutdown(wait=True)
        if dbam is not None:
            # the share of the file this worker had in HBM goes when the worker is done (the stages that follow -- phasing, the indel
            # callers -- need the memory; the loader's work buffers up to 16 GB stay for the next load)
            from .device_bam import release as release_device_bam
            dbam = None
            release_device_bam(params['sam_path'])


# ------------------------------------------------------------------ BGZF (so no bgzip binary is needed)
_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf_write(path, data: bytes):
    """BGZF file through the library's multi-threaded compressor (see vcfio.py)"""
    from . import vcfio
    vcfio.bgzf_write(path, data)


def _sort_key(line, order):
    f = line.split('\t', 2)
    return (order.get(f[0], 1 << 30), int(f[1]))


def call_manager(params, devices=None):
    """Same contract as snpCaller.call_manager (snpCaller.py:213-287): returns the PASS VCF path
    <vcf_path>/<prefix>.snps.vcf.gz (also writes <prefix>.unfiltered.snps.vcf.gz).
    Under torch.distributed (one process per GPU, e.g. torchrun) every rank calls this function: the chunk list
    is sharded over the ranks, each rank works on ITS GPU (engine.local_device: LOCAL_RANK, or devices[local rank]) and
    writes its own worker file, rank 0 merges (other ranks return the path)."""
    from . import shard
    from .engine import local_device
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    if world > 1:                                                     # next to its GPU before anything page-locks memory or starts threads
        from .numa import bind_rank
        bind_rank(local_device(devices, rank))
    chunks_Q = queue.Queue()
    counter_Q = queue.Queue()
    snp_files = []
    wts = shard.depth_weights(params.get('sam_path'), params['chunks_list']) if world > 1 else None      # balance by the alignments held
    for chunk in shard.shard_chunks(params['chunks_list'], rank, world, wts):
        chunks_Q.put(chunk)
    params['intermediate_snp_files_dir'] = os.path.join(params['vcf_path'], 'intermediate_snp_files')
    if rank == 0:
        if os.path.exists(params['intermediate_snp_files_dir']):
            shutil.rmtree(params['intermediate_snp_files_dir'])
        os.makedirs(params['intermediate_snp_files_dir'])
    shard.barrier()
    caller(params, chunks_Q, counter_Q, snp_files, device=local_device(devices, rank), worker_id=rank + 1)
    shard.barrier()
    all_path_ = os.path.join(params['vcf_path'], '%s.snps.vcf.gz' % params['prefix'])
    if rank != 0:
        shard.barrier()
        return all_path_
    snp_files = [os.path.join(params['intermediate_snp_files_dir'], '%s.%d.snps.vcf' % (params['prefix'], r + 1))
                 for r in range(world)]
    all_path = os.path.join(params['vcf_path'], '%s.unfiltered.snps.vcf.gz' % params['prefix'])
    pass_path = os.path.join(params['vcf_path'], '%s.snps.vcf.gz' % params['prefix'])
    if not params.get('suppress_progress'):
        print('\n%s: Combining SNP calls.' % str(datetime.datetime.now()))
    contigs = []
    for x in params['regions_list']:
        if x[0] not in contigs:
            contigs.append(x[0])
    header = VCF_HEADER.format(contigs=''.join('##contig=<ID=%s>\n' % c for c in contigs), sample=params['sample'])
    lines = []
    for fn in snp_files:
        with open(fn) as fd:
            lines.extend(fd.readlines())
    order = {c: i for i, c in enumerate(contigs)}
    keyed = []
    for ln in lines:                                                # CHROM, POS, REF, FILTER of every record
        f = ln.split('\t', 7)
        keyed.append((order.get(f[0], 1 << 30), int(f[1]), len(f[3]), f[6] == 'PASS', ln))
    keyed.sort(key=lambda k: (k[0], k[1]))                          # bcftools sort (:284); stable, keeps E3 duplicates
    from . import vcfio
    for path, recs in ((all_path, keyed), (pass_path, [k for k in keyed if k[3]])):     # bcftools view -f PASS (:285)
        body = ''.join(k[4] for k in recs).encode()
        vcfio.write_vcf_gz_with_csi(path, header, body, contigs, np.array([k[0] for k in recs], np.int64),
                                    np.array([k[1] for k in recs], np.int64), np.array([k[2] for k in recs], np.int64),
                                    np.array([len(k[4].encode()) for k in recs], np.int64))   # + <path>.csi (tabix -p vcf --csi)
    shard.barrier()
    return pass_path
