"""Host-side mirror of the GPU-resident parts of the reference's indel featuriser
(nanocaller_src/generate_indel_pileups.py).

* `scan_indel_candidates(dct, chunk)` = pass 1 of get_indel_testing_candidates (:197-276): the `variants`
  dict {anchor position: 0 (long window) | 1 (small window)}; the per-column work runs in nc_indel_scan, the
  order-dependent `v <= prev` suppression (:249,267,273) is applied here.
* `msa_tensor(rows, ref_row)` = the histogram half of msa() (:57-71) through nc_indel_tensor.
* pass 2 (:306-361): read windows at the anchor columns through the native BAM reader (nc_indel_slices, row a11),
  `msa()` with a pluggable aligner (MUSCLE itself is an external binary, absent here; its aligned rows are the input of
  msa_tensor), `allele_prediction` (row a13) through nc_allele_prediction.  pysam / MUSCLE / parasail parity is unpinned
  (SURVEY.md 8c): the restatements are pinned against independent implementations in tests/.
The impute_indel_phase branch (:278-304) is not covered.
"""
from __future__ import annotations

import ctypes as C
import random
import subprocess

import numpy as np
import torch

from . import _lib

from .engine import get_engine
from .generate_SNP_pileups import _exclude_rows, _resolve
from .pack import pack_world

_PACKS = {}


def pick_variants(col_type, start, win_size):
    """Apply `if v_pos <= prev: continue` (:249) to the per-column decisions; -> {anchor: type} (dict semantics:
    a later detection overwrites an equal anchor)."""
    variants = {}
    prev = 0
    lo = max(1, int(start))
    for c in np.nonzero(col_type >= 0)[0]:
        v = lo + int(c)
        if v <= prev:
            continue
        if col_type[c] == 0:
            prev = v + win_size
            variants[max(1, v - win_size)] = 0                      # :267-268
        else:
            prev = v + 10
            variants[max(1, v - 10)] = 1                            # :273-274
    return variants


def scan_indel_candidates(dct, chunk, device=0, haploid=False):
    """Pass 1 for one chunk (dict, like the reference's per-chunk call) -> {anchor: type}; or for a list of chunks of ONE
    contig and BAM -> list of such dicts, every chunk with the reference's per-chunk semantics, all of them in the same
    kernel launches (nc_indel_scan_batch)."""
    if dct.get("impute_indel_phase"):
        raise NotImplementedError("impute_indel_phase (generate_indel_pileups.py:278-304) is not part of this build")
    chunks = None
    if not isinstance(chunk, dict):
        chunks = list(chunk)
        if not chunks:
            return []
        if any((c["sam_path"], c["chrom"]) != (chunks[0]["sam_path"], chunks[0]["chrom"]) for c in chunks):
            raise ValueError("scan_indel_candidates: a chunk list must be of one BAM and one contig")
    first = chunk if isinstance(chunk, dict) else chunks[0]
    world = _resolve(first["sam_path"], first["chrom"], dct.get("fasta_path"))
    excl_rows = _exclude_rows(dct, first["chrom"])
    key = (id(world), bool(dct.get("supplementary")), device)
    eng = get_engine(device)
    eng.use_torch_stream()
    if key not in _PACKS:
        _PACKS[key] = (eng.upload(pack_world(world, supplementary=bool(dct.get("supplementary")))), world)
    dp = _PACKS[key][0]
    excl = None
    if excl_rows:
        m = np.zeros(dp.n_tiles * dp.tile_size, np.uint8)
        for (a, b) in excl_rows:                                    # IntervalTree.overlaps(pos): a <= pos < b
            m[max(0, a - dp.tile_pos0):max(0, b - dp.tile_pos0)] = 1
        excl = torch.from_numpy(m).to(eng.device)
    if isinstance(chunk, dict):
        col_type = eng.indel_scan(dp, chunk["start"], chunk["end"], mincov=dct["mincov"], win_size=dct["win_size"],
                                  small_win_size=dct["small_win_size"], ins_t=dct["ins_t"], del_t=dct["del_t"], excl=excl, haploid=haploid)
        return pick_variants(col_type, chunk["start"], dct["win_size"])
    cols = eng.indel_scan_batch(dp, [(c["start"], c["end"]) for c in chunks], mincov=dct["mincov"], win_size=dct["win_size"],
                                small_win_size=dct["small_win_size"], ins_t=dct["ins_t"], del_t=dct["del_t"], excl=excl, haploid=haploid)
    return [pick_variants(ct, c["start"], dct["win_size"]) for ct, c in zip(cols, chunks)]


def msa_tensor(rows_list, ref_rows_list, device=0):
    """-> (float64 [S,5,128,2] like msa()'s final_mat (:69-71), list of consensus strings with gaps removed (:64))."""
    eng = get_engine(device)
    eng.use_torch_stream()
    x, cns = eng.indel_tensor(rows_list, ref_rows_list)
    sym = "AGTC"
    return x.cpu().numpy().astype(np.float64), ["".join(sym[c] for c in row) for row in cns]


# ------------------------------------------------------------------------------------------------- pass 2 (a11 - a13)
def nw_cigar(s1, s2, open_=9, extend=1, match=20, mismatch=-10):
    """[(op, count)] of the global alignment, parasail op codes ('=' 7, 'X' 8, 'I' 1, 'D' 2): the library's restatement
    of parasail.nw_trace(s1, s2, open, extend, matrix).cigar (generate_indel_pileups.py:10,79)."""
    L = _lib.lib()
    a, b = s1.encode(), s2.encode()
    cap = len(a) + len(b) + 1
    ops = np.empty(cap, np.int32)
    cnt = np.empty(cap, np.int32)
    n = C.c_int32()
    rc = L.nc_nw_cigar(a, len(a), b, len(b), open_, extend, match, mismatch, _lib.npp(ops), _lib.npp(cnt), cap, C.byref(n))
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("nc_nw_cigar failed (%d)" % rc)
    return [(int(ops[k]), int(cnt[k])) for k in range(n.value)]


def allele_prediction(alt, ref_seq, max_range):
    """(REF, ALT) strings or (None, None): generate_indel_pileups.py:77-127 through nc_allele_prediction."""
    L = _lib.lib()
    a, b = alt.encode(), ref_seq.encode()
    rl, al = C.c_int32(), C.c_int32()
    rc = L.nc_allele_prediction(a, len(a), b, len(b), int(max_range), C.byref(rl), C.byref(al))
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("nc_allele_prediction failed (%d)" % rc)
    if rl.value < 0:
        return (None, None)
    return ref_seq[:rl.value], alt[:al.value]


def muscle_aligner(names, seqs, ref):
    """The reference's aligner call (:24-44): MUSCLE 3.8 as a subprocess on a FASTA of the reads (+ '_SEQ' suffix) and
    the reference row.  -> (aligned read rows in MUSCLE's output order, aligned reference row)."""
    fa = "".join(">%s_SEQ\n%s\n" % (n, s) for n, s in zip(names, seqs)) + ">ref_SEQ\n%s" % ref
    try:
        proc = subprocess.Popen(["muscle", "-quiet", "-gapopen", "1.0", "-maxiters", "1", "-diags1"], stdout=subprocess.PIPE,
                                stdin=subprocess.PIPE, stderr=subprocess.PIPE)
    except FileNotFoundError as e:
        raise RuntimeError("muscle (3.8) is not on PATH: pass aligner=... to msa() / get_indel_testing_candidates()") from e
    out = proc.communicate(input=fa.encode("utf-8"))
    rows, ref_row = [], None
    for rec in out[0].decode("utf-8")[1:].replace("\n", "").split(">"):
        p1, p2 = rec.split("_SEQ")
        if p1 != "ref":
            rows.append(p2)
        else:
            ref_row = p2
    return rows, ref_row


_SYM = {"A": 0, "G": 1, "T": 2, "C": 3, "-": 4}


def msa(seq_list, ref, v_pos, mincov, maxcov, aligner=None, device=0):
    """generate_indel_pileups.py:12-73 -> (flag, indel_flag, final_mat float64 (5,128,2), cns, ref_seq).  Down-sampling
    (unseeded random.sample, as in the reference), name sort, alignment by `aligner(names, seqs, ref)` (default: MUSCLE),
    the histogram / consensus / tensor half on the GPU (nc_indel_tensor)."""
    sample = list(seq_list.keys())
    if len(sample) > maxcov:
        sample = random.sample(sample, min(len(sample), maxcov))
    sample = sorted(sample)
    rows, ref_row = (aligner or muscle_aligner)(sample, [seq_list[n] for n in sample], ref)
    if len(rows) < mincov or ref_row is None:
        return (0, 0, None, None, None)
    mat = np.array([[_SYM[c] for c in r] for r in rows], np.uint8)                # KeyError on 'N', as in the reference (:56)
    ref_codes = np.array([_SYM[c] for c in ref_row], np.uint8)
    x, cns = msa_tensor([mat], [ref_codes], device)
    return (1, 1, x[0], cns[0], ref_row.replace("-", ""))


def get_indel_testing_candidates(dct, chunk, aligner=None, device=0):
    """generate_indel_pileups.py:128-371 for a BAM `chunk['sam_path']` with HP/PS tags:
    -> (pos, x0, x1, x2, alleles, phase); pass 1 on the GPU (nc_indel_scan), pass 2 through the native reader."""
    from .bam import BamFile, read_fasta
    chrom, start, end = chunk["chrom"], chunk["start"], chunk["end"]
    window_before, window_after = 0, 160
    if dct["seq"] == "pacbio":
        window_after = 260
    variants = scan_indel_candidates(dct, chunk, device)
    empty = ([], [], [], [], [], [])
    if not variants:
        return empty
    fasta = read_fasta(dct["fasta_path"], chrom)
    chrom_length = len(fasta)
    lo, hi = max(1, start - 200), end + 400                                        # ref_dict range (:174)
    flag = 0x4 | 0x100 | 0x200 | 0x400 | (0 if dct.get("supplementary") else 0x800)
    bf = BamFile(chunk["sam_path"])
    # hap sets / phase_dict come from a fetch over [start-100000, end+1000] (:178-188); the pileup of pass 2 from
    # [start-10-win, end] (:306): decode the union once
    anchors = sorted(v for v in variants if max(0, start - 10 - dct["win_size"]) < v <= end)
    d = bf.decode(chrom, max(1, start - 100000), end + 1000, anchors=anchors, window_before=window_before,
                  window_after=window_after, keep_mask=flag)
    bf.close()
    names, hap, ps = d["names"], d["hap"], d["ps"]
    max_range = {0: max(10, dct["win_size"]), 1: 10}
    out_pos, x0, x1, x2, alleles, phase = [], [], [], [], [], []
    for v_pos, win in zip(anchors, d["windows"]):
        ref = "".join((fasta[p - 1] if (lo <= p <= hi and fasta[p - 1] in "AGTC") else "N")
                      for p in range(v_pos - window_before, min(chrom_length, v_pos + window_after + 1)))
        if "N" in ref:
            continue
        d_tot, d0, d1 = {}, {}, {}
        for r, text in win:
            d_tot[names[r]] = text
            if hap[r] == 1:
                d0[names[r]] = text
            elif hap[r] == 2:
                d1[names[r]] = text
        f0, _, m0, alt0, ref0 = msa(d0, ref, v_pos, 2, dct["maxcov"], aligner, device)
        f1, _, m1, alt1, ref1 = msa(d1, ref, v_pos, 2, dct["maxcov"], aligner, device)
        ft, _, mt, altt, reft = msa(d_tot, ref, v_pos, dct["mincov"], dct["maxcov"], aligner, device)
        if f0 and f1 and ft:
            out_pos.append(v_pos)
            x0.append(m0); x1.append(m1); x2.append(mt)
            first = next(iter(d0.keys()))
            k = names.index(first)
            phase.append(int(ps[k]) if hap[k] else None)
            mr = max_range[variants[v_pos]]
            alleles.append([allele_prediction(alt0, ref0, mr), allele_prediction(alt1, ref1, mr), allele_prediction(altt, reft, mr)])
    if not out_pos:
        return empty
    return (out_pos, np.array(x0), np.array(x1), np.array(x2), alleles, phase)


def get_indel_testing_candidates_haploid(dct, chunk, aligner=None, device=0):
    """generate_indel_pileups_haploid.py:128-277 -> (pos, x, alleles): one read set per anchor, no HP split."""
    from .bam import BamFile, read_fasta
    chrom, start, end = chunk["chrom"], chunk["start"], chunk["end"]
    window_before, window_after = 0, 160
    if dct["seq"] == "pacbio":
        window_after = 260
    variants = scan_indel_candidates(dct, chunk, device, haploid=True)
    empty = ([], [], [])
    if not variants:
        return empty
    fasta = read_fasta(dct["fasta_path"], chrom)
    chrom_length = len(fasta)
    lo, hi = max(1, start - 200), end + 400
    flag = 0x4 | 0x100 | 0x200 | 0x400 | (0 if dct.get("supplementary") else 0x800)
    anchors = sorted(v for v in variants if max(0, start - 10 - dct["win_size"]) < v <= end)
    bf = BamFile(chunk["sam_path"])
    d = bf.decode(chrom, max(1, start - 10 - dct["win_size"] - window_after), end + 1000, anchors=anchors, window_before=window_before,
                  window_after=window_after, keep_mask=flag)
    bf.close()
    names = d["names"]
    max_range = {0: max(10, dct["win_size"]), 1: 10}
    out_pos, xs, alleles = [], [], []
    for v_pos, win in zip(anchors, d["windows"]):
        ref = "".join((fasta[p - 1] if (lo <= p <= hi and fasta[p - 1] in "AGTC") else "N")
                      for p in range(v_pos - window_before, min(chrom_length, v_pos + window_after + 1)))
        if "N" in ref:
            continue
        d_tot = {names[r]: text for r, text in win}
        ft, _, mt, altt, reft = msa(d_tot, ref, v_pos, dct["mincov"], dct["maxcov"], aligner, device)
        if ft:
            out_pos.append(v_pos)
            xs.append(mt)
            alleles.append(allele_prediction(altt, reft, max_range[variants[v_pos]]))
    if not out_pos:
        return empty
    return (out_pos, np.array(xs), alleles)
