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
* dct['impute_indel_phase'] (:278-304): the column-level predicate (:278-284) is part of the GPU scan (col_type 2); the
  read grouping of the few flagged columns (:285-304) runs here on their pileup strings (`impute_groups`), and pass 2
  takes its two read sets from `extra_variants` (:310-312).
"""
from __future__ import annotations

import bisect
import contextlib
import ctypes as C
import gc
import os
import subprocess

import numpy as np
import torch

from . import _lib

from .engine import get_engine
from .generate_SNP_pileups import _exclude_rows, _resolve, device_pack


def pick_variants(col_type, start, win_size, groups=None, extra=None):
    """Apply `if v_pos <= prev: continue` (:249) to the per-column decisions; -> {anchor: type} (dict semantics:
    a later detection overwrites an equal anchor).  col_type 2 (impute_indel_phase candidates): `groups(v)` returns the
    two read-name collections of :285-300 or None; accepted ones go to `extra` {anchor: (names0, names1)} (:301-304)."""
    variants = {}
    prev = 0
    lo = max(1, int(start))
    idx = np.nonzero(col_type >= 0)[0]
    # plain Python ints; a flagged region is hundreds of consecutive columns, almost all skipped by `v <= prev`: the skip is a
    # bisection over the (ascending) column list instead of a walk
    cols, types = idx.tolist(), col_type[idx].tolist()
    i, n = 0, len(cols)
    while i < n:
        v = lo + cols[i]
        if v <= prev:
            i = bisect.bisect_right(cols, prev - lo, i + 1)
            continue
        t = types[i]
        i += 1
        if t == 0:
            prev = v + win_size
            variants[max(1, v - win_size)] = 0                      # :267-268
        elif t == 1:
            prev = v + 10
            variants[max(1, v - 10)] = 1                            # :273-274
        else:
            sets = groups(v) if groups else None
            if sets is not None:
                prev = v + 10                                       # :301-303
                variants[max(1, v - 10)] = 1
                if extra is not None:
                    extra[max(1, v - 10)] = sets
    return variants


def impute_groups(names, strings, mincov):
    """:285-300 for one column: `names` / `strings` = get_query_names() and the upper-cased get_query_sequences(
    add_indels=True) in pileup order.  Reads are grouped by identical string; the largest group against the runner-up
    (or everything else), or, when one group holds more than 80 % of the reads, its two halves.  -> (names0, names1) or
    None when either side has fewer than mincov reads."""
    tot = len(names)
    groups = {}
    for s, n in zip(strings, names):
        groups.setdefault(s, []).append(n)
    order = sorted(groups, key=lambda g: len(groups[g]), reverse=True)          # stable: ties keep first-seen order
    top = groups[order[0]]
    if len(top) <= 0.8 * tot:
        r0 = set(top)
        r1 = set(groups[order[1]]) if len(groups[order[1]]) >= mincov else set(names) - r0
    else:
        r0, r1 = top[:len(top) // 2], top[len(top) // 2:]
    if len(r0) >= mincov and len(r1) >= mincov:
        return (r0, r1)
    return None


def column_strings(world, sam_path, chrom, cols, supplementary=False):
    """{column: (names, upper-cased pileup strings with indels)} of the kept reads, in file order, for the few columns the
    impute_indel_phase rule looks at.  In-memory worlds carry the inserted bases in meta['ev_ins']; for a BAM file the
    columns are read through the native reader (base + inserted bases = the head of the pass-2 window at that column)."""
    cols = sorted(set(int(c) for c in cols))
    out = {}
    if not cols:
        return out
    flag = 0x4 | 0x100 | 0x200 | 0x400 | (0 if supplementary else 0x800)
    ev_off, ev_pos, ev_len = world.meta["events"]
    if "ev_ins" in world.meta:
        ins_off, ins_bases = world.meta["ev_ins"]
        keep = (world.read_flag & flag) == 0
        rs, re_ = world.read_start, world.read_end
        letters = world.meta.get("letters", {})
        for v in cols:
            names, strs = [], []
            for r in np.nonzero(keep & (rs <= v) & (re_ > v))[0]:
                code = int(world.codes[world.read_off[r] + (v - rs[r])])
                s = letters.get((int(r), v - 1), "AGTC*"[code]).upper()
                e0, e1 = ev_off[r], ev_off[r + 1]
                k = e0 + int(np.searchsorted(ev_pos[e0:e1], v))
                if k < e1 and ev_pos[k] == v:
                    ln = int(ev_len[k])
                    s += ("+%d%s" % (ln, bytes(ins_bases[ins_off[k]:ins_off[k + 1]]).decode().upper())) if ln > 0 else ("-%d%s" % (-ln, "N" * -ln))
                names.append(world.names[r])
                strs.append(s)
            out[v] = (names, strs)
        return out
    if not isinstance(sam_path, str):
        raise ValueError("impute_indel_phase needs the inserted bases: a BAM path, or a world with meta['ev_ins']")
    from .bam import BamFile
    sel = np.isin(ev_pos, np.asarray(cols, ev_pos.dtype)) & (ev_len > 0)
    w_after = 1 + (int(ev_len[sel].max()) if sel.any() else 0)
    bf = BamFile(sam_path)
    d = bf.decode(chrom, cols[0], cols[-1], anchors=cols, window_before=0, window_after=w_after, keep_mask=flag)
    bf.close()
    for v, win in zip(cols, d["windows"]):
        names, strs = [], []
        for r, text in win:
            e0, e1 = d["ev_off"][r], d["ev_off"][r + 1]
            k = e0 + int(np.searchsorted(d["ev_pos"][e0:e1], v))
            deleted = k > e0 and d["ev_len"][k - 1] < 0 and d["ev_pos"][k - 1] - d["ev_len"][k - 1] >= v
            s = "*" if deleted else (text[:1] or "N")
            if k < e1 and d["ev_pos"][k] == v:
                ln = int(d["ev_len"][k])
                s += ("+%d%s" % (ln, text[1:1 + ln])) if ln > 0 else ("-%d%s" % (-ln, "N" * -ln))
            names.append(d["names"][r])
            strs.append(s)
        out[v] = (names, strs)
    return out


def scan_indel_candidates(dct, chunk, device=0, haploid=False, extra_variants=None):
    """Pass 1 for one chunk (dict, like the reference's per-chunk call) -> {anchor: type}; or for a list of chunks of ONE
    contig and BAM -> list of such dicts, every chunk with the reference's per-chunk semantics, all of them in the same
    kernel launches (nc_indel_scan_batch).  With dct['impute_indel_phase'] (diploid only) `extra_variants` (a dict, or a
    list of dicts for a chunk list) receives {anchor: (names0, names1)} of the imputed columns (:304)."""
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
    supp = bool(dct.get("supplementary"))
    eng = get_engine(device)
    eng.use_torch_stream()
    # the SNP path's pack of this contig (same key when no exclusion list is folded into its reference codes)
    dp = device_pack(first["sam_path"], dct.get("fasta_path"), first["chrom"], supp, None, device)[0]
    excl = None
    if excl_rows:
        m = np.zeros(dp.n_tiles * dp.tile_size, np.uint8)
        for (a, b) in excl_rows:                                    # IntervalTree.overlaps(pos): a <= pos < b
            m[max(0, a - dp.tile_pos0):max(0, b - dp.tile_pos0)] = 1
        excl = torch.from_numpy(m).to(eng.device)
    impute = bool(dct.get("impute_indel_phase")) and not haploid
    kw = dict(mincov=dct["mincov"], win_size=dct["win_size"], small_win_size=dct["small_win_size"], ins_t=dct["ins_t"],
              del_t=dct["del_t"], excl=excl, haploid=haploid, impute=impute)
    one = chunks is None
    todo = [chunk] if one else chunks
    cols = [eng.indel_scan(dp, chunk["start"], chunk["end"], **kw)] if one else eng.indel_scan_batch(dp, [(c["start"], c["end"]) for c in chunks], **kw)
    groups = None
    if impute:
        want = np.concatenate([max(1, int(c["start"])) + np.nonzero(ct == 2)[0] for ct, c in zip(cols, todo)])
        strings = column_strings(world, first["sam_path"], first["chrom"], want, supp)
        groups = lambda v: impute_groups(*strings[v], dct["mincov"])
    extras = [extra_variants] if one else (extra_variants if extra_variants is not None else [None] * len(todo))
    if not one and extra_variants is not None and len(extras) != len(todo):
        raise ValueError("scan_indel_candidates: extra_variants must be a list of one dict per chunk")
    out = [pick_variants(ct, c["start"], dct["win_size"], groups, ex) for ct, c, ex in zip(cols, todo, extras)]
    return out[0] if one else out


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


def allele_prediction_batch(alts, ref_seqs, max_ranges, eng=None):
    """[allele_prediction(alt, ref_seq, max_range) for ...] in one native call: on the device when an engine is given (the
    16-lane register aligner with parasail's scoring + one lane per alignment for the allele extraction: same results), else,
    and for windows the device kernel does not cover, on the usable host cores"""
    n = len(alts)
    if n == 0:
        return []
    L = _lib.lib()
    aoff, roff = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.int32)
    np.cumsum([len(a) for a in alts], out=aoff[1:])
    np.cumsum([len(r) for r in ref_seqs], out=roff[1:])
    mr = np.ascontiguousarray(max_ranges, np.int32)
    rl, al = np.empty(n, np.int32), np.empty(n, np.int32)
    a_b, r_b = "".join(alts).encode(), "".join(ref_seqs).encode()
    rc = _lib.NC_ERR_CAPACITY
    if eng is not None:
        rc = L.nc_allele_prediction_device(eng.ctx, n, a_b, _lib.npp(aoff), r_b, _lib.npp(roff), _lib.npp(mr), _lib.npp(rl), _lib.npp(al))
        if rc not in (_lib.NC_OK, _lib.NC_ERR_CAPACITY, _lib.NC_ERR_NOMEM):
            raise _lib.NanoCallerHipError("nc_allele_prediction_device failed (%d): %s" % (rc, eng.last_error() if hasattr(eng, "last_error") else ""))
    # (a batch whose traceback does not fit the device call -- one very long consensus sizes every row of it -- goes to the host aligner
    # like one beyond the kernel's shapes: same results)
    if rc in (_lib.NC_ERR_CAPACITY, _lib.NC_ERR_NOMEM):
        rc = L.nc_allele_prediction_batch(n, a_b, _lib.npp(aoff), r_b, _lib.npp(roff), _lib.npp(mr), _lib.npp(rl), _lib.npp(al))
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("nc_allele_prediction_batch failed (%d)" % rc)
    # plain ints: indexing numpy scalars and slicing with them costs ~2 us per pair, a third of the whole batch at 10^4 anchors
    return [(None, None) if r < 0 else (rs[:r], a_[:a]) for rs, a_, r, a in zip(ref_seqs, alts, rl.tolist(), al.tolist())]


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


def star_aligner(names, seqs, ref, open_=None, extend=None, match=None, mismatch=None):
    """Same interface as muscle_aligner, without the external binary: every read is aligned to the
    reference window (Gotoh, anchored at the window start, free tail; the scoring of the reference's own parasail call) and
    the pairwise alignments are merged in reference coordinates by nc_star_msa -- the longest insertion per reference slot
    makes the columns, shorter ones are left-justified.  Not MUSCLE's algorithm: its rows are not comparable with MUSCLE's
    output, only the calls made from them are.  -> (aligned read rows in input order, aligned reference row)."""
    L = _lib.lib()
    open_, extend, match, mismatch = [d if v is None else v for v, d in zip((open_, extend, match, mismatch), _lib.STAR_SCORING)]
    n = len(seqs)
    raw = "".join(seqs).encode()
    off = np.zeros(n + 1, np.int32)
    np.cumsum([len(q) for q in seqs], out=off[1:])
    rb = ref.encode()
    cap = len(rb) + sum(len(q) for q in seqs) + 1                     # every read base could be an insertion
    cap = min(cap, 4 * len(rb) + 64)
    ncol = C.c_int32()
    while True:
        rows = np.empty((max(n, 1), cap), np.uint8)
        ref_row = np.empty(cap, np.uint8)
        rc = L.nc_star_msa(n, raw, _lib.npp(off), rb, len(rb), int(open_), int(extend), int(match), int(mismatch), cap,
                           _lib.npp(rows), _lib.npp(ref_row), C.byref(ncol))
        if rc == _lib.NC_ERR_CAPACITY:
            cap = ncol.value
            continue
        if rc != _lib.NC_OK:
            raise _lib.NanoCallerHipError("nc_star_msa failed (%d)" % rc)
        break
    sym = np.frombuffer(b"AGTC-N", np.uint8)
    nc = ncol.value
    return [sym[rows[r, :nc]].tobytes().decode() for r in range(n)], sym[ref_row[:nc]].tobytes().decode()


def default_aligner():
    """MUSCLE when it is on PATH (the reference's aligner), else the built-in star aligner"""
    import shutil
    return muscle_aligner if shutil.which("muscle") else star_aligner


_SYM = {"A": 0, "G": 1, "T": 2, "C": 3, "-": 4}


def msa(seq_list, ref, v_pos, mincov, maxcov, aligner=None, device=0):
    """generate_indel_pileups.py:12-73 -> (flag, indel_flag, final_mat float64 (5,128,2), cns, ref_seq).  Down-sampling
    (deterministic: the first `maxcov` reads in pileup order -- the reference draws an unseeded random.sample, :19-20, so its
    result above maxcov is not reproducible; same policy as the SNP path), name sort, alignment by `aligner(names, seqs, ref)`,
    the histogram / consensus / tensor half on the GPU (nc_indel_tensor)."""
    sample = sorted(list(seq_list.keys())[:maxcov])
    if aligner == "device":
        aligner = star_aligner                                      # one set at a time: the host statement (identical rows)
    rows, ref_row = (aligner or default_aligner())(sample, [seq_list[n] for n in sample], ref)
    if len(rows) < mincov or ref_row is None:
        return (0, 0, None, None, None)
    # a read base other than AGTC (N): the reference's symbol table raises KeyError and the chunk is lost (:56); here it
    # counts as a gap at its column (the device rows do the same, nc_msa.hip sym_code)
    mat = np.array([[_SYM.get(c, 4) for c in r] for r in rows], np.uint8)
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
    extra_variants = {}
    variants = scan_indel_candidates(dct, chunk, device, extra_variants=extra_variants)
    empty = ([], [], [], [], [], [])
    if not variants:
        return empty
    lo, hi = max(1, start - 200), end + 400                                        # ref_dict range (:174)
    flag = 0x4 | 0x100 | 0x200 | 0x400 | (0 if dct.get("supplementary") else 0x800)
    # the pileup of pass 2 covers [start-10-win, end] (:306)
    anchors = sorted(v for v in variants if max(0, start - 10 - dct["win_size"]) < v <= end)
    max_range = {0: max(10, dct["win_size"]), 1: 10}
    if aligner is None and default_aligner() is star_aligner:
        aligner = "device"                                                         # no MUSCLE here: star alignment on the GPU
    if aligner == "device":
        # the whole of pass 2 natively: the contig is decoded once (with its query bases) for all its chunks
        ctg = decoded_contig(chunk["sam_path"], chrom, dct["fasta_path"])
        return _pass2_native(dct, variants, extra_variants, anchors, ctg, lo, hi, window_after, max_range, device, haploid=False)
    fasta = read_fasta(dct["fasta_path"], chrom)
    chrom_length = len(fasta)
    bf = BamFile(chunk["sam_path"])
    # hap sets / phase_dict come from a fetch over [start-100000, end+1000] (:178-188): decode the union once
    d = bf.decode(chrom, max(1, start - 100000), end + 1000, anchors=anchors, window_before=window_before,
                  window_after=window_after, keep_mask=flag)
    bf.close()
    names, hap, ps = d["names"], d["hap"], d["ps"]
    out_pos, x0, x1, x2, alleles, phase = [], [], [], [], [], []
    for v_pos, win in zip(anchors, d["windows"]):
        ref = "".join((fasta[p - 1] if (lo <= p <= hi and fasta[p - 1] in "AGTC") else "N")
                      for p in range(v_pos - window_before, min(chrom_length, v_pos + window_after + 1)))
        if "N" in ref:
            continue
        d_tot, d0, d1 = {}, {}, {}
        imputed = extra_variants.get(v_pos)                                      # :310-312
        for r, text in win:
            d_tot[names[r]] = text
            if (names[r] in imputed[0]) if imputed else hap[r] == 1:
                d0[names[r]] = text
            elif (names[r] in imputed[1]) if imputed else hap[r] == 2:
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


_DROP_AGTC = str.maketrans("", "", "AGTC")


def _sample_set(seq_list, mincov, maxcov):
    """the part of msa() before the aligner (:13-23): down-sample to maxcov (the first maxcov reads in pileup order: a
    deterministic stand-in for the reference's unseeded random.sample), sort the names;
    -> (names, seqs) or None when fewer than mincov reads remain"""
    sample = sorted(list(seq_list.keys())[:maxcov])
    if len(sample) < mincov:
        return None
    return sample, [seq_list[n] for n in sample]


_CONTIGS = {}


def decoded_contig(sam_path, chrom, fasta_path):
    """The contig's alignments WITH their query bases, decoded once (nc_bam_decode_regions) and kept for every chunk of the
    contig, + the contig's reference bases.  One contig at a time stays cached."""
    key = (sam_path, chrom, fasta_path)
    if key not in _CONTIGS:
        from .bam import decode_parallel, name_gids, read_fasta, unsupported_counts
        _CONTIGS.clear()
        dec = decode_parallel(sam_path, chrom, keep_seq=True)
        fasta = read_fasta(fasta_path, chrom)
        _CONTIGS[key] = dict(dec=dec, handle=dec["_owner"].handle, fasta=fasta, fasta_b=fasta.encode("ascii"), keep={}, name_idx=None)
        # the same decode serves pass 1 (and the SNP path): register it as the contig's World so that nothing decodes the BAM twice
        from . import generate_SNP_pileups as gsp
        from .synth import World

        def as_world():
            w = World(chrom=chrom, ref=fasta, read_start=dec["read_start"], read_end=dec["read_end"], read_flag=dec["read_flag"],
                      read_off=dec["read_off"], codes=dec["codes"], names=dec["names"])
            w.meta.update(events=(dec["ev_off"], dec["ev_pos"], dec["ev_len"]), hap=dec["hap"], ps=dec["ps"], seq_off=dec["seq_off"], seq=dec["seq"])
            w.meta["unsupported"] = unsupported_counts(dec)          # (bam.read_bam's check: this World serves the SNP path and pass 1 too)
            w.meta["name_gid"] = name_gids(dec)
            return w
        gsp._BAM_WORLDS.get((sam_path, fasta_path, chrom), as_world)
    return _CONTIGS[key]


def _pass2_native(dct, variants, extra_variants, anchors, ctg, lo, hi, window_after, max_range, device, haploid, by_index=False, device_x=False):
    """Pass 2 (:306-361) for all anchors of a chunk: read sets assembled natively (nc_indel_pass2_sets) from the decoded contig,
    every set aligned in ONE device call (star alignment + rows -> tensor), allele strings by nc_allele_prediction_batch.
    -> diploid (pos, x0, x1, x2, alleles, phase) / haploid (pos, x, alleles).  by_index: `variants` / `extra_variants` are keyed
    by the anchor's INDEX in `anchors` (merged chunks) and the kept anchor indices are appended to the result."""
    L = _lib.lib()
    dec = ctg["dec"]
    flag = 0x4 | 0x100 | 0x200 | 0x400 | (0 if dct.get("supplementary") else 0x800)
    if flag not in ctg["keep"]:
        from .pack import pileup_depth_cap
        ctg["keep"][flag] = pileup_depth_cap(dec["read_start"], dec["read_end"], np.ascontiguousarray((dec["read_flag"] & flag) == 0, np.uint8))
    keep = ctg["keep"][flag]
    anc = np.ascontiguousarray(anchors, np.int32)
    imp_idx = imp_off = imp_reads = None
    if extra_variants:
        if ctg["name_idx"] is None:
            ctg["name_idx"] = {}
            for i, nm in enumerate(dec["names"]):                     # every alignment of a name (the reference's collections are by name:
                ctg["name_idx"].setdefault(nm, []).append(i)            # a supplementary alignment at the anchor belongs to its read's side)
        nidx = ctg["name_idx"]
        imp_idx = np.full(len(anchors), -1, np.int32)
        offs, reads = [0], []
        for k, v in enumerate(anchors):
            key = k if by_index else v
            if key in extra_variants:
                imp_idx[k] = (len(offs) - 1) // 2
                for side in extra_variants[key]:
                    for n in side:
                        reads.extend(nidx[n])
                    offs.append(len(reads))
        imp_off = np.ascontiguousarray(offs, np.int32)
        imp_reads = np.ascontiguousarray(reads if reads else [0], np.int32)
    h = C.c_void_p()
    rc = L.nc_indel_pass2_sets(ctg["handle"], _lib.npp(keep), len(anc), _lib.npp(anc), ctg["fasta_b"], len(ctg["fasta"]), int(lo), int(hi),
                               int(window_after), int(dct["mincov"]), int(dct["maxcov"]), 1 if haploid else 0, _lib.npp(imp_idx),
                               _lib.npp(imp_off), _lib.npp(imp_reads), C.byref(h))
    if rc != _lib.NC_OK:
        err = _lib.NanoCallerHipError("nc_indel_pass2_sets failed (%d)" % rc)
        err.status = rc
        raise err
    try:
        v = _lib.Pass2ArraysC()
        L.nc_pass2_view(h, C.byref(v))
        nk, S, ns = v.n_kept, v.sets_per_anchor, v.n_sets
        if nk == 0:
            res = ([], [], []) if haploid else ([], [], [], [], [], [])
            return res + ([],) if by_index else res
        eng = get_engine(device)
        eng.use_torch_stream()
        x, cns_str, _ = eng.star_msa_tensor_flat(ns, C.c_void_p(v.reads), C.c_void_p(v.read_off), C.c_void_p(v.set_read0), C.c_void_p(v.refs),
                                                 C.c_void_p(v.ref_off), min(v.max_cols, 4 * window_after + 64), cns_as_str=True,
                                                 al_dup=None if haploid or os.environ.get("NC_MSA_NO_DEDUP") else C.c_void_p(v.al_dup))
        kept = np.ctypeslib.as_array(C.cast(v.anchor_idx, C.POINTER(C.c_int32)), (nk,)).copy()
        first0 = np.ctypeslib.as_array(C.cast(v.first0, C.POINTER(C.c_int32)), (nk,)).copy()
        ref_off = np.ctypeslib.as_array(C.cast(v.ref_off, C.POINTER(C.c_int32)), (ns + 1,))
        refs_all = C.string_at(v.refs, int(ref_off[ns])).decode("ascii")
        ro = ref_off.tolist()                                       # plain ints: slicing with numpy scalars is ~1 us apiece
        refs = [refs_all[a:b] for a, b in zip(ro, ro[1:])]
    finally:
        L.nc_pass2_free(h)
    kept_l = kept.tolist()
    pos = [int(anchors[k]) for k in kept_l]
    preds = allele_prediction_batch(cns_str, refs, [max_range[variants[k if by_index else int(anchors[k])]] for k in kept_l for _ in range(S)],
                                    eng=eng)
    xh = x.view(nk, S, 5, 128, 2) if device_x else x.cpu().numpy().astype(np.float64).reshape(nk, S, 5, 128, 2)
    tail = (kept_l,) if by_index else ()
    if haploid:
        return (pos, xh[:, 0], preds) + tail
    hap, ps = dec["hap"], dec["ps"]
    alleles = [[preds[3 * k], preds[3 * k + 1], preds[3 * k + 2]] for k in range(nk)]
    phase = [(p if h else None) for h, p in zip(hap[first0].tolist(), ps[first0].tolist())]
    return (pos, xh[:, 0], xh[:, 1], xh[:, 2], alleles, phase) + tail


def get_indel_testing_candidates_batch(dct, chunks, device=0, haploid=False, device_x=False, device_route=True):
    """get_indel_testing_candidates[_haploid] for ALL chunks of one contig and BAM in one go -> list of the per-chunk tuples
    (each exactly what the per-chunk call returns): pass 1 of every chunk in the same launches (nc_indel_scan_batch), the
    anchors of all chunks through ONE native pass-2 call, ONE device star alignment and ONE allele batch.  What
    indelCaller.indel_run uses; the device / native route only (no external aligner).  device_x: the tensors of the tuples are
    torch float32 tensors on the device (no download / float64 conversion / upload round trip on the way to the CNN)."""
    chunks = list(chunks)
    if not chunks:
        return []
    # A whole chromosome is hundreds of chunks and ~10^5 anchors: the read windows of all of them do not fit the native
    # assembler's 1 GiB arrays.  Chunks are independent (a chunk's tuple is what the per-chunk call returns), so long lists go
    # through in groups, and a group that still overflows is halved.
    if len(chunks) > MAX_BATCH_CHUNKS:
        out = []
        for i in range(0, len(chunks), MAX_BATCH_CHUNKS):
            out += get_indel_testing_candidates_batch(dct, chunks[i:i + MAX_BATCH_CHUNKS], device=device, haploid=haploid, device_x=device_x,
                                                      device_route=device_route)
        return out
    try:
        with _gc_paused():
            return _indel_batch(dct, chunks, device, haploid, device_x, device_route)
    except _lib.NanoCallerHipError as e:
        if getattr(e, "status", None) != _lib.NC_ERR_CAPACITY or len(chunks) == 1:
            raise
    h = len(chunks) // 2
    return (get_indel_testing_candidates_batch(dct, chunks[:h], device=device, haploid=haploid, device_x=device_x, device_route=device_route) +
            get_indel_testing_candidates_batch(dct, chunks[h:], device=device, haploid=haploid, device_x=device_x, device_route=device_route))


@contextlib.contextmanager
def _gc_paused():
    """The batch builds ~10^5 result objects (tuples, strings) that all survive: every threshold they cross starts a
    collection that finds nothing, and a full one walks the whole heap (measured: 53 ms of a 125 ms batch of 40 chunks went into
    one generation-2 pass started from the allele list).  Collections are postponed to the end of the call."""
    if os.environ.get("NC_KEEP_GC") or not gc.isenabled():
        yield
        return
    gc.disable()
    try:
        yield
    finally:
        gc.enable()


MAX_BATCH_CHUNKS = 64          # chunks per native pass-2 / alignment call (100 kb chunks at 30x: ~15 k anchors, ~150 MB of read windows)



# ------------------------------------------------------------------------------------------------- device-resident pass 1 + 2
TAIL_CAP = 272                 # query bases kept behind a read's last aligned base (>= the longest window, 260 + rounding)


def device_indel_reads(ctg, flag, dp, device):
    """The per-read inputs of the device pass 2 beyond the read pack: inserted bases of every insertion event, the bases that
    follow the last aligned one, PS tags (nc_indel_pack_build), uploaded once per (contig, flag filter).  -> (IndelReadsC, tensors)"""
    key = ("dev_reads", flag, device)
    if key in ctg and ctg[key][2] is not dp:                         # the pack was rebuilt (LRU eviction, release_contig): these pointers belong to the old one
        del ctg[key]
    if key not in ctg:
        L = _lib.lib()
        keep = ctg["keep"][flag]
        h = C.c_void_p()
        rc = L.nc_indel_pack_build(ctg["handle"], _lib.npp(keep), TAIL_CAP, C.byref(h))
        if rc != _lib.NC_OK:
            raise _lib.NanoCallerHipError("nc_indel_pack_build failed (%d)" % rc)
        try:
            v = _lib.IndelPackArraysC()
            L.nc_indel_pack_view(h, C.byref(v))
            if dp.events is None or dp.reads is None or v.n_reads != dp.events["n_reads"] or v.n_reads != dp.reads["n_reads"]:
                err = _lib.NanoCallerHipError("the contig's read pack carries no indel events / read table for %d kept reads" % v.n_reads)
                err.status = _lib.NC_ERR_CAPACITY                   # the callers fall back to the host-assembled route
                raise err
            dev = get_engine(device).device

            def up(ptr, cnt, dt):
                a = np.frombuffer((C.c_char * (int(cnt) * np.dtype(dt).itemsize)).from_address(ptr), dt) if cnt and ptr else np.zeros(0, dt)
                a = a if a.size else np.zeros(4, dt)
                return torch.from_numpy(a.copy()).to(dev)
            t = dict(ins_off=up(v.ins_off, v.n_events + 1, np.int32), ins_bases=up(v.ins_bases, v.n_ins_bases, np.uint8),
                     tail_off=up(v.tail_off, v.n_reads + 1, np.int32), tail_bases=up(v.tail_bases, v.n_tail_bases, np.uint8),
                     read_ps=up(v.read_ps, v.n_reads, np.int32), read_flag=up(v.read_flag, v.n_reads, np.uint8))
        finally:
            L.nc_indel_pack_free(h)
        ev, rd = dp.events, dp.reads
        st = _lib.IndelReadsC(n_reads=rd["n_reads"], slot_off=rd["slot_off"].data_ptr(), rd_start=rd["rd_start"].data_ptr(), rd_end=rd["rd_end"].data_ptr(),
                              ev_off=ev["ev_off"].data_ptr(), ev_pos=ev["ev_pos"].data_ptr(), ev_len=ev["ev_len"].data_ptr(),
                              ins_off=t["ins_off"].data_ptr(), ins_bases=t["ins_bases"].data_ptr(), tail_off=t["tail_off"].data_ptr(),
                              tail_bases=t["tail_bases"].data_ptr(), read_ps=t["read_ps"].data_ptr(), read_hap=ev["read_hap"].data_ptr(),
                              read_flag=t["read_flag"].data_ptr())
        ctg[key] = (st, t, dp)                                       # dp: keeps the pack's tensors alive with the pointers
    return ctg[key][0]


def indel_sites_device(eng, dp, reads_c, chrom_len, chunks, *, mincov, maxcov, win_size, small_win_size, ins_t, del_t, window_after,
                       haploid=False, excl=None, fetch=True, impute=False):
    """nc_indel_sites_plan + _run (+ _fetch) for a list of (start, end) chunks of one contig -> dict: n, x (device float32
    [n, sets * 5, 128, 2]: the CNN input), and with fetch: pos / chunk / type / phase int32 [n], ref_len / alt_len int32 [n, sets],
    alt (uint8 codes, the ALT prefixes back to back in (site, set) order).  Raises NanoCallerHipError with .status."""
    L = eng.L
    prm = _lib.IndelScanParamsC(mincov=int(mincov), win_size=int(win_size), small_win_size=int(small_win_size), ins_t=float(ins_t),
                                del_t=float(del_t), haploid=1 if haploid else 0, impute=1 if (impute and not haploid) else 0)
    starts = np.ascontiguousarray([c[0] for c in chunks], np.int32)
    ends = np.ascontiguousarray([c[1] for c in chunks], np.int32)
    pc = dp.c_struct()
    n, na = C.c_int32(), C.c_int64()

    def check(rc, what):
        if rc != _lib.NC_OK:
            msg = L.nc_last_error(eng.ctx)
            err = _lib.NanoCallerHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
            err.status = rc
            raise err
    check(L.nc_indel_sites_scoring(eng.ctx, *[int(v) for v in _lib.STAR_SCORING]), "nc_indel_sites_scoring")
    check(L.nc_indel_sites_plan(eng.ctx, C.byref(pc), C.c_void_p(dp.ref_code.data_ptr()), dp.tile_pos0, dp.ref_code.numel(), int(chrom_len),
                                C.byref(reads_c), C.c_void_p(excl.data_ptr()) if excl is not None else None, len(chunks), _lib.npp(starts),
                                _lib.npp(ends), C.byref(prm), int(window_after), int(maxcov), C.byref(n), C.byref(na)), "nc_indel_sites_plan")
    S = 1 if haploid else 3
    N = n.value
    x = torch.empty((N, S * 5, 128, 2), dtype=torch.float32, device=eng.device)
    check(L.nc_indel_sites_run(eng.ctx, C.c_void_p(x.data_ptr())), "nc_indel_sites_run")
    out = dict(n=N, sets=S, x=x, n_alignments=na.value)
    if fetch:
        out.update(indel_sites_fetch(eng, N, S))
    return out


def indel_sites_fetch(eng, N, S):
    L = eng.L
    pos, chunk, typ, phase = (np.empty(max(N, 1), np.int32) for _ in range(4))
    rl, al = np.empty((max(N, 1), S), np.int32), np.empty((max(N, 1), S), np.int32)
    nb = C.c_int64()
    rc = L.nc_indel_sites_fetch(eng.ctx, _lib.npp(pos), _lib.npp(chunk), _lib.npp(typ), _lib.npp(phase), _lib.npp(rl), _lib.npp(al), C.byref(nb))
    if rc != _lib.NC_OK:
        msg = L.nc_last_error(eng.ctx)
        err = _lib.NanoCallerHipError("nc_indel_sites_fetch failed (%d): %s" % (rc, msg.decode() if msg else ""))
        err.status = rc
        raise err
    alt = np.empty(max(int(nb.value), 1), np.uint8)
    if nb.value:
        rc = L.nc_indel_sites_fetch_alt(eng.ctx, _lib.npp(alt), int(nb.value))
        if rc != _lib.NC_OK:
            raise _lib.NanoCallerHipError("nc_indel_sites_fetch_alt failed (%d)" % rc)
    return dict(pos=pos[:N], chunk=chunk[:N], type=typ[:N], phase=phase[:N], ref_len=rl[:N], alt_len=al[:N], alt=alt[:int(nb.value)])


_DEV_INGEST = {}              # one contig at a time: (BAM, contig, FASTA, flag filter, device, file identity) -> (pack, nc_indel_reads, contig dict)


def _device_ingest_contig(dct, sam_path, chrom, supp, device):
    """The contig's read pack WITH the indel sections, made on the device from the BAM file itself (device_bam.py: inflate, record walk, codes,
    events, inserted bases and tails in HBM) -- what decoded_contig + device_pack + device_indel_reads assemble on host threads.  None when the
    input cannot take that route (no .bai, too large, NC_DEVICE_INGEST=0 / dct['device_ingest'] = False): the host route follows."""
    if not dct.get("device_ingest", os.environ.get("NC_DEVICE_INGEST", "1") != "0") or not isinstance(sam_path, str) or not os.path.exists(sam_path):
        return None
    st = os.stat(sam_path)
    key = (sam_path, chrom, dct["fasta_path"], supp, device, st.st_size, st.st_mtime_ns)
    if key not in _DEV_INGEST:
        from .bam import read_fasta_bytes
        from .device_bam import DeviceIngestUnavailable, open_device_bam
        from .wire import indel_reads_struct
        _DEV_INGEST.clear()
        try:
            dbam = open_device_bam(sam_path, device, contigs=[chrom])
        except DeviceIngestUnavailable:
            return None
        fasta_b = read_fasta_bytes(dct["fasta_path"], chrom)
        dp = dbam.pack(dbam.prepare(chrom, fasta_b, supplementary=supp), indel=True, tail_cap=TAIL_CAP)
        _DEV_INGEST[key] = (dp, indel_reads_struct(dp), dict(fasta=fasta_b.decode("ascii"), fasta_b=fasta_b, device_ingest=True))
    return _DEV_INGEST[key]


def _indel_pack_for(dct, chunks, device):
    """(engine, read pack with the indel sections, nc_indel_reads struct, contig dict, exclusion mask or None) of the chunks' BAM and contig: made on the
    device from the BAM file itself where that route is open (_device_ingest_contig), else from the host decode"""
    chrom, sam_path = chunks[0]["chrom"], chunks[0]["sam_path"]
    supp = bool(dct.get("supplementary"))
    eng = get_engine(device)
    eng.use_torch_stream()
    di = _device_ingest_contig(dct, sam_path, chrom, supp, device)
    if di is not None:
        dp, reads_c, ctg = di
    else:
        ctg = decoded_contig(sam_path, chrom, dct["fasta_path"])
        flag = 0x4 | 0x100 | 0x200 | 0x400 | (0 if supp else 0x800)
        if flag not in ctg["keep"]:
            from .pack import pileup_depth_cap
            dec = ctg["dec"]
            ctg["keep"][flag] = pileup_depth_cap(dec["read_start"], dec["read_end"], np.ascontiguousarray((dec["read_flag"] & flag) == 0, np.uint8))
        dp = device_pack(sam_path, dct.get("fasta_path"), chrom, supp, None, device)[0]
        reads_c = device_indel_reads(ctg, flag, dp, device)
    excl = None
    excl_rows = _exclude_rows(dct, chrom)
    if excl_rows:
        m = np.zeros(dp.n_tiles * dp.tile_size, np.uint8)
        for (a, b) in excl_rows:
            m[max(0, a - dp.tile_pos0):max(0, b - dp.tile_pos0)] = 1
        excl = torch.from_numpy(m).to(eng.device)
    return eng, dp, reads_c, ctg, excl


def imputed_chunk_mask(dct, chunks, device):
    """impute_indel_phase (generate_indel_pileups.py:278-304) on the device pipeline: which chunks hold a column that meets the rule's COLUMN-level
    predicate (:278-284; K7's col_type 2 on the resident pack).  A chunk without one takes no imputed anchor, and its `variants` are those of the flag
    off -- it runs on the device pipeline; the others (their read grouping needs the pileup strings, :285-304) take the host-assembled route."""
    eng, dp, _, _, excl = _indel_pack_for(dct, chunks, device)
    cols = eng.indel_scan_batch(dp, [(c["start"], c["end"]) for c in chunks], mincov=dct["mincov"], win_size=dct["win_size"],
                                small_win_size=dct["small_win_size"], ins_t=dct["ins_t"], del_t=dct["del_t"], excl=excl, haploid=False, impute=True)
    return [bool((ct == 2).any()) for ct in cols]


def indel_sites_for_chunks(dct, chunks, device, haploid, fetch=True):
    """The device pipeline for chunks (ascending) of one BAM and contig -> (result dict of indel_sites_device, contig dict)"""
    window_after = 260 if dct["seq"] == "pacbio" else 160
    eng, dp, reads_c, ctg, excl = _indel_pack_for(dct, chunks, device)
    r = indel_sites_device(eng, dp, reads_c, len(ctg["fasta"]), [(c["start"], c["end"]) for c in chunks], mincov=dct["mincov"], maxcov=dct["maxcov"],
                           win_size=dct["win_size"], small_win_size=dct["small_win_size"], ins_t=dct["ins_t"], del_t=dct["del_t"],
                           window_after=window_after, haploid=haploid, excl=excl, fetch=fetch, impute=bool(dct.get("impute_indel_phase")))
    return r, ctg


def _indel_batch_device(dct, chunks, device, haploid, device_x):
    """_indel_batch on the device-resident pipeline (impute_indel_phase: only chunks impute_split_chunks sends here): the per-chunk tuples of the reference's calls"""
    order = sorted(range(len(chunks)), key=lambda k: (chunks[k]["start"], chunks[k]["end"]))
    r, ctg = indel_sites_for_chunks(dct, [chunks[k] for k in order], device, haploid)
    tuples = sites_to_tuples(r, len(chunks), ctg["fasta"], haploid, device_x)
    out = [None] * len(chunks)
    for k, t in zip(order, tuples):
        out[k] = t
    return out


def device_route_ok(dct, chunks, haploid, impute_split=False):
    """the device pipeline covers BAM inputs; its plan takes chunks whose starts AND ends ascend (nc_indel_sites_plan): nested or overlapping chunk
    lists, which sorting by (start, end) does not make monotone, go through the host-assembled route instead of failing the worker.  With
    dct['impute_indel_phase'] (diploid) the route is open only to callers that split the chunk list first (impute_split_chunks: chunks with a
    column that meets the rule's predicate take the host-assembled route, whose read grouping needs the pileup strings)"""
    imputes = bool(dct.get("impute_indel_phase")) and not haploid
    if imputes and os.environ.get("NC_IMPUTE_SPLIT") == "0":
        return False
    if not (isinstance(chunks[0]["sam_path"], str) and (impute_split or not imputes or impute_on_device())
            and not os.environ.get("NC_INDEL_HOST_PASS2")):
        return False
    ends = [c["end"] for c in sorted(chunks, key=lambda c: (c["start"], c["end"]))]
    return all(b >= a for a, b in zip(ends, ends[1:]))


def indel_chunks_vcf_text(dct, chunks, device, haploid, model_kind):
    """candidates -> tensors -> Indel_model -> genotype rules -> VCF record text for chunks of one BAM and contig, without a
    Python object per site: nc_indel_sites_* (device), nc_indel_forward (K9), nc_indel_vcf_format (native host rules).
    -> list of bytes, the records of every chunk (what indelCaller.indel_run writes chunk by chunk)"""
    order = sorted(range(len(chunks)), key=lambda k: (chunks[k]["start"], chunks[k]["end"]))
    r, ctg = indel_sites_for_chunks(dct, [chunks[k] for k in order], device, haploid)
    out = [b""] * len(chunks)
    N, S = r["n"], r["sets"]
    if N == 0:
        return out
    eng = get_engine(device)
    probs = eng.indel_forward(model_kind, r["x"]).cpu().numpy()
    del r["x"]
    L = eng.L
    chrom = chunks[0]["chrom"].encode()
    cap = int(N) * (96 + len(chrom)) + 2 * int(np.maximum(r["ref_len"], 0).sum() + np.maximum(r["alt_len"], 0).sum()) * 2 + 4096
    buf = np.empty(cap, np.uint8)
    nb = C.c_int64()
    coff = np.empty(len(chunks) + 1, np.int64)
    pos, chunk, phase = (np.ascontiguousarray(r[k], np.int32) for k in ("pos", "chunk", "phase"))
    rl, al = np.ascontiguousarray(r["ref_len"], np.int32), np.ascontiguousarray(r["alt_len"], np.int32)
    rc = L.nc_indel_vcf_format(chrom, N, _lib.npp(pos), _lib.npp(chunk), len(chunks), _lib.npp(np.ascontiguousarray(probs, np.float32)), S,
                               _lib.npp(rl), _lib.npp(al), _lib.npp(np.ascontiguousarray(r["alt"], np.uint8)), _lib.npp(phase), ctg["fasta_b"],
                               len(ctg["fasta"]), 1 if haploid else 0, _lib.npp(buf), cap, C.byref(nb), _lib.npp(coff))
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("nc_indel_vcf_format failed (%d)" % rc)
    raw = buf[:nb.value].tobytes()
    co = coff.tolist()
    for j, k in enumerate(order):
        out[k] = raw[co[j]:co[j + 1]]
    return out


def sites_to_tuples(r, n_chunks, fasta, haploid, device_x):
    """the per-chunk tuples get_indel_testing_candidates[_haploid] returns, from the arrays of the device pipeline"""
    N, S = r["n"], r["sets"]
    empty = ([], [], []) if haploid else ([], [], [], [], [], [])
    if N == 0:
        return [empty for _ in range(n_chunks)]
    x = r["x"] if device_x else r["x"].cpu().numpy().astype(np.float64)
    pos, chunk = r["pos"].tolist(), r["chunk"]
    rl, al = r["ref_len"].reshape(-1).tolist(), r["alt_len"].reshape(-1).tolist()
    alt_all = np.frombuffer(b"AGTCN", np.uint8)[r["alt"]].tobytes().decode("ascii")
    alleles, o = [], 0
    for k in range(N):
        p, row = pos[k], []
        for t in range(S):
            a, b = rl[k * S + t], al[k * S + t]
            if a < 0:
                row.append((None, None))
            else:
                row.append((fasta[p - 1:p - 1 + a], alt_all[o:o + b]))
            o += max(b, 0)
        alleles.append(row)
    phase = [v if v else None for v in r["phase"].tolist()]          # PS of set 0's first read; a read without the tag (imputed sets): None (:181-183,349)
    bounds = np.searchsorted(chunk, np.arange(n_chunks + 1))                       # sites are chunk-major
    out = []
    for ci in range(n_chunks):
        a, b = int(bounds[ci]), int(bounds[ci + 1])
        if a == b:
            out.append(empty)
        elif haploid:
            out.append((pos[a:b], x[a:b], [alleles[k][0] for k in range(a, b)]))
        else:
            out.append((pos[a:b], x[a:b, 0:5], x[a:b, 5:10], x[a:b, 10:15], alleles[a:b], phase[a:b]))
    return out

def impute_on_device():
    """[r6] the device pipeline groups the reads of the imputed columns itself (k_impute_flags, k_sets<.., true>); NC_IMPUTE_DEVICE=0: round 5's
    split (chunks with a column that meets the rule's predicate take the host-assembled route)"""
    return os.environ.get("NC_IMPUTE_DEVICE", "1") != "0"


def impute_split_chunks(dct, chunks, device, haploid):
    """dct['impute_indel_phase'] on a chunk list the device pipeline could take, when the device does not group the reads itself
    (NC_IMPUTE_DEVICE=0): -> (indices for the device pipeline, indices for the host-assembled route), or None when there is nothing to split
    (flag off, haploid, route closed, or the device takes the rule).  NC_IMPUTE_SPLIT=0: everything on the host-assembled route"""
    if not dct.get("impute_indel_phase") or haploid or os.environ.get("NC_IMPUTE_SPLIT") == "0" or not device_route_ok(dct, chunks, haploid, impute_split=True):
        return None
    if impute_on_device():
        return None
    order = sorted(range(len(chunks)), key=lambda k: (chunks[k]["start"], chunks[k]["end"]))
    mask = imputed_chunk_mask(dct, [chunks[k] for k in order], device)
    dev = sorted(k for k, m in zip(order, mask) if not m)
    host = sorted(k for k, m in zip(order, mask) if m)
    return dev, host


def _indel_batch(dct, chunks, device, haploid, device_x, device_route=True):
    """device_route=False: the caller has already met a capacity limit of the device pipeline for these chunks (indelCaller.indel_run): straight to
    the host-assembled route instead of computing the expensive groups on the device a second time"""
    chrom, sam_path = chunks[0]["chrom"], chunks[0]["sam_path"]
    split = impute_split_chunks(dct, chunks, device, haploid) if device_route else None
    if split is not None and split[0]:
        # impute_indel_phase: the chunks without a column that meets the rule's predicate are the flag-off problem -> device pipeline
        dev, host = split
        off = dict(dct, impute_indel_phase=False)
        out = [None] * len(chunks)
        for k, t in zip(dev, _indel_batch(off, [chunks[k] for k in dev], device, haploid, device_x, True)):
            out[k] = t
        if host:
            for k, t in zip(host, _indel_batch(dct, [chunks[k] for k in host], device, haploid, device_x, False)):
                out[k] = t
        return out
    if device_route and device_route_ok(dct, chunks, haploid):
        # everything between the column decisions and the CNN input on the device (nc_pipe.hip); a capacity limit of that route
        # (sets of > 1024 alignment columns, chunks of > 130 kb) sends the group through the host-assembled route below
        try:
            return _indel_batch_device(dct, chunks, device, haploid, device_x)
        except _lib.NanoCallerHipError as e:
            if getattr(e, "status", None) != _lib.NC_ERR_CAPACITY:
                raise
    window_after = 260 if dct["seq"] == "pacbio" else 160
    extras = [dict() for _ in chunks]
    ctg = decoded_contig(sam_path, chrom, dct["fasta_path"]) if isinstance(sam_path, str) else None   # before pass 1: one decode for both
    variants = scan_indel_candidates(dct, chunks, device, haploid=haploid, extra_variants=None if haploid else extras)
    max_range = {0: max(10, dct["win_size"]), 1: 10}
    per_chunk, flat_anchor, flat_chunk = [], [], []
    for ci, (c, var) in enumerate(zip(chunks, variants)):
        anc = sorted(v for v in var if max(0, c["start"] - 10 - dct["win_size"]) < v <= c["end"])
        per_chunk.append(anc)
        flat_anchor += anc
        flat_chunk += [ci] * len(anc)
    empty = ([], [], []) if haploid else ([], [], [], [], [], [])
    if not flat_anchor:
        return [empty for _ in chunks]
    if ctg is None:
        raise ValueError("the batched indel featuriser reads a BAM file (chunk['sam_path'])")
    order = np.argsort(np.asarray(flat_anchor), kind="stable")                      # ascending over the contig: one sweep over the reads
    anchors = [flat_anchor[i] for i in order]
    owner = [flat_chunk[i] for i in order]
    # every anchor's window lies inside its own chunk's ref_dict range [start-200, end+400] (:174), so the contig-wide range is
    # the same test; variant types / imputed sets are looked up per owner chunk
    merged_var = _ChunkedLookup([variants[ci] for ci in owner], anchors)
    merged_extra = {}
    if not haploid:
        for k, (v, ci) in enumerate(zip(anchors, owner)):
            if v in extras[ci]:
                merged_extra[k] = extras[ci][v]
    res = _pass2_native(dct, merged_var, merged_extra, anchors, ctg, 1, len(ctg["fasta"]), window_after, max_range, device, haploid,
                        by_index=True, device_x=device_x)
    pos, kept = res[0], res[-1]
    out = [[] for _ in chunks]
    for j, k in enumerate(kept):
        out[owner[k]].append(j)
    tuples = []
    for ci in range(len(chunks)):
        sel = out[ci]
        if not sel:
            tuples.append(empty)
        elif haploid:
            tuples.append(([pos[j] for j in sel], res[1][sel], [res[2][j] for j in sel]))
        else:
            tuples.append(([pos[j] for j in sel], res[1][sel], res[2][sel], res[3][sel], [res[4][j] for j in sel], [res[5][j] for j in sel]))
    return tuples


class _ChunkedLookup:
    """variants lookup by ANCHOR INDEX for a merged anchor list (the same position can be an anchor of two adjacent chunks
    with different types)"""

    def __init__(self, var_of_anchor, anchors):
        self.t = [var[v] for var, v in zip(var_of_anchor, anchors)]

    def __getitem__(self, k):
        return self.t[k]


def __getattr__(name):
    # round-1 location of the haploid function; the reference imports it from generate_indel_pileups_haploid (indelCaller.py:9)
    if name == "get_indel_testing_candidates_haploid":
        from .generate_indel_pileups_haploid import get_indel_testing_candidates_haploid
        return get_indel_testing_candidates_haploid
    raise AttributeError(name)
