"""Host-side mirror of nanocaller_src/generate_indel_pileups_haploid.py: `get_indel_testing_candidates_haploid`.

The haploid featuriser is the diploid one without the HP / PS haplotype split (one read set per anchor, no phase output):
pass 1 on the GPU (nc_indel_scan with haploid = 1, generate_indel_pileups_haploid.py:185-241), pass-2 read windows through
the native BAM reader, star alignment on the device (or MUSCLE / a caller-supplied aligner), rows -> tensor K8 and
allele_prediction (:243-277).  Pinned by the reference's own 3-tuples (tests/golden/indel_pass2.npz).
"""
from __future__ import annotations

import numpy as np

from .engine import get_engine
from .generate_indel_pileups import (_sample_set, allele_prediction, allele_prediction_batch, default_aligner, msa,
                                     scan_indel_candidates, star_aligner)


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
    if aligner is None and default_aligner() is star_aligner:
        aligner = "device"
    if aligner == "device":                                                         # every anchor's read set in one device call
        todo, sets, refs = [], [], []
        for v_pos, win in zip(anchors, d["windows"]):
            a, b = v_pos - window_before, min(chrom_length, v_pos + window_after + 1)
            ref = "".join((fasta[p - 1] if (lo <= p <= hi and fasta[p - 1] in "AGTC") else "N") for p in range(a, b))
            if "N" in ref:
                continue
            picked = _sample_set({names[r]: text for r, text in win}, dct["mincov"], dct["maxcov"])
            if picked is None:
                continue
            todo.append(v_pos)
            sets.append(picked[1])
            refs.append(ref)
        if not todo:
            return empty
        eng = get_engine(device)
        eng.use_torch_stream()
        x, cns_str, _ = eng.star_msa_tensor(sets, refs, cns_as_str=True)
        preds = allele_prediction_batch(cns_str, refs, [max_range[variants[v]] for v in todo])
        return (todo, x.cpu().numpy().astype(np.float64), preds)
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
