"""Host-side mirror of nanocaller_src/generate_indel_pileups_haploid.py: `get_indel_testing_candidates_haploid`.

The haploid featuriser is the diploid one without the HP / PS haplotype split (one read set per anchor, no phase output):
pass 1 on the GPU (nc_indel_scan with haploid = 1, generate_indel_pileups_haploid.py:185-241), pass-2 read windows through
the native BAM reader, star alignment on the device (or MUSCLE / a caller-supplied aligner), rows -> tensor K8 and
allele_prediction (:243-277).  Pinned by the reference's own 3-tuples (tests/golden/indel_pass2.npz).
"""
from __future__ import annotations

import numpy as np

from .generate_indel_pileups import (_pass2_native, allele_prediction, decoded_contig, default_aligner, msa, scan_indel_candidates,
                                     star_aligner)


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
    lo, hi = max(1, start - 200), end + 400
    flag = 0x4 | 0x100 | 0x200 | 0x400 | (0 if dct.get("supplementary") else 0x800)
    anchors = sorted(v for v in variants if max(0, start - 10 - dct["win_size"]) < v <= end)
    max_range = {0: max(10, dct["win_size"]), 1: 10}
    if aligner is None and default_aligner() is star_aligner:
        aligner = "device"
    if aligner == "device":                                                         # every anchor's read set in one device call
        ctg = decoded_contig(chunk["sam_path"], chrom, dct["fasta_path"])
        return _pass2_native(dct, variants, {}, anchors, ctg, lo, hi, window_after, max_range, device, haploid=True)
    fasta = read_fasta(dct["fasta_path"], chrom)
    chrom_length = len(fasta)
    bf = BamFile(chunk["sam_path"])
    d = bf.decode(chrom, max(1, start - 10 - dct["win_size"] - window_after), end + 1000, anchors=anchors, window_before=window_before,
                  window_after=window_after, keep_mask=flag)
    bf.close()
    names = d["names"]
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
