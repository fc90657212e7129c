"""Host-side mirror of the reference's indel caller rules (nanocaller_src/indelCaller.py:59-179).

`indel_vcf_lines` / `indel_vcf_lines_haploid` restate the allele / genotype rules and VCF text of `indel_run`;
the CNN they consume is `Indel_model` / `haploid_Indel_model` (nc_indel_forward) and the (5,128,2) tensors come
from nc_indel_tensor.  Candidate detection, read slicing, MUSCLE and parasail stay on the host side of the
boundary; phasing and the bcftools/rtg merge are out of scope.

Arithmetic note: in the reference `batch_prob_all` is a float32 TensorFlow tensor, so QUAL/GQ are evaluated in
float32 (`1e-6 + 1 - p` etc.); that is reproduced with explicit np.float32 operations.
"""
from __future__ import annotations

import numpy as np

from .weights import get_indel_model  # noqa: F401  (same name as indelCaller.py:26)

rev_gt_map = {0: 'hom-ref', 1: 'hom-alt', 2: 'het-ref', 3: 'het-alt'}       # indelCaller.py:14
_F = np.float32


def _q10(x):
    return _F(-10) * np.log10(_F(x))


def indel_vcf_lines(chrom, pos, probs, alleles_seq, phase, prev=0):
    """Diploid rules (indelCaller.py:87-152).  probs float32 [N,4] (hom-ref, hom-alt, het-ref, het-alt);
    alleles_seq[j] = [(ref0, alt0), (ref1, alt1), (ref_total, alt_total)], entries may be (None, None);
    phase[j] = phase-set id or None.  -> (lines, prev) where prev carries the overlap suppression across batches."""
    probs = np.asarray(probs, np.float32)
    pred = np.argmax(probs, axis=1)
    out = []
    for j in range(len(pos)):
        if not pos[j] > prev:
            continue
        p = probs[j]
        if not p[0] <= 0.95:                                         # :95
            continue
        q = _q10(_F(1e-6) + p[0])                                    # :97
        a0, a1, at = alleles_seq[j]
        if pred[j] == 1 and at[0]:                                   # :100
            gq = _q10(_F(1 + 1e-6) - p[1])
            out.append('%s\t%d\t.\t%s\t%s\t%.2f\tPASS\t.\tGT:GQ\t1/1:%.2f\n' % (chrom, pos[j], at[0], at[1], q, gq))
            prev = pos[j] + max(len(at[0]), len(at[1]))
        elif a0[0] and a1[0]:
            if a0[0] == a1[0] and a0[1] == a1[1]:                     # :109
                gq = _q10(_F(1 + 1e-6) - p[1])
                out.append('%s\t%d\t.\t%s\t%s\t%.2f\tPASS\t.\tGT:GQ\t1/1:%.2f\n' % (chrom, pos[j], a0[0], a0[1], q, gq))
                prev = pos[j] + max(len(a0[0]), len(a0[1]))
            else:                                                    # :115-133 het-alt, alleles padded to one REF
                ref1, alt1 = a0
                ref2, alt2 = a1
                ln = min(len(ref1), len(ref2))
                if len(ref1) > len(ref2):
                    ref = ref1
                    alt2 = alt2 + ref1[ln:]
                else:
                    ref = ref2
                    alt1 = alt1 + ref2[ln:]
                gq = _q10(_F(1 + 1e-6) - p[3])
                if phase[j]:
                    out.append('%s\t%d\t.\t%s\t%s,%s\t%.2f\tPASS\t.\tGT:GQ:PS\t1|2:%.2f:%d\n' % (chrom, pos[j], ref, alt1, alt2, q, gq, phase[j]))
                else:
                    out.append('%s\t%d\t.\t%s\t%s,%s\t%.2f\tPASS\t.\tGT:GQ\t1|2:%.2f\n' % (chrom, pos[j], ref, alt1, alt2, q, gq))
                prev = pos[j] + max(len(ref), len(alt1), len(alt2))
        elif a0[0] or a1[0]:                                         # :135-151
            a, gt = (a0, '0|1') if a0[0] else (a1, '1|0')
            gq = _q10(_F(1 + 1e-6) - p[2])
            if phase[j]:
                out.append('%s\t%d\t.\t%s\t%s\t%.2f\tPASS\t.\tGT:GQ:PS\t%s:%.2f:%d\n' % (chrom, pos[j], a[0], a[1], q, gt, gq, phase[j]))
            else:
                out.append('%s\t%d\t.\t%s\t%s\t%.2f\tPASS\t.\tGT:GQ\t%s:%.2f\n' % (chrom, pos[j], a[0], a[1], q, gt, gq))
            prev = pos[j] + max(len(a[0]), len(a[1]))
    return out, prev


def indel_vcf_lines_haploid(chrom, pos, probs, alleles_seq, prev=0):
    """Haploid rules (indelCaller.py:173-179).  probs float32 [N,1] sigmoid; alleles_seq[j] = (ref, alt)."""
    probs = np.asarray(probs, np.float32).reshape(len(pos), -1)
    out = []
    for j in range(len(pos)):
        at = alleles_seq[j]
        pj = probs[j, 0]
        if pos[j] > prev and pj >= 0.5 and at[0]:
            q = _F(-100) * np.log10(_F(1e-6 + 1) - pj)
            out.append('%s\t%d\t.\t%s\t%s\t%.2f\tPASS\t.\tGT:GQ\t1/1:%.2f\n' % (chrom, pos[j], at[0], at[1], q, q))
            prev = pos[j] + max(len(at[0]), len(at[1]))
    return out, prev


INDEL_VCF_HEADER = (                                                 # indelCaller.py:373-383
    '##fileformat=VCFv4.2\n'
    '##FILTER=<ID=PASS,Description="All filters passed">\n'
    '{contigs}'
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
    '##FORMAT=<ID=GQ,Number=1,Type=Float,Description="Genotype Probability">\n'
    '##FORMAT=<ID=PS,Number=1,Type=Integer,Description="Phase set identifier">\n'
    '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t{sample}\n')
