"""Host-side mirror of the reference's indel caller (nanocaller_src/indelCaller.py:41-189).

`indel_vcf_lines` / `indel_vcf_lines_haploid` restate the allele / genotype rules and VCF text of `indel_run`;
`indel_run` is the worker loop itself: per chunk the candidates, tensors and allele strings of
generate_indel_pileups.get_indel_testing_candidates[_haploid] (window scan K7, star alignment on the device or MUSCLE, rows ->
tensor K8, allele_prediction), `Indel_model` / `haploid_Indel_model` on the GPU (nc_indel_forward, K9), then the rules.
Phasing (WhatsHap) and the bcftools/rtg merge of the SNP and indel files are out of scope.

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


def indel_run(params, indel_dict, job_Q, counter_Q, indel_files_list, device=0, worker_id=1, aligner=None):
    """Worker with the reference's signature (indelCaller.py:41-189): drains ('indel', chunk) jobs from `job_Q`, writes
    <intermediate_indel_files_dir>/<prefix>.<worker>.indel.vcf.  Per chunk: candidates + tensors + allele strings
    (generate_indel_pileups.get_indel_testing_candidates[_haploid]: window scan, star alignment / MUSCLE, K8), the indel CNN on
    the GPU for the whole chunk (the reference feeds batches of 100; the per-site probabilities do not depend on the batching),
    then the genotype rules in batches of 100 with `prev` carried through the chunk."""
    import os
    import queue
    import sys

    import torch

    from . import _lib
    from .engine import get_engine
    from .generate_indel_pileups import get_indel_testing_candidates, get_indel_testing_candidates_haploid
    from .weights import Weights
    curr_vcf_path = os.path.join(params['intermediate_indel_files_dir'], '%s.%d.indel.vcf' % (params['prefix'], worker_id))
    indel_files_list.append(curr_vcf_path)
    model_path = get_indel_model(params['indel_model'])
    if model_path is None:
        print('Invalid indel model name or path', flush=True)          # indelCaller.py:47-49
        sys.exit(1)
    eng = get_engine(device)
    eng.use_torch_stream()
    eng.load_weights(_lib.MODEL_INDEL, Weights(model_path))
    eng.load_weights(_lib.MODEL_INDEL_HAP, Weights(get_indel_model('haploid')))
    batch_size = 100
    with open(curr_vcf_path, 'w') as f:
        while len(indel_dict) > 0 or not job_Q.empty():
            try:
                job = job_Q.get(block=False)
            except queue.Empty:
                if len(indel_dict) > 0:
                    continue
                break
            chunk = job[1]
            chrom = chunk['chrom']
            if chunk['ploidy'] == 'diploid':
                pos, x0, x1, x2, alleles_seq, phase = get_indel_testing_candidates(params, chunk, aligner=aligner, device=device)
                if len(pos) != 0:
                    x_all = np.hstack([x0, x1, x2]).astype(np.float32)                       # :82 -> (n, 15, 128, 2)
                    probs = eng.indel_forward(_lib.MODEL_INDEL, torch.from_numpy(np.ascontiguousarray(x_all)).to(eng.device)).cpu().numpy()
                    prev = 0
                    for b in range(0, len(pos), batch_size):
                        lines, prev = indel_vcf_lines(chrom, pos[b:b + batch_size], probs[b:b + batch_size], alleles_seq[b:b + batch_size],
                                                      phase[b:b + batch_size], prev)
                        f.writelines(lines)
            elif chunk['ploidy'] == 'haploid':
                pos, x, alleles_seq = get_indel_testing_candidates_haploid(params, chunk, aligner=aligner, device=device)
                if len(pos) != 0:
                    probs = eng.indel_forward(_lib.MODEL_INDEL_HAP, torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(eng.device)).cpu().numpy()
                    prev = 0
                    for b in range(0, len(pos), batch_size):
                        lines, prev = indel_vcf_lines_haploid(chrom, pos[b:b + batch_size], probs[b:b + batch_size], alleles_seq[b:b + batch_size], prev)
                        f.writelines(lines)
            f.flush()
            os.fsync(f.fileno())
            counter_Q.put(1)
    return curr_vcf_path
