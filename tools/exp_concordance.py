"""Experiment: why do some planted indels not come back with their exact length? (categories)"""
import collections, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bamio
from nanocaller_amd import generate_indel_pileups as gip

w = bamio.make_pass2_world(seed=41, length=40_000, depth=24, alt_base_frac=0.0)
ev_off, ev_pos, ev_len = w.meta["events"]
tmp = tempfile.mkdtemp()
bam, fa = os.path.join(tmp, "s.bam"), os.path.join(tmp, "s.fa")
bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None))
bamio.write_fasta(fa, w.chrom, w.ref)
truth = collections.Counter()
for r in range(w.n_reads):
    for k in range(ev_off[r], ev_off[r + 1]):
        truth[(int(ev_pos[k]), int(ev_len[k]))] += 1
truth = [k for k, v in sorted(truth.items()) if v >= 5 and 3_000 < k[0] < 37_000]
dct = dict(seq="ont", fasta_path=fa, win_size=40, small_win_size=4, mincov=2, maxcov=160, ins_t=0.4, del_t=0.6, supplementary=False,
           exclude_bed=None, impute_indel_phase=False)
for scoring in [(9, 1, 20, -10), (15, 1, 20, -10), (20, 1, 20, -10), (25, 1, 20, -10), (30, 1, 20, -10), (40, 1, 20, -10), (25, 2, 20, -10), (25, 1, 20, -20), (30, 2, 20, -15)]:
    gip._lib.STAR_SCORING = scoring
    pos, x0, x1, x2, alleles, phase = gip.get_indel_testing_candidates(dct, dict(chrom=w.chrom, start=2_000, end=38_000, sam_path=bam), aligner="device")
    ex = cl = 0
    for (p, ln) in truth:
        diffs = [len(A) - len(R) for a, al in zip(pos, alleles) if a <= p <= a + 60 for (R, A) in al if R is not None]
        ex += ln in diffs
        cl += any(abs(d - ln) <= 3 and d * ln > 0 for d in diffs)
    print("scoring", scoring, "exact %d / %d = %.1f %%, within 3: %d" % (ex, len(truth), 100.0 * ex / len(truth), cl))
gip._lib.STAR_SCORING = (25, 1, 20, -10)
pos, x0, x1, x2, alleles, phase = gip.get_indel_testing_candidates(dct, dict(chrom=w.chrom, start=2_000, end=38_000, sam_path=bam), aligner="device")
tp = [p for p, _ in truth]
cat = collections.Counter()
for i, (p, ln) in enumerate(truth):
    anc = [(a, al) for a, al in zip(pos, alleles) if a <= p <= a + 60]
    if not anc:
        cat["no anchor within 60 bp before the event"] += 1
        continue
    diffs = [len(A) - len(R) for a, al in anc for (R, A) in al if R is not None]
    if ln in diffs:
        cat["exact"] += 1
        continue
    # is there another planted event between the anchor and this one (the walk reports the first indel only)?
    earlier = any(any(a <= q < p for q in tp) for a, _ in anc)
    if earlier:
        cat["another planted event between anchor and this one"] += 1
    elif any(abs(d - ln) <= 3 and d * ln > 0 for d in diffs):
        cat["length off by <= 3"] += 1
    elif not diffs:
        cat["anchor but no allele called"] += 1
    else:
        cat["other length"] += 1
print(len(truth), "planted events with >= 5 carriers;", len(pos), "anchors")
for k, v in cat.most_common():
    print("  %-55s %d" % (k, v))
first = [(p, ln) for (p, ln) in truth if any(a <= p <= a + 60 for a in pos) and not any(any(a <= q < p for q in tp) for a in pos if a <= p <= a + 60)]
ex = sum(1 for (p, ln) in first if ln in [len(A) - len(R) for a, al in zip(pos, alleles) if a <= p <= a + 60 for (R, A) in al if R is not None])
print("events that are the FIRST planted event after an anchor: %d, exact %d = %.1f %%" % (len(first), ex, 100.0 * ex / max(1, len(first))))
