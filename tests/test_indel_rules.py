"""CPU-only: indel allele / genotype rules + VCF text (indelCaller.py:87-179) against lines written by the
REFERENCE's own indel_run() on canned probabilities and allele tuples (tests/golden/indel_caller_vcf.npz)."""
import json
import os

import numpy as np

from nanocaller_amd import indelCaller
from tests.util import GOLD


def _load():
    z = np.load(os.path.join(GOLD, "indel_caller_vcf.npz"))
    alleles = [[tuple(a) for a in site] for site in json.loads(str(z["alleles"]))]
    return z, alleles, json.loads(str(z["phase"]))


def test_diploid_rules_match_reference_indel_run():
    z, alleles, phase = _load()
    pos = z["pos"].tolist()
    # the reference feeds batches of 100 and carries `prev` across them (indelCaller.py:59,75,93)
    lines, prev = [], 0
    for b in range(0, len(pos), 100):
        out, prev = indelCaller.indel_vcf_lines("chr20", pos[b:b + 100], z["probs"][b:b + 100], alleles[b:b + 100],
                                                phase[b:b + 100], prev)
        lines += out
    gold = str(z["vcf_diploid"]).splitlines(keepends=True)
    assert len(gold) > 150 and lines == gold
    gts = {ln.rstrip("\n").split("\t")[9].split(":")[0] for ln in gold}
    assert gts == {"1/1", "1|2", "0|1", "1|0"}
    assert any("GT:GQ:PS" in ln for ln in gold) and any("\tGT:GQ\t" in ln for ln in gold)


def test_haploid_rules_match_reference_indel_run():
    z, alleles, _ = _load()
    pos = z["pos"].tolist()
    lines, _ = indelCaller.indel_vcf_lines_haploid("chr20", pos, z["hap_probs"], [a[2] for a in alleles])
    gold = str(z["vcf_haploid"]).splitlines(keepends=True)
    assert len(gold) > 50 and lines == gold


def test_overlap_suppression_and_homref_skip():
    a = [[("AT", "A"), (None, None), ("AT", "A")]] * 3
    probs = np.float32([[0.1, 0.7, 0.1, 0.1], [0.1, 0.7, 0.1, 0.1], [0.96, 0.02, 0.01, 0.01]])
    lines, prev = indelCaller.indel_vcf_lines("c", [100, 101, 200], probs, a, [None, None, None])
    assert len(lines) == 1 and prev == 102          # 101 <= prev (100 + len('AT')); third site is hom-ref > 0.95
