"""CPU-only: get_chunks against the reference's own output; the converted weight zoo."""
import json
import os

import numpy as np

from nanocaller_amd import weights
from nanocaller_amd.utils import get_chunks
from tests.util import GOLD


def test_get_chunks_matches_reference():
    cases = json.load(open(os.path.join(GOLD, "chunks.json")))
    assert len(cases) >= 6
    for c in cases:
        got = get_chunks([tuple(r) for r in c["regions"]], c["cpu"], **c["kw"])
        assert got == c["chunks"]
    big = [c for c in cases if c["regions"][0][0] == "chr20"][0]
    assert len(big["chunks"]) == 129                                  # BASELINE.json configs[1]: 129 chunks of 500 kb


def test_every_reference_model_name_resolves():
    snp_names = ['NanoCaller1', 'NanoCaller2', 'NanoCaller3', 'ONT-HG001', 'ONT-HG001_GP2.3.8', 'ONT-HG001_GP2.3.8-4.2.2',
                 'ONT-HG001-4_GP4.2.2', 'ONT-HG002', 'ONT-HG002_GP4.2.2_v3.3.2', 'ONT-HG002_GP2.3.4_v3.3.2',
                 'ONT-HG002_GP2.3.4_v4.2.1', 'ONT-HG002_r10.3', 'ONT-HG002_bonito', 'CCS-HG001', 'CCS-HG002',
                 'CCS-HG001-4', 'CLR-HG002', 'haploid']                # snpCaller.py:16-34
    cov = {'ONT-HG002': 48, 'ONT-HG001': 57, 'CCS-HG002': 56, 'CCS-HG001': 57, 'CLR-HG002': 58, 'NanoCaller1': 43,
           'NanoCaller3': 28, 'ONT-HG002_r10.3': 32, 'ONT-HG002_bonito': 51}
    for n in snp_names:
        path, tc = weights.get_SNP_model(n)
        w = weights.Weights(path)
        assert w.kind == (weights.KIND_SNP_HAP if n == 'haploid' else weights.KIND_SNP)
        assert w.flat.size == weights.n_params(w.kind) and np.all(np.isfinite(w.flat))
        if n in cov:
            assert tc == cov[n]
    assert weights.get_SNP_model('NanoCaller2')[0] == weights.get_SNP_model('NanoCaller1')[0]   # snpCaller.py:17
    assert weights.get_SNP_model('no-such-model') == (None, None)
    for n in ['NanoCaller1', 'NanoCaller3', 'ONT-HG001', 'ONT-HG002', 'CCS-HG001', 'CCS-HG002', 'haploid']:
        w = weights.Weights(weights.get_indel_model(n))
        assert w.flat.size == weights.n_params(w.kind)
    w = weights.Weights(weights.get_SNP_model('ONT-HG002')[0])
    assert w.t["conv1_3.k"].shape == (5, 5, 5, 16) and w.t["fc1.k"].shape == (1728, 48) and w.t["A.k"].shape == (17, 2)
