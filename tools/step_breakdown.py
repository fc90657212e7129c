#!/usr/bin/env python3
"""Wall-clock breakdown of one bench step (host + device), for finding host-side overheads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nanocaller_amd import snpCaller, _lib
from nanocaller_amd.engine import get_engine
from nanocaller_amd.synth_device import make_device_workload
from nanocaller_amd.utils import get_chunks
from nanocaller_amd.weights import Weights, get_SNP_model

eng = get_engine(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 64_444_167
pack, info = make_device_workload(eng, L)
chunks = get_chunks([("chr20", 1, L, "diploid")], cpu=16)
params = dict(mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6], snp_model="ONT-HG002",
              seq="ont", supplementary=False, exclude_bed=None, disable_coverage_normalization=False, sam_path=None)
for _ in range(2):
    snpCaller.call_chunks(params, chunks, dpk=pack)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(2):
    t = [T()]
    path, cov = get_SNP_model("ONT-HG002"); w = Weights(path); eng.load_weights(_lib.MODEL_SNP, w); t.append(T())
    sites = eng.snp_scan(pack, [(c['start'], c['end']) for c in chunks], mincov=4, min_allele_freq=0.15, threshold=[0.4, 0.6]); t.append(T())
    eng.snp_featurize(pack, sites, seq="ont", maxcov=160); t.append(T())
    scale, cd = eng.snp_scale(sites, len(chunks), cov); t.append(T())
    probs, gt = eng.snp_forward(_lib.MODEL_SNP, sites.x, sites.ref_code, scale); t.append(T())
    outs = eng.to_host([sites.ref_code, probs, gt, sites.fwd_dp, sites.rev_dp]); t.append(T())
    freq = sites.alt.astype(np.float64) / sites.dp.astype(np.float64); t.append(T())
    names = ["weights", "scan", "featurize", "scale", "forward", "to_host", "freq"]
    print(" ".join("%s=%.2fms" % (n, (b - a) * 1e3) for n, a, b in zip(names, t, t[1:])), "total=%.2fms" % ((t[-1] - t[0]) * 1e3))
