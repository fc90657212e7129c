#!/usr/bin/env python3
"""Host wall-clock of each call inside one overlapped call_chunks step (no device syncs added), to find GPU idle gaps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nanocaller_amd import snpCaller, _lib
from nanocaller_amd.engine import get_engine
from nanocaller_amd.synth_device import make_device_workload
from nanocaller_amd.utils import get_chunks
from nanocaller_amd.weights import get_SNP_model

eng = get_engine(0)
L = 64_444_167
pack, info = make_device_workload(eng, L)
chunks = get_chunks([("chr20", 1, L, "diploid")], cpu=16)
params = dict(mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6], snp_model="ONT-HG002",
              seq="ont", supplementary=False, exclude_bed=None, disable_coverage_normalization=False, sam_path=None)
for _ in range(2):
    snpCaller.call_chunks(params, chunks, dpk=pack)
path, cov = get_SNP_model("ONT-HG002")
for rep in range(3):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    eng.load_weights(_lib.MODEL_SNP, snpCaller._weights(path)); t.append(time.perf_counter())
    sites = eng.snp_scan(pack, [(c['start'], c['end']) for c in chunks], mincov=4, min_allele_freq=0.15, threshold=[0.4, 0.6], async_fetch=True); t.append(time.perf_counter())
    ev = eng.copy_event(); t.append(time.perf_counter())
    eng.snp_featurize(pack, sites, seq="ont", maxcov=160); t.append(time.perf_counter())
    hs = eng.to_host_async([sites.ref_code, sites.fwd_dp, sites.rev_dp]); t.append(time.perf_counter())
    scale, cd = eng.snp_scale(sites, len(chunks), cov); t.append(time.perf_counter())
    r = eng.snp_forward(_lib.MODEL_SNP, sites.x, sites.ref_code, scale, drain=True); t.append(time.perf_counter())
    ev.synchronize(); t.append(time.perf_counter())
    freq = sites.alt.astype(np.float64) / sites.dp.astype(np.float64); t.append(time.perf_counter())
    eng.wait_copies(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    names = ["weights", "scan", "copy_event", "featurize(launch)", "to_host_async", "scale", "forward(launch)", "ev.sync", "freq", "wait_copies", "final_sync"]
    print(" ".join("%s=%.2f" % (n, (b - a) * 1e3) for n, a, b in zip(names, t, t[1:])), "total=%.2fms" % ((t[-1] - t[0]) * 1e3))
