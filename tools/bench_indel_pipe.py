"""Device-resident indel pipeline at scale (experiment / profiling driver): synthetic ONT contig with planted indels generated in HBM ->
nc_indel_sites_plan / _run -> K9 -> fetch -> native rules.  usage: python tools/bench_indel_pipe.py [length] [reps] [--profile]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nanocaller_amd import _lib
from nanocaller_amd import generate_indel_pileups as gip
from nanocaller_amd.engine import get_engine
from nanocaller_amd.synth_device import make_indel_device_workload
from nanocaller_amd.weights import Weights, get_indel_model

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = get_engine(0)
eng.load_weights(_lib.MODEL_INDEL, Weights(get_indel_model("ONT-HG002")))
t0 = time.perf_counter()
pack, reads_c, info = make_indel_device_workload(eng, L, depth=30.0, seed=812)
print("workload: %.1f s, %d reads, %d events, %d inserted bases, codes %.2f GB" % (time.perf_counter() - t0, info["n_reads"], info["n_events"],
                                                                                  info["n_ins_bases"], pack.codes.numel() / 1e9), flush=True)
chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=160)
contig = np.frombuffer(b"AGTCN", np.uint8)[info["tensors"]["ref"].cpu().numpy()[1:]].tobytes()
for rep in range(reps):
    eng.enable_timing(rep == reps - 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = gip.indel_sites_device(eng, pack, reads_c, L, chunks, fetch=False, **kw)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if os.environ.get("NC_PIPE_STOP_AFTER_FILL"):
        continue
    probs = eng.indel_forward(_lib.MODEL_INDEL, r["x"])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    r.update(gip.indel_sites_fetch(eng, r["n"], r["sets"]))
    ph = probs.cpu().numpy()
    t3 = time.perf_counter()
    N, S = r["n"], r["sets"]
    buf = np.empty(N * 110 + 4 * int(np.maximum(r["ref_len"], 0).sum() + np.maximum(r["alt_len"], 0).sum()) + 4096, np.uint8)
    nb = C.c_int64()
    rc = eng.L.nc_indel_vcf_format(b"chr20", N, _lib.npp(np.ascontiguousarray(r["pos"])), _lib.npp(np.ascontiguousarray(r["chunk"])), len(chunks),
                                   _lib.npp(ph), S, _lib.npp(np.ascontiguousarray(r["ref_len"])), _lib.npp(np.ascontiguousarray(r["alt_len"])),
                                   _lib.npp(r["alt"]), _lib.npp(np.ascontiguousarray(r["phase"])), contig, L, 0, _lib.npp(buf), buf.size, C.byref(nb), None)
    assert rc == 0, rc
    t4 = time.perf_counter()
    n_rec = int((buf[:nb.value] == 10).sum())
    print("rep %d: %d sites (%d alignments, %.1f per site), %d records: featuriser %.1f ms, K9 %.1f ms, fetch %.1f ms, rules+text %.1f ms -> %.0f k sites/s"
          % (rep, N, r["n_alignments"], r["n_alignments"] / max(N, 1), n_rec, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3,
             N / (t4 - t0) / 1e3), flush=True)
ms = np.zeros(6, np.float32)
cells = np.zeros(2, np.int64)
eng.L.nc_indel_sites_stage_ms(eng.ctx, _lib.npp(ms), _lib.npp(cells))
print("stages (ms): plan %.2f, windows %.2f, fill %.2f, trace %.2f, tensor %.2f, alleles %.2f; DP cells %.3g + %.3g -> fill %.0f Gcells/s"
      % (*ms, cells[0], cells[1], cells[0] / (ms[2] * 1e-3) / 1e9 if ms[2] > 0 else 0), flush=True)
print("max HBM %.2f GB" % (torch.cuda.max_memory_allocated() / 1e9))
if os.environ.get("NC_PIPE_STOP_AFTER_FILL"):
    sys.exit(0)
# concordance: planted indels (either haplotype) whose length comes back in an allele called at a site up to 60 bp before them
truth = info["truth"].cpu().numpy()
tp = np.nonzero((truth[0] != 0) | (truth[1] != 0))[0]
tp = tp[(tp > 1000) & (tp < L - 1000)]
apos = r["pos"]
rl, al = r["ref_len"], r["alt_len"]
exact = 0
for p_ in tp.tolist():
    lens = {int(truth[0][p_]), int(truth[1][p_])} - {0}
    lo, hi = np.searchsorted(apos, p_ - 60), np.searchsorted(apos, p_, side="right")
    exact += any(rl[k, t] > 0 and (al[k, t] - rl[k, t]) in lens for k in range(lo, hi) for t in range(S))
print("planted indels: %d, exact length recovered in an allele: %d (%.3f)" % (len(tp), exact, exact / max(1, len(tp))))
