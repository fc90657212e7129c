"""Experiment: host timeline of bench.run_units with distinct contigs (where does a unit's wall time go?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nanocaller_amd.engine import get_engine
from nanocaller_amd.utils import get_chunks
from nanocaller_amd.wire import WireUploader
from nanocaller_amd import snpCaller

nd = int(sys.argv[1]) if len(sys.argv) > 1 else 3
eng = get_engine(0)
L = bench.CHR20_LEN
contigs = [bench.Contig(eng, L, 30.0, "ont", 812 + k, keep_pack=False) for k in range(nd)]
chunks = get_chunks([("chr20", 1, L, "diploid")], cpu=16)
params = bench.snp_params("ONT-HG002", "ont")
up = WireUploader(eng)
bench.run_units(eng, up, contigs, 5, params, chunks, 0)
torch.cuda.synchronize()
n_units = 12
marks = []
t00 = time.perf_counter()
prev = None
nxt = up.submit(contigs[0].wire)
for i in range(n_units):
    t0 = time.perf_counter()
    t = nxt
    dpk = up.expand(t)
    t1 = time.perf_counter()
    nxt = up.submit(contigs[(i + 1) % nd].wire) if i + 1 < n_units else None
    t2 = time.perf_counter()
    cur = snpCaller.call_chunks(params, chunks, device=0, dpk=dpk, defer=True)
    t3 = time.perf_counter()
    up.release(t)
    if prev is not None:
        prev.result()
    t4 = time.perf_counter()
    prev = cur
    marks.append((t0 - t00, t1 - t0, t2 - t1, t3 - t2, t4 - t3))
prev.result()
torch.cuda.synchronize()
print("total per unit %.2f ms" % ((time.perf_counter() - t00) / n_units * 1e3))
for m in marks:
    print("start %7.2f  expand %.2f  submit %.2f  call_chunks %.2f  result %.2f" % tuple(x * 1e3 for x in m))
