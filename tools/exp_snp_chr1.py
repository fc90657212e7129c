"""The SNP half of configs[2] alone at chr1 size: wall time per step against the stages' kernel times (experiment driver, GPU)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nanocaller_amd import snpCaller
from nanocaller_amd.engine import get_engine
from nanocaller_amd.utils import get_chunks
from nanocaller_amd.wire import WireUploader

L = int(sys.argv[1]) if len(sys.argv) > 1 else bench.CHR1_LEN
eng = get_engine(0)
up = WireUploader(eng)
c = bench.Contig(eng, L, 30.0, "ont", seed=912, keep_pack=True)
chunks = get_chunks([("chr1", 1, L, "diploid")], cpu=16)
params = bench.snp_params("ONT-HG002", "ont")
for mode in ("resident", "uploaded"):
    for rep in range(4):
        eng.enable_timing(rep == 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "resident":
            r = snpCaller.call_chunks(params, chunks, device=0, dpk=c.pack)
        else:
            t = up.submit(c.wire)
            dpk = up.expand(t)
            r = snpCaller.call_chunks(params, chunks, device=0, dpk=dpk)
            up.release(t)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s rep %d: %d sites, %.1f ms" % (mode, rep, r["n"], dt * 1e3), flush=True)
    ms = eng.last_kernel_ms() if hasattr(eng, "last_kernel_ms") else None
    print("  stage ms:", ms, flush=True)
# pipelined: steps back to back as bench's configs2 SNP-alone leg
import cProfile, pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
t0 = time.perf_counter()
pr.enable()
prev = None
nxt = up.submit(c.wire)
for i in range(5):
    t = nxt
    dpk = up.expand(t)
    nxt = up.submit(c.wire) if i < 4 else None
    cur = snpCaller.call_chunks(params, chunks, device=0, dpk=dpk, defer=True)
    up.release(t)
    if prev is not None:
        prev.result()
    prev = cur
prev.result()
pr.disable()
torch.cuda.synchronize()
print("pipelined: %.1f ms per step" % ((time.perf_counter() - t0) / 5 * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
