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

# bench.run_pairs, SNP half alone (configs[2]'s `snp_ms`), with and without the host thread's VCF text
u = [bench.PairUnit(c, None, chunks, "chr1")]
for label, patch in (("with VCF text on the host thread", None), ("without host text", lambda *a: 0)):
    orig = bench._pair_host_half
    if patch:
        bench._pair_host_half = patch
    bench.run_pairs(up, 0, params, u, 3, True, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_pairs(up, 0, params, u, 8, True, False)
    torch.cuda.synchronize()
    print("run_pairs SNP alone, %s: %.1f ms per step" % (label, (time.perf_counter() - t0) / 8 * 1e3), flush=True)
    bench._pair_host_half = orig
t0 = time.perf_counter()
n = u[0].snp_text(r)
print("snp_text of one chr1-sized result: %.1f ms, %d bytes" % ((time.perf_counter() - t0) * 1e3, n))
