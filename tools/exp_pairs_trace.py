"""Host timeline of bench.run_pairs' SNP half at chr1 size (experiment): where does a step's wall time go?"""
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
c = bench.Contig(eng, L, 30.0, "ont", seed=912, keep_pack=False)
chunks = get_chunks([("chr1", 1, L, "diploid")], cpu=16)
params = bench.snp_params("ONT-HG002", "ont")
T0 = time.perf_counter()
def ts(msg):
    print("%8.1f ms  %s" % ((time.perf_counter() - T0) * 1e3, msg), flush=True)
def enqueue(tk):
    ts("  expand ...")
    dpk = up.expand(tk)
    ts("  call_chunks ...")
    cc = snpCaller.call_chunks(params, chunks, device=0, dpk=dpk, defer=True)
    ts("  call_chunks returned")
    up.release(tk)
    return cc
for variant in ("late", "early"):
    print("variant: the next copy submitted %s" % ("right before its expansion (bench.run_pairs)" if variant == "late" else "one step ahead (before the current step's call_chunks)"))
    torch.cuda.synchronize()
    T0 = time.perf_counter()
    if variant == "late":
        cur = enqueue(up.submit(c.wire))
        for i in range(6):
            ts("step %d: submit next" % i)
            nxt = up.submit(c.wire)
            cn = enqueue(nxt)
            ts("  result(cur) ...")
            cur.result()
            ts("  result done")
            cur = cn
        cur.result()
    else:
        nxt = up.submit(c.wire)
        prev = None
        for i in range(7):
            t = nxt
            ts("step %d: expand" % i)
            dpk = up.expand(t)
            nxt = up.submit(c.wire) if i < 6 else None
            ts("  call_chunks ...")
            cur = snpCaller.call_chunks(params, chunks, device=0, dpk=dpk, defer=True)
            ts("  call_chunks returned")
            up.release(t)
            if prev is not None:
                prev.result()
                ts("  result(prev) done")
            prev = cur
        prev.result()
    torch.cuda.synchronize()
    ts("end")
