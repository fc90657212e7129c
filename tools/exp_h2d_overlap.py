"""Experiment: does a pinned H2D copy on its own stream overlap the hot path's kernels, and at what rate?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nanocaller_amd.engine import get_engine
from nanocaller_amd.utils import get_chunks
from nanocaller_amd import snpCaller

eng = get_engine(0)
L = bench.CHR20_LEN
c = bench.Contig(eng, L, 30.0, "ont", 812, keep_pack=True)
chunks = get_chunks([("chr20", 1, L, "diploid")], cpu=16)
params = bench.snp_params("ONT-HG002", "ont")
for _ in range(3):
    snpCaller.call_chunks(params, chunks, device=0, dpk=c.pack)
dev_buf = torch.empty(c.wire.nbytes, dtype=torch.uint8, device="cuda")
for prio in (0, -1):
    s = torch.cuda.Stream(priority=prio)
    # copies alone
    torch.cuda.synchronize()
    evs = []
    with torch.cuda.stream(s):
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); dev_buf.copy_(c.wire.buf, non_blocking=True); e1.record(s); evs.append((e0, e1))
    torch.cuda.synchronize()
    print("prio %d alone:" % prio, ["%.2f" % a.elapsed_time(b) for a, b in evs])
    # copies while the hot path runs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    prev = None
    for i in range(8):
        cur = snpCaller.call_chunks(params, chunks, device=0, dpk=c.pack, defer=True)
        with torch.cuda.stream(s):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); dev_buf.copy_(c.wire.buf, non_blocking=True); e1.record(s); evs.append((e0, e1))
        if prev is not None:
            prev.result()
        prev = cur
    prev.result()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8 * 1e3
    print("prio %d under compute: step %.2f ms; copies" % (prio, dt), ["%.2f" % a.elapsed_time(b) for a, b in evs])
