"""The indel half of configs[2] at chr1 size on its own, HBM-resident passes one after the other (experiment; run under rocprofv3 --kernel-trace for
tools/stream_gaps.py: where does the GPU idle inside a chr1-sized indel pass?).  usage: exp_indel_chr1_gaps.py [passes]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nanocaller_amd.engine import get_engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
eng = get_engine(0)
job = bench.IndelJob(eng, bench.CHR1_LEN, seed=4913, name=b"chr1", wire=False)
job.gpu_pass(job.pack, job.reads_c)
ms = []
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = job.gpu_pass(job.pack, job.reads_c)
    torch.cuda.synchronize()
    ms.append((time.perf_counter() - t0) * 1e3)
print("chr1-sized indel pass, pack resident, no rules: %s ms, %d sites" % (" ".join("%.1f" % v for v in ms), r["n"]))
