"""DeviceBam.load() of the ONT-like bench file, three times (experiment driver for rocprofv3: where do the load's ~130 ms go?)"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import ont_like_bam
from nanocaller_amd import device_bam
from nanocaller_amd.engine import get_engine

eng = get_engine(0)
tmp = tempfile.mkdtemp()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bam, refs, fasta, st = ont_like_bam.make_files(eng, tmp, n, 9_000_000, depth=30.0, seed0=7000, level=1)
del fasta
for rep in range(int(os.environ.get("NC_EXP_LOADS", 3))):
    device_bam.release()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    db = device_bam.DeviceBam(bam, 0).load()
    torch.cuda.synchronize()
    print("load %d: %.1f ms %s" % (rep, (time.perf_counter() - t0) * 1e3, {k: (round(v * 1e3, 1) if isinstance(v, float) else v) for k, v in device_bam.LAST_LOAD.items()}), flush=True)
