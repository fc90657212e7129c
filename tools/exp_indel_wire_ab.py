"""Indel pass from pinned host memory at chr20 size, steady ms per pass, with and without the deleted columns implied by the events (experiment)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from nanocaller_amd.engine import get_engine  # noqa: E402
from nanocaller_amd.wire import WireUploader  # noqa: E402

eng = get_engine(0)
for mode in ("1", "0", "1", "0"):
    os.environ["NC_WIRE_DEL_IMPLIED"] = mode
    job = bench.IndelJob(eng, 64_444_167)
    up = WireUploader(eng)
    for _ in range(3):
        job.from_host_pass(up, up.submit(job.wire))
    torch.cuda.synchronize()
    n = 12
    t0 = time.perf_counter()
    nxt = up.submit(job.wire)
    for i in range(n):
        tk = nxt
        nxt = up.submit(job.wire) if i + 1 < n else None
        job.from_host_pass(up, tk)
    torch.cuda.synchronize()
    print("deleted columns implied: %s -> wire %.0f MB, %.2f ms per pass" % (mode, job.wire.nbytes / 1e6, (time.perf_counter() - t0) / n * 1e3), flush=True)
    del job, up
    torch.cuda.empty_cache()
