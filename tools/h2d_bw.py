"""PCIe host->device bandwidth of this box for pinned buffers (what bench.py's h2d.achieved_GBs can reach)."""
import time, torch
for mb in (64, 408, 1024):
    h = torch.empty(mb << 20, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    for rep in range(2):
        with torch.cuda.stream(s):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            d.copy_(h, non_blocking=True)
            e1.record(s)
        torch.cuda.synchronize()
    print("H2D %5d MiB pinned: %.2f ms = %.1f GB/s" % (mb, e0.elapsed_time(e1), (mb << 20) / e0.elapsed_time(e1) / 1e6))
    with torch.cuda.stream(s):
        e0.record(s)
        h.copy_(d, non_blocking=True)
        e1.record(s)
    torch.cuda.synchronize()
    print("D2H %5d MiB pinned: %.2f ms = %.1f GB/s" % (mb, e0.elapsed_time(e1), (mb << 20) / e0.elapsed_time(e1) / 1e6))
