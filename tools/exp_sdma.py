"""Experiment: latency of a small D2H copy (and of a kernel) while a large pinned H2D copy is in flight on another stream."""
import time, torch
big_h = torch.empty(400 << 20, dtype=torch.uint8, pin_memory=True)
big_d = torch.empty(400 << 20, dtype=torch.uint8, device="cuda")
small_d = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
small_h = torch.empty(1 << 20, dtype=torch.uint8, pin_memory=True)
x = torch.zeros(1 << 24, device="cuda")
s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
def small_copy():
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(s_dn):
        small_h.copy_(small_d, non_blocking=True)
    s_dn.synchronize()
    return (time.perf_counter() - t) * 1e3
def small_kernel():
    t = time.perf_counter()
    x.add_(1.0)
    torch.cuda.current_stream().synchronize()
    return (time.perf_counter() - t) * 1e3
for _ in range(3):
    small_copy(); small_kernel()
print("idle: small D2H %.3f ms, small kernel %.3f ms" % (small_copy(), small_kernel()))
for mode in ("one 400 MiB copy", "25 copies of 16 MiB"):
    torch.cuda.synchronize()
    with torch.cuda.stream(s_up):
        if mode.startswith("one"):
            big_d.copy_(big_h, non_blocking=True)
        else:
            for k in range(25):
                big_d[k << 24:(k + 1) << 24].copy_(big_h[k << 24:(k + 1) << 24], non_blocking=True)
    t = time.perf_counter()
    with torch.cuda.stream(s_dn):
        small_h.copy_(small_d, non_blocking=True)
    s_dn.synchronize()
    t1 = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    x.add_(1.0)
    torch.cuda.current_stream().synchronize()
    t2 = (time.perf_counter() - t) * 1e3
    s_up.synchronize()
    t3 = (time.perf_counter() - t) * 1e3
    print("%s in flight: small D2H done after %.3f ms, small kernel %.3f ms, big copy done after %.3f ms" % (mode, t1, t2, t3))
