import torch, time, os
x = torch.empty(10_000_000, dtype=torch.uint8, device="cuda")
h = torch.empty(10_000_000, dtype=torch.uint8).pin_memory()
s = torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(2):
    t = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(20):
            h.copy_(x, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
print("D2H 20 x 10 MB: %.2f ms -> %.1f GB/s" % (dt * 1e3, 0.2 / dt))
for rep in range(2):
    t = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(20):
            x.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
print("H2D 20 x 10 MB: %.2f ms -> %.1f GB/s" % (dt * 1e3, 0.2 / dt))
