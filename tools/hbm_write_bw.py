"""HBM write / copy bandwidth of plain torch kernels (context for k_wire_expand's 1 B written per pileup entry)."""
import torch
n = 1_930_000_000
x = torch.empty(n, dtype=torch.uint8, device="cuda")
y = torch.empty(n, dtype=torch.uint8, device="cuda")
xi = x.view(torch.int32)[: n // 4 // 4 * 4]
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: xi.fill_(7)); print("fill int32   %.3f ms  %.2f TB/s written" % (ms, xi.numel() * 4 / ms / 1e9))
ms = t(lambda: x.zero_()); print("zero (memset) %.3f ms  %.2f TB/s written" % (ms, n / ms / 1e9))
ms = t(lambda: y.copy_(x)); print("copy         %.3f ms  %.2f TB/s read + %.2f TB/s written" % (ms, n / ms / 1e9, n / ms / 1e9))
ms = t(lambda: x.view(torch.int64).sum()); print("sum int64    %.3f ms  %.2f TB/s read" % (ms, n / ms / 1e9))
