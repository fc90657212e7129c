"""Can a small-register, memory-bound kernel run BESIDE the persistent SNP trunk (229 VGPRs x 2 waves per SIMD, 130 KB LDS per
CU leave 48 VGPRs and 29 KB per CU)?  Stream A: trunk launches; stream B: copies / fills issued at the same time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanocaller_amd import _lib
from nanocaller_amd.engine import get_engine
from nanocaller_amd.weights import Weights, get_SNP_model
eng = get_engine(0)
eng.load_weights(_lib.MODEL_SNP, Weights(get_SNP_model("ONT-HG002")[0]))
n = 262144 * 2
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.rand((n, 5, 41, 5), device="cuda", generator=g) * 30).to(torch.int16)
rc = torch.randint(0, 4, (n,), device="cuda", dtype=torch.int32)
sc = torch.full((n,), 0.9, device="cuda", dtype=torch.float64)
eng.set_tensor_format(int16=True)
A, B = torch.cuda.Stream(), torch.cuda.Stream()
buf = torch.empty(1_930_000_000, dtype=torch.uint8, device="cuda")
src = torch.empty(1_930_000_000, dtype=torch.uint8, device="cuda")
def trunk():
    with torch.cuda.stream(A):
        eng.use_torch_stream()
        eng.snp_forward(_lib.MODEL_SNP, x, rc, sc)
def side(k):
    with torch.cuda.stream(B):
        for _ in range(k):
            buf.copy_(src)
def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
trunk(); side(1)
t_a = min(timed(trunk) for _ in range(3))
t_b = min(timed(lambda: side(4)) for _ in range(3))
t_ab = min(timed(lambda: (trunk(), side(4))) for _ in range(3))
print("CNN of %d sites alone %.2f ms; 4 copies of 1.93 GB alone %.2f ms; both issued together %.2f ms (serial would be %.2f)" % (n, t_a, t_b, t_ab, t_a + t_b))
