"""Time nc_indel_forward alone (experiment driver for the fused indel trunk): python tools/exp_k10.py [n_sites]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nanocaller_amd import _lib
from nanocaller_amd.engine import get_engine
from nanocaller_amd.weights import Weights, get_indel_model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
eng = get_engine(0)
eng.load_weights(_lib.MODEL_INDEL, Weights(get_indel_model("ONT-HG002")))
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand((n, 15, 128, 2), device="cuda", generator=g) * (torch.rand((n, 15, 128, 2), device="cuda", generator=g) < 0.3)
for _ in range(3):
    p = eng.indel_forward(_lib.MODEL_INDEL, x)
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 10
for _ in range(R):
    p = eng.indel_forward(_lib.MODEL_INDEL, x)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / R * 1e3
print("%s: %d sites, %.3f ms per forward -> %.2f M sites/s, %.1f TFLOP/s" % (os.environ.get("NANOCALLER_HIP_LIB", "default").split("/")[-1], n, ms, n / ms / 1e3, 18_946_752 * n / ms / 1e9), flush=True)
