"""Indel CNN (K9) alone on random tensors: sites/s; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nanocaller_amd import _lib
from nanocaller_amd.engine import get_engine
from nanocaller_amd.weights import Weights, get_indel_model
eng = get_engine(0)
eng.load_weights(_lib.MODEL_INDEL, Weights(get_indel_model("ONT-HG002")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
x = (torch.rand((n, 5, 384, 2), device="cuda") - 0.3).contiguous()
eng.indel_forward(_lib.MODEL_INDEL, x)
torch.cuda.synchronize()
for rep in range(3):
    t = time.perf_counter()
    eng.indel_forward(_lib.MODEL_INDEL, x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("K9: %d sites in %.2f ms -> %.2f M sites/s, %.1f TFLOP/s fp32-equivalent" % (n, dt * 1e3, n / dt / 1e6, 18_946_752 * n / dt / 1e12))
