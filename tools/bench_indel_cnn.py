"""Throughput of the indel CNN (K9) on random tensors (experiment / reporting helper)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nanocaller_amd import _lib
from nanocaller_amd.engine import get_engine
from nanocaller_amd.weights import Weights, get_indel_model
eng = get_engine(0)
for name, kind, shape, flop in (("ONT-HG002", _lib.MODEL_INDEL, (15, 128, 2), 18_946_752), ("haploid", _lib.MODEL_INDEL_HAP, (5, 128, 2), 5_040_688)):
    eng.load_weights(kind, Weights(get_indel_model(name)))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    x = torch.rand((n,) + shape, device="cuda") - 0.3
    eng.indel_forward(kind, x); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        eng.indel_forward(kind, x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print("%-10s %d sites: %.2f ms -> %.0f sites/s, %.1f TFLOP/s" % (name, n, dt * 1e3, n / dt, n * flop / dt / 1e12))
