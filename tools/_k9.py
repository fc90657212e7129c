import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nanocaller_amd import _lib
from nanocaller_amd.engine import get_engine
from nanocaller_amd.weights import Weights, get_indel_model
eng = get_engine(0)
eng.load_weights(_lib.MODEL_INDEL, Weights(get_indel_model("ONT-HG002")))
n = 32768
xx = torch.rand((n, 15, 128, 2), device="cuda") - 0.3
for _ in range(3):
    eng.indel_forward(_lib.MODEL_INDEL, xx)
torch.cuda.synchronize()
