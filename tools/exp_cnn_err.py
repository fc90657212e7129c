"""Per-model error of the SNP CNN kernels against the reference-executed goldens (experiment harness; mirrors
tests/test_cnn_golden.py::test_hip_snp_cnn_equals_reference_call but prints every case)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from nanocaller_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from nanocaller_amd.engine import get_engine
from nanocaller_amd.weights import Weights, get_SNP_model
import test_cnn_golden as T
eng = get_engine(0)
worst = 0.0
for k in range(int(T.ZS["n"])):
    model = str(T.ZS["c%d_model" % k])
    case, n, mode, x, ref_code, depth, dp, out = T._snp_case("c", k)
    path, cov = get_SNP_model(model)
    eng.load_weights(_lib.MODEL_SNP, Weights(path))
    sc = torch.from_numpy(T._scale(cov, depth, dp, mode, n)).cuda()
    probs, gt = eng.snp_forward(_lib.MODEL_SNP, torch.from_numpy(np.ascontiguousarray(x)).cuda(), torch.from_numpy(ref_code).cuda(), sc, scale_mode=mode)
    p, g = probs.cpu().numpy(), gt.cpu().numpy()
    err = max(np.abs(p - out[:, :4, 1]).max(), np.abs(g - out[:, 4, :]).max())
    worst = max(worst, err)
    print("%-28s %-10s n=%5d mode=%d err %.2e" % (model, case, n, mode, err))
for k in range(int(T.ZS["nh"])):
    case, n, mode, x, ref_code, depth, dp, out = T._snp_case("h", k)
    eng.load_weights(_lib.MODEL_SNP_HAP, Weights(get_SNP_model("haploid")[0]))
    sc = torch.from_numpy(T._scale(30.0, depth, dp, mode, n)).cuda()
    probs, _ = eng.snp_forward(_lib.MODEL_SNP_HAP, torch.from_numpy(np.ascontiguousarray(x)).cuda(), torch.from_numpy(ref_code).cuda(), sc, scale_mode=mode)
    err = np.abs(probs.cpu().numpy() - out).max()
    worst = max(worst, err)
    print("%-28s %-10s n=%5d mode=%d err %.2e" % ("haploid", case, n, mode, err))
print("worst %.3e" % worst)
