"""Time the SNP CNN stage (trunk + fc1 + heads) of library builds (experiment): python tools/exp_fc1.py [libs...]; rocprofv3 --kernel-trace --stats gives k6_fc1_h3 alone"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(libpath):
    import torch
    from nanocaller_amd import _lib
    if libpath:
        _lib.LIB_PATH = os.path.abspath(libpath)
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.weights import Weights, get_SNP_model
    eng = get_engine(0)
    eng.load_weights(_lib.MODEL_SNP, Weights(get_SNP_model("ONT-HG002")[0]))
    n = 624622
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.rand((n, 5, 41, 5), device="cuda", generator=g) * 30).to(torch.int16)
    eng.set_tensor_format(int16=True)
    rc = torch.randint(0, 4, (n,), device="cuda", dtype=torch.int32)
    sc = torch.full((n,), 0.9, device="cuda", dtype=torch.float64)
    for _ in range(8):
        eng.snp_forward(_lib.MODEL_SNP, x, rc, sc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.use_torch_stream()
    cur = torch.cuda.current_stream()
    e0.record(cur)
    for _ in range(10):
        eng.snp_forward(_lib.MODEL_SNP, x, rc, sc)
    e1.record(cur)
    torch.cuda.synchronize()
    print("%-28s CNN stage %.3f ms per %d sites" % (os.path.basename(libpath or "in-tree"), e0.elapsed_time(e1) / 10, n), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(sys.argv[2] if len(sys.argv) > 2 else None)
    else:
        for l in (sys.argv[1:] or [""]):
            subprocess.run([sys.executable, __file__, "--one"] + ([l] if l else []), check=False)
