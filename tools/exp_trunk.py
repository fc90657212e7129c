"""Time the SNP CNN trunk kernel of one or more library builds on random sites (experiment harness, not a test).
usage: python tools/exp_trunk.py [build_exp/libnc_X.so ...]   (no args: the in-tree library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(libpath):
    import numpy as np
    import torch
    from nanocaller_amd import _lib
    if libpath:
        _lib.LIB_PATH = os.path.abspath(libpath)
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.weights import Weights, get_SNP_model
    eng = get_engine(0)
    w = Weights(get_SNP_model("ONT-HG002")[0])
    eng.load_weights(_lib.MODEL_SNP, w)
    n = 32768 * 8
    g = torch.Generator(device="cuda").manual_seed(1)
    i16 = os.environ.get("NC_EXP_FP32X") is None                  # product format: int16 site tensors
    x = torch.rand((n, 5, 41, 5), device="cuda", generator=g) * 30
    if i16:
        x = x.to(torch.int16)
        eng.set_tensor_format(int16=True)
    rc = torch.randint(0, 4, (n,), device="cuda", dtype=torch.int32)
    sc = torch.full((n,), 0.9, device="cuda", dtype=torch.float64)
    for exact in ([False, True] if (not libpath and not i16) else [False]):
        eng.set_cnn_precision(exact_fp32=exact)
        eng.snp_forward(_lib.MODEL_SNP, x, rc, sc)
        eng.enable_timing(True)
        ms, nl = 0.0, 0.0
        for _ in range(3):
            eng.snp_forward(_lib.MODEL_SNP, x, rc, sc)
            ms += eng.last_ms(4); nl += eng.last_ms(5)
        eng.enable_timing(False)
        print("%-32s %s trunk %.4f ms/launch" % (os.path.basename(libpath or "in-tree"), "fp32  " if exact else "fp16x3", ms / nl), flush=True)
    if hasattr(_lib.lib(), "nc_debug_trace"):
        import ctypes
        buf = np.zeros((8, 8, 8), np.uint64)
        _lib.lib().nc_debug_trace(ctypes.c_void_p(buf.ctypes.data))
        t0 = buf[:, :, :6][buf[:, :, :6] > 0].min()
        rel = np.where(buf > 0, buf.astype(np.int64) - int(t0), -1)
        for w in range(8):
            print("wave %d:" % w)
            for k in range(8):
                print("   site %d: %s" % (k, " ".join("%7d" % v for v in rel[w, k, :8])))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(sys.argv[2] if len(sys.argv) > 2 else None)
    else:
        libs = sys.argv[1:] or [None]
        for l in libs:
            subprocess.run([sys.executable, __file__, "--one"] + ([l] if l else []), check=False)
