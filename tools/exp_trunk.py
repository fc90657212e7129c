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
    n = int(os.environ.get("NC_EXP_SITES", 32768 * 8))
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
        for _ in range(12):                                        # the first launches after an idle device run 15-25 % slower (clock ramp): not timed
            eng.snp_forward(_lib.MODEL_SNP, x, rc, sc)
        for _ in range(20):
            eng.snp_forward(_lib.MODEL_SNP, x, rc, sc)
            ms += eng.last_ms(4); nl += eng.last_ms(5)
        if os.environ.get("NC_EXP_WALL"):
            for reps in (1, 3, 10):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                tk = 0.0
                for _ in range(reps):
                    eng.snp_forward(_lib.MODEL_SNP, x, rc, sc)
                    tk += eng.last_ms(4)
                e1.record()
                torch.cuda.synchronize()
                print("   %d forwards back to back: %.4f ms each by stream events around them; trunk launches by their dispatch events %.4f ms each" % (reps, e0.elapsed_time(e1) / reps, tk / reps), flush=True)
        if os.environ.get("NC_EXP_ISOLATED"):
            iso = []
            for _ in range(3):
                torch.cuda.synchronize()
                eng.snp_forward(_lib.MODEL_SNP, x, rc, sc)
                torch.cuda.synchronize()
                iso.append(eng.last_ms(4) / max(eng.last_ms(5), 1))
            print("   isolated launches (device idle before each): %s ms" % " ".join("%.4f" % v for v in iso), flush=True)
        eng.enable_timing(False)
        print("%-32s %s trunk %.4f ms/launch" % (os.path.basename(libpath or "in-tree"), "fp32  " if exact else "fp16x3", ms / nl), flush=True)
    if hasattr(_lib.lib(), "nc_debug_trace"):
        import ctypes
        buf = np.zeros((8, 8, 8), np.uint64)
        _lib.lib().nc_debug_trace(ctypes.c_void_p(buf.ctypes.data))
        if os.environ.get("NC_TRACE_CLOCK"):
            b = buf.reshape(128, 4).astype(np.int64)
            t0 = b[:, 0].min()
            en0, p0, en = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0, (b[:, 2] - t0) / 100.0
            hw = buf.reshape(-1)[504:512].astype(np.int64) & 0xffffffff
            print("   HW_ID of block 3's waves 0..7: " + " ".join("w%d:simd%d,wave%d,cu%d,se%d" % (i, (h >> 4) & 3, h & 15, (h >> 8) & 15, (h >> 13) & 7) for i, h in enumerate(hw)))
            print("   blocks: entry %.1f .. %.1f us, P0 %.1f .. %.1f us, loop end %.1f .. %.1f us" % (en0.min(), en0.max(), p0.min(), p0.max(), en.min(), en.max()))
            return
        if os.environ.get("NC_TRACE_P3"):
            b = buf.astype(np.int64)
            names = {4: "C-light", 5: "C-light", 6: "C-heavy", 7: "C-heavy", 0: "conv2a", 1: "conv2b", 2: "conv3a", 3: "conv3b"}
            skew = bool((b[0, 1:7, 3] > 0).all())      # k5_trunk_lin with NC_LIN_SKEW: event 3 = the deferred epilogue of the conv2 / conv3 waves is done
            for w in range(8 if skew else 0):
                rows = []
                for k in range(1, 7):
                    e = b[w, k]
                    if w >= 4:     # 0 start, 2 MFMAs done, 5 before barrier, 6 after
                        rows.append((0, 0, e[2] - e[0], e[5] - e[2], e[6] - e[5], e[6] - b[w, k - 1][6]))
                    elif w < 2:    # 0 start, 3 deferred epilogue done, 1 MFMAs done, 5 before barrier, 6 after
                        rows.append((0, e[3] - e[0], e[1] - e[3], 0, e[6] - e[5], e[6] - b[w, k - 1][6]))
                    else:          # 0 start, 2 staging done, 3 deferred epilogue done, 1 MFMAs done
                        rows.append((e[2] - e[0], e[3] - e[2], e[1] - e[3], 0, e[6] - e[5], e[6] - b[w, k - 1][6]))
                m = np.array(rows).mean(0)
                print("   %-7s commit %5d | epilogue of the previous step %5d | MFMA loop %5d | epilogue %5d | barrier wait %5d | step %5d" % ((names[w],) + tuple(int(v) for v in m)))
            if skew:
                return
            for w in range(8):
                rows = []
                for k in range(1, 7):
                    e = b[w, k]
                    if w >= 4:     # 0 start, 2 MFMAs done, 5 before barrier, 6 after
                        rows.append((0, e[2] - e[0], e[5] - e[2], 0, 0, e[6] - e[5], e[6] - b[w, k - 1][6]))
                    elif w < 2:    # 0 start, 1 MFMAs done, 5 before barrier, 6 after
                        rows.append((0, e[1] - e[0], e[5] - e[1], 0, 0, e[6] - e[5], e[6] - b[w, k - 1][6]))
                    else:          # 0 start, 2 staging done, 1 MFMAs done, 5 before barrier, 6 after
                        rows.append((e[2] - e[0], e[1] - e[2], e[5] - e[1], 0, 0, e[6] - e[5], e[6] - b[w, k - 1][6]))
                m = np.array(rows).mean(0)
                print("   %-7s commit %5d | MFMA loop %5d | epilogue %5d | MFMA loop 2 %5d | epilogue 2 %5d | barrier wait %5d | step %5d" % ((names[w],) + tuple(int(v) for v in m)))
            return
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
