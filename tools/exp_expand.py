"""Time k_wire_expand of library builds on a chr20-sized SNP wire (experiment): python tools/exp_expand.py [libs...]"""
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
    import bench
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.wire import WireUploader
    eng = get_engine(0)
    c = bench.Contig(eng, 64_444_167, 30.0, "ont", seed=812, keep_pack=False)
    up = WireUploader(eng)
    t = up.submit(c.wire)
    up.expand(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record(cur)
    for _ in range(10):
        up.expand(t)
    e1.record(cur)
    torch.cuda.synchronize()
    print("%-24s expand %.3f ms (%.0f GB/s written)" % (os.path.basename(libpath or "in-tree"), e0.elapsed_time(e1) / 10, c.wire.codes_len / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(sys.argv[2] if len(sys.argv) > 2 else None)
    else:
        for l in (sys.argv[1:] or [""]):
            subprocess.run([sys.executable, __file__, "--one"] + ([l] if l else []), check=False)
