"""Two indel passes in flight on ONE GPU: two contexts (own streams, own workspaces), two host threads, the same HBM-resident contig.  Does the
latency-bound third of a pass (K7, windows, tracebacks) hide under the issue-bound rest of the other pass?  (experiment driver, GPU)
usage: python tools/exp_two_ctx.py [length] [passes per thread]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nanocaller_amd import _lib
from nanocaller_amd import generate_indel_pileups as gip
from nanocaller_amd.engine import Engine, get_engine
from nanocaller_amd.synth_device import make_indel_device_workload
from nanocaller_amd.weights import Weights, get_indel_model

L = int(sys.argv[1]) if len(sys.argv) > 1 else 64_444_167
P = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eng0 = get_engine(0)
pack, reads_c, info = make_indel_device_workload(eng0, L, depth=30.0, seed=812)
chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=160)
wgt = Weights(get_indel_model("ONT-HG002"))
engs = [eng0, Engine(0)]
streams = [torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)]
for e in engs:
    e.load_weights(_lib.MODEL_INDEL, wgt)
torch.cuda.synchronize()


def worker(k, n, out):
    with torch.cuda.stream(streams[k]):
        e = engs[k]
        e.use_torch_stream()
        ns = 0
        for _ in range(n):
            r = gip.indel_sites_device(e, pack, reads_c, L, chunks, fetch=False, **kw)
            probs = e.indel_forward(_lib.MODEL_INDEL, r["x"])
            r.update(gip.indel_sites_fetch(e, r["n"], r["sets"]))
            probs.cpu()
            ns += r["n"]
        streams[k].synchronize()
        out[k] = ns


for mode in ("one", "two", "one", "two"):
    out = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "one":
        worker(0, 2 * P, out)
    else:
        th = [threading.Thread(target=worker, args=(k, P, out)) for k in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s context(s): %d passes in %.1f ms = %.2f ms per pass, %.2f M sites/s" % (mode, 2 * P, dt * 1e3, dt * 1e3 / (2 * P), sum(out.values()) / dt / 1e6), flush=True)
