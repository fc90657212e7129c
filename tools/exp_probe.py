"""Which workgroups get onto a CU BESIDE the persistent SNP trunk (one 512-thread workgroup per CU, 229 VGPRs, 130 KB LDS)?
A probe kernel (tools/ubench/probe.hip: R live VGPRs, L bytes of LDS, T threads) is launched on a second stream right after a
trunk launch; every probe workgroup stamps the wall clock at its start.  Reported: how many of its workgroups started before the
trunk finished."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanocaller_amd import _lib
from nanocaller_amd.engine import get_engine
from nanocaller_amd.weights import Weights, get_SNP_model
P = C.CDLL(os.path.join(ROOT, "tools", "ubench", "libprobe.so"))
eng = get_engine(0)
eng.load_weights(_lib.MODEL_SNP, Weights(get_SNP_model("ONT-HG002")[0]))
n = 262144
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.rand((n, 5, 41, 5), device="cuda", generator=g) * 30).to(torch.int16)
rc = torch.randint(0, 4, (n,), device="cuda", dtype=torch.int32)
sc = torch.full((n,), 0.9, device="cuda", dtype=torch.float64)
eng.set_tensor_format(int16=True)
A = torch.cuda.Stream()
B = torch.cuda.Stream(priority=int(os.environ.get("PRIO", "0")))
NB = 2048
start = torch.zeros(NB, dtype=torch.int64, device="cuda"); stop = torch.zeros(NB, dtype=torch.int64, device="cuda"); sink = torch.zeros(4, device="cuda")
mark = torch.zeros(2, dtype=torch.int64, device="cuda")
def run(regs, lds, threads, spin_ns=20000):
    start.zero_(); stop.zero_(); torch.cuda.synchronize()
    with torch.cuda.stream(A):
        eng.use_torch_stream()
        P.probe_launch(C.c_void_p(A.cuda_stream), 8, 0, 1, 64, C.c_void_p(mark.data_ptr()), C.c_void_p(mark.data_ptr() + 8), C.c_void_p(sink.data_ptr()), 0)
        eng.snp_forward(_lib.MODEL_SNP, x, rc, sc)       # one trunk launch (3.4 ms) + fc1 + heads
        e = torch.cuda.Event(); e.record(A)
    time.sleep(0.0005)                                  # the trunk is running
    with torch.cuda.stream(B):
        P.probe_launch(C.c_void_p(B.cuda_stream), regs, lds, NB, threads, C.c_void_p(start.data_ptr()), C.c_void_p(stop.data_ptr()), C.c_void_p(sink.data_ptr()), spin_ns)
    torch.cuda.synchronize()
    t0 = int(mark[0].item()); st = (start.cpu().numpy() - t0) / 100.0      # us
    return st
base = None
for regs, lds, threads in [(8, 0, 256), (11, 0, 256), (17, 0, 256), (20, 0, 256), (23, 0, 256), (26, 0, 256), (30, 0, 256), (100, 0, 256),
                           (17, 16384, 256), (17, 28672, 256), (17, 32768, 256), (17, 0, 512), (17, 0, 1024), (11, 0, 1024), (8, 0, 1024), (17, 0, 64)]:
    st = run(regs, lds, threads)
    # the trunk of 262144 dense sites takes ~3.4 ms from ~0.05 ms on: workgroups that started before 3.0 ms ran beside it
    early = int((st < 3000).sum())
    print("probe R=%3d (see probe.s for the VGPR count) lds=%5d threads=%4d: %4d of %d workgroups started beside the trunk; first %.0f us, median %.0f us, last %.0f us"
          % (regs, lds, threads, early, NB, st.min(), np.median(st), st.max()), flush=True)
