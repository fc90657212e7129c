"""Device inflate (k_huff + k_lz + k_crc32) on the ONT-like bench BAM (tools/ont_like_bam.py): kernel times by HIP events, every member compared
with zlib.  usage: python tools/exp_inflate_ont.py [contigs] [contig length]"""
import os
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import ont_like_bam
from nanocaller_amd import _lib
from nanocaller_amd.engine import get_engine

n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L = int(sys.argv[2]) if len(sys.argv) > 2 else 9_000_000
eng = get_engine(0)
tmp = tempfile.mkdtemp()
bam, refs, fasta, st = ont_like_bam.make_files(eng, tmp, n_contigs, L, depth=30.0, seed0=7000, level=1)
raw = np.fromfile(bam, np.uint8)
Lb = _lib.lib()
cap = raw.size // 1024 + 4096
import ctypes as C
coff, clen, isize = np.empty(cap, np.int64), np.empty(cap, np.int32), np.empty(cap, np.int32)
n = C.c_int64()
assert Lb.nc_bgzf_members(_lib.npp(raw), raw.size, cap, _lib.npp(coff), _lib.npp(clen), _lib.npp(isize), C.byref(n)) == 0
n = int(n.value)
coff, clen, isize = coff[:n], clen[:n], isize[:n]
ooff = np.zeros(n + 1, np.int64)
np.cumsum(isize, out=ooff[1:])
dev = eng.device
d_comp = torch.from_numpy(np.concatenate([raw, np.zeros(64, np.uint8)])).to(dev)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
d_coff, d_clen, d_isize, d_ooff = t(coff), t(clen), t(isize), t(ooff[:-1])
d_out = torch.zeros(int(ooff[-1]) + 64, dtype=torch.uint8, device=dev)
d_st = torch.zeros(n, dtype=torch.int32, device=dev)
d_tok = torch.empty(((n + 63) // 64) << 22, dtype=torch.int32, device=dev)
d_ntok = torch.zeros(n, dtype=torch.int32, device=dev)
eng.use_torch_stream()
args = (n, d_comp.data_ptr(), d_coff.data_ptr(), d_clen.data_ptr(), d_out.data_ptr(), d_ooff.data_ptr(), d_isize.data_ptr(), d_st.data_ptr(), d_tok.data_ptr(), d_ntok.data_ptr())
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    assert Lb.nc_inflate_device_phase(eng.ctx, 1, *args) == 0
    ev[1].record()
    assert Lb.nc_inflate_device_phase(eng.ctx, 2, *args) == 0
    ev[2].record()
    assert Lb.nc_bgzf_crc_device(eng.ctx, *args[:8]) == 0
    ev[3].record()
    torch.cuda.synchronize()
    h, l, c = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])
    print("rep %d: %d members, %.0f MB -> %.0f MB: k_huff %.2f ms, k_lz %.2f ms, k_crc32 %.2f ms = %.1f GB/s of compressed BAM (%.1f inflated)"
          % (rep, n, raw.size / 1e6, ooff[-1] / 1e6, h, l, c, raw.size / (h + l + c) / 1e6, ooff[-1] / (h + l + c) / 1e6), flush=True)
assert int(d_st.count_nonzero().item()) == 0, d_st.cpu().numpy()[:20]
out = d_out.cpu().numpy()
step = max(1, n // 400)
for k in range(0, n, step):                                            # every step-th member against zlib (all of them: NC_EXP_COMPARE_ALL=1)
    want = zlib.decompress(raw[coff[k]:coff[k] + clen[k]].tobytes(), -15)
    assert out[ooff[k]:ooff[k] + isize[k]].tobytes() == want, k
print("members equal to zlib: every %d-th of %d" % (step, n))
