"""Device inflate on a BAM of the bench's from-BAM workload (experiment driver, GPU): members parsed on the host, payloads uploaded, k_inflate timed,
result compared with zlib.  usage: python tools/exp_inflate.py [contigs] [contig length]"""
import os
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bamio
from nanocaller_amd.engine import get_engine
from nanocaller_amd.synth_device import make_device_workload

n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000
eng = get_engine(0)
lut = np.frombuffer(b"AGTCNNNN", np.uint8)
recs, refs = [], []
qrng = np.random.default_rng(3)
for k in range(n_contigs):
    pack, info = make_device_workload(eng, L, depth=30.0, tech="ont", seed=7000 + k)
    codes = pack.codes.cpu().numpy()
    refs.append(("ctg%d" % (k + 1), L))
    s_, e_, base = info["read_start"], info["read_end"], info["read_base"]
    for r in range(info["n_reads"]):
        o = int(base[r]) + int(s_[r])
        n_ = int(e_[r] - s_[r])
        qual = None
        if os.environ.get("NC_EXP_QUAL"):                                 # ONT-like base qualities instead of the absent-quality 0xff run: one literal per base
            qual = qrng.normal(18, 7, n_).clip(1, 50).astype(np.uint8).tobytes()
        recs.append(dict(tid=k, name="r%d_%d" % (k, r), flag=16 if info["strand"][r] else 0, pos0=int(s_[r]) - 1, cigar=[("M", n_)],
                         seq=lut[codes[o:o + n_]].tobytes().decode(), tags={}, qual=qual))
    del pack
tmp = tempfile.mkdtemp()
bam = os.path.join(tmp, "b.bam")
bamio.write_bam(bam, refs[0][0], refs[0][1], recs, other_refs=refs[1:], level=int(os.environ.get("NC_EXP_LEVEL", "6")))
raw = np.fromfile(bam, np.uint8)
t0 = time.perf_counter()
coff, clen, isize, o = [], [], [], 0
rb = raw.tobytes()
while o < len(rb):
    bsize = int.from_bytes(rb[o + 16:o + 18], "little") + 1
    coff.append(o + 18)
    clen.append(bsize - 26)
    isize.append(int.from_bytes(rb[o + bsize - 4:o + bsize], "little"))
    o += bsize
print("%d members, %.1f MB compressed, %.1f MB inflated (host walk %.2f s)" % (len(coff), len(rb) / 1e6, sum(isize) / 1e6, time.perf_counter() - t0))
dev = eng.device
d_comp = torch.from_numpy(np.concatenate([raw, np.zeros(16, np.uint8)])).to(dev)
ooff = np.zeros(len(coff) + 1, np.int64)
np.cumsum(np.asarray(isize, np.int64), out=ooff[1:])
d_coff = torch.tensor(coff, dtype=torch.int64, device=dev)
d_clen = torch.tensor(clen, dtype=torch.int32, device=dev)
d_out = torch.zeros(int(ooff[-1]) + 16, dtype=torch.uint8, device=dev)
d_ooff = torch.from_numpy(ooff[:-1].copy()).to(dev)
d_isize = torch.tensor(isize, dtype=torch.int32, device=dev)
d_st = torch.zeros(len(coff), dtype=torch.int32, device=dev)
d_tok = torch.empty(((len(coff) + 63) // 64) << 22, dtype=torch.int32, device=dev)
d_ntok = torch.zeros(len(coff), dtype=torch.int32, device=dev)
eng.use_torch_stream()
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = eng.L.nc_inflate_device(eng.ctx, len(coff), d_comp.data_ptr(), d_coff.data_ptr(), d_clen.data_ptr(), d_out.data_ptr(), d_ooff.data_ptr(),
                                 d_isize.data_ptr(), d_st.data_ptr(), d_tok.data_ptr(), d_ntok.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("k_inflate: %.2f ms = %.1f GB/s compressed, %.1f GB/s inflated" % (ms, len(rb) / ms / 1e6, sum(isize) / ms / 1e6))
if os.environ.get("NC_INFLATE_DEBUG"):
    st = d_st.cpu().numpy()
    print("debug counter %s: mean %.1f per member, min %d, max %d" % (os.environ["NC_INFLATE_DEBUG"], st.mean(), st.min(), st.max()))
    sys.exit(0)
assert rc == 0 and not d_st.cpu().numpy().any()
t0 = time.perf_counter()
n_cmp = len(coff) if os.environ.get("NC_EXP_COMPARE_ALL") else 400
want = b"".join(zlib.decompress(rb[c:c + n], -15) for c, n in zip(coff[:n_cmp], clen[:n_cmp]))
t_z = time.perf_counter() - t0
got = d_out[:len(want)].cpu().numpy().tobytes()
print("first %d members equal zlib's output:" % n_cmp, got == want, "(zlib: %.0f MB/s inflated on one core)" % (len(want) / t_z / 1e6))
