"""Which alignments differ between the banded and the full-matrix star alignment, and why (debugging aid, GPU).
usage: python tools/exp_band_diff.py [length] [margin]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nanocaller_amd import generate_indel_pileups as gip
from nanocaller_amd.engine import get_engine
from nanocaller_amd.synth_device import make_indel_device_workload
from oracle import oracle

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
margin = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eng = get_engine(0)
pack, reads_c, info = make_indel_device_workload(eng, L, depth=30.0, seed=812)
chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=160)
EW, WS = 176, 160
out = {}
for tag, mode in (("full", 0), ("band", 1)):
    os.environ["NC_PIPE_DUMP"] = "/tmp/ncdump_" + tag
    assert eng.L.nc_indel_sites_band(eng.ctx, mode, margin) == 0
    r = gip.indel_sites_device(eng, pack, reads_c, L, chunks, fetch=True, **kw)
    torch.cuda.synchronize()
    d = {}
    for name, dt in (("trace", np.uint32), ("win", np.uint8), ("n1", np.int32), ("al_site", np.int32), ("site_pos", np.int32), ("site_n2", np.int32)):
        d[name] = np.fromfile("/tmp/ncdump_%s.%s" % (tag, name), dt)
    if mode:
        d["band_lo"] = np.fromfile("/tmp/ncdump_band.band_lo", np.int8)
    out[tag] = d
os.environ.pop("NC_PIPE_DUMP")
f, b = out["full"], out["band"]
A = len(f["n1"])
tf, tb = f["trace"].reshape(A, EW), b["trace"].reshape(A, EW)
n2 = f["site_n2"][f["al_site"]]
mask = np.arange(EW)[None, :] <= n2[:, None]
diff = np.nonzero(((tf != tb) & mask).any(1))[0]
print("%d alignments, %d differ" % (A, len(diff)))
ref_code = pack.ref_code.cpu().numpy()
lut = "AGTCN***#"


def decode(ent, n2_):
    """entries -> (read index per position or -1, insertion (start, len) per slot)"""
    pos = [(int(e) & 0x3ff) - 1 for e in ent[:n2_ + 1]]
    ins = [((int(e) >> 20) & 0x3ff, (int(e) >> 10) & 0x3ff) for e in ent[:n2_ + 1]]
    return pos, ins


def cigar_entries(cig, n1_, n2_):
    """the oracle's cigar (ops 7/8 diag, 1 ins, 2 del) -> the same entry form"""
    pos = [-1] * (n2_ + 1)
    ins = [(0, 0)] * (n2_ + 1)
    i = j = 0
    for op, cnt in cig:
        if op in (7, 8):
            for _ in range(cnt):
                pos[j] = i
                i += 1
                j += 1
        elif op == 2:
            j += cnt
        else:
            ins[j] = (i, cnt)
            i += cnt
    return pos, ins


for a in diff[:8]:
    site = f["al_site"][a]
    v, n2_, n1_ = int(f["site_pos"][site]), int(n2[a]), int(f["n1"][a])
    q = "".join(lut[c] for c in f["win"][a * WS:a * WS + n1_])
    rw = "".join(lut[c] for c in ref_code[v - pack.tile_pos0:v - pack.tile_pos0 + n2_])
    cig = oracle.nw_cigar_free_tail_ref(q, rw, 25, 1, 20, -10)
    po, io = cigar_entries(cig, n1_, n2_)
    pf, i_f = decode(tf[a], n2_)
    pb, ib = decode(tb[a], n2_)
    print("alignment %d site %d pos %d n1 %d n2 %d band_lo %d: full==oracle %s, band==oracle %s" % (a, site, v, n1_, n2_, int(b["band_lo"][a]), (pf, i_f[:n2_]) == (po[:n2_ + 1], io[:n2_]),
                                                                                                 (pb, ib[:n2_]) == (po[:n2_ + 1], io[:n2_])))
    print("  read", q)
    print("  ref ", rw)
    print("  cigar", cig)
    dd = [j for j in range(n2_ + 1) if tf[a][j] != tb[a][j]]
    print("  slots that differ:", dd[:20], "full", [(pf[j], i_f[j]) for j in dd[:6]], "band", [(pb[j], ib[j]) for j in dd[:6]])
