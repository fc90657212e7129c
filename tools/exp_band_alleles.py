"""Banded vs full-matrix allele alignments (debugging aid, GPU). usage: python tools/exp_band_alleles.py [length]"""
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
eng = get_engine(0)
pack, reads_c, info = make_indel_device_workload(eng, L, depth=30.0, seed=812)
chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=160)
out = {}
for tag, mode in (("full", 0), ("band", 1)):
    os.environ["NC_PIPE_DUMP"] = "/tmp/ncdump_" + tag
    assert eng.L.nc_indel_sites_band(eng.ctx, mode, 0) == 0
    r = gip.indel_sites_device(eng, pack, reads_c, L, chunks, fetch=True, **kw)
    torch.cuda.synchronize()
    d = {}
    for name, dt in (("cns", np.uint8), ("ncns", np.int32), ("rlen", np.int32), ("alen", np.int32), ("site_pos", np.int32), ("site_n2", np.int32), ("ab_lo", np.int8), ("ab_counts", np.int32)):
        try:
            d[name] = np.fromfile("/tmp/ncdump_%s.%s" % (tag, name), dt)
        except FileNotFoundError:
            pass
    d["type"] = r["type"]
    out[tag] = d
os.environ.pop("NC_PIPE_DUMP")
f, b = out["full"], out["band"]
print("band counts", b.get("ab_counts"))
n = len(f["ncns"])
same_cns = (f["ncns"] == b["ncns"]) & (f["cns"].reshape(n, 1024) == b["cns"].reshape(n, 1024)).all(1)
print("sets with the same consensus in both runs:", int(same_cns.sum()))
diff = np.nonzero(((f["rlen"] != b["rlen"]) | (f["alen"] != b["alen"])) & same_cns)[0]
print("%d sets, %d differ" % (n, len(diff)))
ref_code = pack.ref_code.cpu().numpy()
lut = "AGTCN***#"
for a in diff[:8]:
    site = a // 3
    v, n2, n1 = int(f["site_pos"][site]), int(f["site_n2"][site]), int(f["ncns"][a])
    q = "".join(lut[c] for c in f["cns"][a * 1024:a * 1024 + n1])
    rw = "".join(lut[c] for c in ref_code[v - pack.tile_pos0:v - pack.tile_pos0 + n2])
    print("set %d site %d pos %d n1 %d n2 %d lo %d type %d: full (%d, %d) band (%d, %d)" % (a, site, v, n1, n2, int(b["ab_lo"][a]), int(f["type"][site]), f["rlen"][a], f["alen"][a], b["rlen"][a], b["alen"][a]))
    print("  cns", q)
    print("  ref", rw)
    print("  full cigar", oracle.nw_cigar_ref(q, rw, 9, 1, 20, -10))
