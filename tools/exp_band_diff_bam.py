"""exp_band_diff.py on the BAM-derived test world of tests/test_indel_pipeline.py (debugging aid, GPU)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bamio
from nanocaller_amd import generate_indel_pileups as gip
from oracle import oracle

d = tempfile.mkdtemp()
w = bamio.make_pass2_world(seed=11, length=150_000, depth=26)
bam, fa = os.path.join(d, "p.bam"), os.path.join(d, "p.fa")
bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None))
bamio.write_fasta(fa, w.chrom, w.ref)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_indel_pipeline import _params
dct = _params(fa)
chunks = [dict(chrom=w.chrom, start=s, end=min(w.length, s + 50_000), sam_path=bam) for s in range(1, w.length, 50_000)]
EW, WS = 176, 160
out = {}
for tag, mode in (("full", "0"), ("band", "1")):
    os.environ["NC_PIPE_DUMP"] = os.path.join(d, tag)
    os.environ["NC_PIPE_BAND"] = mode
    gip.get_indel_testing_candidates_batch(dct, chunks)
    dd = {}
    for name, dt in (("trace", np.uint32), ("win", np.uint8), ("n1", np.int32), ("al_site", np.int32), ("site_pos", np.int32), ("site_n2", np.int32)):
        dd[name] = np.fromfile(os.path.join(d, tag) + "." + name, dt)
    if mode == "1":
        dd["band_lo"] = np.fromfile(os.path.join(d, tag) + ".band_lo", np.int8)
    out[tag] = dd
f, b = out["full"], out["band"]
A = len(f["n1"])
assert np.array_equal(f["n1"], b["n1"]) and np.array_equal(f["win"], b["win"])
tf, tb = f["trace"].reshape(A, EW), b["trace"].reshape(A, EW)
n2 = f["site_n2"][f["al_site"]]
mask = np.arange(EW)[None, :] <= n2[:, None]
diff = np.nonzero(((tf != tb) & mask).any(1))[0]
print("%d alignments, %d differ; n1 histogram of the differing ones: %s" % (A, len(diff), np.bincount(np.minimum(f["n1"][diff] // 20, 8))))
print("band_lo of the differing ones:", np.unique(b["band_lo"][diff], return_counts=True))
lut = "AGTCN***#"


def decode(ent, n2_):
    return [((int(e) & 0x3ff) - 1, (int(e) >> 20) & 0x3ff, (int(e) >> 10) & 0x3ff) for e in ent[:n2_ + 1]]


for a in diff[:6]:
    site = f["al_site"][a]
    v, n2_, n1_ = int(f["site_pos"][site]), int(n2[a]), int(f["n1"][a])
    q = "".join(lut[c] for c in f["win"][a * WS:a * WS + n1_])
    rw = w.ref[v - 1:v - 1 + n2_]
    lo = int(b["band_lo"][a])
    cig = oracle.nw_cigar_free_tail_ref(q, rw, 25, 1, 20, -10)
    cb32 = oracle.nw_cigar_band_free_tail_ref(q, rw, lo, 32, 25, 1, 20, -10)
    cb64 = oracle.nw_cigar_band_free_tail_ref(q, rw, lo, 64, 25, 1, 20, -10)
    print("alignment %d site %d pos %d n1 %d n2 %d band_lo %d" % (a, site, v, n1_, n2_, lo))
    print("  read", q)
    print("  ref ", rw)
    print("  full cigar", cig)
    print("  band32 cigar", cb32)
    print("  band64 cigar", cb64)
    dd = [j for j in range(n2_ + 1) if tf[a][j] != tb[a][j]]
    pf, pb = decode(tf[a], n2_), decode(tb[a], n2_)
    print("  slots that differ:", dd[:12], "full", [pf[j] for j in dd[:6]], "band", [pb[j] for j in dd[:6]])
