"""Where does nc_indel_scan_batch's time go? (experiment)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import bamio
from nanocaller_amd import generate_indel_pileups as gip
from nanocaller_amd.generate_SNP_pileups import device_pack
from nanocaller_amd.engine import get_engine
Lw = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
eng = get_engine(0)
w = bamio.make_pass2_world(seed=5, length=Lw, depth=30)
tmp = tempfile.mkdtemp()
bam, fa = os.path.join(tmp, "i.bam"), os.path.join(tmp, "i.fa")
bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None))
bamio.write_fasta(fa, w.chrom, w.ref)
dp = device_pack(bam, fa, w.chrom, False, None, 0)
dp = dp[0] if isinstance(dp, tuple) else dp
chunks = [(s, min(Lw, s + 100_000 - 1)) for s in range(1, Lw, 100_000)]
kw = dict(mincov=4, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6)
for rep in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = eng.indel_scan_batch(dp, chunks, **kw)
    print("indel_scan_batch %d chunks: %.2f ms" % (len(chunks), (time.perf_counter() - t) * 1e3))
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = eng.indel_scan_batch(dp, chunks[:1], **kw)
    print("indel_scan_batch 1 chunk: %.2f ms" % ((time.perf_counter() - t) * 1e3))
print("n_reads in events", dp.events["n_reads"], "tile", dp.tile_size if hasattr(dp, "tile_size") else "?")
