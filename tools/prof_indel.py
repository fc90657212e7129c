"""cProfile of the batched indel featuriser on a synthetic BAM (where does a contig's pass 1 + pass 2 time go?)"""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import bamio
from nanocaller_amd import generate_indel_pileups as gip
from nanocaller_amd.engine import get_engine

Lw = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
eng = get_engine(0)
w = bamio.make_pass2_world(seed=5, length=Lw, depth=30)
tmp = tempfile.mkdtemp()
bam, fa = os.path.join(tmp, "i.bam"), os.path.join(tmp, "i.fa")
bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None))
bamio.write_fasta(fa, w.chrom, w.ref)
params = dict(seq="ont", fasta_path=fa, win_size=40, small_win_size=4, mincov=4, maxcov=160, ins_t=0.4, del_t=0.6, supplementary=False,
              exclude_bed=None, impute_indel_phase=False)
chunks = [dict(chrom=w.chrom, start=s, end=min(Lw, s + 100_000), ploidy="diploid", sam_path=bam) for s in range(1, Lw, 100_000)]
gip.get_indel_testing_candidates_batch(params, chunks, device_x=True)
torch.cuda.synchronize()
if os.environ.get("NO_GC"):
    import gc
    gc.disable()
t = time.perf_counter()
r = gip.get_indel_testing_candidates_batch(params, chunks, device_x=True)
torch.cuda.synchronize()
print("batch: %.1f ms for %d sites" % ((time.perf_counter() - t) * 1e3, sum(len(x[0]) for x in r)))
pr = cProfile.Profile()
pr.enable()
gip.get_indel_testing_candidates_batch(params, chunks, device_x=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats(os.environ.get("SORT", "tottime")).print_stats(40)
