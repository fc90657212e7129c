"""Wall time of the native / device calls inside the batched indel featuriser (experiment): wraps the ctypes entry points."""
import os, sys, tempfile, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bamio
from nanocaller_amd import _lib
from nanocaller_amd import generate_indel_pileups as gip
from nanocaller_amd.engine import get_engine

Lw = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
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
L = _lib.lib()
T = collections.OrderedDict()
for name in ("nc_indel_scan_batch", "nc_indel_pass2_sets", "nc_pass2_view", "nc_star_msa_tensor_dup", "nc_allele_prediction_device", "nc_pass2_free"):
    f = getattr(L, name)
    def mk(f=f, name=name):
        def g(*a):
            t = time.perf_counter(); r = f(*a); T[name] = T.get(name, 0.0) + time.perf_counter() - t; return r
        return g
    setattr(L, name, mk())
for rep in range(3):
    T.clear()
    torch.cuda.synchronize(); t = time.perf_counter()
    r = gip.get_indel_testing_candidates_batch(params, chunks, device_x=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("batch %.1f ms for %d sites: " % (dt * 1e3, sum(len(x[0]) for x in r)) + ", ".join("%s %.1f" % (k.replace("nc_", ""), v * 1e3) for k, v in T.items()) +
          ", python %.1f" % ((dt - sum(T.values())) * 1e3), flush=True)
