"""Scale check of the indel path (experiment): a multi-megabase synthetic BAM through indelCaller.indel_run's batched route --
more chunks than one native pass-2 group holds -- with timings per stage.  usage: python tools/exp_indel_scale.py [length]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bamio
from nanocaller_amd import _lib, indelCaller
from nanocaller_amd import generate_indel_pileups as gip
from nanocaller_amd.engine import get_engine
from nanocaller_amd.weights import Weights, get_indel_model

Lw = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
eng = get_engine(0)
t = time.perf_counter()
w = bamio.make_pass2_world(seed=5, length=Lw, depth=30)
tmp = tempfile.mkdtemp()
bam, fa = os.path.join(tmp, "i.bam"), os.path.join(tmp, "i.fa")
bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None))
bamio.write_fasta(fa, w.chrom, w.ref)
print("world + files: %.1f s, BAM %.1f MB" % (time.perf_counter() - t, os.path.getsize(bam) / 1e6), flush=True)
params = dict(seq="ont", fasta_path=fa, win_size=40, small_win_size=4, mincov=4, maxcov=160, ins_t=0.4, del_t=0.6, supplementary=False,
              exclude_bed=None, impute_indel_phase=False)
chunks = [dict(chrom=w.chrom, start=s, end=min(Lw, s + 100_000), ploidy="diploid", sam_path=bam) for s in range(1, Lw, 100_000)]
eng.load_weights(_lib.MODEL_INDEL, Weights(get_indel_model("ONT-HG002")))
t = time.perf_counter()
gip.decoded_contig(bam, w.chrom, fa)
print("decode: %.2f s" % (time.perf_counter() - t), flush=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tuples = gip.get_indel_testing_candidates_batch(params, chunks, device=0, device_x=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    x_all = torch.cat([torch.cat([t_[1], t_[2], t_[3]], dim=1) for t_ in tuples if len(t_[0])]).contiguous()
    probs = eng.indel_forward(_lib.MODEL_INDEL, x_all).cpu().numpy()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    n = lines = o = 0
    for c, t_ in zip(chunks, tuples):
        k = len(t_[0])
        if k:
            lines += len(indelCaller.indel_vcf_lines(c["chrom"], t_[0], probs[o:o + k], t_[4], t_[5])[0])
        o += k; n += k
    t3 = time.perf_counter()
    print("%d chunks, %d sites, %d records: featuriser %.1f ms, CNN %.1f ms, rules %.1f ms -> %.1f k sites/s; max HBM %.2f GB"
          % (len(chunks), n, lines, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, n / (t3 - t0) / 1e3, torch.cuda.max_memory_allocated() / 1e9), flush=True)
