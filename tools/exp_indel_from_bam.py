"""The indel product path (indelCaller.indel_run: device pipeline + K9 + native rules) from a BAM FILE, with the contig's pack and per-read
sections made on the device (device_bam.py) or on host threads (experiment driver, GPU): python tools/exp_indel_from_bam.py [contig length] [depth]"""
import os
import queue
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bamio
from nanocaller_amd import device_bam, indelCaller
from nanocaller_amd import generate_indel_pileups as gip
from nanocaller_amd import generate_SNP_pileups as gsp

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 30
t0 = time.perf_counter()
w = bamio.make_pass2_world(seed=11, length=L, depth=depth)
tmp = tempfile.mkdtemp(prefix="nc_indel_bam_")
bam, fa = os.path.join(tmp, "p.bam"), os.path.join(tmp, "p.fa")
bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None), level=1)
bamio.write_fasta(fa, w.chrom, w.ref)
print("world + files: %.1f s, %d reads, BAM %.1f MB" % (time.perf_counter() - t0, w.n_reads, os.path.getsize(bam) / 1e6), flush=True)
texts = {}
for tag, env in (("device_ingest", "1"), ("host_ingest", "0")):
    os.environ["NC_DEVICE_INGEST"] = env
    best = None
    for rep in range(3):
        gsp.release_contig()
        gip._CONTIGS.clear()
        gip._DEV_INGEST.clear()
        device_bam.release()
        d = os.path.join(tmp, "%s%d" % (tag, rep))
        os.makedirs(d)
        params = dict(seq="ont", fasta_path=fa, win_size=40, small_win_size=4, mincov=4, maxcov=160, ins_t=0.4, del_t=0.6, supplementary=False, exclude_bed=None,
                      impute_indel_phase=False, indel_model="ONT-HG002", intermediate_indel_files_dir=d, prefix="t")
        jobs = queue.Queue()
        for s in range(1, w.length, 100_000):
            jobs.put(("indel", dict(chrom=w.chrom, start=s, end=min(w.length, s + 100_000), ploidy="diploid", sam_path=bam)))
        torch.cuda.synchronize()
        prof = os.environ.get("NC_EXP_PROFILE") and rep == 2
        if prof:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
        t0 = time.perf_counter()
        out = indelCaller.indel_run(params, {}, jobs, queue.Queue(), [], aligner="device")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if prof:
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
        texts[tag] = open(out).read()
        best = dt if best is None else min(best, dt)
    print("%s: %.1f ms, %d records" % (tag, best * 1e3, texts[tag].count("\n")), flush=True)
print("identical VCF text:", texts["device_ingest"] == texts["host_ingest"])
