"""The bench's from-BAM block on a smaller file, then the device route's stages one by one (experiment driver, GPU):
python tools/exp_from_bam.py [contigs] [contig length]"""
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nanocaller_amd import device_bam
from nanocaller_amd.bam import read_fasta
from nanocaller_amd.engine import get_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L = int(sys.argv[2]) if len(sys.argv) > 2 else 9_000_000
keep = []
r = bench.extra_from_bam(get_engine(0), 0, n_contigs=n, L=L, keep=keep)
print(json.dumps(r, indent=1))
tmp, bam, fa, regions = keep
for rep in range(3):
    device_bam.release()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    db = device_bam.DeviceBam(bam, 0, contigs=[r_[0] for r_ in regions] if rep == 2 else None)    # (the third: as the caller opens it)
    t1 = time.perf_counter()
    db.load()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("open %.1f ms, load %.1f ms: %s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, {k: (round(v * 1e3, 1) if isinstance(v, float) else v) for k, v in device_bam.LAST_LOAD.items()}))
    for chrom, _, ln, _ in regions[:2]:
        t0 = time.perf_counter()
        ref = read_fasta(fa, chrom)
        t1 = time.perf_counter()
        prep = db.prepare(chrom, ref)
        t2 = time.perf_counter()
        dp = db.pack(prep)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dp = db.pack(prep, codes=dp.codes)
        e1.record()
        torch.cuda.synchronize()
        print("%s: read_fasta %.1f ms, prepare %.1f ms, pack %.1f ms (again, device time %.2f ms), %d reads, codes %.0f MB"
              % (chrom, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, e0.elapsed_time(e1), prep["n_kept"], prep["codes_len"] / 1e6))
if os.environ.get("NC_EXP_PROFILE"):
    import cProfile
    import pstats
    import queue
    from nanocaller_amd import generate_SNP_pileups as gsp
    from nanocaller_amd import snpCaller
    from nanocaller_amd.utils import get_chunks
    for route in ("1", "0"):
        os.environ["NC_DEVICE_INGEST"] = route
        gsp.release_contig()
        device_bam.release()
        d = os.path.join(tmp, "prof" + route)
        os.makedirs(d)
        params = dict(regions_list=regions, sam_path=bam, fasta_path=fa, mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1,
                      threshold=[0.4, 0.6], snp_model="ONT-HG002", cpu=16, prefix="t", sample="S", seq="ont", supplementary=False,
                      exclude_bed=None, suppress_progress=True, disable_coverage_normalization=False,
                      chunks_list=get_chunks(regions, 16), vcf_path=d, intermediate_snp_files_dir=d)
        q = queue.Queue()
        for c in params["chunks_list"]:
            q.put(c)
        pr = cProfile.Profile()
        trace = []

        gpu_ev = []

        def wrap(obj, name, tag):
            f = getattr(obj, name)

            def g(*a, **k):
                ta = time.perf_counter()
                if tag in ("call_chunks", "pack"):
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ea.record()
                r = f(*a, **k)
                if tag in ("call_chunks", "pack"):
                    eb.record()
                    gpu_ev.append((tag, ea, eb))
                trace.append((tag, (ta - t0) * 1e3, (time.perf_counter() - ta) * 1e3))
                return r
            setattr(obj, name, g)
            return f
        saved = [(snpCaller, "call_chunks", wrap(snpCaller, "call_chunks", "call_chunks")), (snpCaller, "_prepare_device", wrap(snpCaller, "_prepare_device", "prepare")),
                 (snpCaller, "_prepare_wire", wrap(snpCaller, "_prepare_wire", "prepare_wire")), (snpCaller, "snp_vcf_text", wrap(snpCaller, "snp_vcf_text", "vcf_text")),
                 (snpCaller.PendingCall, "result", wrap(snpCaller.PendingCall, "result", "result")),
                 (device_bam.DeviceBam, "pack", wrap(device_bam.DeviceBam, "pack", "pack")), (device_bam, "open_device_bam", wrap(device_bam, "open_device_bam", "open+load")),
                 (os, "fsync", wrap(os, "fsync", "fsync"))]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if os.environ.get("NC_EXP_PROFILE") == "2":
            pr.enable()
        snpCaller.caller(params, q, queue.Queue(), [], device=0)
        torch.cuda.synchronize()
        pr.disable()
        print("route %s: %.1f ms" % (route, (time.perf_counter() - t0) * 1e3), {k: (round(v * 1e3, 1) if isinstance(v, float) else v) for k, v in device_bam.LAST_LOAD.items()} if route == "1" else "")
        for o, n_, f in saved:
            setattr(o, n_, f)
        print("  GPU time between the events around:", ", ".join("%s %.2f ms" % (t, a.elapsed_time(b)) for t, a, b in gpu_ev))
        for tag, at, dur in sorted(trace, key=lambda x: x[1]):
            print("  %8.1f ms  +%7.1f  %s" % (at, dur, tag))
        if os.environ.get("NC_EXP_PROFILE") == "2":
            pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
shutil.rmtree(tmp, ignore_errors=True)
