#!/usr/bin/env python3
"""bench.py -- candidate sites/sec through pileup featurisation + CNN (BASELINE.json metric).

A "step" is one pass of the hot path over one batch.  The unit of work is a chr20-sized contig (64,444,167 bp, ONT 30x,
129 chunks of 500 kb -- BASELINE.json configs[1]) whose decoded alignments sit in PINNED HOST MEMORY (the
timed region starts there, as the reference feeds every chunk from the host): transfer (reference-difference wire form,
one PCIe copy per contig on its own stream, a ring of three device slots) -> expansion to the position-addressed codes
in HBM -> column scan -> neighbour selection -> (N,5,41,5) tensors -> coverage scale -> SNP CNN -> per-site results
(pos, probs[4], gt[2], dp, alt, fwd_dp[4], rev_dp[4], ref) back in host memory.

N = 1: a step = ONE contig (configs[1]); consecutive steps take different contigs (--distinct, default 3), so every step's
upload is a real one.  N > 1 (one process per GPU): the contig list of a whole job -- `--total-contigs` (default 8)
chr20-sized contigs, the shape of configs[3] "regions sharded across 8 GPUs" -- is sharded over the ranks in contiguous
blocks; a step = every rank passes once over ITS contigs; the total work per step is the same for every N (strong
scaling), no collective on the data path (regions shard embarrassingly).  `--weak` gives every rank one
contig of its own instead.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 3 --warmup 1
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CHR20_LEN = 64_444_167                 # GRCh38 chr20
SNP_FLOP_PER_SITE = 3_455_760          # SURVEY.md 8d / BASELINE.md section 3 (haploid model: 3,453,696)
TRUNK_FLOP_PER_SITE = 2 * (574_000 + 737_280 + 331_776)   # conv1 (3 kernels) + conv2 + conv3, SURVEY.md Appendix C.1
INDEL_FLOP_PER_SITE = 18_946_752       # SURVEY.md 8d (haploid 5,040,688)
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md
F16_MFMA_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md, dense
# k5_trunk_h3 issues 3 f16 MFMA products per fp32-equivalent product (hi*hi + hi*lo + lo*hi), so the peak its ALGORITHMIC
# FLOP can reach is the dense f16 MFMA peak / 3; the executed v_mfma_f32_16x16x32_f16 per site come from the library
HBM_PEAK_GBS = 8000.0
PCIE_PEAK_GBS = 63.0                   # MI355X_MICROARCH.md: PCIe Gen5 x16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--length", type=int, default=CHR20_LEN, help="contig length (default chr20)")
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--tech", default="ont", choices=["ont", "hifi"])
    ap.add_argument("--model", default="ONT-HG002")
    ap.add_argument("--ploidy", default="diploid", choices=["diploid", "haploid"])
    ap.add_argument("--distinct", type=int, default=3, help="N=1: number of distinct contigs the steps cycle through")
    ap.add_argument("--total-contigs", type=int, default=8, help="N>1: contigs of the whole job, sharded over the ranks")
    ap.add_argument("--weak", action="store_true", help="N>1: one contig per rank (weak scaling) instead of a sharded fixed list")
    ap.add_argument("--resident", action="store_true", help="headline from HBM-resident packs (round-1 definition; not SURVEY 8d's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_configs (HiFi 60x haploid, exact fp32, indel pipeline)")
    ap.add_argument("--no-overlap", action="store_true", help="collect every step's results before the next step is enqueued")
    ap.add_argument("--cpu-sample-chunks", type=int, default=0,
                    help="chunks of contig 0 the CPU baseline runs (one thread each); 0 = one per CPU this process may use (SURVEY 8d: all host cores; the cgroup quota counts)")
    ap.add_argument("--cnn-precision", default="default", choices=["default", "fp32", "fp16x3"],
                    help="trunk kernel: exact fp32 MFMA (k4_conv12) or fp16x3 split precision (k5_trunk_h3)")
    return ap.parse_args()


def usable_cpus():
    """CPUs this process can actually run on: the affinity mask capped by the cgroup CPU quota (a 256-thread box may grant a
    container 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(pack, info, chunks, params, model, gpu_result, n_sample):
    """Time the oracle (CPU port of the reference path, scalar C) on a bounded sample of the same workload,
    and check the GPU results of those chunks against it."""
    from concurrent.futures import ThreadPoolExecutor

    from nanocaller_amd.synth_device import host_sample_for_oracle
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle

    sample = chunks[:n_sample]
    lo = max(1, sample[0]["start"] - 50_000)
    hi = sample[-1]["end"] + 50_000
    h = host_sample_for_oracle(pack, info, lo, hi)
    rr = oracle.RawReads("chr20", h["L"], h["start"], h["end"], h["off"], h["codes"], h["strand"], h["keep"])
    hap = sample[0]["ploidy"] == "haploid"
    path, cov = get_SNP_model("haploid" if hap else model)
    if hap:
        cov = 30.0                                              # hap_train_coverage, snpCaller.py:73
    w = Weights(path)
    oracle.lib()
    cores = min(len(sample), usable_cpus())

    def one(c):
        pos, ref, mat, dp, freq, depth, fwd, rev = oracle.get_snp_testing_candidates(rr, params, c, rc=h["ref_codes"])
        rc = np.argmax(ref, 1).astype(np.int32)
        if hap:
            probs = oracle.snp_hap_forward(w.flat, mat, rc, np.full(len(pos), cov / depth), precision="f32")
        else:
            probs, _ = oracle.snp_forward(w.flat, mat, rc, np.full(len(pos), cov / depth), precision="f32")
        return pos, probs, dp

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(one, sample))
    dt = time.perf_counter() - t0
    n = sum(len(r[0]) for r in res)
    # parity of the GPU run on the same chunks
    pos_ok, max_dp = True, 0.0
    for ci, (pos, probs, dp) in enumerate(res):
        sel = gpu_result["chunk"] == ci
        pos_ok &= bool(np.array_equal(gpu_result["pos"][sel], pos) and np.array_equal(gpu_result["dp"][sel], dp))
        if pos_ok and len(pos):
            max_dp = max(max_dp, float(np.abs(gpu_result["probs"][sel] - probs).max()))
    return dict(value=n / dt, unit="sites/s", cores=cores, kind="port",
                sample="%d chunks of 500 kb (%d sites, %.1f s): oracle/nc_oracle.c scan+tensors+CNN(f32), one thread per chunk; the process may use %d of the box's %d logical CPUs"
                % (len(sample), n, dt, usable_cpus(), os.cpu_count() or 0)), dict(positions_exact=pos_ok, max_abs_dprob=max_dp, sites_checked=n)


class Contig:
    """one chr20-sized unit of work: its transfer form in pinned host memory (+ optionally the expanded pack kept in HBM)"""

    def __init__(self, eng, L, depth, tech, seed, keep_pack):
        from nanocaller_amd.synth_device import make_device_workload, wire_from_device_workload
        t0 = time.perf_counter()
        pack, info = make_device_workload(eng, L, depth=depth, tech=tech, seed=seed)
        self.gen_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        self.wire = wire_from_device_workload(pack, info)
        self.wire_s = time.perf_counter() - t0
        self.entries = info["pileup_entries"]
        info.pop("ref_wire", None)
        self.info = info
        self.pack = pack if keep_pack else None
        del pack
        torch.cuda.empty_cache()


def run_units(eng, uploader, contigs, n_units, params, chunks, local, overlap=True, resident=False):
    """n_units passes of the hot path over contigs[i % len(contigs)], each enqueued behind the previous one's CNN, every
    upload enqueued one unit ahead on the upload stream.  -> (total sites, last result)"""
    from nanocaller_amd import snpCaller
    total, prev, r = 0, None, None
    nc = len(contigs)
    nxt = None if resident else uploader.submit(contigs[0].wire)
    for i in range(n_units):
        c = contigs[i % nc]
        if resident:
            dpk, t = c.pack, None
        else:
            t = nxt
            dpk = uploader.expand(t)
            # the next contig's copy goes out now: it waits (on the device) for its slot, i.e. for unit i-1 to finish, and
            # then runs under this unit's kernels
            nxt = uploader.submit(contigs[(i + 1) % nc].wire) if i + 1 < n_units else None
        cur = snpCaller.call_chunks(params, chunks, device=local, dpk=dpk, defer=overlap)
        if t is not None:
            uploader.release(t)
        if overlap:
            if prev is not None:
                r = prev.result()
                total += int(r["n"])
            prev = cur
        else:
            r = cur
            total += int(r["n"])
    if overlap and prev is not None:
        r = prev.result()
        total += int(r["n"])
    return total, r


def measure(eng, uploader, contigs, units, warm, params, chunks, local, barrier, overlap=True, resident=False):
    run_units(eng, uploader, contigs, warm, params, chunks, local, overlap, resident)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total, r = run_units(eng, uploader, contigs, units, params, chunks, local, overlap, resident)
    torch.cuda.synchronize()
    barrier()
    return total, time.perf_counter() - t0, r


def snp_params(model, tech):
    return dict(mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6], snp_model=model,
                seq="ont" if tech == "ont" else "pacbio", supplementary=False, exclude_bed=None,
                disable_coverage_normalization=False, sam_path=None)


def extra_snp_config(eng, uploader, local, L, depth, tech, model, ploidy, exact_fp32, steps, label):
    """one more SNP configuration, outside the headline's timed region: same pipeline, own workload"""
    from nanocaller_amd.utils import get_chunks
    eng.set_cnn_precision(exact_fp32=exact_fp32)
    try:
        c = Contig(eng, L, depth, tech, seed=4812, keep_pack=True)
        chunks = get_chunks([("chr20", 1, L, ploidy)], cpu=16)
        params = snp_params(model, tech)
        nobar = lambda: None                                                 # noqa: E731
        sites, dt, r = measure(eng, uploader, [c], steps, 2, params, chunks, local, nobar)
        sites_r, dt_r, _ = measure(eng, uploader, [c], steps, 1, params, chunks, local, nobar, resident=True)
        eng.enable_timing(True, trunk_only=True)
        run_units(eng, uploader, [c], steps, params, chunks, local, True, True)
        torch.cuda.synchronize()
        sums, _ = eng.timing_sums()
        eng.enable_timing(False)
        trunk_tf = TRUNK_FLOP_PER_SITE * (sites_r / steps) * steps / (sums[4] * 1e-3) / 1e12 if sums[4] > 0 else 0.0
        peak = FP32_MFMA_PEAK_TFLOPS if exact_fp32 else F16_MFMA_PEAK_TFLOPS / 3.0
        out = {"workload": label, "value": sites / dt, "unit": "sites/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
               "sites_per_step": sites // steps, "hbm_resident_sites_s": sites_r / dt_r, "wire_bytes_per_contig": c.wire.nbytes,
               "pileup_entries": c.entries,
               "roofline": {"bound": "mfma", "kernel": "k4_conv12 (exact fp32 MFMA 16x16x4)" if exact_fp32 else "k5_trunk_h3 (f16x3 split MFMA)",
                            "achieved": trunk_tf, "peak": peak, "unit": "TFLOP/s", "frac": trunk_tf / peak,
                            "avg_launch_ms": sums[4] / max(1.0, sums[5])}}
        del c
        torch.cuda.empty_cache()
        return out
    finally:
        eng.set_cnn_precision(exact_fp32=False)


def extra_indel_config(eng, local):
    """The indel path as candidate sites/s (configs[2]'s second half) through the PRODUCT functions: a synthetic 30x BAM with
    planted indels and HP/PS tags + its FASTA -> per 100 kb chunk: K7 window scan, pass 2 assembled natively from the
    decoded contig (nc_indel_pass2_sets), device star alignment + K8, Indel_model (K9), genotype rules -> VCF records
    (generate_indel_pileups.get_indel_testing_candidates + indelCaller.indel_vcf_lines, what indel_run chains).  BAM
    decoding is logged separately (ingest is outside the metric).  In-run parity: K9 against the f64 oracle."""
    import tempfile

    from nanocaller_amd import _lib, indelCaller
    from nanocaller_amd import generate_indel_pileups as gip
    from nanocaller_amd.generate_SNP_pileups import device_pack, release_contig
    from nanocaller_amd.weights import Weights, get_indel_model
    from oracle import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamio                                                   # BAM / FASTA writer (test tooling)
    Lw = 400_000
    t0 = time.perf_counter()
    w = bamio.make_pass2_world(seed=5, length=Lw, depth=30)
    tmp = tempfile.mkdtemp(prefix="nc_bench_indel_")
    bam, fa = os.path.join(tmp, "i.bam"), os.path.join(tmp, "i.fa")
    bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None))
    bamio.write_fasta(fa, w.chrom, w.ref)
    t_files = time.perf_counter() - t0
    params = dict(seq="ont", fasta_path=fa, win_size=40, small_win_size=4, mincov=4, maxcov=160, ins_t=0.4, del_t=0.6, supplementary=False,
                  exclude_bed=None, impute_indel_phase=False)
    chunks = [dict(chrom=w.chrom, start=s, end=min(Lw, s + 100_000), ploidy="diploid", sam_path=bam) for s in range(1, Lw, 100_000)]
    wgt = Weights(get_indel_model("ONT-HG002"))
    eng.load_weights(_lib.MODEL_INDEL, wgt)
    release_contig()
    t0 = time.perf_counter()
    gip.decoded_contig(bam, w.chrom, fa)                           # ingest: BAM decode (+ query bases) and upload, once per contig
    device_pack(bam, fa, w.chrom, False, None, local)
    torch.cuda.synchronize()
    t_ingest = time.perf_counter() - t0

    def run():
        # what indelCaller.indel_run does with the chunks of one contig: one featuriser call, one CNN call, rules per chunk
        tuples = gip.get_indel_testing_candidates_batch(params, chunks, device=local, device_x=True)    # tensors stay in HBM
        x_all = torch.cat([torch.cat([t[1], t[2], t[3]], dim=1) for t in tuples if len(t[0])]).contiguous()
        probs = eng.indel_forward(_lib.MODEL_INDEL, x_all).cpu().numpy()
        n, lines, o = 0, 0, 0
        for c, t in zip(chunks, tuples):
            k = len(t[0])
            if k:
                lines += len(indelCaller.indel_vcf_lines(c["chrom"], t[0], probs[o:o + k], t[4], t[5])[0])
            o += k
            n += k
        return n, lines, [x_all], [probs], tuples
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_sites, n_lines, xs, ps, tuples = run()
    torch.cuda.synchronize()
    t_run = time.perf_counter() - t0
    # call concordance (how SURVEY 8f judges the aligner that replaces MUSCLE): planted indels carried by >= 5 reads whose exact
    # length comes back in an allele called at an anchor up to 60 bp before them
    import collections
    ev_off, ev_pos, ev_len = w.meta["events"]
    cnt = collections.Counter(zip(ev_pos.tolist(), ev_len.tolist()))
    truth = [k for k, v in sorted(cnt.items()) if v >= 5 and 1_000 < k[0] < Lw - 1_000]
    apos = np.concatenate([np.asarray(t[0], np.int64) for t in tuples if len(t[0])])
    aall = [al for t in tuples for al in t[4]]
    exact = 0
    for (p_, ln) in truth:
        lo_i, hi_i = np.searchsorted(apos, p_ - 60), np.searchsorted(apos, p_, side="right")
        exact += any(R is not None and len(A) - len(R) == ln for k in range(lo_i, hi_i) for (R, A) in aall[k])
    x15 = torch.cat(xs)
    # K9 alone at a batch that fills the chip
    nb = 16384
    xb = x15[torch.arange(nb, device=x15.device) % n_sites].contiguous()
    eng.indel_forward(_lib.MODEL_INDEL, xb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.indel_forward(_lib.MODEL_INDEL, xb)
    torch.cuda.synchronize()
    t_k9 = (time.perf_counter() - t0) / 3
    m = min(n_sites, 256)
    ep = oracle.indel_forward(wgt.flat, torch.cat(xs)[:m].cpu().numpy(), precision="f64")
    k9_err = float(np.abs(np.concatenate(ps)[:m] - ep).max())
    k9_tf = INDEL_FLOP_PER_SITE * nb / t_k9 / 1e12
    release_contig()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return {"workload": "indel path on a %d kb synthetic ONT 30x BAM with planted indels and HP/PS tags, %d chunks of 100 kb: K7 window scan -> "
                        "native pass 2 -> device star alignment + K8 (3 read sets per anchor) -> Indel_model (K9) -> genotype rules; %d candidate "
                        "sites reached the CNN, %d VCF records" % (Lw // 1000, len(chunks), n_sites, n_lines),
            "value": n_sites / t_run, "unit": "candidate sites/s", "sites": n_sites, "ms_per_chunk": t_run / len(chunks) * 1e3,
            "ingest_ms_logged_not_timed": t_ingest * 1e3, "bam_writing_s": t_files,
            "roofline": {"bound": "mfma", "kernel": "K9 indel CNN (k9_conv12_h3 + k8_conv23_h3 + k3_fc1), %d sites per call" % nb,
                         "achieved": k9_tf, "peak": F16_MFMA_PEAK_TFLOPS / 3.0, "unit": "TFLOP/s", "frac": k9_tf / (F16_MFMA_PEAK_TFLOPS / 3.0),
                         "sites_per_s": nb / t_k9},
            "concordance": {"planted_indels_with_5_or_more_carriers": len(truth), "exact_length_recovered": exact,
                            "fraction": exact / max(1, len(truth)), "star_scoring_open_extend_match_mismatch": list(_lib.STAR_SCORING)},
            "parity": {"k9_max_abs_dprob_vs_f64_oracle": k9_err, "sites_checked": m,
                       "note": "the tuples of this path equal the reference's own on the golden worlds (tests/test_pass2_golden.py)"}}


def trunk_traffic_from_profiles():
    """HBM bytes per site of the dominant kernel from the committed PMC passes (profiles/trunk_traffic.json, written from the
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes by tools/pmc_summary.py; FETCH_SIZE doubled as MI355X_MICROARCH.md says)"""
    try:
        with open(os.path.join(ROOT, "profiles", "trunk_traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ        # launched by torch.distributed.run
    # test hooks: several ranks on ONE GPU (a 1-GPU box can exercise the N > 1 code path: NC_BENCH_ONE_GPU=1 puts every rank
    # on device 0 and rendezvous goes over gloo, since RCCL refuses two ranks on one device)
    one_gpu = os.environ.get("NC_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))   # "nccl" is RCCL on ROCm
    from nanocaller_amd import snpCaller
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.utils import get_chunks
    from nanocaller_amd.wire import WireUploader

    eng = get_engine(local)
    exact_fp32 = args.cnn_precision == "fp32"               # library default = fp16x3 split precision
    eng.set_cnn_precision(exact_fp32=exact_fp32)
    L = args.length
    # which contigs does this rank own?  contig k of the job is generated from seed 812 + k on whatever rank owns it
    if world == 1:
        mine = list(range(max(1, args.distinct)))
        scaling, job_contigs = "weak", len(mine)
    elif args.weak:
        mine, scaling, job_contigs = [rank], "weak", world
    else:
        T = max(args.total_contigs, world)
        from nanocaller_amd.shard import shard_range
        mine, scaling, job_contigs = list(shard_range(T, rank, world)), "strong", T
    t_setup = time.perf_counter()
    contigs = [Contig(eng, L, args.depth, args.tech, 812 + k, keep_pack=(i == 0 or args.resident)) for i, k in enumerate(mine)]
    t_setup = time.perf_counter() - t_setup
    chunks = get_chunks([("chr20", 1, L, args.ploidy)], cpu=16)      # 16 = the reference's documented example (--cpu 16)
    params = snp_params(args.model, args.tech)
    uploader = WireUploader(eng, slots=int(os.environ.get("NC_UPLOAD_SLOTS", "3")))
    uploader.timing = True

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()

    per_step = 1 if world == 1 else len(mine)                 # units of work (contigs) this rank passes over per step
    # setup: priming passes size the device / pinned-host buffer pools (untimed, not warmup steps)
    run_units(eng, uploader, contigs, 2, params, chunks, local, True, args.resident)
    run_units(eng, uploader, contigs, max(1, args.warmup * per_step), params, chunks, local, not args.no_overlap, args.resident)
    uploader.h2d_events.clear()
    # The timed region: K steps.  Unit i+1 is enqueued as soon as unit i's CNN is (snpCaller.caller does the same with
    # consecutive contig groups) and its upload one unit ahead on the upload stream; all results are collected (copies
    # complete) before the closing synchronize.
    eng.enable_timing(True, trunk_only=True)                      # live HIP events on the dominant kernel's launches only
    import gc
    gc.collect()
    gc.disable()                                                  # a collector pause is tens of ms: 2-3 steps
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_sites_rank, r = run_units(eng, uploader, contigs, args.steps * per_step, params, chunks, local, not args.no_overlap, args.resident)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    sums, _ = eng.timing_sums()                                   # HIP-event totals over the K steps of the timed region
    trunk_ms, trunk_launches = sums[4], sums[5]
    eng.enable_timing(False)
    h2d_gbs, h2d_ms, h2d_bytes = uploader.h2d_rate()
    from nanocaller_amd.shard import dist_max, dist_sum
    dt = dist_max(dt)                                   # MAX over ranks
    total_sites = dist_sum(n_sites_rank)                # whole-job aggregate over the K steps
    if rank == 0:
        n_units = args.steps * per_step
        c0 = contigs[0]
        pack = c0.pack
        # ---- the same units from HBM-resident packs (round 1's definition of the timed region), contig 0
        rs_sites, rs_dt, r0 = measure(eng, uploader, contigs[:1], n_units, 1, params, chunks, local, lambda: None, not args.no_overlap, True)
        n_sites = int(r0["n"])                          # sites of contig 0: the stage numbers below are per pass over it

        def step():
            return snpCaller.call_chunks(params, chunks, device=local, dpk=pack)
        # stage breakdown (scan / featurize / CNN stage): event pairs around each stage put barrier packets on the stream,
        # so they are taken in their own short loop after the timed region
        eng.enable_timing(True)
        for _ in range(min(3, args.steps)):
            step()
        s2, cnt = eng.timing_sums()
        eng.enable_timing(False)
        stage_ms = np.array([s2[0] / max(1, cnt[0]), s2[1] / max(1, cnt[1]), s2[2] / max(1, cnt[2])])
        # expansion kernel alone
        t = uploader.submit(c0.wire)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            uploader.expand(t)
        e1.record()
        torch.cuda.synchronize()
        uploader.release(t)
        expand_ms = e0.elapsed_time(e1) / 3
        # host genotype rules + VCF record text for one pass' results (native formatter), untimed above: the third number
        # SURVEY.md 8(d) asks for (kernels only / + H2D + D2H / end to end incl. host K6 + VCF text)
        tv = time.perf_counter()
        vcf = snpCaller.snp_vcf_text("chr20", r0["pos"], r0["ref"], r0["probs"], r0["dp"], r0["freq"], r0["fwd_dp"], r0["rev_dp"],
                                     haploid=(args.ploidy == "haploid"), as_array=True)
        vcf_ms = (time.perf_counter() - tv) * 1e3
        # the production loop (snpCaller.caller) formats contig i on a worker thread while the GPU runs contig i+1: measure
        # that pipeline with the uploads inside
        from concurrent.futures import ThreadPoolExecutor

        scratch = np.empty((400 + 5) * max(n_sites, 1) * 5 // 4 + 65536, np.uint8)

        def fmt(res):
            return len(snpCaller.snp_vcf_text("chr20", res["pos"], res["ref"], res["probs"], res["dp"], res["freq"], res["fwd_dp"],
                                              res["rev_dp"], haploid=(args.ploidy == "haploid"), as_array=True, out=scratch))
        with ThreadPoolExecutor(max_workers=1) as pool:
            torch.cuda.synchronize()
            tp = time.perf_counter()
            pend, prev, pipe_sites = None, None, 0
            nxt = uploader.submit(contigs[0].wire)
            for i in range(n_units + 1):
                cur = None
                if i < n_units:
                    tk = nxt
                    dpk_i = uploader.expand(tk)
                    nxt = uploader.submit(contigs[(i + 1) % len(contigs)].wire) if i + 1 < n_units else None
                    cur = snpCaller.call_chunks(params, chunks, device=local, dpk=dpk_i, defer=True)
                    uploader.release(tk)
                if prev is not None:
                    res = prev.result()
                    pipe_sites += int(res["n"])
                    if pend is not None:
                        pend.result()
                    pend = pool.submit(fmt, res)
                prev = cur
            pend.result()
            pipelined_dt = time.perf_counter() - tp
        ms_per_step = dt / args.steps * 1e3
        value = total_sites / dt
        sites_timed = n_sites_rank                                     # this rank's sites inside the timed region
        cnn_tflops = SNP_FLOP_PER_SITE * n_sites / (stage_ms[2] * 1e-3) / 1e12 if stage_ms[2] > 0 else 0.0
        # dominant kernel = fused conv1+conv2+conv3 trunk: algorithmic FLOP of its launches in the timed region / their
        # summed HIP-event durations == FLOP per launch / average launch duration
        n_launch = max(1.0, trunk_launches)
        trunk_tflops = TRUNK_FLOP_PER_SITE * sites_timed / (trunk_ms * 1e-3) / 1e12 if trunk_ms > 0 else 0.0
        scan_bytes = c0.entries + L                            # (d+1) B/column, SURVEY.md 8d
        feat_bytes = (5403 - 2050) * n_sites                  # SURVEY.md 8d's 5,403 B/site with the tensor written as int16 (2,050 B) instead of fp32
        tt = trunk_traffic_from_profiles()
        traffic = None
        if tt and tt.get("kernel") == ("k4_conv12" if exact_fp32 else "k5_trunk_h3"):
            traffic = tt["bytes_per_site"] * sites_timed / n_launch
        common = {"achieved": trunk_tflops, "unit": "TFLOP/s", "traffic": traffic,
                  "traffic_note": ("HBM bytes per launch = bytes per site from the committed PMC passes (%s) x sites per launch of this run"
                                   % tt.get("source", "profiles/trunk_traffic.json")) if traffic else "no PMC pass committed for this kernel build",
                  "launches_in_timed_region": n_launch, "avg_launch_ms": float(trunk_ms / n_launch),
                  "flop_per_launch": TRUNK_FLOP_PER_SITE * sites_timed / n_launch,
                  "cnn_stage_tflops": cnn_tflops, "cnn_stage_ms": float(stage_ms[2])}
        if exact_fp32:
            roofline = {"bound": "mfma", "kernel": "k4_conv12: fused conv1+conv2+conv3 of the SNP CNN, fp32 MFMA 16x16x4",
                        "peak": FP32_MFMA_PEAK_TFLOPS, "frac": trunk_tflops / FP32_MFMA_PEAK_TFLOPS, **common}
        else:
            peak = F16_MFMA_PEAK_TFLOPS / 3.0
            mfma_per_site = eng.trunk_mfma_per_site()
            exec_tflops = mfma_per_site * 16384.0 * sites_timed / (trunk_ms * 1e-3) / 1e12 if trunk_ms > 0 else 0.0
            roofline = {"bound": "mfma", "kernel": "k5_trunk_h3: fused conv1+conv2+conv3 of the SNP CNN, fp32-equivalent via 3 "
                        "f16 MFMA 16x16x32 products (hi*hi + hi*lo + lo*hi), fp32 accumulate",
                        "peak": peak, "peak_note": "dense f16 MFMA peak 2500 TF / 3 products per fp32-equivalent product",
                        "frac": trunk_tflops / peak, "executed_mfma_per_site": mfma_per_site, "executed_f16_mfma_tflops": exec_tflops,
                        "executed_frac_of_f16_peak": exec_tflops / F16_MFMA_PEAK_TFLOPS,
                        "vs_fp32_mfma_peak": trunk_tflops / FP32_MFMA_PEAK_TFLOPS, **common}
        wire_b = c0.wire.nbytes
        out = {
            "metric": "candidate sites/sec (pileup+CNN)", "value": value, "unit": "sites/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32" if exact_fp32 else "f32 (f16x3 split MFMA, f32 accumulate)", "data": "synthetic",
            "config": {"workload": "SNP-only pileup+CNN, synthetic HG002-like %s %gx %s, chr20-sized contigs (%d bp, %d chunks of 500 kb); %s"
                       % (args.tech.upper(), args.depth, args.ploidy, L, len(chunks),
                          ("1 contig per step, %d distinct contigs cycled" % len(contigs)) if world == 1 else
                          ("%d contigs per step sharded over %d GPUs in contiguous blocks (%d on rank 0)" % (job_contigs, world, len(mine))) if scaling == "strong"
                          else "1 contig per GPU per step"),
                       "timed_region": "HBM-resident packs (--resident)" if args.resident else
                       "pinned host memory -> H2D (reference-difference wire form, own stream, ring of three slots) -> expand -> scan -> tensors -> CNN -> results in pinned host memory",
                       "contigs_per_step": job_contigs if world > 1 else 1, "sites_per_contig": n_sites,
                       "pileup_entries_per_contig": c0.entries, "snp_weights": args.model,
                       "tensor_format": "int16 between featuriser and CNN (exact; fp32 with --cnn-precision fp32)", "generator": "synth_v1 seed 812+contig",
                       "setup_s": round(t_setup, 2), "data_gen_s": round(sum(c.gen_s for c in contigs), 2),
                       "host_wire_build_s": round(sum(c.wire_s for c in contigs), 2)},
            "roofline": roofline,
            "h2d": {"wire_bytes_per_contig": wire_b, "bytes_per_pileup_entry": wire_b / c0.entries,
                    "uncompressed_pack_bytes": c0.entries + L, "achieved_GBs": h2d_gbs, "pcie_peak_GBs": PCIE_PEAK_GBS,
                    "copy_ms_per_contig": h2d_ms / max(1, len(uploader.h2d_events)), "copies_timed": len(uploader.h2d_events),
                    "expand_ms": expand_ms, "expand_GBs_written": c0.wire.codes_len / (expand_ms * 1e-3) / 1e9,
                    "note": "copies run on their own stream under the previous contig's compute; expansion is on the compute stream"},
            "three_numbers": {"kernels_only_sites_s": n_sites / ((stage_ms[0] + stage_ms[1] + stage_ms[2] + expand_ms) * 1e-3),
                              "with_h2d_d2h_sites_s": value if not args.resident else None,
                              "hbm_resident_with_d2h_sites_s": rs_sites / rs_dt,
                              "end_to_end_incl_vcf_text_serial_sites_s": n_sites / ((dt / max(1, n_units) * 1e3 + vcf_ms) * 1e-3),
                              "end_to_end_incl_vcf_text_pipelined_sites_s": pipe_sites / pipelined_dt,
                              "vcf_text_ms": vcf_ms, "vcf_bytes": len(vcf),
                              "note": "rank 0, per GPU; with_h2d_d2h = the timed region (= value at N=1); hbm_resident = the same passes over a pack "
                                      "already in HBM (round 1's headline); pipelined = + VCF text of pass i formatted on a host thread while the "
                                      "GPU runs pass i+1 (snpCaller.caller), uploads included"},
            "stages": {"expand_ms": expand_ms, "scan_ms": float(stage_ms[0]), "scan_GBs": scan_bytes / (stage_ms[0] * 1e-3) / 1e9 if stage_ms[0] else 0,
                       "scan_frac_hbm": scan_bytes / (stage_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS if stage_ms[0] else 0,
                       "featurize_ms": float(stage_ms[1]),
                       "featurize_GBs": feat_bytes / (stage_ms[1] * 1e-3) / 1e9 if stage_ms[1] else 0,
                       "featurize_frac_hbm": feat_bytes / (stage_ms[1] * 1e-3) / 1e9 / HBM_PEAK_GBS if stage_ms[1] else 0,
                       "cnn_ms": float(stage_ms[2]), "trunk_ms_per_contig": float(trunk_ms / max(1, n_units))},
        }
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = args.cpu_sample_chunks or min(len(chunks), max(1, usable_cpus()))
            cb, parity = cpu_baseline(pack, c0.info, chunks, params, args.model, r0, n_cpu)
            out["cpu_baseline"] = cb
            out["parity"] = parity
        if world == 1 and not args.no_extra:
            # other configurations, outside the headline's timed region (BASELINE.json configs[4], the exact-fp32 trunk,
            # and the indel half of configs[2]); each with its own workload and roofline
            contigs.clear()
            del c0, pack
            torch.cuda.empty_cache()
            extra = {}
            try:
                extra["hifi60x_haploid"] = extra_snp_config(eng, uploader, local, L, 60.0, "hifi", "CCS-HG002", "haploid", False, 12,
                                                            "SNP-only pileup+CNN, HiFi 60x haploid model (--haploid_genome), pacbio neighbour buckets, chr20-sized contig")
                extra["exact_fp32_trunk"] = extra_snp_config(eng, uploader, local, L, args.depth, args.tech, args.model, args.ploidy, True, 8,
                                                             "headline workload with the exact fp32 MFMA trunk (k4_conv12) on float32 tensors")
                extra["indel_pipeline"] = extra_indel_config(eng, local)
            except Exception as e:                                  # an extra must never take the headline line down
                extra["error"] = "%s: %s" % (type(e).__name__, e)
            out["extra_configs"] = extra
        print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
