#!/usr/bin/env python3
"""bench.py -- candidate sites/sec through pileup featurisation + CNN (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: ALL chunks of a chr20-sized contig (64,444,167 bp,
ONT 30x, 129 chunks of 500 kb -- BASELINE.json configs[1]) whose decoded alignments are already resident in
HBM: column scan -> neighbour selection -> (N,5,41,5) tensors -> coverage scale -> SNP CNN -> per-site
results (pos, probs[4], gt[2], dp, alt, fwd_dp[4], rev_dp[4], ref) back in host memory.
Multi-GPU: one process per GPU, each rank owns an independent region of that size (weak scaling, no
collective on the data path -- regions shard embarrassingly).

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
# HBM bytes per launch of the fused trunk kernel, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
# FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950: profiles/r01b_pmc.md (k4_conv12: r01_final_pmc.md, same bytes)
TRUNK_TRAFFIC_PER_SITE = (2 * 68.53e6 + 215.87e6) / 31231      # k5_trunk_h3, profiles/r01b_pmc.md (measured at 31,231 sites per launch; same per site at 62,462: r01d)
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md
F16_MFMA_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md, dense
# k5_trunk_h3 issues 3 f16 MFMA products per fp32-equivalent product (hi*hi + hi*lo + lo*hi), so the peak its ALGORITHMIC
# FLOP can reach is the dense f16 MFMA peak / 3; it executes 726 v_mfma_f32_16x16x32_f16 per site (zero-weight tap slots incl.)
H3_MFMA_PER_SITE = 13 * 24 + 10 * 27 + 8 * 18      # conv1 24 per 16-position tile, conv2 27 per (tile, half of the channels), conv3 18
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--length", type=int, default=CHR20_LEN, help="contig length per GPU (default chr20)")
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--tech", default="ont", choices=["ont", "hifi"])
    ap.add_argument("--model", default="ONT-HG002")
    ap.add_argument("--ploidy", default="diploid", choices=["diploid", "haploid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="collect every step's results before the next step is enqueued")
    ap.add_argument("--cpu-sample-chunks", type=int, default=16)
    ap.add_argument("--cnn-precision", default="default", choices=["default", "fp32", "fp16x3"],
                    help="trunk kernel: exact fp32 MFMA (k4_conv12) or fp16x3 split precision (k5_trunk_h3)")
    return ap.parse_args()


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
    cores = min(len(sample), os.cpu_count() or 1)

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
    pos_ok, max_dp, off = True, 0.0, 0
    for ci, (pos, probs, dp) in enumerate(res):
        sel = gpu_result["chunk"] == ci
        pos_ok &= bool(np.array_equal(gpu_result["pos"][sel], pos) and np.array_equal(gpu_result["dp"][sel], dp))
        if pos_ok and len(pos):
            max_dp = max(max_dp, float(np.abs(gpu_result["probs"][sel] - probs).max()))
    return dict(value=n / dt, unit="sites/s", cores=cores, kind="port",
                sample="%d chunks of 500 kb (%d sites, %.1f s): oracle/nc_oracle.c scan+tensors+CNN(f32), one thread per chunk"
                % (len(sample), n, dt)), dict(positions_exact=pos_ok, max_abs_dprob=max_dp, sites_checked=n)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ        # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))   # "nccl" is RCCL on ROCm
    from nanocaller_amd import snpCaller
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import make_device_workload
    from nanocaller_amd.utils import get_chunks

    eng = get_engine(local)
    exact_fp32 = args.cnn_precision == "fp32"               # library default = fp16x3 split precision
    eng.set_cnn_precision(exact_fp32=exact_fp32)
    L = args.length
    t_gen = time.perf_counter()
    pack, info = make_device_workload(eng, L, depth=args.depth, tech=args.tech, seed=812 + rank)
    t_gen = time.perf_counter() - t_gen
    chunks = get_chunks([("chr20", 1, L, args.ploidy)], cpu=16)      # 16 = the reference's documented example (--cpu 16)
    params = dict(mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6],
                  snp_model=args.model, seq="ont" if args.tech == "ont" else "pacbio", supplementary=False,
                  exclude_bed=None, disable_coverage_normalization=False, sam_path=None)

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()

    def step():
        return snpCaller.call_chunks(params, chunks, device=local, dpk=pack)

    def run_steps(n):
        """n steps, step i+1 enqueued behind step i's CNN unless --no-overlap; every result is collected before returning"""
        prev, r = None, None
        for _ in range(n):
            cur = snpCaller.call_chunks(params, chunks, device=local, dpk=pack, defer=not args.no_overlap)
            if prev is not None:
                r = prev.result() if not args.no_overlap else prev
            prev = cur
        if prev is not None:
            r = prev.result() if not args.no_overlap else prev
        return r

    run_steps(2)                            # setup: priming calls size the device / pinned-host buffer pools (untimed, not warmup steps)
    run_steps(args.warmup)
    # The timed region: K steps, each enqueued as soon as the previous one's CNN is (snpCaller.caller does the same with
    # consecutive contig groups): step i's results drain and the host turns around while the GPU already runs step i+1's
    # scan.  All K results are collected (copies complete) before the closing synchronize.
    eng.enable_timing(True, trunk_only=True)                      # live HIP events on the dominant kernel's launches only
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = run_steps(args.steps)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    sums, _ = eng.timing_sums()                                   # HIP-event totals over the K steps of the timed region
    trunk_ms, trunk_launches = sums[4], sums[5]
    eng.enable_timing(False)
    n_sites = int(r["n"])
    from nanocaller_amd.shard import dist_max, dist_sum
    dt = dist_max(dt)                       # MAX over ranks
    total_sites = dist_sum(n_sites)         # whole-job aggregate
    # stage breakdown (scan / featurize / CNN stage): event pairs around each stage put barrier packets on the stream, so
    # they are taken in their own short loop after the timed region
    eng.enable_timing(True)
    for _ in range(min(3, args.steps)):
        step()
    sums, cnt = eng.timing_sums()
    eng.enable_timing(False)
    stage_ms = np.array([sums[0] / max(1, cnt[0]), sums[1] / max(1, cnt[1]), sums[2] / max(1, cnt[2]),
                         trunk_ms / max(1, args.steps), trunk_launches / max(1, args.steps)])
    # host genotype rules + VCF record text for one step's results (native formatter), untimed above: it is the third
    # number SURVEY.md 8(d) asks for (kernels only / + D2H / end to end incl. host K6 + VCF text)
    tv = time.perf_counter()
    vcf = snpCaller.snp_vcf_text("chr20", r["pos"], r["ref"], r["probs"], r["dp"], r["freq"], r["fwd_dp"], r["rev_dp"],
                                 haploid=(args.ploidy == "haploid"), as_array=True)
    vcf_ms = (time.perf_counter() - tv) * 1e3
    # the same K steps strictly one after the other (results collected before the next step is enqueued), for comparison
    torch.cuda.synchronize()
    ts = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    sequential_ms = (time.perf_counter() - ts) * 1e3 / args.steps
    # the production loop (snpCaller.caller) formats contig i on a worker thread while the GPU runs contig i+1: measure that
    # pipeline on the same step repeated args.steps times
    from concurrent.futures import ThreadPoolExecutor

    scratch = np.empty((400 + 5) * max(n_sites, 1) * 5 // 4 + 4096, np.uint8)

    def fmt(res):
        return len(snpCaller.snp_vcf_text("chr20", res["pos"], res["ref"], res["probs"], res["dp"], res["freq"], res["fwd_dp"],
                                          res["rev_dp"], haploid=(args.ploidy == "haploid"), as_array=True, out=scratch))
    with ThreadPoolExecutor(max_workers=1) as pool:
        torch.cuda.synchronize()
        tp = time.perf_counter()
        pend = None
        prev = None
        for _ in range(args.steps + 1):
            cur = snpCaller.call_chunks(params, chunks, device=local, dpk=pack, defer=True) if _ < args.steps else None
            if prev is not None:
                res = prev.result()
                if pend is not None:
                    pend.result()
                pend = pool.submit(fmt, res)
            prev = cur
        pend.result()
        pipelined_ms = (time.perf_counter() - tp) * 1e3 / args.steps
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_sites * args.steps / dt
        cnn_tflops = SNP_FLOP_PER_SITE * n_sites / (stage_ms[2] * 1e-3) / 1e12 if stage_ms[2] > 0 else 0.0
        # dominant kernel = fused conv1+conv2+conv3 trunk (k4_conv12): algorithmic FLOP of its launches / their summed
        # HIP-event durations == FLOP per launch / average launch duration
        n_launch = max(1.0, stage_ms[4])
        trunk_tflops = TRUNK_FLOP_PER_SITE * n_sites / (stage_ms[3] * 1e-3) / 1e12 if stage_ms[3] > 0 else 0.0
        scan_bytes = info["pileup_entries"] + L               # (d+1) B/column, SURVEY.md 8d
        feat_bytes = (5403 - 2050) * n_sites                  # SURVEY.md 8d's 5,403 B/site with the tensor written as int16 (2,050 B) instead of fp32
        common = {"achieved": trunk_tflops, "unit": "TFLOP/s", "traffic": TRUNK_TRAFFIC_PER_SITE * n_sites / n_launch,
                  "traffic_note": "HBM bytes per launch from committed PMC passes (profiles/), not re-measured in this run",
                  "launches_per_step": n_launch, "avg_launch_ms": float(stage_ms[3] / n_launch),
                  "flop_per_launch": TRUNK_FLOP_PER_SITE * n_sites / n_launch,
                  "cnn_stage_tflops": cnn_tflops, "cnn_stage_ms": float(stage_ms[2])}
        if exact_fp32:
            roofline = {"bound": "mfma", "kernel": "k4_conv12: fused conv1+conv2+conv3 of the SNP CNN, fp32 MFMA 16x16x4",
                        "peak": FP32_MFMA_PEAK_TFLOPS, "frac": trunk_tflops / FP32_MFMA_PEAK_TFLOPS, **common}
        else:
            peak = F16_MFMA_PEAK_TFLOPS / 3.0
            exec_tflops = H3_MFMA_PER_SITE * 16384.0 * n_sites / (stage_ms[3] * 1e-3) / 1e12 if stage_ms[3] > 0 else 0.0
            roofline = {"bound": "mfma", "kernel": "k5_trunk_h3: fused conv1+conv2+conv3 of the SNP CNN, fp32-equivalent via 3 "
                        "f16 MFMA 16x16x32 products (hi*hi + hi*lo + lo*hi), fp32 accumulate",
                        "peak": peak, "peak_note": "dense f16 MFMA peak 2500 TF / 3 products per fp32-equivalent product",
                        "frac": trunk_tflops / peak, "executed_f16_mfma_tflops": exec_tflops,
                        "executed_frac_of_f16_peak": exec_tflops / F16_MFMA_PEAK_TFLOPS,
                        "vs_fp32_mfma_peak": trunk_tflops / FP32_MFMA_PEAK_TFLOPS, **common}
        out = {
            "metric": "candidate sites/sec (pileup+CNN)", "value": value, "unit": "sites/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if exact_fp32 else "f32 (f16x3 split MFMA, f32 accumulate)", "data": "synthetic",
            "config": {"workload": "SNP-only pileup+CNN, synthetic HG002-like %s %gx %s, chr20-sized contig (%d bp, %d chunks of 500 kb) per GPU"
                       % (args.tech.upper(), args.depth, args.ploidy, L, len(chunks)), "sites_per_gpu": n_sites,
                       "pileup_entries_per_gpu": info["pileup_entries"], "snp_weights": args.model, "tensor_format": "int16 between featuriser and CNN (exact; fp32 with --cnn-precision fp32)", "generator": "synth_v1 seed 812+rank",
                       "data_gen_s": round(t_gen, 2)},
            "roofline": roofline,
            "three_numbers": {"kernels_only_sites_s": n_sites / ((stage_ms[0] + stage_ms[1] + stage_ms[2]) * 1e-3),
                              "with_d2h_sites_s": n_sites / (ms_per_step * 1e-3),
                              "with_d2h_sequential_calls_sites_s": n_sites / (sequential_ms * 1e-3),
                              "end_to_end_incl_vcf_text_sites_s": n_sites / ((ms_per_step + vcf_ms) * 1e-3),
                              "end_to_end_incl_vcf_text_pipelined_sites_s": n_sites / (pipelined_ms * 1e-3),
                              "vcf_text_ms": vcf_ms, "vcf_bytes": len(vcf),
                              "note": "rank 0, per GPU; with_d2h = the timed region (step i+1 enqueued behind step i's CNN, all results collected "
                                      "inside the region); sequential_calls = results collected before the next step is enqueued; "
                                      "pipelined = + VCF text of step i formatted on a host thread while the GPU runs step i+1 (snpCaller.caller)"},
            "stages": {"scan_ms": float(stage_ms[0]), "scan_GBs": scan_bytes / (stage_ms[0] * 1e-3) / 1e9 if stage_ms[0] else 0,
                       "scan_frac_hbm": scan_bytes / (stage_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS if stage_ms[0] else 0,
                       "featurize_ms": float(stage_ms[1]),
                       "featurize_GBs": feat_bytes / (stage_ms[1] * 1e-3) / 1e9 if stage_ms[1] else 0,
                       "cnn_ms": float(stage_ms[2])},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, parity = cpu_baseline(pack, info, chunks, params, args.model, r, args.cpu_sample_chunks)
            out["cpu_baseline"] = cb
            out["parity"] = parity
        print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
