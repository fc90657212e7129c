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
CHR1_LEN = 248_956_422                 # GRCh38 chr1 (BASELINE.json configs[2])
SNP_FLOP_PER_SITE = 3_455_760          # SURVEY.md 8d / BASELINE.md section 3 (haploid model: 3,453,696)
TRUNK_FLOP_PER_SITE = 2 * (574_000 + 737_280 + 331_776)   # conv1 (3 kernels) + conv2 + conv3, SURVEY.md Appendix C.1
INDEL_FLOP_PER_SITE = 18_946_752       # SURVEY.md 8d (haploid 5,040,688)
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md
F16_MFMA_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md, dense
# k5_trunk_p3 issues 3 f16 MFMA products per fp32-equivalent product (hi*hi + hi*lo + lo*hi), so the peak its ALGORITHMIC
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
    ap.add_argument("--no-configs2", action="store_true", help="skip the configs[2] block (chr1-sized SNP + indel in one timed region)")
    ap.add_argument("--configs2-steps", type=int, default=12)
    ap.add_argument("--no-wgs", action="store_true", help="skip the whole-genome N=1 pass (24 contigs at GRCh38 lengths, SNP + indel halves)")
    ap.add_argument("--wgs", action="store_true", help="(default at N=1; kept for explicit invocations)")
    ap.add_argument("--wgs-passes", type=int, default=1)
    ap.add_argument("--wgs-scale", type=float, default=1.0, help="contig lengths = GRCh38 x this (1.0 = the genome)")
    ap.add_argument("--no-indel-leg", action="store_true", help="N>1: skip the indel passes over the sharded contig list")
    ap.add_argument("--indel-passes", type=int, default=3)
    ap.add_argument("--configs2-length", type=int, default=CHR1_LEN)
    ap.add_argument("--no-overlap", action="store_true", help="collect every step's results before the next step is enqueued")
    ap.add_argument("--repeat", type=int, default=5, help="timed regions of --steps steps each; value = the median, min / max reported")
    ap.add_argument("--cpu-sample-chunks", type=int, default=0,
                    help="chunks of contig 0 the CPU baseline runs; 0 = the whole contig (SURVEY 8d: one OS process per usable core pulling chunks from a queue)")
    ap.add_argument("--cnn-precision", default="default", choices=["default", "fp32", "fp16x3"],
                    help="trunk kernel: exact fp32 MFMA (k4_conv12) or fp16x3 split precision (k5_trunk_p3)")
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


def _cpu_worker(q_in, q_out, rr_args, ref_codes, params, wpath, hap, cov):
    """one OS process of the CPU baseline: private weights, chunks pulled from a queue, the CNN in batches of 1000
    (snpCaller.py:80,98-111,238-241)"""
    from nanocaller_amd.weights import Weights
    from oracle import oracle
    w = Weights(wpath)                                         # private copy per process, like the reference's workers
    rr = oracle.RawReads(*rr_args)
    oracle.lib()
    while True:
        item = q_in.get()
        if item is None:
            break
        ci, c = item
        pos, ref, mat, dp, freq, depth, fwd, rev = oracle.get_snp_testing_candidates(rr, params, c, rc=ref_codes)
        n = len(pos)
        probs = np.zeros((n, 4), np.float32)
        if n:
            rc = np.argmax(ref, 1).astype(np.int32)
            sc = np.full(n, cov / depth)
            for b in range(0, n, 1000):
                sl = slice(b, b + 1000)
                if hap:
                    probs[sl] = oracle.snp_hap_forward(w.flat, mat[sl], rc[sl], sc[sl], precision="f32")
                else:
                    probs[sl] = oracle.snp_forward(w.flat, mat[sl], rc[sl], sc[sl], precision="f32")[0]
        q_out.put((ci, np.asarray(pos), probs if ci < 16 else None, np.asarray(dp) if ci < 16 else None, n))


def cpu_baseline_processes(pack, info, chunks, params, model, gpu_result, n_sample):
    """SURVEY 8d's CPU baseline: the oracle (scalar C port of the reference path) driven the way the reference drives itself -- one OS
    process per usable core pulling chunks from a queue, private weights, CNN batches of 1000 -- over the chunks of a whole contig;
    the GPU results of the first 16 chunks are checked against it."""
    import multiprocessing as mp

    from nanocaller_amd.synth_device import host_sample_for_oracle
    from nanocaller_amd.weights import get_SNP_model
    sample = chunks[:n_sample]
    h = host_sample_for_oracle(pack, info, max(1, sample[0]["start"] - 50_000), sample[-1]["end"] + 50_000)
    hap = sample[0]["ploidy"] == "haploid"
    path, cov = get_SNP_model("haploid" if hap else model)
    if hap:
        cov = 30.0
    cores = max(1, min(len(sample), usable_cpus()))
    ctx = mp.get_context("fork")                               # the children run CPU code only (no HIP call after the fork)
    q_in, q_out = ctx.Queue(), ctx.Queue()
    rr_args = ("chr20", h["L"], h["start"], h["end"], h["off"], h["codes"], h["strand"], h["keep"])
    procs = [ctx.Process(target=_cpu_worker, args=(q_in, q_out, rr_args, h["ref_codes"], params, path, hap, cov), daemon=True) for _ in range(cores)]
    t0 = time.perf_counter()
    for p_ in procs:
        p_.start()
    for ci, c in enumerate(sample):
        q_in.put((ci, c))
    for _ in procs:
        q_in.put(None)
    res = [q_out.get() for _ in sample]
    dt = time.perf_counter() - t0
    for p_ in procs:
        p_.join(timeout=10)
    n = sum(r[4] for r in res)
    pos_ok, max_dp, checked = True, 0.0, 0
    for ci, pos, probs, dp, _ in res:
        if probs is None:
            continue
        sel = gpu_result["chunk"] == ci
        pos_ok &= bool(np.array_equal(gpu_result["pos"][sel], pos) and np.array_equal(gpu_result["dp"][sel], dp))
        if pos_ok and len(pos):
            max_dp = max(max_dp, float(np.abs(gpu_result["probs"][sel] - probs).max()))
        checked += len(pos)
    return dict(value=n / dt, unit="sites/s", cores=cores, kind="port", per_core_sites_s=n / dt / cores,
                sample="%d chunks of 500 kb = %s (%d sites, %.1f s): oracle/nc_oracle.c scan + tensors + CNN (f32, batches of 1000), one OS process per "
                       "usable core pulling chunks from a queue, private weights per process (snpCaller.py:238-241); the process may use %d of the box's %d logical CPUs"
                       % (len(sample), "the whole contig" if len(sample) == len(chunks) else "part of the contig", n, dt, usable_cpus(), os.cpu_count() or 0)), \
        dict(positions_exact=pos_ok, max_abs_dprob=max_dp, sites_checked=checked)


def cpu_baseline(pack, info, chunks, params, model, gpu_result, n_sample):
    """The thread variant (round 2's): the oracle on a bounded sample, one thread per chunk, GPU results of those chunks checked against it."""
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

    def __init__(self, eng, L, depth, tech, seed, keep_pack, pool=None):
        """pool: the host builder of the transfer form runs there; finish() waits for it (many-contig set-up)"""
        from nanocaller_amd.synth_device import make_device_workload, wire_from_device_workload
        t0 = time.perf_counter()
        pack, info = make_device_workload(eng, L, depth=depth, tech=tech, seed=seed)
        self.gen_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        self.wire = wire_from_device_workload(pack, info, pool=pool)
        self.wire_s = time.perf_counter() - t0
        self.entries = info["pileup_entries"]
        info.pop("ref_wire", None)
        self.info = info
        self.pack = pack if keep_pack else None
        del pack
        torch.cuda.empty_cache()

    def finish(self):
        if hasattr(self.wire, "result"):
            self.wire = self.wire.result()
        return self


def run_units(eng, uploader, contigs, n_units, params, chunks, local, overlap=True, resident=False):
    """n_units passes of the hot path over contigs[i % len(contigs)], each enqueued behind the previous one's CNN, every
    upload enqueued one unit ahead on the upload stream.  -> (total sites, last result)"""
    from nanocaller_amd import snpCaller
    total, r = 0, None
    pend, depth = [], (2 if os.environ.get("NC_PIPE_CNN", "0") == "1" else 1)
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
            # results are collected `depth` units behind: a pipelined call_chunks enqueues unit i's CNN inside unit i + 1's scan, so waiting for
            # unit i - 1's results right here would hold back the call that enqueues the device's next work
            pend.append(cur)
            if len(pend) > depth:
                r = pend.pop(0).result()
                total += int(r["n"])
        else:
            r = cur
            total += int(r["n"])
    while pend:
        r = pend.pop(0).result()
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


def _lib_kind(ploidy):
    from nanocaller_amd import _lib
    return _lib.MODEL_SNP if ploidy == "diploid" else _lib.MODEL_SNP_HAP


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
        # (8 untimed steps: the leg starts behind its own data generation, an idle device -- the first ~10 forwards after one run 15-25 % slower,
        # profiles/README.md "clock ramp"; with 2 the leg read 39-40 M sites/s on some boxes and 46-47 M on others)
        sites, dt, r = measure(eng, uploader, [c], steps, 8, params, chunks, local, nobar)
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
               "roofline": {"bound": "mfma", "kernel": "k4_conv12 (exact fp32 MFMA 16x16x4)" if exact_fp32 else "k5_trunk_lin (f16 split MFMA, int16 tensors)",
                            "achieved": trunk_tf, "peak": peak, "unit": "TFLOP/s", "frac": trunk_tf / peak,
                            "avg_launch_ms": sums[4] / max(1.0, sums[5])}}
        del c
        torch.cuda.empty_cache()
        return out
    finally:
        eng.set_cnn_precision(exact_fp32=False)


# Vector-issue roofline of the banded alignment fill (k_fill_band).  MI355X_MICROARCH.md: a SIMD issues a wave's plain 32-bit VALU instruction over
# 2 cycles (SIMD-32); packed 16-bit arithmetic (v_pk_*_i16) issues at half that rate (tools/ubench/valu_rate.hip: 1.83 ns against 1.1 ns).  The DP
# recurrence of one cell PAIR (two alignments in the halves of a register, 64 lanes) is 15 packed + 9 plain instructions (7 v_pk_sub, 4 v_pk_max,
# v_pk_min_u16, v_pk_mad, v_pk_add, v_pk_sub; xor, and, 3 shifts, 3 and-or ... the traceback bits) = 78 issue cycles per 128 cells and SIMD.
# peak = 1024 SIMDs x 2.4 GHz x 128 / 78 cells/s; the neighbour shifts (DPP), base streams and stores the kernel also issues count against it.
VALU_PLAIN_CYC, VALU_PK_CYC = 2, 4
FILL_INSTR_PK, FILL_INSTR_PLAIN = 15, 9
CLOCK_HZ = 2.4e9
FILL_PEAK_CELLS_S = 256 * 4 * CLOCK_HZ * 128 / (FILL_INSTR_PK * VALU_PK_CYC + FILL_INSTR_PLAIN * VALU_PLAIN_CYC)


def _indel_wire(eng, pack, reads_c, info, pool=None):
    """the synthetic indel contig as the host would hold it after decoding a BAM: ONE page-locked buffer with the
    reference-difference wire form of the codes, the tile index, the indel events and the bases without a reference column.
    pool: the arrays are fetched here, the host builder runs on the pool -> a Future"""
    from nanocaller_amd.wire import build_wire
    L = info["L"]
    codes_h = pack.codes.cpu().numpy()
    ref_true = info["tensors"]["ref"].cpu().numpy()[1:L + 1]
    ref_wire = ref_true | ((pack.ref_code[1:L + 1].cpu().numpy() == 4).astype(np.uint8) << 3)
    slot = pack.reads["slot_off"].cpu().numpy()
    off = slot[:-1] + (info["read_start"].astype(np.int64) & 15)                # codes[off + p - start]
    t = info["tensors"]
    ev = pack.events
    n = info["n_reads"]
    h = lambda x: x.cpu().numpy()                                                # noqa: E731
    n_ev = info["n_events"]
    extra = dict(ins_off=h(t["ins_off"])[:n_ev + 1], ins_bases=h(t["ins_bases"])[:max(info["n_ins_bases"], 1)], tail_off=h(t["tail_off"]),
                 tail_bases=h(t["tail_bases"]), read_ps=h(t["read_ps"]), read_flag=h(t["read_flag"]))
    events = (h(ev["ev_off"]), h(ev["ev_pos"])[:n_ev], h(ev["ev_len"])[:n_ev])
    rs, re_, ts, strand, hap = info["read_start"], info["read_end"], pack.tile_size, info["strand"], info["hap"]

    def build():
        return build_wire(rs, re_, off, codes_h, None, ref_wire, tile_size=ts, pos_lo=1, pos_hi=L, keep=np.ones(n, np.uint8), strand=strand, hap=hap,
                          events=events, indel_extra=extra)
    return pool.submit(build) if pool is not None else build()


class IndelJob:
    """The indel half of the path over one synthetic contig: an ONT 30x contig with planted indels and HP / PS tags (SURVEY 8d's generator,
    nc_synth_indel_*), 100 kb chunks, its transfer form (reference-difference wire + indel events + bases without a reference column) in
    page-locked host memory, and one pass of the product path over it: expansion -> K7 window scan -> anchors + read sets -> query windows ->
    star alignment (banded) -> tensors + consensus (K8) -> allele_prediction -> Indel_model (K9) -> per-site arrays in host memory; genotype
    rules + VCF text natively on the host."""

    def __init__(self, eng, L, seed=4813, name=b"chr20", haploid=False, window_after=160, wire=True, pool=None):
        from nanocaller_amd import _lib
        from nanocaller_amd.synth_device import make_indel_device_workload
        from nanocaller_amd.weights import Weights, get_indel_model
        self.eng, self.L, self.name = eng, L, name
        t0 = time.perf_counter()
        self.pack, self.reads_c, self.info = make_indel_device_workload(eng, L, depth=30.0, seed=seed)
        self.t_gen = time.perf_counter() - t0
        self.haploid = bool(haploid)
        self.kind = _lib.MODEL_INDEL_HAP if haploid else _lib.MODEL_INDEL
        self.wgt = Weights(get_indel_model("haploid" if haploid else "ONT-HG002"))
        eng.load_weights(self.kind, self.wgt)
        self.chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
        self.kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=window_after, haploid=self.haploid)
        self.contig = np.frombuffer(b"AGTCN", np.uint8)[self.info["tensors"]["ref"].cpu().numpy()[1:]].tobytes()
        t0 = time.perf_counter()
        self.wire = _indel_wire(eng, self.pack, self.reads_c, self.info, pool=pool) if wire else None
        self.t_wire = time.perf_counter() - t0

    def finish(self):
        if hasattr(self.wire, "result"):
            self.wire = self.wire.result()
        return self

    def drop_pack(self):
        """keep only the host-side transfer form (the HBM-resident pack is for the resident / instrumented passes)"""
        self.pack = self.reads_c = None
        torch.cuda.empty_cache()

    def gpu_pass(self, dp, rc):
        from nanocaller_amd import _lib
        from nanocaller_amd import generate_indel_pileups as gip
        eng = self.eng
        dbg = os.environ.get("NC_BENCH_DEBUG") == "1"
        t0 = time.perf_counter()
        r = gip.indel_sites_device(eng, dp, rc, self.L, self.chunks, fetch=False, **self.kw)
        t1 = time.perf_counter()
        probs = eng.indel_forward(self.kind, r["x"])
        t2 = time.perf_counter()
        r.update(gip.indel_sites_fetch(eng, r["n"], r["sets"]))
        t3 = time.perf_counter()
        r["probs"] = probs.cpu().numpy()
        del r["x"]
        if dbg:
            print("  indel pass host ms: plan+run %.1f, K9 enqueue %.1f, fetch %.1f, probs %.1f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (time.perf_counter() - t3) * 1e3),
                  file=sys.stderr, flush=True)
        return r

    def rules(self, r):
        import ctypes as C

        from nanocaller_amd import _lib
        N = r["n"]
        buf = np.empty(N * 110 + 4 * int(np.maximum(r["ref_len"], 0).sum() + np.maximum(r["alt_len"], 0).sum()) + 4096, np.uint8)
        nb = C.c_int64()
        rc = self.eng.L.nc_indel_vcf_format(self.name, N, _lib.npp(np.ascontiguousarray(r["pos"])), _lib.npp(np.ascontiguousarray(r["chunk"])), len(self.chunks),
                                            _lib.npp(r["probs"]), r["sets"], _lib.npp(np.ascontiguousarray(r["ref_len"])),
                                            _lib.npp(np.ascontiguousarray(r["alt_len"])), _lib.npp(r["alt"]), _lib.npp(np.ascontiguousarray(r["phase"])),
                                            self.contig, self.L, 1 if self.haploid else 0, _lib.npp(buf), buf.size, C.byref(nb), None)
        assert rc == 0, rc
        return int((buf[:nb.value] == 10).sum())

    def from_host_pass(self, uploader, tk):
        from nanocaller_amd.wire import indel_reads_struct
        t0 = time.perf_counter()
        dp = uploader.expand(tk)
        t1 = time.perf_counter()
        r = self.gpu_pass(dp, indel_reads_struct(dp))
        uploader.release(tk)
        if os.environ.get("NC_BENCH_DEBUG") == "1":
            print("  indel pass host ms: expand enqueue %.1f, whole pass %.1f" % ((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
        return r


def indel_parity_sample(pack, info, contig, rt, probs, wgt, hi=40_000, max_sites=40):
    """The checker of the indel half (test infrastructure: bench.py's post-timing parity block and tests/test_wgs_slice.py): the sites of `rt` (the
    product's per-site arrays, tensors still on the device) below position `hi` against the oracle's restatement of pass 2 from SAM-like records
    (oracle.read_windows_ref: CIGAR expansion; every star alignment in pure Python on the band the read's CIGAR allows; msa() by the C oracle) and
    against generate_indel_pileups.allele_prediction; K9 against the float64 oracle.  -> (sites checked, tensors + phase exact, alleles exact,
    max |dp| of K9, K9 sites checked)"""
    from nanocaller_amd import generate_indel_pileups as gip
    from oracle import oracle
    S = rt["sets"]
    r1 = int(np.searchsorted(info["read_start"], hi + 400))
    s_, e_ = info["read_start"][:r1], info["read_end"][:r1]
    slot = pack.reads["slot_off"][:r1 + 1].cpu().numpy()
    codes = pack.codes[:int(slot[-1])].cpu().numpy()
    ev_off = pack.events["ev_off"][:r1 + 1].cpu().numpy()
    ev_pos = pack.events["ev_pos"][:int(ev_off[-1])].cpu().numpy()
    ev_len = pack.events["ev_len"][:int(ev_off[-1])].cpu().numpy()
    ins_off = info["tensors"]["ins_off"][:int(ev_off[-1]) + 1].cpu().numpy()
    ins = info["tensors"]["ins_bases"][:max(int(ins_off[-1]), 1)].cpu().numpy()
    recs = oracle.records_from_indel_pack(
        s_, e_, lambda q: codes[int(slot[q]) + (int(s_[q]) & 15):int(slot[q]) + (int(s_[q]) & 15) + int(e_[q] - s_[q])],
        lambda q: list(zip(ev_pos[ev_off[q]:ev_off[q + 1]].tolist(), ev_len[ev_off[q]:ev_off[q + 1]].tolist())),
        lambda q, k: ins[ins_off[int(ev_off[q]) + k]:ins_off[int(ev_off[q]) + k + 1]])
    masked = pack.ref_code[1:hi + 401].cpu().numpy() == 4
    ref_s = "".join(c.lower() if m_ else c for c, m_ in zip(contig[:hi + 400].decode(), masked))
    xh = rt["x"][:max_sites + 24].cpu().numpy()
    alt_all = np.frombuffer(b"AGTCN", np.uint8)[rt["alt"]].tobytes().decode()
    aoff = np.zeros(rt["n"] * S + 1, np.int64)
    np.cumsum(np.maximum(rt["alt_len"].reshape(-1), 0), out=aoff[1:])
    checked, x_exact, alleles_exact = 0, True, True
    for k in range(min(rt["n"], max_sites)):
        p_ = int(rt["pos"][k])
        if p_ > hi:
            break
        got = oracle.indel_site_ref(recs, info["hap"], info["ps"], ref_s, p_, 160, 4, 160, band=True)     # pure Python: no product aligner in the check
        if got is None:
            x_exact = False
            break
        xs, cns, win, phase = got
        x_exact &= bool(np.array_equal(xh[k].reshape(S, 5, 128, 2), xs)) and phase == int(rt["phase"][k])
        mr = 40 if rt["type"][k] == 0 else 10
        for t_ in range(S):
            exp = gip.allele_prediction(cns[t_], win, mr)
            rl, al = int(rt["ref_len"][k, t_]), int(rt["alt_len"][k, t_])
            alleles_exact &= ((None, None) if rl < 0 else (win[:rl], alt_all[aoff[k * S + t_]:aoff[k * S + t_] + al])) == exp
        checked += 1
    m = min(rt["n"], 256)
    ep = oracle.indel_forward(wgt.flat, rt["x"][:m].cpu().numpy(), precision="f64")
    k9_err = float(np.abs(probs[:m].cpu().numpy() - ep).max())
    return checked, bool(x_exact), bool(alleles_exact), k9_err, int(m)


def extra_indel_haploid_config(eng, L, reps=6):
    """The indel half of configs[4]'s shape: --haploid_genome (one read set per site, haploid_Indel_model) with the pacbio preset's 260-base
    windows, over the same chr20-sized synthetic contig, the pack resident in HBM (no upload in the loop: the transfer form is the diploid
    workload's)."""
    job = IndelJob(eng, L, haploid=True, window_after=260, wire=False)
    job.gpu_pass(job.pack, job.reads_c)
    ms, n_rec, r = [], 0, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = job.gpu_pass(job.pack, job.reads_c)
        n_rec = job.rules(r)
        ms.append((time.perf_counter() - t0) * 1e3)
    med = float(np.median(ms))
    st = np.zeros(6, np.int64)
    eng.L.nc_indel_sites_band_stats(eng.ctx, st.ctypes.data_as(__import__("ctypes").c_void_p))
    out = {"workload": "indel half, haploid model + 260-base windows (configs[4]'s shape) on the chr20-sized synthetic ONT-like 30x contig: %d candidate sites, "
                       "%d VCF records per pass" % (r["n"], n_rec),
           "value": r["n"] / (med * 1e-3), "unit": "candidate sites/s (pack resident in HBM; featuriser -> K9 -> fetch -> native rules, one pass after the other)",
           "ms_per_pass_median": med, "pass_ms": [round(x, 2) for x in ms], "sites": int(r["n"]),
           "star_alignments_last_pass": {"on_32_diagonals": int(st[0]), "on_64_diagonals": int(st[1]), "full_matrix": int(st[2]), "edge_touch_rerun": int(st[3])}}
    del job
    torch.cuda.empty_cache()
    return out


def extra_indel_config(eng, uploader, local, L, reps=40):
    """The indel half of configs[2] at chromosome scale, as candidate sites/s: a chr20-sized synthetic ONT 30x contig (IndelJob).  Timed
    region: decoded alignments + the bases without a reference column in PINNED HOST MEMORY -> one H2D copy (own stream, under the
    previous pass) -> the pass -> genotype rules + VCF text (native, on a host thread under the next pass).  Stage times are HIP events of a
    separate instrumented pass; in-run parity against the oracle's restatement on a sample."""
    from concurrent.futures import ThreadPoolExecutor

    from nanocaller_amd import _lib
    from nanocaller_amd import generate_indel_pileups as gip
    from oracle import oracle
    job = IndelJob(eng, L)
    pack, reads_c, info, wgt, chunks, kw, contig, wire = job.pack, job.reads_c, job.info, job.wgt, job.chunks, job.kw, job.contig, job.wire
    t_gen, t_wire = job.t_gen, job.t_wire
    gpu_pass, rules = job.gpu_pass, job.rules

    def from_host_pass(tk):
        return job.from_host_pass(uploader, tk)
    # ---- warm-up (sizes every workspace), then: (a) HBM-resident passes one by one, (b) from pinned host memory, pipelined
    gpu_pass(pack, reads_c)
    resident_ms, n_rec = [], 0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = gpu_pass(pack, reads_c)
        n_rec = rules(r)
        resident_ms.append((time.perf_counter() - t0) * 1e3)
    n_sites = r["n"]
    uploader.h2d_events.clear()
    for _ in range(len(uploader.slots)):                            # sizes EVERY upload slot of the ring (they held the SNP contigs' smaller wires: a slot
        from_host_pass(uploader.submit(wire))                       # that grows inside the timed loop costs an allocation + a device synchronisation, ~30 ms)
    uploader.h2d_events.clear()
    import gc
    gc.collect()
    gc.disable()                                                     # (a full collection of the earlier legs' objects inside the loop: +10 ms on a 15 ms pass, every sixth)
    try:
        with ThreadPoolExecutor(max_workers=1) as pool:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nxt, pend, pass_ms = uploader.submit(wire), None, []
            for i in range(reps):
                tk = nxt
                nxt = uploader.submit(wire) if i + 1 < reps else None    # the next pass's copy runs under this pass's kernels
                tp = time.perf_counter()
                rr = from_host_pass(tk)
                if pend is not None:
                    pend.result()
                pass_ms.append((time.perf_counter() - tp) * 1e3)          # (the first one waits for its own upload: nothing to hide it under)
                pend = pool.submit(rules, rr)                            # rules + text of pass i under pass i + 1
            pend.result()
            torch.cuda.synchronize()
            t_host = time.perf_counter() - t0
    finally:
        gc.enable()
    h2d_gbs, _, _ = uploader.h2d_rate()
    # ---- instrumented pass: per-stage HIP events
    eng.enable_timing(True)
    rt = gip.indel_sites_device(eng, pack, reads_c, L, chunks, fetch=False, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    probs = eng.indel_forward(_lib.MODEL_INDEL, rt["x"])
    e1.record()
    rt.update(gip.indel_sites_fetch(eng, rt["n"], rt["sets"]))
    torch.cuda.synchronize()
    k9_ms = e0.elapsed_time(e1)
    eng.enable_timing(False)
    ms = np.zeros(6, np.float32)
    cells = np.zeros(2, np.int64)
    eng.L.nc_indel_sites_stage_ms(eng.ctx, _lib.npp(ms), _lib.npp(cells))
    ms, cells = [float(v) for v in ms], [int(v) for v in cells]
    band = np.zeros(6, np.int64)
    eng.L.nc_indel_sites_band_stats(eng.ctx, _lib.npp(band))
    band = [int(v) for v in band]
    A = int(rt["n_alignments"])
    S = rt["sets"]
    gbs = lambda nbytes, t_ms: nbytes / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0       # noqa: E731
    k7_bytes = info["pileup_entries"] + L + 8 * info["n_events"]                     # (d+1) B/column + 8 B/event
    win_bytes = A * 2 * 160                                                          # every window read once, written once
    k8_bytes = 2 * A * 128 + n_sites * S * (128 + 5120)                              # R x 128 + 128 + 5,120 B per set (R summed: <= 2 x alignments)
    fill_cells_s = band[4] / (ms[2] * 1e-3) if ms[2] > 0 else 0.0
    k9_tf = INDEL_FLOP_PER_SITE * n_sites / (k9_ms * 1e-3) / 1e12
    stages = {
        "k7_scan_anchors_sets": {"ms": float(ms[0]), "bound": "hbm", "algorithmic_bytes": int(k7_bytes), "achieved_GBs": gbs(k7_bytes, ms[0]),
                                 "frac": gbs(k7_bytes, ms[0]) / HBM_PEAK_GBS, "note": "(d+1) B/column + 8 B/event; the launch set also selects the anchors and builds the read sets"},
        "query_windows": {"ms": float(ms[1]), "bound": "hbm", "algorithmic_bytes": int(win_bytes), "achieved_GBs": gbs(win_bytes, ms[1]),
                          "frac": gbs(win_bytes, ms[1]) / HBM_PEAK_GBS},
        "star_alignment_fill": {"ms": float(ms[2]), "bound": "valu issue", "kernel": "k_fill_band<1> + k_fill_band<2> (anti-diagonal sweep over 32 / 64 diagonals)",
                                "dp_cells": band[4], "dp_cells_full_matrices": int(cells[0]), "achieved_cells_s": fill_cells_s,
                                "peak_cells_s": FILL_PEAK_CELLS_S, "frac": fill_cells_s / FILL_PEAK_CELLS_S,
                                "alignments": {"band32": band[0], "band64": band[1], "full_matrix_by_width": band[2], "full_matrix_after_edge_touch": band[3]},
                                "peak_note": "1024 SIMDs x 2.4 GHz x 128 cells per (%d packed 16-bit instructions x %d cycles + %d plain x %d cycles): the vector-issue "
                                             "time of the DP recurrence alone (MI355X_MICROARCH.md: SIMD-32, 2 cycles per wave instruction; packed 16-bit at half rate)"
                                             % (FILL_INSTR_PK, VALU_PK_CYC, FILL_INSTR_PLAIN, VALU_PLAIN_CYC)},
        "star_alignment_traceback": {"ms": float(ms[3]), "bound": "latency (one dependent 4-bit code per step and alignment)",
                                     "note": "banded tracebacks + the full-matrix fill / traceback of the alignments that do not fit a band"},
        "k8_tensors_consensus": {"ms": float(ms[4]), "bound": "hbm", "algorithmic_bytes": int(k8_bytes), "achieved_GBs": gbs(k8_bytes, ms[4]),
                                 "frac": gbs(k8_bytes, ms[4]) / HBM_PEAK_GBS},
        "allele_prediction": {"ms": float(ms[5]), "bound": "valu issue", "dp_cells_upper": int(cells[1])},
        "k9_indel_cnn": {"ms": float(k9_ms), "bound": "mfma", "achieved": k9_tf, "peak": F16_MFMA_PEAK_TFLOPS / 3.0, "unit": "TFLOP/s",
                         "frac": k9_tf / (F16_MFMA_PEAK_TFLOPS / 3.0), "sites_per_s": n_sites / (k9_ms * 1e-3)},
    }
    # HBM bytes by counter (committed FETCH_SIZE / WRITE_SIZE passes of tools/bench_indel_pipe.py -> profiles/indel_traffic.json, bytes per
    # candidate site): `traffic` per pass of this run; the fill's counters cover both of its launches (star alignment + allele alignment)
    try:
        with open(os.path.join(ROOT, "profiles", "indel_traffic.json")) as f:
            it = json.load(f)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from build_tag import INDEL_SOURCES
        tag_ok, tag_note = traffic_build_check(it, INDEL_SOURCES)
        for name, st in stages.items():
            key = name
            if key in it["stages"] and not tag_ok:
                st["traffic"] = None
                st["traffic_note"] = tag_note
            elif key in it["stages"]:
                st["traffic"] = it["stages"][key]["bytes_per_site"] * n_sites
                st["traffic_note"] = "HBM bytes per pass = bytes per candidate site by counter (%s) x sites of this run" % it.get("source", "profiles/indel_traffic.json")
                if st.get("bound") == "hbm" and st["ms"] > 0:
                    st["traffic_GBs"] = st["traffic"] / (st["ms"] * 1e-3) / 1e9
    except (OSError, ValueError, KeyError):
        pass
    # ---- in-run parity on a sample: pass 2 restated from SAM-like records by the oracle (CIGAR expansion, host star alignment, msa() in C)
    checked, x_exact, alleles_exact, k9_err, m = indel_parity_sample(pack, info, contig, rt, probs, wgt)
    # ---- call concordance (how SURVEY 8f judges the aligner that replaces MUSCLE): planted indels whose exact length comes back in an
    # allele called at a site up to 60 bp before them
    truth = info["truth"].cpu().numpy()
    tp = np.nonzero((truth[0] != 0) | (truth[1] != 0))[0]
    tp = tp[(tp > 1000) & (tp < L - 1000)]
    apos, rl_, al_ = rt["pos"], rt["ref_len"], rt["alt_len"]
    exact = 0
    for p_ in tp.tolist():
        lens = {int(truth[0][p_]), int(truth[1][p_])} - {0}
        lo_i, hi_i = np.searchsorted(apos, p_ - 60), np.searchsorted(apos, p_, side="right")
        exact += any(rl_[k, t_] > 0 and int(al_[k, t_] - rl_[k, t_]) in lens for k in range(lo_i, hi_i) for t_ in range(S))
    rms = np.sort(np.asarray(resident_ms))
    out = {"workload": "indel half of configs[2]: synthetic ONT 30x contig of %d bp with planted indels (1-50 bp) and HP/PS tags, %d chunks of 100 kb; "
                       "%d reads, %d indel events, %d candidate sites reach the CNN (%d read windows aligned), %d VCF records per pass"
                       % (L, len(chunks), info["n_reads"], info["n_events"], n_sites, A, n_rec),
           "value": n_sites * reps / t_host, "unit": "candidate sites/s", "reps": reps, "pass_ms": [round(v, 2) for v in pass_ms], "ms_per_pass": t_host / reps * 1e3,
           "timed_region": "pinned host memory (wire form + indel events + bases without a reference column: %.0f MB, one copy per pass on the upload "
                           "stream) -> expansion -> K7 -> anchors / read sets -> windows -> star alignment -> K8 -> allele_prediction -> K9 -> host arrays -> "
                           "native rules + VCF text (host thread, under the next pass)" % (wire.nbytes / 1e6),
           "hbm_resident_serial": {"sites_s_median": n_sites / (float(np.median(rms)) * 1e-3), "ms_median": float(np.median(rms)), "ms_min": float(rms[0]),
                                   "ms_max": float(rms[-1]), "note": "pack already in HBM; featuriser -> K9 -> fetch -> rules one after the other, nothing overlapped"},
           "h2d": {"bytes_per_pass": wire.nbytes, "achieved_GBs": h2d_gbs},
           "stages": stages, "setup_s": {"generate": round(t_gen, 2), "host_wire_form": round(t_wire, 2)},
           "concordance": {"planted_indels": int(len(tp)), "exact_length_recovered": int(exact), "fraction": exact / max(1, len(tp)),
                           "star_scoring_open_extend_match_mismatch": list(_lib.STAR_SCORING)},
           "parity": {"sites_checked_against_oracle_restatement": checked, "tensors_and_phase_exact": bool(x_exact), "alleles_exact": bool(alleles_exact),
                      "k9_max_abs_dprob_vs_f64_oracle": k9_err, "k9_sites_checked": int(m),
                      "note": "pass 2 restated from SAM-like records (oracle.read_windows_ref: CIGAR expansion), every star alignment in pure Python on the band "
                              "the read's CIGAR allows, as the device runs it (oracle.star_cigars_banded_ref), msa() by the C oracle; the reference-executed "
                              "tuples: tests/test_pass2_golden.py"}}
    del pack, rt, wire
    torch.cuda.empty_cache()
    return out


class PairUnit:
    """one contig of a SNP + indel job: the SNP half's transfer form, the indel half's job (phased reads: its own transfer form), the SNP chunks"""

    def __init__(self, snp, job, chunks, name):
        self.snp, self.job, self.chunks, self.name = snp, job, chunks, name
        self.scratch = {}

    def snp_text(self, res):
        from nanocaller_amd import snpCaller
        n = max(int(res["n"]), 1)
        if self.scratch.get("n", 0) < n:
            self.scratch["buf"], self.scratch["n"] = np.empty((400 + 5) * n * 5 // 4 + 65536, np.uint8), n
        return len(snpCaller.snp_vcf_text(self.name, res["pos"], res["ref"], res["probs"], res["dp"], res["freq"], res["fwd_dp"], res["rev_dp"],
                                          haploid=False, as_array=True, out=self.scratch["buf"]))


STEP_STARTS = []                 # host time at which run_pairs began each step of its last call (steady-state step time = the median difference)


def steady_step_ms():
    d = np.diff(np.asarray(STEP_STARTS))
    return float(np.median(d[1:]) * 1e3) if d.size > 2 else None


def run_pairs(uploader, local, params, units, n_steps, snp_half=True, indel_half=True):
    """n_steps steps, step i over units[i % len(units)]: the SNP pass (upload -> expansion -> scan -> tensors -> SNP CNN) then the indel pass
    (upload -> expansion -> K7 ... K9) of that contig, the reference's order (NanoCaller:25-55).  The copies of step i + 1 (SNP wire, then indel
    wire) are both enqueued before the indel pass of step i starts, so they run under its kernels; the moment the indel pass of step i returns (its
    last results are on the host: the GPU has nothing queued) the SNP half of step i + 1 is enqueued, and only then are step i's results collected and
    handed to the host thread (genotype rules + VCF text of both halves, natively, under the next step): the host's turn-around work runs under the
    next SNP half's kernels instead of beside an idle GPU.  -> (SNP sites, indel sites, indel VCF records)"""
    from concurrent.futures import ThreadPoolExecutor

    from nanocaller_amd import snpCaller
    ns = ni = nrec = 0
    nu = len(units)
    dbg = os.environ.get("NC_BENCH_DEBUG") == "1"
    tdbg = time.perf_counter()
    STEP_STARTS.clear()

    def enqueue_snp(u, tk):
        dpk = uploader.expand(tk)
        # (the indel pass follows this SNP half on the device: its CNN goes out now, not inside the next SNP half's scan)
        c = snpCaller.call_chunks(params, u.chunks, device=local, dpk=dpk, defer=True, pipeline=False if indel_half else None)
        uploader.release(tk)
        return c
    with ThreadPoolExecutor(max_workers=1) as pool:
        pend = None
        cur = enqueue_snp(units[0], uploader.submit(units[0].snp.wire)) if (snp_half and n_steps > 0) else None
        tk_i = uploader.submit(units[0].job.wire) if (indel_half and n_steps > 0) else None
        # SNP alone: nothing runs between two SNP halves, so the copy of step i + 2 goes out before step i + 1 is enqueued (a copy submitted right
        # before its own expansion is waited for in full: tools/exp_pairs_trace.py)
        early = uploader.submit(units[1 % nu].snp.wire) if (snp_half and not indel_half and n_steps > 1) else None
        for i in range(n_steps):
            u = units[i % nu]
            un = units[(i + 1) % nu]
            STEP_STARTS.append(time.perf_counter())
            if dbg:
                mf, _mt = torch.cuda.mem_get_info()
                print("pair step %d %s (snp %s indel %s): +%.1f ms; device memory free %.1f GB, torch reserved %.1f GB (allocated %.1f)"
                      % (i, u.name, snp_half, indel_half, (time.perf_counter() - tdbg) * 1e3, mf / 1e9, torch.cuda.memory_reserved() / 1e9,
                         torch.cuda.memory_allocated() / 1e9), file=sys.stderr, flush=True)
            more = i + 1 < n_steps
            if indel_half:
                nxt_s = uploader.submit(un.snp.wire) if (snp_half and more) else None
            else:
                nxt_s, early = early, (uploader.submit(units[(i + 2) % nu].snp.wire) if (snp_half and i + 2 < n_steps) else None)
            nxt_i = uploader.submit(un.job.wire) if (indel_half and more) else None
            ri = None
            if indel_half:
                ri = u.job.from_host_pass(uploader, tk_i)           # (ends on host waits: the GPU is idle when it returns)
                ni += int(ri["n"])
            cur_next = enqueue_snp(un, nxt_s) if nxt_s is not None else None
            rs = cur.result() if cur is not None else None
            if rs is not None:
                ns += int(rs["n"])
            if pend is not None:
                nrec += pend.result()
                pend = None
            if rs is not None or ri is not None:
                pend = pool.submit(_pair_host_half, u, rs, u, ri)
            cur, tk_i = cur_next, nxt_i
        if pend is not None:
            nrec += pend.result()
    return ns, ni, nrec


def _pair_host_half(us, rs, ui, ri):
    if rs is not None:
        us.snp_text(rs)
    return ui.job.rules(ri) if ri is not None else 0


# GRCh38 primary assembly, chr1..chr22, chrX, chrY (bp): the contig list the reference's `-chrom`-less invocation walks (NanoCaller:25-55, utils.py:6-63)
GRCH38 = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259), ("chr6", 170805979),
          ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
          ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285),
          ("chr19", 58617616), ("chr20", 64444167), ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415)]


def build_pair_units(eng, parts, seed0, seed1):
    """PairUnits of the contigs `parts` = [(name, length), ...] (set-up, untimed): the generator runs on the GPU contig by contig; the host builders of
    the transfer forms (1.2 + 1.8 s per 64 Mb: single threads of numpy + the native builder) run on a pool beside it -- at most `ahead` contigs'
    decoded arrays wait in host memory.  -> (units, SNP wire bytes, indel wire bytes, bp)"""
    from concurrent.futures import ThreadPoolExecutor

    from nanocaller_amd.utils import get_chunks
    units, snp_bytes, indel_bytes, bp = [], 0, 0, 0
    workers = max(1, min(8, len(os.sched_getaffinity(0)) // 2))
    ahead = workers + 2
    with ThreadPoolExecutor(max_workers=workers) as pool:
        for k, (name, L) in enumerate(parts):
            if k >= ahead:
                units[k - ahead].snp.finish()
                units[k - ahead].job.finish()
            snp = Contig(eng, L, 30.0, "ont", seed=seed0 + k, keep_pack=False, pool=pool)
            job = IndelJob(eng, L, seed=seed1 + k, name=name.encode(), pool=pool)
            job.drop_pack()
            units.append(PairUnit(snp, job, get_chunks([(name, 1, L, "diploid")], cpu=16), name))
            bp += L
        for u in units:
            snp_bytes += u.snp.finish().wire.nbytes
            indel_bytes += u.job.finish().wire.nbytes
    return units, snp_bytes, indel_bytes, bp


def wgs_block(eng, uploader, local, model, passes=1, scale=1.0, contigs=None):
    """The metric's own configuration at N = 1: ONE pass over a whole genome -- 24 contigs at the GRCh38 lengths (3.09 Gb), per contig the SNP half
    then the indel half, every wire from PINNED HOST MEMORY, one timed region (the reference walks all regions with snpCaller, then indelCaller:
    NanoCaller:25-55, utils.py:6-83).  Contigs of unequal length: the upload ring and every workspace are sized by chr1 (priming steps, untimed),
    the tails of short contigs and the SNP -> indel hand-over are inside the region.  value = (SNP + indel candidate sites) / wall time."""
    from nanocaller_amd.utils import get_chunks
    t0 = time.perf_counter()
    params = snp_params(model, "ont")
    units, snp_bytes, indel_bytes, bp = build_pair_units(eng, [(name, max(200_000, int(L0 * scale))) for name, L0 in (contigs or GRCH38)], 2000, 6000)
    t_setup = time.perf_counter() - t0
    big = max(units, key=lambda u: u.job.wire.nbytes)
    run_pairs(uploader, local, params, [big], len(uploader.slots))      # sizes every upload slot, workspace and result pool by the largest contig (untimed)
    uploader.h2d_events.clear()
    import gc
    gc.collect()
    gc.disable()
    try:
        eng.enable_timing(True, trunk_only=True)
        s0, _ = eng.timing_sums()
        torch.cuda.synchronize()
        t = time.perf_counter()
        ns, ni, nrec = run_pairs(uploader, local, params, units, passes * len(units))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        s1, _ = eng.timing_sums()
    finally:
        eng.enable_timing(False)
        gc.enable()
    h2d_gbs, _, h2d_bytes = uploader.h2d_rate()
    trunk_ms = s1[4] - s0[4]
    trunk_tf = TRUNK_FLOP_PER_SITE * ns / (trunk_ms * 1e-3) / 1e12 if trunk_ms > 0 else 0.0
    out = {"workload": "whole genome at N=1: %d contigs at GRCh38 lengths x %g (%d bp), synthetic ONT 30x, SNP half + indel half per contig" % (len(units), scale, bp),
           "value": (ns + ni) / dt, "unit": "candidate sites/s (SNP + indel, one timed region over the genome)", "passes": passes,
           "s_per_pass": dt / passes, "snp_sites_per_pass": ns // passes, "indel_sites_per_pass": ni // passes, "sites_per_pass": (ns + ni) // passes,
           "indel_vcf_records_per_pass": nrec // passes, "contigs": len(units), "bp": bp,
           "wire_bytes_snp": snp_bytes, "wire_bytes_indel": indel_bytes, "h2d_achieved_GBs": h2d_gbs,
           "frac_snp_trunk_f16x3": trunk_tf / (F16_MFMA_PEAK_TFLOPS / 3.0), "setup_s": round(t_setup, 1),
           "timed_region": "per contig: SNP wire and indel wire from pinned host memory (copies under the other half's kernels) -> per-site results in host "
                           "memory -> native rules + VCF text on a host thread under the next contig; ring and workspaces sized by chr1 beforehand"}
    del units
    torch.cuda.empty_cache()
    return out


def wgs_sharded_block(eng, uploader, local, model, rank, world, barrier, scale=1.0, contigs=None):
    """BASELINE.json configs[3]: the whole genome (24 contigs at GRCh38 lengths x `scale`), SNP half + indel half, sharded over the ranks by
    shard.shard_plan -- contiguous, contig-aware blocks of the 500 kb chunk list, balanced by scanned columns (a uniform-depth synthetic genome; a BAM's
    depth weights come from its index) -- every rank building and passing over ITS block only, from pinned host memory, no collective on the data path
    (the reference: one worker pool over all chunks, files as the gather medium, snpCaller.py:213-245, 278-285).  A rank's part of a contig it shares with
    a neighbour is a synthetic contig of that part's length.  -> dict on every rank (rank 0 prints it): whole-job sites / MAX rank time, the ranks' times
    and their imbalance (max / mean), per-rank upload rates."""
    import torch.distributed as dist

    from nanocaller_amd.shard import dist_max, dist_sum, shard_plan
    from nanocaller_amd.utils import get_chunks
    spec = [(n, max(500_000, int(L0 * scale))) for n, L0 in (contigs or GRCH38)]
    all_chunks = get_chunks([(n, 1, L, "diploid") for n, L in spec], cpu=16)
    plan = shard_plan(all_chunks, world)
    mine = plan[rank]
    parts = {}
    for c in mine:                                                   # this rank's part of every contig it touches
        a, b = parts.get(c["chrom"], (c["start"], c["end"]))
        parts[c["chrom"]] = (min(a, c["start"]), max(b, c["end"]))
    t0 = time.perf_counter()
    params = snp_params(model, "ont")
    order = [n for n, _ in spec if n in parts]
    units, snp_bytes, indel_bytes, bp = build_pair_units(eng, [(name, parts[name][1] - parts[name][0] + 1) for name in order], 2000 + 100 * rank, 6000 + 100 * rank)
    wire_bytes = snp_bytes + indel_bytes
    t_setup = time.perf_counter() - t0
    ns = ni = nrec = 0
    dt = 0.0
    uploader.h2d_events.clear()
    if units:
        big = max(units, key=lambda u: u.job.wire.nbytes)
        run_pairs(uploader, local, params, [big], len(uploader.slots))  # sizes ring, workspaces and pools by this rank's largest part (untimed)
        uploader.h2d_events.clear()
    import gc
    gc.collect()
    gc.disable()
    try:
        barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        if units:
            ns, ni, nrec = run_pairs(uploader, local, params, units, len(units))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t                                 # this rank's own time; the job's time is the MAX (barrier on both sides)
        barrier()
    finally:
        gc.enable()
    h2d_gbs, _, _ = uploader.h2d_rate()
    dev = "cpu" if dist.get_backend() == "gloo" else torch.device("cuda", local)
    mine_t = torch.tensor([dt, float(ns + ni), float(bp), h2d_gbs, float(len(mine)), float(wire_bytes)], dtype=torch.float64, device=dev)
    allt = [torch.zeros_like(mine_t) for _ in range(world)]
    dist.all_gather(allt, mine_t)
    rows = [[float(v) for v in t_] for t_ in allt]
    times = [r_[0] for r_ in rows]
    del units
    torch.cuda.empty_cache()
    job_dt = dist_max(dt)
    total = dist_sum(ns + ni)
    return {"workload": "configs[3]: %d contigs at GRCh38 lengths x %g (%d bp) sharded over %d ranks by shard_plan, SNP + indel halves" % (len(spec), scale, sum(L for _, L in spec), world),
            "value": total / job_dt if job_dt > 0 else 0.0, "unit": "candidate sites/s (SNP + indel, whole job, MAX over ranks)", "s_per_pass": job_dt, "sites": total,
            "rank_s": [round(v, 4) for v in times], "rank_imbalance_max_over_mean": max(times) / (sum(times) / len(times)) if sum(times) > 0 else None,
            "rank_sites": [int(r_[1]) for r_ in rows], "rank_bp": [int(r_[2]) for r_ in rows], "rank_chunks": [int(r_[4]) for r_ in rows],
            "rank_h2d_GBs": [round(r_[3], 2) for r_ in rows], "rank_wire_bytes": [int(r_[5]) for r_ in rows],
            "host_pinned_read_GBs_all_ranks": sum(r_[5] for r_ in rows) / job_dt / 1e9 if job_dt > 0 else None, "setup_s_rank0": round(t_setup, 1),
            "note": "every rank: its block's contigs (parts) from pinned host memory through its own upload ring; barrier + synchronize on both sides; value = all "
                    "ranks' sites / the slowest rank's time"}


def configs2_block(eng, uploader, local, model, steps=3, L=CHR1_LEN):
    """BASELINE.json configs[2]: "SNP+indel full pipeline on 1 MI355X, HG002 ONT 30x chr1" as ONE timed region.  A step = the SNP pass over a
    chr1-sized contig (248,956,422 bp, 498 chunks of 500 kb; upload -> expansion -> scan -> tensors -> SNP CNN -> per-site results) followed by the
    indel pass over a chr1-sized contig with planted indels and HP / PS tags (2,490 chunks of 100 kb; upload -> expansion -> K7 -> read sets ->
    windows -> banded star alignment -> tensors -> allele_prediction -> indel CNN -> per-site results), both from PINNED HOST MEMORY -- the reference
    runs the two halves one after the other in one invocation (NanoCaller:25-55; the indel half reads the phased BAM: its own decode and upload).
    The copy of each half runs under the kernels of the other; genotype rules + VCF text of both halves run natively on a host thread under the
    next step.  value = (SNP candidate sites + indel candidate sites) / wall time."""
    from concurrent.futures import ThreadPoolExecutor

    from nanocaller_amd import _lib, snpCaller
    from nanocaller_amd import generate_indel_pileups as gip
    from nanocaller_amd.utils import get_chunks
    t0 = time.perf_counter()
    snp = Contig(eng, L, 30.0, "ont", seed=912, keep_pack=False)
    chunks = get_chunks([("chr1", 1, L, "diploid")], cpu=16)
    params = snp_params(model, "ont")
    job = IndelJob(eng, L, seed=4913, name=b"chr1")
    t_setup = time.perf_counter() - t0
    units = [PairUnit(snp, job, chunks, "chr1")]

    def run(n_steps, snp_half=True, indel_half=True):
        return run_pairs(uploader, local, params, units, n_steps, snp_half, indel_half)

    def timed(n_steps, **kw):
        import gc
        gc.collect()
        gc.disable()                                                    # (a collector pause is tens of ms)
        try:
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = run(n_steps, **kw)
            torch.cuda.synchronize()
            return r, time.perf_counter() - t
        finally:
            gc.enable()
    run(len(uploader.slots))                                            # sizes every upload slot, every workspace and the result pools
    uploader.h2d_events.clear()
    eng.enable_timing(True, trunk_only=True)
    s0, _ = eng.timing_sums()
    (ns, ni, nrec), dt = timed(steps)
    steady = steady_step_ms()
    s1, _ = eng.timing_sums()
    eng.enable_timing(False)
    h2d_gbs, _, h2d_bytes = uploader.h2d_rate()
    trunk_ms, trunk_n = s1[4] - s0[4], max(1.0, s1[5] - s0[5])
    trunk_tf = TRUNK_FLOP_PER_SITE * ns / (trunk_ms * 1e-3) / 1e12 if trunk_ms > 0 else 0.0
    (ns_a, _, _), dt_s = timed(steps, indel_half=False)
    (_, ni_a, _), dt_i = timed(steps, snp_half=False)
    # instrumented indel pass over the HBM-resident pack: stage times by HIP events
    eng.enable_timing(True)
    rt = gip.indel_sites_device(eng, job.pack, job.reads_c, L, job.chunks, fetch=False, **job.kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.indel_forward(_lib.MODEL_INDEL, rt["x"])
    e1.record()
    torch.cuda.synchronize()
    k9_ms = e0.elapsed_time(e1)
    gip.indel_sites_fetch(eng, rt["n"], rt["sets"])
    eng.enable_timing(False)
    ms = np.zeros(6, np.float32)
    cells = np.zeros(2, np.int64)
    eng.L.nc_indel_sites_stage_ms(eng.ctx, _lib.npp(ms), _lib.npp(cells))
    band = np.zeros(6, np.int64)
    eng.L.nc_indel_sites_band_stats(eng.ctx, _lib.npp(band))
    n_i = int(rt["n"])
    k9_tf = INDEL_FLOP_PER_SITE * n_i / (k9_ms * 1e-3) / 1e12 if k9_ms > 0 else 0.0
    fill_cells_s = int(band[4]) / (float(ms[2]) * 1e-3) if ms[2] > 0 else 0.0
    peak = F16_MFMA_PEAK_TFLOPS / 3.0
    out = {"workload": "BASELINE.json configs[2]: SNP + indel pipeline over chr1-sized synthetic ONT 30x contigs (%d bp): SNP half %d chunks of 500 kb, %d candidate "
                       "sites per step; indel half (planted indels 1-50 bp, HP/PS tags) %d chunks of 100 kb, %d candidate sites (%d read windows aligned) per step"
                       % (L, len(chunks), ns // steps, len(job.chunks), ni // steps, int(rt["n_alignments"])),
           "value": (ns + ni) / dt, "unit": "candidate sites/s (SNP + indel, one timed region)", "steps": steps, "ms_per_step": dt / steps * 1e3,
           "ms_per_step_steady": steady, "steady_note": "median step-to-step time from the third step on: the first step of the region has nothing to hide its own two uploads under",
           "timed_region": "per step: SNP wire (%.0f MB) and indel wire (%.0f MB) from pinned host memory, each copy under the other half's kernels -> both halves' "
                           "per-site results in host memory -> native rules + VCF text of both on a host thread under the next step"
                           % (snp.wire.nbytes / 1e6, job.wire.nbytes / 1e6),
           "vcf_records_per_step_indel": nrec // steps,
           "snp_half": {"sites_per_step": ns // steps, "ms_per_step_alone": dt_s / steps * 1e3, "sites_s_alone": ns_a / dt_s,
                        "roofline": {"bound": "mfma", "kernel": "k5_trunk_lin", "achieved": trunk_tf, "peak": peak, "unit": "TFLOP/s", "frac": trunk_tf / peak,
                                     "avg_launch_ms": float(trunk_ms / trunk_n), "launches": trunk_n,
                                     "note": "HIP events on the trunk's launches inside the combined timed region"}},
           "indel_half": {"sites_per_step": ni // steps, "ms_per_step_alone": dt_i / steps * 1e3, "sites_s_alone": ni_a / dt_i,
                          "stages_ms": {"k7_scan_anchors_sets": float(ms[0]), "query_windows": float(ms[1]), "star_alignment_fill": float(ms[2]),
                                        "star_alignment_traceback": float(ms[3]), "k8_tensors_consensus": float(ms[4]), "allele_prediction": float(ms[5]),
                                        "k9_indel_cnn": float(k9_ms), "note": "HIP events of one instrumented pass over the HBM-resident pack (stages one after the other)"},
                          "alignments": {"band32": int(band[0]), "band64": int(band[1]), "full_matrix_by_width": int(band[2]), "full_matrix_after_edge_touch": int(band[3])},
                          "roofline": {"bound": "mfma", "kernel": "k10_indel_trunk_h3 + k3_fc1 (K9)", "achieved": k9_tf, "peak": peak, "unit": "TFLOP/s", "frac": k9_tf / peak,
                                       "ms": float(k9_ms)},
                          "roofline_alignment": {"bound": "valu issue", "kernel": "k_fill_band", "achieved": fill_cells_s, "peak": FILL_PEAK_CELLS_S, "unit": "DP cells/s",
                                                 "frac": fill_cells_s / FILL_PEAK_CELLS_S, "dp_cells": int(band[4]), "dp_cells_full_matrices": int(cells[0])}},
           "h2d": {"bytes_per_step": snp.wire.nbytes + job.wire.nbytes, "achieved_GBs": h2d_gbs},
           "setup_s": round(t_setup, 1)}
    del snp, job, rt
    torch.cuda.empty_cache()
    return out


def extra_from_bam(eng, local, n_contigs=4, L=9_000_000, keep=None):
    """From a BAM FILE through the product worker loop: snpCaller.caller (BGZF inflate + record decode + wire build on host threads for
    contig i + 1 while the GPU runs contig i, upload through the three-slot ring) -> candidate sites/s including ingest.  The BAM (4 contigs of
    9 Mb, ONT-like 30x with qualities, CIGARs and tags: ~1.5 GB) is written by test tooling from device-generated reads, streamed contig by contig into the Python writer (~40 s, untimed)."""
    import queue
    import shutil
    import tempfile

    from nanocaller_amd import generate_SNP_pileups as gsp
    from nanocaller_amd import snpCaller
    from nanocaller_amd.synth_device import make_device_workload
    from nanocaller_amd.utils import get_chunks
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamio
    tmp = tempfile.mkdtemp(prefix="nc_bench_bam_")
    t0 = time.perf_counter()
    # the file: ONT-like records (tools/ont_like_bam.py): qualities, the reads' own indels as CIGAR operations (~700 per 10 kb read), soft clips,
    # NM / MD / HP / PS tags -- what a real alignment file makes the inflate and the record decode work through
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ont_like_bam
    bam, refs, fasta, fstats = ont_like_bam.make_files(eng, tmp, n_contigs, L, depth=30.0, seed0=7000, level=1)
    fa = os.path.join(tmp, "b.fa")
    bamio.write_fasta(fa, fasta[0][0], fasta[0][1], extra=fasta[1:])
    del fasta
    t_files = time.perf_counter() - t0
    regions = [(n, 1, ln, "diploid") for n, ln in refs]
    base_params = dict(regions_list=regions, sam_path=bam, fasta_path=fa, mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1,
                       threshold=[0.4, 0.6], snp_model="ONT-HG002", cpu=16, prefix="t", sample="S", seq="ont", supplementary=False,
                       exclude_bed=None, suppress_progress=True, disable_coverage_normalization=False)
    out, texts = {}, {}
    from nanocaller_amd import device_bam
    # device_ingest: the file crosses PCIe and is inflated / cut into records / decoded into the pack in HBM (device_bam.py); host_ingest: the
    # same worker loop with inflate + decode + wire build on the host threads (round 3's route); serial_ingest: that without the pipelining
    for tag, env in (("device_ingest", {"NC_DEVICE_INGEST": "1"}), ("host_ingest", {"NC_DEVICE_INGEST": "0"}),
                     ("serial_ingest", {"NC_DEVICE_INGEST": "0", "NC_SERIAL_INGEST": "1"})):
        for k in ("NC_DEVICE_INGEST", "NC_SERIAL_INGEST"):
            os.environ.pop(k, None)
        os.environ.update(env)
        best = first = None
        for rep in range(3):                                          # first run and best of three: the first run of a route also pays its one-off costs (page-locked
            gsp.release_contig()                                      # buffers of the file's size, first launches); every run starts from the file
            device_bam.release()
            d = os.path.join(tmp, "%s%d" % (tag, rep))
            os.makedirs(d)
            params = dict(base_params, chunks_list=get_chunks(regions, 16), vcf_path=d, intermediate_snp_files_dir=d)
            q = queue.Queue()
            for c in params["chunks_list"]:
                q.put(c)
            files = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            snpCaller.caller(params, q, queue.Queue(), files, device=local)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            texts[tag] = open(files[0], "rb").read()
            n_rec = texts[tag].count(b"\n")
            if first is None:
                first = {"seconds": dt, "sites_s": n_rec / dt}
            if best is None or dt < best["seconds"]:
                best = {"seconds": dt, "sites_s": n_rec / dt, "records": n_rec, "first_run": first}
                if tag == "device_ingest":
                    best["stages_s"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in device_bam.LAST_LOAD.items()}
        out[tag] = best
    for k in ("NC_DEVICE_INGEST", "NC_SERIAL_INGEST"):
        os.environ.pop(k, None)
    gsp.release_contig()
    device_bam.release()
    size = os.path.getsize(bam)
    if keep is None:
        shutil.rmtree(tmp, ignore_errors=True)
    else:
        keep.extend([tmp, bam, fa, regions])                        # (tools/exp_from_bam.py goes on with the files)
    return {"workload": "%d contigs of %d bp, ONT 30x, one ONT-like BAM file (%.0f MB, BGZF level 1: base qualities, %d reads with %.0f CIGAR operations each on average -- "
                        "the reads' own deletions and 3 %% insertions --, soft clips on a third of them, NM / MD / HP / PS tags) + FASTA -> snpCaller.caller -> worker VCF file; "
                        "host threads: %d usable CPUs" % (n_contigs, L, size / 1e6, fstats["reads"], fstats["cigar_ops"] / max(1, fstats["reads"]), usable_cpus()),
            "file": fstats,
            "from_bam_sites_s": out["device_ingest"]["sites_s"],
            "unit": "candidate sites/s incl. file read, H2D of the file, BGZF inflate + record walk + decode in HBM, GPU, rules + text, file write",
            "device_ingest": out["device_ingest"], "host_ingest": out["host_ingest"], "serial_ingest": out["serial_ingest"],
            "vcf_identical_device_vs_host": texts["device_ingest"] == texts["host_ingest"], "bam_writing_s": round(t_files, 1),
            "bam_bytes": size,
            "note": "ingest is outside SURVEY 8d's timed region (its row n1); this is the product worker loop end to end, best of three runs per route over a file "
                    "the test tooling wrote moments before (page cache warm)"}


def extra_from_bam_indel(eng, local, L=8_000_000, depth=30, keep=None):
    """The indel callers from a BAM FILE: indelCaller.indel_run (device pipeline + K9 + native rules + worker file) over one contig with planted
    indels, the contig's pack and per-read sections made in HBM from the file (device_bam.py) or by the host threads' decode +
    nc_indel_pack_build + wire form.  The BAM is written by test tooling (untimed)."""
    import queue
    import shutil
    import tempfile

    from nanocaller_amd import device_bam, indelCaller
    from nanocaller_amd import generate_indel_pileups as gip
    from nanocaller_amd import generate_SNP_pileups as gsp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bamio
    t0 = time.perf_counter()
    w = bamio.make_pass2_world(seed=11, length=L, depth=depth)
    tmp = tempfile.mkdtemp(prefix="nc_bench_ibam_")
    bam, fa = os.path.join(tmp, "p.bam"), os.path.join(tmp, "p.fa")
    bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None), level=1)
    bamio.write_fasta(fa, w.chrom, w.ref)
    t_files = time.perf_counter() - t0
    out, texts = {}, {}
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
            params = dict(seq="ont", fasta_path=fa, win_size=40, small_win_size=4, mincov=4, maxcov=160, ins_t=0.4, del_t=0.6, supplementary=False,
                          exclude_bed=None, impute_indel_phase=False, indel_model="ONT-HG002", intermediate_indel_files_dir=d, prefix="t")
            jobs = queue.Queue()
            for s_ in range(1, w.length, 100_000):
                jobs.put(("indel", dict(chrom=w.chrom, start=s_, end=min(w.length, s_ + 100_000), ploidy="diploid", sam_path=bam)))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            path = indelCaller.indel_run(params, {}, jobs, queue.Queue(), [], aligner="device")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            texts[tag] = open(path).read()
            if best is None or dt < best:
                best = dt
        out[tag] = {"seconds": best, "records": texts[tag].count("\n"), "records_s": texts[tag].count("\n") / best}
    os.environ.pop("NC_DEVICE_INGEST", None)
    gsp.release_contig()
    gip._CONTIGS.clear()
    gip._DEV_INGEST.clear()
    device_bam.release()
    size = os.path.getsize(bam)
    if keep is None:
        shutil.rmtree(tmp, ignore_errors=True)
    else:
        keep.extend([tmp, bam, fa])
    return {"workload": "one contig of %d bp with planted indels, ONT-like %dx, %d reads, one BAM file (%.0f MB, BGZF level 1) + FASTA -> indelCaller.indel_run -> worker "
                        "VCF file" % (L, depth, w.n_reads, size / 1e6),
            "device_ingest": out["device_ingest"], "host_ingest": out["host_ingest"], "vcf_identical_device_vs_host": texts["device_ingest"] == texts["host_ingest"],
            "files_writing_s": round(t_files, 1), "note": "best of three runs per route; every run starts from the file"}


def trunk_traffic_from_profiles():
    """HBM bytes per site of the dominant kernel from the committed PMC passes (profiles/trunk_traffic.json, written from the
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes by tools/pmc_summary.py; FETCH_SIZE doubled as MI355X_MICROARCH.md says)"""
    try:
        with open(os.path.join(ROOT, "profiles", "trunk_traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def traffic_build_check(record, sources):
    """(ok, note): counter passes are reported only when they were taken on the kernel sources this run executes -- the json carries the
    sha-256 tag of those files (tools/build_tag.py), compared with the files beside this script"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from build_tag import build_tag
    have = build_tag(sources, ROOT)
    want = record.get("build_tag")
    if want == have:
        return True, None
    return False, "the committed PMC passes (%s) belong to another build of %s (tag %s, this build %s): not reported" % (
        record.get("source", "profiles"), ", ".join(os.path.basename(s_) for s_ in sources), want, have)


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
    numa = None
    if use_dist and world > 1 and not one_gpu:
        from nanocaller_amd.numa import bind_rank                  # before the engine page-locks memory or starts threads
        numa = bind_rank(local)
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
    # --repeat R: R timed regions of K steps each, every one bracketed by barrier + synchronize; the line reports the MEDIAN region
    # (value, ms_per_step) and the spread
    dts, n_sites_rank, r = [], 0, None
    trunk_ms = trunk_launches = 0.0
    for _ in range(max(1, args.repeat)):
        eng.timing_sums()                                           # (resets nothing: sums are read as differences below)
        s_before, _ = eng.timing_sums()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_sites_rank, r = run_units(eng, uploader, contigs, args.steps * per_step, params, chunks, local, not args.no_overlap, args.resident)
        torch.cuda.synchronize()
        barrier()
        dts.append(time.perf_counter() - t0)
        s_after, _ = eng.timing_sums()
        trunk_ms += s_after[4] - s_before[4]
        trunk_launches += s_after[5] - s_before[5]
    from nanocaller_amd.shard import dist_max as _dmax
    dts = [_dmax(d) for d in dts]                                 # MAX over ranks, region by region
    order = np.argsort(dts)
    dt = dts[order[len(dts) // 2]]
    gc.enable()
    eng.enable_timing(False)                                      # (trunk_ms / trunk_launches: HIP-event totals over ALL the timed regions)
    h2d_gbs, h2d_ms, h2d_bytes = uploader.h2d_rate()
    from nanocaller_amd.shard import dist_max, dist_sum
    total_sites = dist_sum(n_sites_rank)                # whole-job aggregate over the K steps of one region
    # per-rank upload rates (N > 1: the ranks' copies share the host's DRAM and PCIe root complexes)
    per_rank_h2d = None
    if use_dist and world > 1:
        import torch.distributed as dist
        mine_t = torch.tensor([h2d_gbs, float(h2d_bytes)], dtype=torch.float64, device="cpu" if one_gpu else torch.device("cuda", local))
        allt = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t)
        per_rank_h2d = [[float(t[0]), float(t[1])] for t in allt]
    # ---- N > 1: the indel half over the same sharded contig list (configs[3] is SNP + indel): every rank runs the indel pass over ITS contigs,
    # `--indel-passes` times, from pinned host memory, between barriers; no collective on the data path
    indel_leg = None
    if world > 1 and not args.no_indel_leg:
        try:
            jobs = [IndelJob(eng, L, seed=4813 + k) for k in mine]
            for j in jobs:
                j.drop_pack()
            from concurrent.futures import ThreadPoolExecutor

            def indel_units(n_pass):
                sites = 0
                seq = [j for _ in range(n_pass) for j in jobs]
                with ThreadPoolExecutor(max_workers=1) as pool:
                    pend = None
                    nxt = uploader.submit(seq[0].wire) if seq else None
                    for i, j in enumerate(seq):
                        tk = nxt
                        nxt = uploader.submit(seq[i + 1].wire) if i + 1 < len(seq) else None
                        r_i = j.from_host_pass(uploader, tk)
                        sites += int(r_i["n"])
                        if pend is not None:
                            pend.result()
                        pend = pool.submit(j.rules, r_i)
                    if pend is not None:
                        pend.result()
                return sites
            indel_units(1)
            if len(jobs) < len(uploader.slots):
                indel_units(2)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            isites = indel_units(args.indel_passes)
            torch.cuda.synchronize()
            barrier()
            idt = _dmax(time.perf_counter() - t0)
            itot = dist_sum(isites)
            indel_leg = {"value": itot / idt, "unit": "candidate sites/s (indel half, whole job)", "passes": args.indel_passes, "contigs_per_pass": job_contigs,
                         "ms_per_pass_over_the_job": idt / args.indel_passes * 1e3, "sites_per_pass": itot // max(1, args.indel_passes),
                         "wire_bytes_per_contig": jobs[0].wire.nbytes if jobs else None,
                         "note": "every rank: its contigs' indel passes from pinned host memory (upload ring, rules + text on a host thread), barrier + "
                                 "synchronize on both sides, MAX over ranks; outside `value`'s timed region"}
            del jobs
            torch.cuda.empty_cache()
        except Exception as e:                                      # never take the headline down
            indel_leg = {"error": "%s: %s" % (type(e).__name__, e)}
    # ---- N > 1: BASELINE.json configs[3], the whole-genome mix sharded by shard_plan (every rank takes part; rank 0 reports it)
    wgs_sharded = None
    if world > 1 and not args.no_wgs:
        try:
            wgs_sharded = wgs_sharded_block(eng, uploader, local, args.model, rank, world, barrier, args.wgs_scale)
        except Exception as e:                                      # never take the headline down
            wgs_sharded = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                barrier()
            except Exception:
                pass
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
        sites_timed = n_sites_rank * len(dts)                          # this rank's sites inside ALL the timed regions (the trunk events span them)
        cnn_tflops = SNP_FLOP_PER_SITE * n_sites / (stage_ms[2] * 1e-3) / 1e12 if stage_ms[2] > 0 else 0.0
        # dominant kernel = fused conv1+conv2+conv3 trunk: algorithmic FLOP of its launches in the timed region / their
        # summed HIP-event durations == FLOP per launch / average launch duration
        n_launch = max(1.0, trunk_launches)
        trunk_tflops = TRUNK_FLOP_PER_SITE * sites_timed / (trunk_ms * 1e-3) / 1e12 if trunk_ms > 0 else 0.0
        scan_bytes = c0.entries + L                            # (d+1) B/column, SURVEY.md 8d
        feat_bytes = (5403 - 2050) * n_sites                  # SURVEY.md 8d's 5,403 B/site with the tensor written as int16 (2,050 B) instead of fp32
        tt = trunk_traffic_from_profiles()
        traffic, stale_note = None, None
        eng.set_tensor_format(int16=not exact_fp32)                # (as the timed region ran: which trunk kernel does this context launch?)
        mfma_per_site, trunk_kernel = eng.trunk_info()
        eng.set_tensor_format(int16=False)
        if tt and tt.get("kernel") == trunk_kernel:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from build_tag import TRUNK_SOURCES
            tag_ok, stale_note = traffic_build_check(tt, TRUNK_SOURCES)
            if tag_ok:
                traffic = tt["bytes_per_site"] * sites_timed / n_launch
        common = {"achieved": trunk_tflops, "unit": "TFLOP/s", "traffic": traffic,
                  "traffic_note": ("HBM bytes per launch = bytes per site from the committed PMC passes (%s) x sites per launch of this run"
                                   % tt.get("source", "profiles/trunk_traffic.json")) if traffic else (stale_note or "no PMC pass committed for this kernel build"),
                  "launches_in_timed_region": n_launch, "avg_launch_ms": float(trunk_ms / n_launch),
                  "flop_per_launch": TRUNK_FLOP_PER_SITE * sites_timed / n_launch,
                  "cnn_stage_tflops": cnn_tflops, "cnn_stage_ms": float(stage_ms[2])}
        if exact_fp32:
            roofline = {"bound": "mfma", "kernel": "k4_conv12: fused conv1+conv2+conv3 of the SNP CNN, fp32 MFMA 16x16x4",
                        "peak": FP32_MFMA_PEAK_TFLOPS, "frac": trunk_tflops / FP32_MFMA_PEAK_TFLOPS, **common}
        else:
            peak = F16_MFMA_PEAK_TFLOPS / 3.0
            exec_tflops = mfma_per_site * 16384.0 * sites_timed / (trunk_ms * 1e-3) / 1e12 if trunk_ms > 0 else 0.0
            roofline = {"bound": "mfma", "kernel": "%s: fused conv1+conv2+conv3 of the SNP CNN, three roles on three sites; f16 MFMA 16x16x32, f32 accumulate" % trunk_kernel,
                        "kernel_note": "fp32-equivalent products as hi*hi + hi*lo + lo*hi (3 f16 products); k5_trunk_lin runs conv1 on the integer tensor entries "
                                       "(exact in f16) with the coverage scale on the accumulators: 2 products there",
                        "peak": peak, "peak_note": "dense f16 MFMA peak 2500 TF / 3 products per fp32-equivalent product",
                        "frac": trunk_tflops / peak, "executed_mfma_per_site": mfma_per_site, "executed_f16_mfma_tflops": exec_tflops,
                        "executed_frac_of_f16_peak": exec_tflops / F16_MFMA_PEAK_TFLOPS,
                        "vs_fp32_mfma_peak": trunk_tflops / FP32_MFMA_PEAK_TFLOPS, **common}
        wire_b = c0.wire.nbytes
        out = {
            "metric": "candidate sites/sec (pileup+CNN)", "value": value, "unit": "sites/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "repeat": {"regions": len(dts), "value_is": "median region", "ms_per_step_min": min(dts) / args.steps * 1e3, "ms_per_step_max": max(dts) / args.steps * 1e3,
                       "sites_s_min": total_sites / max(dts), "sites_s_max": total_sites / min(dts)},
            "scaling": scaling, "vs_baseline": None, "dtype": "f32" if exact_fp32 else "f32 (f16x3 split MFMA, f32 accumulate)", "data": "synthetic",
            "config": {"workload": "configs[1]: SNP-only pileup+CNN, synthetic HG002-like %s %gx %s, chr20-sized contigs (%d bp)" % (args.tech.upper(), args.depth, args.ploidy, L),
                       "chunks_per_contig": len(chunks), "chunk_bp": 500000, "distinct_contigs_cycled": len(contigs) if world == 1 else None,
                       "sharding": (None if world == 1 else ("%d contigs per step in contiguous blocks over %d GPUs (%d on rank 0)" % (job_contigs, world, len(mine)))
                                    if scaling == "strong" else "1 contig per GPU per step"),
                       "timed_region": "HBM-resident packs (--resident)" if args.resident else
                       "pinned host wire -> H2D (own stream, 3-slot ring) -> expand -> scan -> tensors -> CNN -> results in pinned host memory",
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
                    "note": "copies run on their own stream under the previous contig's compute; expansion is on the compute stream",
                    "per_rank_achieved_GBs": [round(v[0], 2) for v in per_rank_h2d] if per_rank_h2d else None,
                    "host_dram_read_GBs_all_ranks": (sum(v[1] for v in per_rank_h2d) / sum(dts)) / 1e9 if per_rank_h2d else None},
            "three_numbers": {"kernels_only_sites_s": n_sites / ((stage_ms[0] + stage_ms[1] + stage_ms[2] + expand_ms) * 1e-3),
                              "with_h2d_d2h_sites_s": value if not args.resident else None,
                              "hbm_resident_with_d2h_sites_s": rs_sites / rs_dt,
                              "end_to_end_incl_vcf_text_serial_sites_s": n_sites / ((dt / max(1, n_units) * 1e3 + vcf_ms) * 1e-3),
                              "end_to_end_incl_vcf_text_pipelined_sites_s": pipe_sites / pipelined_dt,
                              "vcf_text_ms": vcf_ms, "vcf_bytes": len(vcf),
                              "note": "rank 0, per GPU; with_h2d_d2h = the timed region (= value at N=1); hbm_resident = the same passes over a pack "
                                      "already in HBM (round 1's headline); pipelined = + VCF text of pass i formatted on a host thread while the "
                                      "GPU runs pass i+1 (snpCaller.caller), uploads included"},
            "indel_leg": indel_leg, "wgs_sharded": wgs_sharded,
            "numa": ({"rank0_bound": numa["bound"], "node": numa["node"], "cpus": len(numa["cpus"]) if numa["cpus"] else None, "note": numa["note"]} if numa else None),
            "range_guard": {"x_limit": eng.x_limit(_lib_kind(args.ploidy)), "sites_rerun_on_exact_trunk": int(r.get("range_reruns", 0)) if r else None,
                            "note": "fp16x3 trunk: sites whose scaled tensor exceeds the model's proven-safe input bound are re-run on the exact fp32 trunk "
                                    "(nc_cnn_range_watch); 0 = every result of the timed region is proven inside the fp16 range"},
            "stages": {"expand_ms": expand_ms, "scan_ms": float(stage_ms[0]), "scan_GBs": scan_bytes / (stage_ms[0] * 1e-3) / 1e9 if stage_ms[0] else 0,
                       "scan_frac_hbm": scan_bytes / (stage_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS if stage_ms[0] else 0,
                       "featurize_ms": float(stage_ms[1]),
                       "featurize_GBs": feat_bytes / (stage_ms[1] * 1e-3) / 1e9 if stage_ms[1] else 0,
                       "featurize_frac_hbm": feat_bytes / (stage_ms[1] * 1e-3) / 1e9 / HBM_PEAK_GBS if stage_ms[1] else 0,
                       "cnn_ms": float(stage_ms[2]), "trunk_ms_per_contig": float(trunk_ms / max(1, n_units * len(dts)))},
        }
        if wgs_sharded is not None:
            cfg = out["config"]
            if "error" in wgs_sharded:
                cfg["wgs_sharded_error"] = str(wgs_sharded["error"])[:120]
            else:
                cfg.update({"wgs_sharded_value": wgs_sharded["value"], "wgs_sharded_s_per_pass": wgs_sharded["s_per_pass"], "wgs_sharded_sites": wgs_sharded["sites"],
                            "wgs_sharded_rank_imbalance": wgs_sharded["rank_imbalance_max_over_mean"], "wgs_sharded_scale": args.wgs_scale,
                            "wgs_sharded_h2d_GBs_min": min(wgs_sharded["rank_h2d_GBs"]), "wgs_sharded_h2d_GBs_max": max(wgs_sharded["rank_h2d_GBs"]),
                            "wgs_sharded_host_read_GBs": wgs_sharded["host_pinned_read_GBs_all_ranks"]})
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = args.cpu_sample_chunks or len(chunks)
            cb, parity = cpu_baseline_processes(pack, c0.info, chunks, params, args.model, r0, n_cpu)
            out["cpu_baseline"] = cb
            out["parity"] = parity
            cbt, _ = cpu_baseline(pack, c0.info, chunks, params, args.model, r0, min(len(chunks), max(1, usable_cpus())))
            out["cpu_baseline"]["thread_variant"] = {"value": cbt["value"], "cores": cbt["cores"], "sample": cbt["sample"]}
        if world == 1 and not args.no_configs2:
            # BASELINE.json configs[2] as a first-class block of the line: chr1-sized SNP + indel, one timed region
            contigs.clear()
            c0 = pack = None
            torch.cuda.empty_cache()
            try:
                out["configs2_snp_indel_chr1"] = configs2_block(eng, uploader, local, args.model, args.configs2_steps, args.configs2_length)
            except Exception as e:
                out["configs2_snp_indel_chr1"] = {"error": "%s: %s" % (type(e).__name__, e)}
            # the summary of configs[2] rides INSIDE `config` as SCALARS (a record parser keeps scalar / short-string leaves of `config` only)
            c2 = out["configs2_snp_indel_chr1"]
            cfg = out["config"]
            if "error" in c2:
                cfg["configs2_error"] = str(c2["error"])[:120]
            else:
                cfg.update({"configs2_workload": "configs[2]: SNP+indel, chr1-sized synthetic ONT 30x, one timed region from pinned host memory",
                            "configs2_value": c2["value"], "configs2_ms_per_step": c2["ms_per_step"], "configs2_ms_per_step_steady": c2["ms_per_step_steady"], "configs2_steps": c2["steps"],
                            "configs2_snp_sites": c2["snp_half"]["sites_per_step"], "configs2_indel_sites": c2["indel_half"]["sites_per_step"],
                            "configs2_snp_ms": c2["snp_half"]["ms_per_step_alone"], "configs2_indel_ms": c2["indel_half"]["ms_per_step_alone"],
                            "configs2_indel_sites_s": c2["indel_half"]["sites_s_alone"],
                            "configs2_frac_trunk": c2["snp_half"]["roofline"]["frac"], "configs2_frac_k9": c2["indel_half"]["roofline"]["frac"],
                            "configs2_frac_fill_issue_model": c2["indel_half"]["roofline_alignment"]["frac"],
                            "configs2_h2d_GBs": c2["h2d"]["achieved_GBs"], "configs2_wire_bytes": c2["h2d"]["bytes_per_step"]})
        if world == 1 and not args.no_wgs:
            # the metric's own configuration at N = 1: one pass over 24 contigs at GRCh38 lengths, both halves (scalars in `config` as well)
            contigs.clear()
            c0 = pack = None
            torch.cuda.empty_cache()
            cfg = out["config"]
            try:
                wg = wgs_block(eng, uploader, local, args.model, args.wgs_passes, args.wgs_scale)
                out["wgs_n1"] = wg
                cfg.update({"wgs_workload": wg["workload"][:120], "wgs_value": wg["value"], "wgs_s_per_pass": wg["s_per_pass"], "wgs_sites": wg["sites_per_pass"],
                            "wgs_snp_sites": wg["snp_sites_per_pass"], "wgs_indel_sites": wg["indel_sites_per_pass"], "wgs_contigs": wg["contigs"], "wgs_bp": wg["bp"],
                            "wgs_wire_bytes": wg["wire_bytes_snp"] + wg["wire_bytes_indel"], "wgs_h2d_GBs": wg["h2d_achieved_GBs"],
                            "wgs_frac_trunk": wg["frac_snp_trunk_f16x3"], "wgs_setup_s": wg["setup_s"]})
            except Exception as e:
                out["wgs_n1"] = {"error": "%s: %s" % (type(e).__name__, e)}
                cfg["wgs_error"] = ("%s: %s" % (type(e).__name__, e))[:120]
        if world == 1 and not args.no_extra:
            # other configurations, outside the headline's timed region (BASELINE.json configs[4], the exact-fp32 trunk,
            # and the indel half of configs[2]); each with its own workload and roofline
            contigs.clear()
            c0 = pack = None
            torch.cuda.empty_cache()
            extra = {}
            try:
                extra["hifi60x_haploid"] = extra_snp_config(eng, uploader, local, L, 60.0, "hifi", "CCS-HG002", "haploid", False, 12,
                                                            "SNP-only pileup+CNN, HiFi 60x haploid model (--haploid_genome), pacbio neighbour buckets, chr20-sized contig")
                extra["exact_fp32_trunk"] = extra_snp_config(eng, uploader, local, L, args.depth, args.tech, args.model, args.ploidy, True, 8,
                                                             "headline workload with the exact fp32 MFMA trunk (k4_conv12) on float32 tensors")
                extra["indel_pipeline"] = extra_indel_config(eng, uploader, local, L)
                extra["indel_haploid_260"] = extra_indel_haploid_config(eng, L)
                extra["from_bam"] = extra_from_bam(eng, local)
                extra["from_bam_indel"] = extra_from_bam_indel(eng, local)
            except Exception as e:                                  # an extra must never take the headline line down
                extra["error"] = "%s: %s" % (type(e).__name__, e)
            out["extra_configs"] = extra
        # what a record parser keeps of `config`: scalar leaves and strings of at most 120 characters -- enforce it here
        for k, v in list(out["config"].items()):
            assert not isinstance(v, (dict, list, tuple)), "config[%s] must be a scalar" % k
            if isinstance(v, str) and len(v) > 120:
                out["config"][k] = v[:117] + "..."
        print(json.dumps(out, default=float), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
