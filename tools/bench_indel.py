"""Throughput of the indel-path kernels on synthetic data (reporting helper; the headline bench is the SNP path):
K7 window scan (columns/s), K8 MSA-rows -> tensor (read sets/s), K9 indel CNN (sites/s)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nanocaller_amd import _lib
from nanocaller_amd.engine import get_engine
from nanocaller_amd.pack import pack_world
from nanocaller_amd.synth import add_indels, make_world
from nanocaller_amd.weights import Weights, get_indel_model

eng = get_engine(0)
# ---- K7
L = 3_000_000
w = add_indels(make_world(seed=5, length=L, depth=30, tech="ont", read_len_scale=1.0), seed=5)
dp = eng.upload(pack_world(w))
chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    nflag = 0
    for (a, b) in chunks:
        col = eng.indel_scan(dp, a, b, mincov=4, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6)
        nflag += int((col >= 0).sum())
    dt = time.perf_counter() - t
print("K7 indel window scan: %d chunks of 100 kb, %.1f ms -> %.1f M columns/s (%d flagged columns)" % (len(chunks), dt * 1e3, L / dt / 1e6, nflag))
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    cols = eng.indel_scan_batch(dp, chunks, mincov=4, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6)
    dt = time.perf_counter() - t
print("K7 batched (nc_indel_scan_batch, chunk = grid dimension): %.1f ms -> %.1f M columns/s (%d flagged)" % (dt * 1e3, L / dt / 1e6, sum(int((c >= 0).sum()) for c in cols)))
# ---- n4: star alignment of read sets to their reference windows (device, one call) vs the host statement (per set)
from nanocaller_amd import generate_indel_pileups as gip
rng = np.random.Generator(np.random.PCG64(4))
def _reads(ref, n):
    out = []
    for _ in range(n):
        q = list(ref)
        for p in rng.choice(len(q), size=6, replace=False):
            q[p] = "AGTC"[int(rng.integers(0, 4))]
        p = int(rng.integers(10, 140)); ln = int(rng.choice([-5, -2, -1, 1, 2, 4]))
        if ln > 0: q[p:p] = list("AGTC"[int(rng.integers(0, 4))] * ln)
        else: del q[p:p - ln]
        out.append("".join(q)[:160])
    return out
NS = 3072                                                       # 1,024 anchors x (hap0, hap1, all reads)
refs = ["".join("AGTC"[i] for i in rng.integers(0, 4, size=161)) for _ in range(NS)]
sets = [_reads(refs[s], 30 if s % 3 == 2 else 15) for s in range(NS)]
eng.star_msa_tensor(sets[:30], refs[:30])
torch.cuda.synchronize(); t = time.perf_counter()
x, cns, ncols = eng.star_msa_tensor(sets, refs)
torch.cuda.synchronize(); dt = time.perf_counter() - t
n_al = sum(len(r) for r in sets)
print("n4 device star alignment + tensors: %d read sets (%d alignments of 160 x 161): %.1f ms -> %.0f sets/s, %.2f M alignments/s, %.1f G DP cells/s (incl. host marshalling)"
      % (NS, n_al, dt * 1e3, NS / dt, n_al / dt / 1e6, n_al * 160 * 161 / dt / 1e9))
big_sets, big_refs = sets * 6, refs * 6                          # a contig's worth of anchors in one call: several waves per SIMD
torch.cuda.synchronize(); t = time.perf_counter()
eng.star_msa_tensor(big_sets, big_refs)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("   %d read sets (%d alignments) in one call: %.1f ms -> %.0f sets/s, %.2f M alignments/s, %.1f G DP cells/s"
      % (len(big_sets), 6 * n_al, dt * 1e3, len(big_sets) / dt, 6 * n_al / dt / 1e6, 6 * n_al * 160 * 161 / dt / 1e9))
t = time.perf_counter()
for s in range(0, 96):
    gip.star_aligner(None, sets[s], refs[s])
dt = (time.perf_counter() - t) / 96
print("   host star aligner (nc_star_msa, all usable cores): %.2f ms per set -> %.0f sets/s" % (dt * 1e3, 1 / dt))
# ---- K8
rng = np.random.Generator(np.random.PCG64(1))
S = 4096
rows = [rng.integers(0, 5, size=(30, 170)).astype(np.uint8) for _ in range(S)]
refs = [rng.integers(0, 5, size=170).astype(np.uint8) for _ in range(S)]
eng.indel_tensor(rows[:16], refs[:16])
torch.cuda.synchronize(); t = time.perf_counter()
x, cns = eng.indel_tensor(rows, refs)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("K8 MSA rows -> tensor (incl. host marshalling of %d read sets of 30x170): %.1f ms -> %.0f sets/s" % (S, dt * 1e3, S / dt))
# ---- K9
for name, kind, shape, flop in (("ONT-HG002", _lib.MODEL_INDEL, (15, 128, 2), 18_946_752), ("haploid", _lib.MODEL_INDEL_HAP, (5, 128, 2), 5_040_688)):
    eng.load_weights(kind, Weights(get_indel_model(name)))
    n = 65536
    xx = torch.rand((n,) + shape, device="cuda") - 0.3
    eng.indel_forward(kind, xx); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        eng.indel_forward(kind, xx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print("K9 indel CNN %-10s %d sites: %.2f ms -> %.2f M sites/s, %.1f TFLOP/s" % (name, n, dt * 1e3, n / dt / 1e6, n * flop / dt / 1e12))
