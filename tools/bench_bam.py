"""BAM ingest throughput (row n1): writes a synthetic coordinate-sorted BAM with tests/bamio.py, then times the native
reader (parallel BGZF inflate + CIGAR decode) and the host packer."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bamio
from nanocaller_amd.bam import BamFile, read_bam
from nanocaller_amd.pack import pack_world

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.Generator(np.random.PCG64(3))
ref = "".join("AGTC"[i] for i in rng.integers(0, 4, size=L))
recs, depth, pos = [], 30, 0
n_reads = L * depth // 10_000
starts = np.sort(rng.integers(0, L - 12_000, size=n_reads))
t0 = time.perf_counter()
for k, p0 in enumerate(starts):
    n = int(rng.integers(6_000, 12_000))
    seq = np.frombuffer(ref[p0:p0 + n].encode(), np.uint8).copy()
    mut = rng.random(n) < 0.04
    seq[mut] = np.frombuffer(b"AGTC", np.uint8)[rng.integers(0, 4, size=int(mut.sum()))]
    cig, q, r = [], [], 0
    while r < n:
        m = int(min(n - r, rng.integers(50, 400)))
        cig.append(("M", m)); q.append(seq[r:r + m].tobytes().decode()); r += m
        if r < n:
            if rng.random() < 0.5:
                d = int(rng.integers(1, 6)); cig.append(("D", d)); r += d
            else:
                i = int(rng.integers(1, 6)); cig.append(("I", i)); q.append("A" * i)
    if cig[-1][0] != "M":
        cig.append(("M", 1)); q.append("A")
    recs.append(dict(name="r%d" % k, flag=16 if k & 1 else 0, pos0=int(p0), cigar=cig, seq="".join(q), tags={"HP": 1 + (k & 1), "PS": 1000}))
d = tempfile.mkdtemp()
bam, fa = os.path.join(d, "b.bam"), os.path.join(d, "b.fa")
bamio.write_bam(bam, "c", L + 20_000, recs)
bamio.write_fasta(fa, "c", ref + "A" * 20_000)
print("wrote %s: %.1f MB, %d reads, %.1f s" % (bam, os.path.getsize(bam) / 1e6, len(recs), time.perf_counter() - t0))
for rep in range(3):
    t = time.perf_counter()
    dd = BamFile(bam).decode("c", 1, L + 20_000, keep_seq=False)
    dt = time.perf_counter() - t
    print("decode: %.3f s  -> %.1f M pileup entries/s, %.1f MB/s of BAM" % (dt, len(dd["codes"]) / dt / 1e6, os.path.getsize(bam) / dt / 1e6))
from nanocaller_amd.bam import decode_parallel
for threads in (1, 4, 8, 16, 32):
    t = time.perf_counter()
    dd = decode_parallel(bam, "c", 1, L + 20_000, threads=threads, min_region=50_000)
    dt = time.perf_counter() - t
    print("decode_parallel(%2d threads): %.3f s -> %.1f M pileup entries/s" % (threads, dt, len(dd["codes"]) / dt / 1e6))
t = time.perf_counter()
w = read_bam(bam, fa, "c")
t1 = time.perf_counter()
hp = pack_world(w)
print("read_bam %.3f s + pack %.3f s" % (t1 - t, time.perf_counter() - t1))
