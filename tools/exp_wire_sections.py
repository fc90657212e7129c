"""Section sizes of the indel half's transfer form (experiment): python tools/exp_wire_sections.py [length]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from nanocaller_amd.engine import get_engine  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 64_444_167
eng = get_engine(0)
job = bench.IndelJob(eng, L)
w = job.wire
tot = w.nbytes
print("indel wire of %d bp: %.1f MB; %d reads, %d indel events, %d pileup entries" % (L, tot / 1e6, job.info["n_reads"], job.info["n_events"], job.info["pileup_entries"]))
for name, (off, dt, cnt) in sorted(w.sections.items(), key=lambda kv: -kv[1][2] * np.dtype(kv[1][1]).itemsize):
    nb = cnt * np.dtype(dt).itemsize
    print("  %-14s %10.2f MB  %5.1f %%  (%d x %s)" % (name, nb / 1e6, 100.0 * nb / tot, cnt, np.dtype(dt).name))
ev = w.buf.numpy()[w.sections["events"][0]:w.sections["events"][0] + 2 * w.sections["events"][2]].view(np.uint16)
codes = ev >> 12
print("  wire events by code (A G T C del other):", [int((codes == c).sum()) for c in range(5)], int((codes > 4).sum()))
snp = bench.Contig(eng, L, 30.0, "ont", seed=812, keep_pack=False)
print("SNP wire: %.1f MB" % (snp.wire.nbytes / 1e6))
for name, (off, dt, cnt) in sorted(snp.wire.sections.items(), key=lambda kv: -kv[1][2] * np.dtype(kv[1][1]).itemsize)[:6]:
    nb = cnt * np.dtype(dt).itemsize
    print("  %-14s %10.2f MB  %5.1f %%" % (name, nb / 1e6, 100.0 * nb / snp.wire.nbytes))
