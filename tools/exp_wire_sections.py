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
# what one-byte forms would take (round 6 estimate): code events as (gap to the previous event of the block, 6 bits | base, 2 bits), fillers for gaps
# >= 63, code-4 events on the side; indel events as (distance, 5 bits | length code, 3 bits)
bo = w.buf.numpy()[w.sections["blk_off"][0]:w.sections["blk_off"][0] + 4 * w.sections["blk_off"][2]].view(np.uint32).astype(np.int64)
off = (ev & 0x3ff).astype(np.int64)
blk = np.repeat(np.arange(bo.size - 1), np.diff(bo))
first = np.r_[True, blk[1:] != blk[:-1]]
gap = np.where(first, off, off - np.r_[0, off[:-1]] - 1)
fill = int((gap // 63).sum())
side = int((codes >= 4).sum())
print("one-byte code events: %d events, %d fillers, %d side events -> %.1f MB (now %.1f)" % (ev.size, fill, side, (ev.size - side + fill + 4 * side) / 1e6, 2 * ev.size / 1e6))
d16 = w.buf.numpy()[w.sections["ev_d16"][0]:w.sections["ev_d16"][0] + 2 * w.sections["ev_d16"][2]].view(np.uint16).astype(np.int64)
dist, ln = d16 & 0x7ff, ((d16 >> 11) ^ 16) - 16
big = d16 == 0xFFFF
ok_len = np.isin(ln, [1, 2, 3, -1, -2, -3]) & ~big
fill5 = int((dist[~big] // 31).sum())
print("one-byte indel events: %d events, lengths in +-1..3: %.1f %%, %d fillers (5-bit distance) -> %.1f MB (now %.1f)" % (
    d16.size, 100.0 * ok_len.mean(), fill5, (d16.size + fill5 + 3 * int((~ok_len).sum())) / 1e6, 2 * d16.size / 1e6))
fill6 = int((dist[~big] // 63).sum())
ok2 = np.isin(ln, [1, -1, 2, -2]) & ~big
print("   6-bit distance | 2-bit length code (+1 -1 +2 -2): %.1f %% fit, %d fillers -> %.1f MB" % (100.0 * ok2.mean(), fill6, (d16.size + fill6 + 3 * int((~ok2).sum())) / 1e6))
