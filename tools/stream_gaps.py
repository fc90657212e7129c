"""Idle time between consecutive kernels of a rocprofv3 kernel trace (csv): which hand-overs leave the GPU without work?
usage: python tools/stream_gaps.py <dir with *_kernel_trace.csv> [first_fraction_to_skip]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm.split("(")[0][:40]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
trunks = [r[0] for r in rows if r[2].startswith("k5_trunk")]
rows = [r for r in rows if r[0] >= trunks[int(skip * len(trunks))]]  # steady state: the later part of the trunk launches
gap = collections.Counter()
cnt = collections.Counter()
busy = 0
end = rows[0][0]
for s, e, n in rows:
    prev = getattr(sys.modules[__name__], "_prev", None)
    if s > end and prev is not None:
        gap[(prev, n)] += s - end
        cnt[(prev, n)] += 1
    busy += max(0, e - max(s, end))
    if e > end:
        end = e
        sys.modules[__name__]._prev = n
span = rows[-1][1] - rows[0][0]
print("span %.1f ms, GPU busy %.1f ms (%.1f %%), idle %.1f ms" % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
n_trunk = sum(1 for r in rows if r[2].startswith("k5_trunk")) / 3.0
print("~%.0f contig passes: idle %.3f ms per pass" % (n_trunk, (span - busy) / 1e6 / max(n_trunk, 1)))
for (a, b), g in gap.most_common(14):
    print("  %-40s -> %-40s %8.3f ms total, %6.1f us x %d" % (a, b, g / 1e6, g / 1e3 / cnt[(a, b)], cnt[(a, b)]))

# per-kernel totals in the same window (blit kernels of the runtime's copies show up here: CU time beside the compute kernels)
tot = collections.defaultdict(lambda: [0, 0])
for s_, e_, n_ in rows:
    tot[n_][0] += 1
    tot[n_][1] += e_ - s_
print("kernel totals in the window:")
for n_, (c_, t_) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-44s x %5d  %9.3f ms total  %9.1f us avg" % (n_, c_, t_ / 1e6, t_ / 1e3 / c_))
