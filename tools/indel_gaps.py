"""GPU idle time inside the passes of tools/bench_indel_pipe.py from a rocprofv3 kernel trace: which hand-overs leave the GPU without work?
usage: python tools/indel_gaps.py <dir with *_kernel_trace.csv>"""
import collections
import csv
import glob
import os
import sys

f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm.split("(")[0][:44]))
rows.sort()
# one cycle of the steady state: from one pass's `k_sets<true>` (once per plan) to the next one's
marks = [i for i, r in enumerate(rows) if r[2].startswith("k_sets<true>")]
if len(marks) < 4:
    sys.exit("need at least four passes")
a, b = marks[-3], marks[-2]
win = rows[a:b]
span = win[-1][1] - win[0][0]
gap, cnt = collections.Counter(), collections.Counter()
busy, end, prev = 0, win[0][0], None
for s, e, n in win:
    if s > end and prev is not None:
        gap[(prev, n)] += s - end
        cnt[(prev, n)] += 1
    busy += max(0, e - max(s, end))
    if e > end:
        end, prev = e, n
print("last pass: span %.2f ms, GPU busy %.2f ms (%.1f %%), idle %.2f ms, %d kernels" % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(win)))
for (x, y), g in gap.most_common(16):
    print("  %-36s -> %-36s %8.3f ms total, %7.1f us x %d" % (x, y, g / 1e6, g / 1e3 / cnt[(x, y)], cnt[(x, y)]))
tot = collections.defaultdict(lambda: [0, 0])
for s_, e_, n_ in win:
    tot[n_][0] += 1
    tot[n_][1] += e_ - s_
print("kernel totals in the pass:")
for n_, (c_, t_) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %-44s x %4d  %8.3f ms" % (n_, c_, t_ / 1e6))
