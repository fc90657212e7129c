#!/usr/bin/env python3
"""profiles/<TAG>_pmc.md from the round's counter passes (tools/round_profiles.sh): per-kernel HBM bytes of the indel pipeline and of the SNP path.
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read: MI355X_MICROARCH.md).
usage: pmc_md.py TAG [gpurun_out]"""
import csv
import json
import re
import sys
from collections import defaultdict

tag = sys.argv[1]
O = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"


def sums(path, counter):
    out, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "at::native" in r["Kernel_Name"]:
            continue
        m = re.search(r"\(anonymous namespace\)::(k[0-9]*_[a-zA-Z0-9_]+(?:<[^>]*>)?)", r["Kernel_Name"])
        if m and not re.match(r"k_(reads|truth|selftest|copy)", m.group(1)):      # (not the workload generators / copy kernels)
            out[m.group(1)] += float(r["Counter_Value"]) * 1024.0
            n[m.group(1)] += 1
    return out, n


def table(prefix, passes, title, unit):
    f, nf = sums("%s/%s_%s_FETCH_SIZE/p_counter_collection.csv" % (O, prefix, tag), "FETCH_SIZE")
    w, _ = sums("%s/%s_%s_WRITE_SIZE/p_counter_collection.csv" % (O, prefix, tag), "WRITE_SIZE")
    rows = sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0)))
    lines = ["", title, "", "| kernel | dispatches | read (corrected), GB per %s | written, GB per %s |" % (unit, unit), "|---|---|---|---|"]
    for k in rows:
        rd, wr = 2 * f.get(k, 0) / passes / 1e9, w.get(k, 0) / passes / 1e9
        if rd + wr < 0.005:
            continue
        lines.append("| `%s` | %d | %.2f | %.2f |" % (k, nf.get(k, 0), rd, wr))
    return lines


tt = json.load(open("%s/%s_trunk_traffic.json" % (O, tag)))
it = json.load(open("%s/%s_indel_traffic.json" % (O, tag)))
snp_passes = tt["contig_passes"][0]
out = ["# %s -- HBM traffic by counter of the round's last build" % tag, "",
       "`rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate runs, no other tracing; `tools/round_profiles.sh %s`)." % tag,
       "Counters are KiB, summed over a kernel's dispatches; FETCH_SIZE is doubled (the gfx950 correction of MI355X_MICROARCH.md)."]
out += table("pmci", it["passes"], "## Indel pipeline: `python tools/bench_indel_pipe.py 64444167 2` (%d passes, %d candidate sites, 1.07 M read windows each)"
             % (it["passes"], it["sites_per_pass"]), "pass")
out += ["", "Per stage (`profiles/indel_traffic.json`, bytes per candidate site: what `bench.py` reports as `traffic` of the indel stages):", ""]
for s, v in sorted(it["stages"].items(), key=lambda kv: -kv[1]["bytes_per_site"]):
    out.append("* %s: %.1f KB read + %.1f KB written" % (s, v["read_bytes_per_site_corrected"] / 1e3, v["write_bytes_per_site"] / 1e3))
out += table("pmc", snp_passes, "## SNP path: `python bench.py --no-extra --no-configs2 --no-cpu-baseline --steps 4 --warmup 1 --repeat 1` (%d contig passes of %d sites)"
             % (snp_passes, tt["sites_per_pass"]), "contig pass")
out += ["", "Dominant kernel `%s`: %.0f B/site read + %.0f B/site written = **%.0f B per site** (`profiles/trunk_traffic.json`; algorithmic 2,050 B int16 tensor in + 6,912 B of conv3 "
        "activations out)." % (tt["kernel"], tt["read_bytes_per_site_corrected"], tt["write_bytes_per_site"], tt["bytes_per_site"]),
        "", "Kernel times of the same build: `%s_kernel_stats.csv` (SNP path, bench under `rocprofv3 --kernel-trace --stats`), `%s_indel_kernel_stats.csv` (indel pipeline); bench line: `%s_bench.json`."
        % (tag, tag, tag)]
open("profiles/%s_pmc.md" % tag, "w").write("\n".join(out) + "\n")
print("\n".join(out[:60]))
