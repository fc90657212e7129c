#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: sum of each counter over dispatches.
usage: pmc_summary.py counter_collection.csv [kernel-substring]"""
import csv
import re
import sys
from collections import defaultdict

rows = csv.DictReader(open(sys.argv[1]))
want = sys.argv[2] if len(sys.argv) > 2 else None
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in rows:
    name = r.get("Kernel_Name", "")
    m = re.search(r"\(anonymous namespace\)::(k[0-9]*_[a-z0-9_]+(<[^>]*>)?)", name)
    if not m or "at::native" in name:
        continue
    k = m.group(1)
    if want and want not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r.get("Dispatch_Id", ""))
for k in agg:
    print(k, "dispatches=%d" % len(calls[k]))
    for c, v in sorted(agg[k].items()):
        print("   %-34s %.6g" % (c, v))
