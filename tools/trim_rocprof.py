#!/usr/bin/env python3
"""Trim a rocprofv3 --stats kernel_stats.csv to this library's kernels (+ a one-line 'other' total) so the
summary committed under profiles/ stays small.  usage: trim_rocprof.py in.csv out.csv"""
import csv
import re
import sys

rows = list(csv.reader(open(sys.argv[1])))
hdr, body = rows[0], rows[1:]
pat = re.compile(r"\(anonymous namespace\)::(k[0-9]*_[a-z0-9_]+(<[^>]*>)?)")
ours, other = [], []
pat_mangled = re.compile(r"_ZN12_GLOBAL__N_1\d+(k[0-9]*_[a-z0-9_]+?)(I[A-Za-z0-9_]*E)?v?P")     # kernels rocprofv3 left mangled
for r in body:
    m = pat.search(r[0]) or pat_mangled.search(r[0])
    if m and "at::native" not in r[0]:
        ours.append([m.group(1)] + r[1:])
    else:
        other.append(r)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(hdr)
    w.writerows(ours)
    w.writerow(["<torch / runtime kernels (synthetic data generation, copies)>", sum(int(r[1]) for r in other),
                sum(int(r[2]) for r in other), "", "", "", "", ""])
