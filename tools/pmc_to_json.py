#!/usr/bin/env python3
"""HBM bytes per candidate site of the dominant kernel from two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) over
`bench.py`: -> profiles/trunk_traffic.json, which bench.py reads for roofline.traffic.
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md).
usage: pmc_to_json.py fetch_counter_collection.csv write_counter_collection.csv kernel-substring sites_per_unit out.json source-note"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from build_tag import TRUNK_SOURCES, build_tag


def total(path, want, counter):
    tot, disp = 0.0, set()
    feat = set()
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if r["Counter_Name"] != counter:
            continue
        if "k_featurize" in name:
            feat.add(r["Dispatch_Id"])
        if want in name and "at::native" not in name:
            tot += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return tot, len(disp), len(feat)


f, fd, fu = total(sys.argv[1], sys.argv[3], "FETCH_SIZE")
w, wd, wu = total(sys.argv[2], sys.argv[3], "WRITE_SIZE")
sites = float(sys.argv[4])
out = {"kernel": sys.argv[3], "bytes_per_site": (2 * f * 1024 / fu + w * 1024 / wu) / sites,
       "read_bytes_per_site_corrected": 2 * f * 1024 / fu / sites, "write_bytes_per_site": w * 1024 / wu / sites,
       "fetch_size_raw_KiB": f, "write_size_KiB": w, "launches": [fd, wd], "contig_passes": [fu, wu], "sites_per_pass": sites,
       "source": sys.argv[6], "build_tag": build_tag(TRUNK_SOURCES), "build_tag_of": list(TRUNK_SOURCES)}
json.dump(out, open(sys.argv[5], "w"), indent=1)
print(out)
