#!/usr/bin/env python3
"""HBM bytes per candidate site of every stage of the device-resident indel pipeline from two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE)
over tools/bench_indel_pipe.py -> profiles/indel_traffic.json, which bench.py reads for the `traffic` of extra_configs.indel_pipeline.stages.
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md).
usage: pmc_indel_to_json.py fetch_counter_collection.csv write_counter_collection.csv passes sites_per_pass out.json source-note"""
import csv
import json
import re
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from build_tag import INDEL_SOURCES, build_tag

STAGE = {"k_hap_depth_b": "k7_scan_anchors_sets", "k_yield_rank_b": "k7_scan_anchors_sets", "k_entry_reads": "k7_scan_anchors_sets",
         "k_entry_cursors": "k7_scan_anchors_sets", "k_event_tiles": "k7_scan_anchors_sets", "k_pick": "k7_scan_anchors_sets", "k_sets": "k7_scan_anchors_sets",
         "k_flatten": "k7_scan_anchors_sets", "k_scan_excl": "k7_scan_anchors_sets", "k_windows": "query_windows",
         "k_windows16": "query_windows", "k_window_lists": "query_windows", "k_blk_chunks": "k7_scan_anchors_sets",
         "k_scan_part": "k7_scan_anchors_sets", "k_scan_apply": "k7_scan_anchors_sets",        # (round 5's two-launch scans: plan and allele stage alike, a few KB each)
         # the banded fills of the star alignment AND of allele_prediction (one kernel name; the allele sets are ~1/9 of the cells)
         "k_fill_band": "star_alignment_fill",
         # banded tracebacks + the full-matrix route of the alignments that do not fit a band (star and allele fallbacks share k_fill16q)
         "k_trace_band12": "star_alignment_traceback", "k_fill16q": "star_alignment_traceback", "k_fill16p": "star_alignment_traceback",
         "k_end_cells": "star_alignment_traceback", "k_trace16p": "star_alignment_traceback", "k_band_stats": "star_alignment_traceback",
         "k_site_tensor": "k8_tensors_consensus", "k_allele_trace16p": "allele_prediction", "k_allele_trace_b12": "allele_prediction",
         "k_allele_classes": "allele_prediction", "k_scan_rows": "allele_prediction",
         "k_alt_offsets": "allele_prediction", "k_alt_copy": "allele_prediction", "k10_indel_trunk_h3": "k9_indel_cnn", "k3_fc1": "k9_indel_cnn",
         "k_indel_heads": "k9_indel_cnn"}


def sums(path, counter):
    out = defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"\(anonymous namespace\)::(k[0-9]*_[a-z0-9_]+)", r.get("Kernel_Name", ""))
        if m and "at::native" not in r["Kernel_Name"] and m.group(1) in STAGE:
            out[m.group(1)] += float(r["Counter_Value"]) * 1024.0
    return out


f, w = sums(sys.argv[1], "FETCH_SIZE"), sums(sys.argv[2], "WRITE_SIZE")
passes, sites = float(sys.argv[3]), float(sys.argv[4])
kern, stage = {}, defaultdict(lambda: [0.0, 0.0])
for k in sorted(set(f) | set(w)):
    rd, wr = 2.0 * f.get(k, 0.0) / passes / sites, w.get(k, 0.0) / passes / sites
    kern[k] = {"read_bytes_per_site_corrected": rd, "write_bytes_per_site": wr}
    stage[STAGE[k]][0] += rd
    stage[STAGE[k]][1] += wr
out = {"sites_per_pass": sites, "passes": passes, "kernels": kern,
       "stages": {s: {"read_bytes_per_site_corrected": v[0], "write_bytes_per_site": v[1], "bytes_per_site": v[0] + v[1]} for s, v in stage.items()},
       "source": sys.argv[6], "build_tag": build_tag(INDEL_SOURCES), "build_tag_of": list(INDEL_SOURCES)}
json.dump(out, open(sys.argv[5], "w"), indent=1)
print(json.dumps(out["stages"], indent=1))
