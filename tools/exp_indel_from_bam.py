"""The bench's from-BAM block of the indel callers at another size (experiment driver, GPU): python tools/exp_indel_from_bam.py [contig length] [depth]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nanocaller_amd.engine import get_engine

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print(json.dumps(bench.extra_from_bam_indel(get_engine(0), 0, L=L, depth=depth), indent=1))
