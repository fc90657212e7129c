"""bench.py's indel_pipeline leg on its own (experiment): per-pass wall times of the pipelined from-pinned-memory loop.  usage: exp_indel_only.py [reps]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nanocaller_amd.engine import get_engine
from nanocaller_amd.wire import WireUploader
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
eng = get_engine(0)
uploader = WireUploader(eng, slots=3)
uploader.timing = True
r = bench.extra_indel_config(eng, uploader, 0, bench.CHR20_LEN, reps=reps)
print("value", r["value"], "ms_per_pass", r["ms_per_pass"])
print("pass_ms", r["pass_ms"])
print("resident", r["hbm_resident_serial"])
