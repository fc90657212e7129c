"""bench.py's configs[2] block on its own (experiment; under rocprofv3 --kernel-trace for tools/stream_gaps.py).  usage: exp_configs2_only.py [steps]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nanocaller_amd.engine import get_engine
from nanocaller_amd.wire import WireUploader
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
eng = get_engine(0)
uploader = WireUploader(eng, slots=int(os.environ.get("NC_UPLOAD_SLOTS", "3")))
uploader.timing = True
r = bench.configs2_block(eng, uploader, 0, "ONT-HG002", steps=steps)
print(json.dumps({k: r[k] for k in r if k not in ("note",)})[:3000])
