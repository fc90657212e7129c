"""Headline SNP step and SNP wire size with the difference events one byte each for every pack (NC_WIRE_EVB=2) against the default (experiment)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mode in ("1", "2", "1", "2"):
    env = dict(os.environ, NC_WIRE_EVB=mode)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extra", "--no-configs2", "--no-cpu-baseline", "--no-wgs", "--steps", "60", "--warmup", "5", "--repeat", "3"],
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    import json
    d = json.loads(out)
    print("NC_WIRE_EVB=%s: %.2f M sites/s, %.3f ms per step, wire %.1f MB, expand %.3f ms" % (mode, d["value"] / 1e6, d["ms_per_step"], d["h2d"]["wire_bytes_per_contig"] / 1e6, d["h2d"]["expand_ms"]), flush=True)
