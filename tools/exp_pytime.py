import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from nanocaller_amd import snpCaller, _lib
from nanocaller_amd.engine import get_engine, Engine
acc = {"scan_ret_to_feat": [], "feat_call": [], "scan_call": [], "fetch": []}
o_scan, o_feat = Engine.snp_scan, Engine.snp_featurize
def scan(self, *a, **k):
    t0 = time.perf_counter(); r = o_scan(self, *a, **k); t1 = time.perf_counter()
    acc["scan_call"].append(t1 - t0); self._t_scan_ret = t1; return r
def feat(self, *a, **k):
    t0 = time.perf_counter(); acc["scan_ret_to_feat"].append(t0 - self._t_scan_ret)
    r = o_feat(self, *a, **k); acc["feat_call"].append(time.perf_counter() - t0); return r
Engine.snp_scan, Engine.snp_featurize = scan, feat
sys.argv = ["bench.py", "--no-extra", "--no-configs2", "--no-cpu-baseline", "--steps", "40", "--repeat", "1"]
bench.main()
for k, v in acc.items():
    if v: print(k, "median %.1f us" % (np.median(v[len(v)//2:]) * 1e6), file=sys.stderr)
