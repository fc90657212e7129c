"""Banded star alignment against the full-matrix route on one synthetic indel contig (experiment driver, GPU).
usage: python tools/exp_band.py [length] [margins, e.g. 4,6,8] -> per margin: class counts, stage times, and how many sites / tensors / alleles
differ from the full-matrix run of the same process."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nanocaller_amd import _lib
from nanocaller_amd import generate_indel_pileups as gip
from nanocaller_amd.engine import get_engine
from nanocaller_amd.synth_device import make_indel_device_workload

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
margins = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [6]
haploid = os.environ.get("NC_EXP_HAPLOID") == "1"
window_after = int(os.environ.get("NC_EXP_WINDOW", "160"))
eng = get_engine(0)
pack, reads_c, info = make_indel_device_workload(eng, L, depth=30.0, seed=int(os.environ.get("NC_EXP_SEED", "812")))
chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=window_after, haploid=haploid)


def run(mode, margin, timing):
    assert eng.L.nc_indel_sites_band(eng.ctx, mode, margin) == 0
    eng.enable_timing(timing)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = gip.indel_sites_device(eng, pack, reads_c, L, chunks, fetch=False, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    r.update(gip.indel_sites_fetch(eng, r["n"], r["sets"]))
    eng.enable_timing(False)
    ms = np.zeros(6, np.float32)
    cells = np.zeros(2, np.int64)
    eng.L.nc_indel_sites_stage_ms(eng.ctx, _lib.npp(ms), _lib.npp(cells))
    st = np.zeros(6, np.int64)
    eng.L.nc_indel_sites_band_stats(eng.ctx, _lib.npp(st))
    return r, dt, ms, st


run(0, 0, False)
full, dt_f, _, _ = run(0, 0, False)
_, _, ms_f, _ = run(0, 0, True)
xf = full["x"].clone()
print("full matrix: %d sites, %d alignments, %.2f ms; stages plan %.2f windows %.2f fill %.2f trace %.2f tensor %.2f alleles %.2f"
      % (full["n"], full["n_alignments"], dt_f, *ms_f), flush=True)
for m in margins:
    run(1, m, False)
    r, dt, _, st = run(1, m, False)
    _, _, ms, _ = run(1, m, True)
    same_sites = r["n"] == full["n"] and np.array_equal(r["pos"], full["pos"])
    dx = int((r["x"] != xf).reshape(r["n"], -1).any(1).sum()) if same_sites else -1
    dal = int(((r["ref_len"] != full["ref_len"]) | (r["alt_len"] != full["alt_len"])).any(1).sum()) if same_sites else -1
    alt_same = np.array_equal(r["alt"], full["alt"])
    tot = max(1, int(st[:3].sum()))
    print("margin %d: %.2f ms; classes: B32 %d (%.1f %%), B64 %d (%.1f %%), full by width %d (%.2f %%), re-run after an edge touch %d (%.3f %%); "
          "stages plan %.2f windows %.2f fill %.2f trace %.2f tensor %.2f alleles %.2f; sites equal %s, tensors differing %d, allele lengths differing %d, "
          "ALT bytes equal %s" % (m, dt, st[0], 100 * st[0] / tot, st[1], 100 * st[1] / tot, st[2], 100 * st[2] / tot, st[3], 100 * st[3] / tot, *ms,
                                  same_sites, dx, dal, alt_same), flush=True)
