"""Host-side mirror of the GPU-resident parts of the reference's indel featuriser
(nanocaller_src/generate_indel_pileups.py).

* `scan_indel_candidates(dct, chunk)` = pass 1 of get_indel_testing_candidates (:197-276): the `variants`
  dict {anchor position: 0 (long window) | 1 (small window)}; the per-column work runs in nc_indel_scan, the
  order-dependent `v <= prev` suppression (:249,267,273) is applied here.
* `msa_tensor(rows, ref_row)` = the histogram half of msa() (:57-71) through nc_indel_tensor.
Read slicing, MUSCLE and parasail (pass 2, :306-361) are on the far side of the boundary;
the impute_indel_phase branch (:278-304) is not covered.
"""
from __future__ import annotations

import numpy as np
import torch

from .engine import get_engine
from .generate_SNP_pileups import _exclude_rows, _resolve
from .pack import pack_world

_PACKS = {}


def pick_variants(col_type, start, win_size):
    """Apply `if v_pos <= prev: continue` (:249) to the per-column decisions; -> {anchor: type} (dict semantics:
    a later detection overwrites an equal anchor)."""
    variants = {}
    prev = 0
    lo = max(1, int(start))
    for c in np.nonzero(col_type >= 0)[0]:
        v = lo + int(c)
        if v <= prev:
            continue
        if col_type[c] == 0:
            prev = v + win_size
            variants[max(1, v - win_size)] = 0                      # :267-268
        else:
            prev = v + 10
            variants[max(1, v - 10)] = 1                            # :273-274
    return variants


def scan_indel_candidates(dct, chunk, device=0):
    if dct.get("impute_indel_phase"):
        raise NotImplementedError("impute_indel_phase (generate_indel_pileups.py:278-304) is not part of this build")
    world = _resolve(chunk["sam_path"], chunk["chrom"], dct.get("fasta_path"))
    excl_rows = _exclude_rows(dct, chunk["chrom"])
    key = (id(world), bool(dct.get("supplementary")), device)
    eng = get_engine(device)
    eng.use_torch_stream()
    if key not in _PACKS:
        _PACKS[key] = (eng.upload(pack_world(world, supplementary=bool(dct.get("supplementary")))), world)
    dp = _PACKS[key][0]
    excl = None
    if excl_rows:
        m = np.zeros(dp.n_tiles * dp.tile_size, np.uint8)
        for (a, b) in excl_rows:                                    # IntervalTree.overlaps(pos): a <= pos < b
            m[max(0, a - dp.tile_pos0):max(0, b - dp.tile_pos0)] = 1
        excl = torch.from_numpy(m).to(eng.device)
    col_type = eng.indel_scan(dp, chunk["start"], chunk["end"], mincov=dct["mincov"], win_size=dct["win_size"],
                              small_win_size=dct["small_win_size"], ins_t=dct["ins_t"], del_t=dct["del_t"], excl=excl)
    return pick_variants(col_type, chunk["start"], dct["win_size"])


def msa_tensor(rows_list, ref_rows_list, device=0):
    """-> (float64 [S,5,128,2] like msa()'s final_mat (:69-71), list of consensus strings with gaps removed (:64))."""
    eng = get_engine(device)
    eng.use_torch_stream()
    x, cns = eng.indel_tensor(rows_list, ref_rows_list)
    sym = "AGTC"
    return x.cpu().numpy().astype(np.float64), ["".join(sym[c] for c in row) for row in cns]
