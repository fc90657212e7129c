"""Pin the CPU oracle (oracle/nc_oracle.c) against golden vectors produced by the REFERENCE's own code
(oracle/tools/make_goldens.py, run in the build container).  CPU-only."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests.util import GOLD, SNP_CASES, assert_tuple_matches_gold, load_snp_case


@pytest.mark.parametrize("case", SNP_CASES)
def test_snp_featuriser_matches_reference(case):
    world, dct, region, exclude, gold = load_snp_case(case)
    out = oracle.get_snp_testing_candidates(world, dct, region, exclude=exclude)
    assert_tuple_matches_gold(out, gold)


def test_golden_cases_cover_all_modes():
    seqs = {str(np.load(os.path.join(GOLD, "snp_%s.npz" % c))["seq"]) for c in SNP_CASES}
    assert seqs == {"ont", "short_ont", "ul_ont", "ul_ont_extreme", "pacbio"}


def test_get_cnd_pos_matches_reference():
    z = np.load(os.path.join(GOLD, "cnd_pos.npz"))
    n = int(z["n"])
    assert n >= 50
    for k in range(n):
        left, right = oracle.get_cnd_pos(int(z["c%d_v" % k]), z["c%d_sites" % k], str(z["c%d_seq" % k]))
        assert left == z["c%d_l" % k].tolist(), k
        assert right == z["c%d_r" % k].tolist(), k


def test_indel_tensor_matches_reference_msa():
    z = np.load(os.path.join(GOLD, "indel_msa.npz"))
    sym = "AGTC-"
    for k in range(int(z["n"])):
        out, cns = oracle.indel_tensor(z["m%d_rows" % k], z["m%d_ref" % k])
        gold = z["m%d_mat" % k]
        assert gold.shape == (5, 128, 2)
        assert np.array_equal(out.astype(np.float64), gold), k
        assert "".join(sym[c] for c in cns) == str(z["m%d_cns" % k])


def test_indel_window_scan_matches_reference_pass1():
    from tests.util import indel_scan_cases, load_world
    world = load_world("indel")
    cases = indel_scan_cases()
    assert len(cases) >= 6 and sum(len(c["pos"]) for c in cases) > 100
    for c in cases:
        pos, typ = oracle.indel_scan(world, c["start"], c["end"], mincov=c["mincov"], win_size=c["win_size"],
                                     small_win_size=c["small_win_size"], ins_t=c["ins_t"], del_t=c["del_t"],
                                     exclude=c["exclude"])
        assert np.array_equal(pos, c["pos"]) and np.array_equal(typ, c["type"]), (c["start"], c["end"])


def test_haploid_indel_window_scan_matches_reference_pass1():
    """generate_indel_pileups_haploid.py:185-241: the `variants` dict captured from the reference's own frame"""
    from tests.util import indel_scan_cases, load_world
    world = load_world("indel")
    cases = indel_scan_cases(haploid=True)
    assert len(cases) >= 5 and sum(len(c["pos"]) for c in cases) > 40
    for c in cases:
        pos, typ = oracle.indel_scan(world, c["start"], c["end"], mincov=c["mincov"], win_size=c["win_size"],
                                     small_win_size=c["small_win_size"], ins_t=c["ins_t"], del_t=c["del_t"],
                                     exclude=c["exclude"], haploid=True)
        assert np.array_equal(pos, c["pos"]) and np.array_equal(typ, c["type"]), (c["start"], c["end"])


def test_impute_indel_phase_scan_matches_reference_pass1():
    """generate_indel_pileups.py:197-304 with dct['impute_indel_phase']: `variants` AND `extra_variants` (the two read
    collections per imputed anchor) captured from the reference's own frame; also the plain branch of the Python
    restatement against the C oracle on the same world"""
    from tests.util import indel_impute_cases, load_impute_world
    world = load_impute_world()
    cases = indel_impute_cases()
    assert len(cases) >= 8 and sum(len(c["extra"]) for c in cases) > 300
    for c in cases:
        kw = dict(mincov=c["mincov"], win_size=c["win_size"], small_win_size=c["small_win_size"], ins_t=c["ins_t"], del_t=c["del_t"],
                  exclude=c["exclude"])
        v, x = oracle.indel_scan_impute(world, c["start"], c["end"], **kw)
        assert sorted(v) == c["pos"].tolist() and [v[p] for p in sorted(v)] == c["type"].tolist(), (c["start"], c["end"])
        assert x == c["extra"], (c["start"], c["end"])
    # thresholds nothing imputed can reach: the restatement must agree with the C oracle of the plain rule
    kw = dict(mincov=4, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6)
    v, x = oracle.indel_scan_impute(world, 20_000, 30_000, **dict(kw, mincov=4))
    pos, typ = oracle.indel_scan(world, 20_000, 30_000, **kw)
    assert not x and sorted(v) == pos.tolist() and [v[p] for p in sorted(v)] == typ.tolist() and len(v) > 3
