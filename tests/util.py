"""Shared helpers for the test-suite (loading golden worlds / cases)."""
import os

import numpy as np

from nanocaller_amd.synth import World

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_worlds = {}


def load_world(name):
    if name not in _worlds:
        z = np.load(os.path.join(GOLD, "world_%s.npz" % name))
        R = z["read_start"].shape[0]
        _worlds[name] = World(chrom=str(z["chrom"]), ref=z["ref"].tobytes().decode(), read_start=z["read_start"],
                              read_end=z["read_end"], read_flag=z["read_flag"], read_off=z["read_off"],
                              codes=z["codes"], names=["r%07d" % i for i in range(R)])
        if "ev_off" in z:
            _worlds[name].meta.update(events=(z["ev_off"], z["ev_pos"], z["ev_len"]), hap=z["hap"], ps=z["ps"])
    return _worlds[name]


SNP_CASES = sorted(f[4:-4] for f in os.listdir(GOLD) if f.startswith("snp_") and f.endswith(".npz"))


def load_snp_case(case):
    z = np.load(os.path.join(GOLD, "snp_%s.npz" % case))
    world = load_world(str(z["world"]))
    if "name_id" in z:                                                   # the same alignments with shared read names / other flags (make_goldens.mates_world)
        import copy
        world = copy.copy(world)
        world.names = ["r%07d" % i for i in z["name_id"]]
        world.read_flag = z["read_flag"]
    dct = dict(threshold=[float(z["threshold"][0]), float(z["threshold"][1])], supplementary=bool(z["supplementary"]),
               mincov=int(z["mincov"]), maxcov=int(z["maxcov"]), min_allele_freq=float(z["min_allele_freq"]),
               min_nbr_sites=int(z["min_nbr_sites"]), seq=str(z["seq"]), exclude_bed=None)
    region = dict(chrom=world.chrom, start=int(z["start"]), end=int(z["end"]), ploidy=str(z["ploidy"]))
    exclude = [(world.chrom, int(a), int(b)) for a, b in z["exclude"]]
    gold = dict(pos=z["pos"], ref=z["ref"], mat=z["mat"].astype(np.float32), dp=z["dp"], freq=z["freq"],
                depth=float(z["depth"]), fwd_dp=z["fwd_dp"], rev_dp=z["rev_dp"])
    return world, dct, region, exclude, gold


def assert_tuple_matches_gold(out, gold):
    pos, ref, mat, dp, freq, depth, fwd, rev = out
    n = len(gold["pos"])
    assert len(pos) == n
    if n == 0:
        # reference returns empty lists and depth 0 (generate_SNP_pileups.py:193-197)
        for x in (ref, mat, dp, freq, fwd, rev):
            assert len(x) == 0
        assert depth == 0
        return
    assert np.array_equal(np.asarray(pos), gold["pos"])
    assert np.array_equal(np.asarray(ref), gold["ref"])
    assert np.asarray(mat).dtype == np.float32 and np.array_equal(np.asarray(mat), gold["mat"])
    assert np.array_equal(np.asarray(dp), gold["dp"])
    assert np.array_equal(np.asarray(freq), gold["freq"])          # float64, bit-exact
    assert float(depth) == gold["depth"]
    assert np.array_equal(np.asarray(fwd), gold["fwd_dp"])
    assert np.array_equal(np.asarray(rev), gold["rev_dp"])


def load_impute_world():
    """the indel world with unphased stretches + per-read inserted bases (inputs stored in indel_impute.npz)"""
    if "impute" not in _worlds:
        import copy
        from nanocaller_amd.synth import apply_impute_inputs
        z = np.load(os.path.join(GOLD, "indel_impute.npz"))
        base = load_world("indel")
        w = copy.copy(base)
        w.meta = dict(base.meta)
        _worlds["impute"] = apply_impute_inputs(w, z["hap"], z["ins_off"], z["ins_bases"])
    return _worlds["impute"]


def indel_impute_cases():
    z = np.load(os.path.join(GOLD, "indel_impute.npz"))
    out = []
    for k in range(int(z["n"])):
        xpos = z["s%d_xpos" % k]
        out.append(dict(start=int(z["s%d_start" % k]), end=int(z["s%d_end" % k]), mincov=int(z["s%d_mincov" % k]),
                        win_size=int(z["s%d_win_size" % k]), small_win_size=int(z["s%d_small_win_size" % k]),
                        ins_t=float(z["s%d_ins_t" % k]), del_t=float(z["s%d_del_t" % k]),
                        exclude=[(int(a), int(b)) for a, b in z["s%d_excl" % k]], pos=z["s%d_pos" % k], type=z["s%d_type" % k],
                        extra={int(p): (z["s%d_x%d_0" % (k, j)].tolist(), z["s%d_x%d_1" % (k, j)].tolist()) for j, p in enumerate(xpos)}))
    return out


def indel_scan_cases(haploid=False):
    z = np.load(os.path.join(GOLD, "indel_scan_hap.npz" if haploid else "indel_scan.npz"))
    out = []
    for k in range(int(z["n"])):
        out.append(dict(start=int(z["s%d_start" % k]), end=int(z["s%d_end" % k]), mincov=int(z["s%d_mincov" % k]),
                        win_size=int(z["s%d_win_size" % k]), small_win_size=int(z["s%d_small_win_size" % k]),
                        ins_t=float(z["s%d_ins_t" % k]), del_t=float(z["s%d_del_t" % k]),
                        exclude=[(int(a), int(b)) for a, b in z["s%d_excl" % k]], pos=z["s%d_pos" % k], type=z["s%d_type" % k]))
    return out
