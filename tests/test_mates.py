"""Alignments that share a read name (a split read's primary + supplementary records under dct['supplementary']): the reference keys a column's
pileup, the strand table and the neighbour lookups by NAME (generate_SNP_pileups.py:141-143,175,185,223,232).  Golden: the reference's own function
on the world "ont" with 31 alignments renamed into 14 shared names (tests/golden/snp_ont_mates.npz, oracle/tools/make_goldens.py: mates_world).
CPU: the host tables (pack.name_groups / mate_table, nc_decoded_name_groups); GPU: the featuriser against the golden and the oracle."""
import copy

import numpy as np
import pytest

import bamio
from util import load_snp_case


def _share_names(w, seed, n_groups=10, reach=6_000, primary=False):
    """some alignments of `w` renamed to another alignment's name and flagged supplementary (overlapping it, or within `reach`);
    primary=True: they stay primary records (what paired-end mates look like: kept by the default filter too)"""
    rng = np.random.default_rng(seed)
    w = copy.copy(w)
    names, flag = list(w.names), w.read_flag.copy()
    plain = [r for r in range(w.n_reads) if int(flag[r]) in (0, 16)]
    used, made = set(), 0
    for r1 in rng.permutation(plain).tolist():
        if made >= n_groups:
            break
        if r1 in used:
            continue
        cand = [r for r in plain if r != r1 and r not in used and int(w.read_start[r]) < int(w.read_end[r1]) + reach and int(w.read_start[r1]) < int(w.read_end[r]) + reach]
        if not cand:
            continue
        pick = [cand[int(rng.integers(0, len(cand)))]]
        if made % 3 == 2 and len(cand) > 1:                              # three records of one name
            pick.append([c for c in cand if c != pick[0]][0])
        used.add(r1)
        for r2 in pick:
            used.add(r2)
            names[r2] = names[r1]
            flag[r2] = (0 if primary else 0x800) | (16 if rng.random() < 0.5 else 0)
        made += 1
    w.names, w.read_flag = names, flag
    return w


def test_name_groups_rings_and_the_names_strand():
    from nanocaller_amd.pack import name_groups
    names = ["a", "b", "a", "c", "b", "a", "d", "c"]
    flag = np.array([0x800 | 16, 0, 0, 16, 0x800 | 16, 0x800, 0, 0x800], np.int32)
    keep = np.array([1, 1, 1, 1, 1, 1, 1, 0], np.uint8)                  # the second "c" is not kept: "c" is alone again
    nxt, strand = name_groups(names, flag, keep)
    assert nxt.tolist() == [2, 4, 5, -1, 1, 0, -1, -1]
    # "a": its primary is record 2 (forward) -> all forward; "b": primary 1 forward; the others keep their own bits
    assert strand.tolist() == [0, 0, 0, 1, 0, 0, 0, 0]
    # from the native decode's group ids instead of the names: the same
    gid = np.array([0, 1, 0, 3, 1, 0, -1, 3], np.int32)
    nxt2, strand2 = name_groups(None, flag, keep, gid)
    assert np.array_equal(nxt, nxt2) and np.array_equal(strand, strand2)
    assert name_groups(["x", "y"], np.zeros(2, np.int32), np.ones(2, np.uint8)) == (None, None)
    assert name_groups(None, flag, keep) == (None, None)


def test_mate_table_addresses_the_flagged_tile_entries():
    from nanocaller_amd.pack import pack_world
    from nanocaller_amd.wire import build_wire_from_world
    world = load_snp_case("ont_mates")[0]
    hp = pack_world(world, supplementary=True)
    key, rec = hp.mates
    assert key.size == 31 and np.all(np.diff(key) > 0)
    ent = hp.tile_ent
    flagged = ent[(ent["base_flag"] & 8) != 0]
    assert set(((flagged["base_flag"] & ~15) + (flagged["start"] & ~15)).tolist()) == set(key.tolist())
    # every ring closes on its own name, members ascend in file order
    for i in range(key.size):
        seen, j = [i], int(rec[i, 2])
        while j != i:
            seen.append(j)
            j = int(rec[j, 2])
        assert 2 <= len(seen) <= 3
    # the codes of a member are where the table says: codes[key - floor16(start) + p]
    kept = np.flatnonzero((world.read_flag & 0x704) == 0)
    by_start = {(int(world.read_start[r]), int(world.read_end[r])): r for r in kept}
    for i in range(key.size):
        s, e = int(rec[i, 0]), int(rec[i, 1])
        r = by_start[(s, e)]
        assert np.array_equal(hp.codes[key[i] - (s & ~15) + s:key[i] - (s & ~15) + e], world.read_codes(r))
    wp = build_wire_from_world(world, supplementary=True, pin=False)
    assert np.array_equal(wp.host("mate_key"), key) and np.array_equal(wp.host("mate_rec").reshape(-1, 4), rec)
    assert pack_world(world, supplementary=False).mates is None          # the renamed records are supplementary: filtered, no name is shared


def test_native_name_groups_equal_the_names(tmp_path):
    from nanocaller_amd import bam as nbam
    from nanocaller_amd.pack import name_groups, world_name_gid
    w = _share_names(bamio.make_bam_world(seed=11, length=40_000, depth=10), seed=3)
    path, fa = str(tmp_path / "m.bam"), str(tmp_path / "m.fa")
    bamio.write_bam(path, w.chrom, w.length, bamio.world_to_records(w, np.random.default_rng(1)))
    bamio.write_fasta(fa, w.chrom, w.ref)
    got = nbam.read_bam(path, fa, w.chrom)
    for supp in (False, True):
        keep = ((got.read_flag & (0x704 if supp else 0xF04)) == 0).astype(np.uint8)
        a = name_groups(list(got.names), got.read_flag, keep)
        b = name_groups(None, got.read_flag, keep, world_name_gid(got, supp))
        assert (a[0] is None) == (b[0] is None) == (not supp)
        if supp:
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and int((a[0] >= 0).sum()) >= 20


pytest_gpu = pytest.mark.gpu


@pytest_gpu
def test_shared_names_through_the_wire_and_the_int16_tensors_equal_the_reference_golden():
    import torch
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.wire import WireUploader, build_wire_from_world, upload_wire
    world, dct, region, exclude, gold = load_snp_case("ont_mates")
    eng = get_engine(0)
    was = getattr(eng, "x_int16", False)
    eng.set_tensor_format(int16=True)
    try:
        wp = build_wire_from_world(world, supplementary=True)
        up = WireUploader(eng)
        t = up.submit(wp)
        for dpk in (upload_wire(eng, wp), up.expand(t)):
            assert dpk.mates is not None
            sites = eng.snp_scan(dpk, [(region["start"], region["end"])], mincov=dct["mincov"], min_allele_freq=dct["min_allele_freq"], threshold=dct["threshold"])
            eng.snp_featurize(dpk, sites, seq=dct["seq"], maxcov=dct["maxcov"], min_nbr_sites=dct["min_nbr_sites"])
            torch.cuda.synchronize()
            assert np.array_equal(sites.pos, gold["pos"]) and np.array_equal(sites.dp, gold["dp"])
            assert sites.x.dtype == torch.int16 and np.array_equal(sites.x.cpu().numpy(), gold["mat"].astype(np.int16))
            assert np.array_equal(sites.fwd_dp.cpu().numpy(), gold["fwd_dp"]) and np.array_equal(sites.rev_dp.cpu().numpy(), gold["rev_dp"])
            assert float(np.mean(sites.depth.cpu().numpy().astype(np.float64))) == gold["depth"]
        up.release(t)
        # without the table the same pack gives the per-alignment answer: not the reference's
        dpk.mates = None
        sites = eng.snp_scan(dpk, [(region["start"], region["end"])], mincov=dct["mincov"], min_allele_freq=dct["min_allele_freq"], threshold=dct["threshold"])
        eng.snp_featurize(dpk, sites, seq=dct["seq"], maxcov=dct["maxcov"], min_nbr_sites=dct["min_nbr_sites"])
        assert not np.array_equal(sites.x.cpu().numpy(), gold["mat"].astype(np.int16))
    finally:
        eng.set_tensor_format(int16=was)


@pytest_gpu
@pytest.mark.parametrize("supplementary", [True, False])
def test_shared_names_from_a_bam_file_equal_the_oracle_and_the_caller_takes_them(tmp_path, supplementary):
    """BAM file -> native decode (nc_decoded_name_groups) -> wire -> featuriser, against the oracle keyed by the World's names; snpCaller.caller
    with dct['supplementary'] runs the same file (host route) instead of refusing it"""
    from nanocaller_amd import generate_SNP_pileups as gsp
    from oracle import oracle
    w = _share_names(bamio.make_bam_world(seed=21, length=60_000, depth=14), seed=5, n_groups=16, primary=not supplementary)
    path, fa = str(tmp_path / "m.bam"), str(tmp_path / "m.fa")
    bamio.write_bam(path, w.chrom, w.length, bamio.world_to_records(w, np.random.default_rng(2)))
    bamio.write_fasta(fa, w.chrom, w.ref)
    dct = dict(sam_path=path, fasta_path=fa, exclude_bed=None, threshold=[0.3, 0.7], supplementary=supplementary, mincov=4, maxcov=160, min_allele_freq=0.15,
               min_nbr_sites=1, seq="ont")
    region = dict(chrom=w.chrom, start=5_000, end=55_000, ploidy="diploid")
    gsp.release_contig()
    out = gsp.get_snp_testing_candidates(dct, region)
    exp = oracle.get_snp_testing_candidates(w, dct, region)
    assert len(out[0]) == len(exp[0]) > 50
    assert np.array_equal(out[0], exp[0]) and np.array_equal(np.asarray(out[2]), np.asarray(exp[2]))
    assert np.array_equal(out[6], exp[6]) and np.array_equal(out[7], exp[7]) and float(out[5]) == float(exp[5])
    # the caller's chunk loop (int16 tensors, the wire through the upload ring) keys them by name as well: the strand depths it reports are the oracle's
    from nanocaller_amd import snpCaller
    res = snpCaller.call_chunks(dict(dct, snp_model="ONT-HG002", disable_coverage_normalization=False), [region])
    assert res["n"] == len(exp[0]) and np.array_equal(res["pos"], exp[0])
    assert np.array_equal(res["fwd_dp"], np.asarray(exp[6], res["fwd_dp"].dtype)) and np.array_equal(res["rev_dp"], np.asarray(exp[7], res["rev_dp"].dtype))
    plain = copy.copy(w)
    plain.names = None
    per_alignment = oracle.get_snp_testing_candidates(plain, dct, region)
    assert not np.array_equal(np.asarray(per_alignment[2]), np.asarray(exp[2]))          # (the test is about something)
    # the indel route still refuses what it does not key by name
    from nanocaller_amd import _lib
    world = gsp._resolve(path, w.chrom, fa)
    if world.meta["unsupported"][supplementary][1]:
        with pytest.raises(_lib.NanoCallerHipError):
            gsp._check_supported(world, path, w.chrom, supplementary=supplementary)
    gsp._check_supported(world, path, w.chrom, supplementary=supplementary, by_name=True)
    gsp.release_contig()
