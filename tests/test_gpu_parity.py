"""Parity tests proper: the HIP path (through the C ABI) against the reference goldens and the CPU oracle.
Run on a real MI355X: python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest

from tests.util import GOLD, SNP_CASES, assert_tuple_matches_gold, load_snp_case, load_world

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from nanocaller_amd.engine import get_engine
    return get_engine(0)


def _dct(world, dct, exclude):
    d = dict(dct)
    d["sam_path"] = world
    d["fasta_path"] = None
    d["exclude_bed"] = exclude if exclude else None
    return d


@pytest.mark.parametrize("case", SNP_CASES)
def test_featuriser_matches_reference_golden(eng, case):
    """bit-exact: positions, reference one-hot, integer tensors, dp, float64 freq, depth, strand depths"""
    from nanocaller_amd.generate_SNP_pileups import get_snp_testing_candidates
    world, dct, region, exclude, gold = load_snp_case(case)
    out = get_snp_testing_candidates(_dct(world, dct, exclude), region)
    assert_tuple_matches_gold(out, gold)


@pytest.mark.parametrize("tile_size", [1024, 2048, 4096])
def test_scan_matches_oracle_all_tile_sizes(eng, tile_size):
    from nanocaller_amd.pack import pack_world
    from oracle import oracle
    world = load_world("ont")
    hp = pack_world(world, tile_size=tile_size)
    dpk = eng.upload(hp)
    rc = oracle.ref_codes_with_exclusions(world)
    start, end = 40_000, 95_000
    sites = eng.snp_scan(dpk, [(start, end)], mincov=4, min_allele_freq=0.15, threshold=[0.4, 0.6])
    nbr, cpos, cn, calt = oracle.snp_scan(world, rc, start, end, "diploid", 4, 0.15, [0.4, 0.6])
    assert np.array_equal(eng.fetch_nbr_sites(sites.n_nbr), nbr)
    assert np.array_equal(sites.pos, cpos) and np.array_equal(sites.dp, cn) and np.array_equal(sites.alt, calt)


def test_scan_outputs_regrow_when_a_scan_finds_more_than_the_last(eng):
    """nc_snp_scan sizes its outputs from the PREVIOUS scan of the context (one host round trip per scan instead of two) and
    repeats the compaction with larger buffers when they overflow: a fresh context scans a tiny range first, then the whole
    world, then the tiny range again -- each equal to the oracle."""
    from nanocaller_amd.engine import Engine
    from nanocaller_amd.pack import pack_world
    from oracle import oracle
    world = load_world("ont")
    e2 = Engine(0)                                            # its own context: no buffers from earlier scans
    dpk = e2.upload(pack_world(world))
    rc = oracle.ref_codes_with_exclusions(world)
    for start, end in ((60_000, 60_400), (5_000, 125_000), (60_000, 60_400), (5_000, 125_000)):
        sites = e2.snp_scan(dpk, [(start, end)], mincov=4, min_allele_freq=0.15, threshold=[0.4, 0.6])
        nbr, cpos, cn, calt = oracle.snp_scan(world, rc, start, end, "diploid", 4, 0.15, [0.4, 0.6])
        assert np.array_equal(e2.fetch_nbr_sites(sites.n_nbr), nbr)
        assert np.array_equal(sites.pos, cpos) and np.array_equal(sites.dp, cn) and np.array_equal(sites.alt, calt)


def test_multi_chunk_batch_matches_per_chunk_oracle(eng):
    """A batch of adjacent chunks in one launch == the oracle run chunk by chunk, including the duplicated
    boundary position (quirk E3) and the per-chunk coverage constant (quirk E2)."""
    from nanocaller_amd.pack import pack_world
    from nanocaller_amd.utils import get_chunks
    from oracle import oracle
    world = load_world("ont")
    dpk = eng.upload(pack_world(world))
    chunks = get_chunks([(world.chrom, 5_000, 125_000, "diploid")], cpu=7, max_chunk_size=20_000)
    assert len(chunks) >= 6
    # put two chunk boundaries exactly on known candidate positions (adjacent chunks share one coordinate, E3)
    gold_pos = load_snp_case("ont_dip")[4]["pos"]
    b1, b2 = int(gold_pos[len(gold_pos) // 3]), int(gold_pos[2 * len(gold_pos) // 3])
    chunks = [dict(chrom=world.chrom, start=a, end=b, ploidy="diploid")
              for a, b in ((5_000, 30_000), (30_000, b1), (b1, b2), (b2, 100_000), (100_000, 125_000))]
    sites = eng.snp_scan(dpk, [(c["start"], c["end"]) for c in chunks], mincov=4, min_allele_freq=0.15,
                         threshold=[0.4, 0.6])
    eng.snp_featurize(dpk, sites, seq="ont", maxcov=160)
    scale, chunk_depth = eng.snp_scale(sites, len(chunks), 48.0)
    x = sites.x.cpu().numpy()
    dct = dict(threshold=[0.4, 0.6], mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, seq="ont")
    off = 0
    ndup = 0
    for ci, c in enumerate(chunks):
        pos, ref, mat, dp, freq, depth, fwd, rev = oracle.get_snp_testing_candidates(world, dct, c)
        n = len(pos)
        sl = slice(off, off + n)
        assert np.array_equal(sites.pos[sl], pos) and np.all(sites.chunk[sl] == ci)
        assert np.array_equal(x[sl], mat)
        assert np.array_equal(sites.dp[sl], dp)
        assert chunk_depth[ci] == depth
        assert np.array_equal(scale.cpu().numpy()[sl], np.full(n, 48.0 / depth))
        if n and ci + 1 < len(chunks) and pos[-1] == chunks[ci + 1]["start"]:
            ndup += 1
        off += n
    assert off == sites.n_sites
    assert ndup >= 2, "test world should contain a candidate on a shared chunk boundary"


@pytest.mark.parametrize("maxcov", [50, 300])
def test_maxcov_policy_matches_oracle(eng, maxcov):
    """depth > maxcov: documented deterministic policy (first maxcov reads in coordinate order).  maxcov = 300 runs the
    featuriser instantiation with the 1024-entry read list (maxcov <= 256 uses the 256-entry one)."""
    from nanocaller_amd.generate_SNP_pileups import get_snp_testing_candidates
    from oracle import oracle
    world = load_world("deep")
    dct = dict(threshold=[0.4, 0.6], mincov=4, maxcov=maxcov, min_allele_freq=0.15, min_nbr_sites=1, seq="ont",
               supplementary=False, exclude_bed=None)
    region = dict(chrom=world.chrom, start=6_000, end=18_000, ploidy="diploid")
    a = get_snp_testing_candidates(_dct(world, dct, None), region)
    b = oracle.get_snp_testing_candidates(world, dct, region)
    assert len(a[0]) > 50 and max(a[3]) > 50
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.parametrize("maxcov", [255, 256])
def test_very_deep_pileup_counter_limits(eng, maxcov):
    """depth ~330 with maxcov 255 / 256: the two featuriser instantiations at their boundary (8-bit histogram fields hold
    exactly 255; 256 switches to 16-bit fields), and the scan's byte-lane counters widening every 255 reads"""
    from nanocaller_amd.generate_SNP_pileups import get_snp_testing_candidates
    from nanocaller_amd.synth import make_world
    from oracle import oracle
    world = make_world(seed=77, length=9_000, depth=330, tech="hifi", read_len_scale=0.2)
    dct = dict(threshold=[0.4, 0.6], mincov=4, maxcov=maxcov, min_allele_freq=0.15, min_nbr_sites=1, seq="pacbio",
               supplementary=False, exclude_bed=None)
    region = dict(chrom=world.chrom, start=2_000, end=7_000, ploidy="diploid")
    a = get_snp_testing_candidates(_dct(world, dct, None), region)
    b = oracle.get_snp_testing_candidates(world, dct, region)
    assert len(a[0]) > 3 and max(a[3]) > 256 and np.abs(a[2]).max() >= 200
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))


def _golden_inputs(case):
    _, _, _, _, gold = load_snp_case(case)
    ref_code = np.argmax(gold["ref"], 1).astype(np.int32)
    return gold["mat"], ref_code, gold["depth"]


@pytest.fixture(params=["fp16x3", "fp32"])
def precision(eng, request):
    """both trunk kernels: k5_trunk_h3 (fp16x3 split precision, the default) and k4_conv12 (exact fp32 MFMA)"""
    eng.set_cnn_precision(exact_fp32=(request.param == "fp32"))
    yield request.param
    eng.set_cnn_precision(exact_fp32=False)


@pytest.mark.parametrize("model,case", [("ONT-HG002", "ont_dip"), ("CCS-HG002", "hifi_pacbio_dip"), ("NanoCaller1", "deep_ont")])
def test_snp_cnn_matches_oracle(eng, model, case, precision):
    """per-site softmax probabilities within 1e-4 of the CPU restatement (float64 accumulate); measured ~1e-6"""
    import torch
    from nanocaller_amd import _lib
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle
    path, cov = get_SNP_model(model)
    w = Weights(path)
    eng.load_weights(_lib.MODEL_SNP, w)
    x, ref_code, depth = _golden_inputs(case)
    scale = np.full(len(x), cov / depth)
    for mode in (0, 1):
        probs, gt = eng.snp_forward(_lib.MODEL_SNP, torch.from_numpy(x).cuda(), torch.from_numpy(ref_code).cuda(),
                                    torch.from_numpy(scale).cuda(), scale_mode=mode)
        ep, eg = oracle.snp_forward(w.flat, x, ref_code, scale, scale_mode=mode, precision="f64")
        assert np.abs(probs.cpu().numpy() - ep).max() < 1e-4
        assert np.abs(gt.cpu().numpy() - eg).max() < 1e-4
        assert np.abs(probs.cpu().numpy() - ep).max() < 2e-5, "both trunk kernels should be far inside the 1e-4 contract"
    # genotype-relevant decisions identical
    assert np.array_equal(probs.cpu().numpy() >= 0.5, ep >= 0.5)


def test_every_snp_model_both_trunk_kernels(eng):
    """every shipped SNP model through both trunk kernels (the fp16x3 weight scale S is chosen per model at load time):
    each within 2e-5 of the float64 oracle, and within 1e-5 of each other"""
    import torch
    from nanocaller_amd import _lib
    from nanocaller_amd.weights import SNP_MODEL_FILES, Weights, get_SNP_model
    from oracle import oracle
    x, ref_code, depth = _golden_inputs("ont_dip")
    xd, rd = torch.from_numpy(x).cuda(), torch.from_numpy(ref_code).cuda()
    try:
        for model in sorted(m for m in SNP_MODEL_FILES if m != 'haploid'):
            path, cov = get_SNP_model(model)
            w = Weights(path)
            eng.load_weights(_lib.MODEL_SNP, w)
            scale = np.full(len(x), cov / depth)
            sd = torch.from_numpy(scale).cuda()
            ep, _ = oracle.snp_forward(w.flat, x, ref_code, scale, precision="f64")
            got = {}
            for exact in (False, True):
                eng.set_cnn_precision(exact_fp32=exact)
                got[exact] = eng.snp_forward(_lib.MODEL_SNP, xd, rd, sd)[0].cpu().numpy()
                assert np.abs(got[exact] - ep).max() < 2e-5, (model, exact)
            assert np.abs(got[False] - got[True]).max() < 1e-5, model
    finally:
        eng.set_cnn_precision(exact_fp32=False)


@pytest.mark.parametrize("n", [1, 2, 15, 63, 64, 65, 255, 256, 257, 511, 600])
def test_snp_cnn_odd_site_counts(eng, n, precision):
    """site counts around the persistent kernels' grid / tile boundaries (one 512-thread workgroup per CU, 64-site fc1
    tiles), and the asynchronous drain returning the same numbers as the plain call"""
    import torch
    from nanocaller_amd import _lib
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle
    path, cov = get_SNP_model("ONT-HG002")
    w = Weights(path)
    eng.load_weights(_lib.MODEL_SNP, w)
    x, ref_code, depth = _golden_inputs("ont_dip")
    reps = -(-n // len(x))
    x = np.concatenate([x] * reps)[:n]
    ref_code = np.concatenate([ref_code] * reps)[:n]
    scale = np.linspace(0.5, 2.0, n)
    xd, rd, sd = torch.from_numpy(x).cuda(), torch.from_numpy(ref_code).cuda(), torch.from_numpy(scale).cuda()
    probs, gt = eng.snp_forward(_lib.MODEL_SNP, xd, rd, sd)
    ep, eg = oracle.snp_forward(w.flat, x, ref_code, scale, precision="f64")
    assert np.abs(probs.cpu().numpy() - ep).max() < 2e-5 and np.abs(gt.cpu().numpy() - eg).max() < 2e-5
    _, _, hp, hg = eng.snp_forward(_lib.MODEL_SNP, xd, rd, sd, drain=True)
    eng.wait_copies()
    assert np.array_equal(hp, probs.cpu().numpy()) and np.array_equal(hg, gt.cpu().numpy())


@pytest.mark.parametrize("n", [1, 2, 3, 255, 257, 513, 1300])
def test_three_stage_trunk_equals_two_stage_trunk(eng, n, monkeypatch):
    """k5_trunk_p3 (the default: conv1 | conv2 | conv3 on three consecutive sites, one barrier per site) runs the same MFMA sequence per
    accumulator as k5_trunk_h3 (NC_TRUNK_P3=0): the probabilities are equal bit for bit, for site counts around the pipeline's fill / drain (1, 2, 3 sites
    per workgroup) and the grid's edges, float32 and int16 tensors"""
    import torch
    from nanocaller_amd import _lib
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle
    path, cov = get_SNP_model("ONT-HG002")
    w = Weights(path)
    eng.load_weights(_lib.MODEL_SNP, w)
    eng.set_cnn_precision(exact_fp32=False)
    x, ref_code, depth = _golden_inputs("ont_dip")
    reps = -(-n // len(x))
    x = np.concatenate([x] * reps)[:n]
    ref_code = np.concatenate([ref_code] * reps)[:n]
    scale = np.linspace(0.5, 2.0, n)
    rd, sd = torch.from_numpy(ref_code).cuda(), torch.from_numpy(scale).cuda()
    ep, _ = oracle.snp_forward(w.flat, x, ref_code, scale, precision="f64")
    try:
        for i16 in (False, True):
            xd = torch.from_numpy(x).cuda()
            if i16:
                assert np.array_equal(x, np.rint(x)) and np.abs(x).max() < 32768
                xd = xd.to(torch.int16)
            eng.set_tensor_format(int16=i16)
            got = {}
            monkeypatch.setenv("NC_TRUNK_LIN", "0")                      # (int16 tensors would otherwise take k5_trunk_lin: its own test below)
            for p3 in ("0", "1"):
                monkeypatch.setenv("NC_TRUNK_P3", p3)
                got[p3] = eng.snp_forward(_lib.MODEL_SNP, xd, rd, sd)[0].cpu().numpy()
            assert np.array_equal(got["0"], got["1"]), (n, i16)
            assert np.abs(got["1"] - ep).max() < 2e-5
    finally:
        eng.set_tensor_format(int16=False)


@pytest.mark.parametrize("n", [1, 2, 3, 255, 257, 513, 1300])
def test_linear_conv1_trunk_equals_the_oracle_and_the_pixel_form(eng, n, monkeypatch):
    """k5_trunk_lin (int16 tensors, the product path: conv1 on the integer entries with the coverage scale applied to the accumulators, row 0 and
    channel 4 as u / s in fp16 hi + lo) against the float64 oracle and against k5_trunk_p3 (which multiplies the scale into the operand as the
    reference does, snpCaller.py:93-96): real site tensors and arbitrary int16 ones (negative entries, row 0 / channel 4 not 0 / 1, zero columns),
    scales from 0.05 to 20, both scale modes, no scale array at all; site counts around the pipeline's fill / drain and the grid's edges"""
    import torch
    from nanocaller_amd import _lib
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle
    path, cov = get_SNP_model("ONT-HG002")
    w = Weights(path)
    eng.load_weights(_lib.MODEL_SNP, w)
    eng.set_cnn_precision(exact_fp32=False)
    x, ref_code, depth = _golden_inputs("ont_dip")
    reps = -(-n // len(x))
    x = np.concatenate([x] * reps)[:n].copy()
    ref_code = np.concatenate([ref_code] * reps)[:n]
    rng = np.random.Generator(np.random.PCG64(n))
    for k in range(1, n, 2):                                               # every second site: arbitrary small integers everywhere
        x[k] = rng.integers(-25, 26, size=x[k].shape)
        x[k][:, rng.integers(0, 41, size=6), :] = 0
    scale = np.exp(rng.uniform(np.log(0.05), np.log(20.0), n))
    scale[0] = 1.0
    x_lim = eng.x_limit(_lib.MODEL_SNP)
    for k in range(n):                                                     # keep every site inside the model's proven fp16 range (the guard has its own test)
        m = np.abs(x[k][1:, :, :4]).max() * scale[k]
        if m > 0.9 * x_lim:
            scale[k] *= 0.9 * x_lim / m
    rd = torch.from_numpy(ref_code).cuda()
    xd16 = torch.from_numpy(x.astype(np.int16)).cuda()
    xd32 = torch.from_numpy(x.astype(np.float32)).cuda()
    try:
        for mode in (0, 1):
            for with_scale in (True, False):
                sc = scale if with_scale else np.ones(n)
                sd = torch.from_numpy(sc).cuda()
                ep, eg = oracle.snp_forward(w.flat, x.astype(np.float32), ref_code, sc, scale_mode=mode, precision="f64")
                eng.set_tensor_format(int16=True)
                assert eng.trunk_info()[1] == "k5_trunk_lin"
                flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
                lin_p, lin_g = eng.snp_forward(_lib.MODEL_SNP, xd16, rd, sd if with_scale else None, scale_mode=mode, range_flags=flags)[:2]
                lin_p, lin_g = lin_p.cpu().numpy(), lin_g.cpu().numpy()
                assert int(flags.sum().item()) == 0
                eng.set_tensor_format(int16=False)
                assert eng.trunk_info()[1] == "k5_trunk_p3"
                pix_p = eng.snp_forward(_lib.MODEL_SNP, xd32, rd, sd if with_scale else None, scale_mode=mode)[0].cpu().numpy()
                assert np.abs(lin_p - ep).max() < 2e-5 and np.abs(lin_g - eg).max() < 2e-5, (n, mode, with_scale, np.abs(lin_p - ep).max())
                assert np.abs(lin_p - pix_p).max() < 2e-5, (n, mode, with_scale)
    finally:
        eng.set_tensor_format(int16=False)


def test_linear_conv1_trunk_flags_what_it_cannot_take(eng):
    """the range guard of k5_trunk_lin: a scaled entry beyond the model's proven fp16 range, an unscaled one beyond it, and an integer beyond fp16's
    exact range (2048) mark their site (nc_cnn_range_watch) -- snpCaller.call_chunks re-runs exactly those on the exact fp32 trunk -- and nothing else"""
    import torch
    from nanocaller_amd import _lib
    from nanocaller_amd.weights import Weights, get_SNP_model
    w = Weights(get_SNP_model("ONT-HG002")[0])
    eng.load_weights(_lib.MODEL_SNP, w)
    eng.set_cnn_precision(exact_fp32=False)
    x_lim = eng.x_limit(_lib.MODEL_SNP)
    n = 600
    x = np.zeros((n, 5, 41, 5), np.int16)
    x[:, 0, :, 0] = 1
    x[:, 1, :, 4] = 1
    x[:, 1:, :, :4] = 3
    scale = np.full(n, 1.5)
    x[7, 2, 5, 1] = int(x_lim / 1.5) + 2                   # scaled entry over the limit
    x[300, 0, 40, 3] = int(x_lim) + 2                      # unscaled (row 0) entry over the limit
    x[301, 4, 0, 4] = -(int(x_lim) + 2)                    # unscaled (channel 4) entry over the limit
    x[599, 4, 40, 2] = 2049                                # not exact in fp16
    scale[599] = 1e-3
    x[100, 3, 3, 3] = int(x_lim / 1.5) - 1                 # just inside
    x[101, 4, 40, 0] = 2048
    scale[101] = 1e-3
    try:
        eng.set_tensor_format(int16=True)
        flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
        eng.snp_forward(_lib.MODEL_SNP, torch.from_numpy(x).cuda(), torch.zeros(n, dtype=torch.int32, device="cuda"), torch.from_numpy(scale).cuda(), range_flags=flags)
        assert sorted(torch.nonzero(flags).flatten().tolist()) == [7, 300, 301, 599]
    finally:
        eng.set_tensor_format(int16=False)


def test_snp_hap_cnn_matches_oracle(eng, precision):
    import torch
    from nanocaller_amd import _lib
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle
    path, _ = get_SNP_model("haploid")
    w = Weights(path)
    eng.load_weights(_lib.MODEL_SNP_HAP, w)
    x, ref_code, depth = _golden_inputs("ont_hap")
    scale = np.full(len(x), 30.0 / depth)
    probs, _ = eng.snp_forward(_lib.MODEL_SNP_HAP, torch.from_numpy(x).cuda(), torch.from_numpy(ref_code).cuda(),
                               torch.from_numpy(scale).cuda())
    ep = oracle.snp_hap_forward(w.flat, x, ref_code, scale, precision="f64")
    assert np.abs(probs.cpu().numpy() - ep).max() < 2e-5
    assert np.array_equal(np.argmax(probs.cpu().numpy(), 1), np.argmax(ep, 1))


def test_indel_tensor_and_cnn_match_oracle(eng):
    import torch
    from nanocaller_amd import _lib
    from nanocaller_amd.weights import Weights, get_indel_model
    from oracle import oracle
    z = np.load(os.path.join(GOLD, "indel_msa.npz"))
    n = int(z["n"])
    rows = [z["m%d_rows" % k] for k in range(n)]
    refs = [z["m%d_ref" % k] for k in range(n)]
    x, cns = eng.indel_tensor(rows, refs)
    xh = x.cpu().numpy()
    sym = "AGTC-"
    for k in range(n):
        assert np.array_equal(xh[k].astype(np.float64), z["m%d_mat" % k])          # reference msa() golden, exact
        assert "".join(sym[c] for c in cns[k]) == str(z["m%d_cns" % k])
    # diploid CNN on 15x128x2 (three stacked read sets), haploid on 5x128x2
    rng = np.random.Generator(np.random.PCG64(3))
    x15 = np.concatenate([xh[rng.integers(0, n, size=6)] for _ in range(3)], axis=1).astype(np.float32)
    for name, kind, xin in (("ONT-HG002", _lib.MODEL_INDEL, x15), ("haploid", _lib.MODEL_INDEL_HAP, xh)):
        w = Weights(get_indel_model(name))
        eng.load_weights(kind, w)
        e = oracle.indel_forward(w.flat, xin, precision="f64")
        for exact in (False, True):                                   # split-precision (default) and exact fp32 conv2 / conv3
            eng.set_cnn_precision(exact_fp32=exact)
            p = eng.indel_forward(kind, torch.from_numpy(np.ascontiguousarray(xin)).cuda()).cpu().numpy()
            assert np.abs(p - e).max() < 1e-4, (name, exact)
        eng.set_cnn_precision(exact_fp32=False)


@pytest.mark.parametrize("n", [1, 17, 333])
def test_indel_cnn_split_precision_on_dense_inputs(eng, n):
    """k8_conv23_h3 (fp16 hi/lo planes, three MFMA products per fp32 product) against the float64 oracle and the exact
    fp32 kernels on dense random tensors (every tap and channel exercised, site counts that leave partial tiles)"""
    import torch
    from nanocaller_amd import _lib
    from nanocaller_amd.weights import Weights, get_indel_model
    from oracle import oracle
    rng = np.random.Generator(np.random.PCG64(100 + n))
    for name, kind, rows in (("ONT-HG002", _lib.MODEL_INDEL, 15), ("haploid", _lib.MODEL_INDEL_HAP, 5)):
        w = Weights(get_indel_model(name))
        eng.load_weights(kind, w)
        x = (rng.random((n, rows, 128, 2)) * 1.6 - 0.5).astype(np.float32)
        x[rng.random(x.shape) < 0.3] = 0.0
        e = oracle.indel_forward(w.flat, x, precision="f64")
        xd = torch.from_numpy(x).cuda()
        eng.set_cnn_precision(exact_fp32=False)
        p = eng.indel_forward(kind, xd).cpu().numpy()
        eng.set_cnn_precision(exact_fp32=True)
        q = eng.indel_forward(kind, xd).cpu().numpy()
        eng.set_cnn_precision(exact_fp32=False)
        assert np.abs(q - e).max() < 1e-4 and np.abs(p - e).max() < 1e-4, (name, float(np.abs(p - e).max()))
        assert np.abs(p - q).max() < 2e-5                              # the two kernel families agree far inside the tolerance


def test_end_to_end_vcf_matches_oracle_pipeline(eng, tmp_path):
    """call_manager on the GPU == oracle featuriser + oracle CNN + the same host rules: identical positions
    and genotypes, probabilities within 1e-4."""
    from nanocaller_amd import snpCaller
    from nanocaller_amd.utils import get_chunks
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle
    world = load_world("ont")
    regions = [(world.chrom, 30_000, 110_000, "diploid")]
    chunks = get_chunks(regions, cpu=3)
    params = dict(chunks_list=chunks, regions_list=regions, sam_path=world, fasta_path=None, mincov=4, maxcov=160,
                  min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6], snp_model="ONT-HG002", cpu=1,
                  vcf_path=str(tmp_path), prefix="t", sample="SAMPLE", seq="ont", supplementary=False, exclude_bed=None,
                  suppress_progress=True, disable_coverage_normalization=False)
    out = snpCaller.call_manager(params)
    assert out.endswith("t.snps.vcf.gz") and os.path.exists(out)
    import gzip
    got = [ln for ln in gzip.open(os.path.join(str(tmp_path), "t.unfiltered.snps.vcf.gz"), "rt") if not ln.startswith("#")]
    path, cov = get_SNP_model("ONT-HG002")
    w = Weights(path)
    dct = dict(threshold=[0.4, 0.6], mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, seq="ont")
    exp = []
    for c in chunks:
        pos, ref, mat, dp, freq, depth, fwd, rev = oracle.get_snp_testing_candidates(world, dct, c)
        rc = np.argmax(ref, 1).astype(np.int32)
        probs, _ = oracle.snp_forward(w.flat, mat, rc, np.full(len(pos), cov / depth), precision="f32")
        exp += snpCaller.snp_vcf_lines(world.chrom, pos, rc, probs, dp, freq, fwd, rev)
    exp.sort(key=lambda ln: int(ln.split("\t")[1]))
    assert len(got) == len(exp) and len(got) > 100
    ngt = 0
    for g, e in zip(got, exp):
        gf, ef = g.rstrip("\n").split("\t"), e.rstrip("\n").split("\t")
        assert gf[:5] == ef[:5] and gf[6] == ef[6]                      # CHROM POS ID REF ALT FILTER
        assert gf[9].split(":")[0] == ef[9].split(":")[0]                # GT
        assert gf[9] == ef[9]                                            # depths / VF
        gp = [float(v) for v in gf[7].split(";")[0][3:].split(",")]
        ep = [float(v) for v in ef[7].split(";")[0][3:].split(",")]
        assert max(abs(a - b) for a, b in zip(gp, ep)) <= 1.01e-4
        ngt += gf[6] == "PASS"
    assert ngt > 10


def test_int16_tensors_equal_float32_tensors(eng, monkeypatch):
    """nc_set_tensor_format(ctx, 1): the featuriser's int16 tensors are the float32 tensors value for value; the pixel-form split-precision trunk
    (k5_trunk_p3, NC_TRUNK_LIN=0) returns bit-identical probabilities from either, k5_trunk_lin (the int16 default: conv1 by linearity, other
    roundings) the same within 2e-5; the exact-fp32 trunk refuses int16"""
    import torch
    from nanocaller_amd import _lib
    from nanocaller_amd.pack import pack_world
    from nanocaller_amd.weights import Weights, get_SNP_model
    world = load_world("ont")
    dp = eng.upload(pack_world(world))
    path, cov = get_SNP_model("ONT-HG002")
    eng.load_weights(_lib.MODEL_SNP, Weights(path))
    out = {}
    monkeypatch.setenv("NC_TRUNK_LIN", "0")
    for i16 in (False, True):
        eng.set_tensor_format(int16=i16)
        sites = eng.snp_scan(dp, [(20_000, 60_000), (60_001, 100_003)], mincov=4, min_allele_freq=0.15, threshold=[0.4, 0.6])
        eng.snp_featurize(dp, sites, seq="ont", maxcov=160)
        scale, _ = eng.snp_scale(sites, 2, cov)
        probs, gt = eng.snp_forward(_lib.MODEL_SNP, sites.x, sites.ref_code, scale)
        out[i16] = (sites.x.clone(), probs.clone(), gt.clone(), sites.n_sites)
    assert out[True][0].dtype == torch.int16 and out[False][0].dtype == torch.float32 and out[True][3] == out[False][3] > 300
    assert torch.equal(out[True][0].to(torch.float32), out[False][0])
    assert torch.equal(out[True][1], out[False][1]) and torch.equal(out[True][2], out[False][2])
    monkeypatch.delenv("NC_TRUNK_LIN")
    eng.set_tensor_format(int16=True)
    assert eng.trunk_info()[1] == "k5_trunk_lin"
    lp, lg = eng.snp_forward(_lib.MODEL_SNP, out[True][0], sites.ref_code, scale)
    assert float((lp - out[False][1]).abs().max()) < 2e-5 and float((lg - out[False][2]).abs().max()) < 2e-5
    eng.set_cnn_precision(exact_fp32=True)
    with pytest.raises(Exception):
        eng.snp_forward(_lib.MODEL_SNP, out[True][0], sites.ref_code, scale)
    eng.set_cnn_precision(exact_fp32=False)
    eng.set_tensor_format(int16=False)


def test_deferred_calls_equal_drained_calls(eng):
    """call_chunks(defer=True): several groups enqueued back to back (the next group's scan queued behind the previous
    group's CNN, results collected afterwards, in any order) give bit-identical arrays to one drained call per group;
    the HIP-event totals cover every call."""
    from nanocaller_amd import snpCaller
    from nanocaller_amd.utils import get_chunks
    world = load_world("ont")
    params = dict(sam_path=world, fasta_path=None, mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6],
                  snp_model="ONT-HG002", seq="ont", supplementary=False, exclude_bed=None, disable_coverage_normalization=False)
    groups = [get_chunks([(world.chrom, a, b, "diploid")], cpu=2) for (a, b) in ((20_000, 70_000), (60_000, 118_000), (1, 9_000), (119_000, 119_400))]
    groups.append([dict(chrom=world.chrom, start=world.length + 10, end=world.length + 500, ploidy="diploid")])      # no sites at all
    drained = [snpCaller.call_chunks(params, g) for g in groups]
    eng.enable_timing(True, trunk_only=True)
    pend = [snpCaller.call_chunks(params, g, defer=True) for g in groups]
    got = [p.result() for p in reversed(pend)][::-1]
    sums, cnt = eng.timing_sums()
    eng.enable_timing(False)
    assert sum(r["n"] for r in drained) > 500 and drained[-1]["n"] == 0
    for a, b in zip(drained, got):
        assert a["n"] == b["n"]
        for key in ("pos", "chunk", "ref", "probs", "gt", "dp", "alt", "fwd_dp", "rev_dp", "freq", "chunk_depth"):
            if a["n"]:
                assert np.array_equal(a[key], b[key]), key
    n_calls = sum(1 for r in drained if r["n"])
    assert cnt[4] == n_calls and sums[5] == n_calls and sums[4] > 0.0      # one trunk launch per (small) call, all folded in
    assert pend[0].result() is got[0]                                        # result() is idempotent


def test_one_queue_and_the_library_on_its_own_stream_give_the_same_results(eng, monkeypatch):
    """Engine.use_torch_stream: a thread on torch's default stream is moved to the engine's own torch stream (torch's kernels and the library's on
    one hardware queue); NC_ONE_QUEUE=0 + the default stream is the arrangement of rounds 1-4 (the library on a stream of its own, ordered by the
    default stream's implicit synchronisation).  Pipelined calls give bit-identical arrays either way."""
    import torch
    from nanocaller_amd import snpCaller
    from nanocaller_amd.utils import get_chunks
    world = load_world("ont")
    params = dict(sam_path=world, fasta_path=None, mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6],
                  snp_model="ONT-HG002", seq="ont", supplementary=False, exclude_bed=None, disable_coverage_normalization=False)
    groups = [get_chunks([(world.chrom, a, b, "diploid")], cpu=2) for (a, b) in ((20_000, 70_000), (60_000, 118_000), (1, 9_000))]

    def run():
        pend = [snpCaller.call_chunks(params, g, defer=True) for g in groups]
        return [p.result() for p in pend]
    eng.use_torch_stream()
    assert torch.cuda.current_stream(eng.device).cuda_stream != 0, "the thread should be on the engine's stream"
    one = run()
    try:
        monkeypatch.setenv("NC_ONE_QUEUE", "0")
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream(eng.device))
        two = run()
        torch.cuda.synchronize()
    finally:
        monkeypatch.delenv("NC_ONE_QUEUE")
        eng.use_torch_stream()
    assert torch.cuda.current_stream(eng.device).cuda_stream != 0
    assert sum(r["n"] for r in one) > 400
    for a, b in zip(one, two):
        assert a["n"] == b["n"]
        for key in ("pos", "chunk", "ref", "probs", "gt", "dp", "alt", "fwd_dp", "rev_dp", "freq", "chunk_depth"):
            assert np.array_equal(a[key], b[key]), key


def test_indel_window_scan_matches_reference_pass1(eng):
    """K7: the `variants` dict of the reference's pass 1 (captured from its own frame) and the CPU oracle"""
    from nanocaller_amd.generate_indel_pileups import scan_indel_candidates
    from oracle import oracle
    from tests.util import indel_scan_cases
    world = load_world("indel")
    n = 0
    for c in indel_scan_cases():
        dct = dict(mincov=c["mincov"], win_size=c["win_size"], small_win_size=c["small_win_size"], ins_t=c["ins_t"],
                   del_t=c["del_t"], supplementary=False, impute_indel_phase=False,
                   exclude_bed=[(world.chrom, a, b) for a, b in c["exclude"]] or None)
        got = scan_indel_candidates(dct, dict(chrom=world.chrom, start=c["start"], end=c["end"], sam_path=world))
        assert sorted(got) == c["pos"].tolist() and [got[p] for p in sorted(got)] == c["type"].tolist(), (c["start"], c["end"])
        op, ot = oracle.indel_scan(world, c["start"], c["end"], mincov=c["mincov"], win_size=c["win_size"],
                                   small_win_size=c["small_win_size"], ins_t=c["ins_t"], del_t=c["del_t"], exclude=c["exclude"])
        assert sorted(got) == op.tolist()
        n += len(got)
    assert n > 100


def test_haploid_indel_window_scan_matches_reference_pass1(eng):
    """K7 in haploid mode (generate_indel_pileups_haploid.py:185-241) against the reference's captured `variants`"""
    from nanocaller_amd.generate_indel_pileups import scan_indel_candidates
    from tests.util import indel_scan_cases
    world = load_world("indel")
    n = 0
    for c in indel_scan_cases(haploid=True):
        dct = dict(mincov=c["mincov"], win_size=c["win_size"], small_win_size=c["small_win_size"], ins_t=c["ins_t"],
                   del_t=c["del_t"], supplementary=False, impute_indel_phase=False,
                   exclude_bed=[(world.chrom, a, b) for a, b in c["exclude"]] or None)
        got = scan_indel_candidates(dct, dict(chrom=world.chrom, start=c["start"], end=c["end"], sam_path=world), haploid=True)
        assert sorted(got) == c["pos"].tolist() and [got[p] for p in sorted(got)] == c["type"].tolist(), (c["start"], c["end"])
        n += len(got)
    assert n > 40


@pytest.mark.parametrize("haploid", [False, True])
def test_indel_scan_batch_equals_per_chunk_calls(eng, haploid):
    """nc_indel_scan_batch keeps the per-chunk semantics (fresh window state at every chunk start): identical per-column
    decisions to one nc_indel_scan call per chunk, for ragged chunk lengths incl. one-column chunks"""
    import torch
    from nanocaller_amd.pack import pack_world
    world = load_world("indel")
    dp = eng.upload(pack_world(world))
    chunks = [(1, 9_000), (9_000, 9_000), (9_001, 23_456), (23_400, 41_000), (41_001, world.length), (58_000, 58_001)] + \
             [(a, a + 777) for a in range(2_000, 57_000, 911)]
    kw = dict(mincov=4, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, haploid=haploid)
    excl = torch.zeros(dp.n_tiles * dp.tile_size, dtype=torch.uint8, device="cuda")
    excl[30_000 - dp.tile_pos0:31_500 - dp.tile_pos0] = 1
    # ascending lists run with the chunk as a grid dimension (overlapping, abutting and one-column chunks, chunks past the
    # last read); any other order takes the chunk-by-chunk route
    ascending = sorted([(a, a + 777) for a in range(2_000, 57_000, 911)] + [(a + 400, a + 1_200) for a in range(2_000, 57_000, 1_822)])
    ascending = [(1, 1_500), (1_500, 1_500), (1_501, 2_776)] + ascending + [(58_500, 58_500), (58_501, world.length),
                                                                         (world.length + 5_000, world.length + 6_000)]
    assert all(b[0] >= a[0] and b[1] >= a[1] for a, b in zip(ascending, ascending[1:])) and len(ascending) > 90
    n = 0
    for lst, ex in ((ascending, None), (ascending, excl), (chunks, None)):
        got = eng.indel_scan_batch(dp, lst, excl=ex, **kw)
        assert len(got) == len(lst)
        for (a, b), g in zip(lst, got):
            one = eng.indel_scan(dp, a, b, excl=ex, **kw)
            assert np.array_equal(g, one), (a, b)
            n += int((one >= 0).sum())
    assert n > 300


def test_indel_scan_chunk_list_matches_reference_pass1(eng):
    """the chunk-list form of scan_indel_candidates (one set of launches for all chunks) against the reference's captured
    `variants` of every chunk"""
    from nanocaller_amd.generate_indel_pileups import scan_indel_candidates
    from tests.util import indel_scan_cases
    world = load_world("indel")
    for haploid in (False, True):
        groups = {}
        for c in indel_scan_cases(haploid=haploid):
            key = (c["mincov"], c["win_size"], c["small_win_size"], c["ins_t"], c["del_t"], tuple(map(tuple, c["exclude"])))
            groups.setdefault(key, []).append(c)
        n = 0
        for key, cases in groups.items():
            cases.sort(key=lambda c: (c["start"], c["end"]))
            dct = dict(mincov=key[0], win_size=key[1], small_win_size=key[2], ins_t=key[3], del_t=key[4], supplementary=False,
                       impute_indel_phase=False, exclude_bed=[(world.chrom, a, b) for a, b in key[5]] or None)
            got = scan_indel_candidates(dct, [dict(chrom=world.chrom, start=c["start"], end=c["end"], sam_path=world) for c in cases], haploid=haploid)
            for g, c in zip(got, cases):
                assert sorted(g) == c["pos"].tolist() and [g[p] for p in sorted(g)] == c["type"].tolist(), (c["start"], c["end"])
                n += len(g)
        assert n > 40


def _names_to_idx(sets):
    return {p: (sorted(int(n[1:]) for n in a), sorted(int(n[1:]) for n in b)) for p, (a, b) in sets.items()}


def test_impute_indel_phase_scan_matches_reference_pass1(eng):
    """K7 with dct['impute_indel_phase'] (generate_indel_pileups.py:278-304): `variants` and `extra_variants` captured from
    the reference's own frame -- one call per chunk, and all chunks of equal parameters as one chunk list"""
    from nanocaller_amd.generate_indel_pileups import scan_indel_candidates
    from tests.util import indel_impute_cases, load_impute_world
    world = load_impute_world()
    cases = indel_impute_cases()
    groups = {}
    n = 0
    for c in cases:
        dct = dict(mincov=c["mincov"], win_size=c["win_size"], small_win_size=c["small_win_size"], ins_t=c["ins_t"],
                   del_t=c["del_t"], supplementary=False, impute_indel_phase=True,
                   exclude_bed=[(world.chrom, a, b) for a, b in c["exclude"]] or None)
        extra = {}
        got = scan_indel_candidates(dct, dict(chrom=world.chrom, start=c["start"], end=c["end"], sam_path=world), extra_variants=extra)
        assert sorted(got) == c["pos"].tolist() and [got[p] for p in sorted(got)] == c["type"].tolist(), (c["start"], c["end"])
        assert _names_to_idx(extra) == c["extra"], (c["start"], c["end"])
        n += len(extra)
        key = (c["mincov"], c["win_size"], c["small_win_size"], c["ins_t"], c["del_t"], tuple(map(tuple, c["exclude"])))
        groups.setdefault(key, (dct, []))[1].append(c)
    assert n > 300
    for dct, cs in groups.values():
        cs.sort(key=lambda c: (c["start"], c["end"]))
        extras = [{} for _ in cs]
        got = scan_indel_candidates(dct, [dict(chrom=world.chrom, start=c["start"], end=c["end"], sam_path=world) for c in cs], extra_variants=extras)
        for g, x, c in zip(got, extras, cs):
            assert sorted(g) == c["pos"].tolist() and [g[p] for p in sorted(g)] == c["type"].tolist() and _names_to_idx(x) == c["extra"]
    # the flag off: the plain rule, nothing imputed, on the same unphased world
    from oracle import oracle
    dct = dict(mincov=4, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.4, supplementary=False, impute_indel_phase=False, exclude_bed=None)
    got = scan_indel_candidates(dct, dict(chrom=world.chrom, start=5_000, end=59_000, sam_path=world))
    op, ot = oracle.indel_scan(world, 5_000, 59_000, mincov=4, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.4)
    assert sorted(got) == op.tolist() and [got[p] for p in sorted(got)] == ot.tolist()


@pytest.mark.parametrize("seed", [5, 6])
def test_impute_indel_phase_scan_matches_oracle_on_other_worlds(eng, seed):
    """seeded worlds the goldens do not hold (other depth / read lengths / thresholds) against the Python restatement"""
    from nanocaller_amd.generate_indel_pileups import scan_indel_candidates
    from nanocaller_amd.synth import add_indels, make_world, unphase_blocks
    from oracle import oracle
    w = add_indels(make_world(seed=seed, length=30_000, depth=18 + 10 * (seed % 2), tech="ont", read_len_scale=0.3, odd_flag_frac=0.04), seed=seed)
    w = unphase_blocks(w, [(2_000, 11_000), (17_000, 26_000)], seed=seed, drop=0.7 + 0.05 * (seed % 2))
    n = 0
    for (a, b, kw) in [(1, 30_000, dict(mincov=4, ins_t=0.4, del_t=0.4)), (1_500, 27_000, dict(mincov=3, ins_t=0.25, del_t=0.3)),
                       (9_000, 20_000, dict(mincov=5, ins_t=0.1, del_t=0.1))]:
        dct = dict(win_size=40, small_win_size=4, supplementary=False, impute_indel_phase=True, exclude_bed=[(w.chrom, 5_000, 5_600)], **kw)
        extra = {}
        got = scan_indel_candidates(dct, dict(chrom=w.chrom, start=a, end=b, sam_path=w), extra_variants=extra)
        idx = {nm: i for i, nm in enumerate(w.names)}
        ev, ex = oracle.indel_scan_impute(w, a, b, win_size=40, small_win_size=4, exclude=[(5_000, 5_600)], **kw)
        assert got == ev, (a, b)
        assert {p: (sorted(idx[q] for q in s0), sorted(idx[q] for q in s1)) for p, (s0, s1) in extra.items()} == ex
        n += len(ex)
    assert n > 30


def test_bam_and_fasta_files_end_to_end(eng, tmp_path):
    """real files in (BAM + BAI + FASTA + bgzipped BED), VCF out: identical to the run on the in-memory world"""
    import gzip
    from nanocaller_amd import snpCaller
    from nanocaller_amd.utils import get_chunks
    from tests import bamio
    world = bamio.make_bam_world(seed=9, length=40_000, depth=22)
    rng = np.random.Generator(np.random.PCG64(2))
    bam, fa, bed = str(tmp_path / "s.bam"), str(tmp_path / "r.fa"), str(tmp_path / "x.bed.gz")
    bamio.write_bam(bam, world.chrom, world.length, bamio.world_to_records(world, rng))
    bamio.write_fasta(fa, world.chrom, world.ref)
    snpCaller.bgzf_write(bed, b"chrZ\t1\t50\n%s\t12000\t12800\n" % world.chrom.encode())
    regions = [(world.chrom, 2_000, 38_000, "diploid")]
    outs = []
    for tag, sam, ex in (("files", bam, bed), ("world", world, [(world.chrom, 12_000, 12_800)])):
        vdir = tmp_path / tag
        vdir.mkdir()
        params = dict(chunks_list=get_chunks(regions, cpu=3), regions_list=regions, sam_path=sam, fasta_path=fa, mincov=4,
                      maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6], snp_model="ONT-HG002", cpu=1,
                      vcf_path=str(vdir), prefix="t", sample="S", seq="ont", supplementary=False, exclude_bed=ex,
                      suppress_progress=True, disable_coverage_normalization=False)
        snpCaller.call_manager(params)
        outs.append(gzip.open(str(vdir / "t.unfiltered.snps.vcf.gz"), "rt").read())
    assert outs[0] == outs[1] and outs[0].count("\n") > 100
    assert not any(12_000 <= int(ln.split("\t")[1]) < 12_800 for ln in outs[0].splitlines() if not ln.startswith("#"))


def test_degenerate_inputs(eng):
    """empty / ragged inputs: no reads at all, reads but no candidate, thresholds nothing passes, a chunk past the last read,
    zero-site CNN calls -- the reference's empty-lists contract (generate_SNP_pileups.py:193-197) and no device faults"""
    import torch
    from nanocaller_amd import _lib, snpCaller
    from nanocaller_amd.generate_SNP_pileups import get_snp_testing_candidates
    from nanocaller_amd.synth import make_world
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle
    base = dict(threshold=[0.4, 0.6], mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, seq="ont", supplementary=False,
                exclude_bed=None)
    empty = ([], [], [], [], [], 0, [], [])
    # (1) a contig with (almost) no reads
    w0 = make_world(seed=1, length=5000, depth=0.001, read_len_scale=0.05)
    assert get_snp_testing_candidates(_dct(w0, base, None), dict(chrom=w0.chrom, start=1, end=5000, ploidy="diploid")) == empty
    # (2) normal depth, but nothing can pass: allele frequency above 1, then mincov above the depth
    w = load_world("ont")
    for over in (dict(min_allele_freq=1.5), dict(mincov=10_000)):
        d = dict(base, **over)
        reg = dict(chrom=w.chrom, start=2_000, end=9_000, ploidy="diploid")
        got = get_snp_testing_candidates(_dct(w, d, None), reg)
        exp = oracle.get_snp_testing_candidates(w, d, reg)
        assert got == empty and len(exp[0]) == 0
    # (3) a chunk that starts after the last read / one-column chunks at the contig ends
    for reg in (dict(chrom=w.chrom, start=w.length - 3, end=w.length, ploidy="diploid"), dict(chrom=w.chrom, start=1, end=1, ploidy="diploid"),
                dict(chrom=w.chrom, start=w.length, end=w.length, ploidy="haploid")):
        got = get_snp_testing_candidates(_dct(w, base, None), reg)
        exp = oracle.get_snp_testing_candidates(w, base, reg)
        assert len(got[0]) == len(exp[0])
        for x, y in zip(got, exp):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    # (4) the batched entry point on chunks without candidates, and zero-site CNN calls
    params = dict(base, snp_model="ONT-HG002", disable_coverage_normalization=False, sam_path=w0)
    r = snpCaller.call_chunks(params, [dict(chrom=w0.chrom, start=1, end=2500, ploidy="diploid"), dict(chrom=w0.chrom, start=2500, end=5000, ploidy="diploid")])
    assert r["n"] == 0
    eng.load_weights(_lib.MODEL_SNP, Weights(get_SNP_model("ONT-HG002")[0]))
    z = torch.zeros((0, 5, 41, 5), device="cuda")
    p, g = eng.snp_forward(_lib.MODEL_SNP, z, torch.zeros(0, dtype=torch.int32, device="cuda"), torch.zeros(0, dtype=torch.float64, device="cuda"))
    assert p.shape == (0, 4) and g.shape == (0, 2)


def test_call_chunks_with_min_nbr_sites_filter(eng):
    """min_nbr_sites > 1 drops sites on the host after the batched device pass (generate_SNP_pileups.py:244): the kept
    sites and every per-site array must equal the per-chunk oracle"""
    from nanocaller_amd import snpCaller
    from oracle import oracle
    w = load_world("ont")
    base = dict(threshold=[0.4, 0.6], mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, seq="ont", supplementary=False,
                exclude_bed=None)
    chunks = [dict(chrom=w.chrom, start=1_000, end=30_000, ploidy="diploid"), dict(chrom=w.chrom, start=30_000, end=70_000, ploidy="diploid")]
    # number of real tensor columns of every site (row 0 is the reference one-hot): put the threshold at the median
    ncols = np.concatenate([np.asarray(oracle.get_snp_testing_candidates(w, base, c)[2])[:, 0, :, :4].sum(axis=(1, 2)) for c in chunks])
    base["min_nbr_sites"] = int(np.median(ncols)) + 1
    r = snpCaller.call_chunks(dict(base, snp_model="ONT-HG002", disable_coverage_normalization=False, sam_path=w), chunks)
    exp = [oracle.get_snp_testing_candidates(w, base, c) for c in chunks]
    pos = np.concatenate([np.asarray(e[0], np.int64) for e in exp])
    all_sites = sum(len(oracle.get_snp_testing_candidates(w, dict(base, min_nbr_sites=1), c)[0]) for c in chunks)
    assert 0 < len(pos) < all_sites                                    # the filter really dropped something
    assert np.array_equal(r["pos"], pos) and r["n"] == len(pos)
    assert np.array_equal(r["dp"], np.concatenate([np.asarray(e[3]) for e in exp]))
    assert np.array_equal(r["fwd_dp"], np.concatenate([np.asarray(e[6]) for e in exp]).astype(np.int32))
    assert r["probs"].shape == (len(pos), 4) and r["freq"].shape == (len(pos),)


def test_a_rank_that_owns_part_of_a_contig_decodes_its_span_only(eng, tmp_path, monkeypatch):
    """shard.shard_plan may cut a contig between two ranks: call_chunks then decodes / uploads only the span of its chunks +- the 50 kb
    scan flank (generate_SNP_pileups.contig_span) and returns exactly what the whole-contig decode returns for those chunks"""
    from nanocaller_amd import generate_SNP_pileups as gsp, snpCaller
    from tests import bamio
    world = bamio.make_bam_world(seed=12, length=420_000, depth=14)
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "r.fa")
    bamio.write_bam(bam, world.chrom, world.length, bamio.world_to_records(world, np.random.Generator(np.random.PCG64(2))))
    bamio.write_fasta(fa, world.chrom, world.ref)
    params = dict(sam_path=bam, fasta_path=fa, mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6],
                  snp_model="ONT-HG002", seq="ont", supplementary=False, exclude_bed=None, disable_coverage_normalization=False)
    chunks = [dict(chrom=world.chrom, start=150_000, end=200_000, ploidy="diploid"), dict(chrom=world.chrom, start=200_000, end=260_000, ploidy="diploid")]
    assert gsp.contig_span(bam, world.chrom, chunks) == (100_000, 310_000)
    gsp.release_contig()
    del gsp.DECODES[:]
    part = snpCaller.call_chunks(params, chunks)
    assert gsp.DECODES == [(bam, world.chrom, (100_000, 310_000))]
    gsp.release_contig()
    monkeypatch.setattr(gsp, "contig_span", lambda *a: None)
    whole = snpCaller.call_chunks(params, chunks)
    assert part["n"] == whole["n"] > 100
    for k in ("pos", "chunk", "ref", "dp", "alt", "fwd_dp", "rev_dp", "probs", "gt", "freq"):
        assert np.array_equal(part[k], whole[k]), k
    gsp.release_contig()


def test_reference_skips_are_refused_not_silently_accepted(eng, tmp_path):
    """a kept alignment with a reference skip (CIGAR N): the reference raises KeyError on its '>' pileup symbols (quirk E10); the
    product path answers NC_ERR_UNSUPPORTED instead of coding the skipped positions as deletions"""
    from nanocaller_amd import _lib, generate_SNP_pileups as gsp, snpCaller
    from tests import bamio
    ref = "ACGT" * 2000
    recs = [dict(name="r%d" % k, flag=0, pos0=100 + 7 * k, cigar=[("M", 300)], seq=ref[100 + 7 * k:400 + 7 * k]) for k in range(30)]
    recs.append(dict(name="skip", flag=0, pos0=400, cigar=[("M", 50), ("N", 500), ("M", 50)], seq="A" * 100))
    bam, fa = str(tmp_path / "n.bam"), str(tmp_path / "n.fa")
    bamio.write_bam(bam, "c", len(ref), recs)
    bamio.write_fasta(fa, "c", ref)
    params = dict(sam_path=bam, fasta_path=fa, mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6],
                  snp_model="ONT-HG002", seq="ont", supplementary=False, exclude_bed=None, disable_coverage_normalization=False)
    gsp.release_contig()
    with pytest.raises(_lib.NanoCallerHipError) as e:
        snpCaller.call_chunks(params, [dict(chrom="c", start=1, end=len(ref), ploidy="diploid")])
    assert e.value.status == _lib.NC_ERR_UNSUPPORTED
    gsp.release_contig()


def test_same_name_overlaps_are_keyed_by_name_on_the_snp_route_and_refused_on_the_indel_route(eng, tmp_path):
    """two kept alignments of one read name that overlap on the reference: the reference's per-column dicts hold ONE entry per name
    (generate_SNP_pileups.py:175,185,208).  The SNP route keys such records by name (round 6: pack.name_groups -> nc_snp_set_mates; results
    against the reference's golden and the oracle: tests/test_mates.py); the indel route, whose pass 2 is per alignment, still answers
    NC_ERR_UNSUPPORTED (nc_decoded_check).  A supplementary alignment of the same name only matters when the flag filter keeps it"""
    from nanocaller_amd import _lib, generate_SNP_pileups as gsp, snpCaller
    from tests import bamio
    ref = "ACGT" * 2000
    recs = [dict(name="r%d" % k, flag=0, pos0=100 + 7 * k, cigar=[("M", 300)], seq=ref[100 + 7 * k:400 + 7 * k]) for k in range(30)]
    base = dict(sam_path=None, fasta_path=None, mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6],
                snp_model="ONT-HG002", seq="ont", supplementary=False, exclude_bed=None, disable_coverage_normalization=False)
    chunks = [dict(chrom="c", start=1, end=len(ref), ploidy="diploid")]
    fa = str(tmp_path / "d.fa")
    bamio.write_fasta(fa, "c", ref)
    # (1) the same name twice, both primary
    bam1 = str(tmp_path / "d1.bam")
    bamio.write_bam(bam1, "c", len(ref), sorted(recs + [dict(name="r3", flag=0, pos0=150, cigar=[("M", 200)], seq=ref[150:350])], key=lambda r: r["pos0"]))
    gsp.release_contig()
    assert snpCaller.call_chunks(dict(base, sam_path=bam1, fasta_path=fa), chunks)["n"] >= 0
    w1 = gsp._resolve(bam1, "c", fa)
    with pytest.raises(_lib.NanoCallerHipError) as e:
        gsp._check_supported(w1, bam1, "c")                              # (what the indel route's device_pack() asks)
    assert e.value.status == _lib.NC_ERR_UNSUPPORTED and "same read name" in str(e.value)
    # (2) the second one flagged supplementary: dropped by the default filter, kept and keyed by name with dct['supplementary']
    bam2 = str(tmp_path / "d2.bam")
    bamio.write_bam(bam2, "c", len(ref), sorted(recs + [dict(name="r3", flag=0x800, pos0=150, cigar=[("M", 200)], seq=ref[150:350])], key=lambda r: r["pos0"]))
    gsp.release_contig()
    res = snpCaller.call_chunks(dict(base, sam_path=bam2, fasta_path=fa), chunks)
    assert res["n"] >= 0
    gsp.release_contig()
    assert snpCaller.call_chunks(dict(base, sam_path=bam2, fasta_path=fa, supplementary=True), chunks)["n"] >= 0
    w2 = gsp._resolve(bam2, "c", fa)
    gsp._check_supported(w2, bam2, "c", supplementary=False)
    with pytest.raises(_lib.NanoCallerHipError) as e:
        gsp._check_supported(w2, bam2, "c", supplementary=True)
    assert e.value.status == _lib.NC_ERR_UNSUPPORTED
    gsp.release_contig()


def test_indel_cnn_fused_trunk_at_scale(eng):
    """k10_indel_trunk_h3 with more sites than one workgroup per CU sees in a pass and more than one batch (70,001 diploid sites: 65,536 +
    the rest; 300,001 haploid: 262,144 + the rest): distinct tensors at chosen indices (first / last site of a workgroup's stream, both
    sides of the batch border, the last site) against the float64 oracle; every other site is a copy of one of four tensors and must
    equal that tensor's result bit for bit (a site's result does not depend on its place in the stream)"""
    import torch
    from nanocaller_amd import _lib
    from nanocaller_amd.weights import Weights, get_indel_model
    from oracle import oracle
    rng = np.random.Generator(np.random.PCG64(77))
    for name, kind, rows, n, batch in (("ONT-HG002", _lib.MODEL_INDEL, 15, 70_001, 65_536), ("haploid", _lib.MODEL_INDEL_HAP, 5, 300_001, 262_144)):
        w = Weights(get_indel_model(name))
        eng.load_weights(kind, w)
        base = (rng.random((4, rows, 128, 2)) * (rng.random((4, rows, 128, 2)) < 0.4)).astype(np.float32)
        special = [0, 255, 256, 511, batch - 1, batch, batch + 255, n - 1]
        xs = (rng.random((len(special), rows, 128, 2)) * (rng.random((len(special), rows, 128, 2)) < 0.4)).astype(np.float32)
        xd = torch.from_numpy(base).cuda()[torch.arange(n, device="cuda") % 4].contiguous()
        xd[torch.tensor(special, device="cuda")] = torch.from_numpy(xs).cuda()
        p = eng.indel_forward(kind, xd).cpu().numpy()
        e = oracle.indel_forward(w.flat, np.concatenate([base, xs]), precision="f64")
        assert np.abs(p[special] - e[4:]).max() < 1e-4, name
        plain = np.ones(n, bool)
        plain[special] = False
        for k in range(4):
            rowsk = p[(np.arange(n) % 4 == k) & plain]
            assert np.abs(rowsk[0] - e[k]).max() < 1e-4
            assert np.all(rowsk == rowsk[0]), (name, k)
