"""BASELINE.json's full size (a chr20-sized ONT 30x contig: 64.4 M positions, 1.93 G pileup entries, ~625 k candidate sites)
through size-independent properties -- the oracle cannot run this size in seconds, so what is checked is what must hold at
any size: the transfer form reproduces the pack byte for byte, both routes give bit-identical results, results are
deterministic, positions are ordered and inside their chunks, the tensors satisfy SURVEY Appendix A's invariants, the
probabilities are probabilities, and the first chunks equal the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

L = 64_444_167


@pytest.fixture(scope="module")
def setup():
    import torch
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import make_device_workload, wire_from_device_workload
    from nanocaller_amd.utils import get_chunks
    eng = get_engine(0)
    pack, info = make_device_workload(eng, L, depth=30.0, tech="ont", seed=912)
    wire = wire_from_device_workload(pack, info)
    chunks = get_chunks([("chr20", 1, L, "diploid")], cpu=16)
    params = dict(mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6], snp_model="ONT-HG002", seq="ont",
                  supplementary=False, exclude_bed=None, disable_coverage_normalization=False, sam_path=None)
    yield eng, pack, info, wire, chunks, params
    del pack
    torch.cuda.empty_cache()


def test_wire_form_reproduces_the_full_pack_and_both_routes_agree(setup):
    import torch
    from nanocaller_amd import snpCaller
    from nanocaller_amd.wire import WireUploader
    eng, pack, info, wire, chunks, params = setup
    assert wire.nbytes < 0.25 * info["pileup_entries"]                       # ~0.22 B per pileup entry instead of 1
    up = WireUploader(eng)
    t = up.submit(wire)
    dpk = up.expand(t)
    torch.cuda.synchronize()
    assert dpk.codes.numel() == pack.codes.numel() and torch.equal(dpk.codes, pack.codes)          # 1.93 GB, byte for byte
    assert torch.equal(dpk.ref_code, pack.ref_code) and torch.equal(dpk.tile_off, pack.tile_off)
    assert torch.equal(dpk.tile_ent[:pack.tile_ent.numel()], pack.tile_ent)
    a = snpCaller.call_chunks(params, chunks, dpk=dpk)
    up.release(t)
    b = snpCaller.call_chunks(params, chunks, dpk=pack)
    c = snpCaller.call_chunks(params, chunks, dpk=pack)
    assert a["n"] == b["n"] == c["n"] > 500_000
    for k in ("pos", "chunk", "ref", "dp", "alt", "fwd_dp", "rev_dp", "probs", "gt", "freq"):
        assert np.array_equal(a[k], b[k]), k                                 # uploaded route == resident route, bit for bit
        assert np.array_equal(b[k], c[k]), k                                 # deterministic
    r = b
    # positions: ascending inside a chunk, inside their chunk (both ends inclusive), chunk ids ascending
    ch = r["chunk"].astype(np.int64)
    assert np.all(np.diff(ch) >= 0)
    starts = np.array([c_["start"] for c_ in chunks])[ch]
    ends = np.array([c_["end"] for c_ in chunks])[ch]
    assert np.all((r["pos"] >= starts) & (r["pos"] <= ends))
    same = np.diff(ch) == 0
    assert np.all(np.diff(r["pos"].astype(np.int64))[same] > 0)
    # a position shared by two adjacent chunks is emitted by both (quirk E3)
    shared = set(c_["end"] for c_ in chunks[:-1])
    dup = [p for p in shared if np.count_nonzero(r["pos"] == p) == 2]
    assert len(dup) == np.count_nonzero(np.isin(r["pos"], list(shared))) // 2
    # depths and frequencies
    assert np.all(r["dp"] >= params["mincov"]) and np.all(r["alt"] <= r["dp"])
    assert np.all(r["freq"] >= params["min_allele_freq"]) and np.all(r["freq"] <= 1.0)
    assert np.all(r["fwd_dp"] >= 0) and np.all((r["fwd_dp"] + r["rev_dp"]).sum(1) <= r["dp"])
    # probabilities: the four allele heads and the GT head are softmax outputs
    assert np.all((r["probs"] >= 0) & (r["probs"] <= 1)) and np.all(np.isfinite(r["probs"]))
    assert np.abs(r["gt"].sum(1) - 1).max() < 1e-5
    # planted truth comes back: most sites called with two alleles >= 0.5 sit at het sites of the generator (1 / 1000 bp)
    assert 0.5 * L / 1000 < np.count_nonzero((r["probs"] >= 0.5).sum(1) >= 2) < 2.5 * L / 1000


def device_tensors(eng, pack, chunks, params, haploid=False):
    """the product's int16 site tensors of ALL `chunks` (scan + featuriser through the C ABI, as snpCaller.call_chunks runs them) -> (SnpSites, x int16
    [N, 5, 41, 5] on the device): the full-size tests compare the tensors of the checked chunks with the oracle's, not only what the CNN makes of them"""
    eng.use_torch_stream()
    sites = eng.snp_scan(pack, [(c["start"], c["end"]) for c in chunks], mincov=params["mincov"], min_allele_freq=params["min_allele_freq"],
                         threshold=params["threshold"], haploid=haploid)
    eng.set_tensor_format(int16=True)
    try:
        eng.snp_featurize(pack, sites, seq=params["seq"], maxcov=params["maxcov"], min_nbr_sites=params["min_nbr_sites"])
    finally:
        eng.set_tensor_format(int16=False)
    return sites, sites.x


def assert_chunk_tensors_equal(sites, x, ci, pos, mat):
    import torch
    sel = np.nonzero(sites.chunk == ci)[0]
    assert np.array_equal(sites.pos[sel], pos), ci
    got = x[torch.from_numpy(sel).to(x.device)].cpu().numpy()
    assert got.shape == mat.shape and np.array_equal(got, mat.astype(np.int16)), "site tensors of chunk %d differ from the oracle's" % ci


def test_tensor_invariants_at_full_size(setup):
    """SURVEY Appendix A on every site of the first 40 chunks (~190 k tensors)"""
    from nanocaller_amd import _lib
    eng, pack, info, wire, chunks, params = setup
    eng.use_torch_stream()
    sub = chunks[:40]
    sites = eng.snp_scan(pack, [(c["start"], c["end"]) for c in sub], mincov=4, min_allele_freq=0.15, threshold=[0.4, 0.6])
    eng.set_tensor_format(int16=False)
    eng.snp_featurize(pack, sites, seq="ont", maxcov=160, min_nbr_sites=1)
    x = sites.x.cpu().numpy().reshape(-1, 5, 41, 5)
    n = x.shape[0]
    assert n > 150_000
    real = x[:, 0, :, :4].sum(2) == 1                                          # row 0: one-hot reference base on real columns
    assert np.all((x[:, 0, :, :4].sum(2) == 0) | real) and np.all(x[:, 0, :, 4] == 0)
    assert np.all(real[:, 20])                                                 # the candidate is always at column 20
    pad = ~real
    assert np.all(x[:, 1:, :, :][np.broadcast_to(pad[:, None, :, None], x[:, 1:].shape)] == 0)       # padding columns are all zero
    # the centre column is diagonal: +-count of reads with centre base i, negative iff i is the reference base
    centre = x[:, 1:, 20, :4]
    off = centre.copy()
    off[:, np.arange(4), np.arange(4)] = 0
    assert np.all(off == 0)
    refb = np.argmax(x[:, 0, 20, :4], 1)
    diag = centre[:, np.arange(4), np.arange(4)]
    assert np.all(diag[np.arange(n), refb] <= 0)
    mask = np.ones((n, 4), bool)
    mask[np.arange(n), refb] = False
    assert np.all(diag[mask] >= 0)
    # channel 4 marks the row of the centre's reference base on every real column
    assert np.all(x[np.arange(n), 1 + refb, 20, 4] == 1)
    assert np.all(x[:, 1:, :, 4].sum(1)[real] == 1)
    # counts are bounded by the sampled depth (<= maxcov)
    assert np.abs(x[:, 1:, :, :4]).sum((1, 3)).max() <= 160
    assert np.array_equal(sites.ref_code.cpu().numpy(), refb)


def test_first_chunks_equal_the_oracle(setup):
    from nanocaller_amd import snpCaller
    from nanocaller_amd.synth_device import host_sample_for_oracle
    from nanocaller_amd.weights import Weights, get_SNP_model
    from oracle import oracle
    eng, pack, info, wire, chunks, params = setup
    sub = chunks[:3]
    r = snpCaller.call_chunks(params, chunks, dpk=pack)
    h = host_sample_for_oracle(pack, info, 1, sub[-1]["end"] + 50_000)
    rr = oracle.RawReads("chr20", h["L"], h["start"], h["end"], h["off"], h["codes"], h["strand"], h["keep"])
    path, cov = get_SNP_model("ONT-HG002")
    w = Weights(path)
    sites, x = device_tensors(eng, pack, chunks, params)
    for ci, c in enumerate(sub):
        pos, ref, mat, dp, freq, depth, fwd, rev = oracle.get_snp_testing_candidates(rr, params, c, rc=h["ref_codes"])
        sel = r["chunk"] == ci
        assert np.array_equal(r["pos"][sel], pos) and np.array_equal(r["dp"][sel], dp)
        assert np.array_equal(r["fwd_dp"][sel], fwd) and np.array_equal(r["rev_dp"][sel], rev)
        assert_chunk_tensors_equal(sites, x, ci, pos, mat)
        probs, _ = oracle.snp_forward(w.flat, mat, np.argmax(ref, 1).astype(np.int32), np.full(len(pos), cov / depth), precision="f64")
        assert np.abs(r["probs"][sel] - probs).max() < 1e-4


def test_the_cnn_enqueued_inside_the_next_scan_gives_the_same_results(setup):
    """snpCaller.call_chunks(pipeline=True) (NC_PIPE_CNN=1; off by default: measured slower): a group's CNN is enqueued between nc_snp_scan_begin and
    nc_snp_scan_end of the NEXT group, or by result() when no call follows -- device order scan(k + 1), CNN(k), tensors(k + 1) -- and every result equals
    the unpipelined call's bit for bit"""
    from nanocaller_amd import snpCaller
    eng, pack, info, wire, chunks, params = setup
    groups = [chunks[:12], chunks[12:30], chunks[5:9], chunks[:12]]
    plain = [snpCaller.call_chunks(params, g, dpk=pack, defer=True, pipeline=False).result() for g in groups]
    pend = []
    for k, g in enumerate(groups):
        pend.append(snpCaller.call_chunks(params, g, dpk=pack, defer=True, pipeline=True))
        assert 0 in snpCaller._PENDING_CNN                              # this group's CNN waits for the next scan
        if k == 1:
            assert pend[0].result()["n"] == plain[0]["n"]              # (collected while a later group's CNN is still pending)
    piped = [p.result() for p in pend]
    assert 0 not in snpCaller._PENDING_CNN
    for a, b in zip(plain, piped):
        assert a["n"] == b["n"] > 1000
        for key in ("pos", "chunk", "ref", "dp", "alt", "fwd_dp", "rev_dp", "probs", "gt", "freq"):
            assert np.array_equal(a[key], b[key]), key


def test_hifi_60x_haploid_at_full_size():
    """BASELINE.json configs[4]'s shape: a chr20-sized HiFi 60x contig (3.9 G pileup entries), `pacbio` neighbour buckets,
    haploid model.  The wire form (0.04 B per entry: a HiFi read is 99.8 % reference) reproduces the pack byte for byte, the
    uploaded and the resident route agree bit for bit, and the first chunks equal the oracle."""
    import torch
    from nanocaller_amd import snpCaller
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import host_sample_for_oracle, make_device_workload, wire_from_device_workload
    from nanocaller_amd.utils import get_chunks
    from nanocaller_amd.weights import Weights, get_SNP_model
    from nanocaller_amd.wire import WireUploader
    from oracle import oracle
    eng = get_engine(0)
    pack, info = make_device_workload(eng, L, depth=60.0, tech="hifi", seed=913)
    try:
        wire = wire_from_device_workload(pack, info)
        assert wire.nbytes < 0.06 * info["pileup_entries"]
        chunks = get_chunks([("chr20", 1, L, "haploid")], cpu=16)
        params = dict(mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6], snp_model="CCS-HG002", seq="pacbio",
                      supplementary=False, exclude_bed=None, disable_coverage_normalization=False, sam_path=None)
        up = WireUploader(eng)
        t = up.submit(wire)
        dpk = up.expand(t)
        torch.cuda.synchronize()
        assert torch.equal(dpk.codes, pack.codes) and torch.equal(dpk.ref_code, pack.ref_code)
        a = snpCaller.call_chunks(params, chunks, dpk=dpk)
        up.release(t)
        b = snpCaller.call_chunks(params, chunks, dpk=pack)
        assert a["n"] == b["n"] > 100_000
        for k in ("pos", "chunk", "ref", "dp", "alt", "fwd_dp", "rev_dp", "probs", "freq"):
            assert np.array_equal(a[k], b[k]), k
        assert np.all((b["probs"] >= 0) & (b["probs"] <= 1))
        sub = chunks[:2]
        h = host_sample_for_oracle(pack, info, 1, sub[-1]["end"] + 50_000)
        rr = oracle.RawReads("chr20", h["L"], h["start"], h["end"], h["off"], h["codes"], h["strand"], h["keep"])
        w = Weights(get_SNP_model("haploid")[0])
        sites, x = device_tensors(eng, pack, chunks[:4], params, haploid=True)
        for ci, c in enumerate(sub):
            pos, ref, mat, dp, freq, depth, fwd, rev = oracle.get_snp_testing_candidates(rr, params, c, rc=h["ref_codes"])
            sel = b["chunk"] == ci
            assert np.array_equal(b["pos"][sel], pos) and np.array_equal(b["dp"][sel], dp)
            assert_chunk_tensors_equal(sites, x, ci, pos, mat)
            probs = oracle.snp_hap_forward(w.flat, mat, np.argmax(ref, 1).astype(np.int32), np.full(len(pos), 30.0 / depth), precision="f64")
            assert np.abs(b["probs"][sel] - probs).max() < 1e-4
    finally:
        del pack
        torch.cuda.empty_cache()


def test_snp_half_of_configs2_at_chr1_size():
    """BASELINE.json configs[2]'s SNP half at its full size: a chr1-sized ONT 30x contig (248,956,422 positions, 498 chunks of 500 kb, ~7.5 G pileup
    entries, ~2.4 M candidate sites) through the uploaded route and the resident one -- bit-identical, deterministic, ordered inside their chunks,
    the boundary columns emitted by both neighbours (quirk E3), probabilities that are probabilities -- and the first and the LAST two chunks equal
    to the oracle (the last ones sit behind 2^31 bytes of codes: offsets past 32 bits)"""
    import torch
    from nanocaller_amd import snpCaller
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import host_sample_for_oracle, make_device_workload, wire_from_device_workload
    from nanocaller_amd.utils import get_chunks
    from nanocaller_amd.weights import Weights, get_SNP_model
    from nanocaller_amd.wire import WireUploader
    from oracle import oracle
    eng = get_engine(0)
    L1 = 248_956_422
    pack, info = make_device_workload(eng, L1, depth=30.0, tech="ont", seed=914)
    try:
        assert pack.codes.numel() > (1 << 32)
        chunks = get_chunks([("chr1", 1, L1, "diploid")], cpu=16)
        assert len(chunks) == 498
        params = dict(mincov=4, maxcov=160, min_allele_freq=0.15, min_nbr_sites=1, threshold=[0.4, 0.6], snp_model="ONT-HG002", seq="ont",
                      supplementary=False, exclude_bed=None, disable_coverage_normalization=False, sam_path=None)
        wire = wire_from_device_workload(pack, info)
        up = WireUploader(eng)
        t = up.submit(wire)
        dpk = up.expand(t)
        torch.cuda.synchronize()
        assert torch.equal(dpk.codes, pack.codes)                                  # 7.5 GB, byte for byte
        a = snpCaller.call_chunks(params, chunks, dpk=dpk)
        up.release(t)
        del dpk, wire
        b = snpCaller.call_chunks(params, chunks, dpk=pack)
        assert a["n"] == b["n"] > 2_000_000
        for k in ("pos", "chunk", "ref", "dp", "alt", "fwd_dp", "rev_dp", "probs", "gt", "freq"):
            assert np.array_equal(a[k], b[k]), k
        r = b
        ch = r["chunk"].astype(np.int64)
        assert np.all(np.diff(ch) >= 0) and ch[-1] == 497
        starts = np.array([c_["start"] for c_ in chunks])[ch]
        ends = np.array([c_["end"] for c_ in chunks])[ch]
        assert np.all((r["pos"] >= starts) & (r["pos"] <= ends))
        assert np.all(np.diff(r["pos"].astype(np.int64))[np.diff(ch) == 0] > 0)
        shared = np.array([c_["end"] for c_ in chunks[:-1]])
        hit = np.isin(r["pos"], shared)
        assert np.count_nonzero(hit) % 2 == 0 and np.count_nonzero(hit) > 0         # a shared boundary column is emitted by both chunks
        assert np.all((r["probs"] >= 0) & (r["probs"] <= 1)) and np.all(np.isfinite(r["probs"])) and np.abs(r["gt"].sum(1) - 1).max() < 1e-5
        assert np.all(r["dp"] >= 4) and np.all(r["freq"] >= 0.15)
        path, cov = get_SNP_model("ONT-HG002")
        w = Weights(path)
        sites, x = device_tensors(eng, pack, chunks, params)                          # 2.4 M tensors, 5 GB as int16
        for lo_c, hi_c in ((0, 2), (496, 498)):
            sub = chunks[lo_c:hi_c]
            h = host_sample_for_oracle(pack, info, max(1, sub[0]["start"] - 50_000), min(L1, sub[-1]["end"] + 50_000))
            rr = oracle.RawReads("chr1", h["L"], h["start"], h["end"], h["off"], h["codes"], h["strand"], h["keep"])
            for ci, c in zip(range(lo_c, hi_c), sub):
                pos, ref, mat, dp, freq, depth, fwd, rev = oracle.get_snp_testing_candidates(rr, params, c, rc=h["ref_codes"])
                sel = r["chunk"] == ci
                assert np.array_equal(r["pos"][sel], pos) and np.array_equal(r["dp"][sel], dp), ci
                assert np.array_equal(r["fwd_dp"][sel], fwd) and np.array_equal(r["rev_dp"][sel], rev), ci
                assert_chunk_tensors_equal(sites, x, ci, pos, mat)
                probs, _ = oracle.snp_forward(w.flat, mat, np.argmax(ref, 1).astype(np.int32), np.full(len(pos), cov / depth), precision="f64")
                assert np.abs(r["probs"][sel] - probs).max() < 1e-4, ci
    finally:
        del pack
        torch.cuda.empty_cache()
