"""A whole-genome-shaped slice through the product path: three contigs of UNEQUAL length, per contig the SNP half then the indel half, every wire
from pinned host memory through the upload ring in one pipelined run (bench.py's run_pairs: what `wgs_block` times over 24 contigs at GRCh38 lengths;
the reference walks all regions with snpCaller, then indelCaller: NanoCaller:25-55, utils.py:6-83).  Checked: the pipelined results equal each
contig's own resident pass bit for bit (no state leaks from one contig / one half into the next: ring slots, workspaces sized by another contig,
the SNP -> indel hand-over), and the first and the last chunk of every contig equal the oracle INCLUDING THE TENSORS."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LENGTHS = [("chrA", 5_300_017), ("chrB", 1_250_003), ("chrC", 3_100_999)]          # unequal, none a multiple of the chunk sizes


def test_three_unequal_contigs_snp_and_indel_halves_equal_the_oracle_on_first_and_last_chunks():
    import torch

    import bench
    from nanocaller_amd import snpCaller
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import host_sample_for_oracle
    from nanocaller_amd.utils import get_chunks
    from nanocaller_amd.weights import Weights, get_SNP_model
    from nanocaller_amd.wire import WireUploader
    from oracle import oracle
    from test_full_size import assert_chunk_tensors_equal, device_tensors
    eng = get_engine(0)
    params = bench.snp_params("ONT-HG002", "ont")
    up = WireUploader(eng)
    units, keep = [], []
    for k, (name, L) in enumerate(LENGTHS):
        snp = bench.Contig(eng, L, 30.0, "ont", seed=3100 + k, keep_pack=True)
        job = bench.IndelJob(eng, L, seed=7100 + k, name=name.encode())
        units.append(bench.PairUnit(snp, job, get_chunks([(name, 1, L, "diploid")], cpu=16), name))
    # ---- the pipelined run: results of every step collected by wrapping the host half
    got_snp, got_indel = {}, {}
    orig = bench._pair_host_half

    def spy(us, rs, ui, ri):
        if rs is not None:
            got_snp.setdefault(us.name, []).append(rs)
        if ri is not None:
            got_indel.setdefault(ui.name, []).append(ri)
        return orig(us, rs, ui, ri)
    bench._pair_host_half = spy
    try:
        ns, ni, nrec = bench.run_pairs(up, 0, params, units, 2 * len(units))        # two passes over the three contigs
    finally:
        bench._pair_host_half = orig
    assert ns > 0 and ni > 0 and nrec > 0
    path, cov = get_SNP_model("ONT-HG002")
    w = Weights(path)
    for u in units:
        L = u.snp.info["L"]
        # (1) pipelined == the contig's own resident pass, both passes, both halves
        ref_s = snpCaller.call_chunks(params, u.chunks, dpk=u.snp.pack)
        assert len(got_snp[u.name]) == 2
        for rs in got_snp[u.name]:
            assert rs["n"] == ref_s["n"] > 0
            for key in ("pos", "chunk", "ref", "dp", "alt", "fwd_dp", "rev_dp", "probs", "gt", "freq"):
                assert np.array_equal(rs[key], ref_s[key]), (u.name, key)
        ref_i = u.job.gpu_pass(u.job.pack, u.job.reads_c)
        assert len(got_indel[u.name]) == 2
        for ri in got_indel[u.name]:
            assert ri["n"] == ref_i["n"] > 0
            for key in ("pos", "chunk", "type", "phase", "ref_len", "alt_len", "alt", "probs"):
                assert np.array_equal(ri[key], ref_i[key]), (u.name, key)
        # (2) SNP half: first and last chunk == oracle, tensors included
        sites, x = device_tensors(eng, u.snp.pack, u.chunks, params)
        last = len(u.chunks) - 1
        for ci in sorted({0, last}):
            c = u.chunks[ci]
            h = host_sample_for_oracle(u.snp.pack, u.snp.info, max(1, c["start"] - 50_000), min(L, c["end"] + 50_000))
            rr = oracle.RawReads(u.name, h["L"], h["start"], h["end"], h["off"], h["codes"], h["strand"], h["keep"])
            pos, ref, mat, dp, freq, depth, fwd, rev = oracle.get_snp_testing_candidates(rr, params, c, rc=h["ref_codes"])
            sel = ref_s["chunk"] == ci
            assert np.array_equal(ref_s["pos"][sel], pos) and np.array_equal(ref_s["dp"][sel], dp), (u.name, ci)
            assert np.array_equal(ref_s["fwd_dp"][sel], fwd) and np.array_equal(ref_s["rev_dp"][sel], rev), (u.name, ci)
            assert_chunk_tensors_equal(sites, x, ci, pos, mat)
            probs, _ = oracle.snp_forward(w.flat, mat, np.argmax(ref, 1).astype(np.int32), np.full(len(pos), cov / depth), precision="f64")
            assert np.abs(ref_s["probs"][sel] - probs).max() < 1e-4, (u.name, ci)
        del sites, x
        # (3) indel half: the first sites against the oracle's restatement of pass 2 (tensors, phase, alleles) and K9 against float64
        from nanocaller_amd import _lib
        from nanocaller_amd import generate_indel_pileups as gip
        rt = gip.indel_sites_device(eng, u.job.pack, u.job.reads_c, L, u.job.chunks, fetch=False, **u.job.kw)
        probs_d = eng.indel_forward(_lib.MODEL_INDEL, rt["x"])
        rt.update(gip.indel_sites_fetch(eng, rt["n"], rt["sets"]))
        checked, x_exact, alleles_exact, k9_err, m = bench.indel_parity_sample(u.job.pack, u.job.info, u.job.contig, rt, probs_d, u.job.wgt, hi=25_000, max_sites=12)
        assert checked >= 4 and x_exact and alleles_exact, (u.name, checked, x_exact, alleles_exact)
        assert k9_err < 1e-4, (u.name, k9_err)
        del rt, probs_d
    del units
    torch.cuda.empty_cache()
