"""BASELINE.json configs[2] / configs[4] at their sizes for the INDEL half of the path (csrc/nc_pipe.hip), through size-independent
properties -- the oracle cannot run these sizes in seconds: a chr1-sized ONT 30x contig (248,956,422 bp, 2,490 chunks of 100 kb, ~155 k
candidate sites, ~4.1 M read windows in three balanced alignment groups) and a chr20-sized contig through the haploid model's shape
(--haploid_genome: one read set per site) with the 260-base windows of the pacbio preset (banded like the 160-base ones).
What must hold at any size: results are deterministic, do not depend on how the run loop cuts the alignments into groups, every tensor
column is a frequency distribution, anchors lie inside the window of their chunk, and the first sites equal the oracle's restatement
(pure Python: CIGAR expansion, banded / full star alignment, msa() by the C oracle)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("pos", "chunk", "type", "phase", "ref_len", "alt_len", "alt")


def _workload(L, seed):
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import make_indel_device_workload
    eng = get_engine(0)
    pack, reads_c, info = make_indel_device_workload(eng, L, depth=30.0, seed=seed)
    chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
    return eng, pack, reads_c, info, chunks


def _run(eng, pack, reads_c, L, chunks, **kw):
    from nanocaller_amd import generate_indel_pileups as gip
    r = gip.indel_sites_device(eng, pack, reads_c, L, chunks, mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, **kw)
    return {k: np.array(r[k]) for k in KEYS}, r["x"], r["n"], r["n_alignments"]


def _records(pack, info, hi):
    from oracle import oracle
    r1 = int(np.searchsorted(info["read_start"], hi + 400))
    s, e = info["read_start"][:r1], info["read_end"][:r1]
    slot = pack.reads["slot_off"][:r1 + 1].cpu().numpy()
    codes = pack.codes[:int(slot[-1])].cpu().numpy()
    ev_off = pack.events["ev_off"][:r1 + 1].cpu().numpy()
    ev_pos = pack.events["ev_pos"][:int(ev_off[-1])].cpu().numpy()
    ev_len = pack.events["ev_len"][:int(ev_off[-1])].cpu().numpy()
    ins_off = info["tensors"]["ins_off"][:int(ev_off[-1]) + 1].cpu().numpy()
    ins = info["tensors"]["ins_bases"][:max(int(ins_off[-1]), 1)].cpu().numpy()
    recs = oracle.records_from_indel_pack(
        s, e, lambda r: codes[int(slot[r]) + (int(s[r]) & 15):int(slot[r]) + (int(s[r]) & 15) + int(e[r] - s[r])],
        lambda r: list(zip(ev_pos[ev_off[r]:ev_off[r + 1]].tolist(), ev_len[ev_off[r]:ev_off[r + 1]].tolist())),
        lambda r, k: ins[ins_off[int(ev_off[r]) + k]:ins_off[int(ev_off[r]) + k + 1]])
    ref = np.frombuffer(b"AGTCN", np.uint8)[info["tensors"]["ref"][1:hi + 401].cpu().numpy()].tobytes().decode()
    masked = pack.ref_code[1:hi + 401].cpu().numpy() == 4
    return recs, "".join(c.lower() if m else c for c, m in zip(ref, masked))


def _properties(res, x, n, chunks, S, win_size=40):
    import torch
    # anchors: chunk ids ascend, positions ascend inside a chunk and lie in (start - 10 - win_size, end] of their chunk (:306)
    ch = res["chunk"].astype(np.int64)
    assert np.all(np.diff(ch) >= 0)
    st = np.array([c[0] for c in chunks])[ch]
    en = np.array([c[1] for c in chunks])[ch]
    assert np.all((res["pos"] > st - 10 - win_size) & (res["pos"] <= en))
    same = np.diff(ch) == 0
    assert np.all(np.diff(res["pos"].astype(np.int64))[same] > 0)
    assert set(np.unique(res["type"]).tolist()) <= {0, 1}
    # every used column of every read set: the five symbol frequencies sum to one (msa(), :57-71; the reference one-hot is taken off them)
    xv = x.view(n, S, 5, 128, 2)
    tot = (xv[..., 0] + xv[..., 1]).sum(dim=2)
    used = xv[..., 1].sum(dim=2) > 0
    assert bool(used.any()) and float((tot[used] - 1.0).abs().max()) < 1e-5
    assert bool((xv[..., 1] >= 0).all()) and bool((xv[..., 1].sum(dim=2) <= 1.0 + 1e-6).all())
    # alleles: lengths are -1 (none) or inside the window; ALT bytes are base codes
    assert np.all((res["ref_len"] >= -1) & (res["ref_len"] <= 262)) and np.all(res["alt_len"] >= -1)
    assert res["alt"].size == int(np.maximum(res["alt_len"], 0).sum()) and (res["alt"].size == 0 or int(res["alt"].max()) < 4)
    del torch


def test_indel_pipeline_chr1_sized(monkeypatch):
    import torch
    from oracle import oracle
    L = 248_956_422
    eng, pack, reads_c, info, chunks = _workload(L, 4913)
    a, xa, n, nal = _run(eng, pack, reads_c, L, chunks, window_after=160)
    assert len(chunks) == 2490 and n > 120_000 and nal > 3_000_000
    st = np.zeros(6, np.int64)
    from nanocaller_amd import _lib
    eng.L.nc_indel_sites_band_stats(eng.ctx, _lib.npp(st))
    assert int(st[:3].sum()) == nal and st[0] > 0.85 * nal and st[3] < 0.001 * nal      # nine in ten windows on 32 diagonals
    _properties(a, xa, n, chunks, 3)
    # deterministic, and independent of the cut into groups (default: balanced groups sized to the free memory; here ~9 and ~40 of them,
    # two streams and two buffer sets handed back and forth)
    for group_al in (None, nal // 9 + 1, nal // 40 + 1):
        if group_al:
            monkeypatch.setenv("NC_PIPE_GROUP_AL", str(group_al))
        b, xb, nb, _ = _run(eng, pack, reads_c, L, chunks, window_after=160)
        assert nb == n
        for k in KEYS:
            assert np.array_equal(a[k], b[k]), (k, group_al)
        assert torch.equal(xa, xb), group_al
        del xb
    monkeypatch.delenv("NC_PIPE_GROUP_AL")
    # the first sites against the oracle's restatement, star alignments in pure Python on the bands the device derives
    hi = 25_000
    recs, ref = _records(pack, info, hi)
    xh = xa[:64].cpu().numpy()
    checked = 0
    for k in range(min(n, 64)):
        p = int(a["pos"][k])
        if p > hi:
            break
        got = oracle.indel_site_ref(recs, info["hap"], info["ps"], ref, p, 160, 4, 160, band=True)
        assert got is not None, p
        assert np.array_equal(xh[k].reshape(3, 5, 128, 2), got[0]), p
        assert got[3] == int(a["phase"][k])
        checked += 1
    assert checked >= 8
    del xa, pack
    torch.cuda.empty_cache()


def test_indel_pipeline_haploid_260_base_windows_chr20_sized(monkeypatch):
    """configs[4]'s indel half: --haploid_genome (one read set per site, generate_indel_pileups_haploid.py:128-277) with the pacbio preset's
    260-base windows -- on the CIGAR-derived bands like the 160-base ones (66 blocks of 8 anti-diagonals instead of 41); the first sites against
    the pure-Python restatement of the banded aligner AND against its full-matrix form (the band must not change these)"""
    import torch
    from oracle import oracle
    L = 64_444_167
    eng, pack, reads_c, info, chunks = _workload(L, 5113)
    a, xa, n, nal = _run(eng, pack, reads_c, L, chunks, window_after=260, haploid=True)
    assert n > 8_000 and nal > 200_000
    _properties(a, xa, n, chunks, 1)
    for group_al in (None, nal // 7 + 1):
        if group_al:
            monkeypatch.setenv("NC_PIPE_GROUP_AL", str(group_al))
        b, xb, nb, _ = _run(eng, pack, reads_c, L, chunks, window_after=260, haploid=True)
        assert nb == n
        for k in KEYS:
            assert np.array_equal(a[k], b[k]), (k, group_al)
        assert torch.equal(xa, xb), group_al
        del xb
    monkeypatch.delenv("NC_PIPE_GROUP_AL")
    hi = 45_000
    recs, ref = _records(pack, info, hi)
    xh = xa[:32].cpu().numpy()
    checked = 0
    for k in range(min(n, 32)):
        p = int(a["pos"][k])
        if p > hi or checked >= 4:
            break
        got = oracle.indel_site_ref(recs, info["hap"], info["ps"], ref, p, 260, 4, 160, haploid=True, band=True)      # pure Python, on the bands
        full = oracle.indel_site_ref(recs, info["hap"], info["ps"], ref, p, 260, 4, 160, haploid=True)                # ... and on full matrices
        assert got is not None and full is not None, p
        assert np.array_equal(xh[k].reshape(1, 5, 128, 2), got[0]), p
        assert np.array_equal(got[0], full[0]), p
        checked += 1
    assert checked >= 3
    del xa, pack
    torch.cuda.empty_cache()
