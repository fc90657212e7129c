"""The device-resident indel path (csrc/nc_pipe.hip: nc_indel_sites_plan / _run / _fetch, rows a10-a15 of SURVEY.md 8a) and its host pieces.

CPU: nc_indel_pack_build (bases without a reference column) against a restatement from the decoded query sequences;
nc_decoded_check (reference skips, same-name overlaps -> NC_ERR_UNSUPPORTED); nc_indel_vcf_format against the Python statement
of the reference's rules (indelCaller.indel_vcf_lines[_haploid], itself pinned by the reference's indel_run() text).
GPU: the device pipeline returns, chunk by chunk, exactly the tuples of the host-assembled route (nc_indel_pass2_sets ->
nc_star_msa_tensor_dup -> nc_allele_prediction_device), which tests/test_pass2_golden.py pins to the reference's own outputs."""
import ctypes as C
import os

import numpy as np
import pytest

from nanocaller_amd import _lib, indelCaller
from nanocaller_amd import generate_indel_pileups as gip
from nanocaller_amd.bam import decode_parallel

import bamio


@pytest.fixture(scope="module")
def world_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("pipe")
    w = bamio.make_pass2_world(seed=11, length=150_000, depth=26)
    bam, fa = str(d / "p.bam"), str(d / "p.fa")
    bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None))
    bamio.write_fasta(fa, w.chrom, w.ref)
    return w, bam, fa


def _pack(dec, keep, tail_cap):
    L = _lib.lib()
    h = C.c_void_p()
    assert L.nc_indel_pack_build(dec["_owner"].handle, _lib.npp(keep), tail_cap, C.byref(h)) == _lib.NC_OK
    v = _lib.IndelPackArraysC()
    L.nc_indel_pack_view(h, C.byref(v))

    def arr(ptr, n, dt):
        return np.frombuffer((C.c_char * (int(n) * np.dtype(dt).itemsize)).from_address(ptr), dt).copy() if n else np.zeros(0, dt)
    out = dict(n=v.n_reads, ev_off=arr(v.ev_off, v.n_reads + 1, np.int32), ev_pos=arr(v.ev_pos, v.n_events, np.int32),
               ev_len=arr(v.ev_len, v.n_events, np.int32), ins_off=arr(v.ins_off, v.n_events + 1, np.int32),
               ins=arr(v.ins_bases, v.n_ins_bases, np.uint8), tail_off=arr(v.tail_off, v.n_reads + 1, np.int32),
               tail=arr(v.tail_bases, v.n_tail_bases, np.uint8), ps=arr(v.read_ps, v.n_reads, np.int32), hap=arr(v.read_hap, v.n_reads, np.uint8),
               flag=arr(v.read_flag, v.n_reads, np.uint8))
    L.nc_indel_pack_free(h)
    return out


def test_indel_pack_holds_the_bases_without_a_reference_column(world_files):
    w, bam, fa = world_files
    dec = decode_parallel(bam, w.chrom, keep_seq=True)
    n = len(dec["read_start"])
    keep = np.ascontiguousarray((dec["read_flag"] & 0xF04) == 0, np.uint8)
    keep[::17] = 0
    p = _pack(dec, keep, 40)
    kept = np.nonzero(keep)[0]
    assert p["n"] == len(kept) and n > 300
    n_ins = n_tail = 0
    for k, r in enumerate(kept.tolist()):
        e0, e1 = int(dec["ev_off"][r]), int(dec["ev_off"][r + 1])
        assert p["ev_off"][k + 1] - p["ev_off"][k] == e1 - e0
        w0 = int(p["ev_off"][k])
        assert np.array_equal(p["ev_pos"][w0:w0 + e1 - e0], dec["ev_pos"][e0:e1]) and np.array_equal(p["ev_len"][w0:w0 + e1 - e0], dec["ev_len"][e0:e1])
        assert p["ps"][k] == dec["ps"][r] and p["hap"][k] == dec["hap"][r] and p["flag"][k] == 0
        seq = dec["seq"][dec["seq_off"][r]:dec["seq_off"][r + 1]]
        q, rp = int(dec["qstart"][r]), int(dec["read_start"][r])
        for e in range(e0, e1):                                                   # the query walk of the CIGAR, event by event
            ep, el = int(dec["ev_pos"][e]), int(dec["ev_len"][e])
            q += ep - rp + 1
            rp = ep + 1
            a, b = int(p["ins_off"][w0 + e - e0]), int(p["ins_off"][w0 + e - e0 + 1])
            if el > 0:
                assert np.array_equal(p["ins"][a:b], seq[q:q + el])
                q += el
                n_ins += 1
            else:
                assert a == b
                rp += -el
        q += int(dec["read_end"][r]) - rp
        t0, t1 = int(p["tail_off"][k]), int(p["tail_off"][k + 1])
        assert np.array_equal(p["tail"][t0:t1], seq[q:q + 40])
        n_tail += t1 - t0
    assert n_ins > 100 and n_tail == 2 * len(kept)                                 # world_to_records ends every read with a 2-base soft clip


def test_unsupported_inputs_are_reported_not_accepted(tmp_path):
    """a reference-skip CIGAR and two overlapping alignments with one read name (quirks E10, Appendix A.1): NC_ERR_UNSUPPORTED"""
    L = _lib.lib()
    ref = "ACGT" * 500

    def dec_of(recs, name):
        bam = str(tmp_path / name)
        bamio.write_bam(bam, "c", len(ref), recs)
        return decode_parallel(bam, "c", keep_seq=True)

    def check(dec, keep=None):
        a, b = C.c_int64(), C.c_int64()
        rc = L.nc_decoded_check(dec["_owner"].handle, _lib.npp(keep) if keep is not None else None, C.byref(a), C.byref(b))
        return rc, a.value, b.value
    ok = [dict(name="r1", flag=0, pos0=10, cigar=[("M", 50)], seq="A" * 50), dict(name="r2", flag=16, pos0=30, cigar=[("M", 20), ("D", 3), ("M", 20)], seq="C" * 40)]
    assert check(dec_of(ok, "ok.bam")) == (_lib.NC_OK, 0, 0)
    skip = ok + [dict(name="r3", flag=0, pos0=40, cigar=[("M", 10), ("N", 100), ("M", 10)], seq="G" * 20)]
    d = dec_of(skip, "skip.bam")
    assert check(d) == (_lib.NC_ERR_UNSUPPORTED, 1, 0)
    assert (d["read_flag"] & _lib.FLAG_REFSKIP).tolist() == [0, 0, _lib.FLAG_REFSKIP]
    assert check(d, np.array([1, 1, 0], np.uint8)) == (_lib.NC_OK, 0, 0)           # ... unless the flag filter drops the read
    dup = ok + [dict(name="r1", flag=0x800, pos0=45, cigar=[("M", 30)], seq="T" * 30), dict(name="r2", flag=0x800, pos0=500, cigar=[("M", 30)], seq="T" * 30)]
    d = dec_of(dup, "dup.bam")
    assert check(d) == (_lib.NC_ERR_UNSUPPORTED, 0, 1)                              # r1 twice over [46, 60]; r2's second alignment is elsewhere
    assert check(d, np.ascontiguousarray((d["read_flag"] & 0x800) == 0, np.uint8)) == (_lib.NC_OK, 0, 0)


def test_the_indel_routes_contig_world_carries_the_unsupported_counts(tmp_path):
    """the World that decoded_contig registers for the SNP path / pass 1 (one decode per contig) is checked like bam.read_bam's: with
    --supplementary a read name whose two kept alignments overlap is refused on the indel route too"""
    from nanocaller_amd import generate_SNP_pileups as gsp
    ref = "ACGT" * 500
    recs = [dict(name="r1", flag=0, pos0=10, cigar=[("M", 50)], seq="A" * 50), dict(name="r2", flag=16, pos0=30, cigar=[("M", 40)], seq="C" * 40),
            dict(name="r1", flag=0x800, pos0=45, cigar=[("M", 30)], seq="T" * 30)]
    bam, fa = str(tmp_path / "d.bam"), str(tmp_path / "d.fa")
    bamio.write_bam(bam, "c", len(ref), recs)
    bamio.write_fasta(fa, "c", ref)
    gip._CONTIGS.clear()
    gsp.release_contig()
    gip.decoded_contig(bam, "c", fa)
    w = gsp._BAM_WORLDS.get((bam, fa, "c"), lambda: pytest.fail("decoded_contig did not register its World"))
    assert w.meta["unsupported"] == {False: (0, 0), True: (0, 1)}
    gsp._check_supported(w, bam, "c", supplementary=False)
    with pytest.raises(_lib.NanoCallerHipError) as ei:
        gsp._check_supported(w, bam, "c", supplementary=True)
    assert ei.value.status == _lib.NC_ERR_UNSUPPORTED
    # a World nobody counted for is counted on the spot, not waved through
    del w.meta["unsupported"]
    with pytest.raises(_lib.NanoCallerHipError):
        gsp._check_supported(w, bam, "c", supplementary=True)
    gip._CONTIGS.clear()
    gsp.release_contig()


def _random_sites(rng, n, n_chunks, L, haploid):
    contig = "".join("AGTC"[i] for i in rng.integers(0, 4, size=L))
    chunk = np.sort(rng.integers(0, n_chunks, size=n)).astype(np.int32)
    pos = np.zeros(n, np.int32)
    for c in range(n_chunks):
        sel = np.nonzero(chunk == c)[0]
        pos[sel] = np.sort(rng.integers(1, L - 200, size=len(sel)))                  # close and repeated positions: `prev` matters
    S = 1 if haploid else 3
    rl = rng.integers(1, 30, size=(n, S)).astype(np.int32)
    al = rng.integers(0, 30, size=(n, S)).astype(np.int32)
    none = rng.random((n, S)) < 0.25
    rl[none], al[none] = -1, -1
    same = rng.random(n) < 0.2
    if not haploid:
        rl[same, 1], al[same, 1] = rl[same, 0], al[same, 0]
    tot = int(np.maximum(al, 0).sum())
    alt = rng.integers(0, 4, size=max(tot, 1)).astype(np.uint8)
    off = np.zeros(n * S + 1, np.int64)
    np.cumsum(np.maximum(al.reshape(-1), 0), out=off[1:])
    if not haploid:
        for j in np.nonzero(same)[0]:                                               # equal alleles on both haplotypes (:109)
            if al[j, 0] > 0:
                alt[off[j * 3 + 1]:off[j * 3 + 2]] = alt[off[j * 3]:off[j * 3 + 1]]
    if haploid:
        probs = rng.random((n, 1)).astype(np.float32)
    else:
        probs = rng.dirichlet([0.6, 0.6, 0.6, 0.6], size=n).astype(np.float32)
        probs[rng.random(n) < 0.1, 0] = np.float32(0.97)
    phase = np.where(rng.random(n) < 0.7, rng.integers(1, 10**6, size=n), 0).astype(np.int32)
    return contig, pos, chunk, probs, rl, al, alt, off, phase


@pytest.mark.parametrize("haploid", [False, True])
def test_native_indel_rules_write_the_python_statements_text(haploid):
    L = _lib.lib()
    rng = np.random.default_rng(5 + haploid)
    n, n_chunks, Lc = 4000, 23, 60_000
    contig, pos, chunk, probs, rl, al, alt, off, phase = _random_sites(rng, n, n_chunks, Lc, haploid)
    S = 1 if haploid else 3
    letters = np.frombuffer(b"AGTC", np.uint8)[alt].tobytes().decode()
    exp, exp_off = [], [0]
    for c in range(n_chunks):
        sel = np.nonzero(chunk == c)[0]
        alle = []
        for j in sel:
            row = [(None, None) if rl[j, t] < 0 else (contig[pos[j] - 1:pos[j] - 1 + rl[j, t]], letters[off[j * S + t]:off[j * S + t + 1]]) for t in range(S)]
            alle.append(row[0] if haploid else row)
        prev = 0
        for b in range(0, len(sel), 100):                                           # indel_run's batches of 100 (indelCaller.py:59), prev carried
            s = sel[b:b + 100]
            if haploid:
                lines, prev = indelCaller.indel_vcf_lines_haploid("chrT", pos[s].tolist(), probs[s], alle[b:b + 100], prev)
            else:
                lines, prev = indelCaller.indel_vcf_lines("chrT", pos[s].tolist(), probs[s], alle[b:b + 100], [int(p) if p else None for p in phase[s]], prev)
            exp += lines
        exp_off.append(sum(len(x) for x in exp))
    out = np.empty(n * 400, np.uint8)
    nb = C.c_int64()
    coff = np.empty(n_chunks + 1, np.int64)
    rc = L.nc_indel_vcf_format(b"chrT", n, _lib.npp(pos), _lib.npp(chunk), n_chunks, _lib.npp(probs), S, _lib.npp(rl), _lib.npp(al), _lib.npp(alt),
                               _lib.npp(phase), contig.encode(), Lc, 1 if haploid else 0, _lib.npp(out), out.size, C.byref(nb), _lib.npp(coff))
    assert rc == _lib.NC_OK
    got = out[:nb.value].tobytes().decode()
    assert len(exp) > 800 and got == "".join(exp)
    assert coff.tolist() == exp_off
    if not haploid:
        kinds = {ln.split("\t")[9].split(":")[0] for ln in exp}
        assert kinds == {"1/1", "1|2", "0|1", "1|0"} and any("GT:GQ:PS" in ln for ln in exp) and any(ln.endswith("GT:GQ\t1|2:%s\n" % ln.rstrip("\n").split(":")[-1]) for ln in exp)


def test_native_indel_rules_print_nan_for_a_non_finite_probability():
    """a model that overflowed gives NaN probabilities: the reference's format() writes 'nan' into the quality fields; the native formatter does
    the same instead of shifting a 128-bit integer by the NaN's exponent"""
    L = _lib.lib()
    rng = np.random.default_rng(77)
    n, n_chunks, Lc = 300, 3, 20_000
    contig, pos, chunk, probs, rl, al, alt, off, phase = _random_sites(rng, n, n_chunks, Lc, False)
    probs[::7, 1] = np.nan
    probs[3::11, 2] = np.inf
    letters = np.frombuffer(b"AGTC", np.uint8)[alt].tobytes().decode()
    exp = []
    for c in range(n_chunks):
        sel = np.nonzero(chunk == c)[0]
        alle = [[(None, None) if rl[j, t] < 0 else (contig[pos[j] - 1:pos[j] - 1 + rl[j, t]], letters[off[j * 3 + t]:off[j * 3 + t + 1]]) for t in range(3)] for j in sel]
        with np.errstate(all="ignore"):
            lines, _ = indelCaller.indel_vcf_lines("chrT", pos[sel].tolist(), probs[sel], alle, [int(p) if p else None for p in phase[sel]], 0)
        exp += lines
    out = np.empty(n * 400, np.uint8)
    nb = C.c_int64()
    rc = L.nc_indel_vcf_format(b"chrT", n, _lib.npp(pos), _lib.npp(chunk), n_chunks, _lib.npp(probs), 3, _lib.npp(rl), _lib.npp(al), _lib.npp(alt),
                               _lib.npp(phase), contig.encode(), Lc, 0, _lib.npp(out), out.size, C.byref(nb), None)
    assert rc == _lib.NC_OK
    got = out[:nb.value].tobytes().decode()
    assert got == "".join(exp) and "nan" in got


# ---------------------------------------------------------------------------------------------------------------- GPU
def _params(fa, **kw):
    d = dict(seq="ont", fasta_path=fa, win_size=40, small_win_size=4, mincov=4, maxcov=160, ins_t=0.4, del_t=0.6, supplementary=False,
             exclude_bed=None, impute_indel_phase=False)
    d.update(kw)
    return d


def _same_tuples(got, exp):
    assert len(got) == len(exp)
    n = 0
    for t, e in zip(got, exp):
        assert list(t[0]) == list(e[0])
        n += len(e[0])
        for a, b in zip(t[1:], e[1:]):
            if isinstance(b, np.ndarray):
                assert np.asarray(a).dtype == b.dtype and np.array_equal(np.asarray(a), b)
            else:
                assert list(a) == list(b)
    return n


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["ont", "ont_mincov2_maxcov9", "maxcov300", "pacbio_window", "haploid", "small_chunks", "excluded"])
def test_device_pipeline_returns_the_host_routes_tuples(world_files, variant, monkeypatch):
    w, bam, fa = world_files
    kw, haploid, step = {}, False, 50_000
    if variant == "ont_mincov2_maxcov9":
        kw = dict(mincov=2, maxcov=9, del_t=0.4)                                    # the first-maxcov policy bites: sets are cut
    elif variant == "maxcov300":
        kw = dict(maxcov=300)                                                        # the tensor kernel's 16-bit histogram form (bytes up to 255)
    elif variant == "pacbio_window":
        kw = dict(seq="pacbio", ins_t=0.4, del_t=0.4)                               # window_after 260: the 17-column aligner
    elif variant == "haploid":
        haploid = True
    elif variant == "small_chunks":
        step = 3_000                                                                 # anchors shared by adjacent chunks
    elif variant == "excluded":
        kw = dict(exclude_bed=[(w.chrom, 20_000, 60_000), (w.chrom, 100_000, 100_500)])
    dct = _params(fa, **kw)
    chunks = [dict(chrom=w.chrom, start=s, end=min(w.length, s + step), sam_path=bam) for s in range(1, w.length, step)]
    gip._CONTIGS.clear()
    monkeypatch.setenv("NC_INDEL_HOST_PASS2", "1")
    exp = gip.get_indel_testing_candidates_batch(dct, chunks, haploid=haploid)
    monkeypatch.delenv("NC_INDEL_HOST_PASS2")

    def boom(*a, **k):
        raise AssertionError("the host-assembled route ran where the device pipeline was expected")
    monkeypatch.setattr(gip, "_pass2_native", boom)
    # the host-assembled route aligns on the full matrix: the device pipeline with its band switched off returns the same tuples ...
    monkeypatch.setenv("NC_PIPE_BAND", "0")
    got = gip.get_indel_testing_candidates_batch(dct, chunks, haploid=haploid)
    n_sites = _same_tuples(got, exp)
    assert n_sites > (100 if variant != "excluded" else 50)
    # ... and on the band (the default) the same sites, with the tensors and alleles of all but a few sites identical (a banded alignment is the
    # full-matrix one unless the optimal path leaves the band without the banded path touching its edge: non-homologous stretches, read ends)
    monkeypatch.delenv("NC_PIPE_BAND")
    band = gip.get_indel_testing_candidates_batch(dct, chunks, haploid=haploid)
    same = 0
    for t, e in zip(band, exp):
        assert list(t[0]) == list(e[0])
        for k in range(len(e[0])):
            same += all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) if isinstance(b, np.ndarray) else a[k] == b[k] for a, b in zip(t[1:], e[1:]))
    assert same >= 0.98 * n_sites, (same, n_sites)


@pytest.mark.gpu
def test_impute_indel_phase_splits_the_chunks_between_the_device_pipeline_and_the_host_route(tmp_path, monkeypatch):
    """dct['impute_indel_phase'] (generate_indel_pileups.py:278-304): chunks without a column that meets the rule's predicate run on the device
    pipeline, the others on the host-assembled route (their read grouping needs the pileup strings): the same tuples, and the same indel_run text,
    as everything on the host-assembled route"""
    import queue
    w = bamio.make_pass2_world(seed=23, length=120_000, depth=24, blocks=[(30_000, 60_000)])
    bam, fa = str(tmp_path / "i.bam"), str(tmp_path / "i.fa")
    bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None))
    bamio.write_fasta(fa, w.chrom, w.ref)
    dct = _params(fa, impute_indel_phase=True, mincov=3, ins_t=0.3, del_t=0.3)
    chunks = [dict(chrom=w.chrom, start=s, end=min(w.length, s + 10_000), sam_path=bam) for s in range(1, w.length, 10_000)]
    gip._CONTIGS.clear()
    monkeypatch.setenv("NC_PIPE_BAND", "0")                                          # the host-assembled route aligns on the full matrix
    monkeypatch.setenv("NC_IMPUTE_SPLIT", "0")
    exp = gip.get_indel_testing_candidates_batch(dct, chunks)
    monkeypatch.delenv("NC_IMPUTE_SPLIT")
    monkeypatch.setenv("NC_IMPUTE_DEVICE", "0")                                      # (round 5's route; the default groups on the device: next test)
    dev, host = gip.impute_split_chunks(dct, chunks, 0, False)
    assert len(dev) >= 4 and len(host) >= 2 and sorted(dev + host) == list(range(len(chunks)))
    assert all(30_000 - 10_000 < chunks[k]["start"] < 60_000 + 200 for k in host), [chunks[k]["start"] for k in host]
    calls = []
    real = gip._indel_batch_device
    monkeypatch.setattr(gip, "_indel_batch_device", lambda d, c, *a: (calls.append(len(c)), real(d, c, *a))[1])
    got = gip.get_indel_testing_candidates_batch(dct, chunks)
    assert calls == [len(dev)]
    n = _same_tuples(got, exp)
    assert n > 60
    # imputed anchors exist and sit in the host route's chunks only
    from oracle import oracle
    ev, ex = oracle.indel_scan_impute(w, 28_000, 62_000, mincov=3, win_size=40, small_win_size=4, ins_t=0.3, del_t=0.3)
    assert len(ex) > 3 and all(any(chunks[k]["start"] - 60 <= a <= chunks[k]["end"] for k in host) for a in ex)
    # indelCaller.indel_run: the same file either way
    outs = []
    for tag, split in (("host", "0"), ("split", None)):
        if split:
            monkeypatch.setenv("NC_IMPUTE_SPLIT", split)
        else:
            monkeypatch.delenv("NC_IMPUTE_SPLIT", raising=False)
        d = tmp_path / tag
        d.mkdir()
        params = _params(fa, impute_indel_phase=True, mincov=3, ins_t=0.3, del_t=0.3, indel_model="ONT-HG002", intermediate_indel_files_dir=str(d), prefix="t")
        jobs = queue.Queue()
        for c in chunks:
            jobs.put(("indel", dict(c, ploidy="diploid")))
        outs.append(open(indelCaller.indel_run(params, {}, jobs, queue.Queue(), [], aligner="device")).read())
    assert outs[0] == outs[1] and outs[0].count("\n") > 40


@pytest.mark.gpu
def test_impute_indel_phase_on_the_device_pipeline_equals_the_host_assembled_route(tmp_path, monkeypatch):
    """[r6] dct['impute_indel_phase'] (generate_indel_pileups.py:278-304) entirely on the device pipeline: K7's col_type-2 columns grouped by pileup
    string in HBM (k_impute_flags), the imputed anchors' read sets taken by k_sets<.., true> where it takes the HP tags otherwise -- the tuples and
    the indel_run text of the host-assembled route (whose grouping, gip.impute_groups on the column strings, the reference-captured golden pins:
    tests/test_indel_pass2.py, test_gpu_parity.py::test_impute_indel_phase_scan_matches_reference_pass1)"""
    import queue
    w = bamio.make_pass2_world(seed=23, length=120_000, depth=24, blocks=[(30_000, 60_000)])
    bam, fa = str(tmp_path / "i.bam"), str(tmp_path / "i.fa")
    bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None))
    bamio.write_fasta(fa, w.chrom, w.ref)
    from oracle import oracle
    for kw in (dict(mincov=3, ins_t=0.3, del_t=0.3), dict(mincov=6, ins_t=0.25, del_t=0.25), dict(mincov=2, ins_t=0.4, del_t=0.5, maxcov=12)):
        dct = _params(fa, impute_indel_phase=True, **kw)
        chunks = [dict(chrom=w.chrom, start=s, end=min(w.length, s + 10_000), sam_path=bam) for s in range(1, w.length, 10_000)]
        gip._CONTIGS.clear()
        monkeypatch.setenv("NC_PIPE_BAND", "0")                                      # the host-assembled route aligns on the full matrix
        monkeypatch.setenv("NC_IMPUTE_SPLIT", "0")
        exp = gip.get_indel_testing_candidates_batch(dct, chunks)
        monkeypatch.delenv("NC_IMPUTE_SPLIT")
        real = gip._pass2_native

        def boom(*a, **k):
            raise AssertionError("the host-assembled route ran where the device pipeline was expected")
        monkeypatch.setattr(gip, "_pass2_native", boom)
        got = gip.get_indel_testing_candidates_batch(dct, chunks)
        monkeypatch.setattr(gip, "_pass2_native", real)
        n = _same_tuples(got, exp)
        assert n > 40, (kw, n)
        # imputed anchors exist among them (the unphased block), and so do sites outside it
        ev, ex = oracle.indel_scan_impute(w, 28_000, 62_000, mincov=kw["mincov"], win_size=40, small_win_size=4, ins_t=kw["ins_t"], del_t=kw["del_t"])
        got_pos = {int(p) for t in got for p in t[0]}
        assert len(ex) > 3 and len(got_pos & set(ex)) >= 1 and any(p < 25_000 or p > 65_000 for p in got_pos), (kw, len(ex))
    # round 5's split (NC_IMPUTE_DEVICE=0) still gives the same
    monkeypatch.setenv("NC_IMPUTE_DEVICE", "0")
    assert _same_tuples(gip.get_indel_testing_candidates_batch(dct, chunks), exp) == n
    monkeypatch.delenv("NC_IMPUTE_DEVICE")
    # indelCaller.indel_run: the same file either way
    outs = []
    for tag, split in (("host", "0"), ("device", None)):
        if split:
            monkeypatch.setenv("NC_IMPUTE_SPLIT", split)
        else:
            monkeypatch.delenv("NC_IMPUTE_SPLIT", raising=False)
        d = tmp_path / tag
        d.mkdir()
        params = _params(fa, impute_indel_phase=True, mincov=3, ins_t=0.3, del_t=0.3, indel_model="ONT-HG002", intermediate_indel_files_dir=str(d), prefix="t")
        jobs = queue.Queue()
        for c in chunks:
            jobs.put(("indel", dict(c, ploidy="diploid")))
        outs.append(open(indelCaller.indel_run(params, {}, jobs, queue.Queue(), [], aligner="device")).read())
    assert outs[0] == outs[1] and outs[0].count("\n") > 40


@pytest.mark.gpu
def test_device_pipeline_in_small_groups_is_the_same(world_files, monkeypatch):
    """the run loop cut into many groups of sites (traceback workspace bound) gives the same arrays"""
    w, bam, fa = world_files
    dct = _params(fa)
    chunks = [dict(chrom=w.chrom, start=s, end=min(w.length, s + 50_000), sam_path=bam) for s in range(1, w.length, 50_000)]
    a = gip.get_indel_testing_candidates_batch(dct, chunks)
    monkeypatch.setenv("NC_PIPE_GROUP_AL", "700")
    b = gip.get_indel_testing_candidates_batch(dct, chunks)
    assert _same_tuples(b, a) > 100


@pytest.mark.gpu
def test_device_ingest_feeds_the_indel_pipeline_the_host_ingests_inputs(world_files, monkeypatch):
    """the contig's pack + events + inserted bases + tails made in HBM from the BAM file (device_bam.py) or on host threads (nc_bam_decode,
    nc_indel_pack_build, wire): the same tuples; the default takes the device"""
    from nanocaller_amd import generate_SNP_pileups as gsp
    w, bam, fa = world_files
    dct = _params(fa)
    chunks = [dict(chrom=w.chrom, start=s, end=min(w.length, s + 50_000), sam_path=bam) for s in range(1, w.length, 50_000)]
    gsp.release_contig()
    gip._CONTIGS.clear()
    del gsp.DECODES[:]
    a = gip.get_indel_testing_candidates_batch(dct, chunks)
    assert len(gip._DEV_INGEST) == 1 and gsp.DECODES == [] and not gip._CONTIGS        # nothing was decoded on the host
    monkeypatch.setenv("NC_DEVICE_INGEST", "0")
    gsp.release_contig()
    b = gip.get_indel_testing_candidates_batch(dct, chunks)
    assert not gip._DEV_INGEST and gip._CONTIGS
    assert _same_tuples(b, a) > 100


@pytest.mark.gpu
def test_indel_run_native_text_equals_the_python_rules(world_files, tmp_path, monkeypatch):
    """indelCaller.indel_run on the device pipeline + nc_indel_vcf_format writes the file the tuple route + Python rules write"""
    import queue
    w, bam, fa = world_files
    outs = []
    for tag, env in (("py", "1"), ("native", None)):
        if env:
            monkeypatch.setenv("NC_INDEL_PY_RULES", env)
        else:
            monkeypatch.delenv("NC_INDEL_PY_RULES", raising=False)
        d = tmp_path / tag
        d.mkdir()
        params = _params(fa, indel_model="ONT-HG002", intermediate_indel_files_dir=str(d), prefix="t")
        jobs = queue.Queue()
        for s in range(1, w.length, 40_000):
            jobs.put(("indel", dict(chrom=w.chrom, start=s, end=min(w.length, s + 40_000), ploidy="diploid" if s < 100_000 else "haploid", sam_path=bam)))
        outs.append(open(indelCaller.indel_run(params, {}, jobs, queue.Queue(), [], aligner="device")).read())
    assert outs[0] == outs[1] and outs[0].count("\n") > 60


def _host_sample(pack, reads_c, info, r1):
    """host copies of the first r1 reads of a synthetic device workload, as callables for oracle.records_from_indel_pack"""
    s, e = info["read_start"][:r1], info["read_end"][:r1]
    slot = pack.reads["slot_off"][:r1 + 1].cpu().numpy()
    codes = pack.codes[:int(slot[-1])].cpu().numpy()
    ev_off = pack.events["ev_off"][:r1 + 1].cpu().numpy()
    ev_pos = pack.events["ev_pos"][:int(ev_off[-1])].cpu().numpy()
    ev_len = pack.events["ev_len"][:int(ev_off[-1])].cpu().numpy()
    ins_off = info["tensors"]["ins_off"][:int(ev_off[-1]) + 1].cpu().numpy()
    ins = info["tensors"]["ins_bases"][:int(ins_off[-1])].cpu().numpy()

    def codes_of(r):
        o = int(slot[r]) + (int(s[r]) & 15)
        return codes[o:o + int(e[r] - s[r])]

    def ev_of(r):
        return list(zip(ev_pos[ev_off[r]:ev_off[r + 1]].tolist(), ev_len[ev_off[r]:ev_off[r + 1]].tolist()))

    def ins_of(r, k):
        a = int(ev_off[r]) + k
        return ins[ins_off[a]:ins_off[a + 1]]
    return s, e, codes_of, ev_of, ins_of


@pytest.mark.gpu
def test_device_pipeline_on_the_synthetic_workload_equals_the_oracle_restatement(tmp_path, monkeypatch):
    """the bench workload (generated in HBM, no BAM behind it): sites, tensors, consensus-derived alleles and phase of the device
    pipeline against pass 2 restated from SAM-like records (CIGAR expansion base by base, oracle.read_windows_ref), the star
    alignment in PURE PYTHON for every site checked -- banded exactly as the device runs it (oracle.window_band_ref -> band_of ->
    nw_cigar_band_free_tail_ref, full matrix after an edge touch or for wide bands) --, msa() by the C oracle; no product aligner involved"""
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import make_indel_device_workload
    from oracle import oracle
    eng = get_engine(0)
    L = 260_000
    pack, reads_c, info = make_indel_device_workload(eng, L, depth=28.0, seed=99)
    chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
    kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=160)
    monkeypatch.setenv("NC_PIPE_DUMP", str(tmp_path / "d"))                      # the run's per-alignment arrays as files (a debugging aid of the library)
    r = gip.indel_sites_device(eng, pack, reads_c, L, chunks, **kw)
    monkeypatch.delenv("NC_PIPE_DUMP")
    assert r["n"] > 80
    x = r["x"].cpu().numpy()
    hi = 100_000
    r1 = int(np.searchsorted(info["read_start"], hi + 400))
    s, e, codes_of, ev_of, ins_of = _host_sample(pack, reads_c, info, r1)
    recs = oracle.records_from_indel_pack(s, e, codes_of, ev_of, ins_of)
    ref = np.frombuffer(b"AGTCN", np.uint8)[info["tensors"]["ref"].cpu().numpy()[1:]].tobytes().decode()
    masked = pack.ref_code[1:L + 1].cpu().numpy() == 4
    ref = "".join(c.lower() if m else c for c, m in zip(ref[:hi + 400], masked[:hi + 400]))      # soft-masked runs: not upper-case AGTC
    alt_all = np.frombuffer(b"AGTCN", np.uint8)[r["alt"]].tobytes().decode()
    aoff = np.zeros(r["n"] * 3 + 1, np.int64)
    np.cumsum(np.maximum(r["alt_len"].reshape(-1), 0), out=aoff[1:])
    checked = 0
    how = []
    for k in range(r["n"]):
        p = int(r["pos"][k])
        if p > hi:
            break
        got = oracle.indel_site_ref(recs, info["hap"], info["ps"], ref, p, 160, 4, 160, band=True, how_out=how)
        assert got is not None, p
        xs, cns, win, phase = got
        assert np.array_equal(x[k].reshape(3, 5, 128, 2), xs), p
        assert phase == int(r["phase"][k])
        mr = 40 if r["type"][k] == 0 else 10
        for t in range(3):
            exp = gip.allele_prediction(cns[t], win, mr)
            rl, al = int(r["ref_len"][k, t]), int(r["alt_len"][k, t])
            have = (None, None) if rl < 0 else (win[:rl], alt_all[aoff[k * 3 + t]:aoff[k * 3 + t] + al])
            assert have == exp, (p, t)
        checked += 1
    assert checked >= 32
    assert how.count(32) > 0.7 * len(how) and how.count(64) > 0, (how.count(32), how.count(64), len(how))      # both band widths met
    # the band of every alignment of those sites, as the device derived it from the events of the read, against the CIGAR walk of the oracle
    d = {n: np.fromfile(str(tmp_path / ("d." + n)), dt) for n, dt in (("al_site", np.int32), ("al_read", np.int32), ("band_lo", np.int8), ("site_pos", np.int32), ("n1", np.int32))}
    nb = 0
    for a in range(len(d["al_site"])):
        p = int(d["site_pos"][d["al_site"][a]])
        if p > hi:
            break
        b = oracle.band_of(*oracle.window_band_ref(recs[int(d["al_read"][a])], p, 160), int(d["n1"][a]), min(161, L - p + 1))
        assert int(d["band_lo"][a]) == (b[0] if b else 0), (a, p, b)
        nb += 1
    assert nb > 1000
    # and no site of the oracle's is missing: every anchor position the device kept in the sample range passes the oracle's set tests (above);
    # positions are unique per chunk and ascending
    pos0 = r["pos"][r["chunk"] == 0]
    assert np.all(np.diff(pos0) > 0)


@pytest.mark.gpu
def test_device_pipeline_at_scale_is_deterministic_and_group_invariant(monkeypatch):
    """8 Mb of the bench workload (5 k candidate sites, 130 k read windows): two runs give the same bytes, and so does the run loop cut into
    ~40 groups of alignments (two streams, two buffer sets handed back and forth by events: a race there shows up only with many groups)"""
    import torch
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import make_indel_device_workload
    eng = get_engine(0)
    L = 8_000_000
    pack, reads_c, info = make_indel_device_workload(eng, L, depth=30.0, seed=4242)
    chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
    kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=160)
    keys = ("pos", "chunk", "type", "phase", "ref_len", "alt_len", "alt")

    def run():
        r = gip.indel_sites_device(eng, pack, reads_c, L, chunks, **kw)
        return {k: np.array(r[k]) for k in keys}, r["x"].clone(), r["n"], r["n_alignments"]
    a, xa, n, nal = run()
    assert n > 3000 and nal > 80_000
    b, xb, _, _ = run()
    monkeypatch.setenv("NC_PIPE_GROUP_AL", str(max(1000, nal // 40)))
    c, xc, _, _ = run()
    for other, xo in ((b, xb), (c, xc)):
        for k in keys:
            assert np.array_equal(other[k], a[k]), k
        assert torch.equal(xo, xa)
    # the tensors' frequencies: every used column of every set sums to one over the five symbols (msa(), :57-71), before the reference one-hot is taken off
    x = xa.view(n, -1, 5, 128, 2)
    tot = (x[..., 0] + x[..., 1]).sum(dim=2)
    used = x[..., 1].sum(dim=2) > 0
    assert used.any() and float((tot[used] - 1.0).abs().max()) < 1e-5


@pytest.mark.gpu
def test_a_contig_of_one_group_after_a_contig_of_several_allocates_nothing(monkeypatch):
    """nc_indel_sites_run cuts a contig with more alignments than the group bound into groups of EQUAL size, all smaller than the bound; a later,
    shorter contig may fit ONE group that is larger than any of those.  A run of several groups therefore sizes the per-group buffers for the bound:
    the second contig must not make the library allocate (sized by the group at hand they grew there -- a hipFree + hipMalloc of 9 GB at chr18 of a
    whole-genome pass, a second of idle GPU: profiles/README.md, round 6).  A context of its own: the shared one's buffers are as large as the
    largest test before this one left them."""
    import torch
    from nanocaller_amd.engine import Engine, get_engine
    from nanocaller_amd.synth_device import make_indel_device_workload
    shared = get_engine(0)
    kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=160)

    def contig(L, seed):
        pack, reads_c, info = make_indel_device_workload(shared, L, depth=30.0, seed=seed)
        return pack, reads_c, L, [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)], info      # (info keeps the device arrays alive)

    def library_bytes():
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        return (total - free) - torch.cuda.memory_reserved()

    a, b = contig(8_000_000, 4242), contig(6_900_000, 4243)
    # the alignments of both contigs, counted on the shared context (whatever its buffers hold)
    na = gip.indel_sites_device(shared, a[0], a[1], a[2], a[3], fetch=False, **kw)["n_alignments"]
    nb = gip.indel_sites_device(shared, b[0], b[1], b[2], b[3], fetch=False, **kw)["n_alignments"]
    bound = int(0.92 * na)
    # A: two groups of na / 2 (+ the allocator's 25 % = 0.625 na); B: one group, more than that, within the bound
    assert 0.7 * na < nb <= bound, (na, nb)
    monkeypatch.setenv("NC_PIPE_GROUP_AL", str(bound))
    eng = Engine(0)
    try:
        ra = gip.indel_sites_device(eng, a[0], a[1], a[2], a[3], fetch=False, **kw)
        assert ra["n_alignments"] == na
        del ra
        before = library_bytes()
        rb = gip.indel_sites_device(eng, b[0], b[1], b[2], b[3], fetch=False, **kw)
        assert rb["n_alignments"] == nb
        del rb
        grown = library_bytes() - before
        assert grown < (64 << 20), "the second contig made the library allocate %.1f MB" % (grown / 1e6)
    finally:
        torch.cuda.synchronize()
        eng.close()
        shared.use_torch_stream()


@pytest.mark.gpu
def test_banded_star_alignment_against_the_full_matrix(monkeypatch):
    """the default (every read window aligned on the 32 / 64 diagonals its own CIGAR allows, full matrix after an edge touch) against the
    full matrix for every window: same sites, and all but a few per ten thousand tensors / alleles identical; most windows fit 32 diagonals"""
    import torch
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import make_indel_device_workload
    eng = get_engine(0)
    L = 8_000_000
    pack, reads_c, info = make_indel_device_workload(eng, L, depth=30.0, seed=777)
    chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
    kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=160)
    res = {}
    for mode in (0, 1):
        assert eng.L.nc_indel_sites_band(eng.ctx, mode, 0) == 0
        r = gip.indel_sites_device(eng, pack, reads_c, L, chunks, **kw)
        st = np.zeros(6, np.int64)
        eng.L.nc_indel_sites_band_stats(eng.ctx, _lib.npp(st))
        res[mode] = (r, st)
    assert eng.L.nc_indel_sites_band(eng.ctx, -1, 0) == 0
    (f, sf), (b, sb) = res[0], res[1]
    assert not sf[:4].any() and int(sb[:3].sum()) == b["n_alignments"] == f["n_alignments"]
    assert sb[0] > 0.85 * b["n_alignments"] and sb[1] > 0 and sb[2] < 0.02 * b["n_alignments"] and sb[3] < 0.001 * b["n_alignments"]
    assert f["n"] == b["n"] > 3000
    for k in ("pos", "chunk", "type", "phase"):
        assert np.array_equal(np.asarray(f[k]), np.asarray(b[k])), k
    dx = int((f["x"] != b["x"]).reshape(f["n"], -1).any(1).sum())
    dal = int(((f["ref_len"] != b["ref_len"]) | (f["alt_len"] != b["alt_len"])).any(1).sum())
    assert dx <= 0.002 * f["n"] and dal <= 0.002 * f["n"], (dx, dal, f["n"])


@pytest.mark.gpu
def test_tiled_event_windows_equal_the_atomic_form(monkeypatch):
    """pass 1 of the device pipeline counts the reads with an indel event in each window in LDS, one workgroup per 1024 columns
    (k_event_tiles: margins, clipped interval ends, reads listed in the previous tile); NC_K7_EVENT_ATOMICS=1 keeps the form the host
    route runs (k_event_intervals_w + k_prefix_rows_b, pinned by the golden tests): same sites, same types, same tensors"""
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import make_indel_device_workload
    eng = get_engine(0)
    L = 2_600_000
    pack, reads_c, info = make_indel_device_workload(eng, L, depth=30.0, seed=4711)
    # chunks that do not sit on the tile grid, one shorter than a workgroup's 1024 columns
    cuts = [1, 70_123, 70_900, 171_555, 300_001] + list(range(400_000, L, 100_000)) + [L]
    chunks = [(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=160)
    import torch
    # excluded stretches (ranks skip them): a few columns, most of a workgroup's block, several tiles, and one that ends right where a
    # block begins (its margin lies entirely before the stretch: the walk back over the rank array)
    excl = torch.zeros(pack.ref_code.numel(), dtype=torch.uint8, device=pack.ref_code.device)
    g0 = pack.tile_pos0
    for a, b in ((500_010, 500_020), (640_100, 640_800), (901_000, 907_500), (1_200_000, 1_200_000 + 4096 - (1_200_000 - g0) % 1024)):
        excl[a - g0:b - g0] = 1
    for variant in (dict(), dict(excl=excl), dict(excl=excl, haploid=True)):
        monkeypatch.delenv("NC_K7_EVENT_ATOMICS", raising=False)
        new = gip.indel_sites_device(eng, pack, reads_c, L, chunks, **kw, **variant)
        monkeypatch.setenv("NC_K7_EVENT_ATOMICS", "1")
        old = gip.indel_sites_device(eng, pack, reads_c, L, chunks, **kw, **variant)
        assert new["n"] == old["n"] and new["n"] > 300, variant.keys()
        for key in ("pos", "chunk", "type", "phase", "ref_len", "alt_len"):
            assert np.array_equal(np.asarray(new[key]), np.asarray(old[key])), (key, variant.keys())
        assert bool((new["x"] == old["x"]).all())
        if "excl" in variant:
            pos = np.asarray(new["pos"])
            assert not np.any((pos >= 901_000) & (pos < 907_500))
        # the columns' decisions: threshold tables for depths below 1024 (k_decide_tables), the reference's float64 divide-and-compare beyond --
        # with the tables cut to depths < 12 most columns of this 30x pileup take the division form: same sites
        monkeypatch.delenv("NC_K7_EVENT_ATOMICS", raising=False)
        monkeypatch.setenv("NC_K7_DEC_N", "12")
        cut = gip.indel_sites_device(eng, pack, reads_c, L, chunks, **kw, **variant)
        monkeypatch.delenv("NC_K7_DEC_N")
        for key in ("pos", "chunk", "type", "phase", "ref_len", "alt_len"):
            assert np.array_equal(np.asarray(cut[key]), np.asarray(new[key])), (key, variant.keys())


def _window_dump(tmp_path, tag):
    """the per-alignment arrays NC_PIPE_DUMP left: rows cut to their lengths"""
    d = {n: np.fromfile(str(tmp_path / (tag + "." + n)), dt) for n, dt in (("n1", np.int32), ("band_lo", np.int8), ("al_read", np.int32), ("al_site", np.int32), ("win", np.uint8))}
    A = len(d["n1"])
    ws = len(d["win"]) // A
    win = d["win"].reshape(A, ws)
    rows = [win[a, :d["n1"][a]].tobytes() for a in range(A)]
    return d, rows


@pytest.mark.gpu
@pytest.mark.parametrize("window_after", [160, 260])
def test_windows_by_16_lanes_equal_the_serial_walk(tmp_path, monkeypatch, window_after):
    """k_windows16 (four windows per wave: prefix sums over the read's events, 16-column groups) against the one-lane walk it replaces
    (NC_PIPE_WINDOWS=serial: the round-3 kernel; force16: the new kernel with every window on its serial route): bases, lengths and bands of
    every window, on the synthetic workload (planted indels up to 50 bases, 4 % deletions) at both window lengths"""
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import make_indel_device_workload
    eng = get_engine(0)
    L = 1_500_000
    pack, reads_c, info = make_indel_device_workload(eng, L, depth=30.0, seed=31 + window_after)
    chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
    kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=window_after)
    got = {}
    for mode in ("default", "serial", "force16"):
        if mode == "default":
            monkeypatch.delenv("NC_PIPE_WINDOWS", raising=False)
        else:
            monkeypatch.setenv("NC_PIPE_WINDOWS", mode)
        monkeypatch.setenv("NC_PIPE_DUMP", str(tmp_path / mode))
        r = gip.indel_sites_device(eng, pack, reads_c, L, chunks, **kw)
        monkeypatch.delenv("NC_PIPE_DUMP")
        got[mode] = (_window_dump(tmp_path, mode), r)
    monkeypatch.delenv("NC_PIPE_WINDOWS", raising=False)
    (d0, rows0), r0 = got["serial"]
    assert len(rows0) > 10_000 and int((d0["n1"] < window_after).sum()) > 0 and int((d0["band_lo"] != d0["band_lo"][0]).sum()) > 0
    for mode in ("default", "force16"):
        (d, rows), r = got[mode]
        for k in ("n1", "band_lo", "al_read", "al_site"):
            assert np.array_equal(d[k], d0[k]), (mode, k, int((d[k] != d0[k]).sum()))
        bad = [a for a in range(len(rows0)) if rows[a] != rows0[a]]
        assert not bad, (mode, len(bad), bad[:5])
        assert r["n"] == r0["n"] and bool((r["x"] == r0["x"]).all())
        for k in ("pos", "ref_len", "alt_len"):
            assert np.array_equal(np.asarray(r[k]), np.asarray(r0[k])), (mode, k)


@pytest.mark.gpu
def test_banded_allele_alignments_certify_themselves_or_run_on_the_full_matrix(monkeypatch):
    """allele_prediction's global alignment (exact in the reference: parasail nw_trace, generate_indel_pileups.py:79) runs on a band only where the
    band proves itself: the score of the banded path must exceed what any path that leaves the band can reach (allele_trace_body's bound), else
    the set is re-run on the full matrix.  Every REF / ALT of 8 Mb of the bench workload equals the all-full-matrix run (NC_PIPE_BAND_ALLELES=0),
    with the STAR alignment on its default band in both runs (only NC_PIPE_BAND_ALLELES differs), so that the consensus strings the alleles are read
    from are the same"""
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.synth_device import make_indel_device_workload
    eng = get_engine(0)
    L = 8_000_000
    pack, reads_c, info = make_indel_device_workload(eng, L, depth=30.0, seed=515)
    chunks = [(s, min(L, s + 100_000)) for s in range(1, L, 100_000)]
    kw = dict(mincov=4, maxcov=160, win_size=40, small_win_size=4, ins_t=0.4, del_t=0.6, window_after=160)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NC_PIPE_BAND_ALLELES", mode)
        res[mode] = gip.indel_sites_device(eng, pack, reads_c, L, chunks, **kw)
    monkeypatch.delenv("NC_PIPE_BAND_ALLELES")
    a, b = res["1"], res["0"]
    assert a["n"] == b["n"] > 3000
    for k in ("pos", "type", "ref_len", "alt_len", "alt"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    assert bool((a["x"] == b["x"]).all())
    assert int((np.asarray(a["ref_len"]) > 0).sum()) > 1000                                   # (alleles were called at all)
