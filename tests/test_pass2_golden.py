"""SURVEY.md 8a rows a11 / a13 (and a10 + a12 on the way) against the REFERENCE's own outputs: tests/golden/indel_pass2.npz
holds the full 6-tuple of get_indel_testing_candidates (generate_indel_pileups.py:129-371) and the 3-tuple of
get_indel_testing_candidates_haploid (generate_indel_pileups_haploid.py:128-277) as returned by the reference's code run in
the build container (oracle/tools/make_goldens.py pass2: stub pysam serving SAM-like records, MUSCLE answered by the star
aligner, parasail by the Gotoh aligner); tests/golden/allele_prediction.json holds what the reference's allele_prediction
(:77-127) returned for 801 (alt, ref_seq, max_range) triples.  The library's native code must reproduce them exactly."""
import json
import os

import numpy as np
import pytest

from nanocaller_amd import generate_indel_pileups as gip
from nanocaller_amd.bam import BamFile, read_bam
from oracle import oracle

import bamio

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    z = np.load(os.path.join(GOLD, "indel_pass2.npz"))
    out = []
    for k in range(int(z["n"])):
        dct = json.loads(str(z["c%d_dct" % k]))
        n = len(z["c%d_pos" % k])
        xs = [z["c%d_x%d" % (k, i)] for i in range(3) if ("c%d_x%d" % (k, i)) in z]
        alleles = json.loads(str(z["c%d_alleles" % k]))
        out.append(dict(k=k, world=str(z["c%d_world" % k]), ploidy=str(z["c%d_ploidy" % k]), start=int(z["c%d_start" % k]),
                        end=int(z["c%d_end" % k]), dct=dct, pos=z["c%d_pos" % k].tolist(), xs=xs, n=n,
                        alleles=alleles, phase=json.loads(str(z["c%d_phase" % k]))))
    return z, out


Z, CASES = _cases()


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("pass2gold")
    out = {}
    for wn in ("a", "b"):
        w = bamio.world_from_arrays(Z, "w%s_" % wn)
        bam, fa = str(d / ("%s.bam" % wn)), str(d / ("%s.fa" % wn))
        bamio.write_bam(bam, w.chrom, w.length, bamio.world_to_records(w, None))
        bamio.write_fasta(fa, w.chrom, w.ref)
        out[wn] = (w, bam, fa)
    return out


def _norm_alleles(a):
    """JSON turned the (REF, ALT) tuples into lists"""
    if a and isinstance(a[0], (list, tuple)) and len(a[0]) == 3 and isinstance(a[0][0], (list, tuple)):
        return [[tuple(x) for x in site] for site in a]
    return [tuple(x) for x in a]


def test_allele_prediction_equals_the_reference_outputs():
    """nc_allele_prediction (+ batch) against what the reference's own allele_prediction returned"""
    rec = json.load(open(os.path.join(GOLD, "allele_prediction.json")))
    calls = rec["calls"]
    assert len(calls) > 700 and rec["n_pipeline"] > 400
    n_alt = 0
    for alt, ref, mr, r, a in calls:
        assert not (isinstance(r, str) and r.startswith("!"))
        assert gip.allele_prediction(alt, ref, mr) == (r, a), (alt, ref, mr)
        n_alt += r is not None
    assert n_alt > 200
    got = gip.allele_prediction_batch([c[0] for c in calls], [c[1] for c in calls], [c[2] for c in calls])
    assert got == [(c[3], c[4]) for c in calls]


def _assemble(case, w, bam, fa):
    """pass 2 assembled from the CPU oracle (pass 1, K8) and the library's host code (BAM windows, star alignment, allele
    strings): what the reference's tuple must equal without any GPU"""
    dct = case["dct"]
    world = read_bam(bam, fa, w.chrom)
    kw = dict(mincov=dct["mincov"], win_size=dct["win_size"], small_win_size=dct["small_win_size"], ins_t=dct["ins_t"],
              del_t=dct["del_t"], supplementary=dct["supplementary"])
    hapl = case["ploidy"] == "haploid"
    extra = {}
    if dct["impute_indel_phase"] and not hapl:
        variants, extra_idx = oracle.indel_scan_impute(w, case["start"], case["end"], **kw)
        extra = {p: tuple([w.names[i] for i in side] for side in sets) for p, sets in extra_idx.items()}
    else:
        vp, vt = oracle.indel_scan(world, case["start"], case["end"], haploid=hapl, **kw)
        variants = dict(zip(vp.tolist(), vt.tolist()))
    wa = 260 if dct["seq"] == "pacbio" else 160
    flag = 0x4 | 0x100 | 0x200 | 0x400 | (0 if dct["supplementary"] else 0x800)
    anchors = sorted(v for v in variants if max(0, case["start"] - 10 - dct["win_size"]) < v <= case["end"])
    bf = BamFile(bam)
    d = bf.decode(w.chrom, max(1, case["start"] - 100000), case["end"] + 1000, anchors=anchors, window_before=0, window_after=wa,
                  keep_mask=flag)
    bf.close()
    names, hap, ps = d["names"], d["hap"], d["ps"]
    sym = {"A": 0, "G": 1, "T": 2, "C": 3, "-": 4}
    lo, hi = max(1, case["start"] - 200), case["end"] + 400
    pos, xs, alleles, phase = [], [], [], []
    for a, win in zip(anchors, d["windows"]):
        ref = "".join((w.ref[p - 1] if (lo <= p <= hi and w.ref[p - 1] in "AGTC") else "N") for p in range(a, min(w.length, a + wa + 1)))
        if "N" in ref:
            continue
        sets = [{}, {}, {}]
        imp = extra.get(a)
        for r, text in win:
            sets[2][names[r]] = text
            if (names[r] in imp[0]) if imp else hap[r] == 1:
                sets[0][names[r]] = text
            elif (names[r] in imp[1]) if imp else hap[r] == 2:
                sets[1][names[r]] = text
        todo = [(sets[2], dct["mincov"])] if hapl else [(sets[0], 2), (sets[1], 2), (sets[2], dct["mincov"])]
        res = []
        for s, mc in todo:
            nm = sorted(s)
            if len(nm) < mc:
                res = None
                break
            rows, ref_row = gip.star_aligner(nm, [s[n] for n in nm], ref)
            x, cns = oracle.indel_tensor(np.array([[sym[c] for c in r] for r in rows], np.uint8), np.array([sym[c] for c in ref_row], np.uint8))
            res.append((x, "".join("AGTC"[c] for c in cns if c != 4)))
        if res is None:
            continue
        pos.append(a)
        xs.append([r[0] for r in res])
        mr = {0: max(10, dct["win_size"]), 1: 10}[variants[a]]
        al = [gip.allele_prediction(r[1], ref, mr) for r in res]
        alleles.append(al[0] if hapl else al)
        if not hapl:
            first = next(iter(sets[0]))
            k = names.index(first)
            phase.append(int(ps[k]) if hap[k] else None)
    return pos, xs, alleles, (None if hapl else phase)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%d_%s_%s" % (c["k"], c["world"], c["ploidy"]))
def test_oracle_and_host_pieces_reproduce_the_reference_tuple(files, case):
    w, bam, fa = files[case["world"]]
    pos, xs, alleles, phase = _assemble(case, w, bam, fa)
    assert pos == case["pos"]
    assert alleles == _norm_alleles(case["alleles"])
    assert phase == case["phase"]
    for i, gx in enumerate(case["xs"]):
        got = np.stack([x[i] for x in xs])
        assert np.array_equal(got.astype(np.float32), gx)


@pytest.mark.gpu
@pytest.mark.parametrize("aligner", ["host_star", "device"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%d_%s_%s" % (c["k"], c["world"], c["ploidy"]))
def test_get_indel_testing_candidates_equals_the_reference_tuple(files, case, aligner):
    """the product functions (pass 1 K7 on the GPU, native BAM windows, star alignment on the host cores or on the device,
    K8, nc_allele_prediction) return exactly what the reference's functions returned"""
    w, bam, fa = files[case["world"]]
    dct = dict(case["dct"], fasta_path=fa)
    chunk = dict(chrom=w.chrom, start=case["start"], end=case["end"], sam_path=bam)
    al = gip.star_aligner if aligner == "host_star" else "device"
    if case["ploidy"] == "haploid":
        pos, x, alleles = gip.get_indel_testing_candidates_haploid(dct, chunk, aligner=al)
        xs, phase = [x], None
    else:
        pos, x0, x1, x2, alleles, phase = gip.get_indel_testing_candidates(dct, chunk, aligner=al)
        xs = [x0, x1, x2]
    assert list(pos) == case["pos"]
    assert [tuple(a) if case["ploidy"] == "haploid" else [tuple(t) for t in a] for a in alleles] == _norm_alleles(case["alleles"])
    assert (None if phase is None else list(phase)) == case["phase"]
    if case["n"] == 0:
        return
    for got, gx in zip(xs, case["xs"]):
        assert np.asarray(got).dtype == np.float64 and np.asarray(got).shape == (case["n"], 5, 128, 2)
        assert np.array_equal(np.asarray(got).astype(np.float32), gx)


# ------------------------------------------------------------------------- the reference's indel_run() end to end (e2e_vcf.npz)
ZE = np.load(os.path.join(GOLD, "e2e_vcf.npz"))


def _compare_indel_vcf(got, exp):
    """records identical in CHROM POS REF ALT FILTER FORMAT, GT and PS; QUAL / GQ (%.2f of -10 log10 of float32 probabilities)
    within what a 1e-4 change of the probability allows"""
    assert len(got) == len(exp), (len(got), len(exp))
    for g, e in zip(got, exp):
        gf, ef = g.rstrip("\n").split("\t"), e.rstrip("\n").split("\t")
        assert gf[:5] == ef[:5] and gf[6:9] == ef[6:9], (g, e)
        gs, es = gf[9].split(":"), ef[9].split(":")
        assert gs[0] == es[0] and gs[2:] == es[2:], (g, e)
        for a, b in ((float(gf[5]), float(ef[5])), (float(gs[1]), float(es[1]))):
            pa, pb = 10 ** (-abs(a) / 10), 10 ** (-abs(b) / 10)
            assert abs(pa - pb) <= 2e-4 or abs(a - b) <= 0.02 * max(1.0, abs(b)), (g, e)


@pytest.mark.parametrize("tag", ["indel_a", "indel_b"])
def test_oracle_pipeline_writes_the_reference_indel_runs_vcf(files, tag):
    """pass 2 from the oracle / host pieces + oracle indel CNN + the host rules == the lines of the reference's indel_run()"""
    from nanocaller_amd import indelCaller
    from nanocaller_amd.weights import Weights, get_indel_model
    params = json.loads(str(ZE[tag + "_params"]))
    w, bam, fa = files[str(ZE[tag + "_world"])]
    wd = Weights(get_indel_model(params["indel_model"]))
    wh = Weights(get_indel_model("haploid"))
    lines = []
    for ploidy, a, b in json.loads(str(ZE[tag + "_chunks"])):
        pos, xs, alleles, phase = _assemble(dict(dct=params, ploidy=ploidy, start=a, end=b), w, bam, fa)
        if not pos:
            continue
        if ploidy == "diploid":
            x = np.hstack([np.stack([x[i] for x in xs]) for i in range(3)]).astype(np.float32)
            lines += indelCaller.indel_vcf_lines(w.chrom, pos, oracle.indel_forward(wd.flat, x, precision="f32"), alleles, phase)[0]
        else:
            x = np.stack([x[0] for x in xs]).astype(np.float32)
            lines += indelCaller.indel_vcf_lines_haploid(w.chrom, pos, oracle.indel_forward(wh.flat, x, precision="f32"), alleles)[0]
    exp = str(ZE[tag + "_vcf"]).splitlines()
    assert len(exp) > 20
    _compare_indel_vcf(lines, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["indel_a", "indel_b"])
def test_hip_indel_run_writes_the_reference_indel_runs_vcf(files, tag, tmp_path):
    """indelCaller.indel_run (the product worker loop, everything on the GPU / native host code) on the same BAM and chunks"""
    import queue

    from nanocaller_amd import indelCaller
    params = json.loads(str(ZE[tag + "_params"]))
    w, bam, fa = files[str(ZE[tag + "_world"])]
    params.update(fasta_path=fa, intermediate_indel_files_dir=str(tmp_path), prefix="t")
    jobs, files_out = queue.Queue(), []
    for ploidy, a, b in json.loads(str(ZE[tag + "_chunks"])):
        jobs.put(("indel", dict(chrom=w.chrom, start=a, end=b, ploidy=ploidy, sam_path=bam)))
    path = indelCaller.indel_run(params, {}, jobs, queue.Queue(), files_out, aligner="device")
    _compare_indel_vcf(open(path).read().splitlines(), str(ZE[tag + "_vcf"]).splitlines())


@pytest.mark.gpu
@pytest.mark.parametrize("haploid", [False, True])
def test_batched_featuriser_equals_per_chunk_calls(files, haploid):
    """get_indel_testing_candidates_batch (all chunks of a contig through one pass-1 launch set, one native pass-2 call, one
    alignment call -- what indel_run uses) returns, chunk by chunk, the tuples of the per-chunk functions; adjacent chunks
    share anchors in their overlap zone"""
    from nanocaller_amd.generate_indel_pileups_haploid import get_indel_testing_candidates_haploid
    for wn, kw in (("a", {}), ("b", dict(impute_indel_phase=True, del_t=0.4))):
        w, bam, fa = files[wn]
        dct = dict(seq="ont", win_size=40, small_win_size=4, mincov=2, maxcov=160, ins_t=0.4, del_t=0.6, supplementary=False,
                   exclude_bed=None, impute_indel_phase=False, fasta_path=fa)
        dct.update(kw)
        chunks = [dict(chrom=w.chrom, start=s, end=min(w.length, s + 4_000), sam_path=bam) for s in range(1, w.length, 4_000)]
        got = gip.get_indel_testing_candidates_batch(dct, chunks, haploid=haploid)
        assert len(got) == len(chunks) and sum(len(t[0]) for t in got) > 20
        for c, t in zip(chunks, got):
            e = get_indel_testing_candidates_haploid(dct, c, aligner="device") if haploid else gip.get_indel_testing_candidates(dct, c, aligner="device")
            assert list(t[0]) == list(e[0])
            if len(e[0]):
                for a, b in zip(t[1:], e[1:]):
                    if isinstance(b, np.ndarray):
                        assert np.array_equal(np.asarray(a), b)
                    else:
                        assert list(a) == list(b)
        # device_x=True (what indel_run feeds the CNN): the same tuples with the tensors left in HBM as float32
        dev = gip.get_indel_testing_candidates_batch(dct, chunks, haploid=haploid, device_x=True)
        for t, d in zip(got, dev):
            assert list(t[0]) == list(d[0])
            for a, b in zip(t[1:], d[1:]):
                if isinstance(a, np.ndarray):
                    assert b.is_cuda and np.array_equal(a.astype(np.float32), b.cpu().numpy())
                else:
                    assert list(a) == list(b)


def test_pass2_sets_mark_repeated_alignments(files):
    """nc_pass2_arrays.al_dup (host code, no GPU): every read of an anchor's third ("all reads") set that is in one of its
    haplotype sets points at that earlier alignment -- same read window, same reference window -- and nothing else does"""
    import ctypes as C
    from nanocaller_amd import _lib
    L = _lib.lib()
    n_dup = n_all = 0
    for wn in ("a", "b"):
        w, bam, fa = files[wn]
        ctg = gip.decoded_contig(bam, w.chrom, fa)
        dec = ctg["dec"]
        keep = np.ascontiguousarray((dec["read_flag"] & 0xF04) == 0, np.uint8)
        anc = np.arange(50, w.length - 200, 17, dtype=np.int32)       # > 128 anchors: worker threads, partial results merged
        h = C.c_void_p()
        assert L.nc_indel_pass2_sets(ctg["handle"], _lib.npp(keep), len(anc), _lib.npp(anc), ctg["fasta_b"], len(ctg["fasta"]), 1,
                                     len(ctg["fasta"]), 160, 2, 160, 0, None, None, None, C.byref(h)) == _lib.NC_OK
        v = _lib.Pass2ArraysC()
        L.nc_pass2_view(h, C.byref(v))
        as_i32 = lambda p, n: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), (n,)).copy()   # noqa: E731
        A, ns = v.n_alignments, v.n_sets
        assert v.sets_per_anchor == 3 and ns == 3 * v.n_kept and v.n_kept > 20
        dup, roff, s0 = as_i32(v.al_dup, A), as_i32(v.read_off, A + 1), as_i32(v.set_read0, ns + 1)
        reads = C.string_at(v.reads, int(roff[A]))
        for k in range(v.n_kept):
            a0, a1, a2, a3 = s0[3 * k:3 * k + 4]
            assert (dup[a0:a2] == -1).all()
            for a in range(a2, a3):
                if dup[a] >= 0:
                    assert a0 <= dup[a] < a2 and reads[roff[dup[a]]:roff[dup[a] + 1]] == reads[roff[a]:roff[a + 1]]
                    n_dup += 1
            n_all += a3 - a2
            # every phased read is in the third set (below maxcov): all of them are marked
            assert (dup[a2:a3] >= 0).sum() == a2 - a0
        L.nc_pass2_free(h)
    assert n_dup > 500 and n_dup < n_all


@pytest.mark.gpu
def test_skipping_repeated_alignments_changes_nothing(files, monkeypatch):
    """nc_star_msa_tensor_dup with the duplicate map of nc_indel_pass2_sets (the product path) returns what the plain call
    that aligns every read of every set returns, with and without imputed read sets"""
    for wn, kw in (("a", {}), ("b", dict(impute_indel_phase=True, del_t=0.4))):
        w, bam, fa = files[wn]
        dct = dict(seq="ont", win_size=40, small_win_size=4, mincov=2, maxcov=160, ins_t=0.4, del_t=0.6, supplementary=False,
                   exclude_bed=None, impute_indel_phase=False, fasta_path=fa)
        dct.update(kw)
        chunks = [dict(chrom=w.chrom, start=s, end=min(w.length, s + 4_000), sam_path=bam) for s in range(1, w.length, 4_000)]
        monkeypatch.delenv("NC_MSA_NO_DEDUP", raising=False)
        got = gip.get_indel_testing_candidates_batch(dct, chunks, haploid=False)
        monkeypatch.setenv("NC_MSA_NO_DEDUP", "1")
        exp = gip.get_indel_testing_candidates_batch(dct, chunks, haploid=False)
        assert sum(len(t[0]) for t in got) > 20
        for t, e in zip(got, exp):
            assert list(t[0]) == list(e[0])
            for a, b in zip(t[1:], e[1:]):
                if isinstance(b, np.ndarray):
                    assert np.array_equal(np.asarray(a), b)
                else:
                    assert list(a) == list(b)


@pytest.mark.gpu
def test_long_chunk_lists_go_through_in_groups(files, monkeypatch):
    """a chromosome's worth of chunks exceeds the native pass-2 arrays: the batched featuriser splits the list (and halves a
    group that still overflows); the per-chunk tuples are the same"""
    from nanocaller_amd import _lib
    w, bam, fa = files["a"]
    dct = dict(seq="ont", win_size=40, small_win_size=4, mincov=2, maxcov=160, ins_t=0.4, del_t=0.6, supplementary=False,
               exclude_bed=None, impute_indel_phase=False, fasta_path=fa)
    chunks = [dict(chrom=w.chrom, start=s, end=min(w.length, s + 3_000), sam_path=bam) for s in range(1, w.length, 3_000)]
    assert len(chunks) >= 6
    exp = gip.get_indel_testing_candidates_batch(dct, chunks)

    def same(got):
        assert len(got) == len(exp)
        for t, e in zip(got, exp):
            assert list(t[0]) == list(e[0])
            for a, b in zip(t[1:], e[1:]):
                if isinstance(b, np.ndarray):
                    assert np.array_equal(np.asarray(a), b)
                else:
                    assert list(a) == list(b)
    monkeypatch.setattr(gip, "MAX_BATCH_CHUNKS", 2)
    same(gip.get_indel_testing_candidates_batch(dct, chunks))
    # the native assembler reports NC_ERR_CAPACITY for more than two chunks at once: the list is halved until it fits
    monkeypatch.setattr(gip, "MAX_BATCH_CHUNKS", 64)
    inner, calls = gip._indel_batch, []

    def limited(dct_, chunks_, *a):
        calls.append(len(chunks_))
        if len(chunks_) > 2:
            err = _lib.NanoCallerHipError("nc_indel_pass2_sets failed (-2)")
            err.status = _lib.NC_ERR_CAPACITY
            raise err
        return inner(dct_, chunks_, *a)
    monkeypatch.setattr(gip, "_indel_batch", limited)
    same(gip.get_indel_testing_candidates_batch(dct, chunks))
    assert calls[0] == len(chunks) and max(calls[1:]) < len(chunks) and sum(c for c in calls if c <= 2) == len(chunks)
