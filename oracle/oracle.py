"""ctypes binding of oracle/libnc_oracle.so -- the CPU restatement of the reference path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (nanocaller_amd/) must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODES = {"ont": 0, "short_ont": 1, "ul_ont": 2, "ul_ont_extreme": 3, "pacbio": 4}


def build(force=False):
    so = os.path.join(_HERE, "libnc_oracle.so")
    src = os.path.join(_HERE, "nc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libnc_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class RawReads:
    """Minimal stand-in for synth.World built from raw arrays (used by bench.py's cpu_baseline leg)."""

    def __init__(self, chrom, length, read_start, read_end, read_off, codes, strand, keep):
        self.chrom, self.length = chrom, int(length)
        self.read_start, self.read_end, self.read_off, self.codes = read_start, read_end, read_off, codes
        self.read_flag = (np.asarray(strand, np.int32) * 16) | np.where(np.asarray(keep) != 0, 0, 0x100).astype(np.int32)
        self.ref = None


def _reads(world, supplementary=False):
    from nanocaller_amd.synth import FLAG_FILTER_DEFAULT, FLAG_FILTER_SUPPL
    filt = FLAG_FILTER_SUPPL if supplementary else FLAG_FILTER_DEFAULT
    keep = ((world.read_flag & filt) == 0).astype(np.uint8)
    strand = ((world.read_flag & 0x10) != 0).astype(np.uint8)       # :143 for a primary record; see pack.pack_reads
    return (np.ascontiguousarray(world.read_start, np.int32), np.ascontiguousarray(world.read_end, np.int32),
            np.ascontiguousarray(world.read_off, np.int64), np.ascontiguousarray(world.codes, np.uint8),
            strand, keep)


def _mates(world, keep, strand):
    """Alignments that share a read name (the reference's pileup dicts are keyed by name, generate_SNP_pileups.py:141-143,175,185): -> (mate_next
    int32 [n] or None, strand): mate_next[r] = the next kept alignment of r's name, circular, -1 when the name is r's alone; the strand of every
    alignment of a name is the 0x10 bit of its LAST primary record in file order (strand_dict[qname] is assigned per primary record, :141-143;
    the reference raises KeyError when none is in the chunk's fetch window -- here the alignments of such a name keep their own bits)."""
    names = getattr(world, "names", None)
    if names is None or len(names) != len(world.read_start):
        return None, strand
    groups = {}
    for r in np.nonzero(keep)[0].tolist():
        groups.setdefault(names[r], []).append(r)
    nxt = np.full(len(keep), -1, np.int32)
    strand = strand.copy()
    any_ = False
    for g in groups.values():
        if len(g) < 2:
            continue
        any_ = True
        for a, b in zip(g, g[1:] + g[:1]):
            nxt[a] = b
        prim = [r for r in g if (int(world.read_flag[r]) & 0x900) == 0]
        if prim:
            strand[g] = strand[prim[-1]]
    return (nxt if any_ else None), strand


def ref_codes_with_exclusions(world, exclude=None):
    from nanocaller_amd.synth import world_ref_codes
    rc = world_ref_codes(world).copy()
    if exclude:
        for (c, a, b) in exclude:                 # IntervalTree.overlaps(pos): a <= pos < b
            if c == world.chrom:
                rc[max(0, a - 1):max(0, b - 1)] = 4
    return rc


def get_cnd_pos(v_pos, cnd_pos, seq="ont"):
    nbr = np.ascontiguousarray(cnd_pos, np.int32)
    left = np.zeros(64, np.int32)
    right = np.zeros(64, np.int32)
    nl, nr = C.c_int32(), C.c_int32()
    rc = lib().oracle_get_cnd_pos(C.c_int32(int(v_pos)), _p(nbr, C.c_int32), C.c_int32(nbr.size), MODES[seq],
                                  _p(left, C.c_int32), C.byref(nl), _p(right, C.c_int32), C.byref(nr))
    assert rc == 0
    return left[:nl.value].tolist(), right[:nr.value].tolist()


def snp_scan(world, rc, start, end, ploidy, mincov, min_allele_freq, threshold, supplementary=False):
    rs, re_, ro, codes, strand, keep = _reads(world, supplementary)
    cap = min(world.length + 1, end - start + 2 * 50000 + 2)
    nbr = np.zeros(cap, np.int32)
    cpos = np.zeros(cap, np.int32)
    cn = np.zeros(cap, np.int32)
    calt = np.zeros(cap, np.int32)
    n_nbr, n_cand = C.c_int32(), C.c_int32()
    r = lib().oracle_snp_scan(C.c_int32(rs.size), _p(rs, C.c_int32), _p(re_, C.c_int32), _p(ro, C.c_int64),
                              _p(codes, C.c_uint8), _p(keep, C.c_uint8), _p(rc, C.c_uint8),
                              C.c_int32(world.length), C.c_int32(start), C.c_int32(end),
                              C.c_int(1 if ploidy == "haploid" else 0), C.c_int32(mincov),
                              C.c_double(min_allele_freq), C.c_double(threshold[0]), C.c_double(threshold[1]),
                              C.c_int32(cap), _p(nbr, C.c_int32), C.byref(n_nbr), _p(cpos, C.c_int32),
                              _p(cn, C.c_int32), _p(calt, C.c_int32), C.byref(n_cand))
    assert r == 0, r
    k, m = n_nbr.value, n_cand.value
    return nbr[:k].copy(), cpos[:m].copy(), cn[:m].copy(), calt[:m].copy()


def snp_featurize(world, rc, nbr, cpos, seq, maxcov, min_nbr_sites, supplementary=False):
    rs, re_, ro, codes, strand, keep = _reads(world, supplementary)
    mate_next, strand = _mates(world, keep, strand)
    n = int(cpos.size)
    out_pos = np.zeros(max(n, 1), np.int32)
    out_ref = np.zeros(max(n, 1), np.int32)
    mat = np.zeros((max(n, 1), 5, 41, 5), np.float32)
    fwd = np.zeros((max(n, 1), 4), np.int32)
    rev = np.zeros((max(n, 1), 4), np.int32)
    dep = np.zeros(max(n, 1), np.int32)
    nbr = np.ascontiguousarray(nbr, np.int32)
    cpos = np.ascontiguousarray(cpos, np.int32)
    k = lib().oracle_snp_featurize(C.c_int32(rs.size), _p(rs, C.c_int32), _p(re_, C.c_int32), _p(ro, C.c_int64),
                                   _p(codes, C.c_uint8), _p(strand, C.c_uint8), _p(keep, C.c_uint8),
                                   _p(rc, C.c_uint8), _p(nbr, C.c_int32), C.c_int32(nbr.size),
                                   _p(cpos, C.c_int32), C.c_int32(n), C.c_int(MODES[seq]), C.c_int32(maxcov),
                                   C.c_int32(min_nbr_sites), _p(out_pos, C.c_int32), _p(out_ref, C.c_int32),
                                   _p(mat, C.c_float), _p(fwd, C.c_int32), _p(rev, C.c_int32), _p(dep, C.c_int32),
                                   _p(mate_next, C.c_int32) if mate_next is not None else None)
    assert k >= 0, k
    return out_pos[:k], out_ref[:k], mat[:k], fwd[:k], rev[:k], dep[:k]


def get_snp_testing_candidates(world, dct, region, exclude=None, rc=None):
    """Same 8-tuple as the reference function (generate_SNP_pileups.py:279), computed by the oracle.
    (pos, ref_onehot int32 (N,4), mat f32 (N,5,41,5), dp, freq f64, depth f64, fwd_dp f64 (N,4), rev_dp)"""
    if rc is None:
        rc = ref_codes_with_exclusions(world, exclude)
    nbr, cpos, cn, calt = snp_scan(world, rc, region["start"], region["end"], region["ploidy"],
                                   dct["mincov"], dct["min_allele_freq"], dct["threshold"],
                                   dct.get("supplementary", False))
    if cpos.size == 0:
        return ([], [], [], [], [], 0, [], [])
    pos, ref, mat, fwd, rev, dep = snp_featurize(world, rc, nbr, cpos, dct["seq"], dct["maxcov"],
                                                 dct["min_nbr_sites"], dct.get("supplementary", False))
    if pos.size == 0:
        return ([], [], [], [], [], 0, [], [])
    sel = np.searchsorted(cpos, pos)
    onehot = np.eye(4, dtype=np.int32)[ref]
    return (pos.astype(np.int64), onehot, mat, cn[sel].astype(np.int64),
            calt[sel].astype(np.float64) / cn[sel].astype(np.float64), float(np.mean(dep.astype(np.float64))),
            fwd.astype(np.float64), rev.astype(np.float64))


def _fwd(fn, w, x, ref_code, scale, scale_mode, n_out, with_gt):
    x = np.ascontiguousarray(x, np.float32)
    n = x.shape[0]
    w = np.ascontiguousarray(w, np.float32)
    ref_code = np.ascontiguousarray(ref_code, np.int32)
    scale = np.ascontiguousarray(np.broadcast_to(np.asarray(scale, np.float64), (n,)))
    probs = np.zeros((n, n_out), np.float32)
    if with_gt:
        gt = np.zeros((n, 2), np.float32)
        r = fn(_p(w, C.c_float), C.c_int64(n), _p(x, C.c_float), _p(ref_code, C.c_int32), _p(scale, C.c_double),
               C.c_int(scale_mode), _p(probs, C.c_float), _p(gt, C.c_float))
        assert r == 0
        return probs, gt
    r = fn(_p(w, C.c_float), C.c_int64(n), _p(x, C.c_float), _p(ref_code, C.c_int32), _p(scale, C.c_double),
           C.c_int(scale_mode), _p(probs, C.c_float))
    assert r == 0
    return probs


def snp_forward(w_flat, x, ref_code, scale, scale_mode=0, precision="f32"):
    """-> (probs (N,4) class-1 prob of the A,G,T,C heads, gt (N,2)).  model_architect.py:36-64."""
    fn = lib().oracle_snp_forward_f if precision == "f32" else lib().oracle_snp_forward_d
    return _fwd(fn, w_flat, x, ref_code, scale, scale_mode, 4, True)


def snp_hap_forward(w_flat, x, ref_code, scale, scale_mode=0, precision="f32"):
    fn = lib().oracle_snp_hap_forward_f if precision == "f32" else lib().oracle_snp_hap_forward_d
    return _fwd(fn, w_flat, x, ref_code, scale, scale_mode, 4, False)


def indel_forward(w_flat, x, precision="f32"):
    x = np.ascontiguousarray(x, np.float32)
    n, rows = x.shape[0], x.shape[1]
    w = np.ascontiguousarray(w_flat, np.float32)
    nout = 4 if rows == 15 else 1
    probs = np.zeros((n, nout), np.float32)
    fn = lib().oracle_indel_forward_f if precision == "f32" else lib().oracle_indel_forward_d
    r = fn(_p(w, C.c_float), C.c_int64(n), C.c_int(rows), _p(x, C.c_float), _p(probs, C.c_float))
    assert r == 0
    return probs


def indel_tensor(rows, ref_row):
    rows = np.ascontiguousarray(rows, np.uint8)
    ref_row = np.ascontiguousarray(ref_row, np.uint8)
    out = np.zeros((5, 128, 2), np.float32)
    cns = np.zeros(rows.shape[1] + 1, np.uint8)
    nc = C.c_int32()
    r = lib().oracle_indel_tensor(_p(rows, C.c_uint8), C.c_int32(rows.shape[0]), C.c_int32(rows.shape[1]),
                                  _p(ref_row, C.c_uint8), _p(out, C.c_float), _p(cns, C.c_uint8), C.byref(nc))
    assert r == 0
    return out, cns[:nc.value].copy()


def indel_scan(world, start, end, *, mincov, win_size, small_win_size, ins_t, del_t, exclude=None, supplementary=False, haploid=False):
    """Pass 1 of get_indel_testing_candidates (generate_indel_pileups.py:197-304) -> (anchor positions, types);
    haploid=True: get_indel_testing_candidates_haploid (generate_indel_pileups_haploid.py:185-241)."""
    rs, re_, ro, codes, strand, keep = _reads(world, supplementary)
    ev_off, ev_pos, ev_len = (np.ascontiguousarray(a, np.int32) for a in world.meta["events"])
    hap = np.ascontiguousarray(world.meta["hap"], np.uint8)
    excl = None
    if exclude:
        excl = np.zeros(world.length, np.uint8)
        for (a, b) in exclude:
            excl[max(0, a - 1):max(0, b - 1)] = 1
    cap = end - start + 2
    vp = np.zeros(cap, np.int32)
    vt = np.zeros(cap, np.int32)
    nv = C.c_int32()
    r = lib().oracle_indel_scan(C.c_int32(rs.size), _p(rs, C.c_int32), _p(re_, C.c_int32), _p(keep, C.c_uint8),
                                _p(hap, C.c_uint8), _p(ev_off, C.c_int32), _p(ev_pos, C.c_int32), _p(ev_len, C.c_int32),
                                _p(excl, C.c_uint8) if excl is not None else None, C.c_int32(world.length),
                                C.c_int32(start), C.c_int32(end), C.c_int32(mincov), C.c_int32(win_size),
                                C.c_int32(small_win_size), C.c_double(ins_t), C.c_double(del_t), C.c_int32(1 if haploid else 0), C.c_int32(cap),
                                _p(vp, C.c_int32), _p(vt, C.c_int32), C.byref(nv))
    assert r == 0, r
    return vp[:nv.value].copy(), vt[:nv.value].copy()


def indel_scan_impute(world, start, end, *, mincov, win_size, small_win_size, ins_t, del_t, exclude=None, supplementary=False):
    """Pass 1 of get_indel_testing_candidates WITH dct['impute_indel_phase'] (generate_indel_pileups.py:197-304), restated
    column by column with read-index sets and deques (pure Python: small chunks only) -> (variants {anchor: type},
    extra_variants {anchor: (read indices 0, read indices 1)}).  Pileup strings: base letter of the read's code (code 4 =
    '*'), then '+<n><inserted bases>' / '-<n>N..' of an event on this column (world.meta['ev_ins'] holds the bases)."""
    import collections
    rs, re_, ro, codes, strand, keep = _reads(world, supplementary)
    ev_off, ev_pos, ev_len = world.meta["events"]
    ins_off, ins_bases = world.meta["ev_ins"]
    hap = world.meta["hap"]
    raw = np.asarray(ins_bases, np.uint8).tobytes().decode("ascii")
    excl = np.zeros(world.length + 2, bool)
    for (a, b) in exclude or []:
        excl[max(0, a):max(0, b)] = True                              # IntervalTree.overlaps(pos): a <= pos < b
    lo, hi = max(1, start), min(end, world.length)
    order = np.nonzero(keep)[0]
    ev_at = {}
    for r in order:
        for k in range(int(ev_off[r]), int(ev_off[r + 1])):
            ev_at[(int(r), int(ev_pos[k]))] = k
    classes = [("del", False), ("ins", False), ("del", True), ("ins", True)]
    dq = {(c, h): collections.deque([set()] * (small_win_size if c >= 2 else win_size), small_win_size if c >= 2 else win_size)
          for c in range(4) for h in (0, 1)}
    variants, extra = {}, {}
    prev = 0
    for v in range(lo, hi + 1):
        reads = [int(r) for r in order[(rs[order] <= v) & (re_[order] > v)]]
        if not reads or excl[v]:
            continue
        strs = []
        for r in reads:
            s = "AGTC*"[int(codes[ro[r] + (v - rs[r])])]
            k = ev_at.get((r, v))
            if k is not None:
                ln = int(ev_len[k])
                s += ("+%d%s" % (ln, raw[ins_off[k]:ins_off[k + 1]].upper())) if ln > 0 else ("-%d%s" % (-ln, "N" * -ln))
            strs.append(s)
        r0 = [r for r in reads if hap[r] == 1]
        r1 = [r for r in reads if hap[r] == 2]
        for c, (kind, small) in enumerate(classes):
            for h, rh in ((0, r0), (1, r1)):
                members = set()
                for r in rh:
                    k = ev_at.get((r, v))
                    if k is None:
                        continue
                    ln = int(ev_len[k])
                    if (ln > 0) != (kind == "ins"):
                        continue
                    n = abs(ln)
                    if (n <= 10) if small else (2 < n <= 50):
                        members.add(r)
                dq[(c, h)].append(members)
        if v <= prev:
            continue
        n0, n1, tot = len(r0), len(r1), len(reads)
        if n0 >= mincov and n1 >= mincov:
            f = {(c, h): (len(set.union(*dq[(c, h)])) / n if n > 0 else 0) for c in range(4) for h, n in ((0, n0), (1, n1))}
            if max(f[(0, 0)], f[(0, 1)]) >= del_t or max(f[(1, 0)], f[(1, 1)]) >= ins_t:
                prev = v + win_size
                variants[max(1, v - win_size)] = 0
            elif (max(f[(2, 0)], f[(2, 1)]) >= del_t or max(f[(3, 0)], f[(3, 1)]) >= ins_t or f[(2, 0)] + f[(3, 0)] >= 0.9 or
                  f[(2, 1)] + f[(3, 1)] >= 0.9):
                prev = v + 10
                variants[max(1, v - 10)] = 1
        elif tot >= 2 * mincov:
            two = "".join(s[:2] for s in strs)
            del_f = (two.count("-") + two.count("*")) / tot
            ins_f = two.count("+") / tot
            if del_t <= del_f or ins_t <= ins_f:
                groups = {}
                for s, r in zip(strs, reads):
                    groups.setdefault(s, []).append(r)
                counts = sorted(((g, len(m)) for g, m in groups.items()), key=lambda x: x[1], reverse=True)
                if counts[0][1] <= 0.8 * tot:
                    a = set(groups[counts[0][0]])
                    b = set(groups[counts[1][0]]) if counts[1][1] >= mincov else set(reads) - a
                else:
                    m = groups[counts[0][0]]
                    a, b = m[:counts[0][1] // 2], m[counts[0][1] // 2:]
                if len(a) >= mincov and len(b) >= mincov:
                    prev = v + 10
                    variants[max(1, v - 10)] = 1
                    extra[max(1, v - 10)] = (sorted(a), sorted(b))
    return variants, extra


# ------------------------------------------------------------------------------------------------- indel pass 2 (a11, a13)
# Pure-Python restatements used only by tests.  parasail / pysam are absent from this image: parity with them is UNPINNED;
# these pin the library's native code (nc_nw_cigar, nc_allele_prediction, nc_indel_slices) against an independent
# implementation of the same documented rules.
def nw_cigar_ref(s1, s2, open_=9, extend=1, match=20, mismatch=-10):
    """Gotoh global alignment, gap of length k costs open + (k-1)*extend; ties: diagonal, then D (s2 only), then I (s1 only);
    inside a gap extension is preferred over opening on equal scores.  -> [(op, count)], ops '=' 7, 'X' 8, 'I' 1, 'D' 2
    (generate_indel_pileups.py:79: parasail.nw_trace(alt, ref_seq, 9, 1, sub_mat).cigar)."""
    n1, n2 = len(s1), len(s2)
    NEG = -10 ** 9
    H = [[0] * (n2 + 1) for _ in range(n1 + 1)]
    E = [[NEG] * (n2 + 1) for _ in range(n1 + 1)]
    F = [[NEG] * (n2 + 1) for _ in range(n1 + 1)]
    who = [[0] * (n2 + 1) for _ in range(n1 + 1)]             # 0 diag, 1 D, 2 I
    e_ext = [[False] * (n2 + 1) for _ in range(n1 + 1)]
    f_ext = [[False] * (n2 + 1) for _ in range(n1 + 1)]
    for j in range(1, n2 + 1):
        H[0][j] = E[0][j] = -open_ - (j - 1) * extend
        who[0][j] = 1
        e_ext[0][j] = j > 1
    for i in range(1, n1 + 1):
        H[i][0] = F[i][0] = -open_ - (i - 1) * extend
        who[i][0] = 2
        f_ext[i][0] = i > 1
        for j in range(1, n2 + 1):
            eo, ee = H[i][j - 1] - open_, E[i][j - 1] - extend
            e_ext[i][j] = ee >= eo
            E[i][j] = max(eo, ee)
            fo, fe = H[i - 1][j] - open_, F[i - 1][j] - extend
            f_ext[i][j] = fe >= fo
            F[i][j] = max(fo, fe)
            d = H[i - 1][j - 1] + (match if s1[i - 1] == s2[j - 1] else mismatch)
            h, w = d, 0
            if E[i][j] > h:
                h, w = E[i][j], 1
            if F[i][j] > h:
                h, w = F[i][j], 2
            H[i][j], who[i][j] = h, w
    ops = []
    i, j, state = n1, n2, None
    while i > 0 or j > 0:
        if state is None:
            if who[i][j] == 0:
                ops.append(7 if s1[i - 1] == s2[j - 1] else 8)
                i, j = i - 1, j - 1
                continue
            state = who[i][j]
        if state == 1:
            ops.append(2)
            ext = e_ext[i][j]
            j -= 1
        else:
            ops.append(1)
            ext = f_ext[i][j]
            i -= 1
        if not ext:
            state = None
    ops.reverse()
    out = []
    for o in ops:
        if out and out[-1][0] == o:
            out[-1][1] += 1
        else:
            out.append([o, 1])
    return [(o, c) for o, c in out]


def nw_cigar_free_tail_ref(s1, s2, open_=9, extend=1, match=20, mismatch=-10):
    """nw_cigar_ref anchored at the start only: the alignment ends at the best cell of the last row or last column (ties: the
    corner, then the last row from the corner outwards, then the last column), the unaligned tail is one trailing gap."""
    n1, n2 = len(s1), len(s2)
    if n1 == 0 or n2 == 0:
        return nw_cigar_ref(s1, s2, open_, extend, match, mismatch)
    NEG = -10 ** 9
    H = [[0] * (n2 + 1) for _ in range(n1 + 1)]
    E = [[NEG] * (n2 + 1) for _ in range(n1 + 1)]
    F = [[NEG] * (n2 + 1) for _ in range(n1 + 1)]
    for j in range(1, n2 + 1):
        H[0][j] = E[0][j] = -open_ - (j - 1) * extend
    for i in range(1, n1 + 1):
        H[i][0] = F[i][0] = -open_ - (i - 1) * extend
        for j in range(1, n2 + 1):
            E[i][j] = max(H[i][j - 1] - open_, E[i][j - 1] - extend)
            F[i][j] = max(H[i - 1][j] - open_, F[i - 1][j] - extend)
            H[i][j] = max(H[i - 1][j - 1] + (match if s1[i - 1] == s2[j - 1] else mismatch), E[i][j], F[i][j])
    best, bi, bj = H[n1][n2], n1, n2
    for j in range(n2 - 1, -1, -1):
        if H[n1][j] > best:
            best, bi, bj = H[n1][j], n1, j
    for i in range(n1 - 1, -1, -1):
        if H[i][n2] > best:
            best, bi, bj = H[i][n2], i, n2
    head = nw_cigar_ref(s1[:bi], s2[:bj], open_, extend, match, mismatch)
    tail = ([(2, n2 - bj)] if bj < n2 else []) + ([(1, n1 - bi)] if bi < n1 else [])
    out = [list(t) for t in head]
    for o, c in tail:                                           # the library emits the tail gap before merging runs
        if out and out[-1][0] == o:
            out[-1][1] += c
        else:
            out.append([o, c])
    return [(o, c) for o, c in out]


def star_msa_ref(seqs, ref, open_=9, extend=1, match=20, mismatch=-10, cigars=None):
    """Star alignment restated in pure Python (checks nc_star_msa): pairwise free-tail alignments to `ref`, merged in reference
    coordinates -- per reference slot as many insertion columns as the longest insertion, shorter ones left-justified.
    -> (rows, ref_row) as strings over AGTC-"""
    n_ref = len(ref)
    cig = cigars if cigars is not None else [nw_cigar_free_tail_ref(q, ref, open_, extend, match, mismatch) for q in seqs]
    ins = [0] * (n_ref + 1)
    for c in cig:
        j = 0
        for op, cnt in c:
            if op == 1:
                ins[j] = max(ins[j], cnt)
            else:
                j += cnt
    col, acc = [], 0
    for j in range(n_ref + 1):
        acc += ins[j]
        col.append(acc + j)
    ncol = col[n_ref]
    ref_row = ["-"] * ncol
    for j in range(n_ref):
        ref_row[col[j]] = ref[j]
    rows = []
    for q, c in zip(seqs, cig):
        row = ["-"] * ncol
        i = j = 0
        for op, cnt in c:
            if op in (7, 8):
                for _ in range(cnt):
                    row[col[j]] = q[i]
                    i += 1
                    j += 1
            elif op == 2:
                j += cnt
            else:
                s0 = col[j] - ins[j]
                for t in range(cnt):
                    row[s0 + t] = q[i]
                    i += 1
        assert i == len(q)
        rows.append("".join(row))
    return rows, "".join(ref_row)


def read_windows_ref(records, anchors, window_before, window_after, flag_filter):
    """generate_indel_pileups.py:306-338 on SAM-like records (dict name, flag, pos0, cigar [(op, len)], seq): for each
    anchor (1-based) the [(record index, query_sequence[max(0, q - wb) : q + wa])] of the reads in the pileup there, with
    q = pysam's query_position_or_next, by expanding every CIGAR base by base."""
    out = []
    maps = []
    for r in records:
        ref_to_q = {}                     # 1-based ref pos -> query index or ('next', index)
        rp, qp = r["pos0"] + 1, 0
        pending = []                      # deleted / skipped reference positions waiting for the next aligned query base
        for op, ln in r["cigar"]:
            if op in "M=X":
                for _ in range(ln):
                    for d in pending:
                        ref_to_q[d] = qp
                    pending = []
                    ref_to_q[rp] = qp
                    rp += 1
                    qp += 1
            elif op in "IS":
                qp += ln
            elif op in "DN":
                for _ in range(ln):
                    pending.append(rp)
                    rp += 1
        for d in pending:
            ref_to_q[d] = qp
        maps.append((ref_to_q, r["pos0"] + 1, rp))
    for a in anchors:
        here = []
        for k, r in enumerate(records):
            m, s, e = maps[k]
            if (r["flag"] & flag_filter) or (r["flag"] & 4) or not (s <= a < e):
                continue
            q = m[a]
            here.append((k, r["seq"][max(0, q - window_before):q + window_after]))
        out.append(here)
    return out


# ---- the banded form of the star alignment (the product's default: nc_pipe.hip k_fill_band / k_trace_band), restated independently.
# The band is derived from the read's own CIGAR inside the window; an alignment whose band would be wider than 64 diagonals, or whose
# banded path touches an edge diagonal, is the full-matrix alignment (nw_cigar_free_tail_ref).
BAND_MARGIN = 6


def window_band_ref(record, anchor, window_after):
    """(dmin, dmax) of the diagonals j - i (window column - window base, both 1-based) the CIGAR's own path visits inside
    query_sequence[q : q + window_after], q = query_position_or_next at `anchor`; 0 is always inside (the path starts at the origin).
    Inserted and soft-clipped bases have no column: each lowers the diagonal by one; a deletion raises it by its length, also when it
    directly follows the window's last base."""
    base_col, gap_before = [], []                    # per query base: reference position or None; deleted length right before it
    rp, pend = record["pos0"] + 1, 0
    for op, ln in record["cigar"]:
        if op in "M=X":
            for _ in range(ln):
                base_col.append(rp)
                gap_before.append(pend)
                pend = 0
                rp += 1
        elif op in "IS":
            for _ in range(ln):
                base_col.append(None)
                gap_before.append(pend)
                pend = 0
        elif op in "DN":
            pend += ln
            rp += ln
    gap_before.append(pend)                          # behind the last base
    q = next((k for k, c in enumerate(base_col) if c is not None and c >= anchor), len(base_col))
    n = min(window_after, len(base_col) - q)
    dmin = dmax = d = 0
    if n > 0:
        d = base_col[q] - anchor                     # the anchor itself may be deleted: the window opens on the next aligned base
        dmax = max(dmax, d)
    for t in range(n):
        qi = q + t
        if t > 0:
            d += gap_before[qi]
            dmax = max(dmax, d)
        if base_col[qi] is None:
            d -= 1
            dmin = min(dmin, d)
    if n > 0:
        # what follows the last base at the same column: the rest of an insertion (not emitted), then a deletion
        k = q + n
        while k < len(base_col) and base_col[k] is None and gap_before[k] == 0:
            k += 1
        d += gap_before[k]
        dmax = max(dmax, d)
    return dmin, dmax


def band_of(dmin, dmax, n1, n2, margin=BAND_MARGIN):
    """-> (lo, B) of the band [lo, lo + B) or None (full matrix): B = 32 or 64, the slack split evenly, lo even.  A read that ends inside
    the window (n1 + dmax well below n2) leaves last-row cells to the right of its path, where the free tail may end after a jump: the
    band then reaches the end of the last row (hi >= n2 - n1)."""
    dmax = max(dmax, n2 - n1 - margin + 1)
    w = dmax - dmin
    if w + 2 * margin <= 31:
        B = 32
    elif w + 2 * margin <= 63:
        B = 64
    else:
        return None
    lo = dmin - ((B - 1 - w) >> 1)
    lo -= lo & 1
    return lo, B


def nw_cigar_band_free_tail_ref(s1, s2, lo, B, open_=9, extend=1, match=20, mismatch=-10):
    """nw_cigar_free_tail_ref restricted to the diagonals lo <= j - i < lo + B (everything else is minus infinity), same tie rules;
    -> None when the traceback reads a cell of an edge diagonal (the caller then aligns on the full matrix)"""
    n1, n2 = len(s1), len(s2)
    if n1 == 0 or n2 == 0:
        return nw_cigar_ref(s1, s2, open_, extend, match, mismatch)
    NEG = -10 ** 9
    hi = lo + B - 1
    H = [[NEG] * (n2 + 1) for _ in range(n1 + 1)]
    E = [[NEG] * (n2 + 1) for _ in range(n1 + 1)]
    F = [[NEG] * (n2 + 1) for _ in range(n1 + 1)]
    who = [[0] * (n2 + 1) for _ in range(n1 + 1)]
    e_ext = [[False] * (n2 + 1) for _ in range(n1 + 1)]
    f_ext = [[False] * (n2 + 1) for _ in range(n1 + 1)]
    H[0][0] = 0
    for j in range(1, min(n2, hi) + 1):
        H[0][j] = E[0][j] = -open_ - (j - 1) * extend
        who[0][j] = 1
        e_ext[0][j] = j > 1
    for i in range(1, n1 + 1):
        if -i >= lo:
            H[i][0] = F[i][0] = -open_ - (i - 1) * extend
            who[i][0] = 2
            f_ext[i][0] = i > 1
        for j in range(max(1, i + lo), min(n2, i + hi) + 1):
            eo, ee = H[i][j - 1] - open_, E[i][j - 1] - extend
            e_ext[i][j] = ee >= eo
            E[i][j] = max(eo, ee)
            fo, fe = H[i - 1][j] - open_, F[i - 1][j] - extend
            f_ext[i][j] = fe >= fo
            F[i][j] = max(fo, fe)
            d = H[i - 1][j - 1] + (match if s1[i - 1] == s2[j - 1] else mismatch)
            h, w = d, 0
            if E[i][j] > h:
                h, w = E[i][j], 1
            if F[i][j] > h:
                h, w = F[i][j], 2
            H[i][j], who[i][j] = h, w
    best, bi, bj = NEG * 2, n1, n2
    for j in range(n2, -1, -1):                                 # last row: ties to the larger column
        if lo <= j - n1 <= hi and H[n1][j] > best:
            best, bi, bj = H[n1][j], n1, j
    for i in range(n1 - 1, -1, -1):                             # last column: strictly better only, ties to the larger row
        if lo <= n2 - i <= hi and H[i][n2] > best:
            best, bi, bj = H[i][n2], i, n2
    ops = []
    i, j, state = bi, bj, None
    while i > 0 or j > 0:
        if i > 0 and j > 0 and (j - i - lo <= 0 or j - i - lo >= B - 1):
            return None
        if state is None:
            if who[i][j] == 0:
                ops.append(7 if s1[i - 1] == s2[j - 1] else 8)
                i, j = i - 1, j - 1
                continue
            state = who[i][j]
        if state == 1:
            ops.append(2)
            ext = e_ext[i][j]
            j -= 1
        else:
            ops.append(1)
            ext = f_ext[i][j]
            i -= 1
        if not ext:
            state = None
    ops.reverse()
    out = []
    for o in ops + [2] * (n2 - bj) + [1] * (n1 - bi):
        if out and out[-1][0] == o:
            out[-1][1] += 1
        else:
            out.append([o, 1])
    return [(o, c) for o, c in out]


def star_cigars_banded_ref(seqs, bands, ref, open_=9, extend=1, match=20, mismatch=-10, margin=BAND_MARGIN):
    """the pairwise alignments of the product's star alignment: on the band `bands[k] = (dmin, dmax)` allows, else / after an edge
    touch on the full matrix.  -> (cigars, [how each was aligned: 32, 64, 'width', 'edge'])"""
    cig, how = [], []
    for q, (dmin, dmax) in zip(seqs, bands):
        b = band_of(dmin, dmax, len(q), len(ref), margin)
        c = None
        if b is not None:
            c = nw_cigar_band_free_tail_ref(q, ref, b[0], b[1], open_, extend, match, mismatch)
        how.append(b[1] if c is not None else ("width" if b is None else "edge"))
        cig.append(c if c is not None else nw_cigar_free_tail_ref(q, ref, open_, extend, match, mismatch))
    return cig, how


# ------------------------------------------------------------------------------------------------- device indel pipeline sample
def records_from_indel_pack(read_start, read_end, codes_of, ev_of, ins_of, names=None, flags=None):
    """SAM-like records (what read_windows_ref takes) rebuilt from the pack form of reads: `codes_of(r)` = uint8 code per spanned
    reference position (4 = deleted or N), `ev_of(r)` = [(column, +ins / -del length)], `ins_of(r, k)` = codes of the k-th event's
    inserted bases.  The CIGAR walk of tests/bamio.world_to_records, stated independently of the device's window rebuild."""
    lut = "AGTCNNNN"
    recs = []
    for r in range(len(read_start)):
        s, e = int(read_start[r]), int(read_end[r])
        c = codes_of(r)
        cig, seq, p = [], [], s
        for k, (ep, el) in enumerate(ev_of(r)):
            cig.append(("M", ep - p + 1))
            seq.append("".join(lut[x] for x in c[p - s:ep - s + 1]))
            p = ep + 1
            if el > 0:
                cig.append(("I", el))
                seq.append("".join(lut[x] for x in ins_of(r, k)))
            else:
                cig.append(("D", -el))
                p += -el
        if p < e:
            cig.append(("M", e - p))
            seq.append("".join(lut[x] for x in c[p - s:e - s]))
        recs.append(dict(name=names[r] if names else "r%d" % r, flag=int(flags[r]) if flags is not None else 0, pos0=s - 1, cigar=cig, seq="".join(seq)))
    return recs


def indel_site_ref(records, hap, ps, ref, v_pos, window_after, mincov, maxcov, aligner=None, scoring=(25, 1, 20, -10), haploid=False, band=False,
                   how_out=None):
    """One pass-2 site from records: read sets (first-maxcov policy), star alignment (`aligner(names, seqs, ref)` or the pure-Python
    star_msa_ref; band=True: every pairwise alignment on the band its CIGAR allows, as the device pipeline runs it), msa()'s tensor by
    the C oracle.  -> None when the site fails the set-size tests, else (x float32 [S,5,128,2], [consensus strings], reference window, phase)"""
    win = ref[v_pos - 1:min(len(ref), v_pos + window_after)]
    if any(ch not in "AGTC" for ch in win):
        return None
    here = read_windows_ref(records, [v_pos], 0, window_after, 0)[0]
    sets = [[(k, s) for k, s in here]] if haploid else [[(k, s) for k, s in here if hap[k] == 1], [(k, s) for k, s in here if hap[k] == 2], list(here)]
    sets = [x[:maxcov] for x in sets]
    need = [mincov] if haploid else [2, 2, mincov]
    if any(len(x) < n for x, n in zip(sets, need)):
        return None
    sym = {"A": 0, "G": 1, "T": 2, "C": 3, "-": 4}
    xs, cns = [], []
    for st in sets:
        seqs = [s for _, s in st]
        if aligner:
            rows, ref_row = aligner(["r%d" % k for k, _ in st], seqs, win)
        elif band:
            cig, how = star_cigars_banded_ref(seqs, [window_band_ref(records[k], v_pos, window_after) for k, _ in st], win, *scoring)
            if how_out is not None:
                how_out.extend(how)
            rows, ref_row = star_msa_ref(seqs, win, *scoring, cigars=cig)
        else:
            rows, ref_row = star_msa_ref(seqs, win, *scoring)
        mat = np.array([[sym.get(c, 4) for c in row] for row in rows], np.uint8)
        x, c = indel_tensor(mat, np.array([sym[c] for c in ref_row], np.uint8))
        xs.append(x)
        cns.append("".join("AGTC"[k] for k in c if k < 4))
    phase = 0 if haploid or not sets[0] else int(ps[sets[0][0][0]])
    return np.stack(xs).astype(np.float32), cns, win, phase
