"""CPU-only: the native host packer (nc_pack_plan / nc_pack_fill) -- slot alignment, padding, tile index."""
import ctypes as C

import numpy as np
import pytest

from nanocaller_amd import _lib
from nanocaller_amd.pack import pack_world
from nanocaller_amd.synth import FLAG_FILTER_DEFAULT, make_world
from tests.util import load_world


@pytest.mark.parametrize("tile_size", [1024, 2048, 4096])
def test_pack_roundtrip(tile_size):
    w = make_world(seed=3, length=30_000, depth=15, read_len_scale=0.2)
    hp = pack_world(w, tile_size=tile_size)
    assert hp.tile_pos0 % 16 == 0 and hp.tile_pos0 <= 1 and hp.codes.size % 16 == 0
    keep = (w.read_flag & FLAG_FILTER_DEFAULT) == 0
    ents = hp.tile_ent
    # every kept read appears in exactly the tiles it overlaps, in coordinate order, with its codes recoverable
    seen = {}
    for t in range(hp.n_tiles):
        lo = hp.tile_pos0 + t * tile_size
        e = ents[hp.tile_off[t]:hp.tile_off[t + 1]]
        assert np.all(e["start"] < lo + tile_size) and np.all(e["end"] > lo)
        assert np.all(np.diff(e["start"]) >= 0)
        for x in e:
            seen.setdefault((int(x["start"]), int(x["end"]), int(x["base_flag"])), 0)
            seen[(int(x["start"]), int(x["end"]), int(x["base_flag"]))] += 1
    kept = np.nonzero(keep)[0]
    assert len(seen) == len(kept)
    by_key = {}
    for k in seen:
        by_key.setdefault((k[0], k[1]), []).append(k)
    used = 0
    for i in kept:
        s, e = int(w.read_start[i]), int(w.read_end[i])
        cands = by_key[(s, e)]
        ok = False
        for k in cands:
            base = k[2] & ~15
            assert base % 16 == 0
            got = hp.codes[base + s:base + e]
            if np.array_equal(got, w.read_codes(i)) and (k[2] & 1) == int((w.read_flag[i] & 16) != 0):
                ok = True
                # slot padding outside [start, end) is NC_CODE_ABSENT
                lo16, hi16 = s & ~15, (e + 15) & ~15
                assert np.all(hp.codes[base + lo16:base + s] == 7) and np.all(hp.codes[base + e:base + hi16] == 7)
                ntiles = (e - 1 - hp.tile_pos0) // tile_size - (s - hp.tile_pos0) // tile_size + 1
                assert seen[k] == ntiles
        assert ok
        used += e - s
    assert int((hp.codes != 7).sum()) == used
    # reference codes on the tile grid, soft-masked / N -> 4
    assert hp.ref_code.size == hp.n_tiles * tile_size
    from nanocaller_amd.synth import world_ref_codes
    rc = world_ref_codes(w)
    assert np.array_equal(hp.ref_code[1 - hp.tile_pos0:1 - hp.tile_pos0 + w.length], rc)
    assert (rc == 4).sum() > 100


def test_pack_rejects_unsorted_and_bad_args():
    L = _lib.lib()
    s = np.array([100, 50], np.int32)
    e = np.array([200, 150], np.int32)
    cl, ne = C.c_int64(), C.c_int64()
    tp, nt = C.c_int32(), C.c_int32()
    assert L.nc_pack_plan(2, _lib.npp(s), _lib.npp(e), None, 2048, 1, 1000, C.byref(cl), C.byref(tp), C.byref(nt), C.byref(ne)) == -1
    s2 = np.array([50, 100], np.int32)
    assert L.nc_pack_plan(2, _lib.npp(s2), _lib.npp(e[::-1].copy()), None, 1000, 1, 1000, C.byref(cl), C.byref(tp), C.byref(nt), C.byref(ne)) == -1
    assert L.nc_pack_plan(2, _lib.npp(s2), _lib.npp(e[::-1].copy()), None, 2048, 1, 1000, C.byref(cl), C.byref(tp), C.byref(nt), C.byref(ne)) == 0
    assert nt.value == 1 and ne.value == 2 and cl.value % 16 == 0


def test_pack_exclusion_and_empty_region():
    w = load_world("ont")
    hp = pack_world(w, exclude=[(55_000, 58_000)])
    assert np.all(hp.ref_code[55_000 - hp.tile_pos0:58_000 - hp.tile_pos0] == 4)
    assert hp.ref_code[58_000 - hp.tile_pos0] != 4 or w.ref[58_000 - 1] not in "AGTC"
    empty = make_world(seed=1, length=5000, depth=0.001, read_len_scale=0.05)
    hp2 = pack_world(empty)
    assert hp2.n_tiles >= 1 and hp2.tile_off[-1] == hp2.tile_ent.shape[0]


def test_pack_many_reads_threaded_fill():
    """>= 4096 reads: the packer's copy loop runs on several host threads over disjoint slot ranges -- every kept read's codes
    and the padding between slots must be exactly as in the single-threaded layout (vectorised check)"""
    w = make_world(seed=9, length=120_000, depth=40, read_len_scale=0.04, odd_flag_frac=0.1)
    assert w.n_reads >= 8192
    hp = pack_world(w, tile_size=2048)
    keep = (w.read_flag & FLAG_FILTER_DEFAULT) == 0
    # expected layout: slots of kept reads back to back, base = cursor - floor16(start)
    cur = 0
    exp = np.full(hp.codes.size, 7, np.uint8)
    for i in np.nonzero(keep)[0]:
        s, e = int(w.read_start[i]), int(w.read_end[i])
        lo, hi = s & ~15, (e + 15) & ~15
        exp[cur + (s - lo):cur + (s - lo) + (e - s)] = w.read_codes(i)
        cur += hi - lo
    assert cur + 16 == hp.codes.size and np.array_equal(hp.codes, exp)
    assert hp.tile_off[-1] == len(hp.tile_ent)


def _plp_kept(starts0, ends0, maxcnt):
    """Restatement of htslib's pileup buffer for ONE contig (bam_plp_push / bam_plp_next / bam_plp_auto, sam.c): which records
    enter the buffer.  starts0 / ends0: 0-based start and exclusive end in file order.  The buffer is the list of nodes with a
    dummy tail (the memory pool counts it: `cnt` = records held + 1)."""
    nodes = []                      # held records (beg, end, index), list order = push order
    pos, max_pos = 0, -1
    kept = []
    nxt, n = 0, len(starts0)
    eof = False

    def push(i):
        nonlocal max_pos
        b, e = starts0[i], ends0[i]
        if pos == b and len(nodes) + 1 > maxcnt:
            return
        kept.append(i)
        max_pos = b
        if e > pos:
            nodes.append((b, e, i))

    while True:
        # bam_plp_next: emit columns while the look-ahead record starts beyond the current column
        progressed = False
        while eof or max_pos > pos:
            nodes[:] = [nd for nd in nodes if nd[1] > pos]              # release records that ended at or before this column
            if nodes and pos < nodes[0][0]:
                pos = nodes[0][0]
            else:
                pos += 1
            progressed = True
            if eof and not nodes:
                return kept
        if eof and not nodes:
            return kept
        if nxt < n:
            push(nxt)
            nxt += 1
        else:
            eof = True
        if not progressed and eof and not nodes:
            return kept


def test_pileup_depth_cap_follows_the_htslib_buffer_rule():
    """pack.pileup_depth_cap (what pysam's default max_depth = 8000 does to reads in very deep regions) against a restatement of
    htslib's buffer logic, at small limits so that the rule triggers; below the limit nothing changes"""
    from nanocaller_amd.pack import PILEUP_MAX_DEPTH, pileup_depth_cap
    assert PILEUP_MAX_DEPTH == 8000
    rng = np.random.Generator(np.random.PCG64(77))
    n_dropped = 0
    for case in range(30):
        n = int(rng.integers(50, 400))
        span = int(rng.integers(20, 300))
        s0 = np.sort(rng.integers(0, span, size=n))
        if case % 3 == 0:
            s0 = np.sort(rng.choice(s0[: max(3, n // 10)], size=n))                 # many records per start position
        e0 = s0 + rng.integers(1, 120, size=n)
        flag_keep = (rng.random(n) < 0.9).astype(np.uint8)                          # records the flag filter removed never reach the buffer
        for maxcnt in (3, 8, 25, 10_000):
            idx = np.flatnonzero(flag_keep)
            exp = np.zeros(n, np.uint8)
            exp[idx[_plp_kept(s0[idx].tolist(), e0[idx].tolist(), maxcnt)]] = 1
            got = pileup_depth_cap(s0 + 1, e0 + 1, flag_keep, max_depth=maxcnt)     # the package's arrays are 1-based
            assert np.array_equal(got, exp), (case, maxcnt)
            n_dropped += int(flag_keep.sum() - got.sum())
            if maxcnt == 10_000:
                assert np.array_equal(got, flag_keep)
    assert n_dropped > 2000
    # a pile of 9,000 records on one window: the first 8,000 that are held together stay, later ones starting at a column already
    # at the limit go, a record that is the first of its start position always enters
    s = np.concatenate([np.full(8500, 100), np.full(300, 101), [102], np.full(200, 5000)]).astype(np.int64)
    e = s + 50
    k = pileup_depth_cap(s, e, np.ones(s.size, np.uint8))
    assert k[:8000].all() and not k[8000:8500].any() and k[8500] == 1 and not k[8501:8800].any() and k[8800] == 1 and k[8801:].all()
