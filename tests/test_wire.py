"""The reference-difference transfer form of the read pack (nc_wire_*, nanocaller_amd/wire.py): the host builder against a
numpy restatement of the expansion (CPU), and nc_wire_expand on the GPU byte for byte against nc_pack_fill's codes."""
import os

import numpy as np
import pytest

from nanocaller_amd.pack import pack_world
from nanocaller_amd.wire import build_wire_from_world, ref_wire_from_string

from util import load_world


def _expand_numpy(wp):
    """what nc_wire_expand writes, restated with numpy on the host arrays of a WirePack"""
    rs, re_, so = wp.host("rd_start").astype(np.int64), wp.host("rd_end").astype(np.int64), wp.host("slot_off")
    nib, bo, ev = wp.host("ref_nib"), wp.host("blk_off"), (wp.host("events") if "events" in wp.sections else None)
    refw = np.empty(2 * nib.size, np.uint8)                             # two positions per byte on the wire (nc_wire_ref_unpack)
    refw[0::2], refw[1::2] = nib & 15, nib >> 4
    codes = np.full(wp.codes_len, 7, np.uint8)
    for r in range(wp.n_reads):
        base = so[r] - (rs[r] & ~15)
        ri = np.arange(rs[r], re_[r]) - wp.tile_pos0
        ok = (ri >= 0) & (ri < wp.ref_len)
        codes[base + rs[r]:base + re_[r]] = np.where(ok, refw[np.clip(ri, 0, wp.ref_len - 1)] & 7, 4)
    if "ev_bytes" in wp.sections:
        # the events one byte each (nc_wire_build2): columns skipped since the block's previous event << 2 (63: a filler, no event) | which of the
        # four codes other than the predicted one
        evb = wp.host("ev_bytes").astype(np.int64)
        bo64 = bo.astype(np.int64)
        for b in np.flatnonzero(np.diff(bo64)):
            g = evb[bo64[b]:bo64[b + 1]] >> 2
            k = evb[bo64[b]:bo64[b + 1]] & 3
            offs = np.cumsum(np.where(g == 63, 63, g + 1)) - 1
            is_ev = g != 63
            o, k = b * 1024 + offs[is_ev], k[is_ev]
            pred = codes[o].astype(np.int64)
            assert np.all(pred <= 4) and np.all(offs[is_ev] < 1024)
            codes[o] = np.where(pred < 4, np.where(k == 3, 4, (pred + 1 + k) & 3), k)
    else:
        blk = np.repeat(np.arange(wp.n_blocks), np.diff(bo.astype(np.int64)))
        codes[blk * 1024 + (ev & 0x3ff)] = ev >> 12
    ref_code = np.where(refw & 8, 4, refw & 7).astype(np.uint8)
    return codes, ref_code


def _n_diff_events(wp):
    """difference events of a pack in either form (the one-byte form's fillers are no events)"""
    if "ev_bytes" in wp.sections:
        evb = wp.host("ev_bytes")[:int(wp.host("blk_off")[-1])]
        return int(np.count_nonzero((evb >> 2) != 63))
    return wp.n_events


def _expand_events_numpy(wp):
    """the transfer form of the indel events (nc_indel_events_pack / _pack8) restated: ev_pos / ev_len / ins_off as nc_indel_events_expand[8] writes them"""
    me = wp.meta["indel_events"]
    n_ev, off, start = me["n_ev"], wp.host("ev_off"), wp.host("rd_start")
    big = {int(i): (int(p), int(ln)) for i, p, ln in zip(wp.host("ev_big_idx")[:me["n_big"]], wp.host("ev_big_pos")[:me["n_big"]], wp.host("ev_big_len")[:me["n_big"]])}
    pos, ln = np.zeros(n_ev, np.int32), np.zeros(n_ev, np.int32)

    def two_byte(raw, e, prev):                                         # distance (bits 0-10; 0xFFFF: side table) | signed 5-bit length << 11
        if raw == 0xFFFF:
            return big[e]
        return prev + (raw & 0x7ff), ((raw >> 11) ^ 16) - 16
    if me.get("ev8"):
        # one byte per event: distance << 2 | length code (+1 -1 +2 -2); 0xFF: the next entry of the two-byte array, which starts at read_esc_off[r] for read r
        b8, d16x, reo = wp.host("ev_b8")[:n_ev].astype(np.int64), wp.host("ev_d16").astype(np.int64), wp.host("read_esc_off")
        for r in range(wp.n_indel_reads):
            prev, x = int(start[r]), int(reo[r])
            for e in range(int(off[r]), int(off[r + 1])):
                if b8[e] == 0xFF:
                    prev, ln[e] = two_byte(int(d16x[x]), e, prev)
                    x += 1
                else:
                    prev, ln[e] = prev + int(b8[e] >> 2), (1, -1, 2, -2)[int(b8[e] & 3)]
                pos[e] = prev
            assert x == int(reo[r + 1])
    else:
        raw = wp.host("ev_d16")[:n_ev].astype(np.int64)
        for r in range(wp.n_indel_reads):
            prev = int(start[r])
            for e in range(int(off[r]), int(off[r + 1])):
                prev, ln[e] = two_byte(int(raw[e]), e, prev)
                pos[e] = prev
    ins = np.zeros(n_ev + 1, np.int64)
    np.cumsum(np.maximum(ln, 0), out=ins[1:])
    assert np.array_equal(wp.host("read_ins_off"), ins[off[:wp.n_indel_reads + 1]])
    return pos, ln, ins


def test_indel_events_side_table():
    """distances >= 0xFFFF and lengths beyond a signed byte go through the side table; a too small table is reported with the size it needs"""
    import ctypes as C

    from nanocaller_amd import _lib
    L = _lib.lib()
    start = np.array([100, 500_000], np.int32)
    off = np.array([0, 4, 7], np.int32)
    pos = np.array([100, 70_000, 70_001, 200_000, 500_010, 500_020, 565_555], np.int32)
    ln = np.array([1, -127, 128, -3, -200, 127, 5], np.int32)
    d16, l8, rio = np.zeros(7, np.uint16), np.zeros(7, np.int8), np.zeros(3, np.int32)
    bi, bp, bl = (np.zeros(8, np.int32) for _ in range(3))
    nb = C.c_int64()
    assert L.nc_indel_events_pack(2, _lib.npp(start), _lib.npp(off), _lib.npp(pos), _lib.npp(ln), _lib.npp(d16), _lib.npp(l8), _lib.npp(rio), 2,
                                  _lib.npp(bi), _lib.npp(bp), _lib.npp(bl), C.byref(nb)) == _lib.NC_ERR_CAPACITY and nb.value == 5
    assert L.nc_indel_events_pack(2, _lib.npp(start), _lib.npp(off), _lib.npp(pos), _lib.npp(ln), _lib.npp(d16), _lib.npp(l8), _lib.npp(rio), 8,
                                  _lib.npp(bi), _lib.npp(bp), _lib.npp(bl), C.byref(nb)) == _lib.NC_OK
    assert bi[:5].tolist() == [1, 2, 3, 4, 6] and bp[:5].tolist() == [70_000, 70_001, 200_000, 500_010, 565_555] and bl[:5].tolist() == [-127, 128, -3, -200, 5]
    assert d16.tolist() == [0, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 10, 0xFFFF] and l8[[0, 5]].tolist() == [1, 127]
    assert rio.tolist() == [0, 129, 261]
    bad = pos.copy()
    bad[2] = 60_000
    assert L.nc_indel_events_pack(2, _lib.npp(start), _lib.npp(off), _lib.npp(bad), _lib.npp(ln), _lib.npp(d16), _lib.npp(l8), _lib.npp(rio), 8,
                                  _lib.npp(bi), _lib.npp(bp), _lib.npp(bl), C.byref(nb)) == -1                    # NC_ERR_ARG: the events of a read ascend


CASES = [("ont", False, None, 2048), ("ont", True, [(55_000, 58_000), (30_000, 41_000)], 1024), ("hifi", False, None, 4096),
         ("deep", False, None, 2048), ("indel", False, None, 2048)]


def _consistent_deletions(w):
    """the synthetic indel world with its codes made consistent with its events: every deleted column carries code 4, as a pileup's '*' decodes
    (the stock world plants events without touching the codes: the builder then keeps the plain form)"""
    import copy
    w = copy.copy(w)
    w.codes = w.codes.copy()
    ev_off, ev_pos, ev_len = w.meta["events"]
    for r in range(len(w.read_start)):
        s_, e_ = int(w.read_start[r]), int(w.read_end[r])
        for e in range(int(ev_off[r]), int(ev_off[r + 1])):
            if ev_len[e] < 0:
                a, b = max(int(ev_pos[e]) + 1, s_), min(int(ev_pos[e]) + 1 - int(ev_len[e]), e_)
                w.codes[int(w.read_off[r]) + a - s_:int(w.read_off[r]) + b - s_] = 4
    return w


CASES = CASES + [("indel_consistent", False, None, 2048)]


@pytest.mark.parametrize("evb", ["2", "1", "0"])
@pytest.mark.parametrize("name,supp,excl,tile", CASES)
def test_wire_build_reconstructs_the_packed_codes(name, supp, excl, tile, evb, monkeypatch):
    monkeypatch.setenv("NC_WIRE_EVB", evb)                               # the events one byte each: every pack (default) / packs with indel events / never
    w = _consistent_deletions(load_world("indel")) if name == "indel_consistent" else load_world(name)
    assert ((w.meta or {}).get("events") is None) or name.startswith("indel")
    hp = pack_world(w, supplementary=supp, exclude=excl, tile_size=tile)
    wp = build_wire_from_world(w, supplementary=supp, exclude=excl, tile_size=tile, pin=False)
    codes, ref_code = _expand_numpy(wp)
    me = (wp.meta or {}).get("indel_events")
    assert bool(me and me.get("del_implied")) == (name == "indel_consistent")
    if me and me.get("del_implied"):
        # in blocks inside one read the deleted columns are implied by the read's own deletion events (nc_wire_build_del) and written by the
        # expansion from the block's cursor into the events (nc_wire_expand_del, restated): without that step the codes differ exactly there
        assert not np.array_equal(codes, hp.codes)
        ev_pos, ev_len, _ = _expand_events_numpy(wp)
        off, rs, so = wp.host("ev_off"), wp.host("rd_start").astype(np.int64), wp.host("slot_off")
        be, br_ = wp.host("blk_ev"), wp.host("blk_read")
        n_written = 0
        assert np.any(be != 0xffffffff)
        for b in np.nonzero(be != 0xffffffff)[0]:
            r = int(br_[b])
            p0 = (int(rs[r]) & ~15) + (int(b) * 1024 - int(so[r]))    # position of the block's first byte
            assert int(off[r]) <= int(be[b]) <= int(off[r + 1])
            assert int(be[b]) == int(off[r]) or ev_pos[be[b] - 1] + max(0, -ev_len[be[b] - 1]) < p0       # nothing before the cursor reaches the block
            for e in range(int(be[b]), int(off[r + 1])):
                if ev_pos[e] >= p0 + 1024:
                    break
                if ev_len[e] < 0:
                    a, z = max(int(ev_pos[e]) + 1 - p0, 0), min(int(ev_pos[e]) + 1 - int(ev_len[e]) - p0, 1024)
                    codes[b * 1024 + a:b * 1024 + z] = 4
                    n_written += max(0, z - a)
        os.environ["NC_WIRE_DEL_IMPLIED"] = "0"
        try:
            full = build_wire_from_world(w, supplementary=supp, exclude=excl, tile_size=tile, pin=False)
        finally:
            del os.environ["NC_WIRE_DEL_IMPLIED"]
        assert np.array_equal(_expand_numpy(full)[0], hp.codes)
        assert 0 < _n_diff_events(full) - _n_diff_events(wp) <= n_written            # every dropped event is a deleted column (a deleted column may also carry the reference's code 4: never an event)
    assert wp.codes_len == hp.codes.size and np.array_equal(codes, hp.codes)
    assert np.array_equal(ref_code, hp.ref_code)
    assert np.array_equal(wp.host("tile_off"), hp.tile_off)
    assert wp.host("tile_ent").tobytes() == hp.tile_ent.tobytes()
    assert (wp.tile_size, wp.tile_pos0, wp.n_tiles, wp.n_entries) == (hp.tile_size, hp.tile_pos0, hp.n_tiles, hp.tile_ent.shape[0])
    if hp.ev_off is not None:
        for k, a in (("ev_off", hp.ev_off), ("read_hap", hp.read_hap)):
            assert np.array_equal(wp.host(k)[:a.size], a)
        ev_pos, ev_len, ins_off = _expand_events_numpy(wp)
        assert np.array_equal(ev_pos, hp.ev_pos) and np.array_equal(ev_len, hp.ev_len)
        assert np.array_equal(ins_off[1:], np.cumsum(np.maximum(hp.ev_len, 0)))
        assert 2 * ev_pos.size + 12 * wp.meta["indel_events"]["n_big"] < 0.3 * 12 * max(ev_pos.size, 1) or ev_pos.size < 100
    # the point of it: far fewer bytes than 1 B per pileup entry (ONT worlds: 4 % substitutions + 4 % deletions)
    entries = int((w.read_end - w.read_start).sum())
    bo = wp.host("blk_off")
    if "ev_bytes" in wp.sections:                                        # one byte an event (+ a filler per 63 columns without one)
        assert (name.startswith("indel") or os.environ.get("NC_WIRE_EVB", "2") == "2") and wp.n_events == 0 and "events" not in wp.sections
        assert int(bo[-1]) < 0.125 * entries and int(bo[-1]) < 2 * _n_diff_events(wp)          # (the builder keeps the two-byte form when that is shorter: HiFi)
    else:
        assert 2 * wp.n_events < 0.25 * entries and (name != "ont" or os.environ.get("NC_WIRE_EVB", "2") != "2")
        ev = wp.host("events")
        assert bo[-1] == wp.n_events and np.all((ev >> 12) <= 4)
    assert bo[0] == 0 and np.all(np.diff(bo.astype(np.int64)) >= 0)
    br, so = wp.host("blk_read"), wp.host("slot_off")
    b0 = np.arange(wp.n_blocks, dtype=np.int64) * 1024
    assert np.array_equal(br, np.minimum(np.searchsorted(so[1:], b0, side="right"), wp.n_reads))   # first read with slot end > block start


def test_ref_wire_bytes():
    rw = ref_wire_from_string("AGTCagtcNnR", exclude=[(2, 4)])
    assert rw.tolist() == [0, 1 | 8, 2 | 8, 3, 0 | 8, 1 | 8, 2 | 8, 3 | 8, 4 | 8, 4 | 8, 4 | 8]


def test_indel_events_two_byte_form():
    """l8 = NULL: distance (< 0x7ff) | signed 5-bit length (-16 .. 15) << 11 in ONE uint16 per event; the rest through the side table"""
    import ctypes as C

    from nanocaller_amd import _lib
    L = _lib.lib()
    start = np.array([1000], np.int32)
    off = np.array([0, 8], np.int32)
    pos = np.array([1000, 1000 + 0x7fe, 1000 + 0x7fe + 0x7ff, 5100, 5101, 5102, 5103, 5104], np.int32)
    ln = np.array([15, -16, 3, 16, -17, -1, 1, -16], np.int32)
    d16, rio = np.zeros(8, np.uint16), np.zeros(2, np.int32)
    bi, bp, bl = (np.zeros(8, np.int32) for _ in range(3))
    nb = C.c_int64()
    assert L.nc_indel_events_pack(1, _lib.npp(start), _lib.npp(off), _lib.npp(pos), _lib.npp(ln), _lib.npp(d16), None, _lib.npp(rio), 8,
                                  _lib.npp(bi), _lib.npp(bp), _lib.npp(bl), C.byref(nb)) == _lib.NC_OK
    assert nb.value == 3 and bi[:3].tolist() == [2, 3, 4] and bl[:3].tolist() == [3, 16, -17]            # distance 0x7ff; lengths beyond 5 bits
    raw = d16.astype(np.int64)
    assert raw[[2, 3, 4]].tolist() == [0xFFFF] * 3
    keep = [0, 1, 5, 6, 7]
    assert (raw[keep] & 0x7ff).tolist() == [0, 0x7fe, 1, 1, 1] and (((raw[keep] >> 11) ^ 16) - 16).tolist() == [15, -16, -1, 1, -16]
    assert rio.tolist() == [0, 15 + 3 + 16 + 1]


def test_indel_events_one_byte_form():
    """nc_indel_events_pack8: distance (<= 62) << 2 | length code (+1 -1 +2 -2) in one byte; 0xFF = the next entry of the two-byte array (whose own
    0xFFFF goes on to the side table); read_esc_off = where a read's escapes start"""
    import ctypes as C

    from nanocaller_amd import _lib
    L = _lib.lib()
    start = np.array([1000, 9000], np.int32)
    off = np.array([0, 7, 10], np.int32)
    pos = np.array([1000, 1062, 1125, 1126, 1127, 4000, 4001, 9000, 9010, 9010 + 0x7ff], np.int32)
    ln = np.array([1, -2, 2, 3, -1, 1, -40, 2, -1, 1], np.int32)
    b8, d16x, reo, rio = np.zeros(10, np.uint8), np.zeros(10, np.uint16), np.zeros(3, np.int32), np.zeros(3, np.int32)
    bi, bp, bl = (np.zeros(8, np.int32) for _ in range(3))
    nb, ne = C.c_int64(), C.c_int64()
    args = (2, _lib.npp(start), _lib.npp(off), _lib.npp(pos), _lib.npp(ln), _lib.npp(b8), _lib.npp(d16x))
    assert L.nc_indel_events_pack8(*args, 1, _lib.npp(reo), _lib.npp(rio), 8, _lib.npp(bi), _lib.npp(bp), _lib.npp(bl), C.byref(ne), C.byref(nb)) == _lib.NC_ERR_CAPACITY and ne.value == 5
    assert L.nc_indel_events_pack8(*args, 10, _lib.npp(reo), _lib.npp(rio), 8, _lib.npp(bi), _lib.npp(bp), _lib.npp(bl), C.byref(ne), C.byref(nb)) == _lib.NC_OK
    # events 2 (distance 63), 3 (length 3), 5 (distance 2873), 6 (length -40) and 9 (distance 0x7ff) escape; 5, 6, 9 go on to the side table
    assert b8.tolist() == [0 << 2 | 0, 62 << 2 | 3, 0xFF, 0xFF, 1 << 2 | 1, 0xFF, 0xFF, 0 << 2 | 2, 10 << 2 | 1, 0xFF]
    assert ne.value == 5 and reo.tolist() == [0, 4, 5]
    raw = d16x[:5].astype(np.int64)
    assert raw.tolist()[2:] == [0xFFFF] * 3 and (raw[:2] & 0x7ff).tolist() == [63, 1] and (((raw[:2] >> 11) ^ 16) - 16).tolist() == [2, 3]
    assert nb.value == 3 and bi[:3].tolist() == [5, 6, 9] and bp[:3].tolist() == [4000, 4001, 9010 + 0x7ff] and bl[:3].tolist() == [1, -40, 1]
    assert rio.tolist() == [0, 1 + 2 + 3 + 1, 1 + 2 + 3 + 1 + 2 + 1]


def test_degenerate_worlds():
    from nanocaller_amd.synth import World
    ref = "ACGT" * 40
    z = np.zeros(0, np.int32)
    empty = World(chrom="c", ref=ref, read_start=z, read_end=z, read_flag=z, read_off=np.zeros(1, np.int64), codes=np.zeros(0, np.uint8))
    wp = build_wire_from_world(empty, pin=False)
    assert wp.n_reads == 0 and wp.n_events == 0 and wp.codes_len == 16 and wp.n_blocks == 1
    codes, _ = _expand_numpy(wp)
    assert np.all(codes == 7)
    # many tiny reads in one block, one of them all-different from the reference, one filtered
    n = 120
    rs = (np.arange(n, dtype=np.int32) // 2) + 3
    re_ = rs + 11
    codes_in = np.tile(np.array([4, 0, 1, 2, 3, 4, 4, 0, 1, 2, 3], np.uint8), n)
    flags = np.zeros(n, np.int32)
    flags[7] = 0x400
    w = World(chrom="c", ref=ref, read_start=rs, read_end=re_, read_flag=flags, read_off=np.arange(n + 1, dtype=np.int64) * 11, codes=codes_in)
    hp = pack_world(w)
    wp = build_wire_from_world(w, pin=False)
    c2, rc = _expand_numpy(wp)
    assert np.array_equal(c2, hp.codes) and np.array_equal(rc, hp.ref_code) and wp.n_reads == n - 1


@pytest.fixture(scope="module")
def eng():
    from nanocaller_amd.engine import get_engine
    return get_engine(0)


@pytest.mark.gpu
@pytest.mark.parametrize("evb", ["2", "0"])
@pytest.mark.parametrize("name,supp,excl,tile", CASES)
def test_wire_expand_on_device_equals_direct_pack(eng, name, supp, excl, tile, evb, monkeypatch):
    import torch
    monkeypatch.setenv("NC_WIRE_EVB", evb)                               # the events one byte each (default) / two bytes each
    from nanocaller_amd.wire import WireUploader, upload_wire
    w = _consistent_deletions(load_world("indel")) if name == "indel_consistent" else load_world(name)
    a = eng.upload(pack_world(w, supplementary=supp, exclude=excl, tile_size=tile))
    wp = build_wire_from_world(w, supplementary=supp, exclude=excl, tile_size=tile)
    assert bool(((wp.meta or {}).get("indel_events") or {}).get("del_implied")) == (name == "indel_consistent")     # (then nc_wire_expand_del completes the codes)
    b = upload_wire(eng, wp)
    up = WireUploader(eng)
    t = up.submit(wp)
    c = up.expand(t)
    up.release(t)
    torch.cuda.synchronize()
    for d in (b, c):
        assert torch.equal(d.codes, a.codes) and torch.equal(d.ref_code, a.ref_code)
        assert torch.equal(d.tile_off, a.tile_off) and torch.equal(d.tile_ent[:a.tile_ent.numel()], a.tile_ent)
        assert (d.tile_size, d.tile_pos0, d.n_tiles, d.n_entries, d.pos_lo, d.pos_hi) == (a.tile_size, a.tile_pos0, a.n_tiles, a.n_entries, a.pos_lo, a.pos_hi)
        if a.events is not None:
            for k in ("ev_off", "ev_pos", "ev_len", "read_hap"):
                assert torch.equal(d.events[k][:a.events[k].numel()], a.events[k])
            assert d.events["n_reads"] == a.events["n_reads"]
    if name == "indel_consistent":
        # the separate pass (nc_wire_apply_deletions) completes an array that was expanded WITHOUT the events at hand, and changes nothing in a complete one
        import ctypes as C
        from nanocaller_amd.wire import _views
        v = _views(wp.buf.to(eng.device), wp)
        v.pop("blk_ev")
        from nanocaller_amd import wire as wire_mod
        plain = torch.empty_like(a.codes)
        wire_mod._expand(eng, wp, v, plain, torch.empty_like(a.ref_code))
        torch.cuda.synchronize()
        assert not torch.equal(plain, a.codes)
        for _ in range(2):
            rc = eng.L.nc_wire_apply_deletions(eng.ctx, wp.n_indel_reads, *(C.c_void_p(t_.data_ptr()) for t_ in (
                b.reads["rd_start"], b.reads["rd_end"], b.reads["slot_off"], b.events["ev_off"], b.events["ev_pos"], b.events["ev_len"], plain)))
            assert rc == 0
            torch.cuda.synchronize()
            assert torch.equal(plain, a.codes)
    # slot reuse: a second and third contig through the same two slots, each expanded result checked before the next
    for nm in ("hifi", "ont", "deep"):
        w2 = load_world(nm)
        t2 = up.submit(build_wire_from_world(w2))
        d2 = up.expand(t2)
        up.release(t2)
        torch.cuda.synchronize()
        assert torch.equal(d2.codes, eng.upload(pack_world(w2)).codes)
