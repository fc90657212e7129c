"""Host -> device transfer of a contig's decoded alignments in the reference-difference "wire" form (nc_wire_*, see
include/nanocaller_hip.h and csrc/nc_wire.hip).

SURVEY.md 8(d) starts the timed region at decoded alignments in pinned host memory: the reference feeds every chunk from the
host (snpCaller.py:86, generate_SNP_pileups.py:156).  `build_wire` lays ONE page-locked buffer out per contig -- read table,
reference bytes, difference events, tile index (+ the indel events) -- so that a contig crosses PCIe as a single copy of
~0.2 B per pileup entry (ONT) instead of 1 B; `WireUploader` rings those copies through three device slots on their own stream against the
compute stream and rebuilds the position-addressed codes in HBM (nc_wire_expand) right before the scan.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib
from .engine import DevicePack
from .pack import pileup_depth_cap
from .synth import FLAG_FILTER_DEFAULT, FLAG_FILTER_SUPPL, World

_ALIGN = 256


@dataclass
class WirePack:
    buf: torch.Tensor                      # uint8, page-locked when pin=True: every section back to back
    sections: dict                         # name -> (byte offset, numpy dtype, count)
    tile_size: int
    tile_pos0: int
    n_tiles: int
    n_entries: int
    codes_len: int
    n_reads: int
    n_blocks: int
    n_events: int
    ref_len: int
    pos_lo: int
    pos_hi: int
    n_indel_reads: int = -1                # >= 0: indel event sections present
    meta: dict = field(default_factory=dict)

    @property
    def nbytes(self):
        return int(self.buf.numel())

    def host(self, name):
        off, dt, cnt = self.sections[name]
        return self.buf.numpy()[off:off + cnt * np.dtype(dt).itemsize].view(dt)


def _ref_wire_lut():
    lut = np.full(256, 4 | 8, np.uint8)
    for i, b in enumerate("AGTC"):
        lut[ord(b)] = i                                            # upper case: base code, scanned
        lut[ord(b.lower())] = i | 8                                # lower case: base code, skipped
    return lut


_REF_WIRE_LUT = _ref_wire_lut()


def ref_wire_from_string(ref: str, exclude=None, pos0=1):
    """uint8 per position (index p - pos0): bits 0-2 base code of the letter in either case (A0 G1 T2 C3, else 4), bit 3 =
    the column is skipped by the scan: not an UPPER-case AGTC (`s in 'AGTC'` before .upper(), generate_SNP_pileups.py:137,
    quirk E4) or inside an exclude interval (tree.overlaps(pos): a <= pos < b, :116-119,161)"""
    raw = np.frombuffer(ref.encode("ascii") if isinstance(ref, str) else ref, np.uint8)
    out = _REF_WIRE_LUT[raw]                                       # one table pass (base | skip)
    for (a, b) in exclude or ():
        lo, hi = max(0, int(a) - pos0), max(0, int(b) - pos0)
        out[lo:hi] |= 8
    return out


def build_wire(read_start, read_end, read_off, codes, read_flag, ref_wire_pos1, *, supplementary=False, tile_size=2048,
               pos_lo=None, pos_hi=None, hap=None, events=None, strand=None, keep=None, pin=True, indel_extra=None, names=None, name_gid=None) -> WirePack:
    """read_* / codes as synth.World (coordinate order, codes[read_off[r] + p - read_start[r]]); `ref_wire_pos1`: uint8 per
    position, index p - 1 (ref_wire_from_string).  Flag filter and strand as pack.pack_reads, or given directly (`keep`,
    `strand`).  `indel_extra` (with `events`): dict of the per-read arrays of the device pass 2 for the KEPT reads (ins_off,
    ins_bases, tail_off, tail_bases, read_ps, read_flag: nc_indel_pack_build) -- they ride in the same buffer.
    -> WirePack: one host buffer ready for a single H2D copy."""
    L = _lib.lib()
    rs = np.ascontiguousarray(read_start, np.int32)
    re_ = np.ascontiguousarray(read_end, np.int32)
    ro = np.ascontiguousarray(read_off, np.int64)
    cd = np.ascontiguousarray(codes, np.uint8)
    n = int(rs.shape[0])
    if keep is None:
        flag = np.asarray(read_flag)
        keep = pileup_depth_cap(read_start, read_end, np.ascontiguousarray((flag & (FLAG_FILTER_SUPPL if supplementary else FLAG_FILTER_DEFAULT)) == 0, np.uint8))
        strand = np.ascontiguousarray((flag & 0x10) != 0, np.uint8)
    else:
        keep = np.ascontiguousarray(keep, np.uint8)
        strand = np.ascontiguousarray(strand, np.uint8)
    mates = None
    if (names is not None or name_gid is not None) and read_flag is not None:   # alignments that share read names (pack.name_groups): the name's strand, bit 3
        from .pack import mate_table, name_groups
        nxt, gstrand = name_groups(names, read_flag, keep, name_gid)
        if nxt is not None:
            strand = np.ascontiguousarray(gstrand | ((nxt >= 0).astype(np.uint8) << 3))
            mates = mate_table(nxt, keep, rs, re_)
    if hap is not None:
        strand = np.ascontiguousarray(strand | (np.asarray(hap, np.uint8) & 3) << 1)          # bits 1-2: HP tag
    Lref = int(ref_wire_pos1.shape[0])
    pos_lo = 1 if pos_lo is None else max(1, int(pos_lo))
    pos_hi = max(pos_lo, Lref if pos_hi is None else min(Lref, int(pos_hi)))
    codes_len, n_ent = C.c_int64(), C.c_int64()
    tile_pos0, n_tiles = C.c_int32(), C.c_int32()
    rc = L.nc_pack_plan(n, _lib.npp(rs), _lib.npp(re_), _lib.npp(keep), tile_size, pos_lo, pos_hi, C.byref(codes_len), C.byref(tile_pos0),
                        C.byref(n_tiles), C.byref(n_ent))
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("nc_pack_plan failed (%d)" % rc)
    # reference bytes on the tile grid (what nc_wire_expand and the scan address)
    ref_len = n_tiles.value * tile_size
    ref_grid = np.full(ref_len, 4 | 8, np.uint8)
    a, b = max(1, tile_pos0.value), min(Lref, tile_pos0.value + ref_len - 1)
    if b >= a:
        ref_grid[a - tile_pos0.value:b - tile_pos0.value + 1] = ref_wire_pos1[a - 1:b]
    tile_off = np.empty(n_tiles.value + 1, np.int32)
    tile_ent = np.empty(max(1, n_ent.value), _lib.TILE_ENTRY_DTYPE)
    rc = L.nc_pack_fill(n, _lib.npp(rs), _lib.npp(re_), None, None, _lib.npp(strand), _lib.npp(keep), tile_size, tile_pos0.value,
                        n_tiles.value, None, codes_len.value, _lib.npp(tile_off), _lib.npp(tile_ent), n_ent.value)      # index only
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("nc_pack_fill (index) failed (%d)" % rc)
    h = C.c_void_p()
    del_implied = events is not None and os.environ.get("NC_WIRE_DEL_IMPLIED", "1") != "0"
    # the difference events one byte each (nc_wire_build2; the builder keeps the two-byte form where that is shorter: sparse events, HiFi):
    # NC_WIRE_EVB = 2 (default) every pack, 1 only packs that travel with their indel events, 0 never
    evb = os.environ.get("NC_WIRE_EVB", "2")
    byte_events = evb == "2" or (evb == "1" and events is not None)
    if byte_events:
        e3 = [None, None, None]
        if del_implied:
            e_off, e_pos, e_len = (np.ascontiguousarray(x, np.int32) for x in events)
            e3 = [_lib.npp(e_off), _lib.npp(e_pos) if e_pos.size else None, _lib.npp(e_len) if e_len.size else None]
        rc = L.nc_wire_build2(n, _lib.npp(rs), _lib.npp(re_), _lib.npp(ro), _lib.npp(cd), _lib.npp(keep), _lib.npp(ref_grid), tile_pos0.value, ref_len, *e3, 1, C.byref(h))
        if rc == _lib.NC_ERR_UNSUPPORTED and del_implied:              # codes and events disagree about a deleted column: without the implied deletions
            del_implied = False
            rc = L.nc_wire_build2(n, _lib.npp(rs), _lib.npp(re_), _lib.npp(ro), _lib.npp(cd), _lib.npp(keep), _lib.npp(ref_grid), tile_pos0.value, ref_len, None, None, None, 1, C.byref(h))
    elif del_implied:
        # the reads travel with their indel events: in a block that lies inside one read a deleted column's code is implied by the deletion event
        # and left out of the difference events (nc_wire_expand_del writes it from the events, expanded first)
        e_off, e_pos, e_len = (np.ascontiguousarray(x, np.int32) for x in events)
        rc = L.nc_wire_build_del(n, _lib.npp(rs), _lib.npp(re_), _lib.npp(ro), _lib.npp(cd), _lib.npp(keep), _lib.npp(ref_grid), tile_pos0.value,
                                 ref_len, _lib.npp(e_off), _lib.npp(e_pos) if e_pos.size else None, _lib.npp(e_len) if e_len.size else None, C.byref(h))
        if rc == _lib.NC_ERR_UNSUPPORTED:                               # codes and events disagree about a deleted column: the plain form
            del_implied = False
    if not del_implied and not byte_events:
        rc = L.nc_wire_build(n, _lib.npp(rs), _lib.npp(re_), _lib.npp(ro), _lib.npp(cd), _lib.npp(keep), _lib.npp(ref_grid), tile_pos0.value,
                             ref_len, C.byref(h))
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("nc_wire_build failed (%d)" % rc)
    try:
        v = _lib.WireArraysC()
        L.nc_wire_view(h, C.byref(v))
        assert v.codes_len == codes_len.value, (v.codes_len, codes_len.value)

        def arr(ptr, cnt, dt):
            if not cnt:
                return np.zeros(0, dt)
            return np.frombuffer((C.c_char * (int(cnt) * np.dtype(dt).itemsize)).from_address(ptr), dt)
        meta_ev = None
        parts = [("rd_start", arr(v.rd_start, v.n_reads, np.int32)), ("rd_end", arr(v.rd_end, v.n_reads, np.int32)),
                 ("slot_off", arr(v.slot_off, v.n_reads + 1, np.int64)), ("blk_off", arr(v.blk_off, v.n_blocks + 1, np.uint32)),
                 ("blk_read", arr(v.blk_read, v.n_blocks, np.int32)),
                 (("ev_bytes", np.frombuffer((C.c_char * int(v.n_ev_bytes + 8)).from_address(v.ev_bytes), np.uint8)) if v.ev_bytes
                  else ("events", arr(v.events, v.n_events, np.uint16))), ("ref_nib", ref_grid[0::2] | (ref_grid[1::2] << 4)), ("tile_off", tile_off),
                 ("tile_ent", np.frombuffer(tile_ent[:n_ent.value].tobytes(), np.uint8) if n_ent.value else np.zeros(16, np.uint8))]
        if mates is not None:
            mates = mate_table(nxt, keep, rs, re_, slot_off=arr(v.slot_off, v.n_reads + 1, np.int64))     # (checked against the builder's own slots)
            parts += [("mate_key", mates[0]), ("mate_rec", mates[1].reshape(-1))]
        if del_implied:
            parts.append(("blk_ev", arr(v.blk_ev, v.n_blocks, np.uint32)))     # per block: where the read's deletion events start (nc_wire_expand_del)
        n_indel = -1
        if events is not None:
            ev_off, ev_pos, ev_len = (np.asarray(x) for x in events)
            kept = np.nonzero(keep)[0]
            cnt = (ev_off[1:] - ev_off[:-1])[kept]
            off = np.zeros(kept.size + 1, np.int32)
            np.cumsum(cnt, out=off[1:])
            if kept.size == n:                                        # nothing filtered: the events are already in pack order
                idx = slice(0, int(ev_off[-1]))
            else:
                # (the kept reads' event ranges, one after the other: start of each range - where it lands, repeated, + a running index)
                idx = (np.repeat(ev_off[:-1][kept].astype(np.int64) - off[:-1], cnt) + np.arange(int(off[-1]), dtype=np.int64)) if kept.size else np.zeros(0, np.int64)
            hp = (np.asarray(hap, np.uint8) if hap is not None else np.zeros(n, np.uint8))[kept]
            z = lambda x, dt: np.ascontiguousarray(x, dt) if len(x) else np.zeros(1, dt)      # noqa: E731
            # the events cross PCIe as 3 bytes each (distance to the read's previous event, signed length; a side table for the few that do
            # not fit) + one inserted-base offset per READ: nc_indel_events_expand rebuilds ev_pos / ev_len / ins_off in HBM
            evp, evl = np.ascontiguousarray(ev_pos[idx], np.int32), np.ascontiguousarray(ev_len[idx], np.int32)
            n_ev = int(evp.size)
            kept_start = np.ascontiguousarray(rs[kept], np.int32)
            rio = np.zeros(kept.size + 1, np.int32)
            ev8 = os.environ.get("NC_WIRE_EV8", "1") != "0"
            cap = max(1024, n_ev // 64)
            ecap = max(1024, n_ev // 4)
            b8 = np.empty(max(n_ev, 1), np.uint8) if ev8 else None     # one byte per event, the others through the two-byte array (nc_indel_events_pack8)
            reo = np.zeros(kept.size + 1, np.int32)
            while True:
                bi, bp, bl = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.int32)
                nbig, nesc = C.c_int64(), C.c_int64()
                if ev8:
                    d16 = np.empty(ecap, np.uint16)
                    rc = L.nc_indel_events_pack8(int(kept.size), _lib.npp(kept_start), _lib.npp(off), _lib.npp(evp), _lib.npp(evl), _lib.npp(b8), _lib.npp(d16), ecap,
                                                 _lib.npp(reo), _lib.npp(rio), cap, _lib.npp(bi), _lib.npp(bp), _lib.npp(bl), C.byref(nesc), C.byref(nbig))
                else:
                    d16 = np.empty(max(n_ev, 1), np.uint16)           # two bytes per event: distance | signed 5-bit length << 11 (nc_indel_events_pack with l8 = NULL)
                    rc = L.nc_indel_events_pack(int(kept.size), _lib.npp(kept_start), _lib.npp(off), _lib.npp(evp), _lib.npp(evl), _lib.npp(d16), None,
                                                _lib.npp(rio), cap, _lib.npp(bi), _lib.npp(bp), _lib.npp(bl), C.byref(nbig))
                if rc == _lib.NC_ERR_CAPACITY:
                    cap, ecap = max(cap, int(nbig.value)), max(ecap, int(nesc.value))
                    continue
                if rc != _lib.NC_OK:
                    raise _lib.NanoCallerHipError("nc_indel_events_pack failed (%d)" % rc)
                break
            nb_ = int(nbig.value)
            parts += [("ev_off", z(off, np.int32)), ("ev_d16", z(d16[:int(nesc.value)], np.uint16) if ev8 else d16), ("ev_big_idx", z(bi[:nb_], np.int32)),
                      ("ev_big_pos", z(bp[:nb_], np.int32)), ("ev_big_len", z(bl[:nb_], np.int32)), ("read_ins_off", rio), ("read_hap", z(hp, np.uint8))]
            if ev8:
                parts += [("ev_b8", b8), ("read_esc_off", reo)]
            meta_ev = dict(n_ev=n_ev, n_big=nb_, del_implied=bool(del_implied), ev8=bool(ev8))
            n_indel = int(kept.size)
            if indel_extra is not None:
                if n_ev and int(np.asarray(indel_extra["ins_off"])[n_ev]) != int(rio[-1]):
                    raise _lib.NanoCallerHipError("indel_extra['ins_off'] is not the running sum of the insertion lengths")
                ib = np.ascontiguousarray(indel_extra["ins_bases"], np.uint8)
                if os.environ.get("NC_WIRE_INS_2BIT", "1") != "0" and ib.size:
                    # the inserted bases two bits each (nc_wire_ins_unpack rebuilds the bytes in HBM); the few other letters (code 4) as indices
                    other = np.flatnonzero(ib > 3).astype(np.int32)
                    q = np.zeros((ib.size + 15) // 16 * 16, np.uint8)
                    q[:ib.size] = ib & 3
                    q = q.reshape(-1, 4)
                    parts += [("ins_2b", np.ascontiguousarray(q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6))), ("ins_other", z(other, np.int32))]
                    meta_ev["ins_2b"] = dict(n=int(ib.size), n_other=int(other.size))
                else:
                    parts.append(("ins_bases", z(ib, np.uint8)))
                for name, dt in (("tail_off", np.int32), ("tail_bases", np.uint8), ("read_ps", np.int32), ("read_flag", np.uint8)):
                    parts.append((name, z(indel_extra[name], dt)))
                meta_ev["extra"] = True
        sections, total = {}, 0
        for name, a_ in parts:
            sections[name] = (total, a_.dtype, int(a_.size))
            total += (a_.nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        buf = torch.empty(max(total, _ALIGN), dtype=torch.uint8, pin_memory=bool(pin and torch.cuda.is_available()))
        hb = buf.numpy()
        for name, a_ in parts:
            o = sections[name][0]
            hb[o:o + a_.nbytes] = a_.view(np.uint8).reshape(-1)
        return WirePack(buf=buf, sections=sections, tile_size=tile_size, tile_pos0=tile_pos0.value, n_tiles=n_tiles.value,
                        n_entries=int(n_ent.value), codes_len=int(codes_len.value), n_reads=int(v.n_reads), n_blocks=int(v.n_blocks),
                        n_events=int(v.n_events), ref_len=ref_len, pos_lo=pos_lo, pos_hi=pos_hi, n_indel_reads=n_indel,
                        meta=dict(indel_events=meta_ev) if n_indel >= 0 else {})
    finally:
        L.nc_wire_free(h)


def build_wire_from_world(world: World, supplementary=False, exclude=None, **kw) -> WirePack:
    if "events" in world.meta:
        kw.setdefault("hap", world.meta["hap"])
        kw.setdefault("events", world.meta["events"])
    from .pack import world_name_gid, world_names
    kw.setdefault("names", world_names(world))
    kw.setdefault("name_gid", world_name_gid(world, supplementary))
    return build_wire(world.read_start, world.read_end, world.read_off, world.codes, world.read_flag,
                      ref_wire_from_string(world.ref, exclude), supplementary=supplementary, **kw)


def _views(dev_buf, wp: WirePack):
    out = {}
    for name, (off, dt, cnt) in wp.sections.items():
        nb = cnt * np.dtype(dt).itemsize
        t = dev_buf[off:off + nb]
        out[name] = t if np.dtype(dt) == np.uint8 else t.view({np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.uint32): torch.int32,
                                                                np.dtype(np.uint16): torch.int16, np.dtype(np.int8): torch.int8}[np.dtype(dt)])
    return out


def _expand(eng, wp: WirePack, v, codes, ref_code, scratch=None):
    if "ref_wire" not in v:                                             # the reference bytes travel two per byte: the byte array nc_wire_expand reads
        if scratch is None or scratch.numel() < wp.ref_len:
            scratch = torch.empty(wp.ref_len, dtype=torch.uint8, device=codes.device)
        eng._check(eng.L.nc_wire_ref_unpack(eng.ctx, C.c_void_p(v["ref_nib"].data_ptr()), wp.ref_len, C.c_void_p(scratch.data_ptr())), "nc_wire_ref_unpack")
        v["ref_wire"] = scratch
    evs = v["ev_bytes"] if "ev_bytes" in v else (v["events"] if wp.n_events else v["blk_off"])
    args = (eng.ctx, wp.n_reads, C.c_void_p(v["rd_start"].data_ptr()), C.c_void_p(v["rd_end"].data_ptr()),
            C.c_void_p(v["slot_off"].data_ptr()), C.c_void_p(v["ref_wire"].data_ptr()), wp.tile_pos0, wp.ref_len,
            C.c_void_p(v["blk_off"].data_ptr()), C.c_void_p(v["blk_read"].data_ptr()), C.c_void_p(evs.data_ptr()),
            wp.n_blocks, C.c_void_p(codes.data_ptr()), wp.codes_len, C.c_void_p(ref_code.data_ptr()))
    if "ev_bytes" in v:                                                 # the events one byte each (nc_wire_build2)
        if "blk_ev" in v:
            assert wp.n_indel_reads == wp.n_reads and "ev_pos" in v
            rc = eng.L.nc_wire_expand2(*args, C.c_void_p(v["blk_ev"].data_ptr()), C.c_void_p(v["ev_off"].data_ptr()),
                                       C.c_void_p(v["ev_pos"].data_ptr()), C.c_void_p(v["ev_len"].data_ptr()))
        else:
            rc = eng.L.nc_wire_expand2(*args, None, None, None, None)
    elif "blk_ev" in v:                                                   # deleted columns implied: from the reads' events (_expand_events ran first)
        assert wp.n_indel_reads == wp.n_reads and "ev_pos" in v
        rc = eng.L.nc_wire_expand_del(*args, C.c_void_p(v["blk_ev"].data_ptr()), C.c_void_p(v["ev_off"].data_ptr()),
                                      C.c_void_p(v["ev_pos"].data_ptr()), C.c_void_p(v["ev_len"].data_ptr()))
    else:
        rc = eng.L.nc_wire_expand(*args)
    eng._check(rc, "nc_wire_expand")


def _expand_events(eng, wp: WirePack, v, out):
    """the 3-byte transfer form of the indel events -> ev_pos / ev_len / ins_off int32 in `out` (dict of device tensors, grown on demand),
    put into the views dict `v` under the names the kernels' structs take"""
    me = wp.meta.get("indel_events") if wp.meta else None
    if not me:
        return
    n_ev, dev = me["n_ev"], eng.device
    for name, n in (("ev_pos", n_ev), ("ev_len", n_ev), ("ins_off", n_ev + 1)):
        if out.get(name) is None or out[name].numel() < max(n, 4):
            out[name] = torch.zeros(max(n, 4) + max(n, 4) // 16, dtype=torch.int32, device=dev)
    if n_ev:
        tail = (me["n_big"], C.c_void_p(v["ev_big_idx"].data_ptr()), C.c_void_p(v["ev_big_pos"].data_ptr()), C.c_void_p(v["ev_big_len"].data_ptr()),
                C.c_void_p(v["read_ins_off"].data_ptr()), C.c_void_p(out["ev_pos"].data_ptr()), C.c_void_p(out["ev_len"].data_ptr()),
                C.c_void_p(out["ins_off"].data_ptr()))
        if me.get("ev8"):
            rc = eng.L.nc_indel_events_expand8(eng.ctx, wp.n_indel_reads, C.c_void_p(v["rd_start"].data_ptr()), C.c_void_p(v["ev_off"].data_ptr()),
                                               C.c_void_p(v["ev_b8"].data_ptr()), C.c_void_p(v["ev_d16"].data_ptr()), C.c_void_p(v["read_esc_off"].data_ptr()), *tail)
        else:
            rc = eng.L.nc_indel_events_expand(eng.ctx, wp.n_indel_reads, C.c_void_p(v["rd_start"].data_ptr()), C.c_void_p(v["ev_off"].data_ptr()),
                                              C.c_void_p(v["ev_d16"].data_ptr()), None, *tail)
        eng._check(rc, "nc_indel_events_expand")
    v["ev_pos"], v["ev_len"] = out["ev_pos"][:max(n_ev, 1)], out["ev_len"][:max(n_ev, 1)]
    if me.get("extra"):
        v["ins_off"] = out["ins_off"][:n_ev + 1]
    if me.get("ins_2b"):
        n, no = me["ins_2b"]["n"], me["ins_2b"]["n_other"]
        need = (n + 15) // 16 * 16 + 16
        if out.get("ins_bases") is None or out["ins_bases"].numel() < need:
            out["ins_bases"] = torch.zeros(need + need // 16, dtype=torch.uint8, device=dev)
        eng._check(eng.L.nc_wire_ins_unpack(eng.ctx, C.c_void_p(v["ins_2b"].data_ptr()), n, C.c_void_p(v["ins_other"].data_ptr()), no,
                                            C.c_void_p(out["ins_bases"].data_ptr())), "nc_wire_ins_unpack")
        v["ins_bases"] = out["ins_bases"][:max(n, 1)]


def _device_pack(wp: WirePack, v, codes, ref_code, own_index):
    g = (lambda t: t.clone()) if own_index else (lambda t: t)
    dp = DevicePack(codes=codes, tile_off=g(v["tile_off"]), tile_ent=g(v["tile_ent"]), ref_code=ref_code, tile_size=wp.tile_size,
                    tile_pos0=wp.tile_pos0, n_tiles=wp.n_tiles, n_entries=wp.n_entries, pos_lo=wp.pos_lo, pos_hi=wp.pos_hi)
    if "mate_key" in v:
        dp.mates = (g(v["mate_key"]), g(v["mate_rec"]))
    if wp.n_indel_reads >= 0:
        dp.events = dict(n_reads=wp.n_indel_reads, ev_off=g(v["ev_off"]), ev_pos=g(v["ev_pos"]), ev_len=g(v["ev_len"]), read_hap=g(v["read_hap"]))
        dp.reads = dict(n_reads=wp.n_reads, rd_start=g(v["rd_start"]), rd_end=g(v["rd_end"]), slot_off=g(v["slot_off"]))
        if "ins_off" in v:
            dp.indel = {k: g(v[k]) for k in ("ins_off", "ins_bases", "tail_off", "tail_bases", "read_ps", "read_flag")}
    return dp


def indel_reads_struct(dp: DevicePack):
    """nc_indel_reads over the tensors of a pack that carries the indel sections (build_wire(..., indel_extra=...))"""
    ev, rd, ix = dp.events, dp.reads, dp.indel
    return _lib.IndelReadsC(n_reads=rd["n_reads"], slot_off=rd["slot_off"].data_ptr(), rd_start=rd["rd_start"].data_ptr(), rd_end=rd["rd_end"].data_ptr(),
                            ev_off=ev["ev_off"].data_ptr(), ev_pos=ev["ev_pos"].data_ptr(), ev_len=ev["ev_len"].data_ptr(),
                            ins_off=ix["ins_off"].data_ptr(), ins_bases=ix["ins_bases"].data_ptr(), tail_off=ix["tail_off"].data_ptr(),
                            tail_bases=ix["tail_bases"].data_ptr(), read_ps=ix["read_ps"].data_ptr(), read_hap=ev["read_hap"].data_ptr(),
                            read_flag=ix["read_flag"].data_ptr())


def upload_wire(eng, wp: WirePack) -> DevicePack:
    """One contig, synchronously: copy the wire buffer, expand it into its own codes / ref_code tensors; the tile index and
    the indel events are cloned out so that the 0.2 B/entry wire copy can be freed"""
    eng.use_torch_stream()
    dev = eng.device
    d = wp.buf.to(dev, non_blocking=True)
    v = _views(d, wp)
    codes = torch.empty(wp.codes_len, dtype=torch.uint8, device=dev)
    ref_code = torch.empty(wp.ref_len, dtype=torch.uint8, device=dev)
    _expand_events(eng, wp, v, {})
    _expand(eng, wp, v, codes, ref_code)
    return _device_pack(wp, v, codes, ref_code, own_index=True)


class WireUploader:
    """Uploads through a ring of slots: `submit(wp)` enqueues the single H2D copy of a contig on the upload stream (it waits, on
    the device, until the slot's previous user has been released; with three slots the copy of contig i+1 never waits for contig
    i-1's kernels, measured +1.8 % on the headline against two); `expand(ticket)` makes the compute stream wait for that copy and
    rebuilds the codes in HBM; `release(ticket)` marks -- in compute-stream order -- that the step which used the pack has been
    enqueued completely, so the slot may be overwritten.  The expanded codes / ref_code live in ONE buffer pair that every
    step reuses: all steps run on the same compute stream, so step i+1's expansion is ordered behind step i's last reader."""

    def __init__(self, eng, slots=3):
        self.eng = eng
        self.stream = torch.cuda.Stream(device=eng.device)
        self.slots = [dict(buf=None, free=None) for _ in range(slots)]
        self.turn = 0
        self.codes = None
        self.ref_code = None
        self.ref_bytes = None                      # the reference bytes unpacked from the wire's nibbles (scratch of nc_wire_expand, like codes: one, reused)
        self.ev = {}                               # expanded indel events (ev_pos / ev_len / ins_off), like codes: one set, reused by every step
        self.h2d_events = []                       # (start, stop) event pairs of the copies, for the achieved PCIe rate
        self.timing = False

    def submit(self, wp: WirePack):
        s = self.slots[self.turn % len(self.slots)]
        self.turn += 1
        if s["buf"] is None or s["buf"].numel() < wp.nbytes:
            if s["free"] is not None:
                s["free"].synchronize()
            s["buf"] = torch.empty(wp.nbytes + wp.nbytes // 8, dtype=torch.uint8, device=self.eng.device)
        with torch.cuda.stream(self.stream):
            if s["free"] is not None:
                self.stream.wait_event(s["free"])
            if self.timing:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
            s["buf"][:wp.nbytes].copy_(wp.buf, non_blocking=True)
            done = torch.cuda.Event(enable_timing=self.timing)
            done.record(self.stream)
            if self.timing:
                self.h2d_events.append((e0, done, wp.nbytes))
        return dict(wp=wp, slot=s, done=done)

    def expand(self, ticket) -> DevicePack:
        eng, wp = self.eng, ticket["wp"]
        eng.use_torch_stream()
        cur = torch.cuda.current_stream(eng.device)
        cur.wait_event(ticket["done"])
        if self.codes is None or self.codes.numel() < wp.codes_len:
            self.codes = torch.empty(wp.codes_len + wp.codes_len // 16, dtype=torch.uint8, device=eng.device)
        if self.ref_code is None or self.ref_code.numel() < wp.ref_len:
            self.ref_code = torch.empty(wp.ref_len + wp.ref_len // 16, dtype=torch.uint8, device=eng.device)
        v = _views(ticket["slot"]["buf"], wp)
        codes, ref_code = self.codes[:wp.codes_len], self.ref_code[:wp.ref_len]
        if self.ref_bytes is None or self.ref_bytes.numel() < wp.ref_len:
            self.ref_bytes = torch.empty(wp.ref_len + wp.ref_len // 16, dtype=torch.uint8, device=eng.device)
        _expand_events(eng, wp, v, self.ev)
        _expand(eng, wp, v, codes, ref_code, self.ref_bytes)
        return _device_pack(wp, v, codes, ref_code, own_index=False)

    def release(self, ticket):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.eng.device))
        ticket["slot"]["free"] = ev

    def h2d_rate(self):
        """-> (GB/s over the timed copies, total ms, bytes); call after a synchronize"""
        ms = sum(a.elapsed_time(b) for a, b, _ in self.h2d_events)
        nb = sum(n for _, _, n in self.h2d_events)
        return (nb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0), ms, nb
