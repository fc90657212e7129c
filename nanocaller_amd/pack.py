"""Host side of the packed-alignment layout (see include/nanocaller_hip.h, "read pack").

`pack_reads` turns decoded alignments (read-major codes, the boundary that replaces the pysam objects of
generate_SNP_pileups.py:134-164) into the position-aligned slots + tile index the kernels read.  The heavy
lifting is the library's native packer (nc_pack_plan / nc_pack_fill); this module only applies the pileup
flag filter and builds the padded reference-code array.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .synth import FLAG_FILTER_DEFAULT, FLAG_FILTER_SUPPL, World, world_ref_codes


@dataclass
class HostPack:
    codes: np.ndarray        # uint8 [codes_len]
    tile_size: int
    tile_pos0: int
    n_tiles: int
    tile_off: np.ndarray     # int32 [n_tiles+1]
    tile_ent: np.ndarray     # TILE_ENTRY_DTYPE [n_entries]
    ref_code: np.ndarray     # uint8 [n_tiles*tile_size]; index 0 <-> tile_pos0; 4 = skip column
    pos_lo: int
    pos_hi: int
    # optional indel inputs for the kept reads, in pack order (generate_indel_pileups.py:178-235)
    read_hap: np.ndarray | None = None     # uint8 [n_kept]  0 untagged / 1 / 2
    ev_off: np.ndarray | None = None       # int32 [n_kept+1]
    ev_pos: np.ndarray | None = None       # int32
    ev_len: np.ndarray | None = None       # int32, + insertion / - deletion
    mates: tuple | None = None             # (key int64 [M], rec int32 [M, 4]): alignments that share read names (mate_table)

    @property
    def nbytes(self):
        return self.codes.nbytes + self.tile_off.nbytes + self.tile_ent.nbytes + self.ref_code.nbytes


def _check(rc, what):
    if rc != _lib.NC_OK:
        raise _lib.NanoCallerHipError("%s failed with status %d" % (what, rc))


# pysam's AlignmentFile.pileup(..., max_depth=8000) default, which the reference does not override (generate_SNP_pileups.py:156,
# generate_indel_pileups.py:213): htslib's pileup buffer refuses a read that starts at the column it is about to emit while
# it already holds max_depth reads.
PILEUP_MAX_DEPTH = 8000


def pileup_depth_cap(read_start, read_end, keep, max_depth=PILEUP_MAX_DEPTH):
    """`keep` (uint8 per read, file = coordinate order) with the reads htslib's pileup engine drops cleared.  bam_plp_push:
    a record whose start equals the iterator's current column is skipped when the buffer holds `max_depth` reads; the buffer
    then holds every accepted read that ends after the previous column (reads are released as columns are emitted), and the
    FIRST record of a start position is pushed while the iterator is still before it, so it always enters.  [pysam-doc:
    unpinned, the library is absent from the image.]  Data below max_depth everywhere (any ordinary WGS run) is untouched."""
    keep = np.ascontiguousarray(keep, np.uint8)
    idx = np.flatnonzero(keep)
    if idx.size <= max_depth:
        return keep
    s = np.asarray(read_start, np.int64)[idx]
    e = np.asarray(read_end, np.int64)[idx]
    # reads ahead of read i that are still buffered when it arrives, if nothing had been dropped: an upper bound
    alive = np.arange(idx.size) - np.searchsorted(np.sort(e), s - 1, side="right")
    if int(alive.max()) < max_depth:
        return keep
    import heapq
    keep = keep.copy()
    heap, last = [], None
    for k in range(idx.size):
        sk = int(s[k])
        while heap and heap[0] <= sk - 1:
            heapq.heappop(heap)
        if sk == last and len(heap) >= max_depth:
            keep[idx[k]] = 0
        else:
            heapq.heappush(heap, int(e[k]))
        last = sk
    return keep


def name_groups(names, read_flag, keep, gid=None):
    """Alignments that share a read name among the kept ones -- a split read's primary + supplementary records under dct['supplementary']
    (paired-end mates without it).  The reference keys a column's pileup, the strand table and the neighbour lookups by NAME
    (generate_SNP_pileups.py:141-143,175,185,223,232).  `gid` (int32 [n], optional): the name's first alignment or -1 per alignment, from the
    native decode (nc_decoded_name_groups: no Python walk over 10^6 names); else the names themselves are grouped here.
    -> (nxt int32 [n]: the next kept alignment of the same name, circular, -1 = the name is this alignment's alone; strand uint8 [n]: for the
    members of a shared name the 0x10 bit of the name's LAST primary record in file order (strand_dict[qname] is assigned per primary record; a
    name without one -- the reference raises KeyError -- keeps its records' own bits)); (None, None) when no name is shared."""
    keep = np.asarray(keep) != 0
    flag = np.asarray(read_flag)
    if gid is None:
        if names is None or len(names) != len(keep):
            return None, None
        first, gid = {}, np.full(len(keep), -1, np.int32)
        for r in np.flatnonzero(keep).tolist():
            q = first.setdefault(names[r], r)
            if q != r:
                gid[q] = gid[r] = q
    mem = np.flatnonzero(keep & (np.asarray(gid) >= 0))            # (a member the depth cap dropped leaves its name's ring)
    if mem.size < 2:
        return None, None
    g = np.asarray(gid)[mem]
    order = np.argsort(g, kind="stable")                             # by name, file order inside a name
    mem, g = mem[order], g[order]
    first_of = np.r_[True, g[1:] != g[:-1]]
    last_of = np.r_[first_of[1:], True]
    start_idx = np.maximum.accumulate(np.where(first_of, np.arange(mem.size), 0))
    nxt_in = np.where(last_of, mem[start_idx], np.r_[mem[1:], mem[:1]])
    alone = first_of & last_of
    nxt = np.full(len(keep), -1, np.int32)
    nxt[mem[~alone]] = nxt_in[~alone]
    if not (nxt >= 0).any():
        return None, None
    strand = ((flag & 0x10) != 0).astype(np.uint8)
    # the name's last primary record: walk the members in file order, remembering the latest primary's bit per name
    prim = (flag[mem] & 0x900) == 0
    lastp = np.where(prim, np.arange(mem.size), -1)
    run_end = np.flatnonzero(last_of)
    run_start = np.flatnonzero(first_of)
    for a, b in zip(run_start.tolist(), run_end.tolist()):           # (one step per shared NAME: few)
        if b == a:
            continue
        lp = lastp[a:b + 1].max()
        if lp >= 0:
            strand[mem[a:b + 1]] = strand[mem[lp]]
    return nxt, strand


def mate_table(nxt, keep, read_start, read_end, slot_off=None):
    """the featuriser's table of the alignments that share names (nc_snp_set_mates), from name_groups' `nxt`: -> (key int64 [M] = byte offset of
    the alignment's slot in the pack's codes, ascending; rec int32 [M, 4] = start, end, table index of the next alignment of the name, 0)"""
    kept = np.flatnonzero(keep)
    rs, re_ = np.asarray(read_start, np.int64)[kept], np.asarray(read_end, np.int64)[kept]
    slot = np.zeros(kept.size + 1, np.int64)
    np.cumsum(((re_ + 15) & ~15) - (rs & ~15), out=slot[1:])        # (nc_pack_fill / nc_wire_build: slots in kept order, 16-byte aligned ends)
    if slot_off is not None and not np.array_equal(slot, np.asarray(slot_off, np.int64)):
        raise _lib.NanoCallerHipError("mate_table: the slot layout differs from the packer's")
    rank = np.full(len(keep), -1, np.int64)
    mem = np.flatnonzero(np.asarray(nxt) >= 0)
    rank[mem] = np.arange(mem.size)
    pos_in_kept = np.searchsorted(kept, mem)
    key = np.ascontiguousarray(slot[pos_in_kept], np.int64)
    rec = np.zeros((mem.size, 4), np.int32)
    rec[:, 0], rec[:, 1], rec[:, 2] = rs[pos_in_kept], re_[pos_in_kept], rank[np.asarray(nxt)[mem]]
    return key, rec


def pack_reads(read_start, read_end, read_off, codes, read_flag, ref_codes, *, supplementary=False,
               tile_size=2048, pos_lo=None, pos_hi=None, exclude=None, hap=None, events=None, names=None, name_gid=None) -> HostPack:
    """read_* as in synth.World (coordinate order); ref_codes uint8 [L] (index p-1, 4 = skip).
    `exclude`: iterable of (start, end) half-open intervals whose columns are skipped, the IntervalTree
    test `tree.overlaps(pos)` of generate_SNP_pileups.py:116-119,161."""
    L = _lib.lib()
    rs = np.ascontiguousarray(read_start, np.int32)
    re_ = np.ascontiguousarray(read_end, np.int32)
    ro = np.ascontiguousarray(read_off, np.int64)
    cd = np.ascontiguousarray(codes, np.uint8)
    flag = np.asarray(read_flag)
    filt = FLAG_FILTER_SUPPL if supplementary else FLAG_FILTER_DEFAULT      # :151-154
    keep = pileup_depth_cap(read_start, read_end, np.ascontiguousarray((flag & filt) == 0, np.uint8))
    # strand: the reference looks the read NAME up in a table of the primary alignments' `(flag & 0x910) // 16` (:141-143),
    # i.e. bit 0x10 of a primary record.  With dct['supplementary'] a supplementary record is counted on its own 0x10 bit
    # (the reference takes its primary's, and raises KeyError when that is outside the fetch window).
    strand = np.ascontiguousarray((flag & 0x10) != 0, np.uint8)
    nxt, gstrand = name_groups(names, flag, keep, name_gid)                  # shared read names: the name's strand, bit 3 in the tile entries
    if nxt is not None:
        strand = np.ascontiguousarray(gstrand | ((nxt >= 0).astype(np.uint8) << 3))
    if hap is not None:
        strand = np.ascontiguousarray(strand | (np.asarray(hap, np.uint8) & 3) << 1)   # bits 1-2: HP tag
    n = int(rs.shape[0])
    Lref = int(ref_codes.shape[0])
    pos_lo = 1 if pos_lo is None else max(1, int(pos_lo))
    pos_hi = Lref if pos_hi is None else min(Lref, int(pos_hi))
    if pos_hi < pos_lo:
        pos_hi = pos_lo
    codes_len, n_ent = C.c_int64(), C.c_int64()
    tile_pos0, n_tiles = C.c_int32(), C.c_int32()
    _check(L.nc_pack_plan(n, _lib.npp(rs), _lib.npp(re_), _lib.npp(keep), tile_size, pos_lo, pos_hi,
                          C.byref(codes_len), C.byref(tile_pos0), C.byref(n_tiles), C.byref(n_ent)), "nc_pack_plan")
    out_codes = np.empty(codes_len.value, np.uint8)
    tile_off = np.empty(n_tiles.value + 1, np.int32)
    tile_ent = np.empty(max(1, n_ent.value), _lib.TILE_ENTRY_DTYPE)
    _check(L.nc_pack_fill(n, _lib.npp(rs), _lib.npp(re_), _lib.npp(ro), _lib.npp(cd), _lib.npp(strand), _lib.npp(keep),
                          tile_size, tile_pos0.value, n_tiles.value, _lib.npp(out_codes), codes_len.value,
                          _lib.npp(tile_off), _lib.npp(tile_ent), n_ent.value), "nc_pack_fill")
    # reference codes on the tile grid
    npos = n_tiles.value * tile_size
    rc = np.full(npos, 4, np.uint8)
    a = max(1, tile_pos0.value)
    b = min(Lref, tile_pos0.value + npos - 1)
    if b >= a:
        rc[a - tile_pos0.value:b - tile_pos0.value + 1] = ref_codes[a - 1:b]
    if exclude:
        for (x0, x1) in exclude:
            lo = max(int(x0), tile_pos0.value) - tile_pos0.value
            hi = min(int(x1), tile_pos0.value + npos) - tile_pos0.value
            if hi > lo:
                rc[lo:hi] = 4
    hp = HostPack(codes=out_codes, tile_size=tile_size, tile_pos0=tile_pos0.value, n_tiles=n_tiles.value,
                  tile_off=tile_off, tile_ent=tile_ent[:n_ent.value], ref_code=rc, pos_lo=pos_lo, pos_hi=pos_hi)
    if nxt is not None:
        hp.mates = mate_table(nxt, keep, rs, re_)
    if events is not None:
        ev_off, ev_pos, ev_len = (np.asarray(a) for a in events)
        kept = np.nonzero(keep)[0]
        cnt = (ev_off[1:] - ev_off[:-1])[kept]
        off = np.zeros(kept.size + 1, np.int32)
        np.cumsum(cnt, out=off[1:])
        idx = np.concatenate([np.arange(ev_off[r], ev_off[r + 1]) for r in kept]) if kept.size else np.zeros(0, np.int64)
        hp.ev_off, hp.ev_pos, hp.ev_len = off, np.ascontiguousarray(ev_pos[idx], np.int32), np.ascontiguousarray(ev_len[idx], np.int32)
        hp.read_hap = np.ascontiguousarray((np.asarray(hap, np.uint8) if hap is not None else np.zeros(n, np.uint8))[kept])
    return hp


def pack_world(world: World, **kw) -> HostPack:
    if "events" in world.meta:
        kw.setdefault("hap", world.meta["hap"])
        kw.setdefault("events", world.meta["events"])
    kw.setdefault("names", world_names(world))
    kw.setdefault("name_gid", world_name_gid(world, kw.get("supplementary", False)))
    return pack_reads(world.read_start, world.read_end, world.read_off, world.codes, world.read_flag,
                      world_ref_codes(world), **kw)


def world_names(world):
    """the alignments' read names if the World carries one per alignment (else None: every name is taken to be unique)"""
    nm = getattr(world, "names", None)
    return nm if nm is not None and len(nm) == len(world.read_start) else None


def world_name_gid(world, supplementary):
    """nc_decoded_name_groups' answer for this World under the flag filter, if its decode computed one (bam.read_bam)"""
    g = world.meta.get("name_gid") if getattr(world, "meta", None) else None
    return None if g is None else g.get(bool(supplementary))
