"""Bench-scale synthetic workloads generated directly in HBM (bench tooling, not part of the hot path).

Same distributions as `synth.make_world` (generator synth_v1): read placement / lengths /
strands / haplotypes come from numpy PCG64(seed) on the host (a few hundred thousand reads), the per-base
work (truth haplotypes, substitutions, deletions, systematic-error columns) is done with torch on the GPU
straight into the packed slot layout, because a chr20-sized 30x pile is ~2 GB of codes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .engine import DevicePack


def _read_layout(rng, L, depth, tech):
    if tech == "ont":
        mean_len = 10_000
        n0 = int(depth * L / mean_len * 1.4) + 16
        lens = np.clip(np.exp(rng.normal(np.log(mean_len), 0.6, size=n0)), 1_000, 100_000).astype(np.int64)
    elif tech == "hifi":
        mean_len = 15_000
        n0 = int(depth * L / mean_len * 1.2) + 16
        lens = np.clip(rng.normal(mean_len, 2_000, size=n0), 500, None).astype(np.int64)
    else:
        raise ValueError(tech)
    tot = np.cumsum(lens)
    n = min(int(np.searchsorted(tot, depth * L) + 1), n0)
    lens = lens[:n]
    starts = rng.integers(1 - mean_len // 2, L + 1, size=n)
    ends = np.clip(starts + lens, 2, L + 1)
    starts = np.clip(starts, 1, L)
    keep = ends - starts >= 10
    starts, ends = starts[keep], ends[keep]
    order = np.argsort(starts, kind="stable")
    return starts[order].astype(np.int32), ends[order].astype(np.int32)


def make_device_workload(eng, L, depth=30.0, tech="ont", seed=812, tile_size=2048, het_rate=1 / 1000.0,
                         hom_rate=1 / 2000.0, sys_err_rate=0.01, mask_frac=0.01, chunk_bytes=1 << 27):
    """-> (DevicePack, info dict with host read arrays).  Positions 1..L of one contig."""
    dev = eng.device
    rng = np.random.Generator(np.random.PCG64(seed))
    starts, ends = _read_layout(rng, L, depth, tech)
    R = starts.shape[0]
    strand = rng.integers(0, 2, size=R).astype(np.uint8)
    hap = rng.integers(0, 2, size=R).astype(np.uint8)
    p_sub, p_del = (0.04, 0.04) if tech == "ont" else (0.001, 0.001)

    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    # truth (index p: position p; index 0 unused)
    refc = torch.randint(0, 4, (L + 1,), dtype=torch.uint8, device=dev, generator=g)
    u = torch.rand(L + 1, device=dev, generator=g)
    alt = (refc + torch.randint(1, 4, (L + 1,), dtype=torch.uint8, device=dev, generator=g)) % 4
    het = u < het_rate
    hom = (u >= het_rate) & (u < het_rate + hom_rate)
    which = torch.rand(L + 1, device=dev, generator=g) < 0.5
    hap0 = torch.where(hom | (het & which), alt, refc)
    hap1 = torch.where(hom | (het & ~which), alt, refc)
    haps = torch.stack([hap0, hap1])
    sys_col = torch.rand(L + 1, device=dev, generator=g) < sys_err_rate
    sys_base = (refc + torch.randint(1, 4, (L + 1,), dtype=torch.uint8, device=dev, generator=g)) % 4
    del u, alt, het, hom, which, hap0, hap1

    # slot layout (identical to nc_pack_fill): slot r = [floor16(start), ceil16(end)), packed back to back
    lo16 = starts.astype(np.int64) & ~np.int64(15)
    hi16 = (ends.astype(np.int64) + 15) & ~np.int64(15)
    slot = hi16 - lo16
    slot_off = np.zeros(R + 1, np.int64)
    np.cumsum(slot, out=slot_off[1:])
    codes_len = int(slot_off[-1]) + 16
    base = slot_off[:-1] - lo16
    d_slot_off = torch.from_numpy(slot_off).to(dev)
    d_base = torch.from_numpy(base).to(dev)
    d_start = torch.from_numpy(starts.astype(np.int64)).to(dev)
    d_end = torch.from_numpy(ends.astype(np.int64)).to(dev)
    d_hap = torch.from_numpy(hap.astype(np.int64)).to(dev)
    codes = torch.full((codes_len,), _lib.CODE_ABSENT, dtype=torch.uint8, device=dev)
    total = int(slot_off[-1])
    for b0 in range(0, total, chunk_bytes):
        b1 = min(total, b0 + chunk_bytes)
        idx = torch.arange(b0, b1, device=dev, dtype=torch.int64)
        r = torch.bucketize(idx, d_slot_off, right=True) - 1
        p = idx - d_base[r]
        inside = (p >= d_start[r]) & (p < d_end[r])
        pc = p.clamp(1, L)
        c = haps[d_hap[r], pc]
        e = torch.rand(b1 - b0, device=dev, generator=g)
        sub = e < p_sub
        c = torch.where(sub, (c + torch.randint(1, 4, (b1 - b0,), dtype=torch.uint8, device=dev, generator=g)) % 4, c)
        edge = (p == d_start[r]) | (p == d_end[r] - 1)
        dele = (e >= p_sub) & (e < p_sub + p_del) & ~edge
        c = torch.where(dele, torch.full_like(c, 4), c)
        se = sys_col[pc] & (torch.rand(b1 - b0, device=dev, generator=g) < 0.2)
        c = torch.where(se, sys_base[pc], c)
        codes[b0:b1] = torch.where(inside, c, torch.full_like(c, _lib.CODE_ABSENT))
        del idx, r, p, inside, pc, c, e, sub, edge, dele, se
    # reference codes on the tile grid; soft-masked / N runs are skipped columns (quirk E4)
    tile_pos0 = 0
    n_tiles = L // tile_size + 1
    ref_code = torch.full((n_tiles * tile_size,), 4, dtype=torch.uint8, device=dev)
    ref_code[1:L + 1] = refc[1:]
    n_mask = int(mask_frac * L / 500)
    if n_mask:
        ms = torch.from_numpy(rng.integers(1, max(2, L - 600), size=n_mask)).to(dev)
        mi = (ms[:, None] + torch.arange(500, device=dev)[None, :]).reshape(-1)
        ref_code[mi] = 4
    # tile index from the library's packer (index-only mode)
    L_ = _lib.lib()
    rs = np.ascontiguousarray(starts)
    re_ = np.ascontiguousarray(ends)
    cl, ne = C.c_int64(), C.c_int64()
    tp0, nt = C.c_int32(), C.c_int32()
    rc = L_.nc_pack_plan(R, _lib.npp(rs), _lib.npp(re_), None, tile_size, 1, L, C.byref(cl), C.byref(tp0), C.byref(nt), C.byref(ne))
    assert rc == 0 and cl.value == codes_len and tp0.value == tile_pos0 and nt.value == n_tiles, (rc, cl.value, codes_len)
    tile_off = np.empty(n_tiles + 1, np.int32)
    tile_ent = np.empty(max(1, ne.value), _lib.TILE_ENTRY_DTYPE)
    rc = L_.nc_pack_fill(R, _lib.npp(rs), _lib.npp(re_), None, None, _lib.npp(strand), None, tile_size, tile_pos0, n_tiles,
                         None, codes_len, _lib.npp(tile_off), _lib.npp(tile_ent), ne.value)
    assert rc == 0, rc
    ent_bytes = np.frombuffer(tile_ent[:ne.value].tobytes(), np.uint8).copy()
    pack = DevicePack(codes=codes, tile_off=torch.from_numpy(tile_off).to(dev), tile_ent=torch.from_numpy(ent_bytes).to(dev),
                      ref_code=ref_code, tile_size=tile_size, tile_pos0=tile_pos0, n_tiles=n_tiles, n_entries=int(ne.value),
                      pos_lo=1, pos_hi=L)
    # wire form of the reference (nc_wire_*): true base in bits 0-2, bit 3 = skipped column
    ref_wire = torch.full((n_tiles * tile_size,), 4 | 8, dtype=torch.uint8, device=dev)
    ref_wire[1:L + 1] = refc[1:] | ((ref_code[1:L + 1] == 4).to(torch.uint8) << 3)
    info = dict(L=L, n_reads=R, read_start=starts, read_end=ends, read_base=base, strand=strand, ref_wire=ref_wire, tile_size=tile_size,
                pileup_entries=int((ends.astype(np.int64) - starts).sum()), tech=tech, depth=depth, seed=seed)
    torch.cuda.synchronize(dev)
    return pack, info


def host_sample_for_oracle(pack: DevicePack, info, pos_lo, pos_hi):
    """Copy the reads overlapping [pos_lo, pos_hi] (and the reference codes) back to the host in the
    oracle's read-major form, for bench.py's cpu_baseline leg.  -> dict(start, end, off, codes, strand, ref_codes)"""
    s, e, base = info["read_start"], info["read_end"], info["read_base"]
    sel = np.nonzero((s <= pos_hi) & (e > pos_lo))[0]
    r0, r1 = int(sel.min()), int(sel.max()) + 1                 # contiguous superset (coordinate order)
    lo16 = s[r0:r1].astype(np.int64) & ~np.int64(15)
    byte0 = int(base[r0] + lo16[0])
    hi16_last = (int(e[r1 - 1]) + 15) & ~15
    byte1 = int(base[r1 - 1] + hi16_last)
    raw = pack.codes[byte0:byte1].cpu().numpy()
    off = base[r0:r1] + s[r0:r1].astype(np.int64) - byte0       # codes[off + p - start]
    L = info["L"]
    ref = pack.ref_code[1:L + 1].cpu().numpy()                  # index p-1
    keep = np.zeros(r1 - r0, np.uint8)
    keep[sel - r0] = 1
    return dict(start=np.ascontiguousarray(s[r0:r1]), end=np.ascontiguousarray(e[r0:r1]), off=np.ascontiguousarray(off),
                codes=raw, strand=np.ascontiguousarray(info["strand"][r0:r1]), keep=keep, ref_codes=ref, L=L)


def wire_from_device_workload(pack: DevicePack, info, pin=True, pool=None):
    """The workload as the host would hold it after decoding a BAM: codes copied back to host memory and put into the
    reference-difference transfer form by the library's host builder (nc_wire_build) -- bench.py uploads THIS inside its
    timed region (the timed region starts at decoded alignments in pinned host memory).
    pool (a ThreadPoolExecutor): the arrays are fetched here, the host builder runs on the pool -> a Future of the WirePack
    (set-up of many contigs: the builder is host-only work, the generator of the next contig need not wait for it)."""
    from .wire import build_wire
    L = info["L"]
    codes_h = pack.codes.cpu().numpy()
    ref_h = info["ref_wire"][1:L + 1].cpu().numpy()
    off = info["read_base"] + info["read_start"].astype(np.int64)               # codes[off + p - start]: the slot layout is read-major
    n = info["n_reads"]
    rs, re_, ts, strand = info["read_start"], info["read_end"], info["tile_size"], info["strand"]

    def build():
        return build_wire(rs, re_, off, codes_h, None, ref_h, tile_size=ts, pos_lo=1, pos_hi=L, keep=np.ones(n, np.uint8), strand=strand, pin=pin)
    return pool.submit(build) if pool is not None else build()


# ---------------------------------------------------------------------------------------------------------------- indel workload
def make_indel_device_workload(eng, L, depth=30.0, seed=812, tile_size=2048, het_snp=1 / 1000.0, hom_snp=1 / 2000.0, het_indel=1 / 5000.0,
                               hom_indel=1 / 15000.0, max_len=50, p_sub=0.03, p_del=0.025, p_ins=0.012, carry=0.92, untagged=0.08,
                               mask_frac=0.005):
    """Synthetic ONT contig with planted indels and HP / PS tags, generated in HBM (nc_synth_indel_*): the inputs of the
    device-resident indel pipeline.  -> (DevicePack with .events / .reads, IndelReadsC, info dict).  info['tensors'] keeps the
    device arrays alive; info['truth'] = (hap_indel int8 [2, L + 1] on the device)."""
    from .generate_indel_pileups import TAIL_CAP  # noqa: F401  (documented bound of the tail section; synthetic reads have no soft clips)
    dev = eng.device
    eng.use_torch_stream()
    L_ = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(seed))
    starts, ends = _read_layout(rng, L, depth, "ont")
    R = starts.shape[0]
    strand = rng.integers(0, 2, size=R).astype(np.uint8)
    hap = rng.integers(1, 3, size=R).astype(np.uint8)
    hap[rng.random(R) < untagged] = 0
    ps = np.where(hap > 0, 1 + (starts // 400_000) * 400_000, 0).astype(np.int32)
    lo16 = starts.astype(np.int64) & ~np.int64(15)
    hi16 = (ends.astype(np.int64) + 15) & ~np.int64(15)
    slot_off = np.zeros(R + 1, np.int64)
    np.cumsum(hi16 - lo16, out=slot_off[1:])
    codes_len = int(slot_off[-1]) + 16
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    d_start, d_end, d_slot, d_hap, d_ps = t(starts), t(ends), t(slot_off), t(hap), t(ps)
    ref = torch.zeros(L + 1, dtype=torch.uint8, device=dev)
    hapb = torch.zeros(2 * (L + 1), dtype=torch.uint8, device=dev)
    hapi = torch.zeros(2 * (L + 1), dtype=torch.int8, device=dev)
    P = lambda x: C.c_void_p(x.data_ptr())                              # noqa: E731
    rc = L_.nc_synth_indel_truth(eng.ctx, L, seed, het_snp, hom_snp, het_indel, hom_indel, max_len, P(ref), P(hapb), P(hapi))
    assert rc == 0, rc
    ev_cnt = torch.zeros(R, dtype=torch.int32, device=dev)
    ins_cnt = torch.zeros(R, dtype=torch.int32, device=dev)
    args = (eng.ctx, L, seed, p_sub, p_del, p_ins, carry, R, P(d_start), P(d_end), P(d_slot), P(d_hap), P(hapb), P(hapi))
    rc = L_.nc_synth_indel_reads(*args, 0, P(ev_cnt), P(ins_cnt), None, None, None, None, None, None, None)
    assert rc == 0, rc
    ev_off = torch.zeros(R + 1, dtype=torch.int32, device=dev)
    ev_off[1:] = torch.cumsum(ev_cnt, 0)
    ins_off_read = torch.zeros(R + 1, dtype=torch.int32, device=dev)
    ins_off_read[1:] = torch.cumsum(ins_cnt, 0)
    n_ev, n_ins = int(ev_off[-1]), int(ins_off_read[-1])
    codes = torch.empty(codes_len, dtype=torch.uint8, device=dev)
    codes[codes_len - 16:] = _lib.CODE_ABSENT
    ev_pos = torch.zeros(max(n_ev, 4), dtype=torch.int32, device=dev)
    ev_len = torch.zeros(max(n_ev, 4), dtype=torch.int32, device=dev)
    ins_off = torch.zeros(max(n_ev, 3) + 1, dtype=torch.int32, device=dev)
    ins_bases = torch.zeros(max(n_ins, 4), dtype=torch.uint8, device=dev)
    rc = L_.nc_synth_indel_reads(*args, 1, None, None, P(ev_off), P(ins_off_read), P(codes), P(ev_pos), P(ev_len), P(ins_off), P(ins_bases))
    assert rc == 0, rc
    ins_off[n_ev] = n_ins
    # reference codes on the tile grid; a few soft-masked runs (quirk E4)
    tile_pos0, n_tiles = 0, L // tile_size + 1
    ref_code = torch.full((n_tiles * tile_size,), 4, dtype=torch.uint8, device=dev)
    ref_code[1:L + 1] = ref[1:]
    n_mask = int(mask_frac * L / 500)
    if n_mask:
        ms = torch.from_numpy(rng.integers(1, max(2, L - 600), size=n_mask)).to(dev)
        mi = (ms[:, None] + torch.arange(500, device=dev)[None, :]).reshape(-1)
        ref_code[mi] = 4
    rs, re_ = np.ascontiguousarray(starts), np.ascontiguousarray(ends)
    cl, ne = C.c_int64(), C.c_int64()
    tp0, nt = C.c_int32(), C.c_int32()
    rc = L_.nc_pack_plan(R, _lib.npp(rs), _lib.npp(re_), None, tile_size, 1, L, C.byref(cl), C.byref(tp0), C.byref(nt), C.byref(ne))
    assert rc == 0 and cl.value == codes_len and tp0.value == tile_pos0 and nt.value == n_tiles, (rc, cl.value, codes_len)
    tile_off = np.empty(n_tiles + 1, np.int32)
    tile_ent = np.empty(max(1, ne.value), _lib.TILE_ENTRY_DTYPE)
    sflag = np.ascontiguousarray(strand | (hap << 1))
    rc = L_.nc_pack_fill(R, _lib.npp(rs), _lib.npp(re_), None, None, _lib.npp(sflag), None, tile_size, tile_pos0, n_tiles, None, codes_len,
                         _lib.npp(tile_off), _lib.npp(tile_ent), ne.value)
    assert rc == 0, rc
    ent_bytes = np.frombuffer(tile_ent[:ne.value].tobytes(), np.uint8).copy()
    pack = DevicePack(codes=codes, tile_off=t(tile_off), tile_ent=t(ent_bytes), ref_code=ref_code, tile_size=tile_size, tile_pos0=tile_pos0,
                      n_tiles=n_tiles, n_entries=int(ne.value), pos_lo=1, pos_hi=L)
    pack.events = dict(n_reads=R, ev_off=ev_off, ev_pos=ev_pos, ev_len=ev_len, read_hap=d_hap)
    pack.reads = dict(n_reads=R, rd_start=d_start, rd_end=d_end, slot_off=d_slot)
    tail_off = torch.zeros(R + 1, dtype=torch.int32, device=dev)
    tail_bases = torch.zeros(16, dtype=torch.uint8, device=dev)
    rflag = torch.zeros(R, dtype=torch.uint8, device=dev)
    reads_c = _lib.IndelReadsC(n_reads=R, slot_off=d_slot.data_ptr(), rd_start=d_start.data_ptr(), rd_end=d_end.data_ptr(), ev_off=ev_off.data_ptr(),
                               ev_pos=ev_pos.data_ptr(), ev_len=ev_len.data_ptr(), ins_off=ins_off.data_ptr(), ins_bases=ins_bases.data_ptr(),
                               tail_off=tail_off.data_ptr(), tail_bases=tail_bases.data_ptr(), read_ps=d_ps.data_ptr(), read_hap=d_hap.data_ptr(),
                               read_flag=rflag.data_ptr())
    torch.cuda.synchronize(dev)
    info = dict(L=L, n_reads=R, read_start=starts, read_end=ends, hap=hap, ps=ps, strand=strand, n_events=n_ev, n_ins_bases=n_ins,
                pileup_entries=int((ends.astype(np.int64) - starts).sum()), seed=seed, depth=depth,
                tensors=dict(ins_off=ins_off, ins_bases=ins_bases, tail_off=tail_off, tail_bases=tail_bases, read_ps=d_ps, read_flag=rflag, ref=ref),
                truth=hapi.view(2, L + 1))
    return pack, reads_c, info
