"""DEFLATE on the device (csrc/nc_inflate.hip, one lane per BGZF member) against zlib: every block type and the streams zlib's strategies make,
members at unaligned offsets, the members of the spec-assembled BAM fixture, and damaged inputs (reported, never a crash)."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    return c.compress(data) + c.flush()


def _run(payloads, sizes):
    import torch
    from nanocaller_amd.engine import get_engine
    eng = get_engine(0)
    rng = np.random.default_rng(1)
    off, blob = [], bytearray()
    for p in payloads:
        blob += bytes(rng.integers(0, 256, size=int(rng.integers(0, 7))).astype(np.uint8))      # unaligned starts, junk between members
        off.append(len(blob))
        blob += p
    blob += bytes(16)
    ooff = np.zeros(len(payloads) + 1, np.int64)
    np.cumsum(np.asarray(sizes, np.int64) + 3, out=ooff[1:])                                      # (3 spare bytes between outputs: overruns would show)
    dev = eng.device
    d_comp = torch.from_numpy(np.frombuffer(bytes(blob), np.uint8).copy()).to(dev)
    d_coff = torch.tensor(off, dtype=torch.int64, device=dev)
    d_clen = torch.tensor([len(p) for p in payloads], dtype=torch.int32, device=dev)
    d_out = torch.full((int(ooff[-1]) + 16,), 0xEE, dtype=torch.uint8, device=dev)
    d_ooff = torch.from_numpy(ooff[:-1].copy()).to(dev)
    d_isize = torch.tensor(list(sizes), dtype=torch.int32, device=dev)
    d_st = torch.full((len(payloads),), -1, dtype=torch.int32, device=dev)
    d_tok = torch.empty(((len(payloads) + 63) // 64) << 22, dtype=torch.int32, device=dev)
    d_ntok = torch.zeros(len(payloads), dtype=torch.int32, device=dev)
    eng.use_torch_stream()
    rc = eng.L.nc_inflate_device(eng.ctx, len(payloads), d_comp.data_ptr(), d_coff.data_ptr(), d_clen.data_ptr(), d_out.data_ptr(), d_ooff.data_ptr(),
                                 d_isize.data_ptr(), d_st.data_ptr(), d_tok.data_ptr(), d_ntok.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), ooff, d_st.cpu().numpy()


def _cases():
    rng = np.random.default_rng(7)
    text = (b"chr20\t1234567\t.\tA\tG\t33.10\tPASS\t.\tGT:DP:FQ\t0/1:31:0.4839\n" * 900)[:60_000]
    dna4 = bytes(rng.integers(0, 256, size=65_280).astype(np.uint8))                              # packed bases: incompressible
    qual = bytes((rng.normal(20, 6, size=65_280).clip(0, 60)).astype(np.uint8))                    # quality-like: entropy coded, few matches
    runs = b"".join(bytes([int(b)]) * int(n) for b, n in zip(rng.integers(0, 4, 400), rng.integers(1, 400, 400)))[:65_280]
    far = bytes(rng.integers(0, 256, size=20_000).astype(np.uint8))                                # matches at distance 20,000 / 31,000: older than k_lz's
    far2 = bytes(rng.integers(0, 256, size=31_000).astype(np.uint8))                               # 16 KB ring, read back from HBM
    data = [b"", b"A", b"AB" * 3, bytes(65_280), text, dna4, qual, runs, bytes(range(256)) * 255, far * 3, (far2 * 3)[:65_280], (qual[:9_000] + far[:9_000]) * 3]
    out = []
    for d in data:
        for kw in (dict(level=6), dict(level=1), dict(level=9), dict(level=0), dict(level=6, strategy=zlib.Z_FIXED), dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY),
                   dict(level=6, strategy=zlib.Z_RLE), dict(level=6, strategy=zlib.Z_FILTERED), dict(level=9, mem=1)):
            out.append((d, _deflate(d, **kw)))
    # several deflate blocks in one member (Z_FULL_FLUSH between pieces: stored + dynamic + fixed in one stream)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    multi = c.compress(text[:20_000]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(dna4[:20_000]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(runs[:20_000]) + c.flush()
    out.append((text[:20_000] + dna4[:20_000] + runs[:20_000], multi))
    return out


def test_device_inflate_equals_zlib():
    cases = _cases()
    out, ooff, st = _run([p for _, p in cases], [len(d) for d, _ in cases])
    assert not st.any(), np.nonzero(st)[0][:10]
    for k, (d, _) in enumerate(cases):
        got = out[ooff[k]:ooff[k] + len(d)].tobytes()
        assert got == d, k
        assert out[ooff[k] + len(d):ooff[k] + len(d) + 3].tolist() == [0xEE] * 3, k               # nothing written past the announced size
    kinds = {(p[0] >> 1) & 3 for _, p in cases if p}
    assert kinds == {0, 1, 2}


def test_runs_of_empty_stored_blocks_do_not_starve_the_stream_window():
    """legal deflate: dozens of empty stored blocks in a row (repeated sync flushes; 5 bytes each, no symbol loop runs) between real blocks --
    the per-lane window of the compressed stream is topped up per block header, not only inside the symbol loops"""
    rng = np.random.default_rng(11)
    a = bytes((rng.normal(20, 6, size=9_000).clip(0, 60)).astype(np.uint8))
    b = (b"chr20\t1234567\t.\tA\tG\t33.10\n" * 400)[:9_000]
    cases = []
    for n_empty in (1, 30, 64, 200):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        empty = b"\x00\x00\x00\xff\xff"                                                         # BFINAL = 0, BTYPE = 00, padding; LEN = 0, NLEN = 0xffff
        z = c.compress(a) + c.flush(zlib.Z_SYNC_FLUSH)                                            # (byte-aligned behind a sync flush: zlib itself writes one empty stored block there)
        z += empty * n_empty
        z += c.compress(b) + c.flush(zlib.Z_FULL_FLUSH)
        z += empty * n_empty
        z += c.compress(a[:100]) + c.flush()
        assert zlib.decompress(z, -15) == a + b + a[:100]
        cases.append((a + b + a[:100], z))
    out, ooff, st = _run([p for _, p in cases], [len(d) for d, _ in cases])
    assert not st.any(), st
    for k, (d, _) in enumerate(cases):
        assert out[ooff[k]:ooff[k] + len(d)].tobytes() == d, k


def test_device_inflate_on_the_members_of_the_spec_fixture():
    raw = open(os.path.join(G, "spec.bam"), "rb").read()
    pay, want, o = [], [], 0
    while o < len(raw):
        bsize = int.from_bytes(raw[o + 16:o + 18], "little") + 1
        p = raw[o + 18:o + bsize - 8]
        pay.append(p)
        want.append(zlib.decompress(p, -15))
        assert len(want[-1]) == int.from_bytes(raw[o + bsize - 4:o + bsize], "little")
        o += bsize
    out, ooff, st = _run(pay, [len(w) for w in want])
    assert not st.any() and len(pay) > 30
    assert b"".join(out[ooff[k]:ooff[k] + len(w)].tobytes() for k, w in enumerate(want)) == b"".join(want)


def test_damaged_members_are_reported():
    d = (b"ACGTTGCA" * 4000)[:30_000]
    good = _deflate(d)
    rng = np.random.default_rng(3)
    junk = bytes(rng.integers(0, 256, size=400).astype(np.uint8))
    cases = [(good, len(d)), (good, len(d) - 1), (good, len(d) + 1), (good[:len(good) // 2], len(d)), (junk, 5_000), (b"\x07", 10), (good, len(d))]
    out, ooff, st = _run([p for p, _ in cases], [n for _, n in cases])
    assert st[0] == 0 and st[-1] == 0 and all(st[1:-1] != 0), st
    assert out[ooff[0]:ooff[0] + len(d)].tobytes() == d and out[ooff[6]:ooff[6] + len(d)].tobytes() == d
    for k, (_, n) in enumerate(cases):
        assert out[ooff[k] + n:ooff[k] + n + 3].tolist() == [0xEE] * 3, k                         # even a damaged member stays inside its own output


def _crc_status(payloads, datas, trailers):
    """nc_bgzf_crc_device over hand-laid members: payload + trailer (CRC-32, ISIZE) back to back in one buffer, inflated bytes in another"""
    import torch
    from nanocaller_amd.engine import get_engine
    eng = get_engine(0)
    blob, coff = bytearray(), []
    for p, t in zip(payloads, trailers):
        blob += b"\x55" * (len(blob) % 3)                                                        # unaligned members
        coff.append(len(blob))
        blob += p + t
    blob += bytes(16)
    sizes = [len(d) for d in datas]
    ooff = np.zeros(len(datas) + 1, np.int64)
    np.cumsum(np.asarray(sizes, np.int64) + 1, out=ooff[1:])                                      # (odd gaps: every alignment of a member's first byte)
    raw = np.full(int(ooff[-1]) + 16, 0xEE, np.uint8)
    for k, d in enumerate(datas):
        raw[ooff[k]:ooff[k] + len(d)] = np.frombuffer(d, np.uint8)
    dev = eng.device
    t = lambda a, dt: torch.from_numpy(np.asarray(a, dt)).to(dev)                                 # noqa: E731
    d_comp, d_out = t(np.frombuffer(bytes(blob), np.uint8).copy(), np.uint8), t(raw, np.uint8)
    d_coff, d_clen, d_ooff, d_isize = t(coff, np.int64), t([len(p) for p in payloads], np.int32), t(ooff[:-1].copy(), np.int64), t(sizes, np.int32)
    d_st = torch.zeros(len(datas), dtype=torch.int32, device=dev)
    eng.use_torch_stream()
    rc = eng.L.nc_bgzf_crc_device(eng.ctx, len(datas), d_comp.data_ptr(), d_coff.data_ptr(), d_clen.data_ptr(), d_out.data_ptr(), d_ooff.data_ptr(),
                                  d_isize.data_ptr(), d_st.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    return d_st.cpu().numpy()


def test_device_crc32_equals_zlibs_and_reports_a_damaged_member():
    """k_crc32 (slice-by-4 per lane over 1024-byte slices cut from the member's end, tree combine by x^(8192 * 2^l) mod P) against zlib.crc32 on every
    length class: empty, shorter than a dword, one slice and a bit, exact multiples of 1024, the BGZF maximum 65,280 and 65,536; then the same
    members with one bit of data flipped, and with a wrong trailer: status 7"""
    rng = np.random.default_rng(5)
    lens = [0, 1, 2, 3, 4, 5, 7, 8, 63, 64, 1023, 1024, 1025, 2048, 4097, 10_000, 30_001, 65_279, 65_280, 65_535, 65_536]
    datas = [bytes(rng.integers(0, 256, size=n).astype(np.uint8)) for n in lens]
    pay = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 40))).astype(np.uint8)) for _ in lens]      # (the payload bytes themselves are not read)
    good = [int(zlib.crc32(d)).to_bytes(4, "little") + len(d).to_bytes(4, "little") for d in datas]
    st = _crc_status(pay, datas, good)
    assert not st.any(), [lens[k] for k in np.nonzero(st)[0]]
    flipped = []
    for d in datas:
        if d:
            b = bytearray(d)
            k = int(rng.integers(0, len(b)))
            b[k] ^= 1 << int(rng.integers(0, 8))
            flipped.append(bytes(b))
        else:
            flipped.append(d)
    st = _crc_status(pay, flipped, good)
    assert all((s == 7) == (n > 0) for s, n in zip(st.tolist(), lens)), st
    bad_trailer = [(int(zlib.crc32(d)) ^ 0x00010000).to_bytes(4, "little") + len(d).to_bytes(4, "little") for d in datas]
    st = _crc_status(pay, datas, bad_trailer)
    assert (st == 7).all()
