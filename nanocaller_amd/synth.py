"""Seeded synthetic alignment "worlds" (no real HG002 data is available offline).

A *world* is the decoded-alignment boundary of the hot path: one contig, a
reference string, and reads given read-major as one base code per reference
position they span (A=0 G=1 T=2 C=3, deletion/N=4 -- the code map of the
reference, generate_SNP_pileups.py:104).  It is the single source for

  * the stub ``pysam`` used in THIS container to run the reference featuriser
    and make golden vectors (oracle/tools/),
  * the packed HBM layout consumed by the HIP path (``pack.ReadPack``),
  * the CPU oracle (oracle/).

Generator ``synth_v1`` follows SURVEY.md section 8(d) / BASELINE.md section 3:
numpy ``Generator(PCG64(seed))``, seed 812 by default (the constant the
reference passes to np.random.seed, generate_SNP_pileups.py:201).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

BASES = "AGTC"          # code -> letter, generate_SNP_pileups.py:104 / snpCaller.py:14
CODE_DEL = 4            # '*' and 'N'
SEED = 812

# BAM flag bits that pileup() filters (generate_SNP_pileups.py:151-154)
FLAG_FILTER_DEFAULT = 0x4 | 0x100 | 0x200 | 0x400 | 0x800
FLAG_FILTER_SUPPL = 0x4 | 0x100 | 0x200 | 0x400


@dataclass
class World:
    chrom: str
    ref: str                      # contig sequence; position p (1-based) is ref[p-1]
    read_start: np.ndarray        # int32 [R], 1-based first reference position covered
    read_end: np.ndarray          # int32 [R], exclusive
    read_flag: np.ndarray         # int32 [R], BAM flag
    read_off: np.ndarray          # int64 [R+1], offsets into codes
    codes: np.ndarray             # uint8 [sum(len)], 0..4
    names: list = field(default_factory=list)
    hap: np.ndarray | None = None           # int8 [R] haplotype of origin (0/1), informational
    het_sites: np.ndarray | None = None
    hom_sites: np.ndarray | None = None
    # optional per-read insertion/deletion decorations for the stub pileup strings
    meta: dict = field(default_factory=dict)

    @property
    def n_reads(self) -> int:
        return int(self.read_start.shape[0])

    @property
    def length(self) -> int:
        return len(self.ref)

    def read_codes(self, i: int) -> np.ndarray:
        return self.codes[self.read_off[i]:self.read_off[i + 1]]


def _ref_codes_from_string(ref: str) -> np.ndarray:
    """Upper-case AGTC -> 0..3, everything else (lower-case soft-masked bases, N)
    -> 4.  Mirrors `s.upper() if s in 'AGTC' else '*'` (generate_SNP_pileups.py:137):
    the membership test happens BEFORE upper(), so soft-masked bases are dropped."""
    lut = np.full(256, CODE_DEL, dtype=np.uint8)
    for i, b in enumerate(BASES):
        lut[ord(b)] = i
    return lut[np.frombuffer(ref.encode("ascii"), dtype=np.uint8)]


def make_world(seed: int = SEED, length: int = 60_000, depth: float = 30.0,
               tech: str = "ont", chrom: str = "chr20",
               het_rate: float = 1 / 1000.0, hom_rate: float = 1 / 2000.0,
               sys_err_rate: float = 0.01, softmask_runs: int = 2, n_bases: int = 3,
               odd_flag_frac: float = 0.03, read_len_scale: float = 1.0) -> World:
    """Build a diploid synthetic world.

    ONT: read length lognormal(ln 10 kb, 0.6) clipped to [1 kb, 100 kb], 4 % substitution +
    4 % deletion.  HiFi: N(15 kb, 2 kb), 0.1 % + 0.1 %.  `read_len_scale` shrinks reads for
    the small test worlds.  1 % "systematic error" columns where an extra 20 % of reads carry
    one fixed wrong base (realistic candidate density)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    L = int(length)
    refc = rng.integers(0, 4, size=L, dtype=np.uint8)

    # truth haplotypes
    u = rng.random(L)
    het = u < het_rate
    hom = (u >= het_rate) & (u < het_rate + hom_rate)
    shift = rng.integers(1, 4, size=L, dtype=np.uint8)
    alt = (refc + shift) % 4
    hapc = np.stack([refc.copy(), refc.copy()])
    which = rng.integers(0, 2, size=L)
    hapc[0][het & (which == 0)] = alt[het & (which == 0)]
    hapc[1][het & (which == 1)] = alt[het & (which == 1)]
    hapc[0][hom] = alt[hom]
    hapc[1][hom] = alt[hom]

    # systematic error columns
    sys_col = rng.random(L) < sys_err_rate
    sys_base = (refc + rng.integers(1, 4, size=L, dtype=np.uint8)) % 4

    # reference string with soft-masked runs and a few N (quirk E4)
    ref_arr = np.frombuffer(BASES.encode(), dtype=np.uint8)[refc].copy()
    for _ in range(softmask_runs):
        a = int(rng.integers(0, max(1, L - 300)))
        ref_arr[a:a + 200] += 32  # lower-case
    for _ in range(n_bases):
        ref_arr[int(rng.integers(0, L))] = ord("N")
    ref = ref_arr.tobytes().decode("ascii")

    # reads
    if tech == "ont":
        mean_len = 10_000 * read_len_scale
        p_sub, p_del = 0.04, 0.04
    elif tech == "hifi":
        mean_len = 15_000 * read_len_scale
        p_sub, p_del = 0.001, 0.001
    else:
        raise ValueError(tech)
    n_reads = max(1, int(np.ceil(depth * L / mean_len * 1.15)))
    if tech == "ont":
        lens = np.exp(rng.normal(np.log(mean_len), 0.6, size=n_reads))
        lens = np.clip(lens, 1_000 * read_len_scale, 100_000 * read_len_scale)
    else:
        lens = np.clip(rng.normal(mean_len, 2_000 * read_len_scale, size=n_reads), 500, None)
    lens = np.maximum(lens.astype(np.int64), 20)
    # thin to the requested depth
    tot = np.cumsum(lens)
    n_reads = int(np.searchsorted(tot, depth * L * 1.0) + 1)
    n_reads = min(n_reads, lens.shape[0])
    lens = lens[:n_reads]
    starts = rng.integers(1 - int(mean_len) // 2, L + 1, size=n_reads)
    ends = starts + lens
    starts = np.clip(starts, 1, L)
    ends = np.clip(ends, 2, L + 1)
    keep = ends - starts >= 10
    starts, ends = starts[keep], ends[keep]
    order = np.argsort(starts, kind="stable")
    starts, ends = starts[order].astype(np.int32), ends[order].astype(np.int32)
    R = starts.shape[0]
    strand = rng.integers(0, 2, size=R)
    hap = rng.integers(0, 2, size=R).astype(np.int8)
    flags = (strand * 16).astype(np.int32)
    # a few reads that pileup()'s flag filter removes (secondary, supplementary, dup, qcfail, unmapped)
    odd = rng.random(R) < odd_flag_frac
    odd_bits = rng.choice(np.array([0x100, 0x800, 0x400, 0x200, 0x4]), size=R)
    flags = np.where(odd, flags | odd_bits, flags).astype(np.int32)

    lens = (ends - starts).astype(np.int64)
    off = np.zeros(R + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    total = int(off[-1])
    # per-base position index of every read base
    ridx = np.repeat(np.arange(R, dtype=np.int64), lens)
    pos0 = np.arange(total, dtype=np.int64) - off[ridx] + (starts[ridx].astype(np.int64) - 1)  # 0-based ref index
    codes = hapc[hap[ridx], pos0].astype(np.uint8)
    e = rng.random(total)
    sub = e < p_sub
    codes[sub] = (codes[sub] + rng.integers(1, 4, size=int(sub.sum()), dtype=np.uint8)) % 4
    dele = (e >= p_sub) & (e < p_sub + p_del)
    # first / last base of an alignment is never a deletion
    first = np.zeros(total, dtype=bool)
    first[off[:-1]] = True
    first[off[1:] - 1] = True
    codes[dele & ~first] = CODE_DEL
    se = sys_col[pos0] & (rng.random(total) < 0.2)
    codes[se] = sys_base[pos0[se]]
    nn = rng.random(total) < 2e-4        # read 'N'
    codes[nn & ~first] = CODE_DEL

    names = ["r%07d" % i for i in range(R)]
    return World(chrom=chrom, ref=ref, read_start=starts, read_end=ends, read_flag=flags,
                 read_off=off, codes=codes, names=names, hap=hap,
                 het_sites=np.nonzero(het)[0] + 1, hom_sites=np.nonzero(hom)[0] + 1)


def world_ref_codes(world: World) -> np.ndarray:
    """uint8 [L]: reference code per position (index p-1), 4 where the column is skipped."""
    return _ref_codes_from_string(world.ref)


def world_checksum(world: World) -> str:
    import hashlib
    h = hashlib.sha256()
    h.update(world.ref.encode())
    for a in (world.read_start, world.read_end, world.read_flag, world.read_off, world.codes):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def add_indels(world: World, seed: int = SEED, het_rate: float = 1 / 1500.0, noise_rate: float = 0.004,
               tag_frac: float = 0.85, carry: float = 0.85) -> World:
    """Decorate a world with per-read indel events and HP/PS phasing tags (inputs of the indel candidate scan,
    generate_indel_pileups.py:178-304).  Events are stored read-major in world.meta['events'] =
    (ev_off int32 [R+1], ev_pos int32, ev_len int32 signed: +insertion / -deletion, the marker sits on the column
    BEFORE the inserted / deleted bases as in pysam pileup strings) and world.meta['hap'] (uint8 [R]: 0 untagged,
    1/2 = HP), world.meta['ps'] (int32 [R]); world.meta['deco'] / ['tags'] carry the same for the stub pysam."""
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    R, L = world.n_reads, world.length
    true_hap = world.hap if world.hap is not None else rng.integers(0, 2, size=R)
    tagged = rng.random(R) < tag_frac
    hp = np.where(tagged, true_hap + 1, 0).astype(np.uint8)
    ps = (1000 * (world.read_start // 20000) + 1).astype(np.int32)
    sites = np.nonzero(rng.random(L) < het_rate)[0] + 1
    site_hap = rng.integers(0, 2, size=sites.size)
    site_len = rng.integers(1, 31, size=sites.size) * rng.choice(np.array([-1, 1]), size=sites.size)
    per_read = []
    for r in range(R):
        a, b = int(world.read_start[r]), int(world.read_end[r])
        ev = {}
        lo, hi = np.searchsorted(sites, a), np.searchsorted(sites, b - 1)
        for k in range(lo, hi):
            if site_hap[k] == true_hap[r] and rng.random() < carry:
                ev[int(sites[k])] = int(site_len[k])
        n_noise = rng.poisson(noise_rate * (b - a))
        for p in rng.integers(a, b - 1, size=n_noise) if b - 1 > a else []:
            ev.setdefault(int(p), int(rng.integers(1, 4)) * (1 if rng.random() < 0.5 else -1))
        per_read.append(sorted(ev.items()))
    ev_off = np.zeros(R + 1, np.int32)
    ev_off[1:] = np.cumsum([len(e) for e in per_read])
    ev_pos = np.array([p for e in per_read for p, _ in e], np.int32)
    ev_len = np.array([ln for e in per_read for _, ln in e], np.int32)
    deco = {}
    for r, e in enumerate(per_read):
        for p, ln in e:
            deco[(r, p - 1)] = ("+%d%s" % (ln, "ACGT"[p % 4] * ln)) if ln > 0 else ("-%d%s" % (-ln, "N" * (-ln)))
    world.meta["events"] = (ev_off, ev_pos, ev_len)
    world.meta["hap"] = hp
    world.meta["ps"] = ps
    world.meta["deco"] = deco
    world.meta["tags"] = {r: {"HP": int(hp[r]), "PS": int(ps[r])} for r in range(R) if hp[r]}
    return world


def unphase_blocks(world: World, blocks, seed: int = SEED, drop: float = 0.85, alt_base_frac: float = 0.3) -> World:
    """Inputs of the impute_indel_phase rule (generate_indel_pileups.py:278-304) on a world decorated by add_indels: the
    HP tag of most reads overlapping `blocks` [(start, end)] is removed (unphased stretches), and every insertion event
    gets its inserted bases in world.meta['ev_ins'] = (ins_off int64 [n_events+1], ins_bases uint8 ASCII) - a fraction of
    the reads carries a different base, so that the reads of a column fall into several groups."""
    rng = np.random.Generator(np.random.PCG64(seed + 4242))
    R = world.n_reads
    ev_off, ev_pos, ev_len = world.meta["events"]
    hp = np.array(world.meta["hap"], np.uint8)
    for (a, b) in blocks:
        m = (world.read_start < b) & (world.read_end > a)
        hp[m & (rng.random(R) < drop)] = 0
    n_ev = int(ev_pos.shape[0])
    ins_len = np.where(ev_len > 0, ev_len, 0).astype(np.int64)
    ins_off = np.zeros(n_ev + 1, np.int64)
    np.cumsum(ins_len, out=ins_off[1:])
    letter = np.frombuffer(b"ACGT", np.uint8)[(ev_pos % 4 + np.where(rng.random(n_ev) < alt_base_frac, rng.integers(1, 4, size=n_ev), 0)) % 4]
    ins_bases = np.repeat(letter, ins_len)
    return apply_impute_inputs(world, hp, ins_off, ins_bases)


def apply_impute_inputs(world: World, hp, ins_off, ins_bases) -> World:
    """Install HP tags and inserted bases (and the matching stub-pysam decorations)."""
    ev_off, ev_pos, ev_len = world.meta["events"]
    world.meta["hap"] = np.asarray(hp, np.uint8)
    world.meta["ev_ins"] = (np.asarray(ins_off, np.int64), np.asarray(ins_bases, np.uint8))
    ps = world.meta["ps"]
    world.meta["tags"] = {r: {"HP": int(hp[r]), "PS": int(ps[r])} for r in range(world.n_reads) if hp[r]}
    deco = {}
    raw = np.asarray(ins_bases, np.uint8).tobytes().decode("ascii")
    for r in range(world.n_reads):
        for k in range(int(ev_off[r]), int(ev_off[r + 1])):
            ln, p = int(ev_len[k]), int(ev_pos[k])
            deco[(r, p - 1)] = ("+%d%s" % (ln, raw[ins_off[k]:ins_off[k + 1]])) if ln > 0 else ("-%d%s" % (-ln, "N" * (-ln)))
    world.meta["deco"] = deco
    return world
