"""Host-side mirror of the reference's SNP featuriser interface.

`get_snp_testing_candidates(dct, region)` keeps the call signature, dict keys and the 8-tuple of
/root/reference nanocaller_src/generate_SNP_pileups.py:103-279; the column scan, neighbour selection and
tensor build run in the HIP kernels of libnanocaller_hip.so (nc_snp_scan / nc_snp_featurize).

Alignments: `dct['sam_path']` is a coordinate-sorted BAM (+ optional .bai) decoded by the library's native
reader (nanocaller_amd/bam.py) together with `dct['fasta_path']`; already-decoded alignments are
accepted too (a `synth.World`, or a key registered with `register_alignments`).  `dct['exclude_bed']` is a BED /
bgzipped BED path (rows of the requested contig are used, like tbx.fetch(chrom), :114-116) or a list of
(chrom, start, end) rows.
"""
from __future__ import annotations

import gzip
import os

import numpy as np

from . import _lib
from .engine import get_engine
from .synth import World

_SOURCES = {}


class _LRU:
    """Small bounded cache: a whole-genome run walks contig after contig, and a decoded contig (~1 B per aligned base on
    the host) or its pack (the same in HBM) must not outlive the next ones.  The bound counts entries (contigs x variants);
    NANOCALLER_CONTIG_CACHE overrides it."""

    def __init__(self, cap):
        import threading
        self.cap = max(1, int(os.environ.get("NANOCALLER_CONTIG_CACHE", cap)))
        self.d = {}
        self.lock = threading.Lock()                                 # the caller's ingest threads prepare two contigs side by side

    def get(self, key, make):
        with self.lock:
            if key in self.d:
                self.d[key] = self.d.pop(key)                        # most recently used last
                return self.d[key]
        val = make()                                                 # (outside the lock: a decode takes tens of milliseconds)
        with self.lock:
            self.d[key] = val
            while len(self.d) > self.cap:
                self.d.pop(next(iter(self.d)))
        return val

    def drop(self, pred):
        with self.lock:
            for k in [k for k in self.d if pred(k)]:
                del self.d[k]

    def __len__(self):
        return len(self.d)

    def __contains__(self, key):
        return key in self.d


_BAM_WORLDS = _LRU(2)        # (bam, fasta, contig) -> decoded World
_PACKS = _LRU(3)             # (source, fasta, contig, supplementary, exclusions, device) -> (DevicePack, World)


# BED files of centromere / telomere intervals that ship with the reference (data; `--exclude_bed hg38` of its CLI resolves to
# nanocaller_src/release_data/bed_files/hg38_centro_telo.bed.gz next to the script, NanoCaller:21-22): kept at the same place
BED_SHORTCUTS = ("hg38", "hg19", "mm10", "mm39")
RELEASE_BED_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nanocaller_src", "release_data", "bed_files")


def register_alignments(key, world: World):
    """Make `dct['sam_path'] == key` resolve to decoded alignments."""
    _SOURCES[key] = world
    _PACKS.drop(lambda k: k[0] == key)


def release_contig(chrom=None):
    """Forget the decoded alignments and HBM packs of `chrom` (all contigs when None): callers that walk a genome contig by
    contig call this when they move on (the LRU bound does the same, later)."""
    _BAM_WORLDS.drop(lambda k: chrom is None or k[2] == chrom)
    _PACKS.drop(lambda k: chrom is None or k[2] == chrom)
    import sys
    gip = sys.modules.get(__package__ + ".generate_indel_pileups")       # (the indel route's device-ingested contig: same lifetime)
    if gip is not None:
        for k in [k for k in gip._DEV_INGEST if chrom is None or k[1] == chrom]:
            del gip._DEV_INGEST[k]


DECODES = []                  # (bam, contig, span or None) of every BAM decode of this process (tests: no contig is decoded twice)


def _resolve(sam_path, chrom=None, fasta_path=None, span=None) -> World:
    if isinstance(sam_path, World):
        return sam_path
    if sam_path in _SOURCES:
        return _SOURCES[sam_path]
    if isinstance(sam_path, str) and os.path.exists(sam_path):
        if chrom is None or not fasta_path:
            raise ValueError("decoding %r needs the contig name and dct['fasta_path']" % sam_path)
        from .bam import read_bam

        def make():
            DECODES.append((sam_path, chrom, span))
            return read_bam(sam_path, fasta_path, chrom) if span is None else read_bam(sam_path, fasta_path, chrom, span[0], span[1])
        return _BAM_WORLDS.get((sam_path, fasta_path, chrom) + ((span,) if span else ()), make)
    raise FileNotFoundError("alignments %r: not a BAM file, a World, or a registered key" % (sam_path,))


def _exclude_rows(dct, chrom):
    ex = dct.get("exclude_bed")
    if not ex:
        return None
    if isinstance(ex, (list, tuple)):
        return tuple((int(a), int(b)) for (c, a, b) in ex if c == chrom)
    ex = str(ex)
    if ex in BED_SHORTCUTS:                                          # the names the reference's CLI accepts (NanoCaller:21-22)
        ex = os.path.join(RELEASE_BED_DIR, "%s_centro_telo.bed.gz" % ex)
    with open(ex, "rb") as f:
        head = f.read(18)
    if head[:4] == b"\x1f\x8b\x08\x04" and head[12:14] == b"BC":    # bgzip output, what pysam.TabixFile needs (:113-116)
        from .vcfio import bgzf_read
        text = bgzf_read(ex).decode("ascii", "replace")
    else:
        with (gzip.open if head[:2] == b"\x1f\x8b" else open)(ex, "rt") as f:
            text = f.read()
    rows = []
    for line in text.splitlines():
        t = line.split()
        if len(t) >= 3 and t[0] == chrom:
            rows.append((int(t[1]), int(t[2])))
    return tuple(rows)


def _check_supported(world, sam_path, chrom, supplementary=False, by_name=False):
    """inputs the library does not reproduce are refused, not silently accepted (nc_decoded_check): among the alignments
    the pileup keeps, reference skips in the CIGAR (the reference raises KeyError on their pileup symbols, quirk E10) and same-name
    alignments that overlap on the reference (the reference's per-column dicts are keyed by name: one entry where the pack has two).
    by_name (the SNP featuriser's callers, round 6): alignments that share a read name are keyed by name as the reference does
    (pack.name_groups -> nc_snp_set_mates) when the World carries the names; the indel route still refuses them."""
    mask = 0x704 if supplementary else 0xF04                      # unmapped / secondary / qcfail / duplicate (/ supplementary) are never kept
    keep = (world.read_flag & mask) == 0
    n_skip = int(np.count_nonzero(keep & ((world.read_flag & _lib.FLAG_REFSKIP) != 0)))
    from .pack import world_names
    if by_name and world_names(world) is not None:
        n_dup = 0                                                 # keyed by name downstream
    elif "unsupported" in world.meta:
        n_dup = world.meta["unsupported"].get(bool(supplementary), (0, 0))[1]
    elif world_names(world) is not None:
        from .bam import same_name_overlaps                       # a World nobody counted for: count now rather than assume none
        n_dup = same_name_overlaps(world.names, world.read_start, world.read_end, keep)
    else:
        n_dup = 0
    if n_skip or n_dup:
        what = []
        if n_skip:
            what.append("%d alignments with a reference skip (CIGAR N) would enter the pileup; the reference's code table has no entry for "
                        "their '>' / '<' symbols (generate_SNP_pileups.py:104)" % n_skip)
        if n_dup:
            what.append("%d pairs of kept alignments carry the same read name and overlap on the reference; the reference's per-column "
                        "dicts hold one entry per name (generate_SNP_pileups.py:175,185,208)" % n_dup)
        err = _lib.NanoCallerHipError("%s, contig %s: %s -- NC_ERR_UNSUPPORTED" % (sam_path, chrom, "; ".join(what)))
        err.status = _lib.NC_ERR_UNSUPPORTED
        raise err


def contig_span(sam_path, chrom, chunks):
    """the part of a contig a rank has to decode for `chunks` (all of one contig): None = the whole contig, else (lo, hi) = the
    chunks' extent +- the 50 kb scan flank (generate_SNP_pileups.py:137,156) when that is less than 90 % of the contig (a contig
    split over several ranks, shard.shard_plan)"""
    if not (isinstance(sam_path, str) and os.path.exists(sam_path)):
        return None
    from .bam import BamFile
    bf = BamFile(sam_path)
    try:
        length = bf.get_reference_length(chrom)
    finally:
        bf.close()
    lo = max(1, min(c['start'] for c in chunks) - _lib.FLANK)
    hi = min(length, max(c['end'] for c in chunks) + _lib.FLANK)
    return None if (hi - lo + 1) >= 0.9 * length else (lo, hi)


def device_pack(sam_path, fasta_path, chrom, supplementary=False, excl=None, device=0, span=None, by_name=False):
    """Packed + uploaded alignments of ONE contig -> (DevicePack, World).  The key carries the contig (one BAM holds many),
    the FASTA, the flag filter, the exclusion list and the device; the SNP and the indel path share the entry.
    span (lo, hi): only the alignments overlapping it are decoded and packed (a rank that owns part of a contig)."""
    world = _resolve(sam_path, chrom, fasta_path, span)
    _check_supported(world, sam_path, chrom, supplementary, by_name=by_name)
    if world.chrom != chrom:
        raise ValueError("alignments of contig %r requested, the source holds %r" % (chrom, world.chrom))
    src = id(world) if isinstance(sam_path, World) else sam_path
    key = (src, fasta_path, chrom, bool(supplementary), excl, device) + ((span,) if span else ())

    def make():
        # the contig crosses PCIe in the reference-difference wire form (~0.2 B instead of 1 B per pileup entry) and is
        # expanded to the position-addressed codes in HBM (wire.py, nc_wire_expand)
        from .wire import build_wire_from_world, upload_wire
        kw = dict(pos_lo=span[0], pos_hi=span[1]) if span else {}
        return (upload_wire(get_engine(device), build_wire_from_world(world, supplementary=bool(supplementary), exclude=excl, **kw)), world)
    return _PACKS.get(key, make)


def device_pack_for(dct, chrom, device=0, span=None):
    """Packed + uploaded alignments of contig `chrom` of dct['sam_path'] (cached per source / contig / filter / exclusions / span)."""
    return device_pack(dct["sam_path"], dct.get("fasta_path"), chrom, dct.get("supplementary"), _exclude_rows(dct, chrom), device, span, by_name=True)[0]


def get_snp_testing_candidates(dct, region, device=0):
    """-> (pos, ref_onehot, mat, dp, freq, depth, fwd_dp, rev_dp); empty lists / depth 0 when there is no
    candidate (generate_SNP_pileups.py:193-197, 279)."""
    chrom, start, end, ploidy = region["chrom"], region["start"], region["end"], region["ploidy"]
    eng = get_engine(device)
    eng.use_torch_stream()
    dp_ = device_pack_for(dct, chrom, device)
    sites = eng.snp_scan(dp_, [(start, end)], mincov=dct["mincov"], min_allele_freq=dct["min_allele_freq"],
                         threshold=dct["threshold"], haploid=(ploidy == "haploid"))
    empty = ([], [], [], [], [], 0, [], [])
    if sites.n_sites == 0:
        return empty
    eng.snp_featurize(dp_, sites, seq=dct["seq"], maxcov=dct["maxcov"], min_nbr_sites=dct["min_nbr_sites"])
    valid = sites.valid.cpu().numpy().astype(bool)
    if not valid.any():
        return empty
    mat = sites.x.cpu().numpy()[valid]
    ref = sites.ref_code.cpu().numpy()[valid]
    depth_each = sites.depth.cpu().numpy()[valid]
    n = sites.dp[valid].astype(np.int64)
    alt = sites.alt[valid].astype(np.int64)
    output_ref = np.eye(4, dtype=np.int32)[ref]                                   # :269-270
    return (sites.pos[valid].astype(np.int64), output_ref, mat, n, alt.astype(np.float64) / n.astype(np.float64),
            np.mean(depth_each.astype(np.float64)), sites.fwd_dp.cpu().numpy()[valid].astype(np.float64),
            sites.rev_dp.cpu().numpy()[valid].astype(np.float64))


__all__ = ["get_snp_testing_candidates", "register_alignments", "device_pack_for", "device_pack", "release_contig", "_lib"]
