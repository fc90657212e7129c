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
from .pack import pack_world
from .synth import World

_SOURCES = {}
_PACKS = {}


def register_alignments(key, world: World):
    """Make `dct['sam_path'] == key` resolve to decoded alignments."""
    _SOURCES[key] = world
    for k in [k for k in _PACKS if k[0] == key]:
        del _PACKS[k]


_BAM_WORLDS = {}


def _resolve(sam_path, chrom=None, fasta_path=None) -> World:
    if isinstance(sam_path, World):
        return sam_path
    if sam_path in _SOURCES:
        return _SOURCES[sam_path]
    if isinstance(sam_path, str) and os.path.exists(sam_path):
        if chrom is None or not fasta_path:
            raise ValueError("decoding %r needs the contig name and dct['fasta_path']" % sam_path)
        key = (sam_path, fasta_path, chrom)
        if key not in _BAM_WORLDS:
            from .bam import read_bam
            _BAM_WORLDS[key] = read_bam(sam_path, fasta_path, chrom)
        return _BAM_WORLDS[key]
    raise FileNotFoundError("alignments %r: not a BAM file, a World, or a registered key" % (sam_path,))


def _exclude_rows(dct, chrom):
    ex = dct.get("exclude_bed")
    if not ex:
        return None
    if isinstance(ex, (list, tuple)):
        return tuple((int(a), int(b)) for (c, a, b) in ex if c == chrom)
    opener = gzip.open if str(ex).endswith(".gz") else open          # BGZF is a valid multi-member gzip stream
    rows = []
    with opener(ex, "rt") as f:
        for line in f:
            t = line.split()
            if len(t) >= 3 and t[0] == chrom:
                rows.append((int(t[1]), int(t[2])))
    return tuple(rows)


def device_pack_for(dct, chrom, device=0):
    """Packed + uploaded alignments of a contig (cached per source / filter / exclusion list)."""
    world = _resolve(dct["sam_path"], chrom, dct.get("fasta_path"))
    excl = _exclude_rows(dct, chrom)
    key = (dct["sam_path"] if not isinstance(dct["sam_path"], World) else id(world), bool(dct.get("supplementary")),
           excl, device)
    if key not in _PACKS:
        eng = get_engine(device)
        hp = pack_world(world, supplementary=bool(dct.get("supplementary")), exclude=excl)
        _PACKS[key] = (eng.upload(hp), world)
    return _PACKS[key][0]


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


__all__ = ["get_snp_testing_candidates", "register_alignments", "device_pack_for", "_lib"]
