"""Region / chunk helpers with the reference's names and results (nanocaller_src/utils.py).

Only `get_chunks` is on the hot path: chunk boundaries decide the coverage-normalisation constant (quirk E2)
and the duplicated boundary records (quirk E3), so they must be reproduced exactly.  `get_regions_list` is kept so
that the reference's `run(args)` (NanoCaller:12-56) finds what it imports from this module; contig names and lengths
come from the library's own BAM reader instead of pysam.
"""
import datetime
import os
import shutil
import sys
from subprocess import PIPE, Popen

_HAPLOID_ALWAYS = ('chrY', 'Y', 'chrM', 'M')          # utils.py:54-58
_X = ('chrX', 'X')


def _contigs(bam_path):
    from .bam import BamFile
    bf = BamFile(bam_path)
    try:
        return dict(zip(bf.references, bf.lengths))
    finally:
        bf.close()


def get_regions_list(args):
    """utils.py:6-63 -> tuple of (contig, start, end, ploidy).  Sources in the reference's order of precedence:
    --wgs_contigs ('chr1-22XY' or '1-22XY'), --regions ('chr1' or 'chr1:100-200'), --bed, else every contig of the BAM.
    chrY / chrM are always haploid, chrX with --haploid_X, everything with --haploid_genome."""
    lengths = _contigs(args.bam)
    ploidy = 'haploid' if args.haploid_genome else 'diploid'
    now = lambda: str(datetime.datetime.now())                       # noqa: E731
    out = []
    if getattr(args, 'wgs_contigs', None):
        prefix = 'chr' if args.wgs_contigs == 'chr1-22XY' else ''
        for name in [str(i) for i in range(1, 23)] + ['X', 'Y']:
            if prefix + name in lengths:
                out.append([prefix + name, 1, lengths[prefix + name], ploidy])
    elif getattr(args, 'regions', None):
        for r in args.regions:
            name, _, span = r.partition(':')
            if r.count(':') > 1:
                print('\n%s: Invalid region %s.' % (now(), r), flush=True)
            elif not span:
                if name in lengths:
                    out.append([name, 1, lengths[name], ploidy])
                else:
                    print('\n%s: Contig %s not present in the BAM file.' % (now(), name), flush=True)
            else:
                cord = span.split('-')
                if len(cord) == 2:
                    out.append([name, int(cord[0]), int(cord[1]), ploidy])
                else:
                    print('\n%s: Invalid region %s.' % (now(), r), flush=True)
    elif getattr(args, 'bed', None):
        with open(args.bed) as bed_file:
            for line in bed_file:
                t = line.rstrip('\n').split()
                if not t:
                    continue
                if t[0] in lengths:
                    out.append([t[0], int(t[1]), int(t[2]), ploidy])
                else:
                    print('\n%s: Contig %s not present in the BAM file.' % (now(), t[0]), flush=True)
    else:
        out = [[name, 1, ln, ploidy] for name, ln in lengths.items()]
    if not out:
        print('\n%s: No valid regions found.' % now(), flush=True)
        sys.exit(2)
    for reg in out:
        if reg[0] in _HAPLOID_ALWAYS:
            reg[3] = 'haploid'
        elif reg[0] in _X:
            reg[3] = 'haploid' if args.haploid_X else 'diploid'
    return tuple(tuple(reg) for reg in out)


def get_chunks(regions_list, cpu, max_chunk_size=500000, min_chunk_size=10000):
    total_bases = sum(region[2] - region[1] + 1 for region in regions_list)
    chunksize = min(max_chunk_size, max(min_chunk_size, total_bases // cpu + 1))      # utils.py:72
    chunks_list = []
    for contig, start, end, ploidy in regions_list:
        for chunk in range(start, end, chunksize):                                    # end inclusive downstream
            chunks_list.append({'chrom': contig, 'start': chunk, 'end': min(end, chunk + chunksize), 'ploidy': ploidy})
    return chunks_list


def run_cmd(cmd, verbose=False, output=False, error=False):
    """utils.py:85-103: shell helper for the optional external steps (whatshap)"""
    stream = Popen(cmd, shell=True, stdout=PIPE, stderr=PIPE)
    stdout, stderr = (b.decode('utf-8') for b in stream.communicate())
    if (stderr and error) or verbose:
        print(stderr, flush=True)
    if verbose:
        print(stdout, flush=True)
    if output:
        return stdout
    if error:
        return stderr


def remove_path(path):
    if os.path.isdir(path):
        shutil.rmtree(path)
    elif os.path.exists(path):
        os.remove(path)


def make_and_remove_path(path):
    remove_path(path)
    os.makedirs(path)
