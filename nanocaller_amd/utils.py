"""Chunking helper mirrored from the reference (nanocaller_src/utils.py:67-83).

Only `get_chunks` is on the hot path: chunk boundaries decide the coverage-normalisation constant (quirk E2)
and the duplicated boundary records (quirk E3), so they must be reproduced exactly.
"""


def get_chunks(regions_list, cpu, max_chunk_size=500000, min_chunk_size=10000):
    total_bases = sum(region[2] - region[1] + 1 for region in regions_list)
    chunksize = min(max_chunk_size, max(min_chunk_size, total_bases // cpu + 1))      # utils.py:72
    chunks_list = []
    for contig, start, end, ploidy in regions_list:
        for chunk in range(start, end, chunksize):                                    # end inclusive downstream
            chunks_list.append({'chrom': contig, 'start': chunk, 'end': min(end, chunk + chunksize), 'ploidy': ploidy})
    return chunks_list
