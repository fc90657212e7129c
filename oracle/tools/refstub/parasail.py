"""Stub `parasail` (container-only test tooling).  parasail-python is absent from the build container; its
global affine-gap alignment is supplied by a pluggable BACKEND(s1, s2, open, extend, match, mismatch) ->
[(op, count)] with parasail's CIGAR op codes ('I' 1, 'D' 2, '=' 7, 'X' 8), and handed to the reference in the
encoding its own code decodes (`x & 0xf`, `x >> 4`, generate_indel_pileups.py:80).  Traceback tie-breaking is
the backend's (parasail's is unpinned)."""

BACKEND = None


def matrix_create(alphabet, match, mismatch):
    return (alphabet, match, mismatch)


class _Cigar:
    def __init__(self, ops):
        self.seq = [(int(cnt) << 4) | int(op) for op, cnt in ops]


class _Result:
    def __init__(self, ops):
        self.cigar = _Cigar(ops)


def nw_trace(s1, s2, open, extend, matrix):
    if BACKEND is None:
        raise NotImplementedError("parasail is absent in the build container (SURVEY.md 8c): set parasail.BACKEND")
    _, match, mismatch = matrix
    return _Result(BACKEND(s1, s2, open, extend, match, mismatch))
