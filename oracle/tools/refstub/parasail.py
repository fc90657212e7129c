"""Stub `parasail` (container-only test tooling): only what the reference needs at import."""


def matrix_create(alphabet, match, mismatch):
    return (alphabet, match, mismatch)


def nw_trace(*a, **k):
    raise NotImplementedError("parasail is absent in the build container (SURVEY.md 8c)")
