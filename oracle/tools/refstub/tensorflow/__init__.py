"""Stub `tensorflow` (container-only test tooling): lets /root/reference/nanocaller_src/snpCaller.py
be IMPORTED so its genotype-rule / VCF-text code (snpCaller.py:113-198) can be run on canned
probabilities.  No arithmetic lives here; model classes are replaced by the golden generator."""
from . import keras  # noqa: F401


def concat(*a, **k):
    raise NotImplementedError
