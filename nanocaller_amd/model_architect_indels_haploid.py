"""Host-side mirror of nanocaller_src/model_architect_indels_haploid.py: `haploid_Indel_model` on the HIP CNN (K9)."""
from __future__ import annotations

from . import _lib
from .model_architect_indel import Indel_model


class haploid_Indel_model(Indel_model):
    """model_architect_indels_haploid.py:7-48.  x (B,5,128,2) -> (B,1) sigmoid (probability of an indel allele)."""
    KIND = _lib.MODEL_INDEL_HAP
    ROWS = 5
