"""Host-side mirror of nanocaller_src/model_architect_indel.py: `Indel_model` on the HIP CNN (K9)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .model_architect import _Model


class Indel_model(_Model):
    """model_architect_indel.py:6-48.  x (B,15,128,2) = hstack of the three read-set tensors (indelCaller.py:83) -> (B,4)
    softmax over {hom-ref, hom-alt, het-ref, het-alt}."""
    KIND = _lib.MODEL_INDEL
    ROWS = 15

    def __call__(self, x):
        x = np.ascontiguousarray(x, np.float32)          # the reference hands float64 arrays of f32-rounded values
        if x.ndim != 4 or x.shape[1:] != (self.ROWS, 128, 2):
            raise ValueError("%s expects (B,%d,128,2), got %s" % (type(self).__name__, self.ROWS, x.shape))
        eng = self._engine()
        return eng.indel_forward(self.KIND, torch.from_numpy(x).to(eng.device)).cpu().numpy()
