"""Host-side mirror of nanocaller_src/model_architect_SNP_haploid.py: `haploid_SNP_model` on the HIP CNN."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .model_architect import _Model, _ref_code


class _Probs(np.ndarray):
    """the reference reads the result through `.numpy()` (snpCaller.py:183)"""

    def numpy(self):
        return np.asarray(self)


class haploid_SNP_model(_Model):
    """model_architect_SNP_haploid.py:7-53.  inputs = [x (B,5,41,5), ref (B,4)] -> (B,4) softmax of SELU(fc3)."""
    KIND = _lib.MODEL_SNP_HAP

    def __call__(self, inputs):
        x, ref = inputs
        if self._w is None and len(x) == 1 and not np.any(x):
            return None                            # the reference builds the model by calling it on zeros (snpCaller.py:76-77)
        eng = self._engine()
        dev = eng.device
        xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
        probs, _ = eng.snp_forward(self.KIND, xd, torch.from_numpy(_ref_code(ref)).to(dev), None)
        return probs.cpu().numpy().view(_Probs)
