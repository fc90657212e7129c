"""Host-side mirror of the reference's model classes (nanocaller_src/model_architect*.py).

The four classes keep their names and call conventions (callable on numpy batches, `load_weights`), but
the forward pass is the HIP CNN of libnanocaller_hip.so (nc_snp_forward / nc_indel_forward); weights come
from the converted `.ncw` files (nanocaller_amd/weights.py).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .engine import get_engine
from .weights import Weights


class _Model:
    KIND = None

    def __init__(self, device=0):
        self._device = device
        self._w = None

    def load_weights(self, path):
        """path: an .ncw file (the reference passes a TF checkpoint prefix / .h5, snpCaller.py:71,78)."""
        self._w = Weights(path)
        get_engine(self._device).load_weights(self.KIND, self._w)
        return self

    def expect_partial(self):                     # keras idiom used at snpCaller.py:71
        return self

    def _engine(self):
        if self._w is None:
            raise RuntimeError("%s: call load_weights() first" % type(self).__name__)
        eng = get_engine(self._device)
        eng.use_torch_stream()
        eng.load_weights(self.KIND, self._w)
        return eng


def _ref_code(onehot):
    oh = np.asarray(onehot, np.float32)
    if not np.all((oh == 0) | (oh == 1)) or np.any(oh.sum(1) > 1):
        raise ValueError("reference columns must be a 0/1 one-hot (snpCaller.py:90,111)")
    return np.where(oh.sum(1) == 1, np.argmax(oh, 1), -1).astype(np.int32)


class SNP_model(_Model):
    """model_architect.py:6-64.  inputs = [x (B,5,41,5), A_ref, G_ref, T_ref, C_ref (B,1) each];
    returns out_A, out_G, out_T, out_C, out_GT as (B,2) float32 arrays."""
    KIND = _lib.MODEL_SNP

    def __call__(self, inputs):
        x, a, g, t, c = inputs
        eng = self._engine()
        ref = _ref_code(np.hstack([np.asarray(a), np.asarray(g), np.asarray(t), np.asarray(c)]))
        dev = eng.device
        xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
        probs, gt = eng.snp_forward(self.KIND, xd, torch.from_numpy(ref).to(dev), None)
        p = probs.cpu().numpy()
        outs = [np.stack([1.0 - p[:, k], p[:, k]], axis=1).astype(np.float32) for k in range(4)]
        return outs + [gt.cpu().numpy()]


class haploid_SNP_model(_Model):
    """model_architect_SNP_haploid.py:7-53.  inputs = [x (B,5,41,5), ref (B,4)] -> (B,4) softmax."""
    KIND = _lib.MODEL_SNP_HAP

    def __call__(self, inputs):
        x, ref = inputs
        eng = self._engine()
        dev = eng.device
        xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
        probs, _ = eng.snp_forward(self.KIND, xd, torch.from_numpy(_ref_code(ref)).to(dev), None)
        return probs.cpu().numpy()


class Indel_model(_Model):
    """model_architect_indel.py:6-48.  x (B,15,128,2) -> (B,4) softmax."""
    KIND = _lib.MODEL_INDEL

    def __call__(self, x):
        eng = self._engine()
        xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(eng.device)
        return eng.indel_forward(self.KIND, xd).cpu().numpy()


class haploid_Indel_model(Indel_model):
    """model_architect_indels_haploid.py:7-48.  x (B,5,128,2) -> (B,1) sigmoid."""
    KIND = _lib.MODEL_INDEL_HAP
