"""Host-side mirror of the reference's nanocaller_src/model_architect.py: `SNP_model`.

The class keeps its name and call convention (callable on numpy batches, `load_weights(...).expect_partial()`), but the
forward pass is the HIP CNN of libnanocaller_hip.so (nc_snp_forward); weights come from the converted `.ncw` files
(nanocaller_amd/weights.py).  The three other model classes live in the modules the reference imports them from
(snpCaller.py:6-8, indelCaller.py:6-9): model_architect_SNP_haploid, model_architect_indel, model_architect_indels_haploid.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .engine import get_engine
from .weights import Weights, resolve_weight_file


class _Model:
    KIND = None

    def __init__(self, device=0):
        self._device = device
        self._w = None

    def load_weights(self, path):
        """path: an .ncw file, or the path the reference passes (TF checkpoint prefix / .h5, snpCaller.py:71,78) which is
        mapped to the converted file of that model"""
        self._w = Weights(resolve_weight_file(path, self.KIND))
        get_engine(self._device).load_weights(self.KIND, self._w)
        return self

    def expect_partial(self):                     # keras idiom used at snpCaller.py:71
        return self

    def build(self, input_shape=None):            # indelCaller.py:56
        return None

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


def __getattr__(name):
    # the three other classes were importable from here in round 1
    if name == "haploid_SNP_model":
        from .model_architect_SNP_haploid import haploid_SNP_model
        return haploid_SNP_model
    if name == "Indel_model":
        from .model_architect_indel import Indel_model
        return Indel_model
    if name == "haploid_Indel_model":
        from .model_architect_indels_haploid import haploid_Indel_model
        return haploid_Indel_model
    raise AttributeError(name)
