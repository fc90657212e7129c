"""Flat `.ncw` weight files: the reference model zoo as data, readable without TensorFlow.

Layer order / shapes restate the Keras models:
model_architect.py:6-32 (SNP_model), model_architect_SNP_haploid.py:7-29 (haploid_SNP_model),
model_architect_indel.py:6-24 (Indel_model), model_architect_indels_haploid.py:7-25.
Kernels keep the Keras layouts: Conv2D HWIO [kh,kw,Cin,Cout], Dense [in,out].
Model-name tables mirror snpCaller.py:16-34 and indelCaller.py:17-24.
"""
from __future__ import annotations

import os
import struct

import numpy as np

KIND_SNP, KIND_SNP_HAP, KIND_INDEL, KIND_INDEL_HAP = 0, 1, 2, 3

LAYER_SPECS = {
    KIND_SNP: [
        ("conv1_1", (1, 5, 5, 16)), ("conv1_2", (5, 1, 5, 16)), ("conv1_3", (5, 5, 5, 16)),
        ("conv2", (2, 3, 48, 32)), ("conv3", (2, 3, 32, 64)),
        ("fc1", (1728, 48)), ("fa", (48, 16)),
        ("A", (17, 2)), ("G", (17, 2)), ("T", (17, 2)), ("C", (17, 2)),
        ("fc2", (48, 16)), ("fc3", (24, 8)), ("GT", (8, 2)),
    ],
    KIND_SNP_HAP: [
        ("conv1_1", (1, 5, 5, 16)), ("conv1_2", (5, 1, 5, 16)), ("conv1_3", (5, 5, 5, 16)),
        ("conv2", (2, 3, 48, 32)), ("conv3", (2, 3, 32, 64)),
        ("fc1", (1728, 48)), ("fc2", (48, 16)), ("fc3", (20, 4)),
    ],
    KIND_INDEL: [
        ("conv1_1", (1, 5, 2, 8)), ("conv1_2", (5, 1, 2, 8)), ("conv1_3", (5, 5, 2, 8)),
        ("conv2", (2, 3, 24, 32)), ("conv3", (2, 3, 32, 48)),
        ("fc1", (19344, 32)), ("fc2", (32, 24)), ("fc3", (24, 4)),
    ],
    KIND_INDEL_HAP: [
        ("conv1_1", (1, 5, 2, 8)), ("conv1_2", (5, 1, 2, 8)), ("conv1_3", (5, 5, 2, 8)),
        ("conv2", (2, 3, 24, 32)), ("conv3", (2, 3, 32, 48)),
        ("fc1", (4464, 32)), ("fc2", (32, 24)), ("fc3", (24, 1)),
    ],
}


def n_params(kind: int) -> int:
    return sum(int(np.prod(s)) + s[-1] for _, s in LAYER_SPECS[kind])


assert n_params(KIND_SNP) == 109_370 and n_params(KIND_INDEL) == 634_420
assert n_params(KIND_SNP_HAP) == 108_308 and n_params(KIND_INDEL_HAP) == 158_185

WEIGHT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights")

# model name -> .ncw file (same names the reference accepts; note NanoCaller2 -> NanoCaller1_beta,
# snpCaller.py:17)
SNP_MODEL_FILES = {
    'NanoCaller1': 'snp__NanoCaller1_beta__model-rt-1__ONT_models.ncw',
    'NanoCaller2': 'snp__NanoCaller1_beta__model-rt-1__ONT_models.ncw',
    'NanoCaller3': 'snp__NanoCaller3_beta__model-rt-100__clr_models.ncw',
    'ONT-HG001': 'snp__HG001_guppy4.2.2_giab-3.3.2__model-1__ONT_models.ncw',
    'ONT-HG001_GP2.3.8': 'snp__HG001_guppy2.3.8_giab-3.3.2__model-100__ONT_models.ncw',
    'ONT-HG001_GP2.3.8-4.2.2': 'snp__HG001_guppy2.3.8_guppy4.2.2_giab-3.3.2__model-100__ONT_models.ncw',
    'ONT-HG001-4_GP4.2.2': 'snp__HG001_guppy4.2.2_giab-3.3.2_HG002-4_guppy4.2.2_giab-4.2.1__model-100__ONT_models.ncw',
    'ONT-HG002': 'snp__HG002_guppy4.2.2_giab-4.2.1__model-100__ONT_models.ncw',
    'ONT-HG002_GP4.2.2_v3.3.2': 'snp__HG002_guppy4.2.2_giab-3.3.2__model-100__ONT_models.ncw',
    'ONT-HG002_GP2.3.4_v3.3.2': 'snp__HG002_guppy2.3.4_giab-3.3.2__model-100__ONT_models.ncw',
    'ONT-HG002_GP2.3.4_v4.2.1': 'snp__HG002_guppy2.3.4_giab-4.2.1__model-100__ONT_models.ncw',
    'ONT-HG002_r10.3': 'snp__HG002_r10.3_guppy4.0.11_giab-4.2.1__model-100__ONT_models.ncw',
    'ONT-HG002_bonito': 'snp__HG002_bonito_giab-4.2.1__model-100__ONT_models.ncw',
    'CCS-HG001': 'snp__HG001_giab-3.3.2__model-100__hifi_models.ncw',
    'CCS-HG002': 'snp__HG002_giab-4.2.1__model-100__hifi_models.ncw',
    'CCS-HG001-4': 'snp__HG001_giab-3.3.2_HG002-4_giab-4.2.1__model-100__hifi_models.ncw',
    'CLR-HG002': 'snp__HG002_giab-4.2.1__model-100__clr_models.ncw',
    'haploid': 'snp_hap__CHM13.ncw',
}
INDEL_MODEL_FILES = {
    'NanoCaller1': 'indel__NanoCaller1_beta__model-30__ONT_models.ncw',
    'NanoCaller3': 'indel__NanoCaller3_beta__model-25__hifi_models.ncw',
    'ONT-HG001': 'indel__HG001_guppy4.2_giab-3.3.2__model-100__ONT_models.ncw',
    'ONT-HG002': 'indel__HG002_guppy4.2_giab-4.2.1__model-100__ONT_models.ncw',
    'CCS-HG001': 'indel__HG001_giab-3.3.2__model-100__hifi_models.ncw',
    'CCS-HG002': 'indel__HG002_giab-4.2.1__model-100__hifi_models.ncw',
    'haploid': 'indel_hap__CHM13.ncw',
}


def write_ncw(path, kind, train_coverage, tensors):
    """tensors: list of (name, ndarray f32) in canonical order."""
    hdr = [b"NCW1", struct.pack("<IfI", kind, float(train_coverage), len(tensors))]
    off = 0
    for name, a in tensors:
        dims = list(a.shape) + [1] * (4 - a.ndim)
        hdr.append(struct.pack("<24sI4IQ", name.encode(), a.ndim, *dims, off))
        off += a.size
    with open(path, "wb") as f:
        f.write(b"".join(hdr))
        for _, a in tensors:
            f.write(np.ascontiguousarray(a, dtype="<f4").tobytes())


class Weights:
    """Parsed .ncw: `.kind`, `.train_coverage`, `.t[name]` arrays, `.flat` canonical blob."""

    def __init__(self, path):
        raw = open(path, "rb").read()
        if raw[:4] != b"NCW1":
            raise ValueError("%s: not an NCW1 file" % path)
        self.kind, self.train_coverage, n = struct.unpack_from("<IfI", raw, 4)
        p = 16
        ents = []
        for _ in range(n):
            name, ndim, d0, d1, d2, d3, off = struct.unpack_from("<24sI4IQ", raw, p)
            p += 24 + 4 + 16 + 8
            ents.append((name.rstrip(b"\0").decode(), (d0, d1, d2, d3)[:ndim], off))
        self.flat = np.frombuffer(raw, dtype="<f4", offset=p).copy()
        self.t = {}
        for name, shape, off in ents:
            self.t[name] = self.flat[off:off + int(np.prod(shape))].reshape(shape)
        if self.flat.size != n_params(self.kind):
            raise ValueError("%s: %d floats, expected %d" % (path, self.flat.size, n_params(self.kind)))
        self.path = path


def get_SNP_model(snp_model):
    """Mirror of snpCaller.get_SNP_model (snpCaller.py:36-55): -> (path, train_coverage) or (None, None).
    Only the built-in model names resolve (quirk E11: custom directories never worked upstream);
    a path to an .ncw file is accepted as an extension."""
    if snp_model in SNP_MODEL_FILES:
        path = os.path.join(WEIGHT_DIR, SNP_MODEL_FILES[snp_model])
    elif isinstance(snp_model, str) and snp_model.endswith(".ncw") and os.path.exists(snp_model):
        path = snp_model
    else:
        return None, None
    return path, float(Weights(path).train_coverage)


def get_indel_model(indel_model):
    """Mirror of indelCaller.get_indel_model (indelCaller.py:26-39)."""
    if indel_model in INDEL_MODEL_FILES:
        return os.path.join(WEIGHT_DIR, INDEL_MODEL_FILES[indel_model])
    if isinstance(indel_model, str) and indel_model.endswith(".ncw") and os.path.exists(indel_model):
        return indel_model
    return None


def resolve_weight_file(path, kind):
    """What the model classes' load_weights() accepts: an .ncw file, or a path in the reference's layout -- a TF checkpoint
    prefix `.../release_data/<family>_models/<SNPs|indels>/<dir>/<model-N>` (snpCaller.py:16-34, indelCaller.py:17-24) or a
    haploid Keras `.h5` -- which is mapped to the converted file of the same model (the checkpoint itself is not read)."""
    if path.endswith(".ncw"):
        return path
    if path.endswith(".h5"):
        return os.path.join(WEIGHT_DIR, "snp_hap__CHM13.ncw" if kind in (KIND_SNP, KIND_SNP_HAP) else "indel_hap__CHM13.ncw")
    parts = os.path.normpath(path).split(os.sep)
    if len(parts) >= 4:
        stem = "snp" if kind in (KIND_SNP, KIND_SNP_HAP) else "indel"
        cand = os.path.join(WEIGHT_DIR, "%s__%s__%s__%s.ncw" % (stem, parts[-2], parts[-1], parts[-4]))
        if os.path.exists(cand):
            return cand
    raise FileNotFoundError("no converted weights for %r (expected an .ncw file or a release_data path)" % path)
