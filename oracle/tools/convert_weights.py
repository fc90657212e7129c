#!/usr/bin/env python3
"""Convert the reference's model zoo (DATA, not code) into flat `.ncw` weight files.

Runs ONLY in the build container (reads /root/reference/nanocaller_src/release_data).
TensorFlow is absent, so the TF checkpoint bundle is parsed directly:
`<prefix>.index` is an uncompressed LevelDB-format table whose values are BundleEntryProto
{1:dtype 2:shape 4:offset 5:size}; `<prefix>.data-00000-of-00001` holds raw little-endian f32.
The two Keras `.h5` haploid models are read with /opt/conda/bin/h5dump.

.ncw layout (little endian):
  magic  b"NCW1"  | u32 kind (0 snp, 1 snp_hap, 2 indel, 3 indel_hap) | f32 train_coverage
  u32 n_tensors | n_tensors x { char name[24]; u32 ndim; u32 dims[4]; u64 offset_floats }
  f32 data[...]   in the canonical layer order of nanocaller_amd/weights.py (Keras HWIO /
  [in,out] layouts unchanged).
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from nanocaller_amd.weights import (KIND_INDEL, KIND_INDEL_HAP, KIND_SNP, KIND_SNP_HAP,  # noqa: E402
                                    LAYER_SPECS, write_ncw)

REF = "/root/reference/nanocaller_src"


# ------------------------------------------------------------------ varint / protobuf
def _varint(buf, p):
    r = 0
    s = 0
    while True:
        b = buf[p]
        p += 1
        r |= (b & 0x7F) << s
        if not b & 0x80:
            return r, p
        s += 7


def _parse_proto(buf):
    """-> list of (field, wiretype, value)"""
    out = []
    p = 0
    while p < len(buf):
        key, p = _varint(buf, p)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, p = _varint(buf, p)
        elif wt == 1:
            v = buf[p:p + 8]
            p += 8
        elif wt == 2:
            n, p = _varint(buf, p)
            v = buf[p:p + n]
            p += n
        elif wt == 5:
            v = buf[p:p + 4]
            p += 4
        else:
            raise ValueError("wiretype %d" % wt)
        out.append((f, wt, v))
    return out


# ------------------------------------------------------------------ leveldb table
def _read_block(data, off, size):
    blk = data[off:off + size]
    assert data[off + size] == 0, "compressed block not supported"
    nrestart = struct.unpack("<I", blk[-4:])[0]
    end = len(blk) - 4 - 4 * nrestart
    p = 0
    key = b""
    ents = []
    while p < end:
        shared, p = _varint(blk, p)
        nonshared, p = _varint(blk, p)
        vlen, p = _varint(blk, p)
        key = key[:shared] + blk[p:p + nonshared]
        p += nonshared
        ents.append((key, blk[p:p + vlen]))
        p += vlen
    return ents


def read_tf_checkpoint(prefix):
    idx = open(prefix + ".index", "rb").read()
    dat = open(prefix + ".data-00000-of-00001", "rb").read()
    footer = idx[-48:]
    assert footer[-8:] == bytes.fromhex("57fb808b247547db"), "bad table magic"
    p = 0
    _, p = _varint(footer, p)
    _, p = _varint(footer, p)
    ioff, p = _varint(footer, p)
    isz, p = _varint(footer, p)
    tensors = {}
    for _, handle in _read_block(idx, ioff, isz):
        boff, q = _varint(handle, 0)
        bsz, q = _varint(handle, q)
        for key, val in _read_block(idx, boff, bsz):
            if key == b"" or key.startswith(b"_CHECKPOINTABLE"):
                continue
            dtype, shape, offset, size = 0, [], 0, 0
            for f, wt, v in _parse_proto(val):
                if f == 1:
                    dtype = v
                elif f == 2:
                    for f2, _, v2 in _parse_proto(v):
                        if f2 == 2:
                            d = 0
                            for f3, _, v3 in _parse_proto(v2):
                                if f3 == 1:
                                    d = v3
                            shape.append(d)
                elif f == 4:
                    offset = v
                elif f == 5:
                    size = v
            if dtype != 1:
                continue
            arr = np.frombuffer(dat, dtype="<f4", count=size // 4, offset=offset).reshape(shape)
            tensors[key.decode()] = arr.copy()
    return tensors


def tf_layers(prefix):
    t = read_tf_checkpoint(prefix)
    out = {}
    for k, v in t.items():
        parts = k.split("/")
        if len(parts) >= 2 and parts[1] in ("kernel", "bias") and parts[2] == ".ATTRIBUTES":
            out[(parts[0], parts[1])] = v
    return out


def h5_dataset(path, dset, count):
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.run(["/opt/conda/bin/h5dump", "-d", dset, "-b", "LE", "-o", tmp.name, path],
                       check=True, stdout=subprocess.DEVNULL)
        a = np.fromfile(tmp.name, dtype="<f4")
    assert a.size == count, (dset, a.size, count)
    return a


SNP_MODELS = {  # name -> relative checkpoint prefix (table of snpCaller.py:16-34)
    'NanoCaller1': 'release_data/ONT_models/SNPs/NanoCaller1_beta/model-rt-1',
    'NanoCaller2': 'release_data/ONT_models/SNPs/NanoCaller1_beta/model-rt-1',
    'NanoCaller3': 'release_data/clr_models/SNPs/NanoCaller3_beta/model-rt-100',
    'ONT-HG001': 'release_data/ONT_models/SNPs/HG001_guppy4.2.2_giab-3.3.2/model-1',
    'ONT-HG001_GP2.3.8': 'release_data/ONT_models/SNPs/HG001_guppy2.3.8_giab-3.3.2/model-100',
    'ONT-HG001_GP2.3.8-4.2.2': 'release_data/ONT_models/SNPs/HG001_guppy2.3.8_guppy4.2.2_giab-3.3.2/model-100',
    'ONT-HG001-4_GP4.2.2': 'release_data/ONT_models/SNPs/HG001_guppy4.2.2_giab-3.3.2_HG002-4_guppy4.2.2_giab-4.2.1/model-100',
    'ONT-HG002': 'release_data/ONT_models/SNPs/HG002_guppy4.2.2_giab-4.2.1/model-100',
    'ONT-HG002_GP4.2.2_v3.3.2': 'release_data/ONT_models/SNPs/HG002_guppy4.2.2_giab-3.3.2/model-100',
    'ONT-HG002_GP2.3.4_v3.3.2': 'release_data/ONT_models/SNPs/HG002_guppy2.3.4_giab-3.3.2/model-100',
    'ONT-HG002_GP2.3.4_v4.2.1': 'release_data/ONT_models/SNPs/HG002_guppy2.3.4_giab-4.2.1/model-100',
    'ONT-HG002_r10.3': 'release_data/ONT_models/SNPs/HG002_r10.3_guppy4.0.11_giab-4.2.1/model-100',
    'ONT-HG002_bonito': 'release_data/ONT_models/SNPs/HG002_bonito_giab-4.2.1/model-100',
    'CCS-HG001': 'release_data/hifi_models/SNPs/HG001_giab-3.3.2/model-100',
    'CCS-HG002': 'release_data/hifi_models/SNPs/HG002_giab-4.2.1/model-100',
    'CCS-HG001-4': 'release_data/hifi_models/SNPs/HG001_giab-3.3.2_HG002-4_giab-4.2.1/model-100',
    'CLR-HG002': 'release_data/clr_models/SNPs/HG002_giab-4.2.1/model-100',
}
INDEL_MODELS = {  # indelCaller.py:17-24
    'NanoCaller1': 'release_data/ONT_models/indels/NanoCaller1_beta/model-30',
    'NanoCaller3': 'release_data/hifi_models/indels/NanoCaller3_beta/model-25',
    'ONT-HG001': 'release_data/ONT_models/indels/HG001_guppy4.2_giab-3.3.2/model-100',
    'ONT-HG002': 'release_data/ONT_models/indels/HG002_guppy4.2_giab-4.2.1/model-100',
    'CCS-HG001': 'release_data/hifi_models/indels/HG001_giab-3.3.2/model-100',
    'CCS-HG002': 'release_data/hifi_models/indels/HG002_giab-4.2.1/model-100',
}
# Keras-H5 group names of the haploid models (Appendix C): attribute -> H5 layer name
H5_SNP = {"conv1_1": "C1_1", "conv1_2": "C1_2", "conv1_3": "C1_3", "conv2": "C2", "conv3": "C3",
          "fc1": "C4", "fc2": "C6", "fc3": "C7"}
H5_INDEL = {"conv1_1": "C1_1", "conv1_2": "C1_2", "conv1_3": "C1_3", "conv2": "C2", "conv3": "C3",
            "fc1": "C4", "fc2": "C5", "fc3": "C6"}


def convert_tf(prefix, kind, out_path):
    lay = tf_layers(os.path.join(REF, prefix))
    tensors = []
    for name, kshape in LAYER_SPECS[kind]:
        k = lay[(name, "kernel")]
        b = lay[(name, "bias")]
        assert tuple(k.shape) == tuple(kshape), (name, k.shape, kshape)
        assert b.shape == (kshape[-1],)
        tensors.append((name + ".k", k))
        tensors.append((name + ".b", b))
    cov = 0.0
    cp = os.path.join(REF, prefix + ".coverage")
    if os.path.exists(cp):
        cov = float(open(cp).readline().strip())
    write_ncw(out_path, kind, cov, tensors)
    return sum(t.size for _, t in tensors), cov


def convert_h5(path, kind, names, scope, out_path, cov):
    tensors = []
    for name, kshape in LAYER_SPECS[kind]:
        h5n = names[name]
        base = "/%s/%s%s" % (h5n, scope, h5n)
        k = h5_dataset(path, base + "/kernel:0", int(np.prod(kshape))).reshape(kshape)
        b = h5_dataset(path, base + "/bias:0", kshape[-1])
        tensors.append((name + ".k", k))
        tensors.append((name + ".b", b))
    write_ncw(out_path, kind, cov, tensors)
    return sum(t.size for _, t in tensors)


def main():
    out_dir = os.path.join(os.path.dirname(__file__), "..", "..", "nanocaller_amd", "weights")
    os.makedirs(out_dir, exist_ok=True)
    done = {}
    for name, prefix in SNP_MODELS.items():
        if prefix in done:
            continue
        fn = "snp__" + prefix.split("/")[-2] + "__" + prefix.split("/")[-1] + "__" + prefix.split("/")[1] + ".ncw"
        n, cov = convert_tf(prefix, KIND_SNP, os.path.join(out_dir, fn))
        done[prefix] = fn
        print("%-28s %s floats=%d cov=%g" % (name, fn, n, cov))
    for name, prefix in INDEL_MODELS.items():
        fn = "indel__" + prefix.split("/")[-2] + "__" + prefix.split("/")[-1] + "__" + prefix.split("/")[1] + ".ncw"
        n, _ = convert_tf(prefix, KIND_INDEL, os.path.join(out_dir, fn))
        print("%-28s %s floats=%d" % (name, fn, n))
    n = convert_h5(os.path.join(REF, "release_data/haploid_models/SNPs/CHM13/model.24-0.9985.h5"),
                   KIND_SNP_HAP, H5_SNP, "snp_model/", os.path.join(out_dir, "snp_hap__CHM13.ncw"), 30.0)
    print("haploid SNP floats=%d" % n)
    n = convert_h5(os.path.join(REF, "release_data/haploid_models/indels/CHM13/model.19-0.9811.h5"),
                   KIND_INDEL_HAP, H5_INDEL, "", os.path.join(out_dir, "indel_hap__CHM13.ncw"), 0.0)
    print("haploid indel floats=%d" % n)


if __name__ == "__main__":
    main()
