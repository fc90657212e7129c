"""CPU-only: the CNN restatement in oracle/nc_oracle.c (f32 and f64) against an INDEPENDENT torch-CPU
implementation of the Keras models (TensorFlow is absent, so CNN-vs-TF parity is unpinned; two
independent in-repo readings of the layouts must agree), fed by the real converted weights."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from nanocaller_amd import weights
from oracle import oracle
from tests.util import load_snp_case

SELU = torch.nn.functional.selu


def _conv(x, k, b, stride=(1, 1), same=False):
    # Keras Conv2D: NHWC input, HWIO kernel, cross-correlation; here x is NCHW float64
    w = torch.from_numpy(k.astype(np.float64)).permute(3, 2, 0, 1)
    pad = (k.shape[0] // 2, k.shape[1] // 2) if same else (0, 0)
    return F.conv2d(x, w, torch.from_numpy(b.astype(np.float64)), stride=stride, padding=pad)


def _trunk(t, x):
    c = torch.cat([SELU(_conv(x, t["conv1_1.k"], t["conv1_1.b"], same=True)),
                   SELU(_conv(x, t["conv1_2.k"], t["conv1_2.b"], same=True)),
                   SELU(_conv(x, t["conv1_3.k"], t["conv1_3.b"], same=True))], 1)
    c = SELU(_conv(c, t["conv2.k"], t["conv2.b"], stride=(1, 2)))
    c = SELU(_conv(c, t["conv3.k"], t["conv3.b"], stride=(1, 2)))
    flat = c.permute(0, 2, 3, 1).reshape(c.shape[0], -1)            # Keras Flatten of NHWC
    return SELU(flat @ torch.from_numpy(t["fc1.k"].astype(np.float64)) + torch.from_numpy(t["fc1.b"].astype(np.float64)))


def _dense(x, t, name, act=False):
    y = x @ torch.from_numpy(t[name + ".k"].astype(np.float64)) + torch.from_numpy(t[name + ".b"].astype(np.float64))
    return SELU(y) if act else y


def torch_snp_model(w, x, ref_code, scale):
    xs = x.astype(np.float32).copy()
    xs[:, 1:, :, :4] = xs[:, 1:, :, :4] * np.float32(scale)          # numpy<2 scalar semantics (snpCaller.py:96)
    xt = torch.from_numpy(xs.astype(np.float64)).permute(0, 3, 1, 2)
    fc1 = _trunk(w.t, xt)
    fa = _dense(fc1, w.t, "fa", True)
    onehot = torch.from_numpy(np.eye(4)[ref_code])
    heads = [torch.softmax(_dense(torch.cat([fa, onehot[:, i:i + 1]], 1), w.t, n), 1) for i, n in enumerate("AGTC")]
    fc2 = _dense(fc1, w.t, "fc2", True)
    fc3 = _dense(torch.cat([fc2] + heads, 1), w.t, "fc3", True)
    gt = torch.softmax(_dense(fc3, w.t, "GT"), 1)
    return torch.stack([h[:, 1] for h in heads], 1).numpy(), gt.numpy()


@pytest.mark.parametrize("model,case", [("ONT-HG002", "ont_dip"), ("CCS-HG002", "hifi_pacbio_dip")])
def test_snp_model_two_independent_implementations_agree(model, case):
    path, cov = weights.get_SNP_model(model)
    w = weights.Weights(path)
    gold = load_snp_case(case)[4]
    x = gold["mat"][:96]
    rc = np.argmax(gold["ref"][:96], 1).astype(np.int32)
    s = cov / gold["depth"]
    tp, tg = torch_snp_model(w, x, rc, s)
    p64, g64 = oracle.snp_forward(w.flat, x, rc, s, precision="f64")
    p32, g32 = oracle.snp_forward(w.flat, x, rc, s, precision="f32")
    assert np.abs(tp - p64).max() < 2e-6 and np.abs(tg - g64).max() < 2e-6
    assert np.abs(p32 - p64).max() < 2e-5 and np.abs(g32 - g64).max() < 2e-5
    assert 0.05 < (p64 >= 0.5).mean() < 0.95              # the sample exercises both sides of the decision


def test_haploid_and_indel_models_agree_with_torch():
    path, _ = weights.get_SNP_model("haploid")
    w = weights.Weights(path)
    gold = load_snp_case("ont_hap")[4]
    x = gold["mat"][:64]
    rc = np.argmax(gold["ref"][:64], 1).astype(np.int32)
    s = 30.0 / gold["depth"]
    xs = x.copy()
    xs[:, 1:, :, :4] = xs[:, 1:, :, :4] * np.float32(s)
    fc1 = _trunk(w.t, torch.from_numpy(xs.astype(np.float64)).permute(0, 3, 1, 2))
    fc2 = _dense(fc1, w.t, "fc2", True)
    out = torch.softmax(_dense(torch.cat([fc2, torch.from_numpy(np.eye(4)[rc])], 1), w.t, "fc3", True), 1).numpy()
    assert np.abs(out - oracle.snp_hap_forward(w.flat, x, rc, s, precision="f64")).max() < 2e-6
    rng = np.random.Generator(np.random.PCG64(5))
    for name, rows in (("ONT-HG002", 15), ("haploid", 5)):
        w = weights.Weights(weights.get_indel_model(name))
        x = (rng.random((4, rows, 128, 2)) * (rng.random((4, rows, 128, 2)) < 0.3)).astype(np.float32)
        fc1 = _trunk(w.t, torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2))
        y = _dense(_dense(fc1, w.t, "fc2", True), w.t, "fc3")
        exp = torch.softmax(y, 1).numpy() if rows == 15 else torch.sigmoid(y).numpy()
        assert np.abs(exp - oracle.indel_forward(w.flat, x, precision="f64")).max() < 2e-6


def test_oracle_scale_modes_differ_only_by_rounding():
    path, cov = weights.get_SNP_model("ONT-HG002")
    w = weights.Weights(path)
    gold = load_snp_case("ont_dip")[4]
    x, rc = gold["mat"][:32], np.argmax(gold["ref"][:32], 1).astype(np.int32)
    a, _ = oracle.snp_forward(w.flat, x, rc, cov / gold["depth"], scale_mode=0, precision="f64")
    b, _ = oracle.snp_forward(w.flat, x, rc, cov / gold["depth"], scale_mode=1, precision="f64")
    assert np.abs(a - b).max() < 1e-5
