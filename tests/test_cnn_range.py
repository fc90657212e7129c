"""Range guard of the split-precision (fp16x3) SNP trunk: its epilogues clamp activations to the fp16 range, and a
clamp that fired would be a wrong probability with status 0.  nc_load_weights derives x_limit from the L1 norms of the model's
convolutions; sites whose scaled tensor exceeds it are flagged by the kernel and re-run on the exact fp32 MFMA trunk.  Adversarial but
legal inputs -- 160 reads of one base, per-site scaling with dp = 4 (x 14), random +-160 tensors -- must come back within the 1e-4
contract of the float64 oracle for EVERY shipped model, flagged sites bit-identical to the exact kernel."""
import numpy as np
import pytest
import torch

from nanocaller_amd import _lib
from nanocaller_amd.weights import SNP_MODEL_FILES, Weights, get_SNP_model

MODELS = sorted(set(SNP_MODEL_FILES) - {"NanoCaller2"})


def _tensors(rng, n, hi):
    """integer-valued site tensors with the structure of SURVEY Appendix A (row 0 one-hot, channel 4 flags) and counts up to `hi`"""
    x = np.zeros((n, 5, 41, 5), np.float32)
    ref = rng.integers(0, 4, size=(n, 41))
    x[np.arange(n)[:, None], 0, np.arange(41)[None, :], ref] = 1
    x[:, 1:, :, :4] = rng.integers(-hi, hi + 1, size=(n, 4, 41, 4))
    x[:, 1:, :, 4] = rng.integers(0, 2, size=(n, 4, 41))
    k = n // 4                                                     # a quarter: every read carries the same base everywhere (one saturated plane)
    b = rng.integers(0, 4, size=k)
    x[:k, 1:, :, :4] = 0
    x[np.arange(k), 1 + b, :, b] = hi
    return x, ref[:, 20].astype(np.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
def test_guarded_split_precision_equals_the_oracle_on_adversarial_tensors(model):
    from nanocaller_amd.engine import get_engine
    from oracle import oracle
    eng = get_engine(0)
    path, cov = get_SNP_model(model)
    w = Weights(path)
    kind = w.kind
    eng.load_weights(kind, w)
    xl = eng.x_limit(kind)
    assert 20.0 < xl < 1e5
    rng = np.random.default_rng(17)
    cases = [("benign 30x, chunk scale", 30, np.full(96, cov / 30.0), 0), ("160 reads, chunk scale 1.6", 160, np.full(96, 1.6), 0),
             ("160 reads, per-site scale train_cov / 4", 160, np.full(96, cov / 4.0), 1)]
    for name, hi, scale, mode in cases:
        x, rc = _tensors(rng, 96, hi)
        for fmt16 in (False, True):
            eng.set_cnn_precision(False)
            eng.set_tensor_format(fmt16)
            xd = torch.from_numpy(x.astype(np.int16) if fmt16 else x).to(eng.device)
            rcd, sd = torch.from_numpy(rc).to(eng.device), torch.from_numpy(scale.astype(np.float64)).to(eng.device)
            pg, gg, n_rerun = eng.snp_forward_guarded(kind, xd, rcd, sd, mode)
            eng.set_tensor_format(False)
            eng.set_cnn_precision(True)
            pe, ge = eng.snp_forward(kind, torch.from_numpy(x).to(eng.device), rcd, sd, mode)
            eng.set_cnn_precision(False)
            if kind == _lib.MODEL_SNP:
                po, go = oracle.snp_forward(w.flat, x, rc, scale, scale_mode=mode, precision="f64")
            else:
                po = oracle.snp_hap_forward(w.flat, x, rc, scale, scale_mode=mode, precision="f64")
            pg_h, pe_h = pg.cpu().numpy(), pe.cpu().numpy()
            amax = float(np.abs(x[:, 1:, :, :4] * scale[:, None, None, None]).max())
            assert np.abs(pe_h - po).max() < 1e-4, (model, name)
            assert np.abs(pg_h - po).max() < 1e-4, (model, name, fmt16, n_rerun)
            if amax > xl:
                assert n_rerun > 0, (model, name, amax, xl)
            if hi == 30:
                assert n_rerun == 0, (model, name)                    # ordinary data never leaves the proven range
            if n_rerun == 96:
                assert np.array_equal(pg_h, pe_h)                     # re-run sites ARE the exact kernel's


@pytest.mark.gpu
def test_indel_models_are_proven_in_range_for_frequency_tensors():
    from nanocaller_amd.engine import get_engine
    from nanocaller_amd.weights import INDEL_MODEL_FILES, get_indel_model
    eng = get_engine(0)
    for m in sorted(INDEL_MODEL_FILES):
        w = Weights(get_indel_model(m))
        eng.load_weights(w.kind, w)
        assert eng.x_limit(w.kind) >= 4.0, m                          # msa() tensors are frequencies: |x| <= 1 (conv1 / conv2 outputs are the clamped ones)
