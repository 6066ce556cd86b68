"""GPU: parity at the batch sizes bench.py times: BASELINE.json configs 4 / 5 as one GPU sees them -- B = 8192
utterances x 98 frames in ONE launch for ds_tcn_h256, mdtc_h64 and the 12-class mdtc_h64_global12 (default precision and the
one-product f16 mode of config 5) -- and the GRU at B = 16384 (the layer-major route).  Checked: >= 64 utterances spread over all
persistent rounds (first and last utterance of every 256-utterance round) against the oracle, every output finite, and the first
1024 utterances bit-identical to a call that carries only those (an utterance's result must not depend on its batch)."""
import numpy as np
import pytest
import torch

from oracle import kws_oracle
from tests.test_hip_parity import build
from wekws_amd import pack
from wekws_amd.utils import synth

pytestmark = pytest.mark.gpu


def spots(B, step=256):
    idx = sorted({i for k in range(0, B, step) for i in (k, min(B - 1, k + step - 1))})
    return idx


@pytest.mark.parametrize("name,precision,tol", [("ds_tcn_h256", "default", 1e-4), ("ds_tcn_h256", "f32", 1e-4),
                                                ("mdtc_h64", "default", 1e-4), ("mdtc_h64_global12", "default", 1e-4),
                                                ("mdtc_h64_global12", "f16", 1e-2), ("mdtc_h64", "f16", 1e-2)])
def test_b8192_one_launch_against_the_oracle(name, precision, tol):
    B, T = 8192, 98
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    model = build(cfg, sd).set_precision(precision)
    x = synth.synth_feats(B, T, cfg["input_dim"], seed=11)
    xt = torch.from_numpy(x).cuda()
    y, cache = model(xt)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(cache).all())
    idx = spots(B)
    assert len(idx) >= 64
    ry, rc = kws_oracle.forward(cfg, sd, x[idx], None)
    yh, ch = y[idx].cpu().numpy(), cache[idx].cpu().numpy()
    assert float(np.abs(yh - ry).max()) <= tol * max(1.0, float(np.abs(ry).max()))
    assert float(np.abs(ch - rc).max()) <= tol * max(1.0, float(np.abs(rc).max()))
    # an utterance's result does not depend on the batch it travels in
    y1, c1 = model(xt[:1024].contiguous())
    assert torch.equal(y1, y[:1024]) and torch.equal(c1, cache[:1024])
    y2, c2 = model(xt[5000:5003].contiguous())
    assert torch.equal(y2, y[5000:5003]) and torch.equal(c2, cache[5000:5003])


@pytest.mark.parametrize("precision", ["default", "f32"])
def test_gru_b16384_layer_major_route(precision):
    B, T = 16384, 40
    cfg = dict(synth.MODEL_CONFIGS["gru_2x128"])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    model = build(cfg, sd).set_precision(precision)
    x = synth.synth_feats(B, T, cfg["input_dim"], seed=12)
    h0 = (0.5 * np.random.default_rng(3).standard_normal((2, B, 128))).astype(np.float32)
    xt, ht = torch.from_numpy(x).cuda(), torch.from_numpy(h0).cuda()
    y, hn = model(xt, ht)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(hn).all())
    idx = spots(B, 512)
    ry, rh = kws_oracle.forward(cfg, sd, x[idx], h0[:, idx])
    assert float(np.abs(y[idx].cpu().numpy() - ry).max()) <= 1e-4
    assert float(np.abs(hn[:, idx].cpu().numpy() - rh).max()) <= 1e-4
    y1, h1 = model(xt[:256].contiguous(), ht[:, :256].contiguous())
    assert float((y1 - y[:256]).abs().max()) <= 2e-6 and float((h1 - hn[:, :256]).abs().max()) <= 2e-6   # (another kernel family at B = 256)
