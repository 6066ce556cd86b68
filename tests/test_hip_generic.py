"""GPU parity tests of the ANY-SHAPE path (wekws_amd/csrc/generic.hip.h): configurations the reference's init_model accepts
(wekws/model/kws_model.py:114-170 takes any hidden_dim / kernel_size / num_layers / classifier) and no specialised kernel is
built for.  Before round 5 the library refused them (WEKWS_HIP_EUNSUPPORTED); now they run in exact f32.

  (a) goldens recorded from the LIVE reference (tests/golden/make_generic_golden.py, 13 cases): one-shot and in two chunks;
  (b) the numpy oracle at other seeds / batch sizes, streamed in ragged chunks (incl. chunks shorter than a block's padding);
  (c) FSMN with precision F32 -- served with the reference's own arithmetic instead of the block-floating kernel -- against
      the 13 FSMN goldens, and an FSMN beyond the kernel's limits (48 taps, widths that overflow the LDS tile);
  (d) the path is what runs: effective_precision() says f32 for these models whatever was requested.
Tolerance: north_star's 1e-4 on posteriors (relative to max(1, max|ref|) for logits / caches)."""
import os

import numpy as np
import pytest
import torch

from oracle import kws_oracle
from tests.golden.cases import GENERIC_CASES, shape_case_config
from tests.helpers import CASES, case_in_cache, case_input, case_weights, max_abs
from tests.test_hip_parity import build, run, tol_for
from wekws_amd.utils import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def generic_golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "generic_golden.npz"))


@pytest.mark.parametrize("case", GENERIC_CASES, ids=[c["name"] for c in GENERIC_CASES])
def test_any_shape_vs_live_reference_goldens(case, generic_golden, error_report):
    from wekws_amd import pack
    cfg = shape_case_config(case)
    sd = synth.synth_state_dict(pack.model_spec(cfg), case["wseed"])
    g, name = generic_golden, case["name"]
    assert abs(synth.checksum(sd) - float(g[name + "/wsum"])) <= 1e-6 * abs(float(g[name + "/wsum"]))
    model = build(cfg, sd)
    assert model.effective_precision() == "f32", "these shapes have no specialised kernel: the any-shape path must serve them"
    x = synth.synth_feats(case["B"], case["T"], cfg["input_dim"], seed=case["xseed"])
    y, c = run(model, x)
    gy, gc = g[name + "/y"], g[name + "/cache"]
    assert y.shape == gy.shape and c.shape == gc.shape, (y.shape, c.shape)
    error_report[f"generic/{name}/y"] = max_abs(y, gy)
    error_report[f"generic/{name}/cache_rel"] = max_abs(c, gc) / max(1.0, float(np.abs(gc).max()))
    assert max_abs(y, gy) <= tol_for(gy), max_abs(y, gy)
    assert max_abs(c, gc) <= tol_for(gc), max_abs(c, gc)
    if case.get("split"):
        t1 = case["split"]
        ys, cs = run(model, x, chunks=[t1, case["T"] - t1])
        assert max_abs(ys, g[name + "/y_stream"]) <= tol_for(gy)
        assert max_abs(cs, g[name + "/cache_stream"]) <= tol_for(gc)


@pytest.mark.parametrize("case", [c for c in GENERIC_CASES if c.get("split")], ids=[c["name"] for c in GENERIC_CASES if c.get("split")])
def test_any_shape_ragged_streaming_vs_oracle(case):
    """Other weights, another batch size, ragged chunks 1 / 3 / 10 / 7 / rest (chunks shorter than the blocks' paddings: the
    returned cache is then [old cache tail | new frames]) against the oracle's streaming forward, which the CPU suite pins to
    the live-reference goldens of the same configurations."""
    from wekws_amd import pack
    cfg = shape_case_config(case)
    sd = synth.synth_state_dict(pack.model_spec(cfg), case["wseed"] + 1000)
    model = build(cfg, sd)
    B, T = 5, 41
    x = synth.synth_feats(B, T, cfg["input_dim"], seed=case["xseed"] + 7)
    chunks = [1, 3, 10, 7, 20]
    ys, cs = run(model, x, chunks=chunks)
    h0 = np.zeros((cfg["backbone"]["num_layers"], B, cfg["hidden_dim"]), np.float32) if cfg["backbone"]["type"] == "gru" else None
    ry, rc = kws_oracle.forward_streaming(cfg, sd, x, chunks, h0)
    assert max_abs(ys, ry) <= tol_for(ry), max_abs(ys, ry)
    assert max_abs(cs, rc) <= tol_for(rc), max_abs(cs, rc)
    yo, co = run(model, x)
    assert max_abs(ys, yo) <= 2e-5 * max(1.0, float(np.abs(yo).max()))


FSMN_CASES = [c for c in CASES if c["model"].startswith("fsmn")]


@pytest.mark.parametrize("case", FSMN_CASES, ids=[c["name"] for c in FSMN_CASES])
def test_fsmn_precision_f32_is_exact_f32(case, golden, error_report):
    """FSMN had the block-floating kernel only; a precision-F32 request was served by it (missing 4).  Now it runs
    the reference's own arithmetic -- exact f32 products -- on the any-shape path: every FSMN golden of the live reference,
    incl. the 4-D cache, chunked cases and forward_softmax."""
    cfg, sd = case_weights(case)
    model = build(cfg, sd).set_precision("f32")
    assert model.effective_precision() == "f32"
    x = case_input(case)
    y, cache = run(model, x, case_in_cache(case, cfg), softmax=case.get("softmax", False), chunks=case.get("chunks"))
    gy, gc = golden[case["name"] + "/y"], golden[case["name"] + "/cache"]
    c = cache[:1]
    error_report[f"generic/fsmn_f32/{case['name']}/y"] = max_abs(y, gy)
    assert y.shape == gy.shape and max_abs(y, gy) <= tol_for(gy), max_abs(y, gy)
    assert c.shape == gc.shape and max_abs(c, gc) <= tol_for(gc), max_abs(c, gc)


@pytest.mark.parametrize("name,over", [("fsmn_small", dict(left_order=40, right_order=8)),            # 48 taps > 32
                                       ("fsmn_small", dict(linear_dim=1536, proj_dim=640)),           # beyond the 160 KiB LDS tile
                                       ("fsmn_small", dict(num_layers=18))])                          # deeper than the kernel's table
def test_fsmn_beyond_the_kernels_limits(name, over):
    import copy
    from wekws_amd import pack
    cfg = copy.deepcopy(synth.MODEL_CONFIGS[name])
    cfg["backbone"].update(over)
    sd = synth.synth_state_dict(pack.model_spec(cfg), 77)
    model = build(cfg, sd)
    assert model.effective_precision() == "f32"
    x = synth.synth_feats(3, 30, cfg["input_dim"], seed=5)
    chunks = [9, 1, 20]
    ys, cs = run(model, x, chunks=chunks)
    ry, rc = kws_oracle.forward_streaming(cfg, sd, x, chunks, None)
    assert max_abs(ys, ry) <= tol_for(ry), max_abs(ys, ry)
    assert max_abs(cs, rc) <= tol_for(rc), max_abs(cs, rc)


def test_any_shape_full_batch_and_graph_capture():
    """B = 1024 x 98 frames through the any-shape path (DS-TCN with 320 channels): batch-composition invariance (a sub-batch
    gives the same rows) and the forward is capturable after reserve() like every other path."""
    from wekws_amd import pack
    cfg = shape_case_config(GENERIC_CASES[0])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 5)
    model = build(cfg, sd)
    x = torch.from_numpy(synth.synth_feats(1024, 98, cfg["input_dim"], seed=9)).cuda()
    y, c = model(x)
    y2, c2 = model(x[500:517].contiguous())
    assert torch.equal(y[500:517], y2) and torch.equal(c[500:517], c2)
    ry, rc = kws_oracle.forward(cfg, sd, x[:4].cpu().numpy(), None)
    assert max_abs(y[:4].cpu().numpy(), ry) <= tol_for(ry)
    model.reserve(64, 98)
    xs = x[:64].contiguous()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        model.reserve(64, 98)
        model(xs)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            yg, cg = model(xs)
        gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, y[:64]) and torch.equal(cg, c[:64])
