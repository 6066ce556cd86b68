"""GPU parity tests: the HIP path (wekws_amd.KWSModel -> ctypes -> libwekws_hip.so) against
  (a) the golden vectors recorded from the live reference PyTorch CPU forward (tests/golden),
  (b) the numpy oracle on other seeds / shapes,
  (c) size-independent properties at BASELINE's full batch sizes (streaming == one-shot, batch-composition
      invariance, empty cache == zero cache).
Tolerance: north_star's 1e-4 abs on posterior scores; logits / cache activations use 1e-4 relative to
max(1, max|ref|) since they are unnormalised."""
import os

import numpy as np
import pytest
import torch

from oracle import kws_oracle
from tests.golden.cases import (GRU_INPUT_CASES, HETERO_CASES, SCALE_CASES, SHAPE_CASES, hetero_case_weights,
                                scaled_case_weights, shape_case_config)
from tests.helpers import CASES, case_in_cache, case_input, case_weights, max_abs, random_model_config as _random_model_config
from wekws_amd.model.kws_model import init_model
from wekws_amd.utils import synth

pytestmark = pytest.mark.gpu

POSTERIOR_TOL = 1e-4


def tol_for(ref):
    return POSTERIOR_TOL * max(1.0, float(np.abs(ref).max()))


def build(cfg, sd, device="cuda"):
    m = init_model(cfg)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to(device).eval()


def run(model, x, cache=None, softmax=False, chunks=None):
    dev = next(model.parameters()).device
    xt = torch.from_numpy(x).to(dev)
    fwd = model.forward_softmax if softmax else model.forward
    c = None if cache is None else torch.from_numpy(cache).to(dev)
    if chunks:
        ys, t = [], 0
        for n in chunks:
            y, c = fwd(xt[:, t:t + n]) if c is None else fwd(xt[:, t:t + n], c)
            ys.append(y)
            t += n
        y = torch.cat(ys, dim=1)
    else:
        y, c = fwd(xt) if c is None else fwd(xt, c)
    torch.cuda.synchronize()
    return y.cpu().numpy(), c.cpu().numpy()


@pytest.fixture(scope="module")
def models():
    cache = {}

    def get(case, precision="default"):
        key = (case["model"], case.get("odim"), case["cmvn"], case.get("norm_var", True), case["wseed"], precision)
        if key not in cache:
            cfg, sd = case_weights(case)
            cache[key] = (cfg, sd, build(cfg, sd).set_precision(precision))
        return cache[key]
    return get


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden(case, precision, golden, models, error_report):
    """HIP vs the live-reference golden vectors (58 cases: every backbone/head, ragged T, caches, streaming), in
    both matrix precisions (wekws_hip_precision): exact-f32 MFMA and the fp16 hi/lo split."""
    cfg, sd, model = models(case, precision)
    x = case_input(case)
    y, cache = run(model, x, case_in_cache(case, cfg), softmax=case.get("softmax", False), chunks=case.get("chunks"))
    gy, gc = golden[case["name"] + "/y"], golden[case["name"] + "/cache"]
    assert y.shape == gy.shape
    c = cache if cfg["backbone"]["type"] == "gru" else cache[:1]
    error_report[f"golden/{precision}/{case['name']}/y"] = max_abs(y, gy)
    if c.shape == gc.shape:
        error_report[f"golden/{precision}/{case['name']}/cache_rel"] = max_abs(c, gc) / max(1.0, float(np.abs(gc).max()))
    assert max_abs(y, gy) <= tol_for(gy), f"y err {max_abs(y, gy):.3e}"
    assert c.shape == gc.shape
    assert max_abs(c, gc) <= tol_for(gc), f"cache err {max_abs(c, gc):.3e}"


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("case", SCALE_CASES, ids=[c["name"] for c in SCALE_CASES])
def test_scale_sweep(case, precision, scale_golden, error_report):
    """Operand-scale sweeps (tests/golden/cases.py::scale_state_dict): the same function with weights / activations moved
    by 2^-20 .. 2^+20, goldens from the live reference.  fp32 arithmetic is invariant under such rewriting (the
    reference's outputs are bit-identical to the unscaled model's); the split-fp16 kernels must be too: block floating
    point (conv_stack_f16.hip.h) -- no inf / NaN when activations pass 65504, no loss when operands sink below fp16's
    normal range.  Same 1e-4 bar as every other parity test; the measured error goes to gpurun_out/parity_errors.json."""
    cfg, sd = case_weights(case)
    sd2, xs = scaled_case_weights(case, sd)
    model = build(cfg, sd2).set_precision(precision)
    x = (case_input(case) * np.float32(xs)).astype(np.float32)
    y, _ = run(model, x, case_in_cache(case, cfg), chunks=case.get("chunks"))
    gy = scale_golden[case["name"] + "/y"]
    assert y.shape == gy.shape
    err = max_abs(y, gy) if np.isfinite(y).all() else float("inf")
    error_report[f"scale_sweep/{precision}/{case['name']}"] = err
    assert err <= tol_for(gy), f"y err {err:.3e}"


@pytest.fixture(scope="module")
def shape_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shape_golden.npz"))


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("case", SHAPE_CASES, ids=[c["name"] for c in SHAPE_CASES])
def test_hidden_dims_without_a_kernel(case, precision, shape_golden, error_report):
    """hidden_dim is free in the reference (kws_model.py:114); the kernels are built for 32 / 64 / 128 / 256 channels.  Any
    other width up to 256 runs zero-padded to the next built one (wekws_hip.hip::pad_conv_channels): exact, because a zero
    channel stays zero through the network.  Goldens from the live reference (make_shape_golden.py): one-shot posteriors and
    cache (with the MODEL's channel count), and the same input in two chunks with the carried cache."""
    from wekws_amd.utils import synth as synth_
    cfg = shape_case_config(case)
    from wekws_amd import pack
    sd = synth_.synth_state_dict(pack.model_spec(cfg), case["wseed"])
    assert abs(synth_.checksum(sd) - float(shape_golden[case["name"] + "/wsum"])) <= 1e-6 * abs(float(shape_golden[case["name"] + "/wsum"]))
    model = build(cfg, sd).set_precision(precision)
    x = synth_.synth_feats(case["B"], case["T"], cfg["input_dim"], seed=case["xseed"])
    y, c = run(model, x)
    gy, gc = shape_golden[case["name"] + "/y"], shape_golden[case["name"] + "/cache"]
    assert y.shape == gy.shape and c.shape == gc.shape, (y.shape, c.shape)
    error_report[f"shape/{precision}/{case['name']}"] = max_abs(y, gy)
    assert max_abs(y, gy) <= tol_for(gy), max_abs(y, gy)
    assert max_abs(c, gc) <= tol_for(gc), max_abs(c, gc)
    if case.get("split"):
        t1 = case["split"]
        ys, cs = run(model, x, chunks=[t1, case["T"] - t1])
        assert max_abs(ys, shape_golden[case["name"] + "/y_stream"]) <= tol_for(gy)
        assert max_abs(cs, shape_golden[case["name"] + "/cache_stream"]) <= tol_for(gc)


@pytest.fixture(scope="module")
def hetero_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hetero_golden.npz"))


@pytest.mark.parametrize("case", HETERO_CASES, ids=[c["name"] for c in HETERO_CASES])
def test_heterogeneous_scales(case, hetero_golden, error_report):
    """per-channel power-of-two factors spread over 2^E INSIDE matrices and operand tiles
    (tests/golden/cases.py::hetero_state_dict), goldens from the live reference (which is bit-invariant under the
    rewrite).  Default precision must hold the 1e-4 bar at every spread: inside the envelope (2^20,
    WEKWS_HIP_F16X3_ENVELOPE_LOG2) on the split-fp16 kernels, beyond it because wekws_hip_create routes the model to the
    exact-f32 kernels -- and says so (effective_precision).  Spreads along a matrix's K axis ("kcol") never reach the
    kernels: wekws_hip_create balances the columns.  The error of the split-fp16 kernels WITHOUT the routing (option
    envelope = 0) is recorded, not asserted: it is the measurement of where the envelope really ends
    (gpurun_out/parity_errors.json)."""
    kind, E = case["hetero"]
    cfg, sd = case_weights(case)
    sd2 = hetero_case_weights(case, sd)
    gy = hetero_golden[case["name"] + "/y"]
    x = case_input(case)
    model = build(cfg, sd2)
    import warnings
    with warnings.catch_warnings(record=True) as caught:      # the switch to the f32 kernels is announced, once per handle
        warnings.simplefilter("always")
        y, _ = run(model, x, case_in_cache(case, cfg), chunks=case.get("chunks"))
    spread, eff = model.weight_spread_log2(), model.effective_precision()
    said = [w for w in caught if issubclass(w.category, RuntimeWarning) and "exact-f32 kernels" in str(w.message)]
    assert (len(said) >= 1) == (eff == "f32" and cfg.get("_precision", "default") in ("default", "f16x3")), (eff, len(said))
    if kind == "kcol":
        # factors on the K axis of a matrix are absorbed at wekws_hip_create (balance_operand_channels: every column
        # maximum in [1, 2), the inverse factor folded into the per-channel stage in front): no spread left, no routing
        assert spread <= 12 and eff == "f16x3", (spread, eff)
    else:
        assert spread >= E - 0.5
        assert eff == ("f16x3" if spread <= 20 else "f32"), (spread, eff)
        if E <= 8:
            assert eff == "f16x3"                  # (random rows add 2 .. 5 binades of their own: E = 16 / 20 may land on either side)
    err = max_abs(y, gy) if np.isfinite(y).all() else float("inf")
    error_report[f"hetero/default({eff})/{case['name']}"] = err
    assert err <= tol_for(gy), f"y err {err:.3e} at spread 2^{spread:.1f} ({eff})"
    if eff == "f32":                               # the split-fp16 kernels on their own beyond the envelope, measured
        raw = build(cfg, sd2).set_option("envelope", 0)
        yr, _ = run(raw, x, case_in_cache(case, cfg), chunks=case.get("chunks"))
        assert raw.effective_precision() == "f16x3"
        error_report[f"hetero/f16x3_forced/{case['name']}"] = max_abs(yr, gy) if np.isfinite(yr).all() else float("inf")


@pytest.mark.parametrize("case", GRU_INPUT_CASES, ids=[c["name"] for c in GRU_INPUT_CASES])
def test_gru_out_of_range_inputs(case, hetero_golden, error_report):
    """Features x 2^-24 .. 2^+16 (vanishing / saturating gates): the GRU's per-step feature maximum and its
    pre_alpha * max|x| + pre_beta bound far from the magnitudes the other goldens exercise."""
    cfg, sd = case_weights(case)
    x = (case_input(case) * np.float32(case["xscale"])).astype(np.float32)
    for precision in ("f16x3", "f32"):
        y, _ = run(build(cfg, sd).set_precision(precision), x, case_in_cache(case, cfg), chunks=case.get("chunks"))
        gy = hetero_golden[case["name"] + "/y"]
        err = max_abs(y, gy) if np.isfinite(y).all() else float("inf")
        error_report[f"gru_inputs/{precision}/{case['name']}"] = err
        assert err <= tol_for(gy), f"{precision}: y err {err:.3e}"


@pytest.mark.parametrize("name,B,T", [("ds_tcn_h256", 5, 98), ("ds_tcn_h64", 7, 33), ("tcn_h64", 3, 98),
                                      ("mdtc_h64", 5, 98), ("mdtc_small", 9, 61), ("mdtc_h64_global12", 5, 130),
                                      ("mdtc_small_last12", 6, 98), ("gru_2x128", 19, 40), ("gru_1x128", 3, 98),
                                      ("fsmn_ctc300", 5, 33), ("fsmn_small", 7, 61), ("fsmn_ctc", 2, 17),
                                      ("mdtc_h64_80d", 5, 98), ("mdtc_h64_80d", 3, 20)])
def test_vs_oracle_other_seeds(name, B, T):
    """Different weights (wseed 77) / inputs (xseed 5) / odd batch sizes than the goldens, vs the numpy oracle."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 77)
    x = synth.synth_feats(B, T, cfg["input_dim"], seed=5)
    model = build(cfg, sd)
    y, cache = run(model, x)
    ry, rc = kws_oracle.forward(cfg, sd, x, None)
    assert max_abs(y, ry) <= tol_for(ry)
    assert max_abs(cache, rc) <= tol_for(rc)


@pytest.mark.parametrize("name,B", [("ds_tcn_h256", 1024), ("mdtc_h64", 1024), ("mdtc_h64_global12", 1024),
                                    ("gru_2x128", 256), ("fsmn_ctc300", 512)])
def test_full_batch_properties(name, B):
    """BASELINE-size batches (configs 1-3): spot-check 6 utterances against the oracle, and check the
    size-independent properties: a sub-batch gives bit-identical rows; 10-frame streaming == one-shot."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    T = 98 if cfg["backbone"]["type"] != "gru" else 50
    x = synth.synth_feats(B, T, cfg["input_dim"], seed=3)
    model = build(cfg, sd)
    y, cache = run(model, x)
    assert np.isfinite(y).all() and np.isfinite(cache).all()
    idx = np.array([0, 1, B // 3, B // 2, B - 2, B - 1])
    ry, rc = kws_oracle.forward(cfg, sd, x[idx], None)
    assert max_abs(y[idx], ry) <= tol_for(ry)
    gru = cfg["backbone"]["type"] == "gru"
    csel = cache[:, idx] if gru else cache[idx]
    assert max_abs(csel, rc) <= tol_for(rc)
    # batch-composition invariance: utterances are independent, so any sub-batch reproduces its rows exactly
    ys, cs = run(model, np.ascontiguousarray(x[idx]))
    assert np.array_equal(ys, y[idx])
    assert np.array_equal(cs, csel)
    # streaming in 10-frame chunks == one-shot (per-frame heads only)
    if cfg.get("classifier", {}).get("type", "linear") in ("linear", "identity"):
        chunks = [10] * (T // 10) + ([T % 10] if T % 10 else [])
        yst, cst = run(model, x, chunks=chunks)
        assert max_abs(yst, y) <= 2e-5
        assert max_abs(cst, cache) <= 2e-5 * max(1.0, float(np.abs(cache).max()))


@pytest.mark.parametrize("name,B", [("ds_tcn_h256", 1024), ("mdtc_h64", 1024), ("ds_tcn_h64", 1024), ("mdtc_small", 1024)])
def test_full_batch_streaming_with_long_chunks(name, B):
    """BASELINE-size batches through the context variants (round 5): three seconds of features streamed in 80-frame chunks (the
    Android caller's size) and in ragged chunks above 16 frames equal the one-shot forward of the same 294 frames (which runs the
    tiled long-input path) -- posteriors and final cache; a sub-batch of streams reproduces its rows exactly."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    T = 294
    x = synth.synth_feats(B, T, cfg["input_dim"], seed=4)
    model = build(cfg, sd)
    y, cache = run(model, x)
    assert np.isfinite(y).all() and np.isfinite(cache).all()
    for chunks in ([80, 80, 80, 54], [17, 112, 33, 64, 49, 19]):
        ys, cs = run(model, x, chunks=chunks)
        assert max_abs(ys, y) <= 2e-5, (name, chunks, max_abs(ys, y))
        assert max_abs(cs, cache) <= 2e-5 * max(1.0, float(np.abs(cache).max())), (name, chunks)
    idx = np.array([0, 1, B // 3, B // 2, B - 2, B - 1])
    ys2, cs2 = run(model, np.ascontiguousarray(x[idx]), chunks=[80, 80, 80, 54])
    ys1, cs1 = run(model, x, chunks=[80, 80, 80, 54])
    assert np.array_equal(ys2, ys1[idx]) and np.array_equal(cs2, cs1[idx])
    ry, rc = kws_oracle.forward(cfg, sd, x[idx], None)
    assert max_abs(ys1[idx], ry) <= tol_for(ry) and max_abs(cs1[idx], rc) <= tol_for(rc)


@pytest.mark.parametrize("name", ["ds_tcn_h256", "ds_tcn_h64", "mdtc_h64", "mdtc_h64_80d", "mdtc_small", "tcn_h64", "gru_2x128",
                                  "gru_1x128", "ds_tcn_h64_ctc20", "ds_tcn_h256_ctc300", "fsmn_ctc300", "fsmn_small"])
@pytest.mark.parametrize("precision", ["default", "f32"])
def test_random_chunkings_cross_the_kernel_families(name, precision):
    """Seeded fuzz of the dispatcher: a stream cut into random chunk lengths 1 .. 130 -- streaming-step kernels (<= 16 frames), the
    register-resident context variants (17 .. 112), the LDS-tile kernels and the tiled long-input path (> 112), every hand-over of
    the carried cache between them, random batch sizes -- equals the numpy oracle's one-shot forward of the whole input
    (posteriors <= 1e-4, final cache relative) and the HIP one-shot forward (<= 2e-5)."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    model = build(cfg, sd).set_precision(precision)
    # (WEKWS_FUZZ_SEED: tools/probe/fuzz_chunkings.py walks other seeds through this test)
    rng = np.random.default_rng(20260925 + len(name) + 1000003 * int(os.environ.get("WEKWS_FUZZ_SEED", "0")))
    gru = cfg["backbone"]["type"] == "gru"
    for trial in range(16):
        B = int(rng.choice([1, 2, 3, 5, 17, 70]))
        chunks = []
        while sum(chunks) < 200 and len(chunks) < 12:
            kind = rng.integers(0, 4)
            chunks.append(int(rng.integers(1, 17) if kind == 0 else rng.integers(17, 113) if kind in (1, 2) else rng.integers(113, 131)))
        T = sum(chunks)
        x = synth.synth_feats(B, T, cfg["input_dim"], seed=1000 + trial)
        h0 = np.zeros((cfg["backbone"]["num_layers"], B, cfg["hidden_dim"]), np.float32) if gru else None
        ry, rc = kws_oracle.forward(cfg, sd, x, h0)
        ys, cs = run(model, x, cache=h0, chunks=chunks)
        y1, c1 = run(model, x, cache=h0)
        what = (name, precision, trial, B, chunks)
        assert ys.shape == ry.shape and cs.shape == rc.shape, what
        assert max_abs(ys, ry) <= tol_for(ry), (what, max_abs(ys, ry))
        assert max_abs(cs, rc) <= tol_for(rc), (what, max_abs(cs, rc))
        assert max_abs(ys, y1) <= 2e-5 and max_abs(cs, c1) <= 2e-5 * max(1.0, float(np.abs(c1).max())), what


@pytest.mark.parametrize("seed", range(6))
def test_random_model_shapes_against_the_oracle(seed):
    """Seeded fuzz of the routing between the specialised kernels and the any-shape path (wekws_hip_create): 12 random
    configurations per seed, each with random weights (randomised BatchNorm statistics), a random batch and -- per-frame heads --
    a random cut into two chunks with the carried cache, against the numpy oracle (1e-4 on posteriors / relative on logits and
    caches), in the default precision and exact f32."""
    from wekws_amd import pack
    rng = np.random.default_rng(7000 + seed)
    for trial in range(12):
        cfg, head = _random_model_config(rng)
        sd = synth.synth_state_dict(pack.model_spec(cfg), 500 + 13 * seed + trial)
        B, T = int(rng.choice([1, 2, 3, 9, 9, 260])), int(rng.integers(1, 140) if rng.integers(0, 5) else rng.integers(140, 400))
        if B == 260:
            T = min(T, 120)
        x = synth.synth_feats(B, T, cfg["input_dim"], seed=trial, cmvn_like="cmvn" in cfg)
        gru = cfg["backbone"]["type"] == "gru"
        softmax = head == "linear" and bool(rng.integers(0, 4) == 0)
        # the incoming state: GRU always has one (kws_model.py:73 needs h0); conv backbones start empty or from a random cache
        _, c0 = kws_oracle.forward(cfg, sd, x[:, :1], np.zeros((cfg["backbone"]["num_layers"], B, cfg["hidden_dim"]), np.float32) if gru else None)
        cin = None
        if gru or rng.integers(0, 2):
            cin = (0.5 * np.random.default_rng(trial).standard_normal(c0.shape)).astype(np.float32)
        ry, rc = kws_oracle.forward(cfg, sd, x, cin, softmax=softmax)
        for precision in ("default", "f32"):
            model = build(cfg, sd).set_precision(precision)
            what = (seed, trial, precision, cfg, B, T, cin is not None, softmax)
            y, c = run(model, x, cache=cin, softmax=softmax)
            assert y.shape == ry.shape and c.shape == rc.shape, what
            assert max_abs(y, ry) <= tol_for(ry) and max_abs(c, rc) <= tol_for(rc), (what, max_abs(y, ry), max_abs(c, rc))
            if head == "linear" and T >= 2:
                cut = int(rng.integers(1, T))
                ys, cs = run(model, x, cache=cin, softmax=softmax, chunks=[cut, T - cut])
                assert max_abs(ys, ry) <= tol_for(ry) and max_abs(cs, rc) <= tol_for(rc), (what, cut, max_abs(ys, ry), max_abs(cs, rc))


@pytest.mark.parametrize("layers,ksize,T", [(5, 8, 111), (5, 8, 59), (6, 8, 130), (5, 3, 103), (7, 8, 40)])
def test_ds_tcn_h256_deeper_than_the_recipes(layers, ksize, T):
    """Found by the fuzz above (round 5): hidden_dim 256 with a FIFTH block (dilation 16, padding 112) runs ds256_w16, whose
    cache hand-over without an incoming cache wrote one pass of 64 columns -- right for the recipes' paddings (<= 56), columns
    64 .. 111 of a deeper block's slice were never written.  Posteriors and the whole returned cache against the oracle."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
    cfg["backbone"] = dict(cfg["backbone"], num_layers=layers, kernel_size=ksize)
    sd = synth.synth_state_dict(pack.model_spec(cfg), 77)
    x = synth.synth_feats(3, T, 40, seed=5)
    ry, rc = kws_oracle.forward(cfg, sd, x, None)
    for precision in ("default", "f32"):
        model = build(cfg, sd).set_precision(precision)
        y, c = run(model, x)
        assert max_abs(y, ry) <= tol_for(ry) and max_abs(c, rc) <= tol_for(rc), (precision, max_abs(y, ry), max_abs(c, rc))
        ys, cs = run(model, x, chunks=[T // 3, T - T // 3])
        assert max_abs(ys, ry) <= tol_for(ry) and max_abs(cs, rc) <= tol_for(rc), (precision, "chunks")


@pytest.mark.parametrize("ds", [True, False])
@pytest.mark.parametrize("hidden", [32, 16])
def test_narrow_tcn_runs_as_the_64_wide_kernel(ds, hidden):
    """Found by the fuzz above (round 5): DS-TCN / TCN with hidden_dim 32 were taken for a built width (32 is built for MDTC
    only) and failed at the first forward with WEKWS_HIP_EUNSUPPORTED; they are zero-padded to 64 like every other width."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h64" if ds else "tcn_h64"])
    cfg["hidden_dim"] = hidden
    sd = synth.synth_state_dict(pack.model_spec(cfg), 78)
    x = synth.synth_feats(4, 61, 40, seed=6)
    ry, rc = kws_oracle.forward(cfg, sd, x, None)
    for precision in ("default", "f32"):
        model = build(cfg, sd).set_precision(precision)
        ys, cs = run(model, x, chunks=[10, 30, 21])
        assert cs.shape == rc.shape
        assert max_abs(ys, ry) <= tol_for(ry) and max_abs(cs, rc) <= tol_for(rc), (precision, max_abs(ys, ry), max_abs(cs, rc))


@pytest.mark.parametrize("seed", range(4))
def test_random_fsmn_shapes_against_the_oracle(seed):
    """Seeded fuzz of the FSMN routing (fsmn.py:462-495; the split-fp16 tile kernel, its zero-padded widths and the any-shape
    path): random affine / linear / projection widths, memory orders, depths, class counts, feature widths, batch sizes and a
    random cut into chunks with the carried 4-D cache, against the numpy oracle, default precision and exact f32."""
    from wekws_amd import pack
    rng = np.random.default_rng(9100 + seed)
    for trial in range(8):
        cfg = {"input_dim": int(rng.choice([40, 120, 400, 57])), "output_dim": int(rng.choice([2, 11, 300, 2599])),
               "hidden_dim": 0, "preprocessing": {"type": "none"},
               "backbone": {"type": "fsmn", "input_affine_dim": int(rng.choice([32, 72, 140, 200])), "num_layers": int(rng.integers(1, 7)),
                            "linear_dim": int(rng.choice([64, 100, 250, 300])), "proj_dim": int(rng.choice([24, 40, 128, 160])),
                            "left_order": int(rng.integers(1, 14)), "right_order": int(rng.integers(1, 4)), "left_stride": 1,
                            "right_stride": 1, "output_affine_dim": int(rng.choice([40, 56, 140]))},
               "classifier": {"type": "identity", "dropout": 0.1}, "activation": {"type": "identity"}}
        cfg["hidden_dim"] = cfg["backbone"]["linear_dim"]
        sd = synth.synth_state_dict(pack.model_spec(cfg), 900 + 7 * seed + trial)
        B, T = int(rng.choice([1, 2, 5])), int(rng.integers(1, 120))
        x = synth.synth_feats(B, T, cfg["input_dim"], seed=trial)
        _, c0 = kws_oracle.forward(cfg, sd, x[:, :1], None)
        cin = (0.5 * np.random.default_rng(trial).standard_normal(c0.shape)).astype(np.float32) if rng.integers(0, 2) else None
        ry, rc = kws_oracle.forward(cfg, sd, x, cin)
        for precision in ("default", "f32"):
            model = build(cfg, sd).set_precision(precision)
            what = (seed, trial, precision, cfg, B, T, cin is not None)
            y, c = run(model, x, cache=cin)
            assert y.shape == ry.shape and c.shape == rc.shape, what
            assert max_abs(y, ry) <= tol_for(ry) and max_abs(c, rc) <= tol_for(rc), (what, max_abs(y, ry), max_abs(c, rc))
            if T >= 2:
                cut = int(rng.integers(1, T))
                ys, cs = run(model, x, cache=cin, chunks=[cut, T - cut])
                assert max_abs(ys, ry) <= tol_for(ry) and max_abs(cs, rc) <= tol_for(rc), (what, cut, max_abs(ys, ry), max_abs(cs, rc))


@pytest.mark.parametrize("name", ["ds_tcn_h256", "ds_tcn_h64", "mdtc_h64", "mdtc_small", "tcn_h64", "gru_2x128", "fsmn_small"])
def test_pointers_that_are_only_4_byte_aligned(name):
    """The C ABI takes contiguous float buffers, not 16-byte aligned ones: features and the incoming cache that start 4 / 8 / 12
    bytes into an allocation (a contiguous view of a larger tensor) give the results of aligned copies -- wide loads, and with
    them the kernel of the family, are chosen per call from the pointers."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    model = build(cfg, sd)
    B = 3
    for T in (10, 80, 130):
        x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=T)).cuda()
        if cfg["backbone"]["type"] == "gru":
            c0 = (0.3 * torch.randn(cfg["backbone"]["num_layers"], B, cfg["hidden_dim"], generator=torch.Generator().manual_seed(T))).cuda()
        else:
            _, c0 = model(x[:, :7])
        y_ref, c_ref = model(x, c0)
        for off in (1, 2, 3):
            xb = torch.empty(x.numel() + 4, device="cuda")
            xv = xb[off:off + x.numel()].view_as(x)
            xv.copy_(x)
            cb = torch.empty(c0.numel() + 4, device="cuda")
            cv = cb[off:off + c0.numel()].view_as(c0)
            cv.copy_(c0)
            assert xv.is_contiguous() and xv.data_ptr() % 16 == 4 * off and cv.data_ptr() % 16 == 4 * off
            y, c = model(xv, cv)
            # (another kernel of the family may take the call -- the register-resident ones want 16-byte aligned features --, so
            #  equal up to the summation order of the head, not bit for bit)
            ey, ec = float((y - y_ref).abs().max()), float((c - c_ref).abs().max())
            assert ey <= 2e-5 and ec <= 2e-5 * max(1.0, float(c_ref.abs().max())), (name, T, off, ey, ec)


def test_empty_time_axis_and_empty_batch_like_the_reference():
    """T = 0: the reference's own errors (AssertionError from tcn.py:53 / mdtc.py:112 `assert y.size(2) > self.padding`, RuntimeError from
    torch.nn.GRU / the FSMN's memory conv -- recorded from the live reference in the build container).  B = 0 with T > 0: the conv
    backbones return empty outputs of the right shapes, as the reference does."""
    from wekws_amd import pack
    for name, err in (("ds_tcn_h256", AssertionError), ("tcn_h64", AssertionError), ("mdtc_h64", AssertionError),
                      ("gru_2x128", RuntimeError), ("fsmn_small", RuntimeError)):
        cfg = dict(synth.MODEL_CONFIGS[name])
        model = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 1234))
        with pytest.raises(err):
            model(torch.zeros(2, 0, cfg["input_dim"], device="cuda"))
        if err is AssertionError:
            y, c = model(torch.zeros(0, 5, cfg["input_dim"], device="cuda"))
            assert tuple(y.shape) == (0, 5, cfg["output_dim"]) and tuple(c.shape) == pack.cache_shape(pack.parse_config(cfg), 0)


def test_empty_cache_equals_zero_cache():
    from wekws_amd import pack
    for name in ("ds_tcn_h256", "mdtc_h64", "tcn_h64", "fsmn_small"):
        cfg = dict(synth.MODEL_CONFIGS[name])
        sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
        model = build(cfg, sd)
        x = synth.synth_feats(4, 37, cfg["input_dim"], seed=9)
        y0, c0 = run(model, x)
        y1, c1 = run(model, x, np.zeros(pack.cache_shape(pack.parse_config(cfg), 4), np.float32))
        if name == "mdtc_h64":
            # MDTC h64 without a cache runs mdtc64_g4, whose tile holds frames -off .. 16 NT - 1 - off (the utterance's end
            # aligned with a lane boundary): at T = 37 its padding frames are not the 16-wave kernel's, a block-floating
            # scale may come out a binade apart and the results agree to rounding noise instead of bit for bit
            assert max_abs(c0, c1) <= 3e-6 * max(1.0, float(np.abs(c1).max())) and max_abs(y0, y1) <= 3e-6, name
            continue
        assert np.array_equal(c0, c1), name
        # (DS-TCN h256 without a cache runs ds256_g16, whose keyword head sums in another order than the kernels that take a
        # cache: same state bit for bit, posteriors to a few ulp)
        assert np.array_equal(y0, y1) or (name == "ds_tcn_h256" and max_abs(y0, y1) <= 5e-7), name
    # MDTC h64 where NT divides T (T = 98): mdtc64_g4 (no cache) and mdtc64_w16 (zero cache) hold the same frames, and the
    # state they return is the same bit for bit
    cfg = dict(synth.MODEL_CONFIGS["mdtc_h64"])
    model = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 1234))
    x = synth.synth_feats(4, 98, cfg["input_dim"], seed=9)
    y0, c0 = run(model, x)
    y1, c1 = run(model, x, np.zeros(pack.cache_shape(pack.parse_config(cfg), 4), np.float32))
    assert np.array_equal(c0, c1) and max_abs(y0, y1) <= 5e-7


@pytest.mark.parametrize("B", [255, 257, 300, 1000, 1023])
def test_headline_persistent_tail_vs_oracle(B, error_report):
    """ds256_g16<.., FAST> is a persistent kernel (grid = CUs, `b += gridDim.x`, the next utterance's features prefetched
    under a guard): batches that do NOT fill whole rounds of workgroups, every row against the numpy oracle -- through
    forward (with the returned cache) and through posteriors (score-only launch)."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    x = synth.synth_feats(B, 98, cfg["input_dim"], seed=B)
    model = build(cfg, sd).set_precision("f16x3")
    y, cache = run(model, x)
    ry, rc = kws_oracle.forward(cfg, sd, x, None)
    error_report[f"g16_tail/B{B}/y"] = max_abs(y, ry)
    assert y.shape == ry.shape and max_abs(y, ry) <= POSTERIOR_TOL
    assert max_abs(cache, rc) <= tol_for(rc)
    yp = model.posteriors(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(yp, y)


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("name", ["ds_tcn_h256", "mdtc_h64", "ds_tcn_h64", "mdtc_small", "mdtc_small_global12"])
def test_lane_major_lengths_vs_oracle(name, precision, error_report):
    """The register-resident kernels keep frames lane-major (a lane owns NT consecutive frames; mdtc64_g4 aligns the
    utterance's END with a lane boundary): utterance lengths around every lane / tile boundary against the ORACLE (the T
    sweeps elsewhere compare these kernels with the LDS-tile kernels)."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 4321)
    model = build(cfg, sd).set_precision(precision)
    worst = 0.0
    for T in (2, 6, 7, 8, 13, 14, 15, 97, 99, 111, 112):
        x = synth.synth_feats(3, T, cfg["input_dim"], seed=100 + T)
        y, cache = run(model, x)
        ry, rc = kws_oracle.forward(cfg, sd, x, None)
        assert y.shape == ry.shape and cache.shape == rc.shape, (T, y.shape, cache.shape)
        worst = max(worst, max_abs(y, ry))
        assert max_abs(y, ry) <= POSTERIOR_TOL, (T, max_abs(y, ry))
        assert max_abs(cache, rc) <= tol_for(rc), (T, max_abs(cache, rc))
    error_report[f"lane_major_T/{precision}/{name}"] = worst


def test_forward_stream_is_forward_with_cache():
    from wekws_amd import pack
    for name in ("ds_tcn_h256", "gru_2x128", "fsmn_small"):
        cfg = dict(synth.MODEL_CONFIGS[name])
        sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
        model = build(cfg, sd)
        x = torch.from_numpy(synth.synth_feats(2, 30, cfg["input_dim"], seed=4)).cuda()
        cache, ys = None, []
        for t in range(0, 30, 10):
            y, cache = model.forward_stream(x[:, t:t + 10], cache)
            ys.append(y)
        y1, c1 = model(x[:, :10])
        y2, c2 = model(x[:, 10:20], c1)
        assert torch.equal(ys[0], y1) and torch.equal(ys[1], y2)
        yfull, cfull = model(x)
        assert max_abs(torch.cat(ys, 1).cpu().numpy(), yfull.cpu().numpy()) <= 2e-5
        assert max_abs(cache.cpu().numpy(), cfull.cpu().numpy()) <= 2e-5 * max(1.0, float(cfull.abs().max()))


def test_streaming_kernel_equals_batch_kernel():
    """ds256_stream.hip.h (chunks of <= 16 frames, the stream's cache resident in LDS, coalesced cache I/O) against the
    batch kernel fed the same chunks: same arithmetic in the same order -> bit-identical posteriors and caches.  With
    and without an input cache, T = 1 .. 16, ragged batch sizes, both matrix precisions, several output widths."""
    from wekws_amd import pack as packer
    for K in (None, 1, 16):
        cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
        if K:
            cfg["output_dim"] = K
        sd = synth.synth_state_dict(packer.model_spec(cfg), 77)
        for prec in ("default", "f16"):
            # the batch kernel (ds256_w16: it shares conv_stack_head with the streaming kernel) fed the same chunks
            ref = build(cfg, sd).set_precision(prec).set_option("stream", 0).set_option("g16", 0)
            got = build(cfg, sd).set_precision(prec).set_option("stream", 1)
            for B, T in ((5, 10), (3, 16), (2, 1), (300, 7)):
                x = torch.from_numpy(synth.synth_feats(B, 3 * T, 40, seed=B)).cuda()
                cr = cg = None
                for t in range(0, 3 * T, T):
                    xc = x[:, t:t + T]
                    yr, cr = ref(xc) if cr is None else ref(xc, cr)
                    yg, cg = got(xc) if cg is None else got(xc, cg)
                    assert torch.equal(yr, yg) and torch.equal(cr, cg), (K, prec, B, T, t)
    # MDTC h64 (mdtc64_stream.hip.h): two streams per workgroup, odd stream counts, 40-d and 80-d inputs, pooled head
    for name in ("mdtc_h64", "mdtc_h64_80d", "mdtc_h64_global12"):
        cfg = dict(synth.MODEL_CONFIGS[name])
        sd = synth.synth_state_dict(packer.model_spec(cfg), 78)
        for prec in ("default", "f16"):
            # the batch kernel (mdtc64_w16: same head and tile as the streaming kernel) fed the same chunks
            ref = build(cfg, sd).set_precision(prec).set_option("stream", 0).set_option("g16", 0)
            got = build(cfg, sd).set_precision(prec).set_option("stream", 1)
            for B, T in ((5, 10), (2, 16), (1, 1), (301, 7)):
                x = torch.from_numpy(synth.synth_feats(B, 3 * T, cfg["input_dim"], seed=B)).cuda()
                cr = cg = None
                for t in range(0, 3 * T, T):
                    xc = x[:, t:t + T]
                    yr, cr = ref(xc) if cr is None else ref(xc, cr)
                    yg, cg = got(xc) if cg is None else got(xc, cg)
                    assert torch.equal(yr, yg) and torch.equal(cr, cg), (name, prec, B, T, t)
    # a cache that is only 4-byte aligned (a view into a larger buffer) takes the batch kernel: same numbers
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
    sd = synth.synth_state_dict(packer.model_spec(cfg), 77)
    model = build(cfg, sd)
    x = torch.from_numpy(synth.synth_feats(2, 10, 40, seed=1)).cuda()
    _, c = model(x)
    buf = torch.zeros(c.numel() + 1, device="cuda")
    odd = buf[1:].view_as(c)
    odd.copy_(c)
    assert odd.data_ptr() % 16 != 0
    ya, ca = model(x, c)
    yb, cb = model(x, odd)
    assert torch.equal(ya, yb) and torch.equal(ca, cb)
    # and against the oracle, streamed
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
    sd = synth.synth_state_dict(packer.model_spec(cfg), 5)
    model = build(cfg, sd)
    x = synth.synth_feats(2, 50, 40, seed=3)
    y, c = run(model, x, None, chunks=[10] * 5) if "chunks" in run.__code__.co_varnames else (None, None)
    if y is not None:
        ry, rc = kws_oracle.forward(cfg, sd, x, None)
        assert max_abs(y, ry) <= POSTERIOR_TOL


def test_fsmn_head_slices_equal_single_workgroup():
    """Small FSMN-CTC calls split the vocabulary layer over several workgroups per tile (each recomputes the backbone,
    each writes its own o-tiles of y): the same numbers as one workgroup per tile, caches included."""
    from wekws_amd import pack as packer
    for name in ("fsmn_ctc300", "fsmn_ctc"):
        cfg = dict(synth.MODEL_CONFIGS[name])
        sd = synth.synth_state_dict(packer.model_spec(cfg), 5)
        ref = build(cfg, sd).set_option("head_slices", 0)
        for sl in (-1, 3, 8):
            got = build(cfg, sd).set_option("head_slices", sl)
            for B, T in ((1, 10), (5, 10), (3, 32), (40, 7)):
                x = torch.from_numpy(synth.synth_feats(B, 2 * T, cfg["input_dim"], seed=B)).cuda()
                yr, cr = ref(x[:, :T])
                yg, cg = got(x[:, :T])
                assert torch.equal(yr, yg) and torch.equal(cr, cg), (name, sl, B, T)
                yr, cr = ref(x[:, T:], cr)
                yg, cg = got(x[:, T:], cg)
                assert torch.equal(yr, yg) and torch.equal(cr, cg), (name, sl, B, T)


def test_ds_tcn_ctc_head_slices_equal_single_workgroup():
    """The same split for the DS-TCN CTC recipe's 2599-token head (ds256_mm.hip.h)."""
    from wekws_amd import pack as packer
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256_ctc"])
    sd = synth.synth_state_dict(packer.model_spec(cfg), 5)
    ref = build(cfg, sd).set_option("head_slices", 0)
    for sl in (-1, 3, 8):
        got = build(cfg, sd).set_option("head_slices", sl)
        for B, T in ((1, 10), (5, 16), (2, 98)):
            x = torch.from_numpy(synth.synth_feats(B, T + 10, 40, seed=B)).cuda()
            yr, cr = ref(x[:, :T])
            yg, cg = got(x[:, :T])
            assert torch.equal(yr, yg) and torch.equal(cr, cg), (sl, B, T)
            yr, cr = ref(x[:, T:], cr)
            yg, cg = got(x[:, T:], cg)
            assert torch.equal(yr, yg) and torch.equal(cr, cg), (sl, B, T)


def test_posteriors_only_equals_forward():
    """KWSModel.posteriors (C ABI out_cache = NULL: no cache hand-over) returns the very y of forward."""
    from wekws_amd import pack
    for name, T in (("ds_tcn_h256", 98), ("mdtc_h64", 98), ("tcn_h64", 40), ("gru_2x128", 20), ("fsmn_small", 25),
                    ("ds_tcn_h256", 200), ("ds_tcn_h256_ctc300", 50),
                    # round 4: the register-resident small-recipe kernels (ds64_g4, mdtc_g4<32>) and the GRU wavefront at
                    # a full tile / beyond a lap of its rings
                    ("ds_tcn_h64", 98), ("ds_tcn_h64", 45), ("mdtc_small", 98), ("mdtc_small", 33), ("gru_2x128", 98)):
        cfg = dict(synth.MODEL_CONFIGS[name])
        model = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 1234))
        for B in (3, 300):
            x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=6)).cuda()
            assert torch.equal(model.posteriors(x), model(x)[0]), (name, T, B)


def test_forward_is_graph_capturable():
    """wekws_hip_forward makes no hidden synchronisation or allocation once a stream's workspace exists: after a
    warm-up on the capture stream the streaming step records into a HIP graph and replays bit-identically."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
    model = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 1234)).freeze()
    xs = torch.from_numpy(synth.synth_feats(2, 40, 40, seed=8)).cuda()
    x_static = xs[:, :10].clone()
    cache_static = torch.zeros(pack.cache_shape(pack.parse_config(cfg), 2), device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model(x_static, cache_static)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        y_static, c_out = model(x_static, cache_static)
        cache_static.copy_(c_out)
    cache, cache_static[:] = torch.zeros_like(cache_static), 0.0
    for t in range(0, 40, 10):
        y, cache = model(xs[:, t:t + 10], cache)
        x_static.copy_(xs[:, t:t + 10])
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y_static, y) and torch.equal(cache_static, cache)


@pytest.mark.parametrize("name", ["ds_tcn_h256", "ds_tcn_h64", "mdtc_h64"])
def test_context_variant_is_graph_capturable(name):
    """The same for 80-frame chunks with the carried cache -- the register-resident kernels' context variants (round 5): no
    allocation, no synchronisation, replays bit-identically."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    model = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 1234)).freeze()
    B, Tc, n = 3, 80, 4
    xs = torch.from_numpy(synth.synth_feats(B, Tc * n, cfg["input_dim"], seed=8)).cuda()
    x_static = xs[:, :Tc].clone()
    cache_static = torch.zeros(pack.cache_shape(pack.parse_config(cfg), B), device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model(x_static, cache_static)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        y_static, c_out = model(x_static, cache_static)
        cache_static.copy_(c_out)
    cache, cache_static[:] = torch.zeros_like(cache_static), 0.0
    for t in range(0, Tc * n, Tc):
        y, cache = model(xs[:, t:t + Tc].contiguous(), cache)
        x_static.copy_(xs[:, t:t + Tc])
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y_static, y) and torch.equal(cache_static, cache), (name, t)


def test_long_input_tiling_matches_oracle():
    """T > 112 goes through several LDS tiles that hand the context over via the workspace cache."""
    from wekws_amd import pack
    for name, T in (("ds_tcn_h64", 500), ("mdtc_small_global12", 333), ("tcn_h64", 225), ("fsmn_small", 300)):
        cfg = dict(synth.MODEL_CONFIGS[name])
        sd = synth.synth_state_dict(pack.model_spec(cfg), 5)
        model = build(cfg, sd)
        x = synth.synth_feats(3, T, cfg["input_dim"], seed=11)
        y, cache = run(model, x)
        ry, rc = kws_oracle.forward(cfg, sd, x, None)
        assert max_abs(y, ry) <= tol_for(ry)
        assert max_abs(cache, rc) <= tol_for(rc)


@pytest.mark.parametrize("name", ["ds_tcn_h256", "mdtc_h64", "gru_2x128", "fsmn_small", "tcn_h64"])
def test_two_minutes_of_audio_in_one_call(name):
    """T = 12,000 frames (two minutes; the reference's forward takes any length): 108 tiles handing the context over / a
    12,000-step recurrence in one call, against the oracle's one-shot forward, and equal to the same frames streamed in 1,000-frame
    calls."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 6)
    model = build(cfg, sd)
    T = 12000
    x = synth.synth_feats(1, T, cfg["input_dim"], seed=12)
    h0 = np.zeros((cfg["backbone"]["num_layers"], 1, cfg["hidden_dim"]), np.float32) if cfg["backbone"]["type"] == "gru" else None
    y, cache = run(model, x, cache=h0)
    ry, rc = kws_oracle.forward(cfg, sd, x, h0)
    assert y.shape == ry.shape and max_abs(y, ry) <= tol_for(ry), max_abs(y, ry)
    assert max_abs(cache, rc) <= tol_for(rc)
    ys, cs = run(model, x, cache=h0, chunks=[1000] * 12)
    assert max_abs(ys, y) <= 2e-5 * max(1.0, float(np.abs(y).max())) and max_abs(cs, cache) <= 2e-5 * max(1.0, float(np.abs(cache).max()))


def test_no_cpu_fallback():
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h64"])
    m = init_model(cfg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 10, 40))
    m = m.to("cuda")
    with pytest.raises(ValueError):
        m(torch.zeros(1, 10, 41, device="cuda"))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 10, 40, device="cuda"), torch.zeros(1, 64, 3, device="cuda"))


@pytest.mark.parametrize("name,B,T", [("fsmn_ctc300", 601, 30), ("fsmn_small", 1027, 9), ("fsmn_small", 515, 32),
                                      ("fsmn_ctc300", 1030, 16)])
def test_fsmn_packed_utterances(name, B, T):
    """Short inputs in large batches run 2 or 4 utterances per workgroup (odd B exercises the batch tail): every row
    must equal the unpacked result (a small batch of the same rows) bit for bit, and the oracle within tolerance, with
    a random carried cache."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 21)
    model = build(cfg, sd)
    x = synth.synth_feats(B, T, cfg["input_dim"], seed=13)
    cshape = pack.cache_shape(pack.parse_config(cfg), B)
    cache0 = np.random.default_rng(5).standard_normal(cshape).astype(np.float32)
    y, c = run(model, x, cache0)
    idx = np.array([0, 1, 2, 3, B // 2, B - 3, B - 2, B - 1])
    ys, cs = run(model, np.ascontiguousarray(x[idx]), np.ascontiguousarray(cache0[idx]))   # 8 rows: unpacked path
    assert np.array_equal(y[idx], ys) and np.array_equal(c[idx], cs)
    ry, rc = kws_oracle.forward(cfg, sd, x[idx], cache0[idx])
    assert max_abs(ys, ry) <= tol_for(ry)
    assert max_abs(cs, rc) <= tol_for(rc)
    assert np.isfinite(y).all() and np.isfinite(c).all()


def test_same_model_on_two_streams():
    """ONE model driven from two HIP streams at once, for the paths that need scratch memory per call (GRU layer
    sequences; tile hand-over caches of long conv / FSMN inputs; the global head's running sums): the library keeps one
    workspace per (model, stream), so concurrent calls must equal the serial results bit for bit."""
    from wekws_amd import pack
    for name, B, T in (("gru_2x128", 40, 60), ("ds_tcn_h64", 9, 300), ("mdtc_small_global12", 5, 333),
                       ("fsmn_small", 6, 200)):
        cfg = dict(synth.MODEL_CONFIGS[name])
        m = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 3))
        xs = [torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=s)).cuda() for s in (1, 2)]
        serial = [tuple(t.clone() for t in m(x)) for x in xs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [[], []]
        for it in range(6):
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    outs[i].append(m(xs[i]))
        torch.cuda.synchronize()
        for i in range(2):
            for y, c in outs[i]:
                assert torch.equal(y, serial[i][0]) and torch.equal(c, serial[i][1]), name


def test_generic_kernels_behind_the_specialised_ones(golden):
    """Options w16 = 0 / mdtc16 = 0 / mm = 0 (wekws_hip_set_option) route the headline shapes through the generic 8-wave
    kernel (and CTC heads through the vector-ALU classifier): still the same goldens."""
    for case in CASES:
        if case["model"] not in ("ds_tcn_h256", "mdtc_h64", "ds_tcn_h256_ctc300") or case.get("odim"):
            continue
        cfg, sd = case_weights(case)
        model = build(cfg, sd).set_option("w16", 0).set_option("mdtc16", 0).set_option("mm", 0)
        y, cache = run(model, case_input(case), case_in_cache(case, cfg), softmax=case.get("softmax", False),
                       chunks=case.get("chunks"))
        gy, gc = golden[case["name"] + "/y"], golden[case["name"] + "/cache"]
        assert max_abs(y, gy) <= tol_for(gy), case["name"]
        assert max_abs(cache[:1], gc) <= tol_for(gc), case["name"]


def test_register_resident_kernel_equals_lds_tile_kernel():
    """DS-TCN h256 calls without an incoming cache run ds256_g16 (residual tile in registers, depthwise conv through DPP row
    shifts); option g16 = 0 sends them through ds256_w16 (tile in LDS).  The backbone is the same arithmetic in the same
    order: the returned cache (every block's input) must agree bit for bit, for every tile shape (T = 1 .. 112, i.e. NT =
    1 / 2 / 4 / 7), ragged T, long inputs (tiles after the first carry a cache and take the w16 path either way), both
    precisions that have the kernel.  The keyword head of g16 adds its 256 products per output in a different order (64
    register partial sums instead of conv_stack_head's four LDS walks): posteriors agree to a few ulp, not bit for bit."""
    from wekws_amd import pack
    for name in ("ds_tcn_h256",):
        cfg = dict(synth.MODEL_CONFIGS[name])
        sd = synth.synth_state_dict(pack.model_spec(cfg), 77)
        for prec in ("default", "f16"):
            a = build(cfg, sd).set_precision(prec).set_option("g16", 1)
            b = build(cfg, sd).set_precision(prec).set_option("g16", 0)
            for B, T in ((3, 1), (2, 7), (1, 16), (5, 17), (2, 33), (3, 64), (2, 65), (4, 98), (1, 112), (2, 150), (1, 300)):
                x = synth.synth_feats(B, T, cfg["input_dim"], seed=T)
                ya, ca = run(a, x)
                yb, cb = run(b, x)
                assert np.array_equal(ca, cb), (prec, B, T)
                assert max_abs(ya, yb) <= 5e-7, (prec, B, T, max_abs(ya, yb))
                xt = torch.from_numpy(x).cuda()
                # (a first chunk of <= 16 frames that asks for the cache runs ds256_stream, posteriors() runs g16)
                assert max_abs(a.posteriors(xt).cpu().numpy(), ya) <= (5e-7 if T <= 16 else 0.0), (prec, B, T)
                assert max_abs(a.posteriors(xt).cpu().numpy(), b.posteriors(xt).cpu().numpy()) <= 5e-7, (prec, B, T)


def test_register_resident_kernel_with_incoming_cache():
    """Round 5: later chunks of a stream (an incoming cache, 17 .. 112 frames -- the Android caller sends 80,
    runtime/android/app/src/main/cpp/wekws.cc:84-97) run the CONTEXT variant of ds256_g16 (the blocks' left context in a second
    register tile, reached by a second DPP row shift) instead of ds256_w16; option g16 = 3 keeps them on ds256_w16.  Same
    arithmetic in the same order: the returned cache must agree bit for bit, posteriors to the few ulp of the register head; and
    both against the oracle's streaming forward.  Chunks below and above the paddings (7 / 14 / 28 / 56 frames: T < pad returns
    [old tail | new frames]), ragged T (NT does not divide T), NT = 4 and 7 tiles, batches with a persistent tail, a chunk of
    <= 16 frames in between (ds256_stream), both precisions that have the kernel, posteriors only."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 78)
    for prec in ("default", "f16"):
        a = build(cfg, sd).set_precision(prec)
        b = build(cfg, sd).set_precision(prec).set_option("g16", 3)
        for B, chunks in ((3, [40, 80, 17, 98]), (2, [20, 33, 10, 64, 49]), (1, [112, 112, 21]), (300, [80, 80]), (5, [7, 56, 55, 57, 28])):
            T = sum(chunks)
            x = synth.synth_feats(B, T, cfg["input_dim"], seed=T + B)
            ya, ca = run(a, x, chunks=chunks)
            yb, cb = run(b, x, chunks=chunks)
            assert np.array_equal(ca, cb), (prec, B, chunks, max_abs(ca, cb))
            assert max_abs(ya, yb) <= 5e-7, (prec, B, chunks, max_abs(ya, yb))
            if prec == "default" and B <= 5:
                ry, rc = kws_oracle.forward_streaming(cfg, sd, x, chunks, None)
                assert max_abs(ya, ry) <= POSTERIOR_TOL and max_abs(ca, rc) <= tol_for(rc), (B, chunks, max_abs(ya, ry))
        # posteriors only (out_cache = NULL) with an incoming cache
        xt = torch.from_numpy(synth.synth_feats(4, 160, cfg["input_dim"], seed=9)).cuda()
        y1, c1 = a(xt[:, :80].contiguous())
        y2, _ = a(xt[:, 80:].contiguous(), c1)
        y2p = a._run(xt[:, 80:].contiguous(), c1, False, want_cache=False)[0]
        assert torch.equal(y2, y2p), prec


def test_mdtc_register_resident_kernel_with_incoming_cache():
    """The same for MDTC h64 (mdtc64_g4's context variant, round 5): later chunks of 17 .. 112 frames against mdtc64_w16 (option
    g16 = 3) -- caches bit for bit, posteriors to the register head's few ulp -- and against the oracle's streaming forward.
    Chunk lengths that NT divides and that it does not (then the frames below zero inside lane 0 come from the slice too), chunks
    shorter than the largest padding (32 frames), 40-d and 80-d inputs, a stream-kernel chunk in between, a large batch."""
    from wekws_amd import pack
    for name in ("mdtc_h64", "mdtc_h64_80d", "mdtc_small"):    # (mdtc_small: 32 channels, two waves per utterance)
        cfg = dict(synth.MODEL_CONFIGS[name])
        sd = synth.synth_state_dict(pack.model_spec(cfg), 79)
        for prec in ("default", "f16"):
            if prec == "f16" and name == "mdtc_small":       # (its fallback, the generic kernel, has no one-product mode)
                continue
            a = build(cfg, sd).set_precision(prec)
            b = build(cfg, sd).set_precision(prec).set_option("g16", 3)
            for B, chunks in ((3, [40, 80, 17, 98]), (2, [20, 33, 10, 64, 49]), (1, [112, 112, 21]), (1100, [80, 77]), (5, [7, 28, 31, 33, 19])):
                if name == "mdtc_h64_80d" and B > 5:
                    continue
                T = sum(chunks)
                x = synth.synth_feats(B, T, cfg["input_dim"], seed=T + B)
                ya, ca = run(a, x, chunks=chunks)
                yb, cb = run(b, x, chunks=chunks)
                assert np.array_equal(ca, cb), (name, prec, B, chunks, max_abs(ca, cb))
                assert max_abs(ya, yb) <= 5e-7, (name, prec, B, chunks, max_abs(ya, yb))
                if prec == "default" and B <= 5:
                    ry, rc = kws_oracle.forward_streaming(cfg, sd, x, chunks, None)
                    assert max_abs(ya, ry) <= POSTERIOR_TOL and max_abs(ca, rc) <= tol_for(rc), (name, B, chunks, max_abs(ya, ry))


def test_ds64_register_resident_kernel_with_incoming_cache():
    """And for DS-TCN h64 -- the shape of the trained model the reference ships for Android, whose caller streams 80-frame
    chunks (runtime/android/app/src/main/cpp/wekws.cc:84-97): ds64_g4's context variant against the generic kernel (option
    g16 = 3) -- caches bit for bit -- and against the oracle's streaming forward."""
    from wekws_amd import pack
    for name in ("ds_tcn_h64",):
        cfg = dict(synth.MODEL_CONFIGS[name])
        sd = synth.synth_state_dict(pack.model_spec(cfg), 80)
        for prec in ("default",):                              # (the generic kernel has no one-product mode to compare "f16" with)
            a = build(cfg, sd).set_precision(prec)
            b = build(cfg, sd).set_precision(prec).set_option("g16", 3)
            for B, chunks in ((3, [40, 80, 17, 98]), (2, [20, 33, 10, 64, 49]), (1, [112, 112, 21]), (1100, [80, 77]), (5, [7, 56, 55, 57, 28])):
                if name != "ds_tcn_h64" and B > 5:
                    continue
                T = sum(chunks)
                x = synth.synth_feats(B, T, cfg["input_dim"], seed=T + B)
                ya, ca = run(a, x, chunks=chunks)
                yb, cb = run(b, x, chunks=chunks)
                assert np.array_equal(ca, cb), (name, prec, B, chunks, max_abs(ca, cb))
                assert max_abs(ya, yb) <= 5e-7, (name, prec, B, chunks, max_abs(ya, yb))
                if prec == "default" and B <= 5:
                    ry, rc = kws_oracle.forward_streaming(cfg, sd, x, chunks, None)
                    assert max_abs(ya, ry) <= POSTERIOR_TOL and max_abs(ca, rc) <= tol_for(rc), (name, B, chunks, max_abs(ya, ry))


def test_register_resident_f32_kernel_equals_generic_f32_kernel():
    """Precision F32, DS-TCN h256 keyword configuration without an incoming cache: ds256_g32 (tile in registers, exact-f32
    MFMA) against the generic conv_stack_kernel (option g16 = 0) -- the same products, each rounded once; the sums are
    taken in the same order inside a K group but the generic kernel's epilogue / head differ in association, so the
    comparison is to fp32 rounding noise, for every tile shape, ragged T and long inputs."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 77)
    a = build(cfg, sd).set_precision("f32").set_option("g16", 1)
    b = build(cfg, sd).set_precision("f32").set_option("g16", 0)
    for B, T in ((3, 1), (2, 7), (1, 16), (5, 17), (2, 33), (3, 64), (2, 65), (4, 98), (300, 98), (1, 112), (2, 150)):
        x = synth.synth_feats(B, T, cfg["input_dim"], seed=T)
        ya, ca = run(a, x)
        yb, cb = run(b, x)
        assert max_abs(ya, yb) <= 1e-6, (B, T, max_abs(ya, yb))
        assert max_abs(ca, cb) <= 2e-6 * max(1.0, float(np.abs(cb).max())), (B, T, max_abs(ca, cb))
        xt = torch.from_numpy(x).cuda()
        assert max_abs(a.posteriors(xt).cpu().numpy(), ya) <= 1e-6, (B, T)


def test_mdtc_one_utterance_per_workgroup_kernel_equals_16_wave_kernel():
    """MDTC h64 keyword models without an incoming cache run mdtc64_g4 (one utterance per 4-wave workgroup, residual tile
    in registers); option g16 = 0 sends them through mdtc64_w16.  Same arithmetic: where both kernels see the same set of
    padding frames (NT divides T, e.g. the 98-frame utterance) the returned cache agrees bit for bit; elsewhere the
    block-floating scales may be taken from different don't-care frames and the results agree to rounding noise.  40-d and
    80-d inputs (2 / 3 K steps), every tile shape, ragged T, long inputs, both precisions."""
    from wekws_amd import pack
    for name in ("mdtc_h64", "mdtc_h64_80d", "mdtc_h64_global12"):   # (the last: pooled head, classifier.py:26-28)
        cfg = dict(synth.MODEL_CONFIGS[name])
        sd = synth.synth_state_dict(pack.model_spec(cfg), 79)
        for prec in ("default", "f16"):
            a = build(cfg, sd).set_precision(prec).set_option("g16", 1)
            b = build(cfg, sd).set_precision(prec).set_option("g16", 0)
            for B, T in ((3, 1), (2, 7), (1, 16), (5, 17), (2, 33), (3, 64), (2, 65), (4, 98), (301, 98), (1, 112), (3, 111),
                         (2, 150)):
                x = synth.synth_feats(B, T, cfg["input_dim"], seed=T)
                ya, ca = run(a, x)
                yb, cb = run(b, x)
                tol = 3e-6 if prec == "default" else 2e-3
                assert max_abs(ya, yb) <= tol, (name, prec, B, T, max_abs(ya, yb))
                assert max_abs(ca, cb) <= tol * max(1.0, float(np.abs(cb).max())), (name, prec, B, T, max_abs(ca, cb))
                if T == 98:
                    assert np.array_equal(ca, cb), (name, prec, B, T)
                xt = torch.from_numpy(x).cuda()
                assert max_abs(a.posteriors(xt).cpu().numpy(), ya) <= (tol if T <= 16 else 0.0), (name, prec, B, T)


def test_register_resident_kernels_at_every_length():
    """Every T from 1 to 112 (all lane / register boundaries of the lane-major frame layout, slices longer than the input,
    partial lanes in the cache hand-over) through the register-resident kernels against the LDS-tile kernels (option g16 =
    0): DS-TCN h256 caches bit for bit (both kernels see the same 16 NT frames), MDTC h64 to rounding noise (its tile is
    end-aligned: other padding frames, see test_mdtc_one_utterance_per_workgroup_kernel_equals_16_wave_kernel)."""
    from wekws_amd import pack
    for name, exact in (("ds_tcn_h256", True), ("mdtc_h64", False)):
        cfg = dict(synth.MODEL_CONFIGS[name])
        sd = synth.synth_state_dict(pack.model_spec(cfg), 81)
        a = build(cfg, sd).set_option("g16", 1).set_option("stream", 0)
        b = build(cfg, sd).set_option("g16", 0).set_option("stream", 0)
        for T in range(1, 113):
            x = synth.synth_feats(2, T, cfg["input_dim"], seed=1000 + T)
            ya, ca = run(a, x)
            yb, cb = run(b, x)
            if exact:
                assert np.array_equal(ca, cb), (name, T, max_abs(ca, cb))
                assert max_abs(ya, yb) <= 5e-7, (name, T, max_abs(ya, yb))
            else:
                assert max_abs(ca, cb) <= 3e-6 * max(1.0, float(np.abs(cb).max())), (name, T, max_abs(ca, cb))
                assert max_abs(ya, yb) <= 3e-6, (name, T, max_abs(ya, yb))


def test_ds256_matrix_core_depthwise_variant(golden):
    """Option mm = 1 selects the DS-TCN h256 kernel whose depthwise conv also runs on the matrix cores (ds256_mm.hip.h)
    for keyword heads too: same goldens, same tolerance, including streaming and carried caches."""
    for case in CASES:
        if case["model"] != "ds_tcn_h256" or case.get("odim"):
            continue
        cfg, sd = case_weights(case)
        model = build(cfg, sd).set_option("mm", 1)
        y, cache = run(model, case_input(case), case_in_cache(case, cfg), chunks=case.get("chunks"))
        gy, gc = golden[case["name"] + "/y"], golden[case["name"] + "/cache"]
        assert max_abs(y, gy) <= tol_for(gy), case["name"]
        assert max_abs(cache[:1], gc) <= tol_for(gc), case["name"]


def test_fsmn_f32_request_runs_exact_f32():
    """FSMN's specialised kernel multiplies split-fp16 products with block floating point (fsmn_f16.hip.h: fp32-level accuracy
    at any operand scale, test_scale_sweep).  Until round 4 it also served a precision-F32 request (same numbers as the
    default, effective precision 'f16x3'); since round 5 that request runs the reference's own arithmetic on the any-shape
    path (csrc/generic.hip.h) and the library says so.  The two agree to fp32 rounding; both meet the oracle at 1e-4."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["fsmn_small"])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    x = synth.synth_feats(2, 20, cfg["input_dim"], seed=1)
    y0, c0 = run(build(cfg, sd), x)
    m32 = build(cfg, sd).set_precision("f32")
    y1, c1 = run(m32, x)
    ry, rc = kws_oracle.forward(cfg, sd, x, None)
    assert max_abs(y0, ry) <= tol_for(ry) and max_abs(y1, ry) <= tol_for(ry)
    assert max_abs(y0, y1) <= 2e-5 * max(1.0, float(np.abs(ry).max())) and max_abs(c0, c1) <= 2e-5 * max(1.0, float(np.abs(rc).max()))
    assert m32.effective_precision() == "f32" and build(cfg, sd).effective_precision() == "f16x3"


def test_effective_precision_reports_what_runs():
    """desc.precision is a request; wekws_hip_effective_precision is the truth a parity baseline can rely on."""
    from wekws_amd import pack
    want = {("ds_tcn_h256", "default"): "f16x3", ("ds_tcn_h256", "f32"): "f32", ("ds_tcn_h256", "f16"): "f16",
            ("ds_tcn_h64", "f16"): "f16x3", ("ds_tcn_h64", "f32"): "f32", ("mdtc_h64", "f16"): "f16",
            ("mdtc_small", "f16"): "f16x3", ("gru_2x128", "f32"): "f32", ("gru_2x128", "default"): "f16x3",
            ("gru_2x128", "f16"): "f16x3", ("tcn_h64", "f16x3"): "f16x3", ("fsmn_small", "f16"): "f16x3"}
    for (name, prec), eff in want.items():
        cfg = dict(synth.MODEL_CONFIGS[name])
        m = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 3)).set_precision(prec)
        assert m.effective_precision() == eff, (name, prec)
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
    m = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 3)).set_precision("f16").set_option("w16", 0)
    assert m.effective_precision() == "f16x3"           # the generic kernel has no one-product mode


def test_weight_update_repacks():
    """load_state_dict after the first forward must be honoured (handle is keyed on tensor versions)."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h64"])
    x = synth.synth_feats(2, 20, 40, seed=1)
    sd1 = synth.synth_state_dict(pack.model_spec(cfg), 1)
    sd2 = synth.synth_state_dict(pack.model_spec(cfg), 2)
    model = build(cfg, sd1)
    y1, _ = run(model, x)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    y2, _ = run(model, x)
    r2, _ = kws_oracle.forward(cfg, sd2, x, None)
    assert max_abs(y2, r2) <= POSTERIOR_TOL and max_abs(y1, y2) > 1e-3
    # in-place edits are seen too (tensor version counters) ...
    with torch.no_grad():
        model.classifier.linear.bias.add_(1.0)
    y3, _ = run(model, x)
    assert max_abs(y3, y2) > 1e-3
    # ... unless the caller froze the weights; load_state_dict lifts the promise
    model.freeze()
    with torch.no_grad():
        model.classifier.linear.bias.add_(1.0)
    y4, _ = run(model, x)
    assert np.array_equal(y4, y3)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd1.items()})
    y5, _ = run(model, x)
    assert np.array_equal(y5, y1)
    # assign=True replaces the Parameter objects; still honoured
    model.load_state_dict({k: torch.from_numpy(v).cuda() for k, v in sd2.items()}, assign=True)
    y6, _ = run(model, x)
    assert np.array_equal(y6, y2)


def test_large_vocabulary_head_matches_oracle():
    """CTC-style heads (reference ds_tcn_ctc.yaml: thousands of tokens) do not fit the LDS-staged classifier path;
    the fallback that streams the classifier rows from global memory must give the same numbers."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h64_ctc20"], output_dim=700)
    sd = synth.synth_state_dict(pack.model_spec(cfg), 3)
    model = build(cfg, sd)
    x = synth.synth_feats(3, 45, 40, seed=2)
    y, _ = run(model, x)
    ry, _ = kws_oracle.forward(cfg, sd, x, None)
    assert y.shape == (3, 45, 700) and max_abs(y, ry) <= tol_for(ry)
    ys, _ = run(model, x, softmax=True)
    rs, _ = kws_oracle.forward(cfg, sd, x, None, softmax=True)
    assert max_abs(ys, rs) <= POSTERIOR_TOL


def test_ds_tcn_ctc_vocabulary_head():
    """ds_tcn_ctc.yaml's shape (256 channels -> 2599 tokens): the classifier runs on the matrix cores from the
    activation planes of the all-matrix-core kernel; logits, softmax, streaming and batch tail vs the oracle."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256_ctc"])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 9)
    model = build(cfg, sd)
    x = synth.synth_feats(3, 61, 40, seed=4)
    y, c = run(model, x)
    ry, rc = kws_oracle.forward(cfg, sd, x, None)
    assert y.shape == (3, 61, 2599) and max_abs(y, ry) <= tol_for(ry) and max_abs(c, rc) <= tol_for(rc)
    ys, _ = run(model, x, softmax=True)
    rs, _ = kws_oracle.forward(cfg, sd, x, None, softmax=True)
    assert max_abs(ys, rs) <= POSTERIOR_TOL
    yst, cst = run(model, x, chunks=[7, 20, 1, 33])
    assert max_abs(yst, y) <= 2e-5 * max(1.0, float(np.abs(y).max())) and max_abs(cst, c) <= 2e-5 * max(1.0, float(np.abs(c).max()))


def test_concurrent_streams_and_models():
    """Two models on two HIP streams at once (the library keeps no hidden global state besides the per-model
    workspace): results equal the serial ones bit for bit."""
    from wekws_amd import pack
    ms, xs, serial = [], [], []
    for name, seed in (("ds_tcn_h256", 1), ("mdtc_h64", 2)):
        cfg = dict(synth.MODEL_CONFIGS[name])
        m = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), seed))
        x = torch.from_numpy(synth.synth_feats(64, 98, cfg["input_dim"], seed=seed)).cuda()
        ms.append(m); xs.append(x)
        serial.append(m(x)[0].clone())
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for it in range(4):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                outs[i].append(ms[i](xs[i])[0])
    torch.cuda.synchronize()
    for i in range(2):
        for y in outs[i]:
            assert torch.equal(y, serial[i])


EXPORTED = ["ds_tcn_h64_cmvn", "tcn_h32", "mdtc_small", "mdtc_small_global12", "fsmn_small_ctc", "ds_tcn_h40_nopre_cmvn",
            "fsmn_lorder1_ctc"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", EXPORTED)
def test_exported_models(name):
    """Files written by the reference's exporter recipe (tests/golden/make_onnx_golden.py) -> load_exported -> HIP
    forward == the reference PyTorch model the file was exported from (recorded outputs), with and without a cache,
    and batched (the exported graph is batch 1 only; ours is not)."""
    import os
    from wekws_amd.model.kws_model import load_exported
    here = os.path.dirname(os.path.abspath(__file__))
    gold = np.load(os.path.join(here, "golden", "onnx_golden.npz"))
    model = load_exported(os.path.join(here, "golden", "onnx", name + ".onnx")).cuda()
    x, c = torch.from_numpy(gold[name + "/x"]).cuda(), torch.from_numpy(gold[name + "/cache"]).cuda()
    y, rc = model(x, c)
    assert max_abs(y.cpu().numpy(), gold[name + "/y"]) <= POSTERIOR_TOL
    scale = max(1.0, float(np.abs(gold[name + "/r_cache"]).max()))
    assert max_abs(rc.cpu().numpy(), gold[name + "/r_cache"]) <= 1e-4 * scale
    y0, _ = model(x)                                            # empty cache == zero cache (kws_model.py:67-69)
    assert max_abs(y0.cpu().numpy(), gold[name + "/y_zero_cache"]) <= POSTERIOR_TOL
    xb = torch.cat([x, x.flip(1), x], 0)
    cb = torch.cat([c, torch.zeros_like(c), c], 0)
    yb, rb = model(xb, cb)
    assert torch.equal(yb[0], yb[2]) and max_abs(yb[0:1].cpu().numpy(), gold[name + "/y"]) <= POSTERIOR_TOL
    if name.endswith("_ctc"):                                   # the exported function is forward_softmax
        assert float((y.sum(-1) - 1).abs().max()) < 1e-5


@pytest.mark.gpu
def test_reference_android_asset_hip(error_report):
    """The trained DS-TCN the reference ships as an ORT file, converted by tests/tools/make_ref_asset.py in the build
    container (the asset itself is not in this repo): packed file -> C ABI, streamed in 80-frame chunks.  The only REAL
    weights in the suite.  Two streams (make_ref_asset.py): the 3 randn + 10 noise of round 1, on which the trained model's
    posteriors are 1e-8 .. 1e-5 -- an absolute bar on them would pass for an all-zero output, so the LOGITS are compared
    (descriptor activation = identity: the value one node before the graph's Sigmoid), relative to their size --, and an
    input found by gradient ascent that sweeps the keyword posterior 0 -> 0.998 -> 0, on which the 1e-4 posterior bar
    bites.  Margins, posterior range and the spread of the trained matrices go to gpurun_out/parity_errors.json."""
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "ref_asset")
    if not os.path.exists(os.path.join(root, "kws.wekwship")):
        pytest.skip("build/ref_asset not prepared (python tests/tools/make_ref_asset.py where /root/reference exists)")
    import ctypes
    from wekws_amd import _capi, pack
    from wekws_amd.model.kws_model import _HipHandle
    desc, blob = pack.load_packed(os.path.join(root, "kws.wekwship"))
    exp = np.load(os.path.join(root, "expect.npz"))
    if "logit_kw" not in exp.files:
        pytest.skip("build/ref_asset is from an older make_ref_asset.py: re-run it where /root/reference exists")
    lib = _capi.load()
    dd = {k: int(desc[k]) for k in pack.DESC_FIELDS}
    assert dd["activation"] == 1                                # sigmoid, as the graph ends
    h_post, h_logit = _HipHandle(dd, blob, 0), _HipHandle(dict(dd, activation=0), blob, 0)

    def stream(h, x):
        x = torch.from_numpy(x).cuda()
        caches = [torch.zeros(1, 64, 105, device="cuda"), torch.empty(1, 64, 105, device="cuda")]
        y = torch.empty(1, x.size(1), 1, device="cuda")
        for i, t in enumerate(range(0, x.size(1), 80)):
            xc = x[:, t:t + 80].contiguous()
            _capi.check(lib.wekws_hip_forward(h.ptr, xc.data_ptr(), 1, 80, caches[i & 1].data_ptr(),
                                              y[:, t:t + 80].data_ptr(), caches[(i & 1) ^ 1].data_ptr(), 0,
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "forward")
        torch.cuda.synchronize()
        return y.cpu().numpy(), caches[(i + 1) & 1].cpu().numpy()

    for tag, sfx in (("noise", ""), ("keyword", "_kw")):
        x, gy, gl, gc = exp["x" + sfx], exp["y" + sfx], exp["logit" + sfx], exp["cache" + sfx]
        y, cache = stream(h_post, x)
        logit, cache2 = stream(h_logit, x)
        assert np.array_equal(cache, cache2)
        rel = float(np.max(np.abs(logit - gl) / np.maximum(1.0, np.abs(gl))))
        error_report[f"trained_kws_ort/{tag}/y"] = max_abs(y, gy)
        error_report[f"trained_kws_ort/{tag}/logit_rel"] = rel
        error_report[f"trained_kws_ort/{tag}/logit_range"] = [float(gl.min()), float(gl.max())]
        error_report[f"trained_kws_ort/{tag}/posterior_range"] = [float(gy.min()), float(gy.max())]
        error_report[f"trained_kws_ort/{tag}/cache_rel"] = max_abs(cache, gc) / max(1.0, float(np.abs(gc).max()))
        assert np.abs(logit).max() > 5.0                         # (an all-zero output fails here and on the next line)
        assert rel <= 1e-4, f"{tag}: logit rel err {rel:.3e}"
        assert max_abs(y, gy) <= POSTERIOR_TOL
        assert max_abs(cache, gc) <= 1e-4 * max(1.0, float(np.abs(gc).max()))
        if tag == "keyword":
            mid = (gy > 0.05) & (gy < 0.95)
            assert gy.max() > 0.95 and gy.min() < 0.05 and mid.sum() >= 8          # the input does sweep the range
            error_report["trained_kws_ort/keyword/y_err_on_(0.05,0.95)"] = float(np.abs(y - gy)[mid].max())
            assert float(np.abs(y - gy)[mid].max()) <= POSTERIOR_TOL
    error_report["trained_kws_ort/weight_spread_log2"] = float(lib.wekws_hip_weight_spread_log2(h_post.ptr))
    error_report["trained_kws_ort/effective_precision"] = int(lib.wekws_hip_effective_precision(h_post.ptr))


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,T", [("ds_tcn_h256", 3, 98), ("mdtc_h64", 4, 98), ("mdtc_h64_global12", 3, 61),
                                      ("mdtc_h64_80d", 2, 50), ("ds_tcn_h256", 2, 150)])
def test_precision_f16_mode(name, B, T):
    """WEKWS_HIP_PRECISION_F16 (BASELINE.json config 5: fp16 weights + fp16 MFMA pointwise conv): one fp16 product per
    term.  Checked two ways: (1) tightly against the oracle evaluated with the same roundings (fp16 operands into the
    input Linear / pointwise convs, fp32 everywhere else) -- differences are accumulation order plus the rare fp16
    rounding flip of an activation that differs in its last fp32 bit: <= 5e-4 on posteriors / logits of O(1);
    (2) against the exact fp32 oracle at the accuracy fp16 operands allow: <= 1e-2.  And it must differ from the
    F16X3 result (the mode is really on)."""
    from oracle import folded_oracle
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    model = init_model(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.cuda().eval()
    x = synth.synth_feats(B, T, cfg["input_dim"], seed=3)
    xt = torch.from_numpy(x).cuda()
    y3, c3 = model(xt)
    y16, c16 = model.set_precision("f16")(xt)
    y16, y3 = y16.cpu().numpy(), y3.cpu().numpy()
    desc, blob = pack.pack(cfg, sd)
    exact = folded_oracle.forward(desc, blob, x)
    emu = folded_oracle.forward(desc, blob, x, mm_dtype=np.float16)
    scale = max(1.0, float(np.abs(exact).max()))
    assert max_abs(y3, exact) <= POSTERIOR_TOL * scale
    assert max_abs(y16, emu) <= 5e-4 * scale, max_abs(y16, emu)
    assert max_abs(y16, exact) <= 1e-2 * scale, max_abs(y16, exact)
    if T <= 112:                                  # longer inputs tile through kernels that keep F16X3
        assert max_abs(y16, y3) > 1e-6
    # streaming with the carried cache stays self-consistent in this mode too
    ya, ca = model(xt[:, :40])
    yb, cb = model(xt[:, 40:], ca)
    if y16.ndim == 3:
        assert max_abs(torch.cat([ya, yb], 1).cpu().numpy(), y16) <= 2e-3 * scale


def test_executor_test_loop_on_the_global_head():
    """BASELINE config 1's caller shape: the batch-dict loop of Executor.cv / Executor.test (wekws/utils/executor.py:70-115,
    driven by wekws/bin/compute_accuracy.py:92-98) with the 'ce' criterion of the speech-commands recipes
    (wekws/model/loss.py:167-180 cross_entropy, :91-99 acc_frame) -- restated here on the HIP MDTC + GlobalClassifier(12) models, B = 256 per batch --
    must report the loss and accuracy the oracle's logits give."""
    import torch.nn.functional as F
    from wekws_amd import pack

    def executor_test(forward, loader):           # executor.py:70-115 (num_seen_utts starts at 1: appendix B.10)
        num_seen_utts, total_loss, total_acc = 1, 0.0, 0.0
        with torch.no_grad():
            for batch_dict in loader:
                feats, target = batch_dict["feats"], batch_dict["target"]
                target = target[:, 0] if target.shape[1] == 1 else target
                num_utts = batch_dict["feats_lengths"].size(0)
                if num_utts == 0:
                    continue
                logits = forward(feats)
                loss = F.cross_entropy(logits, target.type(torch.int64))                    # loss.py:178
                pred = logits.max(1, keepdim=True)[1]                                       # loss.py:97-99
                acc = pred.eq(target.long().view_as(pred)).sum().item() * 100.0 / logits.size(0)
                if torch.isfinite(loss):
                    num_seen_utts += num_utts
                    total_loss += loss.item() * num_utts
                    total_acc += acc * num_utts
        return total_loss / num_seen_utts, total_acc / num_seen_utts

    for name in ("mdtc_small_global12", "mdtc_h64_global12"):
        cfg = dict(synth.MODEL_CONFIGS[name])
        sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
        model = build(cfg, sd)
        g = np.random.default_rng(3)
        loader = []
        for i, B in enumerate((256, 256, 97)):                                              # last batch ragged
            feats = synth.synth_feats(B, 98, cfg["input_dim"], seed=20 + i)
            loader.append(dict(keys=[f"utt{i}_{j}" for j in range(B)], feats=torch.from_numpy(feats),
                               target=torch.from_numpy(g.integers(0, 12, size=(B, 1))),
                               feats_lengths=torch.full((B,), 98, dtype=torch.int32),
                               target_lengths=torch.ones(B, dtype=torch.int32)))
        got = executor_test(lambda f: model(f.cuda())[0].cpu(), loader)
        ref = executor_test(lambda f: torch.from_numpy(kws_oracle.forward(cfg, sd, f.numpy(), None)[0]), loader)
        assert abs(got[0] - ref[0]) <= 1e-5 * max(1.0, abs(ref[0])) and got[1] == ref[1], (name, got, ref)


def test_reserve_makes_long_inputs_capturable():
    """a call that needs scratch memory (inputs longer than one LDS tile, every GRU call) used to grow its
    workspace -- synchronise, free, allocate -- inside wekws_hip_forward, which breaks HIP graph capture.  Now: a capture
    without a reservation fails cleanly and names wekws_hip_reserve; after model.reserve(B, T) the call records and
    replays bit-identically."""
    from wekws_amd import _capi, pack
    for name, T in (("ds_tcn_h64", 300), ("gru_2x128", 20)):
        cfg = dict(synth.MODEL_CONFIGS[name])
        model = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 1234)).freeze()
        x = torch.from_numpy(synth.synth_feats(3, T, cfg["input_dim"], seed=8)).cuda()
        y_ref, c_ref = model(x)                       # default stream: its own workspace
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            model.reserve(3, T)                       # sizes the workspace of stream s, outside any capture
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y_static, c_static = model(x)
        y_static.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y_static, y_ref) and torch.equal(c_static, c_ref), name
        assert _capi.load().wekws_hip_workspace_bytes(model._get_handle(x.device).ptr, 3, T) > 0


def test_reserve_covers_every_smaller_shape():
    """the scratch need is not monotonic in (B, T) -- a GRU streaming chunk of <= 16 frames spreads its streams
    over more workgroups than a longer call -- so reserve(B, T) has to hold the maximum over the shapes it claims to cover.
    After reserve(256, 20): 10-frame chunks of 256, 128 and 3 streams capture into a HIP graph (a call that had to grow the
    buffer would fail inside the capture) and replay bit-identically."""
    from wekws_amd import _capi, pack
    cfg = dict(synth.MODEL_CONFIGS["gru_2x128"])
    model = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 1234)).freeze()
    lib, dev = _capi.load(), torch.device("cuda", torch.cuda.current_device())
    h = model._get_handle(dev)
    assert lib.wekws_hip_workspace_bytes(h.ptr, 256, 10) > lib.wekws_hip_workspace_bytes(h.ptr, 256, 20)   # the trap itself
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        model.reserve(256, 20)
    torch.cuda.synchronize()
    for B, T in ((256, 10), (128, 10), (3, 16), (256, 20), (200, 1), (1, 20), (40, 19)):   # (the last three: beyond a lap of the rings)
        x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=B + T)).cuda()
        h0 = torch.from_numpy(synth.synth_feats(2, B, 128, seed=3)).cuda().contiguous()
        y_ref, c_ref = model(x, h0)
        torch.cuda.synchronize()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y_static, c_static = model(x, h0)
        for rep in range(3):         # (every replay is a launch of its own epoch: the hand-over tags come from the device)
            y_static.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(y_static, y_ref) and torch.equal(c_static, c_ref), (B, T, rep)
    model.check()


def _gru_cfg(layers):
    cfg = dict(synth.MODEL_CONFIGS["gru_2x128"])
    cfg["backbone"] = dict(cfg["backbone"], num_layers=layers)
    return cfg


@pytest.mark.gpu
@pytest.mark.parametrize("layers", [1, 2, 3, 4])
def test_gru_wavefront_equals_layer_major(layers, error_report):
    """gru_pipe.hip.h runs the layers of torch.nn.GRU (kws_model.py:128-133) as pipeline stages on different CUs, handing
    sequences over through the workspace inside ONE launch.  Same instructions per column as the layer-major kernels
    (option gru_pipe = 0): bit-identical posteriors and states -- for one stream, partial tiles, time-packed tiles (<= 8
    streams), more tiles than resident slots (several rounds per workgroup), T = 1 .. 3 (shorter than the look-ahead), with
    and without an incoming state, called back to back (the progress words must be zero again after every launch), and
    streamed in chunks.  And against the oracle, so that the pair cannot be wrong together."""
    from wekws_amd import pack
    cfg = _gru_cfg(layers)
    sd = synth.synth_state_dict(pack.model_spec(cfg), 4242 + layers)
    # (option 2: the wavefront also where the default falls back to the layer-major kernels -- more tiles than slots)
    pipe, major = build(cfg, sd).set_option("gru_pipe", 2), build(cfg, sd).set_option("gru_pipe", 0)
    rng = np.random.default_rng(7)
    slots = 256 // (2 * layers)
    # (round 4: the hand-over buffers are rings of 16 steps with credits -- T around and beyond a lap, one stream whose first
    # stage writes a whole lap at once (the case that deadlocked before consumers announced progress before waiting), rounds
    # whose boundaries do not coincide with laps)
    shapes = [(1, 10), (1, 1), (3, 2), (5, 3), (16, 7), (19, 40), (8 * slots, 12), (8 * slots + 1, 12), (16 * slots, 21),
              (16 * slots + 17, 9), (48 * slots + 5, 6), (2, 150), (1, 33), (1, 98), (4, 16), (4, 17), (2, 40), (9, 20), (40 * slots, 37)]
    for rep, (B, T) in enumerate(shapes):
        x = synth.synth_feats(B, T, cfg["input_dim"], seed=50 + rep)
        h0 = None if rep % 3 == 0 else (rng.standard_normal((layers, B, 128)) * (0.5 if rep % 3 == 1 else 3.0)).astype(np.float32)
        y1, c1 = run(pipe, x, h0)
        y0, c0 = run(major, x, h0)
        assert np.isfinite(y1).all()
        assert np.array_equal(y1, y0) and np.array_equal(c1, c0), (layers, B, T, max_abs(y1, y0), max_abs(c1, c0))
        y2, c2 = run(pipe, x, h0)                              # again: no state left behind
        assert np.array_equal(y2, y1) and np.array_equal(c2, c1), (layers, B, T)
        if B <= 64:
            ry, rc = kws_oracle.forward(cfg, sd, x, h0)
            error_report[f"gru_pipe/L{layers}/B{B}_T{T}"] = max_abs(y1, ry)
            assert max_abs(y1, ry) <= POSTERIOR_TOL and max_abs(c1, rc) <= tol_for(rc)
    # With an incoming state the two agree bit for bit where their workgroups see the same bound of |h0| (block floating
    # point: the state planes' scale); at large batches the layer-major kernel takes it over 32 streams, the wavefront over
    # its tile of 16 -- where max|h0| > 1 the last bit of h_n may then differ (posteriors carry 22 bits: equal).
    B, T = 40 * slots, 37
    x = synth.synth_feats(B, T, cfg["input_dim"], seed=77)
    h0 = (rng.standard_normal((layers, B, 128)) * 0.5).astype(np.float32)
    (y1, c1), (y0, c0) = run(pipe, x, h0), run(major, x, h0)
    assert max_abs(y1, y0) <= 2e-7 and max_abs(c1, c0) <= 5e-7, (max_abs(y1, y0), max_abs(c1, c0))
    x = synth.synth_feats(33, 47, cfg["input_dim"], seed=99)
    ys, cs = run(pipe, x, chunks=[10, 10, 10, 10, 7])
    yo, co = run(pipe, x)
    assert max_abs(ys, yo) <= 2e-5 and max_abs(cs, co) <= 2e-5
    ym, cm = run(major, x, chunks=[10, 10, 10, 10, 7])
    assert np.array_equal(ys, ym) and np.array_equal(cs, cm)
    pipe.check()                                              # no bounded wait of any of these launches gave up
    major.check()


@pytest.mark.gpu
@pytest.mark.parametrize("idim", [80, 23, 64])
def test_gru_wavefront_other_feature_widths(idim, error_report):
    """The wavefront has two first-stage code paths per feature layout: <= 64 features in whole 16-byte-aligned octets
    (counted assembly loads, what every 40-d test exercises) and everything else (80-d MFCC front ends: three K steps;
    odd widths: scalar tails; plain loads).  Both against the layer-major kernels (bit for bit) and the oracle, one
    stream, time-packed tiles, full tiles, several rounds, T below / beyond a lap of the hand-over rings."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["gru_2x128"], input_dim=idim)
    if "cmvn" in cfg:
        cfg.pop("cmvn")
    sd = synth.synth_state_dict(pack.model_spec(cfg), 5200 + idim)
    pipe, major = build(cfg, sd).set_option("gru_pipe", 2), build(cfg, sd).set_option("gru_pipe", 0)
    for rep, (B, T) in enumerate([(1, 10), (5, 37), (40, 21), (700, 12), (2100, 33)]):
        x = synth.synth_feats(B, T, idim, seed=300 + rep)
        (y1, c1), (y0, c0) = run(pipe, x), run(major, x)
        assert np.array_equal(y1, y0) and np.array_equal(c1, c0), (idim, B, T, max_abs(y1, y0), max_abs(c1, c0))
        if B <= 64:
            ry, rc = kws_oracle.forward(cfg, sd, x, None)
            error_report[f"gru_pipe/idim{idim}/B{B}_T{T}"] = max_abs(y1, ry)
            assert max_abs(y1, ry) <= POSTERIOR_TOL and max_abs(c1, rc) <= tol_for(rc)
    pipe.check()


@pytest.mark.gpu
@pytest.mark.parametrize("odim", [17, 300])
def test_gru_wavefront_wide_heads(odim, error_report):
    """More than 16 outputs: the last layer cannot fold the head into its step loop (one o-tile) and runs the separate head
    pass over the sequence it kept (gru_pipe.hip.h: `last && !head_in`) -- with softmax on top for the CTC-shaped one."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["gru_2x128"], output_dim=odim)
    sd = synth.synth_state_dict(pack.model_spec(cfg), 6100 + odim)
    pipe, major = build(cfg, sd).set_option("gru_pipe", 2), build(cfg, sd).set_option("gru_pipe", 0)
    for rep, (B, T) in enumerate([(1, 10), (6, 40), (48, 21), (1500, 19)]):
        x = synth.synth_feats(B, T, cfg["input_dim"], seed=400 + rep)
        (y1, c1), (y0, c0) = run(pipe, x), run(major, x)
        assert np.array_equal(y1, y0) and np.array_equal(c1, c0), (odim, B, T, max_abs(y1, y0), max_abs(c1, c0))
        if B <= 64:
            ry, rc = kws_oracle.forward(cfg, sd, x, None)
            error_report[f"gru_pipe/odim{odim}/B{B}_T{T}"] = max_abs(y1, ry)
            assert max_abs(y1, ry) <= POSTERIOR_TOL and max_abs(c1, rc) <= tol_for(rc)
    pipe.check()


# (test_gru_wavefront_epoch_wrap: tests/test_hip_gru_safety.py -- it needs the test build of the library)


@pytest.mark.gpu
def test_gru_wavefront_under_uneven_load():
    """The hand-over must not depend on timing: the same call while another stream keeps part of the GPU busy with a
    long-running kernel of another model, many times, every word compared."""
    from wekws_amd import pack
    cfg = _gru_cfg(2)
    sd = synth.synth_state_dict(pack.model_spec(cfg), 4244)
    pipe, major = build(cfg, sd).set_option("gru_pipe", 2), build(cfg, sd).set_option("gru_pipe", 0)
    cfg2 = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
    other = build(cfg2, synth.synth_state_dict(pack.model_spec(cfg2), 1))
    xo = torch.from_numpy(synth.synth_feats(700, 98, 40, seed=1)).cuda()
    x = torch.from_numpy(synth.synth_feats(301, 30, 40, seed=2)).cuda()
    y0, c0 = major(x)
    side = torch.cuda.Stream()
    for it in range(20):
        with torch.cuda.stream(side):
            for _ in range(3):
                other(xo)
        y1, c1 = pipe(x)
        assert torch.equal(y1, y0) and torch.equal(c1, c0), it
    torch.cuda.synchronize()
    pipe.check()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ds_tcn_h64", "ds_tcn_h64_cmvn1"])
def test_ds64_register_resident_kernel(name, error_report):
    """ds64_g4.hip.h (DS-TCN hidden 64 without an incoming cache: one utterance per 4-wave workgroup, tile in registers)
    against the oracle at batch sizes around the workgroup rounds of a 256-CU part and lengths in every tile size, and
    against the generic LDS-tile kernel (option g16 = 0): same products and scales, so the returned cache agrees to the last
    bits; then the cache it returns must continue a stream exactly like the generic kernel's does."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h64"])
    if name.endswith("cmvn1"):                                   # the shape of the reference's shipped model: CMVN, one output
        cfg["_cmvn"] = True
    else:
        cfg["output_dim"] = 2
    sd = synth.synth_state_dict(pack.model_spec(cfg), 2024)
    fast, slow = build(cfg, sd), build(cfg, sd).set_option("g16", 0)
    worst = 0.0
    # (large batches: every row -- with several workgroups per CU a version of this kernel was wrong in ~3 % of them)
    for B, T in ((1, 98), (3, 1), (5, 7), (2, 16), (4, 31), (7, 33), (3, 64), (301, 98), (1030, 98), (6, 112), (3, 150), (4096, 98),
                 (2048, 49), (2048, 100)):
        x = synth.synth_feats(B, T, cfg["input_dim"], seed=7 * B + T)
        if cfg.get("_cmvn"):
            x = (3 * x + 10).astype(np.float32)
        y, c = run(fast, x)
        ys, cs = run(slow, x)
        assert max_abs(c, cs) <= 3e-6 * max(1.0, float(np.abs(cs).max())) and max_abs(y, ys) <= 2e-6, (B, T, max_abs(c, cs), max_abs(y, ys))
        if B <= 301:
            ry, rc = kws_oracle.forward(cfg, sd, x, None)
            worst = max(worst, max_abs(y, ry))
            assert max_abs(y, ry) <= POSTERIOR_TOL and max_abs(c, rc) <= tol_for(rc), (B, T)
        x2 = synth.synth_feats(B, 10, cfg["input_dim"], seed=T)
        y2, c2 = run(fast, x2, c)                               # (with a cache: the generic kernel in both models)
        y2s, c2s = run(slow, x2, cs)
        assert max_abs(y2, y2s) <= 2e-6 and max_abs(c2, c2s) <= 3e-6 * max(1.0, float(np.abs(c2s).max()))
    error_report[f"ds64_g4/{name}"] = worst


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mdtc_small", "mdtc_small_global12", "mdtc_small_last12", "mdtc_h64", "mdtc_h64_global12"])
def test_mdtc_register_resident_kernels_all_rows(name, error_report):
    """mdtc64_g4.hip.h at C = 64 and (round 4) C = 32 (mdtc_small.yaml: two waves per utterance, eight workgroups per CU)
    against the generic LDS-tile kernel (option g16 = 0), EVERY row of batches that put several rounds of workgroups on every
    CU, all tile sizes, aligned and unaligned lengths; a sample of rows against the oracle."""
    from wekws_amd import pack
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 31)
    fast, slow = build(cfg, sd), build(cfg, sd).set_option("g16", 0)
    worst = 0.0
    for B, T in ((4099, 98), (2048, 49), (2050, 100), (1030, 28), (515, 9), (3, 150)):
        x = synth.synth_feats(B, T, cfg["input_dim"], seed=B + T)
        y, c = run(fast, x)
        ys, cs = run(slow, x)
        scale = max(1.0, float(np.abs(ys).max()))
        assert max_abs(c, cs) <= 3e-6 * max(1.0, float(np.abs(cs).max())), (B, T, max_abs(c, cs))
        assert max_abs(y, ys) <= 3e-6 * scale, (B, T, max_abs(y, ys))
        idx = np.linspace(0, B - 1, 24).astype(int)
        ry, rc = kws_oracle.forward(cfg, sd, x[idx], None)
        worst = max(worst, max_abs(y[idx], ry) / max(1.0, float(np.abs(ry).max())))
        assert max_abs(y[idx], ry) <= tol_for(ry) and max_abs(c[idx], rc) <= tol_for(rc), (B, T)
    error_report[f"mdtc_g4_all_rows/{name}"] = worst
