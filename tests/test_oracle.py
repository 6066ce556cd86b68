"""The numpy oracle (oracle/kws_oracle.py) against every golden vector recorded from the LIVE reference
(tests/golden/make_golden.py).  CPU only.  Also pins wekws_amd.pack.model_spec: the seeded weights are
generated from OUR state_dict spec and their checksum must equal the one recorded from the reference's
own state_dict."""
import numpy as np
import pytest

from oracle import kws_oracle
from tests.helpers import CASES, case_in_cache, case_input, case_weights, max_abs
from wekws_amd.utils import synth

Y_TOL = 5e-6      # oracle vs reference on posteriors / logits (float32 summation-order noise)
CACHE_TOL = 2e-5  # MDTC activations are unnormalised and grow to O(10)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_golden(case, golden):
    cfg, sd = case_weights(case)
    name = case["name"]
    assert abs(synth.checksum(sd) - float(golden[name + "/wsum"])) < 1e-6 * float(golden[name + "/wsum"]), \
        "state_dict spec / synthetic weight stream differs from the one the golden was made with"
    x = case_input(case)
    assert abs(np.abs(x.astype(np.float64)).sum() - float(golden[name + "/xsum"])) < 1e-9 * float(golden[name + "/xsum"])
    cache0 = case_in_cache(case, cfg)
    if case.get("chunks"):
        y, cache = kws_oracle.forward_streaming(cfg, sd, x, case["chunks"], cache0) if not case.get("softmax") else (None, None)
    else:
        y, cache = kws_oracle.forward(cfg, sd, x, cache0, softmax=case.get("softmax", False))
    gy, gc = golden[name + "/y"], golden[name + "/cache"]
    assert y.shape == gy.shape
    assert max_abs(y, gy) <= Y_TOL * max(1.0, float(np.abs(gy).max()))
    c = cache if cfg["backbone"]["type"] == "gru" else cache[:1]
    assert c.shape == gc.shape
    assert max_abs(c, gc) <= CACHE_TOL * max(1.0, float(np.abs(gc).max()))


TORCH_REF_CASES = [c for c in CASES if c["cache"] == "empty" and not c.get("chunks") and not c.get("softmax")
                   and c["model"].split("_")[0] in ("ds", "tcn", "mdtc") and "global" not in c["model"]
                   and "last" not in c["model"]]


@pytest.mark.parametrize("case", TORCH_REF_CASES, ids=[c["name"] for c in TORCH_REF_CASES])
def test_torch_cpu_restatement_matches_reference_golden(case, golden):
    """oracle/torch_ref.py (the reference's ATen call sequence, used as bench.py's cpu_baseline) against the same
    live-reference goldens."""
    import torch
    from oracle import torch_ref
    cfg, sd = case_weights(case)
    y, cache = torch_ref.forward(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, torch.from_numpy(case_input(case)))
    gy, gc = golden[case["name"] + "/y"], golden[case["name"] + "/cache"]
    assert max_abs(y.numpy(), gy) <= Y_TOL * max(1.0, float(np.abs(gy).max()))
    assert max_abs(cache.numpy()[:1], gc) <= CACHE_TOL * max(1.0, float(np.abs(gc).max()))


from tests.golden.cases import SCALE_CASES, scaled_case_weights  # noqa: E402


@pytest.mark.parametrize("case", SCALE_CASES, ids=[c["name"] for c in SCALE_CASES])
def test_oracle_matches_scale_sweep_golden(case, scale_golden):
    """The operand-scale sweeps (tests/golden/cases.py::scale_state_dict, goldens from the live reference): the fp32
    oracle follows the reference through weights / activations moved by 2^-20 .. 2^+20."""
    cfg, sd = case_weights(case)
    sd2, xs = scaled_case_weights(case, sd)
    name = case["name"]
    assert abs(synth.checksum(sd2) - float(scale_golden[name + "/wsum"])) <= 1e-6 * float(scale_golden[name + "/wsum"])
    x = (case_input(case) * np.float32(xs)).astype(np.float32)
    cache0 = case_in_cache(case, cfg)
    if case.get("chunks"):
        y, _ = kws_oracle.forward_streaming(cfg, sd2, x, case["chunks"], cache0)
    else:
        y, _ = kws_oracle.forward(cfg, sd2, x, cache0)
    gy = scale_golden[name + "/y"]
    assert y.shape == gy.shape
    assert max_abs(y, gy) <= Y_TOL * max(1.0, float(np.abs(gy).max()))


from tests.golden.cases import GRU_INPUT_CASES, HETERO_CASES, hetero_case_weights  # noqa: E402


@pytest.fixture(scope="module")
def hetero_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hetero_golden.npz"))


@pytest.mark.parametrize("case", HETERO_CASES + GRU_INPUT_CASES, ids=[c["name"] for c in HETERO_CASES + GRU_INPUT_CASES])
def test_oracle_matches_hetero_golden(case, hetero_golden):
    """Heterogeneous per-channel scales inside matrices / operand tiles (tests/golden/cases.py::hetero_state_dict) and the
    GRU's out-of-range inputs, goldens from the live reference: the fp32 oracle follows."""
    cfg, sd = case_weights(case)
    name = case["name"]
    if case.get("hetero"):
        sd = hetero_case_weights(case, sd)
        assert abs(synth.checksum(sd) - float(hetero_golden[name + "/wsum"])) <= 1e-6 * float(hetero_golden[name + "/wsum"])
    x = (case_input(case) * np.float32(case.get("xscale", 1.0))).astype(np.float32)
    cache0 = case_in_cache(case, cfg)
    if case.get("chunks"):
        y, _ = kws_oracle.forward_streaming(cfg, sd, x, case["chunks"], cache0)
    else:
        y, _ = kws_oracle.forward(cfg, sd, x, cache0)
    gy = hetero_golden[name + "/y"]
    assert y.shape == gy.shape
    assert max_abs(y, gy) <= Y_TOL * max(1.0, float(np.abs(gy).max()))


TORCH_REF_STREAM = [c for c in CASES if c.get("chunks") and not c.get("softmax")
                    and c["model"] in ("ds_tcn_h256", "ds_tcn_h64", "tcn_h64", "mdtc_h64", "mdtc_small", "gru_2x128")]


@pytest.mark.parametrize("case", TORCH_REF_STREAM, ids=[c["name"] for c in TORCH_REF_STREAM])
def test_torch_cpu_restatement_streams_like_the_reference(case, golden):
    """oracle/torch_ref.py with a carried cache (bench.py's CPU per-frame-latency baseline) against the streaming traces
    recorded from the live reference, GRU included."""
    import torch
    from oracle import torch_ref
    cfg, sd = case_weights(case)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    x = torch.from_numpy(case_input(case))
    c0 = case_in_cache(case, cfg)
    cache = None if c0 is None else torch.from_numpy(c0)
    ys, t = [], 0
    for n in case["chunks"]:
        y, cache = torch_ref.forward(cfg, tsd, x[:, t:t + n], cache)
        ys.append(y)
        t += n
    y = torch.cat(ys, dim=1).numpy()
    gy, gc = golden[case["name"] + "/y"], golden[case["name"] + "/cache"]
    assert max_abs(y, gy) <= Y_TOL * max(1.0, float(np.abs(gy).max()))
    c = cache.numpy() if cfg["backbone"]["type"] == "gru" else cache.numpy()[:1]
    assert max_abs(c, gc) <= CACHE_TOL * max(1.0, float(np.abs(gc).max()))


from tests.golden.cases import SHAPE_CASES, shape_case_config  # noqa: E402


@pytest.mark.parametrize("case", SHAPE_CASES, ids=[c["name"] for c in SHAPE_CASES])
def test_oracle_matches_odd_width_golden(case):
    """Conv backbones with a hidden_dim none of the kernels is built for (goldens from the live reference,
    tests/golden/make_shape_golden.py): the oracle follows the reference at any width, one-shot and in two chunks."""
    import os
    from wekws_amd import pack
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shape_golden.npz"))
    cfg = shape_case_config(case)
    sd = synth.synth_state_dict(pack.model_spec(cfg), case["wseed"])
    name = case["name"]
    assert abs(synth.checksum(sd) - float(g[name + "/wsum"])) <= 1e-6 * abs(float(g[name + "/wsum"]))
    x = synth.synth_feats(case["B"], case["T"], cfg["input_dim"], seed=case["xseed"])
    y, c = kws_oracle.forward(cfg, sd, x, None)
    gy, gc = g[name + "/y"], g[name + "/cache"]
    assert y.shape == gy.shape and c.shape == gc.shape
    assert max_abs(y, gy) <= Y_TOL * max(1.0, float(np.abs(gy).max()))
    assert max_abs(c, gc) <= Y_TOL * max(1.0, float(np.abs(gc).max()))
    if case.get("split"):
        t1 = case["split"]
        ys, cs = kws_oracle.forward_streaming(cfg, sd, x, [t1, case["T"] - t1], None)
        assert max_abs(ys, g[name + "/y_stream"]) <= Y_TOL * max(1.0, float(np.abs(gy).max()))
        assert max_abs(cs, g[name + "/cache_stream"]) <= Y_TOL * max(1.0, float(np.abs(gc).max()))


from tests.golden.cases import GENERIC_CASES  # noqa: E402


@pytest.mark.parametrize("case", GENERIC_CASES, ids=[c["name"] for c in GENERIC_CASES])
def test_oracle_matches_any_shape_golden(case):
    """Shapes beyond every shipped recipe (wider than 256 channels, kernel sizes above 8 / 5, 33 residual blocks, GRU hidden 192,
    five GRU layers, pooled heads on a GRU; goldens from the live reference, tests/golden/make_generic_golden.py): the oracle
    follows the reference there too -- it is what tests/test_hip_generic.py compares the any-shape HIP path with at other sizes."""
    import os
    from wekws_amd import pack
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "generic_golden.npz"))
    cfg = shape_case_config(case)
    sd = synth.synth_state_dict(pack.model_spec(cfg), case["wseed"])
    name = case["name"]
    assert abs(synth.checksum(sd) - float(g[name + "/wsum"])) <= 1e-6 * abs(float(g[name + "/wsum"]))
    x = synth.synth_feats(case["B"], case["T"], cfg["input_dim"], seed=case["xseed"])
    y, c = kws_oracle.forward(cfg, sd, x, None)
    gy, gc = g[name + "/y"], g[name + "/cache"]
    assert y.shape == gy.shape and c.shape == gc.shape
    assert max_abs(y, gy) <= Y_TOL * max(1.0, float(np.abs(gy).max()))
    assert max_abs(c, gc) <= Y_TOL * max(1.0, float(np.abs(gc).max()))
    if case.get("split"):
        t1 = case["split"]
        ys, cs = kws_oracle.forward_streaming(cfg, sd, x, [t1, case["T"] - t1], None)
        assert max_abs(ys, g[name + "/y_stream"]) <= Y_TOL * max(1.0, float(np.abs(gy).max()))
        assert max_abs(cs, g[name + "/cache_stream"]) <= Y_TOL * max(1.0, float(np.abs(gc).max()))


@pytest.mark.parametrize("seed", range(4))
def test_oracle_against_the_live_reference_on_random_configurations(seed):
    """The GPU fuzz tests (tests/test_hip_parity.py::test_random_model_shapes_against_the_oracle) trust the oracle on random
    configurations no golden covers.  Where the reference tree is present (the build container; not the GPU box) the oracle is run
    against the LIVE reference model on those very configurations -- random incoming caches, a chunk cut, CMVN, NoSubsampling,
    forward_softmax.  (tools/probe/fuzz_oracle_vs_reference.py is the same loop over hundreds of seeds: worst 7.2e-7.)"""
    import contextlib
    import io
    import os
    import sys
    if not os.path.isdir("/root/reference/wekws"):
        pytest.skip("reference tree not present (GPU box)")
    import torch
    sys.path.insert(0, "/root/reference")
    try:
        from wekws.model.cmvn import GlobalCMVN
        from wekws.model.kws_model import init_model as ref_init_model
    finally:
        sys.path.remove("/root/reference")
    from tests.helpers import random_model_config
    rng = np.random.default_rng(7000 + seed)
    for trial in range(12):
        cfg, head = random_model_config(rng)
        with contextlib.redirect_stdout(io.StringIO()):
            model = ref_init_model({k: v for k, v in cfg.items() if k not in ("_cmvn", "cmvn")})
        if cfg.get("_cmvn"):
            model.global_cmvn = GlobalCMVN(torch.zeros(cfg["input_dim"]), torch.ones(cfg["input_dim"]), cfg["cmvn"]["norm_var"])
        sd = synth.synth_state_dict(synth.module_spec(model), 500 + 13 * seed + trial)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.eval()
        B, T = int(rng.choice([1, 2, 3, 9])), int(rng.integers(1, 200))
        x = synth.synth_feats(B, T, cfg["input_dim"], seed=trial, cmvn_like="cmvn" in cfg)
        gru = cfg["backbone"]["type"] == "gru"
        softmax = head == "linear" and bool(rng.integers(0, 4) == 0)
        fwd = model.forward_softmax if softmax else model.forward
        with torch.no_grad():
            h00 = torch.zeros(cfg["backbone"]["num_layers"], B, cfg["hidden_dim"]) if gru else None
            _, c0 = model(torch.from_numpy(x[:, :1]), h00) if gru else model(torch.from_numpy(x[:, :1]))
            cin = (0.5 * np.random.default_rng(trial).standard_normal(tuple(c0.shape))).astype(np.float32) \
                if gru or rng.integers(0, 2) else None
            args = (torch.from_numpy(cin),) if cin is not None else ()
            ty, tc = fwd(torch.from_numpy(x), *args)
        oy, oc = kws_oracle.forward(cfg, sd, x, cin, softmax=softmax)
        what = (seed, trial, cfg, B, T, softmax)
        assert oy.shape == tuple(ty.shape) and oc.shape == tuple(tc.shape), what
        assert max_abs(oy, ty.numpy()) <= 2e-5 * max(1.0, float(ty.abs().max())), what
        assert max_abs(oc, tc.numpy()) <= 2e-5 * max(1.0, float(tc.abs().max())), what
        if head == "linear" and T >= 2 and not softmax:
            cut = int(rng.integers(1, T))
            with torch.no_grad():
                y1, c1 = model(torch.from_numpy(x[:, :cut]), *args)
                y2, c2 = model(torch.from_numpy(x[:, cut:]), c1)
            oys, ocs = kws_oracle.forward_streaming(cfg, sd, x, [cut, T - cut], cin)
            assert max_abs(oys, torch.cat([y1, y2], 1).numpy()) <= 2e-5 * max(1.0, float(ty.abs().max())), (what, cut)
            assert max_abs(ocs, c2.numpy()) <= 2e-5 * max(1.0, float(c2.abs().max())), (what, cut)
