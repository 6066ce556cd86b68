"""Exported-model readers (wekws_amd/utils/onnx_model.py, onnx_lower.py) -- CPU side.

Fixtures: tests/golden/onnx/*.onnx are written by the reference's exporter recipe on the live reference model
(tests/golden/make_onnx_golden.py), onnx_golden.npz holds the reference PyTorch outputs for a seeded input / cache.
Chain of evidence:  torch reference == numpy graph executor (pins parser + executor)  and
torch reference == oracle forward of the *lowered* (config, state_dict) (pins the recogniser).  The HIP leg is in
tests/test_hip_parity.py::test_exported_models."""
import os

import numpy as np
import pytest

from oracle import kws_oracle, onnx_graph_oracle
from wekws_amd import pack
from wekws_amd.utils import onnx_model
from wekws_amd.utils.onnx_lower import load_model_file, lower

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["ds_tcn_h64_cmvn", "tcn_h32", "mdtc_small", "mdtc_small_global12", "fsmn_small_ctc", "ds_tcn_h40_nopre_cmvn", "fsmn_lorder1_ctc"]
REF_ORT = "/root/reference/runtime/android/app/src/main/assets/kws.ort"


def fixture(name):
    return os.path.join(HERE, "golden", "onnx", name + ".onnx")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "onnx_golden.npz"))


@pytest.mark.parametrize("name", CASES)
def test_graph_executor_matches_reference_torch(name, gold):
    # export_onnx.py:87-94 checks torch vs onnxruntime at atol 1e-6 on its dummy input; same bar here (logit heads
    # are O(1) so 1e-6 holds for them too)
    g = onnx_model.load_graph(fixture(name))
    assert g.inputs == ["input", "cache"] and g.outputs == ["output", "r_cache"]
    assert set(g.meta) == {"cache_dim", "cache_len"}
    out = onnx_graph_oracle.run(g, dict(input=gold[name + "/x"], cache=gold[name + "/cache"]))
    assert np.abs(out["output"] - gold[name + "/y"]).max() <= 1e-6
    assert np.abs(out["r_cache"] - gold[name + "/r_cache"]).max() <= 3e-6
    out = onnx_graph_oracle.run(g, dict(input=gold[name + "/x"], cache=np.zeros_like(gold[name + "/cache"])))
    assert np.abs(out["output"] - gold[name + "/y_zero_cache"]).max() <= 1e-6


@pytest.mark.parametrize("name", CASES)
def test_lowered_model_matches_reference_torch(name, gold):
    cfg, sd, info = load_model_file(fixture(name))
    assert info["softmax"] == name.endswith("_ctc")
    if info["softmax"]:
        cfg["_exported_softmax"] = True
    # the recovered state_dict has exactly the reference's names and shapes for the recovered config
    spec = dict(pack.model_spec(cfg))
    assert set(spec) == set(sd)
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in spec.items())
    y, c = kws_oracle.forward(cfg, sd, gold[name + "/x"], gold[name + "/cache"])
    assert np.abs(y - gold[name + "/y"]).max() <= 1e-6
    assert np.abs(c - gold[name + "/r_cache"]).max() <= 3e-6
    # metadata the C++ runtime reads (keyword_spotting.cc:33-40) == geometry of the recovered model
    shape = pack.cache_shape(pack.parse_config(cfg), 1)
    assert (int(info["meta"]["cache_dim"]), int(info["meta"]["cache_len"])) == (shape[1], shape[2])
    # and the folded blob packs
    desc, blob = pack.pack(cfg, sd)
    assert blob.size == pack.blob_elems(desc)
    assert desc["activation"] == (pack.ACT_SOFTMAX if info["softmax"] else desc["activation"])


def test_recovered_configs():
    cfg, _, _ = load_model_file(fixture("mdtc_small"))
    assert cfg["backbone"] == dict(type="mdtc", num_stack=3, stack_size=4, kernel_size=5, hidden_dim=32, causal=True)
    cfg, _, _ = load_model_file(fixture("ds_tcn_h64_cmvn"))
    assert cfg["backbone"]["ds"] and cfg["backbone"]["num_layers"] == 4 and cfg["cmvn"] == dict(norm_var=True)
    cfg, _, _ = load_model_file(fixture("mdtc_small_global12"))
    assert cfg["classifier"]["type"] == "global" and cfg["output_dim"] == 12
    cfg, _, _ = load_model_file(fixture("fsmn_small_ctc"))
    assert cfg["backbone"]["left_order"] == 5 and cfg["backbone"]["right_order"] == 1 and cfg["input_dim"] == 120


def test_malformed_files_are_refused(tmp_path):
    data = open(fixture("tcn_h32"), "rb").read()
    with pytest.raises(onnx_model.ModelFileError):
        onnx_model.parse_onnx(data[:len(data) // 2])               # truncated
    with pytest.raises(onnx_model.ModelFileError):
        onnx_model.parse_ort(b"\x00" * 64)                          # no ORTM identifier
    with pytest.raises(onnx_model.ModelFileError):
        onnx_model.parse_onnx(b"")                                  # no graph
    # a graph that is not a wekws export: drop the residual Add of block 0
    g = onnx_model.parse_onnx(data)
    victim = next(n for n in g.nodes if n.op == "Add" and all(i not in g.init for i in n.inputs))
    for n in g.nodes:
        n.inputs = [victim.inputs[0] if i == victim.outputs[0] else i for i in n.inputs]
    g.nodes.remove(victim)
    with pytest.raises(onnx_model.ModelFileError, match="unrecognised"):
        lower(g)
    # metadata that contradicts the graph
    g = onnx_model.parse_onnx(data)
    g.meta["cache_len"] = "104"
    with pytest.raises(onnx_model.ModelFileError, match="cache_dim/cache_len"):
        lower(g)


def test_ort_flatbuffer_reader_roundtrip():
    # no ORT writer exists here, so the .ort container is exercised through a minimal FlatBuffers image built by hand:
    # InferenceSession{ort_version, model{graph{...}}} with one initializer and one node
    from tests.helpers import build_tiny_ort
    data, expect = build_tiny_ort()
    g = onnx_model.parse_ort(data)
    assert g.producer == "onnxruntime 1.12.0" and g.meta == {"cache_dim": "4"}
    assert g.inputs == ["input"] and g.outputs == ["output"]
    assert [n.op for n in g.nodes] == ["Relu"] and g.nodes[0].attrs == {"alpha": 0.5, "axes": [1, 2], "mode": "x"}
    assert np.array_equal(g.init["w"], expect)


@pytest.mark.skipif(not os.path.exists(REF_ORT), reason="reference tree not present (GPU box)")
def test_reference_android_asset():
    """The one trained model the reference ships: an ORT-optimised DS-TCN (FusedConv / FusedMatMul).  Its recognised
    form must compute what the graph computes."""
    g = onnx_model.load_graph(REF_ORT)
    assert {n.op for n in g.nodes} >= {"FusedConv", "FusedMatMul", "Sigmoid"}
    cfg, sd, info = lower(g)
    assert cfg["backbone"] == dict(type="tcn", ds=True, num_layers=4, kernel_size=8, dropout=0.0)
    assert (cfg["input_dim"], cfg["hidden_dim"], cfg["output_dim"]) == (40, 64, 1) and cfg["_cmvn"]
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((1, 80, 40)) * 3 + 10).astype(np.float32)      # Android chunk: 80 frames
    cache = (rng.standard_normal((1, 64, 105)) * 0.5).astype(np.float32)
    out = onnx_graph_oracle.run(g, dict(input=x, cache=cache))
    y, c = kws_oracle.forward(cfg, sd, x, cache)
    assert np.abs(y - out["output"]).max() <= 1e-6 and np.abs(c - out["r_cache"]).max() <= 3e-6
    assert out["output"].shape == (1, 80, 1) and 0.0 < out["output"].min() and out["output"].max() < 1.0
