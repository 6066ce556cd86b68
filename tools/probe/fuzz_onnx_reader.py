#!/usr/bin/env python3
"""Development tool (build container only: needs /root/reference): random configurations through the reference exporter's recipe
(tests/golden/make_onnx_golden.py) -> wekws_amd.utils.onnx_lower.load_model_file -> oracle forward of the lowered (config,
state_dict), against the live reference model's outputs.    PYTHONPATH=/root/reference:/root/repo python tools/probe/fuzz_onnx_reader.py [n]"""
import contextlib
import io
import os
import subprocess
import sys
import tempfile
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from torch.onnx._internal.torchscript_exporter import onnx_proto_utils  # noqa: E402

onnx_proto_utils._add_onnxscript_fn = lambda proto, opsets: proto
from wekws.model.kws_model import init_model  # noqa: E402  (the reference)
from wekws.model.cmvn import GlobalCMVN  # noqa: E402

from oracle import kws_oracle  # noqa: E402
from tests.golden.make_onnx_golden import metadata_entry  # noqa: E402
from tests.helpers import random_model_config as _random_model_config  # noqa: E402
from wekws_amd import pack  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402
from wekws_amd.utils.onnx_lower import load_model_file  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
MODEL_CONVERT = os.path.join(ROOT, "runtime", "build", "model_convert")
rng = np.random.default_rng(31337)
bad = done = 0
tmp = tempfile.mkdtemp()
while done < n:
    if rng.integers(0, 4) == 0:
        proj = int(rng.choice([24, 40, 64]))
        cfg = {"input_dim": int(rng.choice([40, 120])), "output_dim": int(rng.choice([2, 11, 50])), "hidden_dim": proj,
               "preprocessing": {"type": "none"},
               "backbone": {"type": "fsmn", "input_affine_dim": int(rng.choice([32, 72])), "num_layers": int(rng.integers(1, 5)),
                            "linear_dim": int(rng.choice([64, 100])), "proj_dim": proj, "left_order": int(rng.integers(1, 12)),
                            "right_order": int(rng.integers(1, 4)), "left_stride": 1, "right_stride": 1,
                            "output_affine_dim": int(rng.choice([40, 56]))},
               "classifier": {"type": "identity", "dropout": 0.1}, "activation": {"type": "identity"}}
        head = "identity"
    else:
        cfg, head = _random_model_config(rng)
        if cfg["backbone"]["type"] == "gru":          # (no `padding` attribute: export_onnx.py:56 raises)
            continue
    # (forward_softmax is exported for CTC recipes, whose activation is the identity: export_onnx.py:46-48, ds_tcn_ctc.yaml:41-42)
    softmax = head in ("linear", "identity") and cfg.get("activation", {}).get("type") == "identity" and bool(rng.integers(0, 2))
    with contextlib.redirect_stdout(io.StringIO()):
        model = init_model(cfg)
    if cfg.get("_cmvn"):
        model.global_cmvn = GlobalCMVN(torch.zeros(cfg["input_dim"]), torch.ones(cfg["input_dim"]), cfg["cmvn"]["norm_var"])
    sd = synth.synth_state_dict(synth.module_spec(model), 100 + done)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    if softmax:
        model.forward = model.forward_softmax
    is_fsmn = cfg["backbone"]["type"] == "fsmn"
    cache = torch.zeros(1, model.hdim, model.backbone.padding)
    if is_fsmn:
        cache = cache.unsqueeze(-1).expand(-1, -1, -1, cfg["backbone"]["num_layers"])
    path = os.path.join(tmp, f"m{done}.onnx")
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(model, (torch.randn(1, 100, cfg["input_dim"]), cache), path, input_names=["input", "cache"],
                              output_names=["output", "r_cache"], dynamic_axes={"input": {1: "T"}, "output": {1: "T"}},
                              opset_version=13, verbose=False, do_constant_folding=True, dynamo=False)
    except Exception as e:   # (configurations the exporter itself cannot trace are not the reader's business)
        print("export failed", type(e).__name__, str(e)[:100], cfg)
        continue
    with open(path, "ab") as f:
        f.write(metadata_entry("cache_dim", str(model.hdim)))
        f.write(metadata_entry("cache_len", str(model.backbone.padding)))
    done += 1
    T = int(rng.integers(1, 120))
    g = torch.Generator().manual_seed(done)
    x = torch.randn(1, T, cfg["input_dim"], generator=g)
    c = torch.randn(tuple(cache.shape), generator=g) * 0.5
    with torch.no_grad():
        y, rc = model(x, c)
    try:
        cfg2, sd2, info = load_model_file(path)
        if info["softmax"]:
            cfg2["_exported_softmax"] = True
        assert info["softmax"] == softmax, ("softmax flag", info["softmax"], softmax)
        spec = dict(pack.model_spec(cfg2))
        assert set(spec) == set(sd2), "names"
        y2, c2 = kws_oracle.forward(cfg2, sd2, x.numpy(), c.numpy(), softmax=softmax)
        ey, ec = float(np.abs(y2 - y.numpy()).max()), float(np.abs(c2 - rc.numpy()).max())
        tol = 1e-5 * max(1.0, float(np.abs(y.numpy()).max()))
        assert y2.shape == tuple(y.shape) and ey <= tol and ec <= 1e-5 * max(1.0, float(rc.abs().max())), (ey, ec)
        # the C++ reader (runtime/kws/model_file.cc) must write the very descriptor and blob the Python reader + packer produce
        if os.path.exists(MODEL_CONVERT):
            outp = os.path.join(tmp, "m.wekwship")
            r = subprocess.run([MODEL_CONVERT, path, outp], capture_output=True, text=True)
            assert r.returncode == 0, "model_convert: " + r.stderr[:200]
            desc, blob = pack.load_packed(outp)
            pdesc, pblob = pack.pack(cfg2, sd2)
            assert {k: int(desc[k]) for k in pack.DESC_FIELDS} == {k: int(pdesc[k]) for k in pack.DESC_FIELDS}, "C++ descriptor"
            assert np.array_equal(blob.view(np.uint32), pblob.view(np.uint32)), "C++ blob bits"
    except Exception as e:
        bad += 1
        print("FAIL", type(e).__name__, str(e)[:300], cfg, "softmax", softmax, flush=True)
print(f"onnx reader fuzz: {done} exported models, {bad} failures")
