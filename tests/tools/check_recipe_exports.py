#!/usr/bin/env python3
"""Build-container only (needs /root/reference): export EVERY recipe config of examples/hi_xiaowen/s0/conf with the
reference's exporter call (random-init weights, full recipe sizes -- up to 3.8 MB, too large to commit as fixtures),
read each file back with wekws_amd.utils.onnx_lower and compare the recognised model (numpy oracle) with the live
PyTorch model on a random input + cache.  The committed small fixtures are tests/golden/onnx/ (make_onnx_golden.py).

    PYTHONPATH=/root/reference:/root/repo python tests/tools/check_recipe_exports.py
"""
import contextlib
import io
import os
import sys
import tempfile
import warnings

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, "/root/reference")
from torch.onnx._internal.torchscript_exporter import onnx_proto_utils  # noqa: E402

onnx_proto_utils._add_onnxscript_fn = lambda proto, opsets: proto      # see tests/golden/make_onnx_golden.py
from wekws.model.kws_model import init_model  # noqa: E402
from oracle import kws_oracle  # noqa: E402
from wekws_amd.utils.onnx_lower import load_model_file  # noqa: E402

CONF = "/root/reference/examples/hi_xiaowen/s0/conf/"


def main():
    worst = 0.0
    for f in sorted(os.listdir(CONF)):
        cfg = yaml.load(open(CONF + f), Loader=yaml.FullLoader)
        mc, dc = cfg["model"], cfg["dataset_conf"]
        (mc.get("cmvn") or {}).pop("cmvn_file", None)
        dim = dc.get("fbank_conf", {}).get("num_mel_bins", 40)
        if dc.get("feats_type") == "mfcc":
            dim = dc["mfcc_conf"]["num_ceps"]
        if dc.get("context_expansion"):
            ce = dc["context_expansion_conf"]
            dim *= ce["left"] + ce["right"] + 1
        ctc = cfg["training_config"].get("criterion", "max_pooling") == "ctc"
        mc.setdefault("input_dim", dim)                       # train.py fills these two in from the data
        mc.setdefault("output_dim", 2599 if ctc else 2)
        with contextlib.redirect_stdout(io.StringIO()):
            m = init_model(mc)
        if not hasattr(m.backbone, "padding"):
            print(f"{f}: backbone has no .padding -- export_onnx.py:56 cannot export it")
            continue
        m.eval()
        if ctc:
            m.forward = m.forward_softmax
        c = torch.zeros(1, m.hdim, m.backbone.padding)
        if mc["backbone"]["type"] == "fsmn":
            c = c.unsqueeze(-1).expand(-1, -1, -1, mc["backbone"]["num_layers"])
        with tempfile.TemporaryDirectory() as d, warnings.catch_warnings():
            warnings.simplefilter("ignore")
            path = os.path.join(d, "m.onnx")
            torch.onnx.export(m, (torch.randn(1, 100, mc["input_dim"]), c), path, input_names=["input", "cache"],
                              output_names=["output", "r_cache"], dynamic_axes={"input": {1: "T"}, "output": {1: "T"}},
                              opset_version=13, do_constant_folding=True, dynamo=False)
            size = os.path.getsize(path)
            cfg2, sd, info = load_model_file(path)
        if info["softmax"]:
            cfg2["_exported_softmax"] = True
        x, cc = torch.randn(1, 50, mc["input_dim"]), torch.randn(tuple(c.shape)) * 0.3
        with torch.no_grad():
            y, rc = m(x, cc)
        yy, rr = kws_oracle.forward(cfg2, sd, x.numpy(), cc.numpy())
        ey, ec = float(np.abs(yy - y.numpy()).max()), float(np.abs(rr - rc.numpy()).max())
        worst = max(worst, ey)
        print(f"{f}: {size} B  {cfg2['backbone']['type']}{' +softmax' if info['softmax'] else ''}  "
              f"y err {ey:.1e}  cache err {ec:.1e}")
    assert worst <= 1e-6
    print("all recognised")


if __name__ == "__main__":
    main()
