#!/usr/bin/env python3
"""Generate tests/golden/shape_golden.npz from the LIVE reference: conv backbones whose hidden_dim is none of the widths the
kernels are built for (kws_model.py:114 takes any hidden_dim; the library runs them zero-padded to the next built width,
wekws_hip.hip::pad_conv_channels).  Per case: one-shot forward without a cache, and the same input in two chunks with the
carried cache (the cache the caller sees keeps the model's own channel count).

Runs only in the build container (needs /root/reference); the GPU box consumes the committed .npz.

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_shape_golden.py
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")

from tests.golden.make_golden import build_reference  # noqa: E402  (imports the reference)
from tests.golden.cases import SHAPE_CASES, shape_case_config  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402


def main():
    torch.set_num_threads(4)
    out = {}
    for case in SHAPE_CASES:
        cfg = shape_case_config(case)
        model, sd = build_reference(copy.deepcopy(cfg), case["wseed"])
        x = synth.synth_feats(case["B"], case["T"], cfg["input_dim"], seed=case["xseed"])
        with torch.no_grad():
            xt = torch.from_numpy(x)
            gru = cfg["backbone"]["type"] == "gru"            # (torch.nn.GRU wants its h0: the reference's streaming callers pass zeros)
            h0 = torch.zeros(cfg["backbone"]["num_layers"], case["B"], cfg["hidden_dim"]) if gru else None
            y, c = model(xt, h0) if gru else model(xt)
        name = case["name"]
        out[name + "/y"] = y.numpy().astype(np.float32)
        out[name + "/cache"] = c.numpy().astype(np.float32)
        out[name + "/wsum"] = np.float64(synth.checksum(sd))
        d = float("nan")
        if case.get("split"):                                 # (per-frame heads: the same input in two chunks)
            with torch.no_grad():
                t1 = case["split"]
                ya, ca = model(xt[:, :t1], h0) if gru else model(xt[:, :t1])
                yb, cb = model(xt[:, t1:], ca)
            out[name + "/y_stream"] = torch.cat([ya, yb], 1).numpy().astype(np.float32)
            out[name + "/cache_stream"] = cb.numpy().astype(np.float32)
            d = float((y - torch.cat([ya, yb], 1)).abs().max())
        print(f"{name:28s} y{tuple(y.shape)} cache{tuple(c.shape)} |y - y_stream|max = {d:.2e}")
    path = os.path.join(HERE, "shape_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(SHAPE_CASES), "cases; torch", torch.__version__)


if __name__ == "__main__":
    main()
