#!/usr/bin/env python3
"""Generate tests/golden/generic_golden.npz from the LIVE reference: configurations init_model accepts (kws_model.py:114-170)
that no specialised kernel is built for -- wider than 256 channels (MDTC: 128), kernel sizes above 8 / 5, more residual blocks
than the block-floating kernels track, GRU hidden sizes above 128, more than 4 GRU layers, pooled heads on a GRU
(tests/golden/cases.py::GENERIC_CASES).  The library runs them on its any-shape exact-f32 path (wekws_amd/csrc/generic.hip.h).
Per case: one-shot forward without a cache and, for per-frame heads, the same input in two chunks with the carried cache.

Runs only in the build container (needs /root/reference); the GPU box consumes the committed .npz.

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_generic_golden.py
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
from tests.golden.cases import GENERIC_CASES, shape_case_config  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402


def main():
    torch.set_num_threads(4)
    out = {}
    for case in GENERIC_CASES:
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
    path = os.path.join(HERE, "generic_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(GENERIC_CASES), "cases; torch", torch.__version__)


if __name__ == "__main__":
    main()
