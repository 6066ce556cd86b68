#!/usr/bin/env python3
"""Build-container only: turn the one trained model the reference ships (runtime/android/app/src/main/assets/kws.ort)
into build/ref_asset/{kws.wekwship, expect.npz} for tests/test_hip_parity.py::test_reference_android_asset_hip.
build/ is git-ignored but travels to the GPU box, where /root/reference does not exist.  expect.npz = the graph run by
the numpy executor (oracle/onnx_graph_oracle.py) on a seeded 3-chunk stream."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import onnx_graph_oracle  # noqa: E402
from wekws_amd.bin import export_packed  # noqa: E402
from wekws_amd.utils import onnx_model  # noqa: E402

SRC = "/root/reference/runtime/android/app/src/main/assets/kws.ort"


def main():
    out = os.path.join(ROOT, "build", "ref_asset")
    os.makedirs(out, exist_ok=True)
    export_packed.main(["--exported", SRC, "--output", os.path.join(out, "kws.wekwship")])
    g = onnx_model.load_graph(SRC)
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((1, 240, 40)) * 3 + 10).astype(np.float32)
    cache = np.zeros((1, int(g.meta["cache_dim"]), int(g.meta["cache_len"])), np.float32)
    ys = []
    for t in range(0, 240, 80):                                   # the Android app feeds 80-frame chunks
        o = onnx_graph_oracle.run(g, dict(input=x[:, t:t + 80], cache=cache))
        ys.append(o["output"])
        cache = o["r_cache"]
    np.savez(os.path.join(out, "expect.npz"), x=x, y=np.concatenate(ys, 1), cache=cache)
    print("wrote", out)


if __name__ == "__main__":
    main()
