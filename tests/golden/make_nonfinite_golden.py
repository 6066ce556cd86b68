#!/usr/bin/env python3
"""Generate tests/golden/nonfinite_golden.npz from the LIVE reference (build container only): what KWSModel.forward returns for
features / caches that hold NaN, +Inf or -Inf.  Stored per case: y (float32, NaNs and all) and, for the returned cache, the
class of every element (0 finite, 1 NaN, 2 +Inf, 3 -Inf; int8 -- compresses to nothing) plus its finite values' checksum.

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_nonfinite_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")

from tests.golden.make_golden import build_reference  # noqa: E402  (imports the reference)
from tests.golden.nonfinite_cases import CASES, case_config, classify, poisoned_input  # noqa: E402


def run_reference(model, case, x, cache0):
    fwd = model.forward_softmax if case.get("softmax") else model.forward
    with torch.no_grad():
        xt = torch.from_numpy(x)
        cache = torch.from_numpy(cache0) if cache0 is not None else None
        if case.get("chunks"):
            ys, t = [], 0
            for n in case["chunks"]:
                y, cache = fwd(xt[:, t:t + n]) if cache is None else fwd(xt[:, t:t + n], cache)
                ys.append(y)
                t += n
            y = torch.cat(ys, dim=1)
        else:
            y, cache = fwd(xt) if cache is None else fwd(xt, cache)
    return y.numpy().astype(np.float32), cache.numpy().astype(np.float32)


def main():
    torch.set_num_threads(4)
    out = {}
    for case in CASES:
        cfg = case_config(case)
        model, _ = build_reference(cfg, case["wseed"])
        x, cache0 = poisoned_input(case, cfg)
        y, cache = run_reference(model, case, x, cache0)
        cy, cc = classify(y), classify(cache)
        out[case["name"] + "/y"] = y
        out[case["name"] + "/cache_class"] = cc
        out[case["name"] + "/cache_finite_sum"] = np.float64(np.abs(np.where(cc == 0, cache, 0).astype(np.float64)).sum())
        print(f"{case['name']:34s} y{tuple(y.shape)} classes y: nan {int((cy == 1).sum())} +inf {int((cy == 2).sum())} "
              f"-inf {int((cy == 3).sum())} | cache: nan {int((cc == 1).sum())} +inf {int((cc == 2).sum())} -inf {int((cc == 3).sum())}")
    path = os.path.join(HERE, "nonfinite_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(CASES), "cases; torch", torch.__version__)


if __name__ == "__main__":
    main()
