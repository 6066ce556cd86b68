#!/usr/bin/env python3
"""Generate tests/golden/scale_golden.npz from the LIVE reference: the operand-scale sweeps of
tests/golden/cases.py::SCALE_CASES (models rewritten with exact power-of-two factors, see scale_state_dict).

Runs only in the build container (needs /root/reference); the GPU box consumes the committed .npz.

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_scale_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")

from tests.golden.make_golden import build_reference  # noqa: E402  (imports the reference)
from wekws_amd.utils import synth  # noqa: E402
from tests.golden.cases import SCALE_CASES, case_config, case_input, case_in_cache, scaled_case_weights  # noqa: E402


def run(model, x, cache0, chunks):
    with torch.no_grad():
        xt = torch.from_numpy(x)
        cache = torch.from_numpy(cache0) if cache0 is not None else None
        if chunks:
            ys, t = [], 0
            for n in chunks:
                y, cache = model(xt[:, t:t + n]) if cache is None else model(xt[:, t:t + n], cache)
                ys.append(y)
                t += n
            return torch.cat(ys, dim=1).numpy()
        y, _ = model(xt) if cache is None else model(xt, cache)
        return y.numpy()


def main():
    torch.set_num_threads(4)
    out, worst = {}, 0.0
    for case in SCALE_CASES:
        cfg = case_config(case)
        model, sd = build_reference(cfg, case["wseed"])
        x = case_input(case)
        cache0 = case_in_cache(case, cfg)
        y_base = run(model, x, cache0, case.get("chunks"))
        sd2, xs = scaled_case_weights(case, sd)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
        y = run(model, (x * np.float32(xs)).astype(np.float32), cache0, case.get("chunks"))
        assert np.isfinite(y).all(), case["name"]
        d = float(np.abs(y - y_base).max())          # exact power-of-two rewriting: the reference itself is invariant
        worst = max(worst, d)
        out[case["name"] + "/y"] = y.astype(np.float32)
        out[case["name"] + "/wsum"] = np.float64(synth.checksum(sd2))
        print(f"{case['name']:42s} y{tuple(y.shape)} [min,max]=[{y.min():.4g},{y.max():.4g}]  |y - y_unscaled|max={d:.2e}")
    assert worst <= 2e-6, worst
    path = os.path.join(HERE, "scale_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(SCALE_CASES), "cases; torch", torch.__version__)


if __name__ == "__main__":
    main()
