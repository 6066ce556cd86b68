#!/usr/bin/env python3
"""Generate tests/golden/hetero_golden.npz from the LIVE reference: the heterogeneous-scale rewrites of
tests/golden/cases.py::HETERO_CASES (per-channel power-of-two factors spread over 2^E inside matrices and operand tiles)
and the GRU's out-of-range-input cases.  Also asserts what makes the rewrites meaningful: the reference's own output does
not move under them.

Runs only in the build container (needs /root/reference); the GPU box consumes the committed .npz.

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_hetero_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")

from tests.golden.make_golden import build_reference  # noqa: E402  (imports the reference)
from tests.golden.make_scale_golden import run  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402
from tests.golden.cases import (GRU_INPUT_CASES, HETERO_CASES, case_config, case_input, case_in_cache,  # noqa: E402
                                hetero_case_weights)


def main():
    torch.set_num_threads(4)
    out, worst = {}, 0.0
    for case in HETERO_CASES:
        cfg = case_config(case)
        model, sd = build_reference(cfg, case["wseed"])
        x = case_input(case)
        cache0 = case_in_cache(case, cfg)
        y_base = run(model, x, cache0, case.get("chunks"))
        sd2 = hetero_case_weights(case, sd)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
        y = run(model, x, cache0, case.get("chunks"))
        assert np.isfinite(y).all(), case["name"]
        d = float(np.abs(y - y_base).max())          # exact power-of-two rewriting: the reference itself is invariant
        worst = max(worst, d)
        out[case["name"] + "/y"] = y.astype(np.float32)
        out[case["name"] + "/wsum"] = np.float64(synth.checksum(sd2))
        print(f"{case['name']:42s} y{tuple(y.shape)} [min,max]=[{y.min():.4g},{y.max():.4g}]  |y - y_unscaled|max={d:.2e}")
    assert worst <= 2e-6, worst
    for case in GRU_INPUT_CASES:
        cfg = case_config(case)
        model, sd = build_reference(cfg, case["wseed"])
        x = (case_input(case) * np.float32(case["xscale"])).astype(np.float32)
        y = run(model, x, case_in_cache(case, cfg), case.get("chunks"))
        assert np.isfinite(y).all(), case["name"]
        out[case["name"] + "/y"] = y.astype(np.float32)
        print(f"{case['name']:42s} y{tuple(y.shape)} [min,max]=[{y.min():.4g},{y.max():.4g}]")
    path = os.path.join(HERE, "hetero_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(HETERO_CASES) + len(GRU_INPUT_CASES), "cases; torch", torch.__version__)


if __name__ == "__main__":
    main()
