#!/usr/bin/env python3
"""Not a test: prints per-case HIP-vs-golden errors without stopping at the first failure (first GPU bring-up)."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import CASES, case_in_cache, case_input, case_weights, max_abs  # noqa: E402
from tests.test_hip_parity import build, run  # noqa: E402

golden = np.load(os.path.join(ROOT, "tests", "golden", "model_golden.npz"))
bad = 0
for case in CASES:
    try:
        cfg, sd = case_weights(case)
        model = build(cfg, sd).set_precision(os.environ.get('WEKWS_PRECISION', 'default'))
        y, cache = run(model, case_input(case), case_in_cache(case, cfg), softmax=case.get("softmax", False),
                       chunks=case.get("chunks"))
        gy, gc = golden[case["name"] + "/y"], golden[case["name"] + "/cache"]
        c = cache if cfg["backbone"]["type"] == "gru" else cache[:1]
        ey, ec = max_abs(y, gy), max_abs(c, gc)
        flag = "" if (ey <= 1e-4 * max(1, np.abs(gy).max()) and ec <= 1e-4 * max(1, np.abs(gc).max())) else "  <<<<<< FAIL"
        bad += bool(flag)
        print(f"{case['name']:36s} y_err={ey:.2e} cache_err={ec:.2e} |y|max={np.abs(gy).max():.3g}{flag}", flush=True)
    except Exception:
        bad += 1
        print(f"{case['name']:36s} EXCEPTION", flush=True)
        traceback.print_exc()
print("failures:", bad)
