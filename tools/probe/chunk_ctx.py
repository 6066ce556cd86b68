#!/usr/bin/env python3
"""DS-TCN h256, chunks WITH an incoming cache: time per call for a library variant (WEKWS_HIP_LIB); first chunk beside it."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build, timeit  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
cfg, m = build("ds_tcn_h256")
row = {"tag": tag}
for B, T in ((1024, 98), (1024, 80), (1, 80)):
    x = torch.from_numpy(synth.synth_feats(B, T, 40, seed=1)).cuda()
    _, c = m(x)
    row[f"B{B}T{T}_first"] = round(timeit(lambda: m(x), warm=3, reps=12, group=10)[0], 4)
    row[f"B{B}T{T}_cache"] = round(timeit(lambda: m(x, c), warm=3, reps=12, group=10)[0], 4)
print(json.dumps(row), flush=True)
