#!/usr/bin/env python3
"""Chunks of 80 frames (the Android caller's size, runtime/android/app/src/main/cpp/wekws.cc:84-97): the first chunk (no incoming
cache: register-resident kernels) against the following ones (incoming cache: LDS-tile kernels), DS-TCN h256 and MDTC h64."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build, timeit  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

for name in ("ds_tcn_h256", "mdtc_h64"):
    cfg, m = build(name)
    for B in (1, 256, 1024):
        x = torch.from_numpy(synth.synth_feats(B, 80, cfg["input_dim"], seed=1)).cuda()
        _, c = m(x)
        first = timeit(lambda: m(x), warm=3, reps=15, group=10)[0]
        cont = timeit(lambda: m(x, c), warm=3, reps=15, group=10)[0]
        print(json.dumps(dict(model=name, B=B, T=80, first_chunk_ms=round(first, 4), with_cache_ms=round(cont, 4),
                              ratio=round(cont / first, 3))), flush=True)
