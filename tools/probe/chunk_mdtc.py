#!/usr/bin/env python3
"""MDTC h64, 80-frame chunks with an incoming cache: the context variant of mdtc64_g4 against mdtc64_w16 (option g16 = 3) over B."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build, timeit  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

for name in ("mdtc_h64", "ds_tcn_h256"):
    cfg, a = build(name)
    _, b = build(name)
    b.set_option("g16", 3)
    for B in (1, 4, 16, 64, 256):
        for T in (80, 33):
            x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=1)).cuda()
            _, c = a(x)
            ta = timeit(lambda: a(x, c), warm=3, reps=12, group=10)[0]
            tb = timeit(lambda: b(x, c), warm=3, reps=12, group=10)[0]
            print(json.dumps(dict(model=name, B=B, T=T, ctx_ms=round(ta, 4), w16_ms=round(tb, 4), speedup=round(tb / ta, 3))), flush=True)
