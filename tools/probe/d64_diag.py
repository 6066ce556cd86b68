#!/usr/bin/env python3
"""Development tool: where do ds64_g4's posteriors differ from the LDS-tile kernel's in a large batch?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_hip_parity import build, run  # noqa: E402
from wekws_amd import pack  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h64"])
cfg["output_dim"] = 2
sd = synth.synth_state_dict(pack.model_spec(cfg), 2024)
fast, slow = build(cfg, sd), build(cfg, sd).set_option("g16", 0)
for B, T in ((1030, 98), (4096, 98), (301, 98)):
    x = synth.synth_feats(B, T, 40, seed=7 * B + T)
    ys, cs = run(slow, x)
    for rep in range(3):
        y, c = run(fast, x)
        d = np.abs(y - ys)
        bad = np.argwhere(d > 1e-5)
        bs = sorted(set(bad[:, 0].tolist()))
        print(f"B={B} rep={rep}: {len(bs)} utterances differ; first {bs[:12]}; cache max diff {np.abs(c - cs).max():.1e}")
        for b in bs[:4]:
            tt = sorted(set(bad[bad[:, 0] == b][:, 1].tolist()))
            kk = sorted(set(bad[bad[:, 0] == b][:, 2].tolist()))
            print(f"   b={b} (b%256={b % 256}): frames {tt[:20]}{'...' if len(tt) > 20 else ''} n={len(tt)} outputs {kk} max {d[b].max():.2e}")
