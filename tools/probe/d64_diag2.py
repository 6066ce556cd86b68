#!/usr/bin/env python3
"""Development tool: ds64_g4 with spills -- are posteriors-only calls (out_cache = NULL) affected too?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_hip_parity import build  # noqa: E402
from wekws_amd import pack  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h64"])
cfg["output_dim"] = 2
sd = synth.synth_state_dict(pack.model_spec(cfg), 2024)
fast, slow = build(cfg, sd), build(cfg, sd).set_option("g16", 0)
B, T = 4096, 98
x = torch.from_numpy(synth.synth_feats(B, T, 40, seed=7 * B + T)).cuda()
ys = slow.posteriors(x)
torch.cuda.synchronize()
for rep in range(3):
    y = fast.posteriors(x)
    torch.cuda.synchronize()
    nb = int(((y - ys).abs() > 1e-5).any(dim=2).any(dim=1).sum())
    print(f"posteriors only rep={rep}: {nb} utterances differ")
for rep in range(3):
    y, c = fast(x)
    torch.cuda.synchronize()
    nb = int(((y - ys).abs() > 1e-5).any(dim=2).any(dim=1).sum())
    print(f"with cache out rep={rep}: {nb} utterances differ")
