#!/usr/bin/env python3
"""Development tool: what a NaN / Inf feature does to the posteriors (the reference propagates it: torch.relu(nan) = nan)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

for name in ("ds_tcn_h256", "mdtc_h64", "gru_2x128", "ds_tcn_h64"):
    cfg, m = build(name)
    for val in (float("nan"), float("inf")):
        x = synth.synth_feats(4, 98, cfg["input_dim"], seed=1)
        x[1, 40, 7] = val
        xt = torch.from_numpy(x).cuda()
        y, c = m(xt) if name != "gru_2x128" else m(xt, torch.zeros(2, 4, 128, device="cuda"))
        y = y.cpu().numpy()
        fin = [bool(np.isfinite(y[b]).all()) for b in range(4)]
        print(name, val, "finite per utterance:", fin, "utt1 frames non-finite:", int((~np.isfinite(y[1])).any(axis=-1).sum()))
