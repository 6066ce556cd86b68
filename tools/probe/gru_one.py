#!/usr/bin/env python3
"""30 wavefront forwards of GRU 2x128 at B = 1024 x 98 frames and nothing else (for rocprofv3 --pmc passes: one kernel shape)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

cfg, pipe = build("gru_2x128")
pipe.set_option("gru_pipe", 2)
x = torch.from_numpy(synth.synth_feats(1024, 98, 40, seed=1)).cuda()
for _ in range(30):
    pipe(x)
torch.cuda.synchronize()
pipe.check()
