#!/usr/bin/env python3
"""Run one synthetic model a few times on cuda:0 (profiling target for rocprofv3).
    python tools/run_model.py fsmn_ctc 1024 33 [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wekws_amd import pack  # noqa: E402
from wekws_amd.model.kws_model import init_model  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

name, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 8
cfg = dict(synth.MODEL_CONFIGS[name])
m = init_model(cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(pack.model_spec(cfg), 1234).items()})
m = m.cuda().eval()
x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=1)).cuda()
for _ in range(iters):
    y, c = m(x)
torch.cuda.synchronize()
print(name, tuple(y.shape), float(y.abs().mean()))
