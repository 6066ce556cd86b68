#!/usr/bin/env python3
"""Development tool: 200 steps of 4096 DS-TCN h256 streams x one 10-frame chunk with the carried cache -- the workload of
bench.py's `rooflines_other[ds256_stream_kernel]` -- for rocprofv3 counter passes:
    tools/pmc.sh strm python tools/probe/stream4096.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

cfg, m = build("ds_tcn_h256")
x = torch.from_numpy(synth.synth_feats(4096, 10, 40, seed=2)).cuda()
_, c = m(x)
for _ in range(200):
    _, c = m(x, c)
torch.cuda.synchronize()
print("done", float(c.abs().max()))
