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
ts = []
for _ in range(15):                                       # (timing: 15 groups of 20 steps between two events)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        _, c = m(x, c)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 20)
ts.sort()
print(os.environ.get("WEKWS_HIP_LIB", "product"), f"4096 streams x 10 frames: median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f}", float(c.abs().max()))
