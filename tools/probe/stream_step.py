#!/usr/bin/env python3
"""Development tool: one 10-frame chunk with the carried cache on the DS-TCN h256 streaming-step kernel, B = 1 / 256 / 4096
(median of 15 groups of 20 steps between two events).  WEKWS_HIP_LIB selects the library: A/B of tools/abvar.sh variants."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

cfg, m = build(sys.argv[1] if len(sys.argv) > 1 else "ds_tcn_h256")
out = []
for B in (1, 256, 4096):
    x = torch.from_numpy(synth.synth_feats(B, 10, cfg["input_dim"], seed=2)).cuda()
    _, c = m(x)
    for _ in range(100):
        _, c = m(x, c)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            _, c = m(x, c)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20)
    ts.sort()
    out.append(f"B={B}: {ts[len(ts) // 2] * 1e3:.1f} us (min {ts[0] * 1e3:.1f})")
print(os.environ.get("WEKWS_HIP_LIB", "product"), " | ".join(out), float(c.abs().max()))
