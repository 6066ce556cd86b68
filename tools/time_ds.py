#!/usr/bin/env python3
"""Development tool: median kernel time of the DS-TCN h256 headline shape (B=1024, T=98) + a checksum of a small result, for
A/B runs of kernel variants on one box (tools/abvar.sh).   WEKWS_HIP_LIB=... python tools/time_ds.py [precision]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wekws_amd import pack  # noqa: E402
from wekws_amd.model.kws_model import init_model  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

name = os.environ.get("MODEL", "ds_tcn_h256")
cfg = dict(synth.MODEL_CONFIGS[name])
sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
m = init_model(cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda().eval().set_precision(sys.argv[1] if len(sys.argv) > 1 else "default").freeze()
xs = synth.synth_feats(3, 98, cfg["input_dim"], seed=5)
y, c = m(torch.from_numpy(xs).cuda())
err = float(y.double().abs().sum())      # checksum: variants of a kernel must agree on it (parity itself: pytest -m gpu)
x = torch.from_numpy(synth.synth_feats(1024, 98, cfg["input_dim"], seed=1)).cuda()
if os.environ.get("NOCACHE"):
    _m = m
    m = _m.posteriors
for _ in range(10):
    m(x)
ts = []
for _ in range(12):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        m(x)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 10)
print(f"{os.environ.get('WEKWS_HIP_LIB', 'product')}: median {np.median(ts):.4f} ms  min {min(ts):.4f}  checksum {err:.9f}")
