#!/usr/bin/env python3
"""Development tool: A/B kernel time of the DS-TCN h256 headline shape (B=1024, T=98) for several per-model option sets in ONE
process, interleaved round-robin after a common preheat, so that clock ramp and box-to-box drift cancel.
    python tools/time_ds.py [precision] -- "" roles=4 "roles=4,w16=1"
Each option set is a comma-separated list for KWSModel.set_option.  Prints the median / min ms per set and a checksum of a
small result (variants of a kernel must agree on it; parity itself: pytest -m gpu).  NOCACHE=1 times posteriors only.
WEKWS_HIP_LIB selects a variant library (tools/abvar.sh)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wekws_amd import pack  # noqa: E402
from wekws_amd.model.kws_model import init_model  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

args = sys.argv[1:]
precision = "default"
if "--" in args:
    k = args.index("--")
    if k > 0:
        precision = args[0]
    optsets = args[k + 1:]
else:
    if args:
        precision = args[0]
    optsets = [os.environ.get("OPTS", "")]
name = os.environ.get("MODEL", "ds_tcn_h256")
B = int(os.environ.get("BATCH", "1024"))
cfg = dict(synth.MODEL_CONFIGS[name])
sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
xs = torch.from_numpy(synth.synth_feats(3, 98, cfg["input_dim"], seed=5)).cuda()
x = torch.from_numpy(synth.synth_feats(B, 98, cfg["input_dim"], seed=1)).cuda()
models, sums = [], []
for opts in optsets:
    m = init_model(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda().eval().set_precision(precision).freeze()
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        m.set_option(k, int(v))
    y, c = m(xs)
    sums.append(float(y.double().abs().sum()) + float(c.double().abs().sum()) * 1e-3)
    models.append(m.posteriors if os.environ.get("NOCACHE") else m)
t0 = time.time()
while time.time() - t0 < 0.5:                       # preheat: the GPU's clocks ramp for ~0.3 s out of idle
    for m in models:
        for _ in range(20):
            m(x)
    torch.cuda.synchronize()
ts = [[] for _ in models]
for _ in range(15):
    for i, m in enumerate(models):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            m(x)
        b.record()
        torch.cuda.synchronize()
        ts[i].append(a.elapsed_time(b) / 20)
lib = os.environ.get("WEKWS_HIP_LIB", "product")
for opts, t, s in zip(optsets, ts, sums):
    print(f"{lib} [{opts}]: median {np.median(t):.4f} ms  min {min(t):.4f}  checksum {s:.9f}")
