#!/usr/bin/env python3
"""Per-phase clock64() sums of the register-resident DS-TCN kernel (ds256_g16.hip.h built with -DWEKWS_G16_STAMPS by
tools/abvar.sh):  WEKWS_HIP_LIB=build/var/libst.so python tools/probe/stamps_g16.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

cfg, m = build("ds_tcn_h256")
m.set_option("g16", 1)
names = ["pre", "block top", "depthwise -> planes", "barrier waits", "matrix phase", "epilogue", "cache hand-over", "head"]
x = torch.from_numpy(synth.synth_feats(1024, 98, 40, seed=1)).cuda()
for _ in range(50):
    y, c = m(x)
torch.cuda.synchronize()
d = c[0].flatten()[:128].cpu().numpy().reshape(16, 8)
print("sums over the utterances of workgroup 0 (/ 4 at B = 1024); columns:", ", ".join(names))
for w in range(16):
    print(f"wave {w:2d}", " ".join(f"{int(v):7d}" for v in d[w]), "total", int(d[w].sum()))
print("max    ", " ".join(f"{int(v):7d}" for v in d.max(0)))
print("min    ", " ".join(f"{int(v):7d}" for v in d.min(0)))
