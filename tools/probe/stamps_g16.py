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
d = c[0].flatten()[:16].cpu().numpy()
for role, off in (("wave 0", 0), ("wave 9", 8)):
    print(role, "(sum over the utterances of workgroup 0; / 4 at B = 1024)", " ".join(f"[{n}]={int(v)}" for n, v in zip(names, d[off:off + 8])), "total", int(d[off:off + 8].sum()))
