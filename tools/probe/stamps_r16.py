#!/usr/bin/env python3
"""Per-phase clock64() sums of the role-split DS-TCN kernel (ds256_r16.hip.h built with -DWEKWS_R16_STAMPS by
tools/abvar.sh):  WEKWS_HIP_LIB=build/var/libst.so python tools/probe/stamps_r16.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

cfg, m = build("ds_tcn_h256")
m.set_option("roles", 1)
names = ["pre", "block top", "wait first K step", "work (mfma | produce)", "wait in K loop", "epilogue", "wait epilogue", "head"]
x = torch.from_numpy(synth.synth_feats(1024, 98, 40, seed=1)).cuda()
for _ in range(50):
    y, c = m(x)
torch.cuda.synchronize()
d = c[0].flatten()[:16].cpu().numpy()
for role, off in (("M-wave 0", 0), ("P-wave 8", 8)):
    print(role, " ".join(f"[{n}]={int(v)}" for n, v in zip(names, d[off:off + 8])), "total", int(d[off:off + 8].sum()))
