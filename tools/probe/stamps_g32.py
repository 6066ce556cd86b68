#!/usr/bin/env python3
"""Per-phase clock64() sums of the exact-f32 register-resident DS-TCN kernel (ds256_g32.hip.h built with
-DWEKWS_G16_STAMPS by tools/abvar.sh):  WEKWS_HIP_LIB=build/var/libst32.so python tools/probe/stamps_g32.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wekws_amd import pack  # noqa: E402
from wekws_amd.model.kws_model import init_model  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
m = init_model(cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(pack.model_spec(cfg), 1234).items()})
m = m.cuda().eval().set_precision("f32").freeze()
names = ["pre", "block top", "depthwise -> planes", "barrier waits", "matrix phase", "epilogue", "cache hand-over", "head"]
x = torch.from_numpy(synth.synth_feats(1024, 98, 40, seed=1)).cuda()
for _ in range(50):
    y, c = m(x)
torch.cuda.synchronize()
d = c[0].flatten()[:128].cpu().numpy().reshape(16, 8)[[0, 9]].flatten()
for role, off in (("wave 0", 0), ("wave 9", 8)):
    print(role, "(sum over the utterances of workgroup 0)", " ".join(f"[{n}]={int(v)}" for n, v in zip(names, d[off:off + 8])),
          "total", int(d[off:off + 8].sum()))
