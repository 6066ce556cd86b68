#!/usr/bin/env python3
"""Development tool: the fbank kernel beside tenants on another stream -- which tenants change its results, and where (frames, bins)?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_hip_parity import build  # noqa: E402
from wekws_amd import pack  # noqa: E402
from wekws_amd.frontend import Fbank  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

cfg = dict(synth.MODEL_CONFIGS["mdtc_h64"])
model = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 11))
x = torch.from_numpy(synth.synth_feats(512, 98, cfg["input_dim"], seed=4)).cuda()
pcm = torch.from_numpy(synth.synth_pcm(768, 16000, seed=5, kind="noise")).cuda()
fb = Fbank(40)
ref = fb(pcm).clone()
big = torch.randn(64 * 1024 * 1024, device="cuda")
ma = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
torch.cuda.synchronize()
s_fb, s_t = torch.cuda.Stream(), torch.cuda.Stream()


def tenant_none():
    pass


def tenant_mdtc():
    for _ in range(6):
        model(x)


def tenant_elementwise():
    for _ in range(3):
        big.mul_(1.0001)


def tenant_matmul():
    for _ in range(3):
        torch.matmul(ma, ma)


for name, tenant in (("none", tenant_none), ("mdtc_h64 forward", tenant_mdtc), ("elementwise", tenant_elementwise), ("fp16 matmul", tenant_matmul)):
    bad_frames, bins, lanes = 0, np.zeros(40, np.int64), None
    maxd = 0.0
    for _ in range(20):
        with torch.cuda.stream(s_t):
            tenant()
        with torch.cuda.stream(s_fb):
            got = [fb(pcm) for _ in range(4)]
        torch.cuda.synchronize()
        for g in got:
            d = g.view(torch.int32) != ref.view(torch.int32)
            bad_frames += int(d.any(dim=-1).sum().item())
            bins += d.sum(dim=(0, 1)).cpu().numpy()
            maxd = max(maxd, float((g - ref).abs().max().item()))
    print(f"tenant {name:18s}: {bad_frames:8d} frames differ of {20 * 4 * 768 * 98}; max |diff| {maxd:.3e}; per mel bin: {bins.tolist()}", flush=True)
