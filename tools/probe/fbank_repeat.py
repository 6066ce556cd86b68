#!/usr/bin/env python3
"""Development tool: is fbank_kernel deterministic under load?  The same 1024 / 8192 x 1 s batch 40 times, every output compared
bit for bit with the first run's, and with the same utterances run in batches of 8 (other workgroup / wave assignment)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wekws_amd.frontend import Fbank  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

dev = torch.device("cuda", 0)
for B in (1024, 8192):
    fb = Fbank(num_bins=40, device=dev)
    pcm = torch.from_numpy(synth.synth_pcm(B, 16000, seed=3, kind="noise")).to(dev)
    first = fb(pcm).clone()
    torch.cuda.synchronize()
    nbad = 0
    for rep in range(40):
        y = fb(pcm)
        nbad += int((y.view(torch.int32) != first.view(torch.int32)).any(dim=-1).sum().item())
    small = torch.cat([fb(pcm[i:i + 8]).clone() for i in range(0, min(B, 1024), 8)])
    d = int((small.view(torch.int32) != first[:small.shape[0]].view(torch.int32)).any(dim=-1).sum().item())
    print(f"B={B}: frames differing from the first run over 40 repeats: {nbad}; frames differing batch-of-8 vs full batch: {d}", flush=True)
