#!/usr/bin/env python3
"""Development tool: fbank_kernel time per 1024 x 1 s (HIP events, many calls)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wekws_amd.frontend import Fbank  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

dev = torch.device("cuda", 0)
NB = int(os.environ.get("FBANK_BINS", "40"))
for B in (1024, 8192):
    fb = Fbank(num_bins=NB, device=dev)
    pcm = torch.from_numpy(synth.synth_pcm(B, 16000, seed=0, kind="noise")).to(dev)
    for _ in range(200):
        fb(pcm)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fb(pcm)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 50 * 1e3)
    print(os.environ.get("WEKWS_HIP_LIB", "product"), f"bins={NB} B={B}: median {np.median(ts):.1f} us  min {min(ts):.1f}", flush=True)
