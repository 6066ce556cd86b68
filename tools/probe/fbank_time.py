#!/usr/bin/env python3
"""fbank kernel time for a library variant (WEKWS_DBG_LIB) and its error against the default library's features."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wekws_amd import _capi
if os.environ.get("WEKWS_DBG_LIB"):
    _capi._LIB_PATH = os.environ["WEKWS_DBG_LIB"]
from tools.bench_configs import timeit
from wekws_amd.frontend import Fbank
from wekws_amd.utils import synth
fb = Fbank(40)
for B in (1024, 8192):
    pcm = torch.from_numpy(synth.synth_pcm(B, 16000, seed=3)).cuda()
    med, p10, p90 = timeit(lambda: fb(pcm), warm=3, reps=15, group=10)
    f = fb(pcm); torch.cuda.synchronize()
    print(json.dumps(dict(lib=os.environ.get("WEKWS_DBG_LIB", "default"), B=B, ms=round(med, 5), checksum=float(f.double().sum()))), flush=True)
