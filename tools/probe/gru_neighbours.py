#!/usr/bin/env python3
"""What slows a GRU wavefront step from ~1.3 us (64 CUs busy: B = 256) to ~1.7 us (all 256 CUs: B = 1024)?  The B = 256 call
alone, next to 190 workgroups that only HOLD their CUs (asleep), and next to 190 workgroups that keep every SIMD busy with
matrix + vector instructions and no memory traffic.  Needs the hooks library:
    WEKWS_HIP_LIB=wekws_amd/lib/libwekws_hip_hooks.so python tools/probe/gru_neighbours.py"""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build  # noqa: E402
from wekws_amd import _capi  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

lib = _capi.load()
lib.wekws_hip_debug_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.wekws_hip_debug_hog.restype = ctypes.c_int
cfg, pipe = build("gru_2x128")
pipe.set_option("gru_pipe", 2)
side = torch.cuda.Stream()
for B in (256, 1024):
    x = torch.from_numpy(synth.synth_feats(B, 98, 40, seed=1)).cuda()
    for _ in range(200):
        pipe(x)
    torch.cuda.synchronize()
    for mode, ms in (("alone", 0), ("asleep", 60), ("busy", -60)):
        if B == 1024 and mode != "alone":
            continue
        if ms:
            assert lib.wekws_hip_debug_hog(0, 190, ms, ctypes.c_void_p(side.cuda_stream)) == 0, _capi.last_error()
            time.sleep(0.01)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            pipe(x)
        b.record()
        torch.cuda.synchronize()
        print(json.dumps(dict(B=B, neighbours=mode, ms_per_call=round(a.elapsed_time(b) / 100, 4))), flush=True)
pipe.check()
