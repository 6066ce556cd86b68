#!/usr/bin/env python3
"""GRU 2x128 wavefront: one library build per process (WEKWS_HIP_LIB), equality with the layer-major kernels first, then times.
    WEKWS_HIP_LIB=build/var/libgl2.so python tools/probe/gru_l2_ab.py <tag>      -> one JSON line"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build, timeit  # noqa: E402
from wekws_amd import _capi  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
    cfg, pipe = build("gru_2x128")
    _, major = build("gru_2x128")
    pipe.set_option("gru_pipe", 2)
    major.set_option("gru_pipe", 0)
    row = {"tag": tag, "lib": os.path.basename(_capi.lib_path())}
    bad = 0
    for B, T, it in ((1024, 98, 30), (4096, 98, 5), (37, 40, 10)):
        x = torch.from_numpy(synth.synth_feats(B, T, 40, seed=B)).cuda()
        y0, c0 = major(x)
        for _ in range(it):
            y1, c1 = pipe(x)
            bad += int(not (torch.equal(y1, y0) and torch.equal(c1, c0)))
    x = torch.from_numpy(synth.synth_feats(1, 10, 40, seed=3)).cuda()
    y0, c0 = major(x)
    ym, cm = major(x, c0)
    for _ in range(50):
        y1, c1 = pipe(x, c0)
        bad += int(not (torch.equal(y1, ym) and torch.equal(c1, cm)))
    row["mismatches"] = bad
    try:
        pipe.check()
        row["health"] = "ok"
    except Exception as e:
        row["health"] = str(e)[:200]
    for B, T in ((1024, 98), (4096, 98), (256, 98), (1, 10), (256, 10)):
        x = torch.from_numpy(synth.synth_feats(B, T, 40, seed=1)).cuda()
        h = torch.zeros(2, B, 128, device="cuda")
        fn = (lambda: pipe(x, h)) if T <= 16 else (lambda: pipe(x))
        med, p10, p90 = timeit(fn, warm=3, reps=15, group=10 if B < 4096 else 3)
        row[f"B{B}xT{T}_ms"] = round(med, 5)
        if T > 16:
            row[f"B{B}xT{T}_Mutt_s"] = round(B / med / 1e3, 3)
        else:
            row[f"B{B}xT{T}_us_per_frame"] = round(med * 1e3 / T, 3)
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
