#!/usr/bin/env python3
"""GRU 2x128: the layer wavefront (gru_pipe.hip.h) against the layer-major kernels (option gru_pipe = 0), same box, same
process: batch throughput and streaming-chunk latency.  One JSON line per shape."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wekws_amd import _capi  # noqa: E402
if os.environ.get('WEKWS_DBG_LIB'):
    _capi._LIB_PATH = os.environ['WEKWS_DBG_LIB']
from tools.bench_configs import build, timeit  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402


def main():
    names = sys.argv[1:] or ["gru_2x128"]
    for name in names:
        cfg, m = build(name)
        L = cfg["backbone"]["num_layers"]
        for B, T in ((1, 10), (1, 10), (256, 10), (1024, 98)):
            x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=1)).cuda()
            h = torch.zeros(L, B, 128, device="cuda")
            row = dict(model=name, B=B, T=T)
            for tag, opt in (("pipe", 2), ("major", 0)):
                m.set_option("gru_pipe", opt)
                if T <= 16:                                   # streaming: carried state
                    def step():
                        m(x, h)
                else:
                    def step():
                        m(x)
                med, p10, p90 = timeit(step, warm=3, reps=15, group=10 if B < 4096 else 3)
                row[tag + "_ms"] = round(med, 5)
                row[tag + "_utt_per_s"] = round(B / med * 1e3)
            row["speedup"] = round(row["major_ms"] / row["pipe_ms"], 3)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
