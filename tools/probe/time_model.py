#!/usr/bin/env python3
"""Development tool: forward time of the named models at B x 98 frames (HIP events, median of groups):  time_model.py [B] name..."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build, timeit  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

args = sys.argv[1:]
Bs = [int(a) for a in args if a.isdigit()] or [1024]
for name in [a for a in args if not a.isdigit()] or ["mdtc_h64"]:
    cfg, m = build(name)
    for B in Bs:
        x = torch.from_numpy(synth.synth_feats(B, 98, cfg["input_dim"], seed=1)).cuda()
        med, p10, p90 = timeit(lambda: m(x), warm=5, reps=25, group=20)
        print(json.dumps(dict(lib=os.environ.get("WEKWS_HIP_LIB", "product"), model=name, B=B, ms=round(med, 5), p10=round(p10, 5),
                              p90=round(p90, 5), utt_per_s=round(B / med * 1e3))), flush=True)
