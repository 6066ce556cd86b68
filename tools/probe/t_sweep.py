#!/usr/bin/env python3
"""DS-TCN h256, B = 1024: time per call over chunk lengths around multiples of the lane width (7 frames): first chunk and with cache."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build, timeit
from wekws_amd.utils import synth
cfg, m = build("ds_tcn_h256")
for T in (70, 77, 80, 84, 91, 98, 100, 112):
    x = torch.from_numpy(synth.synth_feats(1024, T, 40, seed=1)).cuda()
    _, c = m(x)
    f = timeit(lambda: m(x), warm=3, reps=10, group=10)[0]
    w = timeit(lambda: m(x, c), warm=3, reps=10, group=10)[0]
    p = timeit(lambda: m.posteriors(x), warm=3, reps=10, group=10)[0]
    print(json.dumps(dict(T=T, first_ms=round(f, 4), cache_ms=round(w, 4), posteriors_only_ms=round(p, 4))), flush=True)
