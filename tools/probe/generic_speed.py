#!/usr/bin/env python3
"""Throughput of the any-shape path (csrc/generic.hip.h) on a few configurations beyond the specialised kernels; one JSON line each."""
import copy
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import timeit  # noqa: E402
from wekws_amd import pack  # noqa: E402
from wekws_amd.model.kws_model import init_model  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

CASES = [("ds_tcn_h256", dict(hidden=512), 1024), ("ds_tcn_h256", dict(hidden=320), 1024), ("mdtc_h64", dict(hidden=256), 1024),
         ("gru_2x128", dict(hidden=256), 256), ("fsmn_ctc", dict(f32=True), 256), ("tcn_h64", dict(hidden=288), 256)]
for name, over, B in CASES:
    cfg = copy.deepcopy(synth.MODEL_CONFIGS[name])
    if over.get("hidden"):
        cfg["hidden_dim"] = over["hidden"]
        if "hidden_dim" in cfg["backbone"]:
            cfg["backbone"]["hidden_dim"] = over["hidden"]
    m = init_model(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(pack.model_spec(cfg), 1).items()})
    m = m.cuda().eval()
    if over.get("f32"):
        m.set_precision("f32")
    m.freeze()
    T = 32 if name.startswith("fsmn") else 98
    x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=1)).cuda()
    med, p10, p90 = timeit(lambda: m(x), warm=2, reps=5, group=2)
    print(json.dumps(dict(model=name, over=over, B=B, T=T, precision=m.effective_precision(), ms=round(med, 3),
                          utts_per_s=round(B / med * 1e3, 1))), flush=True)
