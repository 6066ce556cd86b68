#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): how close is bench.py's cpu_baseline "port" (oracle/torch_ref.py) to what it
stands for, the reference's own KWSModel.forward on PyTorch CPU?  Same weights, same batches, 1 / 4 / 8 threads; writes
profiles/r04_cpu_port_vs_reference.json.  (Round-3 review: the port was 8 .. 45 % slower than the real thing.)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from wekws.model.kws_model import init_model as ref_init_model  # noqa: E402  (the live reference)
from oracle import torch_ref  # noqa: E402
from wekws_amd import pack  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402


def rate(fn, nb, seconds=4.0):
    fn()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        fn()
        n += nb
    return n / (time.perf_counter() - t0)


def main():
    name, B, T = "ds_tcn_h256", 128, 98
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    ref = ref_init_model(dict(cfg))
    ref.load_state_dict(tsd, strict=False)
    ref.eval()
    x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=0))
    with torch.no_grad():
        yr, _ = ref(x)
        yp, _ = torch_ref.forward(cfg, tsd, x)
    out = {"model": name, "batch": B, "T": T, "max_abs_diff_port_vs_reference": float((yr - yp).abs().max()),
           "host_threads": os.cpu_count(), "rows": []}
    for th in (1, 4, 8):
        if th > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(th)
        with torch.no_grad():
            r = rate(lambda: ref(x), B)
            p = rate(lambda: torch_ref.forward(cfg, tsd, x), B)
        out["rows"].append({"threads": th, "reference_utts_per_s": round(r, 1), "port_utts_per_s": round(p, 1),
                            "port_over_reference": round(p / r, 3)})
        print(out["rows"][-1], flush=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r04_cpu_port_vs_reference.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
