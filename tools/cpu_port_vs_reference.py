#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): how close is bench.py's cpu_baseline "port" (oracle/torch_ref.py) to what it
stands for, the reference's own KWSModel.forward on PyTorch CPU?  Same weights, same batches, 1 / 4 / 8 threads; writes
profiles/r06_cpu_port_vs_reference.json (DS-TCN h256, MDTC h64, GRU 2x128).  (Round-3 review: the port was 8 .. 45 % slower than the real thing.)"""
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


def one(name, B, T, threads):
    cfg = dict(synth.MODEL_CONFIGS[name])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    ref = ref_init_model(dict(cfg))
    ref.load_state_dict(tsd, strict=False)
    ref.eval()
    x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=0))
    # the GRU needs an explicit h0 in the reference (the default in_cache raises)
    h0 = torch.zeros(cfg["backbone"]["num_layers"], B, cfg["hidden_dim"]) if cfg["backbone"]["type"] == "gru" else None
    call_ref = (lambda: ref(x, h0)) if h0 is not None else (lambda: ref(x))
    call_port = (lambda: torch_ref.forward(cfg, tsd, x, h0)) if h0 is not None else (lambda: torch_ref.forward(cfg, tsd, x))
    with torch.no_grad():
        yr, cr = call_ref()
        yp, cp = call_port()
    out = {"model": name, "batch": B, "T": T, "max_abs_diff_port_vs_reference": float((yr - yp).abs().max()),
           "max_abs_diff_cache": float((cr - cp).abs().max()), "rows": []}
    for th in threads:
        torch.set_num_threads(th)
        with torch.no_grad():
            r = rate(call_ref, B)
            p = rate(call_port, B)
        out["rows"].append({"threads": th, "reference_utts_per_s": round(r, 1), "port_utts_per_s": round(p, 1),
                            "port_over_reference": round(p / r, 3)})
        print(name, out["rows"][-1], flush=True)
    return out


def main():
    threads = [t for t in (1, 4, 8) if t <= (os.cpu_count() or 1)]
    out = {"what": "bench.py's cpu_baseline port (oracle/torch_ref.py) vs the live reference KWSModel.forward "
                   "(/root/reference/wekws/model/kws_model.py:65-76) on the build container, same weights and batches",
           "host_threads": os.cpu_count(), "torch": torch.__version__,
           "models": [one(n, 128, 98, threads) for n in ("ds_tcn_h256", "mdtc_h64", "gru_2x128")]}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r06_cpu_port_vs_reference.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
