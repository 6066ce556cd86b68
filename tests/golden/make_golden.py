#!/usr/bin/env python3
"""Generate tests/golden/model_golden.npz from the LIVE reference.

Runs only in the build container (needs /root/reference on PYTHONPATH); the GPU
box consumes the committed .npz.  For every case below it builds the reference
``wekws.model.kws_model.init_model(cfg)`` (PyTorch CPU fp32, eval), loads the
deterministic synthetic state_dict from ``wekws_amd.utils.synth`` and records
the reference outputs.  Inputs and weights are NOT stored: tests regenerate
them from the same seeds (a checksum of each is stored to catch RNG drift).

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_golden.py
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")

from wekws.model.kws_model import init_model  # noqa: E402  (the reference)
from wekws_amd.utils import synth  # noqa: E402
from tests.golden.cases import CASES, case_config, case_input, case_in_cache  # noqa: E402


def build_reference(cfg, seed):
    cfg = dict(cfg)
    with contextlib.redirect_stdout(io.StringIO()):  # MDTC prints its receptive field
        model = init_model(cfg)
    if cfg.get("_cmvn"):
        from wekws.model.cmvn import GlobalCMVN
        idim = cfg["input_dim"]
        model.global_cmvn = GlobalCMVN(torch.zeros(idim), torch.ones(idim), cfg["cmvn"]["norm_var"])
    sd = synth.synth_state_dict(synth.module_spec(model), seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    return model, sd


def main():
    torch.set_num_threads(4)
    out = {}
    for case in CASES:
        name = case["name"]
        cfg = case_config(case)
        model, sd = build_reference(cfg, case["wseed"])
        x = case_input(case)
        cache0 = case_in_cache(case, cfg)
        fwd = model.forward_softmax if case.get("softmax") else model.forward
        with torch.no_grad():
            xt = torch.from_numpy(x)
            if case.get("chunks"):
                ys, cache, t = [], (torch.from_numpy(cache0) if cache0 is not None else None), 0
                for n in case["chunks"]:
                    if cache is None:
                        y, cache = fwd(xt[:, t:t + n])
                    else:
                        y, cache = fwd(xt[:, t:t + n], cache)
                    ys.append(y)
                    t += n
                assert t == x.shape[1]
                y = torch.cat(ys, dim=1)
            else:
                if cache0 is None:
                    y, cache = fwd(xt)
                else:
                    y, cache = fwd(xt, torch.from_numpy(cache0))
        y, cache = y.numpy(), cache.numpy()
        out[name + "/y"] = y.astype(np.float32)
        # keep fixtures small: full cache for GRU (tiny), utterance 0 only for conv backbones
        if cfg["backbone"]["type"] == "gru":
            out[name + "/cache"] = cache.astype(np.float32)
        else:
            out[name + "/cache"] = cache[:1].astype(np.float32)
        out[name + "/wsum"] = np.float64(synth.checksum(sd))
        out[name + "/xsum"] = np.float64(np.abs(x.astype(np.float64)).sum())
        print(f"{name:42s} y{tuple(y.shape)} cache{tuple(cache.shape)} "
              f"y[min,max]=[{y.min():.4g},{y.max():.4g}]")
    path = os.path.join(HERE, "model_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(CASES), "cases; torch", torch.__version__)


if __name__ == "__main__":
    main()
