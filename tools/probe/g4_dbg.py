#!/usr/bin/env python3
"""All rows of the register-resident kernels against the generic kernel at large batches: python tools/probe/g4_dbg.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wekws_amd import pack, _capi
if os.environ.get("WEKWS_DBG_LIB"):
    _capi._LIB_PATH = os.environ["WEKWS_DBG_LIB"]
from wekws_amd.model.kws_model import init_model
from wekws_amd.utils import synth
for name, odim in (("mdtc_h64", 2), ("ds_tcn_h64", 2), ("ds_tcn_h64", 1), ("ds_tcn_h256", 2)):
    cfg = dict(synth.MODEL_CONFIGS[name]); cfg["output_dim"] = odim
    sd = synth.synth_state_dict(pack.model_spec(cfg), 2024)
    ms = []
    for opt in (1, 0):
        m = init_model(cfg); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        ms.append(m.cuda().eval().set_option("g16", opt))
    for B, T in ((1024, 98), (4096, 98), (2048, 49), (2048, 28), (2048, 100)):
        x = torch.from_numpy(synth.synth_feats(B, T, 40, seed=B + T)).cuda()
        (y1, c1), (y0, c0) = [m(x) for m in ms]
        torch.cuda.synchronize()
        d = (y1 - y0).abs().amax(dim=(1, 2)).cpu().numpy()
        dc = (c1 - c0).abs().amax(dim=(1, 2)).cpu().numpy()
        print(name, "B", B, "T", T, "max dy %.2e" % d.max(), "utterances with dy > 1e-5:", int((d > 1e-5).sum()), " cache: max %.2e" % dc.max(), "bad", int((dc > 1e-4).sum()), flush=True)
