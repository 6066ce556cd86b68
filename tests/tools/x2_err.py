import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import kws_oracle as ko
from tests.helpers import CASES, case_weights, case_input, case_in_cache
gold = np.load("/root/repo/tests/golden/model_golden.npz")
F32 = np.float32
orig_pw, orig_lin = ko.pointwise_conv, ko.linear
def r16(a): return a.astype(np.float16).astype(F32)
mode = sys.argv[1]
def pw(x, w, b):
    w = np.asarray(w, F32)
    if mode == "act_hi":   x = r16(x)
    if mode == "w_hi":     w = r16(w)
    if mode == "both_hi":  x, w = r16(x), r16(w)
    return orig_pw(x, w, b)
def lin(x, w, b):
    w = np.asarray(w, F32)
    if mode == "act_hi":   x = r16(np.asarray(x, F32))
    if mode == "w_hi":     w = r16(w)
    if mode == "both_hi":  x, w = r16(np.asarray(x, F32)), r16(w)
    return orig_lin(x, w, b)
ko.pointwise_conv, ko.linear = pw, lin
worst = {}
for c in CASES:
    if c["model"] not in ("ds_tcn_h256", "mdtc_h64", "ds_tcn_h64", "mdtc_small", "ds_tcn_h256_ctc300"): continue
    cfg, sd = case_weights(c)
    if c.get("chunks"):
        y, cache = ko.forward_streaming(cfg, sd, case_input(c), c["chunks"], case_in_cache(c, cfg))
    else:
        y, cache = ko.forward(cfg, sd, case_input(c), case_in_cache(c, cfg), softmax=c.get("softmax", False))
    e = float(np.abs(y - gold[c["name"] + "/y"]).max())
    worst[c["model"]] = max(worst.get(c["model"], 0), e)
print(mode, {k: f"{v:.2e}" for k, v in worst.items()})
