#!/usr/bin/env python3
"""Build-container only: turn the one trained model the reference ships (runtime/android/app/src/main/assets/kws.ort)
into build/ref_asset/{kws.wekwship, expect.npz} for tests/test_hip_parity.py::test_reference_android_asset_hip.
build/ is git-ignored but travels to the GPU box, where /root/reference does not exist.

expect.npz = the graph run by the numpy executor (oracle/onnx_graph_oracle.py) on two seeded 3-chunk streams:
  x / y / logit / cache            3 randn + 10 noise (round 1's input): the trained model answers with posteriors of
                                   1e-8 .. 1e-5, so only the LOGITS (one node before the Sigmoid) say anything there;
  x_kw / y_kw / logit_kw / cache_kw  an input that drives the keyword posterior THROUGH its range (round-3 review: the
                                   noise check passes for an all-zero output).  It is found by gradient ascent on the
                                   features through the torch port of the reference forward (oracle/torch_ref.py) on the
                                   state_dict recovered from the asset, then blended with the noise by a raised-cosine
                                   ramp so that the posterior sweeps 0 -> 1 -> 0 over the 240 frames.  The expected
                                   values come from the graph executor, not from the port that found the input.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import onnx_graph_oracle, torch_ref  # noqa: E402
from wekws_amd.bin import export_packed  # noqa: E402
from wekws_amd.model.kws_model import load_exported  # noqa: E402
from wekws_amd.utils import onnx_model  # noqa: E402

SRC = "/root/reference/runtime/android/app/src/main/assets/kws.ort"
T, CHUNK = 240, 80


def keyword_input(noise):
    """Features in the range real log-mel frames have (CMVN mean +- 3 sigma) that the trained model scores as its keyword."""
    m = load_exported(SRC)
    cfg, sd = m._cfg, {k: v.detach().clone() for k, v in m.state_dict().items()}
    mean, std = sd["global_cmvn.mean"], 1.0 / sd["global_cmvn.istd"]
    fwd = torch_ref.forward.__wrapped__                           # the port without its no_grad() wrapper
    cfg_logit = dict(cfg, activation=dict(type="identity"))
    x = torch.from_numpy(noise.copy()).requires_grad_(True)
    opt = torch.optim.Adam([x], lr=0.05)
    for it in range(400):
        opt.zero_grad()
        logit, _ = fwd(cfg_logit, sd, x)
        loss = -logit[:, 60:180].clamp(max=6.0).mean()           # push the middle two seconds to a confident keyword
        loss.backward()
        opt.step()
        with torch.no_grad():
            x.copy_(torch.maximum(torch.minimum(x, mean + 3 * std), mean - 3 * std))
    xk = x.detach().numpy()
    ramp = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(T) / (T - 1))  # 0 -> 1 -> 0
    return (noise + ramp[None, :, None] * (xk - noise)).astype(np.float32)


def stream(g, x, logit_name):
    cache = np.zeros((1, int(g.meta["cache_dim"]), int(g.meta["cache_len"])), np.float32)
    ys, ls = [], []
    for t in range(0, T, CHUNK):                                  # the Android app feeds 80-frame chunks
        o = onnx_graph_oracle.run(g, dict(input=x[:, t:t + CHUNK], cache=cache), also=(logit_name,))
        ys.append(o["output"])
        ls.append(o[logit_name])
        cache = o["r_cache"]
    return np.concatenate(ys, 1), np.concatenate(ls, 1), cache


def main():
    out = os.path.join(ROOT, "build", "ref_asset")
    os.makedirs(out, exist_ok=True)
    export_packed.main(["--exported", SRC, "--output", os.path.join(out, "kws.wekwship")])
    g = onnx_model.load_graph(SRC)
    sig = [n for n in g.toposorted() if n.op == "Sigmoid" and n.outputs[0] == "output"]
    assert len(sig) == 1
    logit_name = sig[0].inputs[0]
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((1, T, 40)) * 3 + 10).astype(np.float32)
    y, logit, cache = stream(g, x, logit_name)
    torch.manual_seed(0)
    x_kw = keyword_input(x)
    y_kw, logit_kw, cache_kw = stream(g, x_kw, logit_name)
    print("noise input: posterior %.2e .. %.2e, logits %.2f .. %.2f" % (y.min(), y.max(), logit.min(), logit.max()))
    print("keyword input: posterior %.3f .. %.3f, logits %.2f .. %.2f; frames inside (0.05, 0.95): %d" %
          (y_kw.min(), y_kw.max(), logit_kw.min(), logit_kw.max(), int(((y_kw > 0.05) & (y_kw < 0.95)).sum())))
    assert y_kw.max() > 0.95 and y_kw.min() < 0.05 and ((y_kw > 0.05) & (y_kw < 0.95)).sum() >= 8
    np.savez(os.path.join(out, "expect.npz"), x=x, y=y, logit=logit, cache=cache,
             x_kw=x_kw, y_kw=y_kw, logit_kw=logit_kw, cache_kw=cache_kw)
    print("wrote", out)


if __name__ == "__main__":
    main()
