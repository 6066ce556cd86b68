#!/usr/bin/env python3
"""Development tool (build container only: needs /root/reference): the numpy oracle (oracle/kws_oracle.py) against the LIVE reference
model on random configurations -- the configurations, batches, incoming caches and chunk cuts of
tests/test_hip_parity.py::test_random_model_shapes_against_the_oracle, whose GPU side trusts the oracle on exactly these.
    PYTHONPATH=/root/reference:/root/repo python tools/probe/fuzz_oracle_vs_reference.py [seeds]"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from wekws.model.kws_model import init_model  # noqa: E402  (the reference)
from wekws.model.cmvn import GlobalCMVN  # noqa: E402

from oracle import kws_oracle  # noqa: E402
from tests.helpers import random_model_config as _random_model_config  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = n = 0
worst = 0.0
for seed in range(nseeds):
    rng = np.random.default_rng(7000 + seed)
    for trial in range(12):
        cfg, head = _random_model_config(rng)
        with contextlib.redirect_stdout(io.StringIO()):
            model = init_model({k: v for k, v in cfg.items() if not k.startswith("_")} if "_cmvn" not in cfg else
                               {k: v for k, v in cfg.items() if k not in ("_cmvn", "cmvn")})
        if cfg.get("_cmvn"):
            model.global_cmvn = GlobalCMVN(torch.zeros(cfg["input_dim"]), torch.ones(cfg["input_dim"]), cfg["cmvn"]["norm_var"])
        sd = synth.synth_state_dict(synth.module_spec(model), 500 + 13 * seed + trial)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.eval()
        B, T = int(rng.choice([1, 2, 3, 9])), int(rng.integers(1, 200))
        x = synth.synth_feats(B, T, cfg["input_dim"], seed=trial, cmvn_like="cmvn" in cfg)
        gru = cfg["backbone"]["type"] == "gru"
        softmax = head == "linear" and bool(rng.integers(0, 4) == 0)
        with torch.no_grad():
            h00 = torch.zeros(cfg["backbone"]["num_layers"], B, cfg["hidden_dim"]) if gru else None
            _, c0 = model(torch.from_numpy(x[:, :1]), h00) if gru else model(torch.from_numpy(x[:, :1]))
        cin = None
        if gru or rng.integers(0, 2):
            cin = (0.5 * np.random.default_rng(trial).standard_normal(tuple(c0.shape))).astype(np.float32)
        fwd = model.forward_softmax if softmax else model.forward
        cut = int(rng.integers(1, T)) if T >= 2 else None
        with torch.no_grad():
            ty, tc = fwd(torch.from_numpy(x), torch.from_numpy(cin)) if cin is not None else fwd(torch.from_numpy(x))
            if cut and head == "linear":
                y1, c1 = fwd(torch.from_numpy(x[:, :cut]), torch.from_numpy(cin)) if cin is not None else fwd(torch.from_numpy(x[:, :cut]))
                y2, c2 = fwd(torch.from_numpy(x[:, cut:]), c1)
                tys, tcs = torch.cat([y1, y2], 1), c2
        n += 1
        try:
            oy, oc = kws_oracle.forward(cfg, sd, x, cin, softmax=softmax)
            sy = max(1.0, float(ty.abs().max()))
            sc = max(1.0, float(tc.abs().max()))
            ey, ec = float(np.abs(oy - ty.numpy()).max()) / sy, float(np.abs(oc - tc.numpy()).max()) / sc
            worst = max(worst, ey, ec)
            assert oy.shape == tuple(ty.shape) and oc.shape == tuple(tc.shape), "shapes"
            assert ey <= 2e-5 and ec <= 2e-5, (ey, ec)
            if cut and head == "linear":
                oys, ocs = kws_oracle.forward_streaming(cfg, sd, x, [cut, T - cut], cin) if not softmax else (None, None)
                if oys is not None:
                    e2 = float(np.abs(oys - tys.numpy()).max()) / sy
                    e3 = float(np.abs(ocs - tcs.numpy()).max()) / sc
                    worst = max(worst, e2, e3)
                    assert e2 <= 2e-5 and e3 <= 2e-5, ("streamed", e2, e3)
        except Exception as e:
            bad += 1
            print("FAIL", type(e).__name__, str(e)[:200], cfg, B, T, cut, softmax, flush=True)
print(f"oracle vs live reference: {n} configurations, {bad} failures, worst relative error {worst:.2e}")
