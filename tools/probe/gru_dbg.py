#!/usr/bin/env python3
"""Where do the GRU wavefront and the layer-major kernels differ?  python tools/probe/gru_dbg.py L B T [h0scale]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wekws_amd import _capi, pack  # noqa: E402
if os.environ.get("WEKWS_DBG_LIB"):
    _capi._LIB_PATH = os.environ["WEKWS_DBG_LIB"]
from wekws_amd.model.kws_model import init_model  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402


def main():
    L, B, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    hs = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    cfg = dict(synth.MODEL_CONFIGS["gru_2x128"])
    cfg["backbone"] = dict(cfg["backbone"], num_layers=L)
    sd = synth.synth_state_dict(pack.model_spec(cfg), 4242 + L)
    ms = []
    for opt in (2, 0):
        m = init_model(cfg)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        ms.append(m.cuda().eval().set_option("gru_pipe", opt))
    x = torch.from_numpy(synth.synth_feats(B, T, 40, seed=58)).cuda()
    h0 = (torch.randn(L, B, 128, device="cuda") * hs) if hs else None
    for it in range(3):
        (y1, c1), (y0, c0) = [m(x) if h0 is None else m(x, h0) for m in ms]
        torch.cuda.synchronize()
        d = (y1 - y0).abs()
        bad = (d.amax(dim=(1, 2)) > 0).nonzero().flatten().cpu().numpy()
        print(f"it {it}: max |dy| {float(d.max()):.3e}  max |dh| {float((c1 - c0).abs().max()):.3e}  differing streams {bad.size}"
              f" tiles {sorted(set((bad // 16).tolist()))[:40]}")
        if bad.size:
            s = int(bad[0])
            tt = (d[s].amax(dim=1) > 0).nonzero().flatten().cpu().numpy()
            print("   stream", s, "first differing steps", tt[:10], "dy there", d[s, tt[:5]].cpu().numpy().ravel())


if __name__ == "__main__":
    main()
