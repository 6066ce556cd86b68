#!/usr/bin/env python3
"""Development tool: per-block difference of the returned cache between mdtc64_g4 (option g16 = 1) and mdtc64_w16 (g16 = 0):
the cache slices are the blocks' inputs, so the first differing slice names the block that went wrong.
    python tools/probe/dbg_g4.py [model] [T]        (PREC=f16 for the fp16 mode)"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wekws_amd import pack
from wekws_amd.model.kws_model import init_model
from wekws_amd.utils import synth
name = sys.argv[1] if len(sys.argv) > 1 else "mdtc_h64"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 98
cfg = dict(synth.MODEL_CONFIGS[name])
sd = synth.synth_state_dict(pack.model_spec(cfg), 79)
def mk(g):
    m = init_model(cfg); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.cuda().eval().set_option("g16", g)
import os
a, b = mk(1), mk(0)
if os.environ.get("PREC"):
    a.set_precision(os.environ["PREC"]); b.set_precision(os.environ["PREC"])
x = torch.from_numpy(synth.synth_feats(3, T, cfg["input_dim"], seed=T)).cuda()
ya, ca = a(x); yb, cb = b(x)
print("y diff", float((ya - yb).abs().max()))
pads = [4] + [4, 8, 16, 32] * 4
o = 0
for i, p in enumerate(pads):
    d = (ca[:, :, o:o + p] - cb[:, :, o:o + p]).abs()
    print(f"block {i} pad {p}: max diff {float(d.max()):.3e}  (ref max {float(cb[:, :, o:o+p].abs().max()):.3f})", "worst ch", int(d.amax(dim=(0, 2)).argmax()), "worst col", int(d.amax(dim=(0, 1)).argmax()))
    o += p
o = 4 + 4 + 8 + 16 + 32
d = (ca[0, :, o:o + 4] - cb[0, :, o:o + 4]).abs().cpu().numpy()
np.set_printoptions(linewidth=200, precision=3, suppress=True)
print("block 5 input, utterance 0: |diff| per channel (rows of 16 = waves), cols 0..3 summed")
print(d.sum(1).reshape(4, 16))
print("g4 values ch 0..7:", ca[0, :8, o:o + 4].cpu().numpy().round(3).tolist())
print("w16 values ch 0..7:", cb[0, :8, o:o + 4].cpu().numpy().round(3).tolist())
