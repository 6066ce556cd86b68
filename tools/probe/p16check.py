import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from wekws_amd import pack
from wekws_amd.model.kws_model import init_model
from wekws_amd.utils import synth
from oracle import kws_oracle
cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
sd = synth.synth_state_dict(pack.model_spec(cfg), 77)
def build(**o):
    m = init_model(cfg); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
    for k, v in o.items(): m.set_option(k, v)
    return m
a, b = build(p16=1), build(p16=0)
for B, T in ((3, 98), (2, 64), (1, 112), (4, 50), (2, 150)):
    x = synth.synth_feats(B, T, 40, seed=T)
    xt = torch.from_numpy(x).cuda()
    ya, ca = a(xt); yb, cb = b(xt)
    ry, rc = kws_oracle.forward(cfg, sd, x, None)
    print(B, T, "p16 vs g16: y", float((ya - yb).abs().max()), "cache", float((ca - cb).abs().max()),
          "| p16 vs oracle: y", float(np.abs(ya.cpu().numpy() - ry).max()), "cache", float(np.abs(ca.cpu().numpy() - rc).max()),
          "| g16 vs oracle y", float(np.abs(yb.cpu().numpy() - ry).max()))
