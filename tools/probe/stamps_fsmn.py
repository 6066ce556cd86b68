import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from wekws_amd import pack
from wekws_amd.model.kws_model import init_model
from wekws_amd.utils import synth
cfg = dict(synth.MODEL_CONFIGS["fsmn_ctc"])
m = init_model(cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(pack.model_spec(cfg), 1234).items()})
m = m.cuda().eval()
x = torch.from_numpy(synth.synth_feats(1024, 32, 400, seed=1)).cuda()
for _ in range(5): y, c = m(x)
torch.cuda.synchronize()
names = ["xload","in1","in2"] + [f"L{l}.{p}" for l in range(4) for p in ("proj","memc","memw","aff")] + ["out1","out2","TOTAL"]
for b in (0, 600):
    d = c[b].flatten()[:len(names)].cpu().numpy()
    print("block", b, " ".join(f"{n}={int(v)}" for n, v in zip(names, d)))
