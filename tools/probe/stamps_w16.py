import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from tools.bench_configs import build
from wekws_amd.utils import synth
cfg, m = build("ds_tcn_h256")
names = ["outcache(in produce)", "produce_rest", "bar1", "mfma", "bar2", "pre", "blockprologue+loop", "epilogue"]
x = torch.from_numpy(synth.synth_feats(1024, 98, 40, seed=1)).cuda()
for _ in range(4): y, c = m(x)
torch.cuda.synchronize()
d = c[0].flatten()[:8].cpu().numpy()
print(" ".join(f"{n}={int(v)}" for n, v in zip(names, d)), "total", int(d.sum()))
