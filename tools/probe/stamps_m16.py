import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from tools.bench_configs import build
from wekws_amd.utils import synth
cfg, m = build("mdtc_h64")
names = ["produce", "bar1", "gemm1", "bar2", "mid", "bar3", "gemm2+epi", "bar4", "pre", "zsum+head"]
x = torch.from_numpy(synth.synth_feats(1024, 98, 40, seed=1)).cuda()
for _ in range(4): y, c = m(x)
torch.cuda.synchronize()
d = c[0].flatten()[:10].cpu().numpy()
print(" ".join(f"{n}={int(v)}" for n, v in zip(names, d)), "total", int(d.sum()))
