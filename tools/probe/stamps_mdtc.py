import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from tools.bench_configs import build
from wekws_amd.utils import synth
cfg, m = build("mdtc_h64")
names = ["produce0(+pre)", "produce1+mfma_a0", "mfma_a1", "mid_epi", "gemm2+epi2", "x", "zsum", "head"]
for B, T in ((1024, 98), (1, 10)):
    x = torch.from_numpy(synth.synth_feats(B, T, 40, seed=1)).cuda()
    y, c = m(x)
    for _ in range(3):
        y, c = m(x, c) if B == 1 else m(x)
    torch.cuda.synchronize()
    d = c[0].flatten()[:8].cpu().numpy()
    print(f"B={B} T={T}", " ".join(f"{n}={int(v)}" for n, v in zip(names, d)), "total", int(d.sum()))
