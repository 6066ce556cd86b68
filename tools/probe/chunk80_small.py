import json, os, sys, torch
sys.path.insert(0, "/root/repo") if os.path.exists("/root/repo/tools") else sys.path.insert(0, os.getcwd())
from tools.bench_configs import build, timeit
from wekws_amd.utils import synth
for name in ("ds_tcn_h64", "mdtc_small"):
    cfg, m = build(name)
    for B in (1, 256, 1024):
        x = torch.from_numpy(synth.synth_feats(B, 80, cfg["input_dim"], seed=1)).cuda()
        _, c = m(x)
        f = timeit(lambda: m(x), warm=3, reps=12, group=10)[0]
        w = timeit(lambda: m(x, c), warm=3, reps=12, group=10)[0]
        print(json.dumps(dict(model=name, B=B, T=80, first_ms=round(f, 4), cache_ms=round(w, 4), ratio=round(w / f, 2))), flush=True)
