"""Development tool (GPU box): one launch of B utterances far beyond what the suite runs (index arithmetic, grid sizes, workspace
sizing): outputs finite, and the first / last / middle utterances bit-identical to the same utterances run as a small batch.
    python tools/probe/huge_batch.py [B] name..."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

args = [a for a in sys.argv[1:] if a != "--more"]
B = next((int(a) for a in args if a.isdigit()), 70000)
T = 98
for name in [a for a in args if not a.isdigit()] or ["ds_tcn_h256", "mdtc_h64", "ds_tcn_h64", "gru_2x128", "tcn_h64", "mdtc_h64_global12"]:
    cfg, m = build(name)
    base = torch.from_numpy(synth.synth_feats(4096, T, cfg["input_dim"], seed=7)).cuda()
    x = base.repeat((B + 4095) // 4096, 1, 1)[:B].contiguous()
    x[-8:] = torch.from_numpy(synth.synth_feats(8, T, cfg["input_dim"], seed=8)).cuda()      # the tail is its own data
    gru = cfg["backbone"]["type"] == "gru"
    for prec in ("default", "f32"):
        m.set_precision(prec)
        y, c = m(x)
        torch.cuda.synchronize()
        idx = torch.tensor(list(range(8)) + list(range(B // 2, B // 2 + 8)) + list(range(B - 8, B)), device="cuda")
        ys, cs = m(x[idx].contiguous())
        csel = c[:, idx] if gru else c[idx]
        ok = bool(torch.isfinite(y).all()) and bool(torch.isfinite(c).all())
        same = torch.equal(y[idx], ys) and torch.equal(csel, cs)
        # repeats of the 4096-utterance block must repeat bit for bit
        rep = torch.equal(y[:4096], y[4096:8192]) if B >= 8192 else True
        print(f"{name:20s} {prec:8s} B={B}: finite={ok} sub-batch identical={same} repeated block identical={rep} "
              f"(y {tuple(y.shape)}, cache {tuple(c.shape)} = {c.numel() * 4 / 2**30:.1f} GiB)", flush=True)
        del y, c, ys, cs
        torch.cuda.empty_cache()

if "--more" in sys.argv or not [a for a in args if not a.isdigit()]:
    # CTC-sized heads: more than 2^31 logits in one call
    for name, Bc in (("ds_tcn_h256_ctc", 9000), ("fsmn_ctc", 9000)):
        cfg, m = build(name)
        x = torch.from_numpy(synth.synth_feats(512, T, cfg["input_dim"], seed=7)).cuda().repeat((Bc + 511) // 512, 1, 1)[:Bc].contiguous()
        y, c = m(x)
        torch.cuda.synchronize()
        idx = torch.tensor(list(range(4)) + list(range(Bc - 4, Bc)), device="cuda")
        ys, cs = m(x[idx].contiguous())
        print(f"{name:20s} B={Bc}: y {tuple(y.shape)} = {y.numel() / 2**31:.2f} x 2^31 elements, finite={bool(torch.isfinite(y).all())} "
              f"sub-batch identical={torch.equal(y[idx], ys) and torch.equal(c[idx], cs)} repeated block identical={torch.equal(y[:512], y[512:1024])}", flush=True)
        del y, c, ys, cs
        torch.cuda.empty_cache()
    # streaming steps: more streams than 2^31 cache elements
    for name in ("ds_tcn_h256", "mdtc_h64", "gru_2x128"):
        cfg, m = build(name)
        Bs = 90000
        gru = cfg["backbone"]["type"] == "gru"
        x = torch.from_numpy(synth.synth_feats(1024, 20, cfg["input_dim"], seed=9)).cuda().repeat((Bs + 1023) // 1024, 1, 1)[:Bs].contiguous()
        y1, c1 = m(x[:, :10].contiguous())
        y2, c2 = m(x[:, 10:].contiguous(), c1)
        torch.cuda.synchronize()
        idx = torch.tensor(list(range(4)) + list(range(Bs - 4, Bs)), device="cuda")
        s1, d1 = m(x[idx, :10].contiguous())
        s2, d2 = m(x[idx, 10:].contiguous(), d1)
        csel = c2[:, idx] if gru else c2[idx]
        print(f"{name:20s} {Bs} streams x 2 chunks of 10 frames: cache {tuple(c2.shape)} = {c2.numel() / 2**31:.2f} x 2^31 elements, "
              f"finite={bool(torch.isfinite(y2).all()) and bool(torch.isfinite(c2).all())} sub-batch identical={torch.equal(y2[idx], s2) and torch.equal(csel, d2)} "
              f"repeated block identical={torch.equal(y2[:1024], y2[1024:2048])}", flush=True)
        del y1, c1, y2, c2
        torch.cuda.empty_cache()
    # fbank: more than 2^31 samples in one call
    from wekws_amd.frontend import Fbank
    fb = Fbank(num_bins=40)
    Bf = 140000
    pcm = torch.from_numpy(synth.synth_pcm(1000, 16000, seed=1, kind="noise")).cuda().repeat(Bf // 1000, 1).contiguous()
    f = fb(pcm)
    torch.cuda.synchronize()
    print(f"fbank B={Bf}: {pcm.numel() / 2**31:.2f} x 2^31 samples -> {tuple(f.shape)}, finite={bool(torch.isfinite(f).all())} "
          f"sub-batch identical={torch.equal(f[-8:], fb(pcm[-8:].contiguous()))} repeated block identical={torch.equal(f[:1000], f[-1000:])}", flush=True)
