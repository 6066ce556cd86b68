#!/usr/bin/env python3
"""The small recipes: the register-resident kernels (default) against the generic LDS-tile kernel (option g16 = 0), same
box, same process.  One JSON line per model and batch."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs import build, timeit  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402


def main():
    for name in sys.argv[1:] or ["ds_tcn_h64", "mdtc_small", "mdtc_small_global12", "mdtc_h64"]:
        cfg, m = build(name)
        for B in (1024, 8192):
            x = torch.from_numpy(synth.synth_feats(B, 98, cfg["input_dim"], seed=1)).cuda()
            row = dict(model=name, B=B, T=98)
            for tag, opt in (("resident", 1), ("generic", 0)):
                m.set_option("g16", opt)
                med, p10, p90 = timeit(lambda: m(x), warm=3, reps=15, group=10)
                row[tag + "_ms"] = round(med, 5)
                row[tag + "_utt_per_s"] = round(B / med * 1e3)
                med, p10, p90 = timeit(lambda: m.posteriors(x), warm=3, reps=15, group=10)
                row[tag + "_score_only_ms"] = round(med, 5)
            row["speedup"] = round(row["generic_ms"] / row["resident_ms"], 3)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
