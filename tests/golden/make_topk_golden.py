#!/usr/bin/env python3
"""Generate tests/golden/topk_golden.npz: torch ``softmax(-1).topk(k)`` on seeded logits (the numeric part of the
reference's first beam prune), and check with the reference's OWN ``ctc_prefix_beam_search`` that decoding from only
the k recorded values per frame reproduces decoding from the full posterior matrix.  Build container only.

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_topk_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")
from wekws.model.loss import ctc_prefix_beam_search  # noqa: E402  (the reference)
from tests.golden.topk_cases import CASES, case_logits  # noqa: E402


def main():
    out = {}
    for name, rows, K, k, scale in CASES:
        x = torch.from_numpy(case_logits(rows, K, scale))
        probs = x.softmax(-1)
        tv, ti = probs.topk(k)
        out[name + "/probs"] = tv.numpy()
        out[name + "/idx"] = ti.numpy()
        # contract check: the reference decoder sees the same hypotheses from the k best values alone
        if k >= 3:
            sparse = torch.zeros_like(probs)
            sparse.scatter_(1, ti[:, :3], tv[:, :3])
            full = ctc_prefix_beam_search(probs, torch.tensor([rows]), None, 3, 20)
            thin = ctc_prefix_beam_search(sparse, torch.tensor([rows]), None, 3, 20)
            assert [(h[0], h[1]) for h in full] == [(h[0], h[1]) for h in thin], name
        print(f"{name:14s} rows={rows} K={K} k={k} pmax={float(tv.max()):.4f}")
    path = os.path.join(HERE, "topk_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
