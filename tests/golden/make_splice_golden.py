#!/usr/bin/env python3
"""Generate tests/golden/splice_golden.npz from the reference's OWN context_expansion / frame_skip.

wekws/dataset/init_dataset.py cannot be imported here (it imports the un-installed ``wenet`` package at module
level), so this script lifts the source text of exactly those two functions out of the file with ``ast`` and
executes them unchanged with torch.  Build container only (needs /root/reference).

    PYTHONPATH=/root/repo python tests/golden/make_splice_golden.py
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from tests.golden.splice_cases import CASES, RAISING, case_input  # noqa: E402

SRC = "/root/reference/wekws/dataset/init_dataset.py"


def load_reference_functions():
    text = open(SRC).read()
    tree = ast.parse(text)
    ns = {"torch": torch}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("context_expansion", "frame_skip"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, "exec"), ns)
    return ns["context_expansion"], ns["frame_skip"]


def main():
    ctx, skip_fn = load_reference_functions()
    out = {}
    for name, B, T, F, left, right, skip in CASES:
        x = case_input(B, T, F)
        lens = torch.full((B,), T, dtype=torch.int32)
        sample = {"feats": torch.from_numpy(x), "feats_lengths": lens}
        sample = ctx(sample, left=left, right=right)
        sample = skip_fn(sample, skip_rate=skip)
        y = sample["feats"].contiguous().numpy()
        out[name + "/y"] = y
        out[name + "/lens"] = sample["feats_lengths"].numpy()
        out[name + "/xsum"] = np.float64(np.abs(x.astype(np.float64)).sum())
        print(f"{name:20s} x{x.shape} -> y{y.shape} lens {sample['feats_lengths'].tolist()}")
    for name, B, T, F, left, right, skip in RAISING:            # what the reference does with left >= T: recorded, not assumed
        sample = {"feats": torch.from_numpy(case_input(B, T, F)), "feats_lengths": torch.full((B,), T, dtype=torch.int32)}
        try:
            skip_fn(ctx(sample, left=left, right=right), skip_rate=skip)
            out["raises/" + name] = np.int32(0)
        except IndexError as e:
            out["raises/" + name] = np.int32(1)
            print(f"{name:20s} IndexError: {e}")
    path = os.path.join(HERE, "splice_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
