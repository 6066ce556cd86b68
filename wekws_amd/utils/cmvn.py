"""CMVN statistics file readers (init-time host code feeding the packer).

Reference behaviour: wekws/utils/cmvn.py:23-45 (json accumulators -> mean, 1/std with a 1e-20 variance
floor) and :48-93 (Kaldi nnet ``<AddShift>`` / ``<Rescale>`` / ``<Splice>`` text: mean = -shift,
istd = scale, tiled once per splice offset).  Both return a (2, dim) float64 array [mean, istd]."""
from __future__ import annotations

import json
import re

import numpy as np

_BRACKET = re.compile(r"\[(.*?)\]")


def load_cmvn(json_cmvn_file: str) -> np.ndarray:
    with open(json_cmvn_file) as f:
        stats = json.load(f)
    n = float(stats["frame_num"])
    mean = np.asarray(stats["mean_stat"], dtype=np.float64) / n
    var = np.asarray(stats["var_stat"], dtype=np.float64) / n - mean * mean
    istd = 1.0 / np.sqrt(np.maximum(var, 1.0e-20))
    return np.stack([mean, istd])


def _vector_after(lines, idx, expect_dim=None):
    vals = [float(s) for s in _BRACKET.findall(lines[idx + 1])[0].split()]
    if expect_dim is not None and len(vals) != expect_dim:
        raise ValueError(f"line {idx + 2}: expected {expect_dim} values, found {len(vals)}")
    return vals


def load_kaldi_cmvn(cmvn_file: str) -> np.ndarray:
    with open(cmvn_file) as f:
        lines = f.readlines()
    mean = istd = None
    copies = None
    for i, line in enumerate(lines):
        seg = line.strip().split(" ")
        if "AddShift" in line:
            mean = -np.asarray(_vector_after(lines, i, int(seg[1])), dtype=np.float64)
        elif "Rescale" in line:
            istd = np.asarray(_vector_after(lines, i, int(seg[1])), dtype=np.float64)
        elif "Splice" in line:
            offs = _vector_after(lines, i)
            if len(offs) * int(seg[2]) != int(seg[1]):
                raise ValueError("inconsistent <Splice> dimensions")
            copies = len(offs)
    if mean is None or istd is None:
        raise ValueError(f"{cmvn_file}: <AddShift>/<Rescale> not found")
    if copies is None:  # the reference hits an unbound variable here; a file without <Splice> means no tiling
        copies = 1
    return np.tile(np.stack([mean, istd]), (1, copies))
