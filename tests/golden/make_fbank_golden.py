#!/usr/bin/env python3
"""Generate tests/golden/fbank_golden.npz from the REFERENCE front-end compiled into oracle/_ref/
(make -C oracle; needs /root/reference, so build container only).  Inputs are regenerated from seeds by
the tests (wekws_amd.utils.synth.synth_pcm); only the reference outputs are stored.

    python tests/golden/make_fbank_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

from oracle import fbank_oracle  # noqa: E402
from tests.golden.fbank_cases import FBANK_CASES, fbank_input  # noqa: E402


def main():
    assert fbank_oracle.have_ref(), "run `make -C oracle` in the build container first"
    out = {}
    for case in FBANK_CASES:
        pcm = fbank_input(case)
        feats = np.stack([fbank_oracle.ref_fbank(p, case["num_bins"], case["sample_rate"], case["first_push"]) for p in pcm])
        out[case["name"]] = feats.astype(np.float32)
        out[case["name"] + "/xsum"] = np.float64(np.abs(pcm.astype(np.float64)).sum())
        print(f"{case['name']:28s} {feats.shape} [{feats.min():.4f}, {feats.max():.4f}]")
    path = os.path.join(HERE, "fbank_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
