"""Development tool (GPU box): for the seeds of the fbank framing fuzz, where the product is further than the tolerance from the float32
reference: how far are the product and the reference from the float64 evaluation of the same pipeline (oracle/fbank_oracle.py::fbank_f64)?
    python tools/probe/fbank_fuzz_diag.py LO HI"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import tests.test_hip_fbank as t
from oracle import fbank_oracle
from wekws_amd.frontend import Fbank

lo, hi = int(sys.argv[1]), int(sys.argv[2])
nover = nbad = ncase = 0
worst = (0.0, None)
for seed in range(lo, hi):
    for what, pcm in t.framing_cases(seed):
        _, trial, sr, flen, shift, bins, window, B, nsamp, kind = what
        if fbank_oracle.has_empty_filter(bins, sr, flen) or nsamp < flen:
            continue
        ncase += 1
        fb = Fbank(num_bins=bins, sample_rate=sr, frame_length=flen, frame_shift=shift, window=window)
        got = fb(torch.from_numpy(np.ascontiguousarray(pcm)).cuda()).cpu().numpy()
        tol = t.TOL if bins <= 40 else t.TOL80
        w = 0 if window == "hamming" else 1
        for i in range(B):
            ref = fbank_oracle.fbank(pcm[i], bins, sr, flen, shift, w)
            err = np.abs(got[i] - ref)
            near = ref >= ref.max(axis=-1, keepdims=True) - 13.8
            if float(err[near].max()) <= tol and float(err.max()) <= 20 * tol:
                continue
            nover += 1
            f64 = fbank_oracle.fbank_f64(pcm[i], bins, sr, flen, shift, w)
            bound = t.framing_bound(ref, tol); e_near = float((err - bound).max()); e_all = e_near; raw = float(err[near].max())
            gp, rp = np.abs(got[i] - f64), np.abs(ref - f64)
            k = np.unravel_index(np.argmax(np.where(near, err, 0)), err.shape)
            line = (f"{what} utt {i}: |got-ref| near {raw:.2e} all {err.max():.2e}; at the worst near bin: |got-f64| {gp[k]:.2e} |ref-f64| {rp[k]:.2e}; "
                    f"counted (bins where the product is further from f64 than the reference): near {e_near:.2e} all {e_all:.2e}")
            if e_near > 0:
                nbad += 1
                print("STILL OVER", line, flush=True)
            elif nover <= 12:
                print("excused  ", line, flush=True)
            if e_near > worst[0]:
                worst = (e_near, what)
print(f"seeds {lo} .. {hi - 1}: {ncase} configurations, {nover} utterances over the plain bar, {nbad} still over with the float64 excuse; worst counted near-error {worst}")
