#!/usr/bin/env python3
"""Development tool (diagnosis builds -DWEKWS_D64_EARLY -DWEKWS_D64_DUMP of ds64_g4): for the utterances whose posteriors differ from the
LDS-tile kernel's, where does the head go wrong -- the lane's partial sums (registers) or their way through LDS?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_hip_parity import build, run  # noqa: E402
from wekws_amd import pack  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h64"])
cfg["output_dim"] = 2
sd = synth.synth_state_dict(pack.model_spec(cfg), 2024)
fast, slow = build(cfg, sd), build(cfg, sd).set_option("g16", 0)
B, T = 4096, 98
x = synth.synth_feats(B, T, 40, seed=7 * B + T)
ys, cs = run(slow, x)
for rep in range(3):
    y, c = run(fast, x)
    d = np.abs(y - ys)
    bad = sorted(set(np.argwhere(d > 1e-5)[:, 0].tolist()))
    print(f"rep {rep}: {len(bad)} utterances differ; first {bad[:10]}")
    c = c.reshape(B, -1)
    for b in bad[:6]:
        regs = c[b, :256 * 16].reshape(256, 16)        # per thread: yp00 yp01 hv0..3 w0.xyzw yp10 yp11
        rows = c[b, 4096:4096 + 28 * 16].reshape(28, 16)  # per reduce thread th (column th >> 1, output th & 1): the 16 partial rows
        p0 = regs[:, 6] * regs[:, 2]
        for k in range(1, 4):
            p0 = np.float32(regs[:, 6 + k] * regs[:, 2 + k] + p0)
        wrong_reg = np.argwhere(np.abs(p0 - regs[:, 0]) > 1e-6 * (1 + np.abs(p0)))[:, 0]
        # LDS: partial row r = tid >> 4, lane l15 = tid & 15 wrote yp[0][k] to column 14 l15 + k; reduce thread th = 14 l15 + k (l15 < 2 in the dump)
        lds_bad = []
        for th in range(28):
            l15, rem = divmod(th, 14)
            tt, k = divmod(rem, 2)
            if tt not in (0, 5):                         # dumped registers: yp[0][*] and yp[1][*] only (tt = 0, 1)
                pass
            if tt > 1:
                continue
            for r in range(16):
                tid = r * 16 + l15
                want = regs[tid, (0 if tt == 0 else 10) + k]
                if rows[th, r] != want:
                    lds_bad.append((th, r, float(rows[th, r]), float(want)))
        tfr = sorted(set(np.argwhere(d[b] > 1e-5)[:, 0].tolist()))
        print(f"  b={b}: frames {tfr[:16]} max {d[b].max():.2e}; threads whose yp[0][0] != w0 . hv[0] recomputed from the dumped registers: "
              f"{wrong_reg.tolist()[:20]} (n={len(wrong_reg)}); LDS rows that differ from the registers written: {lds_bad[:8]} (n={len(lds_bad)})")
        if len(wrong_reg):
            t0 = int(wrong_reg[0])
            print(f"     thread {t0}: dumped yp00 {regs[t0, 0]!r} recomputed {p0[t0]!r} hv {regs[t0, 2:6]} w0 {regs[t0, 6:10]}")
