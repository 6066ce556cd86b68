#!/usr/bin/env python3
"""Per-stage wall-clock stamps of the GRU wavefront (slot 0), from the measurement build:
   make -C wekws_amd/csrc EXTRA=-DWEKWS_GRU_PIPE_STAMPS OUT=../../build/var/libwekws_stamps.so OBJDIR=../../build/var/obj_stamps
   python tools/probe/gru_stamps.py [B] [T]
Rows: 0 PI chunk top, 1 after the barrier, 2 after the drain; R_l: 4l+3 step top, 4l+4 after the MFMAs, 4l+5 before /
4l+6 after ensure; I_1: 11 step top, 12 after the MFMAs, 13 after the drain, 14 after a blocking ensure.  Unit: 10 ns."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wekws_amd import _capi  # noqa: E402
_capi._LIB_PATH = os.environ.get("WEKWS_DBG_LIB") or os.path.join(ROOT, "build", "var", "libwekws_stamps.so")
from tools.bench_configs import build  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 98
    cfg, m = build("gru_2x128")
    lib = _capi.load()
    x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=1)).cuda()
    for _ in range(20):
        m(x)
    torch.cuda.synchronize()
    buf = np.zeros(16 * 1024, np.uint64)
    lib.wekws_hip_debug_gru_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.wekws_hip_debug_gru_stamps(buf.ctypes.data, buf.size) == 0
    s = buf.reshape(16, 1024).astype(np.int64)
    t0 = s[s > 0].min()
    np.set_printoptions(linewidth=250)
    names = {0: "PI top", 1: "PI barrier", 2: "PI drained", 3: "R0 top", 4: "R0 mfma", 5: "R0 pre-ensure", 6: "R0 ensured",
             7: "R1 top", 8: "R1 mfma", 9: "R1 pre-ensure", 10: "R1 ensured", 11: "I1 top", 12: "I1 mfma", 13: "I1 drained",
             14: "I1 ensured"}
    print(f"B={B} T={T}; times in us since the first stamp")
    for k in range(14):
        row = s[k, :T]
        if not (row > 0).any():
            continue
        v = np.where(row > 0, (row - t0) / 100.0, np.nan)
        print(f"{names[k]:14s}", " ".join(f"{a:7.2f}" for a in v[:min(T, 24)]), "...", " ".join(f"{a:7.2f}" for a in v[max(T - 4, 24):T]))
    near = [("near" if s[14, 100 + st] > 0 else "far" if s[14, 200 + st] > 0 else "-") for st in range(8)]
    print("stores of stage 0..7:", near, "handshake done at", [round((s[15, st] - t0) / 100.0, 2) if s[15, st] > 0 else None for st in range(8)])
    for k in (3, 7, 11):
        row = s[k, :T]
        d = np.diff(row[row > 0]) / 100.0
        if d.size:
            print(f"{names[k]:14s} step median {np.median(d):.2f} us, mean {d.mean():.2f}, max {d.max():.2f}")


if __name__ == "__main__":
    main()
