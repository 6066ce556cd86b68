#!/usr/bin/env python3
"""Generate tests/golden/det_golden.npz by executing the reference's OWN threshold loop.

wekws/bin/compute_det.py keeps the DET arithmetic in its ``if __name__ == '__main__':`` block (lines 79-106), reading
its inputs from files.  This script lifts the ``while threshold <= 1.0:`` statement (and the two assignments in front of
it) out of that block with ``ast`` and executes them UNCHANGED on in-memory score tables (what load_label_and_score,
compute_det.py:20-52, would return: {key: [float, ...]}); the lines it writes to ``fout`` are the golden.
Build container only (needs /root/reference).

    PYTHONPATH=/root/repo python tests/golden/make_det_golden.py
"""
import ast
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from tests.golden.det_cases import CASES, case_data  # noqa: E402

SRC = "/root/reference/wekws/bin/compute_det.py"


def reference_loop():
    tree = ast.parse(open(SRC).read())
    main = [n for n in tree.body if isinstance(n, ast.If)][-1]               # if __name__ == '__main__':
    with_stmt = [n for n in main.body if isinstance(n, ast.With)][-1]         # with open(args.stats_file, 'w') as fout:
    body = with_stmt.body                                                     # keyword = ...; threshold = 0.0; while ...
    assert any(isinstance(n, ast.While) for n in body)
    return compile(ast.Module(body=body, type_ignores=[]), SRC, "exec")


def main():
    code = reference_loop()
    out = {}
    for case in CASES:
        name, B, T, K, kw, ws, step, ragged = case
        s, lengths, is_kw, dur = case_data(*case)
        keyword_table = {f"utt{b}": s[b, :lengths[b], kw].tolist() for b in range(B) if is_kw[b] and lengths[b] > 0}
        filler_table = {f"utt{b}": s[b, :lengths[b], kw].tolist() for b in range(B) if not is_kw[b]}
        fout = io.StringIO()
        ns = dict(args=types.SimpleNamespace(keyword="KW", step=step), window_shift=ws, keyword_table=keyword_table,
                  filler_table=filler_table, filler_duration=dur, fout=fout)
        exec(code, ns)
        rows = np.asarray([[float(v) for v in ln.split()] for ln in fout.getvalue().splitlines()], np.float64)
        out[name + "/rows"] = rows
        out[name + "/ssum"] = np.float64(np.abs(s.astype(np.float64)).sum())
        # per-utterance facts the device ops are checked on (plain Python on the same lists)
        out[name + "/max"] = np.asarray([max(s[b, :lengths[b], kw].tolist()) if lengths[b] else -np.inf for b in range(B)], np.float32)
        print(f"{name:22s} {rows.shape[0]} thresholds; FA/h at 0.5: {rows[rows.shape[0] // 2, 1]:.3f}; FRR at 0.5: {rows[rows.shape[0] // 2, 2]:.3f}")
    path = os.path.join(HERE, "det_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
