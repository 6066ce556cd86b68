"""oracle/det_oracle.py against tests/golden/det_golden.npz -- the stats lines produced by the reference's own threshold
loop (wekws/bin/compute_det.py:79-106, lifted with ast by tests/golden/make_det_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import det_oracle
from tests.golden.det_cases import CASES, case_data

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "det_golden.npz"))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_det_oracle_matches_reference_loop(case):
    name, B, T, K, kw, ws, step, ragged = case
    s, lengths, is_kw, dur = case_data(*case)
    assert abs(np.abs(s.astype(np.float64)).sum() - float(GOLD[name + "/ssum"])) < 1e-9
    keyword_table = {b: s[b, :lengths[b], kw].tolist() for b in range(B) if is_kw[b] and lengths[b] > 0}
    filler_table = {b: s[b, :lengths[b], kw].tolist() for b in range(B) if not is_kw[b]}
    rows = np.asarray(det_oracle.det_stats(keyword_table, filler_table, dur, step, ws))
    gold = GOLD[name + "/rows"]
    assert rows.shape == gold.shape
    assert np.abs(rows - gold).max() <= 5.1e-7          # the reference prints with {:.6f}
    mx, am = det_oracle.max_pool(s, lengths)
    assert np.array_equal(mx[:, kw], GOLD[name + "/max"])
    for b in range(B):
        if lengths[b]:
            assert s[b, am[b, kw], kw] == mx[b, kw] and not (s[b, :am[b, kw], kw] == mx[b, kw]).any()
