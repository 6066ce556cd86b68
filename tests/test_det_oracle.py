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


@pytest.mark.parametrize("empty", ["keywords", "filler_audio", "both"])
def test_empty_tables_die_like_the_reference(empty):
    """compute_det.py:97-104 assigns the rates only under `if len(keyword_table) != 0` / `if filler_duration != 0` and then formats
    them: NameError at the first row (false_alarm_per_hour is evaluated first).  The oracle raises the same error; where the reference
    tree is present its own loop (lifted like tests/golden/make_det_golden.py does) is executed beside it."""
    import io
    import types
    kt = {} if empty in ("keywords", "both") else {"k0": [0.2, 0.9]}
    dur = 0.0 if empty in ("filler_audio", "both") else 36.0
    ft = {"f0": [0.1, 0.6, 0.3]}
    want = "false_reject_rate" if empty == "keywords" else "false_alarm_per_hour"
    with pytest.raises(NameError, match=want):
        det_oracle.det_stats(kt, ft, dur)
    if os.path.exists("/root/reference/wekws/bin/compute_det.py"):
        from tests.golden.make_det_golden import reference_loop
        ns = dict(args=types.SimpleNamespace(keyword="KW", step=0.01), window_shift=50, keyword_table=kt, filler_table=ft,
                  filler_duration=dur, fout=io.StringIO())
        with pytest.raises(NameError, match=want):
            exec(reference_loop(), ns)
