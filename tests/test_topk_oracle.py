"""CPU: numpy softmax + top-k against the goldens recorded from torch (tests/golden/make_topk_golden.py)."""
import os

import numpy as np
import pytest

from oracle import topk_oracle
from tests.golden.topk_cases import CASES, case_logits

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "topk_golden.npz"))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_topk_oracle_matches_torch(case):
    name, rows, K, k, scale = case
    p, i = topk_oracle.softmax_topk(case_logits(rows, K, scale), k)
    assert np.array_equal(i, GOLD[name + "/idx"])
    assert np.abs(p - GOLD[name + "/probs"]).max() <= 2e-7
