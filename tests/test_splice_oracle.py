"""CPU: the numpy restatement of context_expansion / frame_skip against the goldens recorded from the
reference's own functions (tests/golden/make_splice_golden.py).  Pure data movement -> bit-exact."""
import os

import numpy as np
import pytest

from oracle import splice_oracle
from tests.golden.splice_cases import CASES, case_input

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "splice_golden.npz"))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_splice_oracle_matches_reference(case):
    name, B, T, F, left, right, skip = case
    x = case_input(B, T, F)
    assert abs(float(np.abs(x.astype(np.float64)).sum()) - float(GOLD[name + "/xsum"])) < 1e-9
    y = splice_oracle.splice_skip(x, left, right, skip)
    assert y.shape == GOLD[name + "/y"].shape
    assert np.array_equal(y, GOLD[name + "/y"])
    assert np.array_equal(splice_oracle.lengths(np.full(B, T), right, skip), GOLD[name + "/lens"])


def test_closed_form():
    """out[b][i][(lag+left)*F + f] = feats[b][max(i*skip + lag, 0)][f] -- the formula the HIP kernel implements."""
    x = case_input(2, 37, 12, seed=3)
    left, right, skip = 3, 2, 4
    y = splice_oracle.splice_skip(x, left, right, skip)
    To = -(-(37 - right) // skip)
    assert y.shape == (2, To, 6 * 12)
    for i in range(To):
        for k, lag in enumerate(range(-left, right + 1)):
            assert np.array_equal(y[:, i, k * 12:(k + 1) * 12], x[:, max(i * skip + lag, 0)])
