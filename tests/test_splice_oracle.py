"""CPU: the numpy restatement of context_expansion / frame_skip against the goldens recorded from the
reference's own functions (tests/golden/make_splice_golden.py).  Pure data movement -> bit-exact."""
import os

import numpy as np
import pytest

from oracle import splice_oracle
from tests.golden.splice_cases import CASES, RAISING, case_input

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


@pytest.mark.parametrize("case", RAISING, ids=[c[0] for c in RAISING])
def test_left_context_not_shorter_than_the_utterance_raises_like_the_reference(case):
    """left >= T: the reference's left-margin loop (init_dataset.py:45-48) reads feats_ctx[:, left] -> IndexError; the golden file
    records that the live reference did raise, the restatement raises the same way."""
    name, B, T, F, left, right, skip = case
    assert int(GOLD["raises/" + name]) == 1
    with pytest.raises(IndexError):
        splice_oracle.splice_skip(case_input(B, T, F), left, right, skip)


def test_frame_count_of_the_c_abi_matches_the_oracle_for_every_small_length():
    """wekws_hip_splice_frames is host arithmetic: swept on the CPU against the oracle's output shape, including utterances shorter
    than their right context (the reference's negative slice keeps 2 T - right frames)."""
    from wekws_amd import _capi
    lib = _capi.load()
    for T in range(0, 14):
        for right in range(0, 30):
            for skip in (1, 2, 3, 7):
                want = splice_oracle.splice_skip(np.zeros((1, T, 1), np.float32), 0, right, skip).shape[1]
                assert int(lib.wekws_hip_splice_frames(T, right, skip)) == want, (T, right, skip)
