"""CPU: the numpy oracle against the live reference's outputs for NaN / +Inf / -Inf features and caches
(tests/golden/make_nonfinite_golden.py): same class (finite / NaN / +Inf / -Inf) at every position of y and of the returned cache,
same finite values.  The GPU side of the same cases is tests/test_hip_nonfinite.py."""
import os

import numpy as np
import pytest

from oracle import kws_oracle
from tests.golden.nonfinite_cases import CASES, classify, poisoned_input
from tests.helpers import case_weights

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def nf_golden():
    return np.load(os.path.join(HERE, "golden", "nonfinite_golden.npz"))


def run_oracle(case, cfg, sd, x, cache0):
    with np.errstate(all="ignore"):
        if case.get("chunks"):
            assert not case.get("softmax")
            return kws_oracle.forward_streaming(cfg, sd, x, case["chunks"], cache0)
        return kws_oracle.forward(cfg, sd, x, cache0, softmax=case.get("softmax", False))


def check_against(y, cache, gy, gcc, tol=5e-6):
    assert y.shape == gy.shape and cache.shape == gcc.shape
    assert np.array_equal(classify(y), classify(gy)), "finite / NaN / +Inf / -Inf classes of y differ"
    fin = np.isfinite(gy)
    if fin.any():
        assert float(np.abs(y[fin].astype(np.float64) - gy[fin]).max()) <= tol * max(1.0, float(np.abs(gy[fin]).max()))
    assert np.array_equal(classify(cache), gcc), "classes of the returned cache differ"


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_propagates_nonfinite_like_the_reference(case, nf_golden):
    cfg, sd = case_weights(case)
    x, cache0 = poisoned_input(case, cfg)
    y, cache = run_oracle(case, cfg, sd, x, cache0)
    name = case["name"]
    check_against(y, cache, nf_golden[name + "/y"], nf_golden[name + "/cache_class"])
    fsum = float(np.abs(np.where(classify(cache) == 0, cache, 0).astype(np.float64)).sum())
    assert abs(fsum - float(nf_golden[name + "/cache_finite_sum"])) <= 2e-5 * max(1.0, float(nf_golden[name + "/cache_finite_sum"]))
