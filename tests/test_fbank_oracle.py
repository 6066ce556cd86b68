"""CPU: the plain-C fbank oracle against (a) the golden vectors recorded from the reference front-end and
(b) the compiled reference itself when oracle/_ref is present."""
import os

import numpy as np
import pytest

from oracle import fbank_oracle
from tests.golden.fbank_cases import FBANK_CASES, fbank_input

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fgolden():
    return np.load(os.path.join(ROOT, "tests", "golden", "fbank_golden.npz"))


@pytest.mark.parametrize("case", FBANK_CASES, ids=[c["name"] for c in FBANK_CASES])
def test_oracle_matches_golden(case, fgolden):
    pcm = fbank_input(case)
    assert abs(np.abs(pcm.astype(np.float64)).sum() - float(fgolden[case["name"] + "/xsum"])) < 1e-6
    sr = case["sample_rate"]
    got = np.stack([fbank_oracle.fbank(p, case["num_bins"], sr, sr // 1000 * 25, sr // 1000 * 10) for p in pcm])
    ref = fgolden[case["name"]]
    assert got.shape == ref.shape
    # same algorithm, same float32 operation order: bit-exact on this toolchain; allow 1 ulp-ish slack for libm
    assert float(np.abs(got - ref).max()) <= 2e-6


@pytest.mark.skipif(not fbank_oracle.have_ref(), reason="oracle/_ref not built (reference tree absent)")
def test_oracle_matches_compiled_reference():
    rng = np.random.default_rng(7)
    for n in (400, 559, 560, 8000, 16000, 23456):
        pcm = np.round(rng.standard_normal(n) * 2000).astype(np.float32)
        ref = fbank_oracle.ref_fbank(pcm, 40, 16000, first_push=n // 3)
        got = fbank_oracle.fbank(pcm, 40)
        assert ref.shape == got.shape and float(np.abs(ref - got).max()) <= 2e-6


def test_short_input_yields_no_frames():
    assert fbank_oracle.fbank(np.zeros(399, np.float32)).shape == (0, 40)


@pytest.mark.parametrize("case", FBANK_CASES, ids=[c["name"] for c in FBANK_CASES])
def test_float64_evaluation_is_the_same_pipeline(case, fgolden):
    """fbank_f64 (the arbiter of the GPU fuzz test: is a difference the reference's own rounding noise?) is the pipeline of the
    goldens: within the reference's documented float32 FFT noise of them (2.4e-5 at 40 bins, 1.7e-4 at 80: tests/test_hip_fbank.py),
    and identical in shape and in the floor (log FLT_EPSILON) for silence."""
    pcm = fbank_input(case)
    sr = case["sample_rate"]
    got = np.stack([fbank_oracle.fbank_f64(p, case["num_bins"], sr, sr // 1000 * 25, sr // 1000 * 10) for p in pcm])
    ref = fgolden[case["name"]]
    assert got.shape == ref.shape
    assert float(np.abs(got - ref).max()) <= (4e-4 if case["num_bins"] == 80 else 1e-4)


def test_float64_evaluation_windows_and_floor():
    rng = np.random.default_rng(11)
    pcm = np.round(rng.standard_normal(4000) * 3000).astype(np.float32)
    for win in (0, 1):
        a = fbank_oracle.fbank(pcm, 40, 8000, 173, 61, win)
        b = fbank_oracle.fbank_f64(pcm, 40, 8000, 173, 61, win)
        assert a.shape == b.shape and float(np.abs(a - b).max()) <= 1e-4
    z = fbank_oracle.fbank_f64(np.zeros(800, np.float32), 23)
    assert np.array_equal(z.astype(np.float32), fbank_oracle.fbank(np.zeros(800, np.float32), 23))
