"""GPU: NaN / +Inf / -Inf in the features or in a carried cache -- the HIP path returns what the reference returns
(torch.relu(nan) = nan and IEEE arithmetic: wekws/model/tcn.py:101-114, mdtc.py:95-121, kws_model.py:65-76): the same class
(finite / NaN / +Inf / -Inf) at every position of y and of the returned cache, finite values within the 1e-4 bar, and every OTHER
utterance of the batch bit-identical to the same call without the poison.  Goldens: tests/golden/nonfinite_golden.npz, recorded
from the live reference (make_nonfinite_golden.py); the oracle is pinned to them by tests/test_nonfinite_oracle.py.
How: wekws_amd/csrc/nonfinite.hip.h."""
import os

import numpy as np
import pytest
import torch

from tests.golden.nonfinite_cases import CASES, classify, poisoned_input
from tests.golden.cases import case_in_cache, case_input
from tests.helpers import case_weights, random_model_config as _random_model_config
from tests.test_hip_parity import build, run
from tests.test_nonfinite_oracle import run_oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def nf_golden():
    return np.load(os.path.join(HERE, "golden", "nonfinite_golden.npz"))


def same_classes_and_values(a, ref, what, tol=1e-4):
    assert a.shape == ref.shape, what
    ca, cr = classify(a), classify(ref)
    if not np.array_equal(ca, cr):
        bad = np.argwhere(ca != cr)
        raise AssertionError(f"{what}: {len(bad)} positions differ in class, first {bad[0].tolist()}: got {ca[tuple(bad[0])]} "
                             f"want {cr[tuple(bad[0])]}")
    fin = cr == 0
    if fin.any():
        err = float(np.abs(a[fin].astype(np.float64) - ref[fin]).max())
        assert err <= tol * max(1.0, float(np.abs(ref[fin]).max())), f"{what}: finite values off by {err:.3e}"


def poisoned_rows(case, cfg):
    """Batch indices that carry poison (GRU caches are (L, B, H): the batch index is the second one)."""
    gru = cfg["backbone"]["type"] == "gru"
    return sorted({(p[2] if (p[0] == "cache" and gru) else p[1]) for p in case["poison"]})


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_nonfinite_inputs_propagate_like_the_reference(case, precision, nf_golden):
    cfg, sd = case_weights(case)
    model = build(cfg, sd).set_precision(precision)
    x, cache0 = poisoned_input(case, cfg)
    y, cache = run(model, x, cache0, softmax=case.get("softmax", False), chunks=case.get("chunks"))
    # (a) the live reference's classes and y
    name = case["name"]
    same_classes_and_values(y, nf_golden[name + "/y"], "y vs reference golden")
    assert np.array_equal(classify(cache), nf_golden[name + "/cache_class"]), "cache classes vs reference golden"
    # (b) the oracle, values of the cache included
    ry, rc = run_oracle(case, cfg, sd, x, cache0)
    same_classes_and_values(y, ry, "y vs oracle")
    same_classes_and_values(cache, rc, "cache vs oracle")
    # (c) the utterances without poison: bit-identical to the same call on clean inputs
    xc, cc = case_input(case), case_in_cache(case, cfg)
    yc, cachec = run(model, xc, cc, softmax=case.get("softmax", False), chunks=case.get("chunks"))
    bad = poisoned_rows(case, cfg)
    clean = [b for b in range(case["B"]) if b not in bad]
    assert clean, "every case keeps at least one clean utterance"
    gru = cfg["backbone"]["type"] == "gru"
    cl, clc = (cache[:, clean], cachec[:, clean]) if gru else (cache[clean], cachec[clean])
    # One utterance per workgroup (every DS-TCN h256 kernel): the neighbours run the fast path, bit for bit.  Kernels that pack
    # 2 .. 4 utterances into a workgroup (hidden_dim <= 64 on the LDS-tile kernels, the MDTC streaming step) re-compute the
    # workgroup's other utterances with the same exact-f32 routine, and the GRU kernels share a per-step operand scale among 16
    # streams (the poisoned element enters as 0): there the neighbours are equal to fp32 rounding, not to the bit.
    if case["model"].startswith("ds_tcn_h256"):
        assert np.array_equal(y[clean], yc[clean]) and np.array_equal(cl, clc), "clean utterances changed"
    else:
        assert float(np.abs(y[clean].astype(np.float64) - yc[clean]).max()) <= 5e-6 * max(1.0, float(np.abs(yc[clean]).max()))
        assert float(np.abs(cl.astype(np.float64) - clc).max()) <= 5e-6 * max(1.0, float(np.abs(clc).max()))


@pytest.mark.parametrize("precision", ["f16x3", "f32", "f16"])
@pytest.mark.parametrize("seed", range(16))
def test_nonfinite_fuzz(seed, precision):
    """Random configurations around the kernel-family thresholds (zero-padded widths / kernel sizes, the any-shape path, every
    head), random batch / chunking / cache, a few poisoned elements: classes equal the oracle's everywhere."""
    from oracle import kws_oracle
    from wekws_amd import pack
    from wekws_amd.utils import synth
    rng = np.random.default_rng([0xBAD, seed])
    for trial in range(4):
        cfg, head = _random_model_config(rng)
        sd = synth.synth_state_dict(pack.model_spec(cfg), 4000 + 7 * seed + trial)
        model = build(cfg, sd).set_precision(precision)
        B, T = int(rng.integers(2, 6)), int(rng.integers(1, 140))
        x = synth.synth_feats(B, T, cfg["input_dim"], seed=seed, cmvn_like="cmvn" in cfg)
        vals = [np.nan, np.inf, -np.inf]
        for _ in range(int(rng.integers(1, 4))):
            x[int(rng.integers(1, B)), int(rng.integers(0, T)), int(rng.integers(0, cfg["input_dim"]))] = vals[int(rng.integers(0, 3))]
        gru = cfg["backbone"]["type"] == "gru"
        cache0 = np.zeros((cfg["backbone"]["num_layers"], B, cfg["hidden_dim"]), np.float32) if gru else None
        chunks = None
        if head == "linear" and T >= 4 and rng.random() < 0.5:
            cuts = sorted(set(int(c) for c in rng.integers(1, T, size=int(rng.integers(1, 4)))))
            chunks = [b - a for a, b in zip([0] + cuts, cuts + [T])]
        y, cache = run(model, x, cache0, chunks=chunks)
        with np.errstate(all="ignore"):
            ry, rc = (kws_oracle.forward_streaming(cfg, sd, x, chunks, cache0) if chunks else kws_oracle.forward(cfg, sd, x, cache0))
        tol = 1e-4 if precision != "f16" else 2e-2
        what = f"seed {seed} trial {trial} {precision} B={B} T={T} chunks={chunks} {cfg}"
        same_classes_and_values(y, ry, "y: " + what, tol)
        same_classes_and_values(cache, rc, "cache: " + what, tol)
        assert np.isfinite(y[0]).all(), "utterance 0 carries no poison: " + what
