"""GPU: the DET scoring entries of the C ABI (wekws_hip_score_maxpool / wekws_hip_det_false_alarms, wekws_amd/det.py)
-- bit-exact against the Python oracle on maxima, first arg-max frames and alarm counts, and against the stats lines of
the reference's own loop (tests/golden/det_golden.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import det_oracle
from tests.golden.det_cases import CASES, case_data
from wekws_amd import det

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "det_golden.npz"))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_det_reductions_bit_exact(case):
    name, B, T, K, kw, ws, step, ragged = case
    s, lengths, is_kw, dur = case_data(*case)
    st = torch.from_numpy(s).cuda()
    lt = torch.from_numpy(lengths).cuda()
    mx, am = det.max_pool_scores(st, lt)
    rmx, ram = det_oracle.max_pool(s, lengths)
    assert np.array_equal(mx.cpu().numpy(), rmx) and np.array_equal(am.cpu().numpy(), ram)
    th = det.det_thresholds(step)
    assert th.tolist() == det_oracle.thresholds(step)
    al = det.false_alarm_counts(st, kw, th, ws, lt).cpu().numpy()
    want = np.asarray([[det_oracle.false_alarms(s[b, :lengths[b], kw].tolist(), t, ws) for t in th] for b in range(B)])
    assert np.array_equal(al, want)
    # with lengths = None every utterance is T frames long
    if not ragged:
        mx2, am2 = det.max_pool_scores(st)
        assert np.array_equal(mx2.cpu().numpy(), rmx) and np.array_equal(am2.cpu().numpy(), ram)
    # the stats file of compute_det.py, from the device reductions (keyword utterances with no frames are not in the table)
    ok = torch.from_numpy(lengths > 0)
    rows = det.det_stats(st[ok], lt[ok], is_kw[lengths > 0], kw, dur, step, ws)
    gold = GOLD[name + "/rows"]
    assert np.asarray(rows).shape == gold.shape and np.abs(np.asarray(rows) - gold).max() <= 5.1e-7


def test_det_on_model_posteriors():
    """End of the score.py -> compute_det.py chain on the device: forward -> max pool, vs the oracle on the copied scores."""
    from wekws_amd import pack
    from wekws_amd.model.kws_model import init_model
    from wekws_amd.utils import synth
    cfg = dict(synth.MODEL_CONFIGS["ds_tcn_h256"])
    m = init_model(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(pack.model_spec(cfg), 1234).items()})
    m = m.cuda().eval()
    x = torch.from_numpy(synth.synth_feats(64, 98, 40, seed=2)).cuda()
    y = m.posteriors(x)
    mx, am = det.max_pool_scores(y)
    rmx, ram = det_oracle.max_pool(y.cpu().numpy())
    assert np.array_equal(mx.cpu().numpy(), rmx) and np.array_equal(am.cpu().numpy(), ram)


def test_det_through_the_score_text_file():
    """the reference's chain is score.py -> '{:.6f}' text -> compute_det.py, so its comparisons see scores
    rounded to six decimals: 0.4999997 is written "0.500000" and counts as >= 0.5.  The `_text` scan (text_format=True)
    must reproduce that chain -- here recorded by formatting the float32 scores exactly as score.py:134-135 does, parsing
    them back like compute_det.py:53-57 and running the pinned oracle loops on the parsed lists -- including scores
    planted within 5e-7 of thresholds on both sides, where the plain float32 scan answers differently."""
    rng = np.random.default_rng(7)
    B, T, K, kw, ws, step = 24, 60, 2, 1, 5, 0.01
    s = rng.random((B, T, K)).astype(np.float32)
    th = det.det_thresholds(step)
    near = np.float32(th[rng.integers(1, len(th) - 1, size=(B, 12))])
    delta = np.float32(rng.choice([-4e-7, -2e-7, 2e-7, 4e-7], size=(B, 12)))
    cols = rng.integers(0, T, size=(B, 12))
    for b in range(B):
        s[b, cols[b], kw] = near[b] + delta[b]
    lengths = rng.integers(T // 2, T + 1, size=B).astype(np.int32)
    is_kw = rng.random(B) < 0.4
    dur = 3600.0
    st, lt = torch.from_numpy(s).cuda(), torch.from_numpy(lengths).cuda()
    parsed = [[float(tok) for tok in " ".join("{:.6f}".format(x) for x in s[b, :lengths[b], kw].tolist()).split()]
              for b in range(B)]                                   # score.py:134-135 -> compute_det.py:53-57
    want = np.asarray([[det_oracle.false_alarms(parsed[b], t, ws) for t in th] for b in range(B)])
    got_text = det.false_alarm_counts(st, kw, th, ws, lt, text_format=True).cpu().numpy()
    got_raw = det.false_alarm_counts(st, kw, th, ws, lt).cpu().numpy()
    assert np.array_equal(got_text, want)
    assert not np.array_equal(got_raw, want)                       # the planted scores do flip counts
    rows = det.det_stats(st, lt, is_kw, kw, dur, step, ws, text_format=True)
    # the stats rows from the parsed lists, with the reference's own loop structure (compute_det.py:79-104)
    ref_rows, frr, fah = [], 0.0, 0.0
    for t in th:
        nfr = sum(1 for b in range(B) if is_kw[b] and float(max(parsed[b])) < t)
        nfa = sum(det_oracle.false_alarms(parsed[b], t, ws) for b in range(B) if not is_kw[b])
        if is_kw.any():
            frr = nfr / int(is_kw.sum())
        fah = max(nfa, 1e-6) / (dur / 3600.0)
        ref_rows.append((float(t), fah, frr))
    assert rows == ref_rows


@pytest.mark.parametrize("seed", range(3))
def test_random_det_shapes(seed):
    """Seeded fuzz of the two reductions: random batch / length / keyword counts, ragged lengths (some empty), window sizes and
    threshold steps, bit-exact against the Python oracle."""
    rng = np.random.default_rng(800 + seed)
    for _ in range(12):
        B, T, K = int(rng.integers(1, 50)), int(rng.integers(1, 200)), int(rng.integers(1, 5))
        kw, ws, step = int(rng.integers(0, K)), int(rng.integers(1, 120)), float(rng.choice([0.01, 0.05, 0.13]))
        s = np.random.default_rng(int(rng.integers(0, 1000))).random((B, T, K), dtype=np.float32)
        lengths = rng.integers(0, T + 1, size=B).astype(np.int32)
        st, lt = torch.from_numpy(s).cuda(), torch.from_numpy(lengths).cuda()
        mx, am = det.max_pool_scores(st, lt)
        rmx, ram = det_oracle.max_pool(s, lengths)
        assert np.array_equal(mx.cpu().numpy(), rmx) and np.array_equal(am.cpu().numpy(), ram), (seed, B, T, K)
        th = det.det_thresholds(step)
        al = det.false_alarm_counts(st, kw, th, ws, lt).cpu().numpy()
        want = np.asarray([[det_oracle.false_alarms(s[b, :lengths[b], kw].tolist(), t, ws) for t in th] for b in range(B)])
        assert np.array_equal(al, want), (seed, B, T, K, kw, ws, step)


def test_empty_tables_die_like_the_reference():
    """No keyword utterance / no filler audio: the reference's stats loop dies with NameError (compute_det.py:97-104; executed beside
    the oracle in tests/test_det_oracle.py); det_stats raises the same."""
    s = torch.rand(4, 20, 2, device="cuda")
    ln = torch.full((4,), 20, dtype=torch.int32)
    with pytest.raises(NameError, match="false_reject_rate"):
        det.det_stats(s, ln, [False] * 4, 0, 36.0)
    with pytest.raises(NameError, match="false_alarm_per_hour"):
        det.det_stats(s, ln, [True, False, False, True], 0, 0.0)
    assert len(det.det_stats(s, ln, [True, False, False, True], 0, 36.0)) == len(det.det_thresholds(0.01))
    ln[3] = 0                                                  # a keyword utterance without frames: `max(score_list)` of an empty list
    with pytest.raises(ValueError, match="empty sequence"):
        det.det_stats(s, ln, [True, False, False, True], 0, 36.0)
    with pytest.raises(ValueError, match="empty sequence"):
        det_oracle.det_stats({"k0": [0.5], "k3": []}, {"f1": [0.1]}, 36.0)
    ln[3], ln[1] = 20, 0                                       # a filler utterance without frames is fine (its scan loop does nothing)
    assert len(det.det_stats(s, ln, [True, False, False, True], 0, 36.0)) == len(det.det_thresholds(0.01))
