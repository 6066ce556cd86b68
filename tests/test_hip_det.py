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
