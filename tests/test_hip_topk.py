"""GPU: wekws_hip_softmax_topk (the CTC decoder's first beam prune) against torch's softmax().topk() goldens, the
oracle on other shapes, and the model's own forward_softmax."""
import os

import numpy as np
import pytest
import torch

from oracle import topk_oracle
from tests.golden.topk_cases import CASES, case_logits
from wekws_amd import ctc, pack
from wekws_amd.model.kws_model import init_model
from wekws_amd.utils import synth

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "topk_golden.npz"))
TOL = 1e-6   # posteriors; well inside north_star's 1e-4


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_topk_golden(case):
    name, rows, K, k, scale = case
    p, i = ctc.softmax_topk(torch.from_numpy(case_logits(rows, K, scale)).cuda(), k)
    torch.cuda.synchronize()
    assert i.dtype == torch.int64 and np.array_equal(i.cpu().numpy(), GOLD[name + "/idx"])
    assert np.abs(p.cpu().numpy() - GOLD[name + "/probs"]).max() <= TOL


def test_topk_shapes_ties_and_errors():
    x = case_logits(1024 * 33, 2599, 2.0, seed=5)
    p, i = ctc.softmax_topk(torch.from_numpy(x).cuda().view(1024, 33, 2599), 3)
    rp, ri = topk_oracle.softmax_topk(x, 3)
    assert p.shape == (1024, 33, 3) and np.array_equal(i.cpu().numpy().reshape(-1, 3), ri)
    assert np.abs(p.cpu().numpy().reshape(-1, 3) - rp).max() <= TOL
    # equal values: lower index first; K < k: padded with (-1, 0)
    t = torch.tensor([[1.0, 5.0, 5.0, 0.0, 5.0]], device="cuda")
    p, i = ctc.softmax_topk(t, 3)
    assert i.tolist() == [[1, 2, 4]] and abs(float(p.sum()) - 3 * float(torch.softmax(t, -1)[0, 1])) < 1e-6
    p, i = ctc.softmax_topk(torch.tensor([[0.5, 1.5]], device="cuda"), 4)
    assert i.tolist() == [[1, 0, -1, -1]] and p[0, 2:].tolist() == [0.0, 0.0]
    with pytest.raises(Exception):
        ctc.softmax_topk(t, 9)
    with pytest.raises(ValueError):
        ctc.softmax_topk(torch.zeros(2, 3), 1)


def test_fsmn_logits_to_first_beam_prune():
    """FSMN-CTC logits -> fused prune == the reference call shape forward_softmax(...).topk(3) + the 0.05 filter."""
    cfg = dict(synth.MODEL_CONFIGS["fsmn_ctc"])
    m = init_model(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(pack.model_spec(cfg), 7).items()})
    m = m.cuda().eval()
    x = torch.from_numpy(synth.synth_feats(1, 40, 400, seed=3) * 4.0).cuda()
    logits, _ = m(x)
    probs, _ = m.forward_softmax(x)
    tv, ti = probs[0].topk(3)
    p, i = ctc.softmax_topk(logits[0], 3)
    assert torch.equal(i, ti) and float((p - tv).abs().max()) <= TOL
    pruned = ctc.first_beam_prune(logits[0], 3, keywords_tokenset=None)
    assert len(pruned) == 40
    for t, kept in enumerate(pruned):
        want = [(float(a), int(b)) for a, b in zip(tv[t].tolist(), ti[t].tolist()) if a > 0.05]
        assert [b for _, b in kept] == [b for _, b in want]


@pytest.mark.parametrize("seed", range(3))
def test_random_topk_shapes(seed):
    """Seeded fuzz: random row counts, class counts (below, at and far above the wave width), k and logit scales against the
    oracle: indices exact, probabilities <= 1e-6."""
    rng = np.random.default_rng(600 + seed)
    for _ in range(25):
        rows, K = int(rng.integers(1, 300)), int(rng.choice([1, 2, 3, 12, 63, 64, 65, 300, 2599, 5000]))
        k, scale = int(rng.integers(1, 9)), float(rng.choice([0.1, 1.0, 8.0, 40.0]))
        x = case_logits(rows, K, scale, seed=int(rng.integers(0, 1000)))
        p, i = ctc.softmax_topk(torch.from_numpy(x).cuda(), k)
        kk = min(k, K)
        rp, ri = topk_oracle.softmax_topk(x, kk)
        # (distinct random logits: no ties, so the order is unique; K < k: the library pads with (-1, 0))
        gi, gp = i.cpu().numpy(), p.cpu().numpy()
        assert gi.shape == (rows, k) and np.array_equal(gi[:, :kk], ri), (seed, rows, K, k, scale)
        assert np.abs(gp[:, :kk] - rp).max() <= TOL, (seed, rows, K, k, scale)
        assert (gi[:, kk:] == -1).all() and (gp[:, kk:] == 0).all(), (seed, rows, K, k, scale)
