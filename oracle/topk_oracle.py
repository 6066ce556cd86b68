"""CPU oracle for the fused softmax + top-k  --  TEST INFRASTRUCTURE, NOT PRODUCT.

numpy restatement of the first beam prune's numeric part: ``logits.softmax(2)`` (wekws/bin/stream_kws_ctc.py:488)
followed by ``probs.topk(score_beam_size)`` per frame (wekws/model/loss.py:236-238).  Parity status: PINNED --
``tests/golden/make_topk_golden.py`` records torch's ``softmax(-1).topk(k)`` on seeded logits and checks, with the
reference's own ``ctc_prefix_beam_search``, that decoding from ONLY those k values per frame gives the same
hypotheses as decoding from the full posterior matrix; ``tests/test_topk_oracle.py`` checks this file against the
recorded values.
"""
import numpy as np


def softmax_topk(logits, k):
    """(..., K) -> (probs (..., k) float32 descending, index (..., k) int64); equal values: lower index first."""
    x = np.asarray(logits, np.float32)
    m = x.max(axis=-1, keepdims=True)
    e = np.exp((x - m).astype(np.float32)).astype(np.float32)
    p = (e / e.sum(axis=-1, keepdims=True, dtype=np.float32)).astype(np.float32)
    order = np.argsort(-x, axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(p, order, axis=-1), order.astype(np.int64)
