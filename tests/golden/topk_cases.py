"""softmax + top-k parity cases shared by the golden generator and the tests."""
import numpy as np

# (name, rows, K, k, scale)   scale: logit spread (larger = peakier posteriors)
CASES = [
    ("ctc2599_k3", 66, 2599, 3, 3.0),    # hi_xiaowen CTC vocabulary, score_beam_size 3 (loss.py:210)
    ("ctc2599_k8", 9, 2599, 8, 1.0),
    ("small_k1", 17, 20, 1, 2.0),
    ("k_eq_K", 5, 3, 3, 1.0),
    ("K_lt_64", 12, 37, 4, 4.0),
    ("wide", 3, 10007, 5, 2.0),
]


def case_logits(rows, K, scale, seed=0):
    g = np.random.default_rng([0x70B4, seed, rows, K])
    return (g.standard_normal((rows, K)) * scale).astype(np.float32)
