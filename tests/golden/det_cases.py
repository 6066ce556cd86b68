"""Seeded score tables for the DET scoring tests (shared by the golden generator, the oracle test and the GPU test)."""
import numpy as np

# (name, B, T, K, keyword column, window_shift, step, ragged lengths?)
CASES = [
    ("b16_t98_k2", 16, 98, 2, 1, 50, 0.01, False),
    ("b40_t98_k2_ragged", 40, 98, 2, 0, 50, 0.01, True),
    ("b9_t300_k12_ws7", 9, 300, 12, 5, 7, 0.05, True),
    ("b5_t1_k1", 5, 1, 1, 0, 50, 0.01, False),
    ("b33_t64_k3_ws1", 33, 64, 3, 2, 1, 0.1, True),
]


def case_data(name, B, T, K, kw, ws, step, ragged):
    g = np.random.default_rng([0xDE7, B, T, K])
    # posteriors with plateaus and exact ties (quantised) so that arg-max tie-breaking and `>=` edges are exercised
    s = g.random((B, T, K), dtype=np.float32)
    s = np.where(g.random((B, T, K)) < 0.5, np.round(s * 20) / 20, s).astype(np.float32)
    lengths = g.integers(0 if T > 1 else 1, T + 1, size=B).astype(np.int32) if ragged else np.full(B, T, np.int32)
    if ragged:
        lengths[0] = T
    is_kw = (np.arange(B) % 3 == 0)
    dur = float(sum(lengths[b] for b in range(B) if not is_kw[b])) * 0.01
    return s, lengths, is_kw, dur
