"""Context-expansion / frame-skip parity cases shared by the golden generator and the tests."""
import numpy as np

# (name, B, T, F, left, right, skip)
CASES = [
    ("fsmn_ctc", 3, 98, 80, 2, 2, 3),        # examples/hi_xiaowen/s0/conf/fsmn_ctc.yaml:21-25
    ("fsmn_ctc_T100", 2, 100, 80, 2, 2, 3),
    ("expand_only", 2, 50, 40, 1, 1, 1),     # the function defaults
    ("skip_only", 2, 50, 40, 0, 0, 2),
    ("left_only", 1, 20, 40, 3, 0, 1),
    ("right_only", 1, 20, 40, 0, 3, 2),
    ("odd_dim", 2, 31, 23, 2, 1, 2),         # F not a multiple of 4 (scalar copy path)
    ("T_eq_left_plus_1", 1, 4, 8, 3, 1, 1),  # shortest input the reference's left-margin loop can index (T > left)
    ("big_skip", 1, 33, 16, 2, 2, 7),
]


def case_input(B, T, F, seed=0):
    g = np.random.default_rng([0x5911CE, seed, B, T, F])
    return g.standard_normal((B, T, F)).astype(np.float32)
