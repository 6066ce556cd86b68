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
    # degenerate lengths (found by the seeded fuzz over 1,000 more seeds, round 6): an utterance no longer than its right context
    ("T_eq_right", 2, 3, 8, 0, 3, 1),        # T - right = 0: empty
    ("T_lt_right", 2, 3, 79, 0, 4, 1),       # T - right = -1: the negative slice keeps 2 frames, right-hand blocks wrapped by torch.roll
    ("T_lt_right_skip", 3, 5, 40, 2, 7, 2),  # 2 T - right = 3 frames, every second kept, left margin inside
    ("T_lt_right_none", 1, 2, 8, 1, 4, 1),   # 2 T - right = 0: empty
    ("T_lt_right_far", 1, 2, 8, 0, 5, 1),    # 2 T - right < 0: empty
]

# left >= T (left >= 1): the reference's left-margin loop reads feats_ctx[:, left] and raises IndexError; recorded in the golden file
# as `raises/<name>` = 1 by make_splice_golden.py, mirrored by frontend.splice_skip (IndexError) and wekws_hip_splice (EINVAL)
RAISING = [
    ("left_eq_T", 2, 3, 8, 3, 0, 1),
    ("left_gt_T", 1, 3, 8, 4, 1, 2),
    ("left_gt_T_right_gt_T", 1, 2, 8, 2, 4, 1),
    ("T_one_left_one", 3, 1, 40, 1, 0, 1),
]


def case_input(B, T, F, seed=0):
    g = np.random.default_rng([0x5911CE, seed, B, T, F])
    return g.standard_normal((B, T, F)).astype(np.float32)
