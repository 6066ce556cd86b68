"""CPU oracle for the DET scoring reductions  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Plain-Python restatement of wekws/bin/compute_det.py:79-106 (the threshold loop of its __main__ block) on score lists
as wekws/bin/score.py:128-137 produces them.  Parity status: PINNED -- tests/golden/make_det_golden.py lifts that very
loop out of the reference file with ``ast``, executes it unchanged on seeded score tables and records the stats lines;
tests/test_det_oracle.py checks this file against them.
"""
import numpy as np


def thresholds(step=0.01):
    out, th = [], 0.0                      # compute_det.py:78-79, :105
    while th <= 1.0:
        out.append(th)
        th += step
    return out


def max_pool(scores, lengths=None):
    """(B, T, K) -> (max (B, K) float32, first arg-max (B, K) int64): compute_det.py:84 `score = max(score_list)`."""
    s = np.asarray(scores, np.float32)
    B, T, K = s.shape
    mx = np.full((B, K), -np.inf, np.float32)
    am = np.full((B, K), -1, np.int64)
    for b in range(B):
        n = T if lengths is None else int(min(max(lengths[b], 0), T))
        for k in range(K):
            lst = s[b, :n, k].tolist()
            if lst:
                m = max(lst)
                mx[b, k] = m
                am[b, k] = lst.index(m)
    return mx, am


def false_alarms(score_list, threshold, window_shift):
    """compute_det.py:88-96 for one filler utterance and one threshold."""
    n, i = 0, 0
    while i < len(score_list):
        if score_list[i] >= threshold:
            n += 1
            i += window_shift
        else:
            i += 1
    return n


def det_stats(keyword_table, filler_table, filler_duration, step=0.01, window_shift=50):
    """The (threshold, false_alarm_per_hour, false_reject_rate) rows of compute_det.py:79-106."""
    rows = []
    # compute_det.py:97-104 assigns the two rates only under `if len(keyword_table) != 0` / `if filler_duration != 0` and then formats
    # them: with no keyword utterance / no filler audio the reference dies with NameError at its first row (arguments are evaluated
    # left to right: false_alarm_per_hour first)
    if filler_duration == 0:
        raise NameError("name 'false_alarm_per_hour' is not defined")
    if len(keyword_table) == 0:
        raise NameError("name 'false_reject_rate' is not defined")
    for threshold in thresholds(step):
        num_false_reject = sum(1 for sl in keyword_table.values() if float(max(sl)) < threshold)
        num_false_alarm = sum(false_alarms(sl, threshold, window_shift) for sl in filler_table.values())
        if len(keyword_table) != 0:
            false_reject_rate = num_false_reject / len(keyword_table)
        num_false_alarm = max(num_false_alarm, 1e-6)
        if filler_duration != 0:
            false_alarm_per_hour = num_false_alarm / (filler_duration / 3600.0)
        rows.append((threshold, false_alarm_per_hour, false_reject_rate))
    return rows
