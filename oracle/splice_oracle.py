"""CPU oracle for context expansion + frame skip  --  TEST INFRASTRUCTURE, NOT PRODUCT.

numpy restatement of ``context_expansion`` and ``frame_skip`` of the reference's batch pipeline
(wekws/dataset/init_dataset.py:24-52 and :54-68; the per-utterance twins are
wekws/dataset/processor.py:267-311).  Parity status: PINNED -- ``tests/golden/make_splice_golden.py`` executes the
reference's own two functions (their source text, lifted out of init_dataset.py because the module's
``wenet`` import is not installable here) and commits the outputs as ``tests/golden/splice_golden.npz``;
``tests/test_splice_oracle.py`` checks this file against every one of them, bit for bit.
"""
import numpy as np


def context_expansion(feats, left=1, right=1):
    """init_dataset.py:24-52.  feats (B, T, D) -> (B, T - right, D * (left + right + 1)).
    Block ``lag`` of output frame t is input frame t + lag (torch.roll, :40-43); for t + lag < 0 the block is
    overwritten with frame 0 (the "replication pad" loop :45-48 copies feats_ctx[:, left, :D] = feats[:, 0]);
    the wrapped-around tail is cut by dropping the last ``right`` frames (:50)."""
    feats = np.asarray(feats, np.float32)
    B, T, D = feats.shape
    out = np.zeros((B, T, D * (left + right + 1)), np.float32)
    for k, lag in enumerate(range(-left, right + 1)):
        out[:, :, k * D:(k + 1) * D] = np.roll(feats, -lag, axis=1)
    for idx in range(left):
        for cpx in range(left - idx):
            out[:, idx, cpx * D:(cpx + 1) * D] = out[:, left, :D]
    return out[:, :T - right]


def frame_skip(feats, skip_rate=1):
    """init_dataset.py:54-68: keep every ``skip_rate``-th frame."""
    return np.asarray(feats)[:, ::skip_rate, :]


def splice_skip(feats, left, right, skip):
    return np.ascontiguousarray(frame_skip(context_expansion(feats, left, right), skip))


def lengths(feats_lengths, right, skip):
    """init_dataset.py:51 and :64-65."""
    return np.ceil((np.asarray(feats_lengths) - right) / skip).astype(np.int16)
