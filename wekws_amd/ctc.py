"""Device side of the CTC keyword decoder: the first beam prune of ``ctc_prefix_beam_search``
(wekws/model/loss.py:236-251) -- per frame, the ``score_beam_size`` largest softmax posteriors and their token ids --
computed from the LOGITS in one kernel (``wekws_hip_softmax_topk``), so that a streaming caller
(wekws/bin/stream_kws_ctc.py:486-494) moves 8*k bytes per frame to the host instead of the whole posterior row.
The prefix search itself (loss.py:253-312) is host control logic and stays in the reference's Python."""
from __future__ import annotations

import ctypes
from typing import Tuple

import torch

from wekws_amd import _capi


def softmax_topk(logits: torch.Tensor, k: int = 3) -> Tuple[torch.Tensor, torch.Tensor]:
    """(..., K) float32 logits on a ROCm device -> (probs (..., k) float32, index (..., k) int64), i.e.
    ``logits.softmax(-1).topk(k)`` without materialising the softmax.  Equal values: lower index first."""
    if not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() < 1:
        raise ValueError("logits must be a float32 tensor on a ROCm device (no CPU fallback)")
    lib = _capi.load()
    x = logits.contiguous()
    K = int(x.size(-1))
    rows = x.numel() // K if K else 0
    probs = torch.empty(x.shape[:-1] + (k,), dtype=torch.float32, device=x.device)
    idx = torch.empty(x.shape[:-1] + (k,), dtype=torch.int32, device=x.device)
    if rows:
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _capi.check(lib.wekws_hip_softmax_topk(x.data_ptr(), rows, K, int(k), probs.data_ptr(), idx.data_ptr(),
                                               ctypes.c_void_p(stream)), "wekws_hip_softmax_topk")
    return probs, idx.to(torch.int64)


def first_beam_prune(logits: torch.Tensor, score_beam_size: int = 3, keywords_tokenset=None, min_prob: float = 0.05):
    """Per frame of a (T, K) logit matrix: the (prob, token) pairs that survive loss.py:236-251 -- top
    ``score_beam_size`` posteriors, prob > 0.05, token in ``keywords_tokenset`` if given.  Returns a list (one entry per
    frame) of lists of (prob, token); only T * k values cross PCIe."""
    probs, idx = softmax_topk(logits, score_beam_size)
    probs, idx = probs.cpu().tolist(), idx.cpu().tolist()
    out = []
    for pr, ix in zip(probs, idx):
        out.append([(p, i) for p, i in zip(pr, ix)
                    if p > min_prob and i >= 0 and (keywords_tokenset is None or i in keywords_tokenset)])
    return out
