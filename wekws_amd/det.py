"""Device side of the DET evaluation: the score reductions of wekws/bin/compute_det.py:79-106 on the per-frame
posteriors that wekws/bin/score.py:128-137 would write to a text file -- per-utterance maxima (false rejects) and
sliding-window alarm counts (false alarms) -- so that an evaluation loop keeps the (B, T, K) score matrix on the GPU.
Comparisons only: bit-exact with the reference's Python on the same float32 scores."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from wekws_amd import _capi


def det_thresholds(step: float = 0.01) -> np.ndarray:
    """The thresholds compute_det.py:78-79,105 visits: `threshold = 0.0; while threshold <= 1.0: ...; threshold += step`
    (float64 accumulation, so the list is exactly the reference's)."""
    out, th = [], 0.0
    while th <= 1.0:
        out.append(th)
        th += step
    return np.asarray(out, np.float64)


def _check(scores: torch.Tensor, lengths: Optional[torch.Tensor]):
    if not scores.is_cuda or scores.dtype != torch.float32 or scores.dim() != 3:
        raise ValueError("scores must be a (B, T, K) float32 tensor on a ROCm device (no CPU fallback)")
    s = scores.contiguous()
    ln = None
    if lengths is not None:
        ln = lengths.to(device=s.device, dtype=torch.int32).contiguous()
        if ln.numel() != s.size(0):
            raise ValueError("lengths must have one entry per utterance")
    return s, ln


def max_pool_scores(scores: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(B, T, K) posteriors -> (max (B, K) float32, first arg-max frame (B, K) int64): `max(score_list)` of
    compute_det.py:84 for every utterance and keyword, over the first lengths[b] frames (score.py:131)."""
    s, ln = _check(scores, lengths)
    B, T, K = (int(v) for v in s.shape)
    mx = torch.empty((B, K), dtype=torch.float32, device=s.device)
    am = torch.empty((B, K), dtype=torch.int32, device=s.device)
    if B and T and K:
        stream = torch.cuda.current_stream(s.device).cuda_stream
        _capi.check(_capi.load().wekws_hip_score_maxpool(s.data_ptr(), B, T, K, ln.data_ptr() if ln is not None else None,
                                                         mx.data_ptr(), am.data_ptr(), ctypes.c_void_p(stream)),
                    "wekws_hip_score_maxpool")
    return mx, am.to(torch.int64)


def round_like_score_file(values: np.ndarray) -> np.ndarray:
    """float64 values as compute_det.py sees scores after score.py:134-135 wrote them as '{:.6f}' text."""
    return np.asarray([float("{:.6f}".format(v)) for v in np.asarray(values, np.float64).ravel()],
                      np.float64).reshape(np.shape(values))


def false_alarm_counts(scores: torch.Tensor, keyword: int, thresholds: Sequence[float], window_shift: int = 50,
                       lengths: Optional[torch.Tensor] = None, text_format: bool = False) -> torch.Tensor:
    """(B, T, K) posteriors -> (B, n_thr) int32 alarm counts of column `keyword`: the scan of compute_det.py:88-96.
    text_format: compare the scores rounded to six decimals, i.e. as the reference's score text file carries them."""
    s, ln = _check(scores, lengths)
    B, T, K = (int(v) for v in s.shape)
    th = torch.from_numpy(np.ascontiguousarray(thresholds, np.float64)).to(s.device)
    out = torch.empty((B, th.numel()), dtype=torch.int32, device=s.device)
    if B and T and K:
        stream = torch.cuda.current_stream(s.device).cuda_stream
        lib = _capi.load()
        fn = lib.wekws_hip_det_false_alarms_text if text_format else lib.wekws_hip_det_false_alarms
        _capi.check(fn(s.data_ptr(), B, T, K, int(keyword), ln.data_ptr() if ln is not None else None, th.data_ptr(),
                       int(th.numel()), int(window_shift), out.data_ptr(), ctypes.c_void_p(stream)),
                    "wekws_hip_det_false_alarms")
    return out


def det_stats(scores: torch.Tensor, lengths: Optional[torch.Tensor], is_keyword: Sequence[bool], keyword: int,
              filler_duration: float, step: float = 0.01, window_shift: int = 50,
              text_format: bool = False) -> List[Tuple[float, float, float]]:
    """The rows compute_det.py:97-104 writes to its stats file -- (threshold, false alarms per hour, false reject rate)
    -- for one keyword column of a scored batch: `is_keyword[b]` says whether utterance b's transcript is the keyword
    (compute_det.py:45-50).  Only the (B,) maxima and the (B, n_thr) counts leave the device.
    text_format=True reproduces the stats file of the reference's score.py -> compute_det.py chain bit for bit: there the
    scores pass through '{:.6f}' text (score.py:134-135) before they are compared."""
    kw = np.asarray(list(is_keyword), bool)
    # the reference assigns the two rates only under `if len(keyword_table) != 0` / `if filler_duration != 0` (compute_det.py:97-101)
    # and then formats them: an evaluation set without a keyword utterance, or without filler audio, dies there with NameError
    if filler_duration == 0:
        raise NameError("name 'false_alarm_per_hour' is not defined")
    if not kw.any():
        raise NameError("name 'false_reject_rate' is not defined")
    if lengths is not None and bool((lengths.cpu().numpy()[kw] <= 0).any()):
        raise ValueError("max() arg is an empty sequence")       # compute_det.py:84 `max(score_list)` of a keyword utterance without frames
    th = det_thresholds(step)
    mx, _ = max_pool_scores(scores, lengths)
    alarms = false_alarm_counts(scores, keyword, th, window_shift, lengths, text_format=text_format).cpu().numpy()
    mk = mx[:, keyword].cpu().numpy().astype(np.float64)[kw]
    if text_format:
        mk = round_like_score_file(mk)                       # max and a monotonic rounding commute
    rows = []
    false_reject_rate = false_alarm_per_hour = 0.0
    for j, t in enumerate(th):
        num_false_reject = int((mk < t).sum())
        num_false_alarm = int(alarms[~kw, j].sum())
        if kw.any():
            false_reject_rate = num_false_reject / int(kw.sum())
        num_false_alarm = max(num_false_alarm, 1e-6)
        if filler_duration != 0:
            false_alarm_per_hour = num_false_alarm / (filler_duration / 3600.0)
        rows.append((float(t), float(false_alarm_per_hour), float(false_reject_rate)))
    return rows
