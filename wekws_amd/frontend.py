"""Batched on-device front-end: log-mel filterbank (MI355X replacement for ``wenet::Fbank::Compute``,
runtime/core/frontend/fbank.h:138-198, behind ``FeaturePipeline::AcceptWaveform``'s framing rule,
runtime/core/frontend/feature_pipeline.cc:30-47) and the context-expansion / frame-skip step of the data pipeline
(wekws/dataset/init_dataset.py:24-68).  Thin host wrappers over the C ABI (wekws_hip_fbank_*, wekws_hip_splice);
no CPU fallback."""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from wekws_amd import _capi

WINDOWS = dict(hamming=0, povey=1)


class Fbank:
    """feats = Fbank(num_bins=40)(pcm)  with pcm (B, nsamp) float32 in int16 scale on a ROCm device
    (the runtime never divides by 32768: runtime/core/frontend/wav.h:98-102) -> (B, frames, num_bins)."""

    def __init__(self, num_bins: int = 40, sample_rate: int = 16000, frame_length: Optional[int] = None,
                 frame_shift: Optional[int] = None, window: str = "hamming", device="cuda"):
        # FeaturePipelineConfig: 25 ms window, 10 ms shift (feature_pipeline.h:34-39)
        self.num_bins = num_bins
        self.sample_rate = sample_rate
        self.frame_length = frame_length if frame_length is not None else sample_rate // 1000 * 25
        self.frame_shift = frame_shift if frame_shift is not None else sample_rate // 1000 * 10
        if window not in WINDOWS:
            raise ValueError(f"window must be one of {sorted(WINDOWS)}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("wekws_amd.frontend.Fbank runs on the MI355X HIP path only (no CPU fallback)")
        lib = _capi.load()
        cfg = _capi.FbankCfg()
        cfg.num_bins, cfg.sample_rate = num_bins, sample_rate
        cfg.frame_length, cfg.frame_shift, cfg.window = self.frame_length, self.frame_shift, WINDOWS[window]
        self._lib = lib
        self._ptr = ctypes.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _capi.check(lib.wekws_hip_fbank_create(ctypes.byref(cfg), idx, ctypes.byref(self._ptr)), "wekws_hip_fbank_create")

    def __del__(self):
        try:
            if self._ptr:
                self._lib.wekws_hip_fbank_destroy(self._ptr)
                self._ptr = ctypes.c_void_p()
        except Exception:
            pass

    def num_frames(self, nsamp: int) -> int:
        return int(self._lib.wekws_hip_fbank_num_frames(self._ptr, int(nsamp)))

    def __call__(self, pcm: torch.Tensor) -> torch.Tensor:
        """(B, nsamp) PCM on a ROCm device -> (B, frames, num_bins) log-mel.  float32 in int16 scale (wav.h:98-102), or
        int16 as FeaturePipeline::AcceptWaveform(std::vector<int16_t>) receives it (feature_pipeline.cc:49-55): widened
        in the kernel's registers -- 2 bytes per sample over PCIe and HBM, the same features bit for bit."""
        if pcm.dim() != 2 or not pcm.is_cuda or pcm.dtype not in (torch.float32, torch.int16):
            raise ValueError("pcm must be a (B, nsamp) float32 or int16 tensor on a ROCm device")
        pcm = pcm.contiguous()
        B, n = int(pcm.size(0)), int(pcm.size(1))
        nf = self.num_frames(n)
        feats = torch.empty((B, nf, self.num_bins), dtype=torch.float32, device=pcm.device)
        if B and nf:
            stream = torch.cuda.current_stream(pcm.device).cuda_stream
            fn = self._lib.wekws_hip_fbank_compute if pcm.dtype == torch.float32 else self._lib.wekws_hip_fbank_compute_i16
            _capi.check(fn(self._ptr, pcm.data_ptr(), B, n, feats.data_ptr(), ctypes.c_void_p(stream)), "wekws_hip_fbank_compute")
        return feats

    def leftover(self, nsamp: int) -> Tuple[int, int]:
        """(frames, first sample to keep) for a streaming caller: the reference keeps
        samples[frame_shift * frames:] for the next push (feature_pipeline.cc:41-44)."""
        nf = self.num_frames(nsamp)
        return nf, self.frame_shift * nf


def splice_skip(feats: torch.Tensor, left: int = 0, right: int = 0, skip: int = 1,
                feats_lengths: Optional[torch.Tensor] = None):
    """context_expansion(left, right) followed by frame_skip(skip) (wekws/dataset/init_dataset.py:24-68) as ONE
    gather on the device: (B, T, F) -> (B, ceil((T - right) / skip), (left + right + 1) * F).
    With ``feats_lengths`` also returns the reference's updated lengths (:51 then :64-65).
    Degenerate lengths behave like the reference: ``left >= T`` raises IndexError (its left-margin loop), ``T < right``
    returns the ``2 T - right`` wrapped frames its negative slice keeps (include/wekws_hip.h)."""
    if feats.dim() != 3 or not feats.is_cuda or feats.dtype != torch.float32:
        raise ValueError("feats must be a (B, T, F) float32 tensor on a ROCm device (no CPU fallback)")
    lib = _capi.load()
    feats = feats.contiguous()
    B, T, F = (int(v) for v in feats.shape)
    if left >= 1 and left >= T:                 # init_dataset.py:45-48 reads feats_ctx[:, left]: the reference's own error
        raise IndexError(f"index {int(left)} is out of bounds for dimension 1 with size {T}")
    To = int(lib.wekws_hip_splice_frames(T, int(right), int(skip)))
    out = torch.empty((B, To, (left + right + 1) * F), dtype=torch.float32, device=feats.device)
    if B and To:
        stream = torch.cuda.current_stream(feats.device).cuda_stream
        _capi.check(lib.wekws_hip_splice(feats.data_ptr(), B, T, F, int(left), int(right), int(skip), out.data_ptr(),
                                         ctypes.c_void_p(stream)), "wekws_hip_splice")
    if feats_lengths is None:
        return out
    lens = torch.ceil((feats_lengths - right) / skip).to(dtype=torch.int16)
    return out, lens


def context_expansion(feats: torch.Tensor, left: int = 1, right: int = 1) -> torch.Tensor:
    """wekws/dataset/init_dataset.py:24-52 on a (B, T, F) device batch."""
    return splice_skip(feats, left, right, 1)


def frame_skip(feats: torch.Tensor, skip_rate: int = 1) -> torch.Tensor:
    """wekws/dataset/init_dataset.py:54-68 on a (B, T, F) device batch."""
    return splice_skip(feats, 0, 0, skip_rate)


def dct_lifter(logmel: torch.Tensor, num_ceps: int, cepstral_lifter: float = 22.0) -> torch.Tensor:
    """The MFCC tail of torchaudio.compliance.kaldi.mfcc (DCT-II 'ortho' with Kaldi's first column, first ``num_ceps``
    cepstra, cepstral lifter) on (..., num_mel_bins) log-mel rows -> (..., num_ceps); wekws_hip_dct_lifter."""
    if not logmel.is_cuda or logmel.dtype != torch.float32:
        raise ValueError("logmel must be a float32 tensor on a ROCm device (no CPU fallback)")
    lib = _capi.load()
    logmel = logmel.contiguous()
    nb = int(logmel.shape[-1])
    rows = logmel.numel() // nb if nb else 0
    out = torch.empty(tuple(logmel.shape[:-1]) + (int(num_ceps),), dtype=torch.float32, device=logmel.device)
    stream = torch.cuda.current_stream(logmel.device).cuda_stream
    _capi.check(lib.wekws_hip_dct_lifter(logmel.data_ptr(), rows, nb, int(num_ceps), float(cepstral_lifter),
                                         out.data_ptr(), ctypes.c_void_p(stream)), "wekws_hip_dct_lifter")
    return out


class Mfcc:
    """``kaldi.mfcc(waveform * (1 << 15), num_ceps, num_mel_bins, frame_length, frame_shift, energy_floor=0.0,
    sample_frequency)`` of the reference's MDTC recipes (wekws/dataset/processor.py:134-169) on the device: Povey-window
    fbank (wekws_hip_fbank_*) followed by the DCT / lifter kernel.  ``pcm`` is (B, nsamp) float32 already in int16 scale.
    Parity with torchaudio is unpinned (see oracle/kaldi_feats_oracle.py)."""

    def __init__(self, num_ceps: int = 80, num_mel_bins: int = 80, sample_rate: int = 16000, cepstral_lifter: float = 22.0,
                 device="cuda"):
        if not 0 < num_ceps <= num_mel_bins:
            raise ValueError("need 0 < num_ceps <= num_mel_bins")
        self.num_ceps, self.cepstral_lifter = int(num_ceps), float(cepstral_lifter)
        self.fbank = Fbank(num_mel_bins, sample_rate, window="povey", device=device)

    def __call__(self, pcm: torch.Tensor) -> torch.Tensor:
        return dct_lifter(self.fbank(pcm), self.num_ceps, self.cepstral_lifter)
