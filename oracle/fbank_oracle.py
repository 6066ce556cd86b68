"""ctypes access to the fbank oracles  --  TEST INFRASTRUCTURE, NOT PRODUCT (see oracle/fbank_oracle.c).

fbank(wave, num_bins)      the plain-C restatement (oracle/_build/libfbank_oracle.so)
ref_fbank(wave, num_bins)  the reference's own C++ front-end compiled into oracle/_ref/ (None if absent)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = os.path.join(_HERE, "_build", "libfbank_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_fbank.so")
_port = _ref = None


def _load_port():
    global _port
    if _port is None:
        if not os.path.exists(_PORT):
            raise RuntimeError(f"{_PORT} missing: run `make -C oracle`")
        _port = C.CDLL(_PORT)
        _port.wekws_oracle_fbank.restype = C.c_int
        _port.wekws_oracle_fbank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return _port


def have_ref() -> bool:
    return os.path.exists(_REF)


def num_frames(nsamp, frame_length=400, frame_shift=160):
    return 0 if nsamp < frame_length else 1 + (nsamp - frame_length) // frame_shift


def fbank(wave, num_bins=40, sample_rate=16000, frame_length=400, frame_shift=160, window=0):
    wave = np.ascontiguousarray(wave, dtype=np.float32)
    nf = num_frames(wave.size, frame_length, frame_shift)
    out = np.empty((nf, num_bins), np.float32)
    if nf:
        n = _load_port().wekws_oracle_fbank(wave.ctypes.data, wave.size, num_bins, sample_rate, frame_length,
                                            frame_shift, window, out.ctypes.data)
        assert n == nf
    return out


def ref_fbank(wave, num_bins=40, sample_rate=16000, first_push=0):
    """Reference front-end (25 ms / 10 ms framing fixed by FeaturePipelineConfig, feature_pipeline.h:34-39)."""
    global _ref
    if not have_ref():
        return None
    if _ref is None:
        _ref = C.CDLL(_REF)
        _ref.ref_fbank.restype = C.c_int
        _ref.ref_fbank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    wave = np.ascontiguousarray(wave, dtype=np.float32)
    nf = num_frames(wave.size, sample_rate // 1000 * 25, sample_rate // 1000 * 10)
    out = np.empty((nf, num_bins), np.float32)
    n = _ref.ref_fbank(wave.ctypes.data, wave.size, num_bins, sample_rate, first_push, out.ctypes.data)
    assert n == nf, (n, nf)
    return out


def has_empty_filter(num_bins=40, sample_rate=16000, frame_length=400):
    """Does a triangular filter of this bank cover no FFT bin?  The reference's constructor CHECK-fails then (fbank.h:51-81;
    same float32 arithmetic as there), and wekws_hip_fbank_create refuses the configuration."""
    n = 1
    while n < frame_length:
        n *= 2
    f32 = np.float32
    mel = lambda f: f32(1127.0) * np.log(f32(1.0) + f32(f) / f32(700.0), dtype=np.float32)   # noqa: E731
    lo, hi = mel(f32(20.0)), mel(f32(sample_rate // 2))
    delta = f32((hi - lo) / f32(num_bins + 1))
    width = f32(sample_rate) / f32(n)
    m = np.array([mel(width * f32(i)) for i in range(n // 2)], np.float32)
    for b in range(num_bins):
        left, right = lo + f32(b) * delta, lo + f32(b + 2) * delta
        if not np.any((m > left) & (m < right)):
            return True
    return False


def fbank_f64(wave, num_bins=40, sample_rate=16000, frame_length=400, frame_shift=160, window=0):
    """The same pipeline evaluated in float64 (tables as the reference builds them: float32 mel weights, the window rounded to
    float32; per-frame arithmetic and the FFT in double): what both the reference's float32 recurrence-twiddle FFT and the HIP
    kernel's exactly-rounded one approximate.  Used by the fuzz tests to tell the reference's own rounding noise from a defect."""
    wave = np.ascontiguousarray(wave, dtype=np.float32)
    nf = num_frames(wave.size, frame_length, frame_shift)
    n = 1
    while n < frame_length:
        n *= 2
    nb = n // 2
    f32 = np.float32
    mel = lambda f: f32(1127.0) * np.log(f32(1.0) + f32(f) / f32(700.0), dtype=np.float32)   # noqa: E731
    lo, hi = mel(f32(20.0)), mel(f32(sample_rate // 2))
    delta = f32((hi - lo) / f32(num_bins + 1))
    width = f32(sample_rate) / f32(n)
    m = np.array([mel(width * f32(i)) for i in range(nb)], np.float32)
    W = np.zeros((num_bins, nb), np.float32)
    for b in range(num_bins):
        left, center, right = f32(lo + f32(b) * delta), f32(lo + f32(b + 1) * delta), f32(lo + f32(b + 2) * delta)
        inside = (m > left) & (m < right)
        up = (m - left) / (center - left)
        down = (right - m) / (right - center)
        W[b] = np.where(inside, np.where(m <= center, up, down), f32(0)).astype(np.float32)
    a = 2.0 * np.pi / (frame_length - 1)
    i = np.arange(frame_length, dtype=np.float64)
    win = (0.54 - 0.46 * np.cos(a * i)) if window == 0 else np.power(0.5 - 0.5 * np.cos(a * i), 0.85)
    win = win.astype(np.float32).astype(np.float64)
    out = np.empty((nf, num_bins), np.float64)
    for f in range(nf):
        x = wave[f * frame_shift:f * frame_shift + frame_length].astype(np.float64)
        x = x - x.mean()
        c = np.float64(np.float32(0.97))                             # fbank.h:122-127 (the constant is 0.97f)
        x = np.concatenate([[x[0] - c * x[0]], x[1:] - c * x[:-1]])
        spec = np.fft.fft(np.concatenate([x * win, np.zeros(n - frame_length)]))[:nb]
        power = spec.real ** 2 + spec.imag ** 2
        e = W.astype(np.float64) @ power
        out[f] = np.log(np.maximum(e, np.finfo(np.float32).eps))
    return out
