"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the training-side features: torchaudio.compliance.kaldi.fbank / .mfcc.

PARITY PINNED TO A THIRD-PARTY PORT (not to torchaudio itself).  The reference computes its training / scoring features with torchaudio
(wekws/dataset/processor.py:134-203: `kaldi.fbank(waveform * (1 << 15), num_mel_bins, frame_length, frame_shift, dither,
energy_floor=0.0, sample_frequency)` and `kaldi.mfcc(..., num_ceps, num_mel_bins, ...)`; streaming twin
wekws/bin/stream_kws_ctc.py:354-360).  torchaudio is a requirements.txt dependency without a pinned version and is not
installed in this environment (no network), and the reference ships no golden features.  PINNED, since round 5, against an
INDEPENDENT implementation of the same algorithm that IS present here: Hugging Face transformers' numpy port of kaldi.fbank
(transformers.audio_utils, what SeamlessM4TFeatureExtractor uses when torchaudio is missing; validated by its authors against
torchaudio) plus scipy's DCT -- tests/golden/kaldi_golden.npz, tests/golden/make_kaldi_golden.py; this file agrees with it to
float32 rounding (<= 2e-6 on the log-mel).  Not checked against torchaudio ITSELF: "HF's port == torchaudio" is its authors'
claim.  This file restates the published algorithm of torchaudio/compliance/kaldi.py
(torchaudio 2.x, BSD-2; itself a port of Kaldi's feature-window.cc / mel-computations.cc / feature-mfcc.cc) with the
defaults those two calls leave in place:
    window_type 'povey' (hann(periodic=False) ** 0.85), remove_dc_offset, preemphasis 0.97 with a replicated first
    sample, snip_edges (1 + (n - 400) // 160 frames), round_to_power_of_two (512-point real FFT), power spectrum,
    mel banks 20 Hz .. Nyquist on the 1127 ln(1 + f/700) scale with triangular weights in the mel domain, log of
    max(energy, float32 eps); MFCC = log-mel @ DCT-II('ortho', first column sqrt(1/num_mel_bins))[:, :num_ceps], times
    the cepstral lifter 1 + 11 sin(pi i / 22); no energy term, no mean subtraction, dither 0.
It serves two purposes: an independent cross-check of oracle/fbank_oracle.c's Povey mode (which follows the C++
runtime's code structure instead), and the checker for the on-device DCT / lifter step.  Float64 inside, float32 out.
"""
import numpy as np

EPS = float(np.finfo(np.float32).eps)


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, np.float64) / 700.0)


def mel_banks(num_bins, padded, sample_rate, low_freq=20.0, high_freq=0.0):
    """get_mel_banks: (num_bins, padded // 2) triangular weights (the Nyquist column is added as zeros by fbank)."""
    nfft = padded // 2
    nyquist = 0.5 * sample_rate
    if high_freq <= 0.0:
        high_freq += nyquist
    width = sample_rate / padded
    lo, hi = mel_scale(low_freq), mel_scale(high_freq)
    delta = (hi - lo) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = lo + b * delta, lo + (b + 1.0) * delta, lo + (b + 2.0) * delta
    mel = mel_scale(width * np.arange(nfft, dtype=np.float64))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return np.maximum(0.0, np.minimum(up, down))


def _frames(wave, frame_length, frame_shift):
    wave = np.asarray(wave, np.float64)
    n = wave.shape[-1]
    m = 0 if n < frame_length else 1 + (n - frame_length) // frame_shift
    idx = np.arange(frame_length)[None, :] + frame_shift * np.arange(m)[:, None]
    return wave[idx]                                                     # (m, frame_length)


def fbank(wave, num_mel_bins=40, sample_rate=16000, frame_length_ms=25.0, frame_shift_ms=10.0):
    """wave: (n,) samples already in int16 scale (the reference multiplies by 1 << 15).  -> (frames, num_mel_bins)."""
    win, shift = int(sample_rate * frame_length_ms * 0.001), int(sample_rate * frame_shift_ms * 0.001)
    padded = 1 << (win - 1).bit_length()
    x = _frames(wave, win, shift)
    if x.shape[0] == 0:
        return np.zeros((0, num_mel_bins), np.float32)
    x = x - x.mean(axis=1, keepdims=True)                                # remove_dc_offset
    prev = np.concatenate([x[:, :1], x[:, :-1]], axis=1)                 # replicate-padded shift by one
    x = x - 0.97 * prev                                                  # preemphasis
    hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win) / (win - 1))  # torch.hann_window(periodic=False)
    x = x * hann ** 0.85                                                 # povey
    x = np.concatenate([x, np.zeros((x.shape[0], padded - win))], axis=1)
    power = np.abs(np.fft.rfft(x, axis=1)) ** 2                          # (m, padded/2 + 1)
    banks = np.concatenate([mel_banks(num_mel_bins, padded, sample_rate), np.zeros((num_mel_bins, 1))], axis=1)
    return np.log(np.maximum(power @ banks.T, EPS)).astype(np.float32)


def dct_matrix(num_ceps, num_mel_bins):
    """_get_dct_matrix: create_dct(num_mel_bins, num_mel_bins, 'ortho') with column 0 := sqrt(1/num_mel_bins)."""
    n = np.arange(num_mel_bins, dtype=np.float64)
    k = np.arange(num_mel_bins, dtype=np.float64)[:, None]
    dct = np.cos(np.pi / num_mel_bins * (n + 0.5) * k)                   # (k, n)
    dct[0] *= 1.0 / np.sqrt(2.0)
    dct *= np.sqrt(2.0 / num_mel_bins)
    m = dct.T.copy()                                                     # (n_mels, n_mfcc): right-multiplied
    m[:, 0] = np.sqrt(1.0 / num_mel_bins)
    return m[:, :num_ceps]


def lifter_coeffs(num_ceps, cepstral_lifter=22.0):
    return 1.0 + 0.5 * cepstral_lifter * np.sin(np.pi * np.arange(num_ceps) / cepstral_lifter)


def dct_lifter(logmel, num_ceps, cepstral_lifter=22.0):
    """The MFCC tail on given log-mel rows: (rows, num_mel_bins) -> (rows, num_ceps)."""
    logmel = np.asarray(logmel, np.float64)
    out = logmel @ dct_matrix(num_ceps, logmel.shape[-1])
    if cepstral_lifter != 0.0:
        out = out * lifter_coeffs(num_ceps, cepstral_lifter)
    return out.astype(np.float32)


def mfcc(wave, num_ceps=80, num_mel_bins=80, sample_rate=16000, frame_length_ms=25.0, frame_shift_ms=10.0):
    return dct_lifter(fbank(wave, num_mel_bins, sample_rate, frame_length_ms, frame_shift_ms), num_ceps)
