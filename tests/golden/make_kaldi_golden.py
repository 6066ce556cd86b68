#!/usr/bin/env python3
"""Generate tests/golden/kaldi_golden.npz: the TRAINING-side features of the reference -- torchaudio.compliance.kaldi.fbank /
.mfcc as wekws/dataset/processor.py:134-203 calls them (Povey window, dither 0 at test time, waveform * (1 << 15)) -- from
INDEPENDENT third-party implementations that exist in this container:

  * log-mel: `transformers.audio_utils` (spectrogram / mel_filter_bank(mel_scale="kaldi", triangularize_in_mel_space=True) /
    window_function("povey")) in exactly the configuration Hugging Face's SeamlessM4TFeatureExtractor uses to REPLACE
    torchaudio.compliance.kaldi.fbank when torchaudio is absent (feature_extraction_seamless_m4t.py: frame 400, hop 160, FFT 512,
    power 2, no centring, pre-emphasis 0.97, DC removal, mel floor 1.192092955078125e-07, 20 Hz .. 8 kHz); its authors
    validate that port against torchaudio.  Version recorded in the file.
  * MFCC tail: scipy.fft.dct(type 2, norm "ortho") -- the matrix torchaudio.functional.create_dct builds, whose first column
    kaldi.mfcc overwrites with sqrt(1 / num_mel_bins), which IS the ortho value -- and the cepstral lifter written out from
    Kaldi's formula 1 + 0.5 Q sin(pi i / Q), Q = 22.

torchaudio itself is still not installable here (no network), so this is "pinned against an independent implementation of
the same published algorithm", one step short of "pinned against the reference's own dependency"; DESIGN.md says so.

    python tests/golden/make_kaldi_golden.py
"""
import os
import sys

import numpy as np
import scipy
import scipy.fft
import transformers
from transformers.audio_utils import mel_filter_bank, spectrogram, window_function

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from wekws_amd.utils import synth  # noqa: E402

CASES = [dict(name="noise_s3", kind="noise", seed=3, n=16000), dict(name="noise_s5", kind="noise", seed=5, n=16000),
         dict(name="sine", kind="sine", seed=0, n=16000), dict(name="ramp", kind="ramp", seed=0, n=12345),
         dict(name="short", kind="noise", seed=8, n=400), dict(name="silence", kind="silence", seed=0, n=4000)]


def case_pcm(c):
    if c["kind"] == "silence":
        return np.zeros(c["n"], np.float32)
    if c["kind"] == "ramp":                                    # an int16 ramp with a wrap: exercises DC removal and pre-emphasis
        return ((np.arange(c["n"]) * 37) % 30000 - 15000).astype(np.float32)
    return synth.synth_pcm(1, c["n"], seed=c["seed"], kind=c["kind"])[0]


def hf_kaldi_fbank(pcm, bins):
    window = window_function(400, "povey", periodic=False)
    mel = mel_filter_bank(num_frequency_bins=257, num_mel_filters=bins, min_frequency=20, max_frequency=8000, sampling_rate=16000,
                          norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    return spectrogram(pcm.astype(np.float64), window, frame_length=400, hop_length=160, fft_length=512, power=2.0, center=False,
                       preemphasis=0.97, mel_filters=mel, log_mel="log", mel_floor=1.192092955078125e-07,
                       remove_dc_offset=True).T.astype(np.float32)


def scipy_mfcc(logmel, num_ceps, q=22.0):
    c = scipy.fft.dct(logmel.astype(np.float64), type=2, norm="ortho", axis=-1)[:, :num_ceps]
    i = np.arange(num_ceps, dtype=np.float64)
    return (c * (1.0 + 0.5 * q * np.sin(np.pi * i / q))).astype(np.float32)


def main():
    out = {"versions": np.array(f"transformers {transformers.__version__}; scipy {scipy.__version__}; numpy {np.__version__}")}
    for c in CASES:
        pcm = case_pcm(c)
        for bins in (40, 80):
            f = hf_kaldi_fbank(pcm, bins)
            out[f"{c['name']}/fbank{bins}"] = f
            print(f"{c['name']:10s} fbank{bins} {f.shape}  range {f.min():8.3f} .. {f.max():8.3f}")
        out[f"{c['name']}/mfcc80"] = scipy_mfcc(out[f"{c['name']}/fbank80"], 80)
    path = os.path.join(HERE, "kaldi_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", str(out["versions"]))


if __name__ == "__main__":
    main()
