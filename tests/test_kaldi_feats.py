"""Training-side features (torchaudio.compliance.kaldi.fbank / .mfcc, wekws/dataset/processor.py:134-203).
PARITY UNPINNED on the reference side: torchaudio is not installable here and the reference ships no golden features.
What can be checked: two independent restatements agree with each other (oracle/fbank_oracle.c in Povey mode follows
the C++ runtime's code structure, oracle/kaldi_feats_oracle.py follows torchaudio's published algorithm in float64),
and the device path agrees with both."""
import numpy as np
import pytest

from oracle import fbank_oracle, kaldi_feats_oracle as kf
from wekws_amd.utils import synth


@pytest.mark.parametrize("kind", ["noise", "sine"])
@pytest.mark.parametrize("bins", [40, 80])
def test_two_restatements_of_povey_fbank_agree(kind, bins):
    pcm = synth.synth_pcm(1, 16000, seed=3, kind=kind)[0]
    a = fbank_oracle.fbank(pcm, bins, window=1)          # float32, recurrence twiddles like the runtime's fft.cc
    b = kf.fbank(pcm, bins)                              # float64 inside
    assert a.shape == b.shape == (98, bins)
    # the float32 FFT of the runtime-structured port is ~1e-4 from a float64 evaluation on near-empty bins
    assert float(np.abs(a - b).max()) <= 1e-3


def test_dct_matrix_properties():
    # orthonormal DCT-II except for Kaldi's rescaled first column; lifter leaves c0 alone
    m = kf.dct_matrix(80, 80)
    g = m.T @ m
    assert np.allclose(g[1:, 1:], np.eye(79), atol=1e-12)
    assert np.allclose(m[:, 0], np.sqrt(1.0 / 80))
    lif = kf.lifter_coeffs(80)
    assert lif[0] == 1.0 and np.isclose(lif[11], 1.0 + 11.0 * np.sin(np.pi * 11 / 22))
    assert kf.mfcc(np.zeros(16000, np.float32)).shape == (98, 80)
    assert kf.fbank(np.zeros(100, np.float32)).shape == (0, 40)            # shorter than one window


@pytest.mark.gpu
@pytest.mark.parametrize("bins", [40, 80])
def test_hip_povey_fbank(bins):
    import torch
    from wekws_amd.frontend import Fbank
    pcm = np.concatenate([synth.synth_pcm(3, 16000, seed=5, kind="noise"), synth.synth_pcm(1, 16000, kind="sine")])
    got = Fbank(bins, window="povey")(torch.from_numpy(pcm).cuda()).cpu().numpy()
    for i in range(pcm.shape[0]):
        # the kernel (exact twiddles) sits between the two: closer to the float64 evaluation than the float32
        # recurrence-twiddle port is (measured 1.3e-4 / 3.2e-4 vs 4.4e-4 / 5.6e-4 on these inputs)
        assert float(np.abs(got[i] - kf.fbank(pcm[i], bins)).max()) <= 5e-4
        assert float(np.abs(got[i] - fbank_oracle.fbank(pcm[i], bins, window=1)).max()) <= 1e-3


@pytest.mark.gpu
def test_hip_mfcc():
    import torch
    from wekws_amd.frontend import Mfcc, dct_lifter
    rng = np.random.default_rng(0)
    # the DCT / lifter kernel alone, ragged row counts, fewer cepstra than bins, lifter off
    for rows, nb, nc, q in ((1, 80, 80, 22.0), (37, 80, 80, 22.0), (1000, 40, 13, 22.0), (50, 23, 23, 0.0)):
        x = (rng.standard_normal((rows, nb)) * 5 + 10).astype(np.float32)
        got = dct_lifter(torch.from_numpy(x).cuda(), nc, q).cpu().numpy()
        ref = kf.dct_lifter(x, nc, q)
        assert got.shape == ref.shape and float(np.abs(got - ref).max()) <= 2e-5 * float(np.abs(ref).max())
    # end to end: pcm -> 80 MFCCs (the MDTC recipes' features)
    pcm = synth.synth_pcm(4, 16000, seed=9, kind="noise")
    got = Mfcc(80, 80)(torch.from_numpy(pcm).cuda()).cpu().numpy()
    assert got.shape == (4, 98, 80)
    for i in range(4):
        ref = kf.mfcc(pcm[i])
        assert float(np.abs(got[i] - ref).max()) <= 2e-3      # 80 log-mels at <= 4e-4 each through an orthonormal DCT + lifter <= 12
    with pytest.raises(Exception):
        dct_lifter(torch.zeros(4, 80, device="cuda"), 81)
