"""GPU: wekws_hip_splice (context expansion + frame skip) against the reference goldens and the oracle; the op is
a gather, so every comparison is bit-exact.  Also the fbank -> splice -> FSMN chain against the oracles."""
import os

import numpy as np
import pytest
import torch

from oracle import kws_oracle, splice_oracle
from tests.golden.splice_cases import CASES, RAISING, case_input
from wekws_amd import frontend, pack
from wekws_amd.model.kws_model import init_model
from wekws_amd.utils import synth

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "splice_golden.npz"))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_splice_golden(case):
    name, B, T, F, left, right, skip = case
    x = torch.from_numpy(case_input(B, T, F)).cuda()
    y, lens = frontend.splice_skip(x, left, right, skip, feats_lengths=torch.full((B,), T, dtype=torch.int32))
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), GOLD[name + "/y"])
    assert np.array_equal(lens.numpy(), GOLD[name + "/lens"])


def test_splice_large_and_edges():
    x = case_input(512, 98, 80, seed=9)
    xt = torch.from_numpy(x).cuda()
    y = frontend.splice_skip(xt, 2, 2, 3).cpu().numpy()
    assert np.array_equal(y, splice_oracle.splice_skip(x, 2, 2, 3))
    assert np.array_equal(frontend.context_expansion(xt[:4], 1, 1).cpu().numpy(), splice_oracle.context_expansion(x[:4], 1, 1))
    assert np.array_equal(frontend.frame_skip(xt[:4], 3).cpu().numpy(), splice_oracle.frame_skip(x[:4], 3))
    # T == right: nothing left; B = 0
    assert frontend.splice_skip(xt[:2, :2], 1, 2, 3).shape == (2, 0, 320)
    assert frontend.splice_skip(xt[:0], 2, 2, 3).shape == (0, 32, 400)
    # unaligned view -> the scalar path
    z = frontend.splice_skip(xt[:3, 1:, :][:, :, :79].contiguous(), 1, 2, 2).cpu().numpy()
    assert np.array_equal(z, splice_oracle.splice_skip(x[:3, 1:, :79], 1, 2, 2))
    with pytest.raises(ValueError):
        frontend.splice_skip(torch.zeros(1, 4, 8), 1, 1, 1)


@pytest.mark.parametrize("case", RAISING, ids=[c[0] for c in RAISING])
def test_left_context_not_shorter_than_the_utterance(case):
    """left >= T: IndexError like the reference (recorded in the golden file); the C ABI refuses with EINVAL."""
    import ctypes
    from wekws_amd import _capi
    name, B, T, F, left, right, skip = case
    assert int(GOLD["raises/" + name]) == 1
    x = torch.from_numpy(case_input(B, T, F)).cuda()
    with pytest.raises(IndexError):
        frontend.splice_skip(x, left, right, skip)
    lib = _capi.load()
    out = torch.empty(max(1, B * T * (left + right + 1) * F), device="cuda")
    rc = lib.wekws_hip_splice(x.data_ptr(), B, T, F, left, right, skip, out.data_ptr(), ctypes.c_void_p(0))
    assert rc == -1 and b"IndexError" in lib.wekws_hip_last_error()      # WEKWS_HIP_EINVAL


def test_fbank80_splice_fsmn_chain():
    """pcm -> HIP fbank(80) -> HIP splice(2,2)/skip 3 -> HIP FSMN-CTC, each stage's input being the previous HIP
    output, against the oracles fed the same way (fsmn_ctc.yaml recipe shape: 1 s of audio -> 32 x 400 -> 32 x 300)."""
    from oracle import fbank_oracle
    cfg = dict(synth.MODEL_CONFIGS["fsmn_ctc300"])
    sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
    model = init_model(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.cuda().eval()
    pcm = synth.synth_pcm(3, 16000, seed=4)
    feats = frontend.Fbank(num_bins=80)(torch.from_numpy(pcm).cuda())
    ref_feats = np.stack([fbank_oracle.fbank(p, 80) for p in pcm])
    assert np.abs(feats.cpu().numpy() - ref_feats).max() <= 1e-3
    x = frontend.splice_skip(feats, 2, 2, 3)
    assert x.shape == (3, 32, 400)
    y, cache = model(x)
    torch.cuda.synchronize()
    ry, rc = kws_oracle.forward(cfg, sd, splice_oracle.splice_skip(feats.cpu().numpy(), 2, 2, 3), None)
    assert np.abs(y.cpu().numpy() - ry).max() <= 1e-4 * max(1.0, float(np.abs(ry).max()))
    assert cache.shape == (3, 128, 11, 4)
    assert np.abs(cache.cpu().numpy() - rc).max() <= 1e-4 * max(1.0, float(np.abs(rc).max()))


@pytest.mark.parametrize("seed", [0, 1, 2, 260, 270, 1259])
def test_random_splice_shapes(seed):
    """Seeded fuzz: random batch / length / width / context / skip (init_dataset.py:24-68 takes any), bit-exact against the oracle.
    Seeds 260 / 270 / 1259: the first ones of tools/probe/fuzz_all.py's 1,000-seed run that drew an utterance no longer than its
    left (IndexError in the reference) or right context (its negative slice) -- the product clamped both until then."""
    rng = np.random.default_rng(300 + seed)
    for _ in range(25):
        B, T, F = int(rng.integers(1, 40)), int(rng.integers(1, 130)), int(rng.choice([1, 3, 23, 40, 79, 80, 120]))
        left, right, skip = int(rng.integers(0, 6)), int(rng.integers(0, 6)), int(rng.integers(1, 6))
        x = case_input(B, T, F, seed=int(rng.integers(0, 1000)))
        if left >= 1 and left >= T:                              # the reference's left-margin loop raises (init_dataset.py:45-48)
            with pytest.raises(IndexError):
                splice_oracle.splice_skip(x, left, right, skip)
            with pytest.raises(IndexError):
                frontend.splice_skip(torch.from_numpy(x).cuda(), left, right, skip)
            continue
        y = frontend.splice_skip(torch.from_numpy(x).cuda(), left, right, skip).cpu().numpy()
        want = splice_oracle.splice_skip(x, left, right, skip)
        assert y.shape == want.shape and np.array_equal(y, want), (seed, B, T, F, left, right, skip)
