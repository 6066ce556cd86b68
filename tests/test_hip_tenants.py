"""GPU: every kernel family returns the same bits whether it has the device to itself or shares its CUs with an MFMA-heavy tenant
on another stream.

Why: the gfx950 hazard of wekws_amd/csrc/pk_safe.hip.h showed that an instruction's result can depend on what OTHER waves of the
SIMD are issuing.  The parity tests run one forward at a time; this one puts an MDTC forward that half-fills the CUs on a second
stream (its matrix phases are the neighbours that hazard needs) and compares each model's posteriors and returned cache, one-shot
and as a stream of chunks, with the solo run -- bit for bit, 12 overlapping rounds each."""
import numpy as np
import pytest
import torch

from tests.test_hip_parity import build
from wekws_amd import pack
from wekws_amd.utils import synth

pytestmark = pytest.mark.gpu

MODELS = ["ds_tcn_h64", "ds_tcn_h256", "tcn_h64", "mdtc_h64", "mdtc_small", "mdtc_small_global12", "gru_2x128", "fsmn_ctc", "ds_tcn_h256_ctc"]


def _forward(model, x, chunks):
    if chunks is None:
        y, c = model(x)
        return y, c
    ys, c, t = [], None, 0
    for n in chunks:
        y, c = model(x[:, t:t + n]) if c is None else model(x[:, t:t + n], c)
        ys.append(y)
        t += n
    return torch.cat(ys, dim=1), c


TENANTS = {"mdtc_h64": 512, "gru_2x128": 256, "ds_tcn_h64": 1024}   # model -> batch that leaves room on every CU beside it


@pytest.fixture(scope="module", params=sorted(TENANTS))
def tenant(request):
    """An MFMA-heavy forward for the second stream: MDTC h64 (four-wave workgroups, matrix and vector phases interleaved across the
    workgroups of a CU), the GRU layer wavefront (stage workgroups polling each other's rings between MFMA steps), DS-TCN h64."""
    name = request.param
    cfg = dict(synth.MODEL_CONFIGS[name])
    m = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 99))
    B = TENANTS[name]
    x = torch.from_numpy(synth.synth_feats(B, 98, cfg["input_dim"], seed=8)).cuda()
    if cfg["backbone"]["type"] == "gru":
        h0 = torch.zeros(cfg["backbone"]["num_layers"], B, cfg["hidden_dim"], device="cuda")
        return (lambda xx: m(xx, h0)), x
    return m, x


@pytest.mark.parametrize("precision", ["default", "f32"])
@pytest.mark.parametrize("name", MODELS)
def test_bit_identical_beside_an_mfma_tenant(name, precision, tenant):
    cfg = dict(synth.MODEL_CONFIGS[name])
    model = build(cfg, synth.synth_state_dict(pack.model_spec(cfg), 21)).set_precision(precision)
    pooled = "global" in name or "last" in name
    gru = cfg["backbone"]["type"] == "gru"
    B, T = 192, 98
    x = torch.from_numpy(synth.synth_feats(B, T, cfg["input_dim"], seed=3)).cuda()
    if gru:
        h0 = torch.zeros(cfg["backbone"]["num_layers"], B, cfg["hidden_dim"], device="cuda")
        fwd = lambda xx, c=None: model(xx, h0 if c is None else c)      # noqa: E731
    else:
        fwd = model
    modes = [None] if pooled else [None, [40, 10, 48], [7, 91]]
    solo = []
    for ch in modes:
        y, c = _forward(fwd, x, ch)
        solo.append((y.clone(), c.clone()))
    tm, tx = tenant
    torch.cuda.synchronize()
    s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
    for rnd in range(12):
        with torch.cuda.stream(s_b):
            for _ in range(5):
                tm(tx)
        with torch.cuda.stream(s_a):
            got = [_forward(fwd, x, ch) for ch in modes]
        torch.cuda.synchronize()
        for (y, c), (ys, cs), ch in zip(got, solo, modes):
            assert torch.equal(y.view(torch.int32), ys.view(torch.int32)), (name, precision, ch, rnd, float((y - ys).abs().max()))
            assert torch.equal(c.view(torch.int32), cs.view(torch.int32)), (name, precision, ch, rnd)


def test_front_end_and_post_processing_beside_an_mfma_tenant(tenant):
    """The small kernels either side of the model -- fbank (Hamming / Povey, 40 / 80 bins, float / int16 PCM), MFCC (DCT + lifter),
    splice + skip, softmax + top-k, the DET reductions -- beside the same tenant: bit-identical to their solo runs."""
    from wekws_amd import ctc, det
    from wekws_amd.frontend import Fbank, Mfcc, splice_skip
    pcm = torch.from_numpy(synth.synth_pcm(384, 16000, seed=12, kind="noise")).cuda()
    pcm16 = pcm.round().clamp(-32768, 32767).to(torch.int16)
    logits = torch.from_numpy(np.random.default_rng(5).standard_normal((64, 50, 2599)).astype(np.float32) * 3).cuda()
    scores = torch.rand(256, 98, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    fb40, fb80, fbp, mf = Fbank(40), Fbank(80), Fbank(80, window="povey"), Mfcc(80, 80)

    def work():
        f80 = fb80(pcm)
        p, i = ctc.softmax_topk(logits, 3)
        mx, am = det.max_pool_scores(scores)
        return [fb40(pcm), f80, fb40(pcm16), fbp(pcm), mf(pcm), splice_skip(f80, 2, 2, 3), p, i, mx, am]

    solo = [t.clone() for t in work()]
    tm, tx = tenant
    torch.cuda.synchronize()
    s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
    for rnd in range(12):
        with torch.cuda.stream(s_b):
            for _ in range(8):
                tm(tx)
        with torch.cuda.stream(s_a):
            got = work()
        torch.cuda.synchronize()
        for k, (g, r) in enumerate(zip(got, solo)):
            assert torch.equal(g, r) or torch.equal(g.view(torch.int32), r.view(torch.int32)), (rnd, k)
