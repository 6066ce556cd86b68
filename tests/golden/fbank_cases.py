"""Front-end parity cases shared by the golden generator (compiled reference) and the tests."""
from wekws_amd.utils import synth


def _c(name, kind, B=2, nsamp=16000, num_bins=40, seed=0, first_push=0, sample_rate=16000):
    return dict(name=name, kind=kind, B=B, nsamp=nsamp, num_bins=num_bins, seed=seed, first_push=first_push,
                sample_rate=sample_rate)


FBANK_CASES = [
    _c("noise_1s", "noise", B=3),
    _c("noise_1s_two_pushes", "noise", B=2, seed=1, first_push=5000),   # leftover rule, feature_pipeline.cc:41-44
    _c("sine_1s", "sine", B=1),
    _c("ramp_1s", "ramp", B=2),                                         # full int16 range, DC offset
    _c("silence_1s", "silence", B=1),                                   # FLT_EPSILON floor -> -15.942385
    _c("noise_80bins", "noise", B=2, num_bins=80, seed=2),
    _c("noise_exactly_one_frame", "noise", B=2, nsamp=400, seed=3),
    _c("noise_ragged_tail", "noise", B=2, nsamp=16000 + 97, seed=4),    # samples that do not fill a frame
    _c("noise_2p5s", "noise", B=1, nsamp=40000, seed=5),
    # 8 kHz audio: 200-sample frames, the reference transforms 256 points (fbank.h:43,117-119); 4 kHz: 100 samples, 128 points
    _c("noise_8k_1s", "noise", B=3, nsamp=8000, seed=6, sample_rate=8000),
    _c("ramp_8k_23bins", "ramp", B=2, nsamp=8000, num_bins=23, seed=7, sample_rate=8000),
    _c("noise_8k_two_pushes", "noise", B=2, nsamp=12000, seed=8, first_push=3000, sample_rate=8000),
    _c("noise_4k", "noise", B=2, nsamp=4000, num_bins=23, seed=9, sample_rate=4000),
]


def fbank_input(case):
    return synth.synth_pcm(case["B"], case["nsamp"], seed=case["seed"], kind=case["kind"])
