"""Deterministic synthetic weights / inputs shared by tests, bench.py and the
golden-fixture generator (there is no network for real checkpoints).

Every tensor is drawn from its own numpy PCG64 stream seeded by
``crc32(name) ^ seed`` so the values depend only on (name, shape, seed), not on
iteration order or on torch's RNG.  BatchNorm running statistics are
randomised (defaults of 0/1 would hide BN-folding bugs).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

# The model sections of the reference recipes (input_dim / output_dim are
# injected by the reference's train.py:134-153; here they are explicit).
MODEL_CONFIGS: Dict[str, dict] = {
    # examples/hi_xiaowen/s0/conf/ds_tcn.yaml:27-36  (headline ~300 k-param model)
    "ds_tcn_h256": dict(input_dim=40, output_dim=2, hidden_dim=256,
                        preprocessing=dict(type="linear"),
                        backbone=dict(type="tcn", ds=True, num_layers=4, kernel_size=8, dropout=0.1)),
    # examples/hey_snips/s0/conf/ds_tcn.yaml
    "ds_tcn_h64": dict(input_dim=40, output_dim=1, hidden_dim=64,
                       preprocessing=dict(type="linear"),
                       backbone=dict(type="tcn", ds=True, num_layers=4, kernel_size=8, dropout=0.1)),
    # examples/hi_xiaowen/s0/conf/tcn.yaml
    "tcn_h64": dict(input_dim=40, output_dim=2, hidden_dim=64,
                    preprocessing=dict(type="linear"),
                    backbone=dict(type="tcn", ds=False, num_layers=4, kernel_size=8, dropout=0.1)),
    # examples/hi_xiaowen/s0/conf/mdtc.yaml:28-38 (at 40-d)
    "mdtc_h64": dict(input_dim=40, output_dim=2, hidden_dim=64,
                     preprocessing=dict(type="linear"),
                     backbone=dict(type="mdtc", num_stack=4, stack_size=4, kernel_size=5,
                                   hidden_dim=64, causal=True)),
    # examples/hi_xiaowen/s0/conf/mdtc.yaml as trained: 80-d MFCC input (mdtc.yaml:11)
    "mdtc_h64_80d": dict(input_dim=80, output_dim=2, hidden_dim=64,
                         preprocessing=dict(type="linear"),
                         backbone=dict(type="mdtc", num_stack=4, stack_size=4, kernel_size=5,
                                       hidden_dim=64, causal=True)),
    # examples/hi_xiaowen/s0/conf/mdtc_small.yaml:28-38
    "mdtc_small": dict(input_dim=40, output_dim=2, hidden_dim=32,
                       preprocessing=dict(type="linear"),
                       backbone=dict(type="mdtc", num_stack=3, stack_size=4, kernel_size=5,
                                     hidden_dim=32, causal=True)),
    # examples/speechcommand_v1/s0/conf/mdtc.yaml:28-41 (at 40-d, 12 classes: BASELINE config 5)
    "mdtc_h64_global12": dict(input_dim=40, output_dim=12, hidden_dim=64,
                              preprocessing=dict(type="linear"),
                              backbone=dict(type="mdtc", num_stack=4, stack_size=4, kernel_size=5,
                                            hidden_dim=64, causal=True),
                              classifier=dict(type="global", dropout=0.5)),
    # BASELINE config 1: the ~38 k "tiny" model buildable from reference parts
    "mdtc_small_global12": dict(input_dim=40, output_dim=12, hidden_dim=32,
                                preprocessing=dict(type="linear"),
                                backbone=dict(type="mdtc", num_stack=3, stack_size=4, kernel_size=5,
                                              hidden_dim=32, causal=True),
                                classifier=dict(type="global", dropout=0.5)),
    "mdtc_small_last12": dict(input_dim=40, output_dim=12, hidden_dim=32,
                              preprocessing=dict(type="linear"),
                              backbone=dict(type="mdtc", num_stack=3, stack_size=4, kernel_size=5,
                                            hidden_dim=32, causal=True),
                              classifier=dict(type="last", dropout=0.5)),
    # examples/hi_xiaowen/s0/conf/gru.yaml:26-32
    "gru_2x128": dict(input_dim=40, output_dim=2, hidden_dim=128,
                      preprocessing=dict(type="linear"),
                      backbone=dict(type="gru", num_layers=2)),
    "gru_1x128": dict(input_dim=40, output_dim=2, hidden_dim=128,
                      preprocessing=dict(type="linear"),
                      backbone=dict(type="gru", num_layers=1)),
    # CTC-style head: identity activation + forward_softmax (kws_model.py:78-90, 204-210)
    "ds_tcn_h64_ctc20": dict(input_dim=40, output_dim=20, hidden_dim=64,
                             preprocessing=dict(type="linear"),
                             backbone=dict(type="tcn", ds=True, num_layers=4, kernel_size=8, dropout=0.1),
                             activation=dict(type="identity")),
    # examples/hi_xiaowen/s0/conf/ds_tcn_ctc.yaml:32-43: the headline body with a CTC token head (2599 tokens, 955 k
    # params) and identity activation; the 300-token twin keeps the committed fixtures small
    "ds_tcn_h256_ctc": dict(input_dim=40, output_dim=2599, hidden_dim=256,
                            preprocessing=dict(type="linear"),
                            backbone=dict(type="tcn", ds=True, num_layers=4, kernel_size=8, dropout=0.1),
                            activation=dict(type="identity")),
    "ds_tcn_h256_ctc300": dict(input_dim=40, output_dim=300, hidden_dim=256,
                               preprocessing=dict(type="linear"),
                               backbone=dict(type="tcn", ds=True, num_layers=4, kernel_size=8, dropout=0.1),
                               activation=dict(type="identity")),
    # examples/hi_xiaowen/s0/conf/fsmn_ctc.yaml:36-56 (400-d spliced fbank, 2599 CTC tokens, 756 k params)
    "fsmn_ctc": dict(input_dim=400, output_dim=2599, hidden_dim=128,
                     preprocessing=dict(type="none"),
                     backbone=dict(type="fsmn", input_affine_dim=140, num_layers=4, linear_dim=250, proj_dim=128,
                                   left_order=10, right_order=2, left_stride=1, right_stride=1,
                                   output_affine_dim=140),
                     classifier=dict(type="identity", dropout=0.1), activation=dict(type="identity")),
    # same body, 300 tokens: keeps the committed fixtures small
    "fsmn_ctc300": dict(input_dim=400, output_dim=300, hidden_dim=128,
                        preprocessing=dict(type="none"),
                        backbone=dict(type="fsmn", input_affine_dim=140, num_layers=4, linear_dim=250, proj_dim=128,
                                      left_order=10, right_order=2, left_stride=1, right_stride=1,
                                      output_affine_dim=140),
                        classifier=dict(type="identity", dropout=0.1), activation=dict(type="identity")),
    # odd sizes everywhere: exercises every zero-padding rule of the packed operands
    "fsmn_small": dict(input_dim=120, output_dim=11, hidden_dim=64,
                       preprocessing=dict(type="none"),
                       backbone=dict(type="fsmn", input_affine_dim=72, num_layers=2, linear_dim=100, proj_dim=40,
                                     left_order=5, right_order=1, left_stride=1, right_stride=1,
                                     output_affine_dim=56),
                       classifier=dict(type="identity", dropout=0.1), activation=dict(type="identity")),
}


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(name.encode()) & 0xFFFFFFFF, seed & 0xFFFFFFFF])


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    """One deterministic float32 tensor for state_dict entry ``name``."""
    g = _rng(name, seed)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "running_mean":
        return (g.standard_normal(shape) * 0.5).astype(np.float32)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if name == "global_cmvn.mean":
        return (10.0 + g.standard_normal(shape)).astype(np.float32)
    if name == "global_cmvn.istd":
        return g.uniform(0.2, 0.5, shape).astype(np.float32)
    if leaf.startswith("weight_ih") or leaf.startswith("weight_hh") or leaf.startswith("bias_ih") \
            or leaf.startswith("bias_hh"):  # nn.GRU: U(-1/sqrt(H), 1/sqrt(H)), H = shape[0] / 3
        b = 1.0 / np.sqrt(shape[0] / 3.0)
        return g.uniform(-b, b, shape).astype(np.float32)
    if name == "classifier.linear.weight":
        # per-frame head: smaller than torch's default so the sigmoid is exercised around its
        # sensitive mid-range instead of saturating (MDTC sums 4 non-negative stack outputs)
        b = 0.3 / np.sqrt(shape[1])
        return g.uniform(-b, b, shape).astype(np.float32)
    if name == "classifier.linear.bias":
        return (g.standard_normal(shape) * 1.0).astype(np.float32)
    if leaf == "weight" and len(shape) == 1:  # BatchNorm gamma
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "weight":  # Linear / Conv1d: U(-1/sqrt(fan_in), +)
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        return g.uniform(-b, b, shape).astype(np.float32)
    if leaf == "bias":
        return (g.standard_normal(shape) * 0.1).astype(np.float32)
    raise ValueError(f"no synthesis rule for state_dict entry {name!r}")


def synth_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 1234) -> Dict[str, np.ndarray]:
    """``spec`` = iterable of (name, shape) in state_dict order."""
    return {name: synth_tensor(name, shape, seed) for name, shape in spec}


def module_spec(module) -> list:
    """(name, shape) list of a torch module's state_dict."""
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


def synth_feats(B: int, T: int, idim: int = 40, seed: int = 0, cmvn_like: bool = False) -> np.ndarray:
    """Synthetic fbank-shaped features: randn, or 3*randn+10 (log-mel-like, for CMVN cases)."""
    g = np.random.default_rng([0xFEA75, seed])
    x = g.standard_normal((B, T, idim)).astype(np.float32)
    if cmvn_like:
        x = (3.0 * x + 10.0).astype(np.float32)
    return x


def synth_pcm(B: int, nsamp: int = 16000, seed: int = 0, kind: str = "noise") -> np.ndarray:
    """Synthetic audio in int16 scale as float32 (the runtime never divides by 32768:
    runtime/core/frontend/feature_pipeline.cc:49-55)."""
    g = np.random.default_rng([0x9C3, seed])
    if kind == "noise":
        x = np.clip(np.round(g.standard_normal((B, nsamp)) * 3000.0), -32767, 32767)
    elif kind == "sine":
        t = np.arange(nsamp, dtype=np.float64) / 16000.0
        x = np.round(1000.0 * np.sin(2 * np.pi * 440.0 * t))[None, :].repeat(B, 0)
    elif kind == "ramp":
        x = ((np.arange(nsamp)[None, :] * 7 + np.arange(B)[:, None] * 131) % 65536 - 32768).astype(np.float64)
    elif kind == "silence":
        x = np.zeros((B, nsamp))
    else:
        raise ValueError(kind)
    return x.astype(np.float32)


def checksum(sd: Dict[str, np.ndarray]) -> float:
    """Order-independent fingerprint of a synthetic state_dict (guards the fixtures
    against an RNG-stream change)."""
    return float(sum(float(np.abs(v.astype(np.float64)).sum()) for v in sd.values()))
