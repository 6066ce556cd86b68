"""Host-side weight packer: reference ``state_dict`` + ``configs['model']`` -> (descriptor, float32 blob)
in the layout ``include/wekws_hip.h`` documents.

Everything that is constant at inference time is folded here, in float64, and rounded to float32 once:
  * GlobalCMVN ``(x - mean) * istd``  (wekws/model/cmvn.py:45-48) into the first Linear,
  * every eval-mode ``BatchNorm1d``  ``(x - rm) / sqrt(rv + eps) * w + b``  (wekws/model/tcn.py:81,108,111;
    wekws/model/mdtc.py:47,86,92) into the convolution in front of it,
  * ``Dropout`` is the identity in eval mode and disappears.
The state_dict key names are the reference's; ``model_spec`` reproduces them from
the config alone so that ``wekws_amd.model.kws_model.init_model`` can build a ``load_state_dict``-compatible
module without importing the reference.
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Tuple

import numpy as np

ABI_VERSION = 2          # include/wekws_hip.h; the descriptor / blob layout is unchanged since 1 (load_packed reads both)
BACKBONE = dict(ds_tcn=0, tcn=1, mdtc=2, gru=3, fsmn=4)
HEAD = dict(linear=0, glob=1, last=2, identity=3)
BN_EPS = 1e-5
HEAD_HIDDEN = 64  # nn.Linear(hidden_dim, 64) in wekws/model/kws_model.py:181-186

DESC_FIELDS = ("abi_version", "backbone", "idim", "hdim", "odim", "num_layers", "num_stack", "stack_size",
               "kernel_size", "preproc_relu", "head", "head_hidden", "activation", "precision", "aux0", "aux1")
# FSMN reuses the generic slots (include/wekws_hip.h): hdim = linear_dim, num_stack = proj_dim, kernel_size =
# left_order, stack_size = right_order, aux0 = input_affine_dim, aux1 = output_affine_dim
PRECISION = dict(default=0, f32=1, f16x3=2, f16=3)  # enum wekws_hip_precision
ACT_SOFTMAX = 2  # enum wekws_hip_activation: the model is forward_softmax (exported CTC graphs, export_onnx.py:46-48)


class ConfigError(ValueError):
    pass


def parse_config(configs: Mapping) -> dict:
    """configs['model'] of the reference (wekws/model/kws_model.py:97-214) -> flat description."""
    idim, odim, hdim = int(configs["input_dim"]), int(configs["output_dim"]), int(configs["hidden_dim"])
    prep = configs["preprocessing"]["type"]
    if prep not in ("linear", "none"):
        # cnn1d_s1 exists in the reference but cannot run
        raise ConfigError(f"Unknown preprocessing type {prep}")
    bb = configs["backbone"]
    bt = bb["type"]
    d = dict(abi_version=ABI_VERSION, idim=idim, hdim=hdim, odim=odim, num_layers=0, num_stack=0, stack_size=0,
             kernel_size=0, preproc_relu=1 if prep == "linear" else 0, prep=prep, aux0=0, aux1=0,
             precision=PRECISION[str(configs.get("_precision", "default"))])
    if bt == "gru":
        d.update(backbone=BACKBONE["gru"], num_layers=int(bb["num_layers"]), kind="gru")
    elif bt == "tcn":
        ds = bool(bb.get("ds", False))
        d.update(backbone=BACKBONE["ds_tcn" if ds else "tcn"], num_layers=int(bb["num_layers"]),
                 kernel_size=int(bb.get("kernel_size", 8)), kind="ds_tcn" if ds else "tcn")
    elif bt == "mdtc":
        if not bb.get("causal", True):
            raise ConfigError("mdtc: only causal=True is supported (all reference recipes use it)")
        if int(bb["hidden_dim"]) != hdim:
            raise ConfigError("mdtc: backbone.hidden_dim must equal hidden_dim")
        d.update(backbone=BACKBONE["mdtc"], num_stack=int(bb["num_stack"]), stack_size=int(bb["stack_size"]),
                 kernel_size=int(bb["kernel_size"]), kind="mdtc")
    elif bt == "fsmn":
        # the reference builds every FSMNBlock with stride 1 (fsmn.py:381-383 drops left_stride / right_stride),
        # and a block with right_order 0 cannot run there (empty slice at fsmn.py:231); follow both
        if prep != "none":
            raise ConfigError("fsmn: only preprocessing none is supported (the reference recipe, fsmn_ctc.yaml:38-39)")
        if int(bb["right_order"]) < 1 or int(bb["left_order"]) < 1:
            raise ConfigError("fsmn: left_order and right_order must be >= 1")
        hdim = int(bb["linear_dim"])
        d.update(backbone=BACKBONE["fsmn"], kind="fsmn", hdim=hdim, num_layers=int(bb["num_layers"]),
                 num_stack=int(bb["proj_dim"]), kernel_size=int(bb["left_order"]), stack_size=int(bb["right_order"]),
                 aux0=int(bb["input_affine_dim"]), aux1=int(bb["output_affine_dim"]))
    else:
        raise ConfigError(f"Unknown body type {bt}")
    if d["kind"] == "fsmn":
        # FSMN ends in its own out_linear2 (-> output_dim): only the identity classifier is shape-consistent
        if "classifier" not in configs or configs["classifier"]["type"] != "identity":
            raise ConfigError("fsmn: classifier must be identity (the backbone already maps to output_dim)")
        d.update(head=HEAD["identity"], activation=0, head_hidden=0)
        if "activation" in configs and configs["activation"]["type"] != "identity":
            raise ConfigError(f"Unknown activation type {configs['activation']['type']}")
        cm = configs.get("cmvn", {}) or {}
        d["cmvn"] = bool(configs.get("_cmvn")) or bool(cm.get("cmvn_file"))
        d["norm_var"] = bool(cm.get("norm_var", True))
        _exported_softmax(configs, d)
        return d
    if prep == "none" and idim != hdim:
        raise ConfigError("preprocessing none needs input_dim == hidden_dim")
    if "classifier" in configs:
        ct = configs["classifier"]["type"]
        if ct not in ("global", "last", "identity"):
            raise ConfigError(f"Unknown classifier type {ct}")
        d.update(head=HEAD["glob" if ct == "global" else ct], activation=0,
                 head_hidden=HEAD_HIDDEN if ct in ("global", "last") else 0)
        if ct == "identity" and odim != hdim:
            raise ConfigError("identity classifier needs output_dim == hidden_dim")
    else:
        d.update(head=HEAD["linear"], activation=1, head_hidden=0)
    if "activation" in configs:
        if configs["activation"]["type"] != "identity":
            raise ConfigError(f"Unknown activation type {configs['activation']['type']}")
        d["activation"] = 0
    cm = configs.get("cmvn", {}) or {}
    d["cmvn"] = bool(configs.get("_cmvn")) or bool(cm.get("cmvn_file"))
    d["norm_var"] = bool(cm.get("norm_var", True))
    _exported_softmax(configs, d)
    return d


def _exported_softmax(configs: Mapping, d: dict) -> None:
    """configs['_exported_softmax'] (set by the exported-file reader, wekws_amd/utils/onnx_lower.py): forward itself
    ends in the softmax, as in the graph wekws/bin/export_onnx.py:46-48 traces for CTC recipes."""
    if configs.get("_exported_softmax"):
        if d["activation"] != 0 or d["head"] not in (HEAD["linear"], HEAD["identity"]):
            raise ConfigError("_exported_softmax needs a per-frame head with identity activation")
        d["activation"] = ACT_SOFTMAX


def mdtc_blocks(d: dict) -> List[Tuple[str, int]]:
    """(state_dict prefix, dilation) of every TCNBlock in execution order (mdtc.py:247-275)."""
    out = [("backbone.preprocessor.", 1)]
    for s in range(d["num_stack"]):
        for j in range(d["stack_size"]):
            out.append((f"backbone.blocks.{s}.res_blocks.{j}.", 2 ** j))
    return out


def cache_shape(d: dict, B: int) -> Tuple[int, ...]:
    if d["kind"] == "gru":
        return (d["num_layers"], B, d["hdim"])
    if d["kind"] == "fsmn":  # (B, proj_dim, lorder - 1 + rorder, layers)  -- export_onnx.py:55-60, fsmn.py:495
        return (B, d["num_stack"], d["kernel_size"] - 1 + d["stack_size"], d["num_layers"])
    k = d["kernel_size"]
    if d["kind"] == "mdtc":
        P = sum((k - 1) * dil for _, dil in mdtc_blocks(d))
    else:
        P = sum((k - 1) * 2 ** i for i in range(d["num_layers"]))
    return (B, d["hdim"], P)


def _bn(prefix: str, C: int) -> List[Tuple[str, tuple]]:
    return [(prefix + ".weight", (C,)), (prefix + ".bias", (C,)), (prefix + ".running_mean", (C,)),
            (prefix + ".running_var", (C,)), (prefix + ".num_batches_tracked", ())]


def model_spec(configs: Mapping) -> List[Tuple[str, tuple]]:
    """(name, shape) of every state_dict entry of the reference model for this config, in order."""
    d = parse_config(configs)
    C, I, K, ks = d["hdim"], d["idim"], d["odim"], d["kernel_size"]
    spec: List[Tuple[str, tuple]] = []
    if d["cmvn"]:
        spec += [("global_cmvn.mean", (I,)), ("global_cmvn.istd", (I,))]
    if d["prep"] == "linear":
        spec += [("preprocessing.out.0.weight", (C, I)), ("preprocessing.out.0.bias", (C,))]
    if d["kind"] == "fsmn":
        A1, A2, D, lo, ro = d["aux0"], d["aux1"], d["num_stack"], d["kernel_size"], d["stack_size"]
        spec += [("backbone.in_linear1.linear.weight", (A1, I)), ("backbone.in_linear1.linear.bias", (A1,)),
                 ("backbone.in_linear2.linear.weight", (C, A1)), ("backbone.in_linear2.linear.bias", (C,))]
        for l in range(d["num_layers"]):
            p = f"backbone.fsmn.{l}."
            spec += [(p + "0.linear.weight", (D, C)), (p + "1.conv_left.weight", (D, 1, lo, 1)),
                     (p + "1.conv_right.weight", (D, 1, ro, 1)),
                     (p + "2.linear.weight", (C, D)), (p + "2.linear.bias", (C,))]
        spec += [("backbone.out_linear1.linear.weight", (A2, C)), ("backbone.out_linear1.linear.bias", (A2,)),
                 ("backbone.out_linear2.linear.weight", (K, A2)), ("backbone.out_linear2.linear.bias", (K,))]
        return spec
    if d["kind"] == "gru":
        for l in range(d["num_layers"]):
            spec += [(f"backbone.weight_ih_l{l}", (3 * C, C)), (f"backbone.weight_hh_l{l}", (3 * C, C)),
                     (f"backbone.bias_ih_l{l}", (3 * C,)), (f"backbone.bias_hh_l{l}", (3 * C,))]
    elif d["kind"] == "ds_tcn":
        for i in range(d["num_layers"]):
            p = f"backbone.network.{i}.cnn."
            spec += [(p + "0.weight", (C, 1, ks)), (p + "0.bias", (C,))] + _bn(p + "1", C)
            spec += [(p + "3.weight", (C, C, 1)), (p + "3.bias", (C,))] + _bn(p + "4", C)
    elif d["kind"] == "tcn":
        for i in range(d["num_layers"]):
            p = f"backbone.network.{i}.cnn."
            spec += [(p + "0.weight", (C, C, ks)), (p + "0.bias", (C,))] + _bn(p + "1", C)
    else:
        for p, _ in mdtc_blocks(d):
            spec += [(p + "conv1.conv.weight", (C, 1, ks)), (p + "conv1.conv.bias", (C,))] + _bn(p + "conv1.bn", C)
            spec += [(p + "conv1.pointwise.weight", (C, C, 1)), (p + "conv1.pointwise.bias", (C,))]
            spec += _bn(p + "bn1", C)
            spec += [(p + "conv2.weight", (C, C, 1)), (p + "conv2.bias", (C,))] + _bn(p + "bn2", C)
    if d["head"] == HEAD["linear"]:
        spec += [("classifier.linear.weight", (K, C)), ("classifier.linear.bias", (K,))]
    elif d["head"] in (HEAD["glob"], HEAD["last"]):
        spec += [("classifier.classifier.0.weight", (HEAD_HIDDEN, C)), ("classifier.classifier.0.bias", (HEAD_HIDDEN,)),
                 ("classifier.classifier.3.weight", (K, HEAD_HIDDEN)), ("classifier.classifier.3.bias", (K,))]
    return spec


def _f64(sd, name):
    v = sd[name]
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float64)


def _bn_affine(sd, prefix):
    """eval BatchNorm1d as y = s * x + t."""
    s = _f64(sd, prefix + ".weight") / np.sqrt(_f64(sd, prefix + ".running_var") + BN_EPS)
    t = _f64(sd, prefix + ".bias") - _f64(sd, prefix + ".running_mean") * s
    return s, t


def pack(configs: Mapping, sd: Mapping) -> Tuple[dict, np.ndarray]:
    """-> (descriptor dict with DESC_FIELDS, float32 blob)."""
    d = parse_config(configs)
    C, I = d["hdim"], d["idim"]
    parts: List[np.ndarray] = []

    if d["kind"] == "fsmn":
        return _pack_fsmn(d, sd)

    # ---- preprocessing (+ CMVN):  W (x - mean) * istd + b = (W * istd) x + (b - (W * istd) mean)
    if d["prep"] == "linear":
        W, b = _f64(sd, "preprocessing.out.0.weight"), _f64(sd, "preprocessing.out.0.bias")
    else:
        W, b = np.eye(C, I), np.zeros(C)
    if d["cmvn"]:
        mean = _f64(sd, "global_cmvn.mean")
        istd = _f64(sd, "global_cmvn.istd") if d["norm_var"] else np.ones(I)
        W = W * istd[None, :]
        b = b - W @ mean
    parts += [W, b]

    if d["kind"] == "gru":
        for l in range(d["num_layers"]):
            parts += [_f64(sd, f"backbone.weight_ih_l{l}"), _f64(sd, f"backbone.weight_hh_l{l}"),
                      _f64(sd, f"backbone.bias_ih_l{l}"), _f64(sd, f"backbone.bias_hh_l{l}")]
    elif d["kind"] == "ds_tcn":
        for i in range(d["num_layers"]):
            p = f"backbone.network.{i}.cnn."
            s1, t1 = _bn_affine(sd, p + "1")
            s2, t2 = _bn_affine(sd, p + "4")
            wd, bd = _f64(sd, p + "0.weight")[:, 0, :], _f64(sd, p + "0.bias")
            wp, bp = _f64(sd, p + "3.weight")[:, :, 0], _f64(sd, p + "3.bias")
            parts += [wd * s1[:, None], bd * s1 + t1, wp * s2[:, None], bp * s2 + t2]
    elif d["kind"] == "tcn":
        for i in range(d["num_layers"]):
            p = f"backbone.network.{i}.cnn."
            s1, t1 = _bn_affine(sd, p + "1")
            parts += [_f64(sd, p + "0.weight") * s1[:, None, None], _f64(sd, p + "0.bias") * s1 + t1]
    else:
        for p, _ in mdtc_blocks(d):
            sa, ta = _bn_affine(sd, p + "conv1.bn")
            s1, t1 = _bn_affine(sd, p + "bn1")
            s2, t2 = _bn_affine(sd, p + "bn2")
            wd, bd = _f64(sd, p + "conv1.conv.weight")[:, 0, :], _f64(sd, p + "conv1.conv.bias")
            w1, b1 = _f64(sd, p + "conv1.pointwise.weight")[:, :, 0], _f64(sd, p + "conv1.pointwise.bias")
            w2, b2 = _f64(sd, p + "conv2.weight")[:, :, 0], _f64(sd, p + "conv2.bias")
            parts += [wd * sa[:, None], bd * sa + ta, w1 * s1[:, None], b1 * s1 + t1, w2 * s2[:, None], b2 * s2 + t2]

    if d["head"] == HEAD["linear"]:
        parts += [_f64(sd, "classifier.linear.weight"), _f64(sd, "classifier.linear.bias")]
    elif d["head"] in (HEAD["glob"], HEAD["last"]):
        parts += [_f64(sd, "classifier.classifier.0.weight"), _f64(sd, "classifier.classifier.0.bias"),
                  _f64(sd, "classifier.classifier.3.weight"), _f64(sd, "classifier.classifier.3.bias")]
    blob = np.concatenate([np.ravel(p) for p in parts]).astype(np.float32)
    desc = {k: int(d[k]) for k in DESC_FIELDS}
    return desc, np.ascontiguousarray(blob)


def _pack_fsmn(d: dict, sd: Mapping) -> Tuple[dict, np.ndarray]:
    """FSMN blob: in1 W(A1,I) b | in2 W(C,A1) b | per layer: Wp(D,C), taps(D, lo+ro), Wa(C,D), ba | out1 W(A2,C) b |
    out2 W(K,A2) b.  CMVN folds into in_linear1; the memory block's identity path (fsmn.py:236-237) folds into its
    taps: out[t] = sum_j taps[j] x_pad[t + j] with taps = [wl_0 .. wl_{lo-1} + 1 | wr_0 .. wr_{ro-1}]."""
    I, lo, ro = d["idim"], d["kernel_size"], d["stack_size"]
    W, b = _f64(sd, "backbone.in_linear1.linear.weight"), _f64(sd, "backbone.in_linear1.linear.bias")
    if d["cmvn"]:
        mean = _f64(sd, "global_cmvn.mean")
        istd = _f64(sd, "global_cmvn.istd") if d["norm_var"] else np.ones(I)
        W = W * istd[None, :]
        b = b - W @ mean
    parts = [W, b, _f64(sd, "backbone.in_linear2.linear.weight"), _f64(sd, "backbone.in_linear2.linear.bias")]
    for l in range(d["num_layers"]):
        p = f"backbone.fsmn.{l}."
        taps = np.concatenate([_f64(sd, p + "1.conv_left.weight")[:, 0, :, 0],
                               _f64(sd, p + "1.conv_right.weight")[:, 0, :, 0]], axis=1)
        taps[:, lo - 1] += 1.0
        parts += [_f64(sd, p + "0.linear.weight"), taps, _f64(sd, p + "2.linear.weight"), _f64(sd, p + "2.linear.bias")]
    parts += [_f64(sd, "backbone.out_linear1.linear.weight"), _f64(sd, "backbone.out_linear1.linear.bias"),
              _f64(sd, "backbone.out_linear2.linear.weight"), _f64(sd, "backbone.out_linear2.linear.bias")]
    blob = np.concatenate([np.ravel(p) for p in parts]).astype(np.float32)
    return {k: int(d[k]) for k in DESC_FIELDS}, np.ascontiguousarray(blob)


def blob_elems(desc: Mapping) -> int:
    """Python twin of wekws_hip_blob_elems (checked against the library in tests/test_capi.py)."""
    C, I, K, ks = desc["hdim"], desc["idim"], desc["odim"], desc["kernel_size"]
    bb = desc["backbone"]
    if bb == BACKBONE["fsmn"]:
        A1, A2, D, ro = desc["aux0"], desc["aux1"], desc["num_stack"], desc["stack_size"]
        return (A1 * I + A1 + C * A1 + C + desc["num_layers"] * (D * C + D * (ks + ro) + C * D + C)
                + A2 * C + A2 + K * A2 + K)
    n = C * I + C
    if bb == BACKBONE["ds_tcn"]:
        n += desc["num_layers"] * (C * ks + C + C * C + C)
    elif bb == BACKBONE["tcn"]:
        n += desc["num_layers"] * (C * C * ks + C)
    elif bb == BACKBONE["mdtc"]:
        n += (1 + desc["num_stack"] * desc["stack_size"]) * (C * ks + C + 2 * (C * C + C))
    else:
        n += desc["num_layers"] * (6 * C * C + 6 * C)
    if desc["head"] == HEAD["linear"]:
        n += K * C + K
    elif desc["head"] in (HEAD["glob"], HEAD["last"]):
        hh = desc["head_hidden"]
        n += hh * C + hh + K * hh + K
    return n


# ------------------------------------------------------------------------------------------------
# Packed-model file: what a C / C++ host reads instead of a torch checkpoint.
#   bytes 0..7    magic b"WEKWSHIP"
#   then          16 x int32 little-endian = struct wekws_hip_desc
#   then          uint64 n_elems, n_elems x float32 little-endian = the folded blob
# ------------------------------------------------------------------------------------------------
MAGIC = b"WEKWSHIP"


def save_packed(path: str, desc: Mapping, blob: np.ndarray) -> None:
    fields = [int(desc.get(k, 0)) for k in DESC_FIELDS]
    blob = np.ascontiguousarray(blob, dtype="<f4")
    if blob.size != blob_elems(desc):
        raise ValueError(f"blob has {blob.size} floats, descriptor needs {blob_elems(desc)}")
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(np.asarray(fields, dtype="<i4").tobytes())
        f.write(np.asarray([blob.size], dtype="<u8").tobytes())
        f.write(blob.tobytes())


def load_packed(path: str) -> Tuple[dict, np.ndarray]:
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a packed wekws_hip model")
        fields = np.frombuffer(f.read(64), dtype="<i4")
        n = int(np.frombuffer(f.read(8), dtype="<u8")[0])
        blob = np.frombuffer(f.read(4 * n), dtype="<f4").astype(np.float32)
    if blob.size != n or fields.size != 16:
        raise ValueError(f"{path}: truncated")
    desc = {k: int(v) for k, v in zip(DESC_FIELDS, fields)}
    if desc["abi_version"] not in (1, ABI_VERSION) or blob.size != blob_elems(desc):
        raise ValueError(f"{path}: ABI version / size mismatch")
    desc["abi_version"] = ABI_VERSION                           # (files written under version 1 carry the same layout)
    return desc, blob
