"""CPU oracle for the WeKws model forward  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a numpy (float32) restatement of the reference's inference
forward.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path
(``wekws_amd``) never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` runs the *live*
reference (``/root/reference/wekws/model``, PyTorch CPU fp32) on seeded
weights/inputs and commits its outputs under ``tests/golden/*.npz``;
``tests/test_oracle.py`` checks this restatement against every one of those
fixtures (the reference itself ships no golden vectors).

Each function cites the reference file:line it restates (paths relative to
/root/reference).  Weights come in as a ``state_dict``-shaped mapping
``name -> ndarray`` with exactly the reference's key names.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
BN_EPS = F32(1e-5)  # nn.BatchNorm1d default eps (wekws/model/tcn.py:81,108,111; mdtc.py:47,86,92)


# --------------------------------------------------------------------------- #
# small pieces
# --------------------------------------------------------------------------- #
def _f(a):
    return np.ascontiguousarray(a, dtype=F32)


def batchnorm_eval(x, sd, prefix):
    """nn.BatchNorm1d in eval mode on (B, C, T): per-channel affine with the
    running statistics.  wekws/model/tcn.py:81,108,111 ; mdtc.py:47,86,92."""
    w = _f(sd[prefix + ".weight"])[None, :, None]
    b = _f(sd[prefix + ".bias"])[None, :, None]
    m = _f(sd[prefix + ".running_mean"])[None, :, None]
    v = _f(sd[prefix + ".running_var"])[None, :, None]
    return (x - m) / np.sqrt(v + BN_EPS) * w + b


def causal_concat(x, cache, pad):
    """Left context for a causal conv.  Empty cache == zero left padding.
    wekws/model/tcn.py:49-54 ; wekws/model/mdtc.py:108-112.
    x: (B, C, T); cache: (B, C, pad) or None.  Returns (u, new_cache)."""
    B, C, T = x.shape
    if cache is None or cache.size == 0:
        u = np.concatenate([np.zeros((B, C, pad), F32), x], axis=2)
    else:
        assert cache.shape == (B, C, pad), (cache.shape, (B, C, pad))
        u = np.concatenate([_f(cache), x], axis=2)
    return u, u[:, :, u.shape[2] - pad:].copy()


def depthwise_conv(u, w, b, dilation):
    """nn.Conv1d(C, C, k, dilation=d, groups=C) (cross-correlation, no padding).
    wekws/model/tcn.py:102-107 ; wekws/model/mdtc.py:35-44.
    u: (B, C, pad+T), w: (C, 1, k), b: (C,)."""
    k = w.shape[2]
    T = u.shape[2] - (k - 1) * dilation
    acc = np.zeros((u.shape[0], u.shape[1], T), F32)
    for j in range(k):
        acc += _f(w[:, 0, j])[None, :, None] * u[:, :, j * dilation:j * dilation + T]
    return acc + _f(b)[None, :, None]


def full_conv(u, w, b, dilation):
    """nn.Conv1d(C, C, k, dilation=d) (dense).  wekws/model/tcn.py:76-80."""
    k = w.shape[2]
    T = u.shape[2] - (k - 1) * dilation
    acc = np.zeros((u.shape[0], w.shape[0], T), F32)
    for j in range(k):
        acc += np.einsum("oc,bct->bot", _f(w[:, :, j]), u[:, :, j * dilation:j * dilation + T],
                         optimize=True).astype(F32)
    return acc + _f(b)[None, :, None]


def pointwise_conv(x, w, b):
    """nn.Conv1d(C, C_out, 1).  wekws/model/tcn.py:110 ; mdtc.py:48-53,82-84."""
    B, C, T = x.shape
    y = np.matmul(_f(w[:, :, 0])[None], x)  # (1,O,C) @ (B,C,T) -> (B,O,T)
    return y.astype(F32) + _f(b)[None, :, None]


def relu(x):
    return np.maximum(x, F32(0))


def sigmoid(x):
    return (F32(1) / (F32(1) + np.exp(-x.astype(F32)))).astype(F32)


def linear(x, w, b):
    """torch.nn.Linear on the last axis."""
    return (np.matmul(x, _f(w).T) + _f(b)).astype(F32)


# --------------------------------------------------------------------------- #
# backbones
# --------------------------------------------------------------------------- #
def tcn_forward(cfg, sd, x, in_cache):
    """TCN.forward + Block.forward + {Ds,}CnnBlock.cnn.
    wekws/model/tcn.py:139-166 (stack), :35-61 (block), :75-84 / :101-114 (cnn).
    x: (B, T, C) -> (B, T, C); cache (B, C, sum(pad_i))."""
    bb = cfg["backbone"]
    L = bb["num_layers"]
    k = bb.get("kernel_size", 8)
    ds = bb.get("ds", False)
    h = np.transpose(x, (0, 2, 1))
    caches, off = [], 0
    for i in range(L):
        d = 2 ** i
        pad = (k - 1) * d
        c_in = None if in_cache is None or in_cache.size == 0 else in_cache[:, :, off:off + pad]
        u, c_out = causal_concat(h, c_in, pad)
        p = f"backbone.network.{i}.cnn."
        if ds:
            a = depthwise_conv(u, sd[p + "0.weight"], sd[p + "0.bias"], d)
            a = relu(batchnorm_eval(a, sd, p + "1"))
            a = pointwise_conv(a, sd[p + "3.weight"], sd[p + "3.bias"])
            a = relu(batchnorm_eval(a, sd, p + "4"))
        else:
            a = full_conv(u, sd[p + "0.weight"], sd[p + "0.bias"], d)
            a = relu(batchnorm_eval(a, sd, p + "1"))
        h = a + h  # residual after the ReLU, nothing after the add (tcn.py:60)
        caches.append(c_out)
        off += pad
    return np.transpose(h, (0, 2, 1)), np.concatenate(caches, axis=2)


def _mdtc_block(sd, prefix, h, cache, k, d):
    """TCNBlock.forward + DSDilatedConv1d.forward.
    wekws/model/mdtc.py:95-121 and :55-59."""
    pad = (k - 1) * d
    u, c_out = causal_concat(h, cache, pad)
    a = depthwise_conv(u, sd[prefix + "conv1.conv.weight"], sd[prefix + "conv1.conv.bias"], d)
    a = batchnorm_eval(a, sd, prefix + "conv1.bn")  # no ReLU between dw and pw
    a = pointwise_conv(a, sd[prefix + "conv1.pointwise.weight"], sd[prefix + "conv1.pointwise.bias"])
    a = relu(batchnorm_eval(a, sd, prefix + "bn1"))
    a = pointwise_conv(a, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])
    a = batchnorm_eval(a, sd, prefix + "bn2")
    return relu(a + h), c_out  # residual BEFORE the final ReLU (mdtc.py:117-118)


def mdtc_forward(cfg, sd, x, in_cache):
    """MDTC.forward / TCNStack.forward.  wekws/model/mdtc.py:242-276, :181-198.
    ``num_stack`` stacks of ``stack_size`` blocks, dilations 2^0..2^(stack_size-1)
    (argument swap documented in SURVEY.md appendix B.4)."""
    bb = cfg["backbone"]
    S, J, k = bb["num_stack"], bb["stack_size"], bb["kernel_size"]
    h = np.transpose(x, (0, 2, 1))
    have = in_cache is not None and in_cache.size > 0
    caches, off = [], 0

    def take(pad):
        nonlocal off
        c = in_cache[:, :, off:off + pad] if have else None
        off += pad
        return c

    h, c = _mdtc_block(sd, "backbone.preprocessor.", h, take((k - 1) * 1), k, 1)
    h = relu(h)  # mdtc.py:256 (no-op after the block's own ReLU)
    caches.append(c)
    z = None
    for s in range(S):
        for j in range(J):
            d = 2 ** j
            h, c = _mdtc_block(sd, f"backbone.blocks.{s}.res_blocks.{j}.", h, take((k - 1) * d), k, d)
            caches.append(c)
        z = h.copy() if z is None else z + h  # sum of stack outputs (mdtc.py:270-273)
    return np.transpose(z, (0, 2, 1)), np.concatenate(caches, axis=2)


def gru_forward(cfg, sd, x, in_cache):
    """torch.nn.GRU(hdim, hdim, num_layers, batch_first=True) as instantiated at
    wekws/model/kws_model.py:128-133; PyTorch gate order (r, z, n):
      r = s(W_ir x + b_ir + W_hr h + b_hr); z likewise;
      n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1-z) n + z h.
    in_cache = h0 (L, B, H).  The reference raises on the empty-cache sentinel
    (SURVEY.md B.1); the oracle treats it as h0 = 0 like the product does."""
    L = cfg["backbone"]["num_layers"]
    B, T, H = x.shape
    if in_cache is None or in_cache.size == 0:
        h0 = np.zeros((L, B, H), F32)
    else:
        h0 = _f(in_cache)
    seq = x
    hn = []
    for l in range(L):
        wi, wh = _f(sd[f"backbone.weight_ih_l{l}"]), _f(sd[f"backbone.weight_hh_l{l}"])
        bi, bh = _f(sd[f"backbone.bias_ih_l{l}"]), _f(sd[f"backbone.bias_hh_l{l}"])
        gi_all = (np.matmul(seq, wi.T) + bi).astype(F32)  # (B, T, 3H)
        h = h0[l]
        out = np.empty((B, T, H), F32)
        for t in range(T):
            gh = (np.matmul(h, wh.T) + bh).astype(F32)
            gi = gi_all[:, t]
            r = sigmoid(gi[:, :H] + gh[:, :H])
            zg = sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]).astype(F32)
            h = ((F32(1) - zg) * n + zg * h).astype(F32)
            out[:, t] = h
        hn.append(h)
        seq = out
    return seq, np.stack(hn, axis=0)


def fsmn_forward(cfg, sd, x, in_cache):
    """FSMN.forward -- wekws/model/fsmn.py:462-495: in_linear1 -> in_linear2 -> ReLU ->
    fsmn_layers x [LinearTransform (no bias) -> FSMNBlock memory -> AffineTransform -> ReLU]
    -> out_linear1 -> out_linear2.  FSMNBlock.forward (fsmn.py:214-253): with
    P = (lorder-1) + rorder, x_pad = [cache (P) | p (T)],
      out[t] = x_pad[t + lorder-1] + sum_k wl[k] x_pad[t + k] + sum_k wr[k] x_pad[t + lorder + k],
    i.e. the block output lags its input by ``rorder`` frames; new cache = last P columns of x_pad.
    The blocks are always built with stride 1 (``_build_repeats`` passes literal 1, 1 at
    fsmn.py:381-383 and drops left_stride / right_stride), which this restatement follows.
    x: (B, T, idim) -> (B, T, odim); cache (B, proj_dim, P, layers) (4-D, layer index last)."""
    bb = cfg["backbone"]
    L, lo, ro = int(bb["num_layers"]), int(bb["left_order"]), int(bb["right_order"])
    P = (lo - 1) + ro
    B, T, _ = x.shape
    h = linear(x, sd["backbone.in_linear1.linear.weight"], sd["backbone.in_linear1.linear.bias"])
    h = relu(linear(h, sd["backbone.in_linear2.linear.weight"], sd["backbone.in_linear2.linear.bias"]))
    have = in_cache is not None and in_cache.size > 0
    caches = []
    for l in range(L):
        pre = f"backbone.fsmn.{l}."
        p = np.matmul(h, _f(sd[pre + "0.linear.weight"]).T).astype(F32)  # (B, T, D), no bias
        D = p.shape[2]
        left = _f(in_cache[:, :, :, l]) if have else np.zeros((B, D, P), F32)
        xp = np.concatenate([left, np.transpose(p, (0, 2, 1))], axis=2)  # (B, D, P + T)
        wl = _f(sd[pre + "1.conv_left.weight"])[:, 0, :, 0]  # (D, lorder)
        wr = _f(sd[pre + "1.conv_right.weight"])[:, 0, :, 0]  # (D, rorder)
        m = xp[:, :, lo - 1:lo - 1 + T].copy()
        for k in range(lo):
            m += wl[None, :, k, None] * xp[:, :, k:k + T]
        for k in range(ro):
            m += wr[None, :, k, None] * xp[:, :, lo + k:lo + k + T]
        caches.append(xp[:, :, xp.shape[2] - P:].copy())
        h = relu(linear(np.transpose(m, (0, 2, 1)), sd[pre + "2.linear.weight"], sd[pre + "2.linear.bias"]))
    h = linear(h, sd["backbone.out_linear1.linear.weight"], sd["backbone.out_linear1.linear.bias"])
    y = linear(h, sd["backbone.out_linear2.linear.weight"], sd["backbone.out_linear2.linear.bias"])
    return y, np.stack(caches, axis=3)


# --------------------------------------------------------------------------- #
# whole model
# --------------------------------------------------------------------------- #
def classifier_kind(cfg):
    """Which head init_model builds.  wekws/model/kws_model.py:175-210."""
    if "classifier" in cfg:
        kind = cfg["classifier"]["type"]  # global | last | identity
        act = "identity"
    else:
        kind = "linear"
        act = "sigmoid"
    if "activation" in cfg:
        assert cfg["activation"]["type"] == "identity"
        act = "identity"
    return kind, act


def forward(cfg, sd, x, in_cache=None, softmax=False):
    """KWSModel.forward (softmax=False) / KWSModel.forward_softmax (True).
    wekws/model/kws_model.py:65-76 and :78-90.
    x: (B, T, idim) float32.  Returns (y, out_cache) as float32 ndarrays."""
    x = _f(x)
    # 1. GlobalCMVN.forward -- wekws/model/cmvn.py:45-48
    if "global_cmvn.mean" in sd:
        x = x - _f(sd["global_cmvn.mean"])
        if cfg.get("cmvn", {}).get("norm_var", True):
            x = x * _f(sd["global_cmvn.istd"])
    # 2. preprocessing -- wekws/model/subsampling.py:53-57 (linear) / :35-36 (none)
    if cfg["preprocessing"]["type"] == "linear":
        h = relu(linear(x, sd["preprocessing.out.0.weight"], sd["preprocessing.out.0.bias"]))
    else:
        h = x
    # 3. backbone
    bt = cfg["backbone"]["type"]
    if bt == "tcn":
        h, cache = tcn_forward(cfg, sd, h, in_cache)
    elif bt == "mdtc":
        h, cache = mdtc_forward(cfg, sd, h, in_cache)
    elif bt == "gru":
        h, cache = gru_forward(cfg, sd, h, in_cache)
    elif bt == "fsmn":
        h, cache = fsmn_forward(cfg, sd, h, in_cache)
    else:
        raise ValueError(bt)
    # 4. classifier -- wekws/model/classifier.py:26-28, :38-40, :63-67
    kind, act = classifier_kind(cfg)
    if kind == "linear":
        y = linear(h, sd["classifier.linear.weight"], sd["classifier.linear.bias"])
    elif kind in ("global", "last"):
        m = h.mean(axis=1, dtype=F32) if kind == "global" else h[:, -1, :]
        m = relu(linear(m, sd["classifier.classifier.0.weight"], sd["classifier.classifier.0.bias"]))
        y = linear(m, sd["classifier.classifier.3.weight"], sd["classifier.classifier.3.bias"])
    elif kind == "identity":
        y = h
    else:
        raise ValueError(kind)
    # 5. activation -- wekws/model/kws_model.py:196-210
    if act == "sigmoid":
        y = sigmoid(y)
    if softmax or cfg.get("_exported_softmax"):  # forward_softmax: x.softmax(2)  (kws_model.py:89); exported CTC
        # graphs are forward_softmax (export_onnx.py:46-48)
        e = np.exp(y - y.max(axis=2, keepdims=True))
        y = (e / e.sum(axis=2, keepdims=True)).astype(F32)
    return _f(y), _f(cache)


def forward_streaming(cfg, sd, x, chunk_sizes, in_cache=None):
    """Chunked calls of ``forward`` carrying the cache, as the streaming callers do
    (wekws/bin/stream_kws_ctc.py:486-487 ; runtime/core/kws/keyword_spotting.cc:63-94)."""
    ys, cache, t = [], in_cache, 0
    for n in chunk_sizes:
        y, cache = forward(cfg, sd, x[:, t:t + n], cache)
        ys.append(y)
        t += n
    assert t == x.shape[1]
    return np.concatenate(ys, axis=1), cache
