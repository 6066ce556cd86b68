"""TEST INFRASTRUCTURE ONLY -- a numpy executor for the operator graphs the reference's exporter writes.

It runs a parsed graph (wekws_amd.utils.onnx_model.Graph) node by node, each operator restated from its published
ONNX definition (ai.onnx opset 13 operator docs; com.microsoft FusedConv / FusedMatMul from ONNX Runtime 1.12's
ContribOperators.md, the version runtime/core/cmake/onnxruntime.cmake:1-16 pins).  ONNX Runtime itself is not available
in this environment, so it stands in for `ort.InferenceSession.run` of wekws/bin/export_onnx.py:80-85 and of
runtime/core/kws/keyword_spotting.cc:77-79.  Pinning: tests/test_onnx_reader.py checks it against the PyTorch outputs
of the live reference model each tests/golden/onnx/*.onnx was exported from (same check as export_onnx.py:87-94,
atol 1e-6 -> here <= 2e-6 on probabilities / 1e-5 relative on logits).

Only tests/ may import this module; the product path lowers a graph to the packed-weights format instead
(wekws_amd/utils/onnx_lower.py) and never executes graphs on the CPU.
"""
import numpy as np

_CAST = {1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}


def _conv(x, w, b, attrs):
    """N-d cross-correlation, NCHW / OIHW, groups, dilations, zero pads, strides 1 (Conv-11)."""
    nd = w.ndim - 2
    dil = list(attrs.get("dilations", [1] * nd))
    pads = list(attrs.get("pads", [0] * 2 * nd))
    strides = list(attrs.get("strides", [1] * nd))
    group = int(attrs.get("group", 1))
    assert all(s == 1 for s in strides), "strides != 1 never occur in these models"
    if any(pads):
        x = np.pad(x, [(0, 0), (0, 0)] + [(pads[i], pads[i + nd]) for i in range(nd)])
    B, C = x.shape[:2]
    O, Cg = w.shape[:2]
    ks = w.shape[2:]
    out_sp = [x.shape[2 + i] - (ks[i] - 1) * dil[i] for i in range(nd)]
    assert all(n >= 0 for n in out_sp), "input shorter than the receptive field"
    y = np.zeros([B, O] + out_sp, np.float32)
    og = O // group
    for gi in range(group):
        xs = x[:, gi * Cg:(gi + 1) * Cg]
        ws = w[gi * og:(gi + 1) * og]
        for tap in np.ndindex(*ks):
            sl = tuple(slice(tap[i] * dil[i], tap[i] * dil[i] + out_sp[i]) for i in range(nd))
            y[:, gi * og:(gi + 1) * og] += np.einsum("oc,bc...->bo...", ws[(slice(None), slice(None)) + tap],
                                                     xs[(slice(None), slice(None)) + sl], dtype=np.float32)
    if b is not None:
        y += b.reshape([1, O] + [1] * nd)
    return y


def _slice(x, starts, ends, axes=None, steps=None):
    """Slice-13: clamp like numpy's slice objects do (INT64_MAX ends, negative starts)."""
    axes = list(range(len(starts))) if axes is None else [int(a) for a in axes]
    steps = [1] * len(starts) if steps is None else [int(s) for s in steps]
    idx = [slice(None)] * x.ndim
    for s, e, a, st in zip(starts, ends, axes, steps):
        idx[a] = slice(int(s), int(e), st)
    return x[tuple(idx)]


def run(graph, feeds, also=()):
    """feeds: {'input': (1,T,idim) f32, 'cache': ...} -> dict of every graph output (+ the intermediate values named
    in `also`, e.g. the logits one node before the final Sigmoid)."""
    v = dict(graph.init)
    v.update({k: np.asarray(a) for k, a in feeds.items()})
    v[""] = None
    for n in graph.toposorted():
        i = [v[name] for name in n.inputs]
        a = n.attrs
        op = n.op
        if op == "Sub":
            o = i[0] - i[1]
        elif op == "Mul":
            o = i[0] * i[1]
        elif op == "Add":
            o = i[0] + i[1]
        elif op == "Neg":
            o = -i[0]
        elif op == "Relu":
            o = np.maximum(i[0], 0)
        elif op == "Sigmoid":
            o = (1.0 / (1.0 + np.exp(-i[0].astype(np.float64)))).astype(np.float32)
        elif op == "Softmax":
            ax = int(a.get("axis", -1))
            e = np.exp(i[0] - i[0].max(axis=ax, keepdims=True))
            o = e / e.sum(axis=ax, keepdims=True)
        elif op == "MatMul":
            o = np.matmul(i[0], i[1])
        elif op == "FusedMatMul":        # alpha * op(A) @ op(B); trans* swap the last two axes
            A = np.swapaxes(i[0], -1, -2) if a.get("transA", 0) else i[0]
            Bm = np.swapaxes(i[1], -1, -2) if a.get("transB", 0) else i[1]
            assert not a.get("transBatchA", 0) and not a.get("transBatchB", 0)
            o = np.float32(a.get("alpha", 1.0)) * np.matmul(A, Bm)
        elif op == "Gemm":
            A = i[0].T if a.get("transA", 0) else i[0]
            Bm = i[1].T if a.get("transB", 0) else i[1]
            o = np.float32(a.get("alpha", 1.0)) * (A @ Bm)
            if len(i) > 2 and i[2] is not None:
                o = o + np.float32(a.get("beta", 1.0)) * i[2]
        elif op == "Transpose":
            o = np.transpose(i[0], a["perm"])
        elif op == "Concat":
            o = np.concatenate(i, axis=int(a["axis"]))
        elif op == "Slice":
            o = _slice(i[0], i[1], i[2], i[3] if len(i) > 3 else None, i[4] if len(i) > 4 else None)
        elif op in ("Conv", "FusedConv"):
            o = _conv(i[0], i[1], i[2] if len(i) > 2 else None, a)
            act = a.get("activation") if op == "FusedConv" else None
            if act == "Relu":
                o = np.maximum(o, 0)
            elif act:
                raise NotImplementedError("FusedConv activation " + act)
        elif op == "ReduceMean":
            o = i[0].mean(axis=tuple(int(x) for x in a["axes"]), keepdims=bool(a.get("keepdims", 1)),
                          dtype=np.float32)
        elif op == "Shape":
            o = np.array(i[0].shape, np.int64)
        elif op == "ConstantOfShape":
            val = a.get("value")
            val = np.zeros(1, np.float32) if val is None else np.asarray(val)
            o = np.full([int(x) for x in i[0]], val.ravel()[0], val.dtype)
        elif op == "Gather":
            o = np.take(i[0], i[1], axis=int(a.get("axis", 0)))
        elif op == "Unsqueeze":
            o = i[0]
            for ax in sorted(int(x) for x in (i[1] if len(i) > 1 else a["axes"])):
                o = np.expand_dims(o, ax)
        elif op == "Squeeze":
            o = np.squeeze(i[0], axis=tuple(int(x) for x in (i[1] if len(i) > 1 else a["axes"])))
        elif op == "Cast":
            o = i[0].astype(_CAST[int(a["to"])])
        else:
            raise NotImplementedError("operator %s (node %s)" % (op, n.name))
        v[n.outputs[0]] = o
    return {k: v[k] for k in list(graph.outputs) + list(also)}
