"""Exported operator graph -> the reference's ``configs['model']`` dict + ``state_dict``.

The product never executes graphs: an exported wekws model is one of four fixed topologies (DS-TCN / TCN: tcn.py:139-166,
MDTC: mdtc.py:242-276, FSMN: fsmn.py:462-495) with a LinearClassifier / GlobalClassifier / LastClassifier head
(classifier.py) -- so the file is *recognised*: the tracer below walks the dataflow of the graph from 'input' to
'output', checks every step against the structure the reference's forward would have traced (cache slices, receptive
fields, residual wiring, stack sum), and reads the constants off into state_dict names.  Anything it does not
recognise raises ModelFileError; nothing is guessed.

The exporter folds eval BatchNorm into the preceding convolution (do_constant_folding, export_onnx.py:69), so the
recovered state_dict carries the folded convolution weights and *identity* BatchNorm entries (mean 0, var 1-eps, weight
1, bias 0): loaded into KWSModel (or the reference's own model class) it reproduces the exported function, not the
pre-fold parameter values.  Handles both containers of wekws_amd/utils/onnx_model.py (.onnx, and ORT-optimised .ort
with FusedConv / FusedMatMul).
"""
from typing import Dict, List, Optional, Tuple

import numpy as np

from .onnx_model import Graph, ModelFileError, Node, load_graph

BN_EPS = 1e-5          # torch.nn.BatchNorm1d default, the value every reference block is built with
HEAD_HIDDEN = 64       # kws_model.py:181-186


def _fail(msg):
    raise ModelFileError("unrecognised wekws graph: " + msg)


class _Tracer:
    def __init__(self, g: Graph):
        self.g = g
        self.prod: Dict[str, Node] = {}
        self.cons: Dict[str, List[Node]] = {}
        for n in g.nodes:
            for o in n.outputs:
                self.prod[o] = n
            for i in n.inputs:
                self.cons.setdefault(i, []).append(n)

    # ---- small helpers
    def const(self, name) -> Optional[np.ndarray]:
        return self.g.init.get(name)

    def users(self, t, *ops) -> List[Node]:
        return [n for n in self.cons.get(t, []) if not ops or n.op in ops]

    def only_user(self, t, *ops) -> Optional[Node]:
        """The single consumer of t if it is one of `ops` (t must have no other consumer)."""
        c = self.cons.get(t, [])
        return c[0] if len(c) == 1 and c[0].op in ops else None

    def slice_range(self, n: Node) -> Tuple[int, int, int]:
        """(axis, start, end) of a constant unit-step single-axis Slice-13."""
        ps = [self.const(x) for x in n.inputs[1:]]
        if len(ps) < 3 or any(p is None or p.size != 1 for p in ps) or (len(ps) > 3 and int(ps[3][0]) != 1):
            _fail("Slice %s is not a constant single-axis unit-step slice" % n.name)
        return int(ps[2][0]), int(ps[0][0]), int(ps[1][0])

    def cache_window(self, t, axis) -> Optional[Tuple[int, int]]:
        """If t is (a slice of a slice of ...) the graph input 'cache' along `axis`: its absolute [start, end)."""
        if t == "cache":
            return 0, 1 << 62
        n = self.prod.get(t)
        if n is None:
            return None
        if n.op == "Cast":
            return self.cache_window(n.inputs[0], axis)
        if n.op != "Slice":
            return None
        inner = self.cache_window(n.inputs[0], axis)
        if inner is None:
            return None
        ax, s, e = self.slice_range(n)
        if ax != axis or s < 0 or e < 0:
            _fail("cache slice %s on axis %d" % (n.name, ax))
        return inner[0] + s, min(inner[0] + e, inner[1])

    # ---- layer recognisers; each returns the tensor name the layer produces
    def take_cmvn(self, t):
        """GlobalCMVN (cmvn.py:45-48): x - mean, then * istd when norm_var."""
        sub = self.only_user(t, "Sub")
        if sub is None or sub.inputs[0] != t or self.const(sub.inputs[1]) is None:
            return t, None
        mean = self.const(sub.inputs[1]).astype(np.float32).ravel()
        t = sub.outputs[0]
        mul = self.only_user(t, "Mul")
        if mul is not None:
            other = [x for x in mul.inputs if x != t]
            if len(other) == 1 and self.const(other[0]) is not None:
                return mul.outputs[0], (mean, self.const(other[0]).astype(np.float32).ravel(), True)
        return t, (mean, np.ones_like(mean), False)

    def take_linear(self, t, from_channels_first=False):
        """x @ W^T + b in any of its exported spellings -> (W (out,in), b or None, output tensor)."""
        n = self.only_user(t, "MatMul", "Gemm", "FusedMatMul") if not from_channels_first else \
            next(iter(self.users(t, "FusedMatMul")), None)
        if n is None or n.inputs[0] != t:
            return None
        W = self.const(n.inputs[1])
        if W is None or W.ndim != 2:
            return None
        if n.op == "Gemm":
            if n.attrs.get("transA", 0) or float(n.attrs.get("alpha", 1.0)) != 1.0 or \
                    float(n.attrs.get("beta", 1.0)) != 1.0:
                _fail("Gemm %s with transA / alpha / beta" % n.name)
            W = W if n.attrs.get("transB", 0) else W.T
            b = self.const(n.inputs[2]) if len(n.inputs) > 2 and n.inputs[2] else None
            return W.astype(np.float32), None if b is None else b.astype(np.float32), n.outputs[0]
        if n.op == "FusedMatMul":
            if bool(n.attrs.get("transA", 0)) != from_channels_first or n.attrs.get("transBatchA", 0) or \
                    n.attrs.get("transBatchB", 0) or float(n.attrs.get("alpha", 1.0)) != 1.0:
                _fail("FusedMatMul %s attributes" % n.name)
            W = W if n.attrs.get("transB", 0) else W.T
        else:
            W = W.T
        out, b = n.outputs[0], None
        add = self.only_user(out, "Add")
        if add is not None:
            other = [x for x in add.inputs if x != out]
            if len(other) == 1 and self.const(other[0]) is not None and self.const(other[0]).ndim == 1:
                b, out = self.const(other[0]).astype(np.float32), add.outputs[0]
        return np.ascontiguousarray(W, np.float32), b, out

    def take_relu(self, t):
        r = self.only_user(t, "Relu")
        return (r.outputs[0], True) if r is not None else (t, False)

    def take_conv(self, t):
        """Conv / FusedConv consuming t -> dict(W, b, dil, group, relu, out)."""
        cn = self.users(t, "Conv", "FusedConv")
        if len(cn) != 1:
            return None
        n = cn[0]
        W = self.const(n.inputs[1])
        if W is None or n.inputs[0] != t:
            return None
        b = self.const(n.inputs[2]) if len(n.inputs) > 2 and n.inputs[2] else np.zeros(W.shape[0], np.float32)
        if any(n.attrs.get("pads", [0])) or any(s != 1 for s in n.attrs.get("strides", [1])) or \
                n.attrs.get("auto_pad", "NOTSET") != "NOTSET":
            _fail("convolution %s with pads / strides" % n.name)
        act = n.attrs.get("activation") if n.op == "FusedConv" else None
        if act not in (None, "", "Relu"):
            _fail("FusedConv activation %s" % act)
        out, relu = n.outputs[0], act == "Relu"
        if not relu:
            out, relu = self.take_relu(out)
        return dict(W=W.astype(np.float32), b=b.astype(np.float32), dil=[int(x) for x in n.attrs.get("dilations", [1])],
                    group=int(n.attrs.get("group", 1)), relu=relu, out=out)


def _bn_identity(sd, prefix, C):
    sd[prefix + ".weight"] = np.ones(C, np.float32)
    sd[prefix + ".bias"] = np.zeros(C, np.float32)
    sd[prefix + ".running_mean"] = np.zeros(C, np.float32)
    sd[prefix + ".running_var"] = np.full(C, 1.0 - BN_EPS, np.float32)
    sd[prefix + ".num_batches_tracked"] = np.zeros((), np.int64)


def _trace_conv_blocks(tr: _Tracer, h: str):
    """Residual blocks of TCN / MDTC from the (B,C,T) tensor h on: tcn.py:35-61, mdtc.py:95-121."""
    blocks = []
    while True:
        cats = [n for n in tr.users(h, "Concat") if int(n.attrs.get("axis", 0)) == 2 and len(n.inputs) == 2 and
                n.inputs[1] == h and tr.cache_window(n.inputs[0], 2) is not None]
        if not cats:
            break
        if len(cats) != 1:
            _fail("tensor %s is padded by more than one cache slice" % h)
        s, e = tr.cache_window(cats[0].inputs[0], 2)
        u = cats[0].outputs[0]
        keep = [n for n in tr.users(u, "Slice")]
        if len(keep) != 1 or tr.slice_range(keep[0])[:2] != (2, -(e - s)):
            _fail("block at cache offset %d does not emit its last %d frames as the new cache" % (s, e - s))
        convs, t = [], u
        while True:
            c = tr.take_conv(t)
            if c is None:
                break
            convs.append(c)
            t = c["out"]
        adds = [n for n in tr.users(t, "Add") if h in n.inputs]
        if not convs or len(adds) != 1:
            _fail("block at cache offset %d has no residual connection" % s)
        t, post = adds[0].outputs[0], 0
        while True:
            t, r = tr.take_relu(t)
            if not r:
                break
            post += 1
        blocks.append(dict(off=s, pad=e - s, convs=convs, post_relu=post, out=t, new_cache=keep[0].outputs[0]))
        h = t
    return blocks, h


def _check_cache_order(tr: _Tracer, pieces: List[str], axis: int):
    """r_cache must be the blocks' new caches concatenated in block order (tcn.py:165, mdtc.py:274; nested Concats for
    MDTC stacks)."""
    def leaves(t):
        n = tr.prod.get(t)
        if n is not None and n.op == "Concat" and int(n.attrs["axis"]) in (axis, -1):     # the last axis either way
            return [x for i in n.inputs for x in leaves(i)]
        return [t]
    if leaves("r_cache") != pieces:
        _fail("r_cache is not the per-block caches in block order")


def _lower_head(tr: _Tracer, h: str, channels_first: bool, cfg: dict, sd: dict, C: int):
    """Classifier + activation from the backbone output h."""
    x = h
    lin = None
    if channels_first:
        lin = tr.take_linear(h, from_channels_first=True)      # ORT folds the transpose into FusedMatMul(transA)
        if lin is None:
            tp = tr.users(h, "Transpose")
            if len(tp) != 1 or list(tp[0].attrs.get("perm", [])) != [0, 2, 1]:
                _fail("backbone output is not transposed back to (B,T,C)")
            x = tp[0].outputs[0]
    if lin is None:
        pool = tr.users(x, "ReduceMean", "Gather")
        if pool:
            p = pool[0]
            if p.op == "ReduceMean":
                if list(p.attrs.get("axes", [])) != [1] or int(p.attrs.get("keepdims", 1)) != 0:
                    _fail("ReduceMean head is not a mean over frames")
                kind = "global"
            else:
                idx = tr.const(p.inputs[1])
                if int(p.attrs.get("axis", 0)) != 1 or idx is None or idx.size != 1 or int(idx.ravel()[0]) != -1:
                    _fail("Gather head is not x[:, -1, :]")
                kind = "last"
            l1 = tr.take_linear(p.outputs[0])
            if l1 is None or l1[0].shape != (HEAD_HIDDEN, C) or l1[1] is None:
                _fail("pooled classifier: first Linear")
            t, r = tr.take_relu(l1[2])
            l2 = tr.take_linear(t)
            if not r or l2 is None or l2[1] is None or l2[2] != "output":
                _fail("pooled classifier: second Linear must produce 'output'")
            sd["classifier.classifier.0.weight"], sd["classifier.classifier.0.bias"] = l1[0], l1[1]
            sd["classifier.classifier.3.weight"], sd["classifier.classifier.3.bias"] = l2[0], l2[1]
            cfg["classifier"] = dict(type=kind, dropout=0.0)
            cfg["activation"] = dict(type="identity")
            cfg["output_dim"] = int(l2[0].shape[0])
            return False
        lin = tr.take_linear(x)
    if lin is None or lin[0].shape[1] != C or lin[1] is None:
        _fail("no classifier found after the backbone")
    sd["classifier.linear.weight"], sd["classifier.linear.bias"] = lin[0], lin[1]
    cfg["output_dim"] = int(lin[0].shape[0])
    return _lower_activation(tr, lin[2], cfg)


def _lower_activation(tr: _Tracer, t: str, cfg: dict) -> bool:
    """Sigmoid (kws_model.py:196-199), nothing (activation identity), or forward_softmax's softmax (:78-90)."""
    if t == "output":
        cfg["activation"] = dict(type="identity")
        return False
    n = tr.only_user(t, "Sigmoid", "Softmax")
    if n is not None and n.op == "Sigmoid" and tr.only_user(n.outputs[0], "Softmax") is not None:
        # forward_softmax of a model whose activation is the sigmoid: the exporter takes forward_softmax for CTC recipes
        # (export_onnx.py:46-48), and those set activation identity (ds_tcn_ctc.yaml:41-42, fsmn_ctc.yaml:55-56)
        _fail("Softmax on top of a Sigmoid (forward_softmax of a sigmoid model) is no recipe of the reference")
    if n is None or n.outputs[0] != "output":
        _fail("the classifier does not end in 'output'")
    if n.op == "Sigmoid":
        return False
    if int(n.attrs.get("axis", -1)) not in (2, -1):
        _fail("Softmax over axis %s" % n.attrs.get("axis"))
    cfg["activation"] = dict(type="identity")
    return True


def _lower_conv_family(tr: _Tracer, t: str, cfg: dict, sd: dict) -> bool:
    tp = tr.only_user(t, "Transpose")
    if tp is None or list(tp.attrs.get("perm", [])) != [0, 2, 1]:
        _fail("no (B,T,C)->(B,C,T) transpose in front of the backbone")
    blocks, h = _trace_conv_blocks(tr, tp.outputs[0])
    if not blocks:
        _fail("no residual block found")
    C = blocks[0]["convs"][0]["W"].shape[0]
    off = 0
    for b in blocks:
        if b["off"] != off:
            _fail("cache offsets are not cumulative (%d, expected %d)" % (b["off"], off))
        off += b["pad"]
    shapes = [[(c["W"].shape, c["group"], c["relu"]) for c in b["convs"]] for b in blocks]
    ks = blocks[0]["convs"][0]["W"].shape[2]
    dw, pw, full = ((C, 1, ks), C), ((C, C, 1), 1), ((C, C, ks), 1)

    def is_(b, pattern):
        return len(b) == len(pattern) and all((w, g) == p[0] and r == p[1] for (w, g, r), p in zip(b, pattern))
    for b in blocks:
        if b["pad"] != (ks - 1) * b["convs"][0]["dil"][0] or any(c["dil"] != [1] for c in b["convs"][1:]):
            _fail("receptive field of the block at cache offset %d" % b["off"])
    if all(is_(s, [(dw, True), (pw, True)]) for s in shapes) or all(is_(s, [(full, True)]) for s in shapes):
        ds = len(shapes[0]) == 2
        for i, b in enumerate(blocks):
            if b["convs"][0]["dil"][0] != 2 ** i or b["post_relu"]:
                _fail("TCN block %d: dilation / activation" % i)
            p = "backbone.network.%d.cnn." % i
            sd[p + "0.weight"], sd[p + "0.bias"] = b["convs"][0]["W"], b["convs"][0]["b"]
            _bn_identity(sd, p + "1", C)
            if ds:
                sd[p + "3.weight"], sd[p + "3.bias"] = b["convs"][1]["W"], b["convs"][1]["b"]
                _bn_identity(sd, p + "4", C)
        cfg["backbone"] = dict(type="tcn", ds=ds, num_layers=len(blocks), kernel_size=int(ks), dropout=0.0)
    elif all(is_(s, [(dw, False), (pw, True), (pw, False)]) for s in shapes):
        dil = [b["convs"][0]["dil"][0] for b in blocks]
        body = dil[1:]
        size = body.index(1, 1) if 1 in body[1:] else len(body)
        if dil[0] != 1 or blocks[0]["post_relu"] != 2 or not body or len(body) % size or \
                body != [2 ** j for j in range(size)] * (len(body) // size) or \
                any(b["post_relu"] != 1 for b in blocks[1:]):
            _fail("MDTC block dilations %s" % dil)
        nstack = len(body) // size
        names = ["backbone.preprocessor."] + ["backbone.blocks.%d.res_blocks.%d." % (s, j)
                                              for s in range(nstack) for j in range(size)]
        for p, b in zip(names, blocks):
            c1, c2, c3 = b["convs"]
            sd[p + "conv1.conv.weight"], sd[p + "conv1.conv.bias"] = c1["W"], c1["b"]
            _bn_identity(sd, p + "conv1.bn", C)
            sd[p + "conv1.pointwise.weight"], sd[p + "conv1.pointwise.bias"] = c2["W"], c2["b"]
            _bn_identity(sd, p + "bn1", C)
            sd[p + "conv2.weight"], sd[p + "conv2.bias"] = c3["W"], c3["b"]
            _bn_identity(sd, p + "bn2", C)
        # output = sum of the stack outputs (mdtc.py:270-273): zeros_like + every stack's last block
        ends = {blocks[(s + 1) * size]["out"] for s in range(nstack)}

        def terms(t):
            n = tr.prod.get(t)
            if n is not None and n.op == "Add" and t not in ends:
                return [x for i in n.inputs for x in terms(i)]
            return [t]
        top = [n for n in tr.users(h, "Add")]
        while top and tr.users(top[0].outputs[0], "Add"):
            top = tr.users(top[0].outputs[0], "Add")
        if len(top) != 1:
            _fail("MDTC stack outputs are not summed")
        leaves = terms(top[0].outputs[0])
        zeros = [x for x in leaves if x not in ends]
        if set(leaves) - set(zeros) != ends or len(leaves) - len(zeros) != nstack or \
                any(tr.prod.get(z) is None or tr.prod[z].op != "ConstantOfShape" for z in zeros):
            _fail("MDTC output is not the sum of its %d stacks" % nstack)
        h = top[0].outputs[0]
        cfg["backbone"] = dict(type="mdtc", num_stack=nstack, stack_size=size, kernel_size=int(ks), hidden_dim=int(C),
                               causal=True)
    else:
        _fail("residual blocks match neither TCN, DS-TCN nor MDTC")
    _check_cache_order(tr, [b["new_cache"] for b in blocks], 2)
    cfg["hidden_dim"] = int(C)
    return _lower_head(tr, h, True, cfg, sd, C)


def _lower_fsmn(tr: _Tracer, t: str, cfg: dict, sd: dict) -> bool:
    """fsmn.py:462-495: in_linear1, in_linear2, ReLU, [LinearTransform, FSMNBlock, AffineTransform, ReLU] * L,
    out_linear1, out_linear2."""
    l1 = tr.take_linear(t)
    l2 = tr.take_linear(l1[2]) if l1 is not None and l1[1] is not None else None
    if l2 is None or l2[1] is None:
        _fail("FSMN input affine layers")
    t, r = tr.take_relu(l2[2])
    if not r:
        _fail("FSMN: ReLU after in_linear2")
    sd["backbone.in_linear1.linear.weight"], sd["backbone.in_linear1.linear.bias"] = l1[0], l1[1]
    sd["backbone.in_linear2.linear.weight"], sd["backbone.in_linear2.linear.bias"] = l2[0], l2[1]
    A1, C = l1[0].shape[0], l2[0].shape[0]
    layers, caches = 0, []
    D = lo = ro = None
    while True:
        lin = tr.take_linear(t)
        if lin is None:
            _fail("FSMN layer %d: projection" % layers)
        uq = tr.only_user(lin[2], "Unsqueeze")
        if uq is None:
            break                                                   # this Linear is out_linear1
        if lin[1] is not None:
            _fail("FSMN layer %d: LinearTransform carries a bias" % layers)
        tp = tr.only_user(uq.outputs[0], "Transpose")
        if tp is None or list(tp.attrs["perm"]) != [0, 3, 2, 1]:
            _fail("FSMN layer %d: (B,T,1,D)->(B,D,T,1) transpose" % layers)
        x4 = tp.outputs[0]
        cats = [n for n in tr.users(x4, "Concat") if int(n.attrs["axis"]) == 2 and n.inputs[1] == x4]
        if len(cats) != 1 or tr.cache_window(cats[0].inputs[0], 3) != (layers, layers + 1):
            _fail("FSMN layer %d does not read cache[..., %d]" % (layers, layers))
        u = cats[0].outputs[0]
        left = right = keep = ident = None
        for s in tr.users(u, "Slice"):
            c = tr.take_conv(s.outputs[0])
            if c is not None:
                left = (c, tr.slice_range(s), s)
            elif tr.users(s.outputs[0], "Slice"):
                c = tr.take_conv(tr.users(s.outputs[0], "Slice")[0].outputs[0])
                if c is None:
                    _fail("FSMN layer %d: right-context convolution" % layers)
                right = c
            elif all(tr.const(x) is not None for x in s.inputs[1:]) and tr.slice_range(s)[2] >= (1 << 62):
                keep = (s, tr.slice_range(s))
            else:
                ident = (s, tr.slice_range(s))
        if ident is None and left is not None and left[0]["W"].ndim == 4 and left[0]["W"].shape[2] == 1:
            # left_order 1: the window of the left taps IS the identity window (fsmn.py:231,235 slice the same range; the
            # exporter keeps one Slice for both)
            ident = (left[2], left[1])
        if left is None or right is None or keep is None or ident is None:
            _fail("FSMN layer %d: memory block" % layers)
        wl, wr = left[0]["W"], right["W"]
        if wl.ndim != 4 or wl.shape[1] != 1 or wl.shape[3] != 1 or wr.shape[:2] != wl.shape[:2] or \
                left[0]["dil"] != [1, 1] or right["dil"] != [1, 1] or left[0]["relu"] or right["relu"] or \
                np.any(left[0]["b"]) or np.any(right["b"]):
            _fail("FSMN layer %d: memory taps" % layers)
        if (D, lo, ro) != (None, None, None) and (D, lo, ro) != (wl.shape[0], wl.shape[2], wr.shape[2]):
            _fail("FSMN layers differ in memory shape")
        D, lo, ro = wl.shape[0], wl.shape[2], wr.shape[2]
        if lin[0].shape[0] != D or keep[1][:2] != (2, -(lo - 1 + ro)) or left[1] != (2, 0, -ro) or \
                ident[1] != (2, lo - 1, -ro):
            _fail("FSMN layer %d: cache length / left window" % layers)
        # the two Adds (x + left, + right: fsmn.py:237-246) end in the transpose back
        t2 = right["out"]
        a = tr.only_user(t2, "Add")
        a0 = tr.only_user(left[0]["out"], "Add")
        if a is None or a0 is None or a0.outputs[0] not in a.inputs or ident[0].outputs[0] not in a0.inputs:
            _fail("FSMN layer %d: memory sum" % layers)
        tp = tr.only_user(a.outputs[0], "Transpose")
        sq = tr.only_user(tp.outputs[0], "Squeeze") if tp is not None else None
        aff = tr.take_linear(sq.outputs[0]) if sq is not None else None
        if aff is None or aff[1] is None or aff[0].shape != (C, D):
            _fail("FSMN layer %d: AffineTransform" % layers)
        t, r = tr.take_relu(aff[2])
        if not r:
            _fail("FSMN layer %d: ReLU" % layers)
        p = "backbone.fsmn.%d." % layers
        sd[p + "0.linear.weight"] = lin[0]
        sd[p + "1.conv_left.weight"], sd[p + "1.conv_right.weight"] = wl, wr
        sd[p + "2.linear.weight"], sd[p + "2.linear.bias"] = aff[0], aff[1]
        caches.append(keep[0].outputs[0])
        layers += 1
    o2 = tr.take_linear(lin[2]) if lin[1] is not None else None
    if not layers or o2 is None or o2[1] is None:
        _fail("FSMN output affine layers")
    sd["backbone.out_linear1.linear.weight"], sd["backbone.out_linear1.linear.bias"] = lin[0], lin[1]
    sd["backbone.out_linear2.linear.weight"], sd["backbone.out_linear2.linear.bias"] = o2[0], o2[1]
    _check_cache_order(tr, caches, 3)
    cfg["backbone"] = dict(type="fsmn", input_affine_dim=int(A1), num_layers=layers, linear_dim=int(C),
                           proj_dim=int(D), left_order=int(lo), right_order=int(ro), left_stride=1, right_stride=1,
                           output_affine_dim=int(lin[0].shape[0]))
    cfg["hidden_dim"] = int(tr.g.meta.get("cache_dim", D))
    cfg["output_dim"] = int(o2[0].shape[0])
    cfg["classifier"] = dict(type="identity", dropout=0.0)
    return _lower_activation(tr, o2[2], cfg)


def lower(g: Graph):
    """-> (configs['model'] dict, state_dict of numpy arrays, info).  info['softmax'] tells whether the exported
    function is forward_softmax (export_onnx.py:46-48) rather than forward.  Model files are untrusted input: whatever
    a damaged graph trips over in here surfaces as ModelFileError."""
    try:
        return _lower(g)
    except ModelFileError:
        raise
    except (KeyError, IndexError, TypeError, AttributeError, ValueError, OverflowError, RecursionError) as e:
        raise ModelFileError("unrecognised wekws graph: %s: %s" % (type(e).__name__, e))


def _lower(g: Graph):
    if g.inputs != ["input", "cache"] or g.outputs != ["output", "r_cache"]:
        _fail("graph inputs %s / outputs %s are not the exporter's input,cache / output,r_cache" %
              (g.inputs, g.outputs))
    tr = _Tracer(g)
    cfg: dict = {}
    sd: Dict[str, np.ndarray] = {}
    t, cm = tr.take_cmvn("input")
    if cm is not None:
        sd["global_cmvn.mean"], sd["global_cmvn.istd"] = cm[0], cm[1]
        cfg["_cmvn"] = True
        cfg["cmvn"] = dict(norm_var=cm[2])
        cfg["input_dim"] = int(cm[0].size)
    first = tr.take_linear(t)
    if first is None:
        tp = tr.only_user(t, "Transpose")
        if tp is not None and list(tp.attrs.get("perm", [])) == [0, 2, 1]:
            # NoSubsampling (subsampling.py:35-36) in front of a (B,C,T) backbone: the features are the hidden tile
            cfg["preprocessing"] = dict(type="none")
            softmax = _lower_conv_family(tr, t, cfg, sd)
            if cfg.setdefault("input_dim", cfg["hidden_dim"]) != cfg["hidden_dim"]:
                _fail("CMVN width %d in front of a %d-channel backbone without a preprocessing Linear" % (cfg["input_dim"], cfg["hidden_dim"]))
            return _finish(g, cfg, sd, softmax)
        _fail("the first layer is not a Linear")
    cfg.setdefault("input_dim", int(first[0].shape[1]))
    after, relu = tr.take_relu(first[2]) if first[1] is not None else (first[2], False)
    if relu and tr.only_user(after, "Transpose") is not None:
        # LinearSubsampling1 (subsampling.py:45-57) then the (B,C,T) backbones
        sd["preprocessing.out.0.weight"], sd["preprocessing.out.0.bias"] = first[0], first[1]
        cfg["preprocessing"] = dict(type="linear")
        softmax = _lower_conv_family(tr, after, cfg, sd)
    else:
        cfg["preprocessing"] = dict(type="none")
        softmax = _lower_fsmn(tr, t, cfg, sd)
    return _finish(g, cfg, sd, softmax)


def _finish(g: Graph, cfg: dict, sd: dict, softmax: bool):
    want = g.meta.get("cache_len")
    info = dict(softmax=softmax, producer=g.producer, meta=dict(g.meta))
    if want is not None:
        from .. import pack
        shape = pack.cache_shape(pack.parse_config(cfg), 1)
        if int(want) != shape[2] or int(g.meta.get("cache_dim", shape[1])) != shape[1]:
            _fail("metadata cache_dim/cache_len %s/%s do not match the recovered model %s" %
                  (g.meta.get("cache_dim"), want, shape))
    return cfg, sd, info


def load_model_file(path: str):
    """Read an exported .onnx / .ort file -> (configs['model'], state_dict, info)."""
    return lower(load_graph(path))
