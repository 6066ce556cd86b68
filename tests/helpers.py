"""Shared by the CPU and GPU tests: seeded weights / inputs for a golden case (no reference import)."""
import numpy as np

from tests.golden.cases import CASES, case_config, case_in_cache, case_input  # noqa: F401
from wekws_amd import pack
from wekws_amd.utils import synth


def case_weights(case):
    cfg = case_config(case)
    sd = synth.synth_state_dict(pack.model_spec(cfg), case["wseed"])
    return cfg, sd


def by_name(name):
    for c in CASES:
        if c["name"] == name:
            return c
    raise KeyError(name)


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0


# ---------------------------------------------------------------------------------------------------------------------
# A forward-laid-out FlatBuffers writer, just enough to hand-build an ORT-format image for the .ort reader test
# (uoffsets point forward from the referencing slot, so parents are written before their children).
class _FlatWriter:
    def __init__(self, ident=b"ORTM"):
        import struct
        self.s = struct
        self.b = bytearray(4) + ident

    def _align(self, n):
        while len(self.b) % n:
            self.b.append(0)

    def table(self, fields):
        """fields: [(slot, fmt | 'ref', value | emitter)] -> position of the table."""
        s = self.s
        nslots = max(f[0] for f in fields) + 1
        offs, cur = {}, 4
        for slot, fmt, _ in fields:
            size = 4 if fmt == "ref" else s.calcsize(fmt)
            cur = (cur + size - 1) // size * size
            offs[slot] = cur
            cur += size
        self._align(8)
        vt = len(self.b)
        self.b += s.pack("<HH", 4 + 2 * nslots, cur) + b"".join(s.pack("<H", offs.get(i, 0)) for i in range(nslots))
        self._align(8)
        t = len(self.b)
        self.b += s.pack("<i", t - vt) + bytes(cur - 4)
        for slot, fmt, val in fields:
            if fmt != "ref":
                s.pack_into("<" + fmt, self.b, t + offs[slot], val)
        for slot, fmt, val in fields:
            if fmt == "ref":
                child = val()
                s.pack_into("<I", self.b, t + offs[slot], child - (t + offs[slot]))
        return t

    def string(self, text):
        self._align(4)
        p = len(self.b)
        self.b += self.s.pack("<I", len(text)) + text.encode() + b"\0"
        return p

    def scalars(self, fmt, vals):
        size = self.s.calcsize(fmt)
        while (len(self.b) + 4) % max(size, 4):
            self.b.append(0)
        p = len(self.b)
        self.b += self.s.pack("<I", len(vals)) + b"".join(self.s.pack("<" + fmt, v) for v in vals)
        return p

    def refs(self, emitters):
        self._align(4)
        p = len(self.b)
        self.b += self.s.pack("<I", len(emitters)) + bytes(4 * len(emitters))
        for i, e in enumerate(emitters):
            slot = p + 4 + 4 * i
            self.s.pack_into("<I", self.b, slot, e() - slot)
        return p

    def finish(self, root):
        self.s.pack_into("<I", self.b, 0, root)
        return bytes(self.b)


def build_tiny_ort():
    """InferenceSession{ort_version, model{opset, graph{1 initializer, 1 node with f / ints / s attributes, inputs,
    outputs}, metadata_props}} in the slot layout of ort.fbs (ORT 1.12) -> (bytes, the initializer's values)."""
    w = _FlatWriter()
    S = lambda text: (lambda: w.string(text))                                         # noqa: E731
    SV = lambda texts: (lambda: w.refs([S(t) for t in texts]))                        # noqa: E731
    T = lambda fields: (lambda: w.table(fields))                                      # noqa: E731
    vals = np.arange(6, dtype=np.float32).reshape(2, 3) - 2.5
    tensor = T([(0, "ref", S("w")), (2, "ref", lambda: w.scalars("q", [2, 3])), (3, "i", 1),
                (4, "ref", lambda: w.scalars("B", list(vals.tobytes())))])
    attrs = [T([(0, "ref", S("alpha")), (2, "i", 1), (3, "f", 0.5)]),
             T([(0, "ref", S("axes")), (2, "i", 7), (9, "ref", lambda: w.scalars("q", [1, 2]))]),
             T([(0, "ref", S("mode")), (2, "i", 3), (5, "ref", S("x"))])]
    node = T([(0, "ref", S("n0")), (4, "I", 0), (5, "ref", S("Relu")), (8, "ref", SV(["input"])),
              (9, "ref", SV(["output"])), (10, "ref", lambda: w.refs(attrs))])
    graph = T([(0, "ref", lambda: w.refs([tensor])), (2, "ref", lambda: w.refs([node])), (5, "ref", SV(["input", "w"])),
               (6, "ref", SV(["output"]))])
    model = T([(0, "q", 7), (1, "ref", lambda: w.refs([T([(0, "ref", S("")), (1, "q", 13)])])), (7, "ref", graph),
               (9, "ref", lambda: w.refs([T([(0, "ref", S("cache_dim")), (1, "ref", S("4"))])]))])
    root = w.table([(0, "ref", S("1.12.0")), (1, "ref", model)])
    return w.finish(root), vals


def random_model_config(rng):
    """A configuration init_model accepts (kws_model.py:97-214), drawn around the thresholds between the specialised kernels and
    the any-shape path: hidden sizes on and off the built widths, kernel sizes, depths, feature widths, class counts, heads."""
    kind = str(rng.choice(["ds", "tcn", "mdtc", "gru"]))
    idim = int(rng.choice([40, 80, 23, 64]))
    odim = int(rng.choice([1, 2, 3, 12, 20]))
    cfg = {"input_dim": idim, "output_dim": odim, "preprocessing": {"type": "linear"}}
    if kind in ("ds", "tcn"):
        h = int(rng.choice([16, 32, 64, 96, 128, 256, 256, 320] if kind == "ds" else [16, 32, 64, 80, 128]))
        cfg["hidden_dim"] = h
        cfg["backbone"] = {"type": "tcn", "ds": kind == "ds", "num_layers": int(rng.integers(1, 8)),
                           "kernel_size": int(rng.choice([3, 5, 8, 8, 8, 9])), "dropout": 0.1}
    elif kind == "mdtc":
        h = int(rng.choice([16, 32, 48, 64, 64, 128, 160]))
        cfg["hidden_dim"] = h
        cfg["backbone"] = {"type": "mdtc", "num_stack": int(rng.integers(1, 6)), "stack_size": int(rng.choice([1, 2, 3, 4, 4, 5, 6])),
                           "kernel_size": int(rng.choice([3, 5, 5, 5, 7])), "hidden_dim": h, "causal": True}
    else:
        cfg["hidden_dim"] = int(rng.choice([32, 64, 128, 128, 160]))
        cfg["backbone"] = {"type": "gru", "num_layers": int(rng.integers(1, 6))}
    head = str(rng.choice(["linear", "linear", "global", "last"]))
    if head != "linear":
        cfg["classifier"] = {"type": head, "dropout": 0.5}
    if rng.integers(0, 4) == 0:
        cfg["activation"] = {"type": "identity"}
    if rng.integers(0, 3) == 0:                       # GlobalCMVN in front (cmvn.py:45-48), with or without the variance
        cfg["cmvn"] = {"norm_var": bool(rng.integers(0, 2))}
        cfg["_cmvn"] = True                          # (statistics come as buffers of the state dict, not from a cmvn_file)
    if rng.integers(0, 6) == 0:                       # NoSubsampling (subsampling.py:35-36): features ARE the hidden tile
        cfg["preprocessing"] = {"type": "none"}
        cfg["input_dim"] = cfg["hidden_dim"]
    return cfg, head
