"""Readers for the model files the reference's exporters write, without the onnx / onnxruntime / flatbuffers packages.

Two container formats carry the same thing -- a small operator graph plus its constant tensors:
  * ``.onnx``  protobuf ``ModelProto`` written by wekws/bin/export_onnx.py:62-77 (torch.onnx.export, opset 13, inputs
    'input' (1,T,idim) / 'cache', outputs 'output' / 'r_cache', metadata_props 'cache_dim' / 'cache_len' -- the strings
    runtime/core/kws/keyword_spotting.cc:33-40 reads back);
  * ``.ort``   ONNX Runtime's FlatBuffers format of the same graph after its offline optimiser (Conv+Relu ->
    com.microsoft FusedConv, MatMul+Transpose -> FusedMatMul), e.g. runtime/android/app/src/main/assets/kws.ort.
Both decode into ``Graph`` (plain python / numpy).  ``wekws_amd.utils.onnx_lower`` turns a Graph into the reference's
``configs['model']`` dict + ``state_dict`` so that the file drops into KWSModel / the packed-weights writer.

Protobuf field numbers follow onnx.proto3 (onnx 1.12, IR version 7/8); the FlatBuffers slots follow
onnxruntime/core/flatbuffers/schema/ort.fbs (ORT 1.12, the version runtime/core/cmake/onnxruntime.cmake pins).
Only what these exporters emit is handled; anything else raises ``ModelFileError`` rather than being guessed.
"""
import struct
from typing import Dict, List, Optional

import numpy as np


class ModelFileError(ValueError):
    pass


class Node:
    __slots__ = ("op", "name", "domain", "inputs", "outputs", "attrs")

    def __init__(self, op, name, domain, inputs, outputs, attrs):
        self.op, self.name, self.domain, self.inputs, self.outputs, self.attrs = op, name, domain, inputs, outputs, attrs

    def __repr__(self):
        a = {k: (v if not isinstance(v, np.ndarray) else "tensor%s" % (v.shape,)) for k, v in self.attrs.items()}
        return "%s(%s -> %s %s)" % (self.op, ",".join(self.inputs), ",".join(self.outputs), a)


class Graph:
    def __init__(self):
        self.nodes: List[Node] = []
        self.init: Dict[str, np.ndarray] = {}
        self.inputs: List[str] = []       # graph inputs that are not initializers, in declaration order
        self.outputs: List[str] = []
        self.meta: Dict[str, str] = {}
        self.producer = ""
        self.opset: Dict[str, int] = {}

    def toposorted(self) -> List[Node]:
        """Nodes in an order where every input is produced before use (.onnx files already are; ORT stores nodes by
        index, which its optimiser does not keep topological)."""
        ready = set(self.init) | set(self.inputs) | {""}
        pending, out = list(self.nodes), []
        while pending:
            rest = []
            for n in pending:
                if all(i in ready for i in n.inputs):
                    out.append(n)
                    ready.update(n.outputs)
                else:
                    rest.append(n)
            if len(rest) == len(pending):
                raise ModelFileError("graph has a cycle or a dangling input: %s" % rest[0])
            pending = rest
        return out


# ------------------------------------------------------------------------------------------------ protobuf (.onnx)
_ONNX_DTYPE = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_,
               10: np.float16, 11: np.float64}


def _varint(buf, pos):
    n = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, pos
        shift += 7
        if shift > 70:
            raise ModelFileError("malformed varint")


def _fields(buf):
    """Yield (field number, wire type, value) of one serialised message; value is int (varint / fixed) or a
    memoryview (length-delimited)."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            if pos + n > end:
                raise ModelFileError("truncated protobuf field %d" % fno)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ModelFileError("unsupported protobuf wire type %d" % wt)
        yield fno, wt, v


def _s64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _packed_ints(wt, v):
    if wt == 0:
        return [_s64(v)]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_s64(x))
    return out


def _pb_tensor(buf):
    dims, dtype, name, raw = [], 1, "", None
    f32, i32, i64, f64 = [], [], [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_ints(wt, v)
        elif fno == 2:
            dtype = v
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = bytes(v)
        elif fno == 4:
            f32.append(np.frombuffer(bytes(v), "<f4") if wt == 2 else np.array([struct.unpack("<f", struct.pack("<I", v))[0]], "<f4"))
        elif fno == 5:
            i32 += _packed_ints(wt, v)
        elif fno == 7:
            i64 += _packed_ints(wt, v)
        elif fno == 10:
            f64.append(np.frombuffer(bytes(v), "<f8"))
        elif fno in (13, 14) and (fno == 13 or v):
            raise ModelFileError("tensor %r uses external data" % name)
    if dtype not in _ONNX_DTYPE:
        raise ModelFileError("tensor %r: unsupported element type %d" % (name, dtype))
    dt = np.dtype(_ONNX_DTYPE[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, dt.newbyteorder("<")).astype(dt)
    elif f32:
        arr = np.concatenate(f32).astype(dt)
    elif f64:
        arr = np.concatenate(f64).astype(dt)
    elif i64:
        arr = np.array(i64, dt)
    elif i32:
        arr = np.array(i32, np.int64).astype(dt)
    else:
        arr = np.zeros(0, dt)
    n = int(np.prod(dims)) if dims else 1
    if arr.size != n:
        raise ModelFileError("tensor %r: %d elements for dims %s" % (name, arr.size, dims))
    return name, arr.reshape(dims)


def _pb_attr(buf):
    name, typ = "", 0
    f = i = s = t = None
    floats, ints, strings = [], [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            f = struct.unpack("<f", struct.pack("<I", v))[0]
        elif fno == 3:
            i = _s64(v)
        elif fno == 4:
            s = bytes(v)
        elif fno == 5:
            t = _pb_tensor(v)[1]
        elif fno == 7:
            floats += list(np.frombuffer(bytes(v), "<f4")) if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]]
        elif fno == 8:
            ints += _packed_ints(wt, v)
        elif fno == 9:
            strings.append(bytes(v))
        elif fno == 20:
            typ = v
        elif fno in (6, 11):
            raise ModelFileError("attribute %r holds a sub-graph (control flow is not part of these models)" % name)
    # AttributeProto.AttributeType: FLOAT 1, INT 2, STRING 3, TENSOR 4, FLOATS 6, INTS 7, STRINGS 8
    val = {1: f, 2: i, 3: s.decode() if s is not None else None, 4: t, 6: [float(x) for x in floats],
           7: ints, 8: [x.decode() for x in strings]}.get(typ)
    if val is None and typ not in (6, 7, 8):
        val = next((x for x in (t, s.decode() if s is not None else None, i, f) if x is not None), None)
    return name, val


def _pb_node(buf):
    ins, outs, name, op, domain, attrs = [], [], "", "", "", {}
    for fno, wt, v in _fields(buf):
        if fno == 1:
            ins.append(bytes(v).decode())
        elif fno == 2:
            outs.append(bytes(v).decode())
        elif fno == 3:
            name = bytes(v).decode()
        elif fno == 4:
            op = bytes(v).decode()
        elif fno == 5:
            k, val = _pb_attr(v)
            attrs[k] = val
        elif fno == 7:
            domain = bytes(v).decode()
    return Node(op, name, domain, ins, outs, attrs)


def _pb_value_name(buf):
    for fno, wt, v in _fields(buf):
        if fno == 1:
            return bytes(v).decode()
    return ""


def parse_onnx(data: bytes) -> Graph:
    g = Graph()
    buf = memoryview(data)
    graph = None
    try:
        for fno, wt, v in _fields(buf):
            if fno == 7:
                graph = v
            elif fno == 2:
                g.producer = bytes(v).decode()
            elif fno == 8:
                dom, ver = "", 0
                for f2, _, v2 in _fields(v):
                    if f2 == 1:
                        dom = bytes(v2).decode()
                    elif f2 == 2:
                        ver = v2
                g.opset[dom] = ver
            elif fno == 14:
                kv = {f2: bytes(v2).decode() for f2, _, v2 in _fields(v)}
                g.meta[kv.get(1, "")] = kv.get(2, "")
        if graph is None:
            raise ModelFileError("no GraphProto in the file (not an ONNX model?)")
        ins = []
        for fno, wt, v in _fields(graph):
            if fno == 1:
                g.nodes.append(_pb_node(v))
            elif fno == 5:
                name, arr = _pb_tensor(v)
                g.init[name] = arr
            elif fno == 11:
                ins.append(_pb_value_name(v))
            elif fno == 12:
                g.outputs.append(_pb_value_name(v))
            elif fno == 15:
                raise ModelFileError("sparse initializers are not supported")
    except (IndexError, struct.error, UnicodeDecodeError, TypeError, ValueError, KeyError, AttributeError, OverflowError) as e:
        if isinstance(e, ModelFileError):
            raise
        raise ModelFileError("malformed ONNX file: %s" % e)
    g.inputs = [n for n in ins if n not in g.init]
    # opset-13 exporters emit constants as Constant nodes: fold them into the initializer table
    keep = []
    for n in g.nodes:
        if n.op == "Constant" and isinstance(n.attrs.get("value"), np.ndarray):
            g.init[n.outputs[0]] = n.attrs["value"]
        else:
            keep.append(n)
    g.nodes = keep
    return g


# ------------------------------------------------------------------------------------------------ FlatBuffers (.ort)
class _FB:
    """Just enough FlatBuffers: tables with vtables, scalars, strings, vectors."""

    def __init__(self, data: bytes):
        self.b = data

    def u32(self, p):
        return struct.unpack_from("<I", self.b, p)[0]

    def i32(self, p):
        return struct.unpack_from("<i", self.b, p)[0]

    def root(self):
        return self.u32(0)

    def field(self, table, slot) -> Optional[int]:
        """Absolute position of field `slot` of the table at `table`, or None when absent (default value)."""
        vt = table - self.i32(table)
        vsize = struct.unpack_from("<H", self.b, vt)[0]
        off = 4 + 2 * slot
        if off + 2 > vsize:
            return None
        rel = struct.unpack_from("<H", self.b, vt + off)[0]
        return table + rel if rel else None

    def scalar(self, table, slot, fmt, default=0):
        p = self.field(table, slot)
        return struct.unpack_from(fmt, self.b, p)[0] if p is not None else default

    def indirect(self, p):
        return p + self.u32(p)

    def table(self, table, slot):
        p = self.field(table, slot)
        return self.indirect(p) if p is not None else None

    def string_at(self, p):
        p = self.indirect(p)
        n = self.u32(p)
        return self.b[p + 4:p + 4 + n].decode()

    def string(self, table, slot, default=""):
        p = self.field(table, slot)
        return self.string_at(p) if p is not None else default

    def vector(self, table, slot):
        """(position of element 0, length) or (0, 0)."""
        p = self.field(table, slot)
        if p is None:
            return 0, 0
        p = self.indirect(p)
        return p + 4, self.u32(p)

    def strings(self, table, slot):
        p, n = self.vector(table, slot)
        return [self.string_at(p + 4 * i) for i in range(n)]

    def tables(self, table, slot):
        p, n = self.vector(table, slot)
        return [self.indirect(p + 4 * i) for i in range(n)]

    def array(self, table, slot, dtype):
        p, n = self.vector(table, slot)
        return np.frombuffer(self.b, np.dtype(dtype).newbyteorder("<"), n, p).astype(dtype) if n else np.zeros(0, dtype)


def _ort_tensor(fb, t):
    # table Tensor { name; doc_string; dims:[int64]; data_type:int32; raw_data:[ubyte]; string_data; ... }
    name = fb.string(t, 0)
    dims = [int(x) for x in fb.array(t, 2, np.int64)]
    dtype = fb.scalar(t, 3, "<i")
    if dtype not in _ONNX_DTYPE:
        raise ModelFileError("tensor %r: unsupported element type %d" % (name, dtype))
    dt = np.dtype(_ONNX_DTYPE[dtype])
    p, n = fb.vector(t, 4)
    count = int(np.prod(dims)) if dims else 1
    if n != count * dt.itemsize:
        raise ModelFileError("tensor %r: %d raw bytes for dims %s" % (name, n, dims))
    return name, np.frombuffer(fb.b, dt.newbyteorder("<"), count, p).astype(dt).reshape(dims)


def _ort_attr(fb, a):
    # table Attribute { name; doc_string; type:int32; f:float; i:int64; s:string; t:Tensor; g:Graph; floats; ints;
    #                   strings; tensors; graphs }
    name, typ = fb.string(a, 0), fb.scalar(a, 2, "<i")
    if typ == 1:
        return name, fb.scalar(a, 3, "<f", 0.0)
    if typ == 2:
        return name, fb.scalar(a, 4, "<q", 0)
    if typ == 3:
        return name, fb.string(a, 5)
    if typ == 4:
        return name, _ort_tensor(fb, fb.table(a, 6))[1]
    if typ == 6:
        return name, [float(x) for x in fb.array(a, 8, np.float32)]
    if typ == 7:
        return name, [int(x) for x in fb.array(a, 9, np.int64)]
    if typ == 8:
        return name, fb.strings(a, 10)
    raise ModelFileError("attribute %r: unsupported type %d" % (name, typ))


def parse_ort(data: bytes) -> Graph:
    if len(data) < 8 or data[4:8] != b"ORTM":
        raise ModelFileError("not an ORT-format model (file identifier 'ORTM' missing)")
    fb = _FB(data)
    g = Graph()
    try:
        sess = fb.root()                         # table InferenceSession { ort_version; model; ... }
        g.producer = "onnxruntime " + fb.string(sess, 0)
        model = fb.table(sess, 1)
        # table Model { ir_version; opset_import; producer_name; producer_version; domain; model_version; doc_string;
        #               graph; graph_doc_string; metadata_props }
        for o in fb.tables(model, 1):
            g.opset[fb.string(o, 0)] = fb.scalar(o, 1, "<q")
        graph = fb.table(model, 7)
        for e in fb.tables(model, 9):
            g.meta[fb.string(e, 0)] = fb.string(e, 1)
        # table Graph { initializers; node_args; nodes; max_node_index; node_edges; inputs; outputs; ... }
        for t in fb.tables(graph, 0):
            name, arr = _ort_tensor(fb, t)
            g.init[name] = arr
        # table Node { name; doc_string; domain; since_version; index; op_type; type; execution_provider_type;
        #              inputs; outputs; attributes; input_arg_counts; implicit_inputs }
        nodes = []
        for n in fb.tables(graph, 2):
            attrs = dict(_ort_attr(fb, a) for a in fb.tables(n, 10))
            nodes.append((fb.scalar(n, 4, "<I"),
                          Node(fb.string(n, 5), fb.string(n, 0), fb.string(n, 2), fb.strings(n, 8), fb.strings(n, 9),
                               attrs)))
        g.nodes = [n for _, n in sorted(nodes, key=lambda x: x[0])]
        g.inputs = [n for n in fb.strings(graph, 5) if n not in g.init]
        g.outputs = fb.strings(graph, 6)
    except (struct.error, IndexError, UnicodeDecodeError, ValueError, TypeError, KeyError, AttributeError, OverflowError) as e:
        if isinstance(e, ModelFileError):
            raise
        raise ModelFileError("malformed ORT file: %s" % e)
    g.nodes = g.toposorted()
    return g


def load_graph(path: str) -> Graph:
    with open(path, "rb") as f:
        data = f.read()
    if data[4:8] == b"ORTM":
        return parse_ort(data)
    return parse_onnx(data)
