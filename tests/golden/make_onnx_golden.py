#!/usr/bin/env python3
"""Generate tests/golden/onnx/*.onnx + onnx_golden.npz with the reference's OWN exporter recipe.

What the reference does (wekws/bin/export_onnx.py:38-77): init_model(configs['model']), forward := forward_softmax for
CTC recipes (:46-48), eval, dummy input (1,100,idim) + cache (1,hdim,padding) [FSMN: expanded to (...,num_layers),
:59-60], torch.onnx.export(..., input_names ['input','cache'], output_names ['output','r_cache'], dynamic axis T,
opset 13, constant folding) (:62-69), then two metadata_props 'cache_dim' / 'cache_len' (:72-77).

That script cannot be imported here (it imports onnx and onnxruntime at module level, neither is installed), so this
generator issues the same export call on the live reference model.  Two accommodations, both outside the reference:
  * torch's legacy exporter serialises the ModelProto in C++ and only afterwards asks the `onnx` package to splice
    onnx-script functions in; that post-step is stubbed out (there are no such functions in these graphs);
  * the metadata is appended on the protobuf wire (ModelProto.metadata_props = field 14, StringStringEntry
    {key=1,value=2}) -- appending a repeated field to a serialised message is what onnx.save would produce.
Recorded next to the files: a seeded input/cache and the reference PyTorch outputs for them (T differs from the
export's dummy T to exercise the dynamic axis).

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_onnx_golden.py [case names: only those]
"""
import contextlib
import io
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")

from torch.onnx._internal.torchscript_exporter import onnx_proto_utils  # noqa: E402

onnx_proto_utils._add_onnxscript_fn = lambda proto, opsets: proto

from wekws.model.kws_model import init_model  # noqa: E402  (the reference)
from wekws.model.cmvn import GlobalCMVN  # noqa: E402
from wekws_amd.utils import synth  # noqa: E402

# small members of every family the exporter can handle (GRU has no `padding` attribute: export_onnx.py:56 raises)
CASES = {
    "ds_tcn_h64_cmvn": dict(cfg=dict(synth.MODEL_CONFIGS["ds_tcn_h64"], _cmvn=True, cmvn=dict(norm_var=True)),
                            T=37),
    "tcn_h32": dict(cfg=dict(input_dim=40, output_dim=2, hidden_dim=32, preprocessing=dict(type="linear"),
                             backbone=dict(type="tcn", ds=False, num_layers=4, kernel_size=8, dropout=0.1)), T=23),
    "mdtc_small": dict(cfg=dict(synth.MODEL_CONFIGS["mdtc_small"]), T=41),
    "mdtc_small_global12": dict(cfg=dict(synth.MODEL_CONFIGS["mdtc_small_global12"]), T=50),
    # the exporter sizes the cache with model.hdim (export_onnx.py:55), so FSMN exports need hidden_dim == proj_dim
    "fsmn_small_ctc": dict(cfg=dict(synth.MODEL_CONFIGS["fsmn_small"], hidden_dim=40), T=19, softmax=True),
    # round 5 (found by tools/probe/fuzz_onnx_reader.py): NoSubsampling in front of a conv backbone -- the features are the
    # hidden tile, CMVN in front --, and an FSMN with left_order 1 (the exporter keeps ONE Slice for the left taps' window and
    # the identity window)
    "ds_tcn_h40_nopre_cmvn": dict(cfg=dict(input_dim=40, output_dim=2, hidden_dim=40, preprocessing=dict(type="none"),
                                           backbone=dict(type="tcn", ds=True, num_layers=3, kernel_size=8, dropout=0.1),
                                           _cmvn=True, cmvn=dict(norm_var=True)), T=29),
    "fsmn_lorder1_ctc": dict(cfg=dict(synth.MODEL_CONFIGS["fsmn_small"], hidden_dim=40,
                                      backbone=dict(synth.MODEL_CONFIGS["fsmn_small"]["backbone"], left_order=1, right_order=2)),
                             T=17, softmax=True),
}


def varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def metadata_entry(key, value):
    k, v = key.encode(), value.encode()
    body = b"\x0a" + varint(len(k)) + k + b"\x12" + varint(len(v)) + v
    return b"\x72" + varint(len(body)) + body            # field 14, wire type 2


def main():
    os.makedirs(os.path.join(HERE, "onnx"), exist_ok=True)
    only = set(sys.argv[1:])                     # names on the command line: (re)make only those, keep the rest of the npz
    npz = os.path.join(HERE, "onnx_golden.npz")
    out = dict(np.load(npz)) if only and os.path.exists(npz) else {}
    for name, case in CASES.items():
        if only and name not in only:
            continue
        cfg = dict(case["cfg"])
        with contextlib.redirect_stdout(io.StringIO()):
            model = init_model(cfg)
        if cfg.get("_cmvn"):
            model.global_cmvn = GlobalCMVN(torch.zeros(cfg["input_dim"]), torch.ones(cfg["input_dim"]),
                                           cfg["cmvn"]["norm_var"])
        sd = synth.synth_state_dict(synth.module_spec(model), 4321)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.eval()
        if case.get("softmax"):
            model.forward = model.forward_softmax
        is_fsmn = cfg["backbone"]["type"] == "fsmn"
        dummy = torch.randn(1, 100, cfg["input_dim"])
        cache = torch.zeros(1, model.hdim, model.backbone.padding)
        if is_fsmn:
            cache = cache.unsqueeze(-1).expand(-1, -1, -1, cfg["backbone"]["num_layers"])
        path = os.path.join(HERE, "onnx", name + ".onnx")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(model, (dummy, cache), path, input_names=["input", "cache"],
                              output_names=["output", "r_cache"],
                              dynamic_axes={"input": {1: "T"}, "output": {1: "T"}}, opset_version=13, verbose=False,
                              do_constant_folding=True, dynamo=False)
        with open(path, "ab") as f:
            f.write(metadata_entry("cache_dim", str(model.hdim)))
            f.write(metadata_entry("cache_len", str(model.backbone.padding)))
        g = torch.Generator().manual_seed(7)
        x = torch.randn(1, case["T"], cfg["input_dim"], generator=g)
        if cfg.get("_cmvn"):
            x = 3 * x + 10
        c = torch.randn(tuple(cache.shape), generator=g) * 0.5
        with torch.no_grad():
            y, rc = model(x, c)
            y0, rc0 = model(x, torch.zeros(tuple(cache.shape)))
        out[name + "/x"] = x.numpy()
        out[name + "/cache"] = c.numpy()
        out[name + "/y"] = y.numpy()
        out[name + "/r_cache"] = rc.numpy()
        out[name + "/y_zero_cache"] = y0.numpy()
        print(name, os.path.getsize(path), "bytes; y", tuple(y.shape), "cache", tuple(rc.shape))
    np.savez_compressed(os.path.join(HERE, "onnx_golden.npz"), **out)


if __name__ == "__main__":
    main()
