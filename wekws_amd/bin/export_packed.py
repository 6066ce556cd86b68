#!/usr/bin/env python3
"""Reference artefacts -> packed model file for the C / C++ runtime (runtime/).

    python -m wekws_amd.bin.export_packed --config exp/ds_tcn/config.yaml --checkpoint exp/ds_tcn/avg_30.pt \
        --output ds_tcn.wekwship [--precision f32|f16x3]
    python -m wekws_amd.bin.export_packed --exported exp/ds_tcn/avg_30.onnx --output ds_tcn.wekwship
    python -m wekws_amd.bin.export_packed --exported runtime/android/app/src/main/assets/kws.ort --output kws.wekwship

Plays the role of wekws/bin/export_onnx.py:37-77 in the reference flow (config.yaml written by train.py:150-153 +
a state_dict checkpoint -> the file the runtime loads), minus ONNX: BatchNorm / CMVN are folded on the host
(wekws_amd/pack.py) and the result is the descriptor + float32 blob that wekws_hip_create consumes.  Host-only: no GPU
needed.  Prints the cache geometry the exporter records as ONNX metadata (cache_dim / cache_len).

--exported takes what the reference's own exporters wrote instead (export_onnx.py's .onnx, or an ORT-format .ort of
it): the graph is recognised and its constants re-packed (wekws_amd/utils/onnx_model.py, onnx_lower.py; no onnx /
onnxruntime packages involved), so a deployed reference model moves to the HIP runtime without its training artefacts."""
import argparse

import torch
import yaml

from wekws_amd import pack
from wekws_amd.model.kws_model import init_model, load_exported


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", help="config.yaml of the experiment (its 'model' section is used)")
    ap.add_argument("--checkpoint", help="state_dict checkpoint (.pt)")
    ap.add_argument("--exported", help="exported model of the reference (.onnx / .ort) instead of config + checkpoint")
    ap.add_argument("--output", required=True, help="packed model file to write")
    ap.add_argument("--precision", default="default", choices=sorted(pack.PRECISION))
    args = ap.parse_args(argv)
    if args.exported:
        if args.config or args.checkpoint:
            ap.error("--exported replaces --config / --checkpoint")
        model = load_exported(args.exported).set_precision(args.precision)
        mcfg = model._cfg
    else:
        if not (args.config and args.checkpoint):
            ap.error("give --config and --checkpoint, or --exported")
        with open(args.config) as f:
            configs = yaml.load(f, Loader=yaml.FullLoader)
        mcfg = dict(configs["model"] if "model" in configs else configs)
        mcfg["_precision"] = args.precision
        model = init_model(mcfg)
        state = torch.load(args.checkpoint, map_location="cpu")
        model.load_state_dict(state)
    desc, blob = model.packed()
    pack.save_packed(args.output, desc, blob)
    shape = pack.cache_shape(pack.parse_config(mcfg), 1)
    print(f"wrote {args.output}: {blob.size} float32 ({blob.nbytes / 1e6:.2f} MB), backbone={mcfg['backbone']['type']} "
          f"idim={desc['idim']} hdim={desc['hdim']} odim={desc['odim']} cache shape per stream {shape}")


if __name__ == "__main__":
    main()
