#!/usr/bin/env python3
"""Reference artefacts -> packed model file for the C / C++ runtime (runtime/).

    python -m wekws_amd.bin.export_packed --config exp/ds_tcn/config.yaml --checkpoint exp/ds_tcn/avg_30.pt \
        --output ds_tcn.wekwship [--precision f32|f16x3]

Plays the role of wekws/bin/export_onnx.py:37-77 in the reference flow (config.yaml written by train.py:150-153 +
a state_dict checkpoint -> the file the runtime loads), minus ONNX: BatchNorm / CMVN are folded on the host
(wekws_amd/pack.py) and the result is the descriptor + float32 blob that wekws_hip_create consumes.  Host-only: no GPU
needed.  Prints the cache geometry the exporter records as ONNX metadata (cache_dim / cache_len)."""
import argparse

import torch
import yaml

from wekws_amd import pack
from wekws_amd.model.kws_model import init_model


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", required=True, help="config.yaml of the experiment (its 'model' section is used)")
    ap.add_argument("--checkpoint", required=True, help="state_dict checkpoint (.pt)")
    ap.add_argument("--output", required=True, help="packed model file to write")
    ap.add_argument("--precision", default="default", choices=sorted(pack.PRECISION))
    args = ap.parse_args(argv)
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
