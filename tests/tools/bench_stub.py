#!/usr/bin/env python3
"""CPU stand-in for `bench.py --gpus N` (tests/test_dist.py): drives bench.py's own launcher (bench.self_launch ->
torch.distributed.run on 127.0.0.1) with gloo ranks and the forward replaced by a sleep -- rendezvous, weight broadcast,
barrier-bracketed timing, MAX over ranks, one line from rank 0.  Measures nothing; the GPU path is bench.py itself."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(bench.self_launch(args, script=__file__, need_gpus=False))
    import torch
    import torch.distributed as dist
    from wekws_amd import pack, parallel
    from wekws_amd.model.kws_model import init_model
    from wekws_amd.utils import synth
    rank, world, _ = parallel.init_distributed("gloo")
    assert world == args.gpus
    cfg = dict(synth.MODEL_CONFIGS["mdtc_small"])
    model = init_model(cfg)
    if rank == 0:
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(pack.model_spec(cfg), 1234).items()})
    parallel.broadcast_weights(model, src=0, device=torch.device("cpu"))
    wsum = float(np.abs(model.packed()[1].astype(np.float64)).sum())
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))                       # rank 1 is slower: the line must carry the MAX
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ws = torch.tensor([wsum], dtype=torch.float64)
    dist.all_reduce(ws, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": round(args.batch * world * args.steps / float(tt.item()), 1),
                          "unit": "utts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(tt.item()) / args.steps * 1e3, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "stub",
                          "config": {"workload": "launcher plumbing test (forward stubbed)"},
                          "weights_identical_on_all_ranks": abs(float(ws.item()) - wsum) < 1e-9}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
