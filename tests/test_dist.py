"""CPU, world_size 2 over gloo: the utterance-parallel plumbing used by bench.py --gpus N -- contiguous
sharding, the one-time broadcast of the folded weight blob from the rank that holds the checkpoint, and the
optional score gather.  (The forward itself has no collective.)"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wekws_amd import pack, parallel
from wekws_amd.model.kws_model import init_model
from wekws_amd.utils import synth


def test_shard_range_tiles_exactly():
    for n in (0, 1, 7, 8, 1024, 8192, 8191):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = synth.MODEL_CONFIGS["mdtc_small"]
    model = init_model(cfg)  # every rank starts from its own random weights
    if rank == 0:            # only rank 0 has the "checkpoint"
        sd = synth.synth_state_dict(pack.model_spec(cfg), 1234)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    parallel.broadcast_weights(model, src=0, device=torch.device("cpu"))
    blob = model._packed_blob
    # shard a global batch and "score" it with a stand-in (the HIP forward needs a GPU): gather must restore order
    lo, hi = parallel.shard_range(11, rank, world)
    y_local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1, 1).repeat(1, 3, 2)
    y = parallel.gather_scores(y_local, 11, dst=0)
    q.put((rank, float(np.abs(blob.astype(np.float64)).sum()), blob.size,
           None if y is None else y[:, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_weight_broadcast_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = synth.MODEL_CONFIGS["mdtc_small"]
    _, want = pack.pack(cfg, synth.synth_state_dict(pack.model_spec(cfg), 1234))
    for rank, s, n, y in res:
        assert n == want.size and abs(s - float(np.abs(want.astype(np.float64)).sum())) < 1e-9
    assert res[0][3] == [float(i) for i in range(11)] and res[1][3] is None
