"""CPU, world_size 2 over gloo: the utterance-parallel plumbing used by bench.py --gpus N -- contiguous
sharding, the one-time broadcast of the weights from the rank that holds the checkpoint, and the
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
    blob = model.packed()[1]     # every rank now holds rank 0's tensors in its module and packs them itself
    # a checkpoint loaded AFTER the broadcast must win (a broadcast blob used to shadow it for ever)
    sd2 = synth.synth_state_dict(pack.model_spec(cfg), 99)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    assert np.array_equal(model.packed()[1], pack.pack(cfg, sd2)[1])
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


def _worker8(rank, world, port, q, ckpt_dir):
    """One of EIGHT ranks of configs[3]'s launch shape (one process per GPU of a node), CPU + gloo: only rank 0 can read the
    checkpoint FILE (the others get a path that does not exist -- a node-local disk the weights were never copied to), the
    global batch is uneven (8191 utterances), every rank "scores" its shard and rank 0 gathers in order."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w, local = parallel.init_distributed(backend="gloo")
    assert (r, w, local) == (rank, world, rank)
    from wekws_amd.utils.checkpoint import load_checkpoint
    cfg = synth.MODEL_CONFIGS["ds_tcn_h64"]
    model = init_model(cfg)
    path = os.path.join(ckpt_dir if rank == 0 else os.path.join(ckpt_dir, f"not_mounted_on_rank_{rank}"), "final.pt")
    if rank == 0:
        load_checkpoint(model, path)
    else:
        assert not os.path.exists(path)
    parallel.broadcast_weights(model, src=0, device=torch.device("cpu"))
    blob = model.packed()[1]
    n = 8191
    lo, hi = parallel.shard_range(n, rank, world)
    y_local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1)
    y = parallel.gather_scores(y_local, n, dst=0)
    ok = None if y is None else bool(torch.equal(y[:, 0], torch.arange(n, dtype=torch.float32)))
    q.put((rank, hi - lo, float(np.abs(blob.astype(np.float64)).sum()), ok))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_uneven_batch_one_checkpoint(tmp_path):
    """The launch shape of BASELINE configs 4 / 5 (8 ranks) without an 8-GPU box: an uneven global batch (8191), a checkpoint
    file only rank 0 can read, weights identical everywhere after the ONE broadcast, scores gathered in utterance order."""
    cfg = synth.MODEL_CONFIGS["ds_tcn_h64"]
    sd = synth.synth_state_dict(pack.model_spec(cfg), 77)
    ref = init_model(cfg)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    torch.save(ref.state_dict(), str(tmp_path / "final.pt"))
    _, want = pack.pack(cfg, sd)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q, str(tmp_path))) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sizes = [r[1] for r in res]
    assert sum(sizes) == 8191 and max(sizes) - min(sizes) == 1 and sizes == sorted(sizes, reverse=True)
    for rank, _, ssum, ok in res:
        assert abs(ssum - float(np.abs(want.astype(np.float64)).sum())) < 1e-9, rank
        assert ok is (True if rank == 0 else None)


def test_load_packed_is_superseded_by_later_weight_changes():
    """load_packed() installs a folded blob; load_state_dict or an in-place edit afterwards must be honoured, and
    packed() must describe what is actually running."""
    cfg = synth.MODEL_CONFIGS["ds_tcn_h64"]
    m = init_model(cfg)
    _, blob_a = pack.pack(cfg, synth.synth_state_dict(pack.model_spec(cfg), 5))
    m.load_packed(blob_a)
    assert np.array_equal(m.packed()[1], blob_a)
    sd_b = synth.synth_state_dict(pack.model_spec(cfg), 6)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_b.items()})
    assert np.array_equal(m.packed()[1], pack.pack(cfg, sd_b)[1])
    m.load_packed(blob_a)
    with torch.no_grad():
        m.classifier.linear.bias.add_(1.0)       # in-place edit after load_packed: the module's tensors win again
    assert not np.array_equal(m.packed()[1], blob_a)


def test_load_packed_survives_moving_the_module():
    """`m.load_packed(blob); m.cuda()` (or .to(dtype)) replaces every parameter / buffer object -- version
    counters restart -- but does not change a weight: the installed blob must stay what packed() reports and what runs,
    not be silently dropped in favour of the module's own (possibly default-initialised) tensors.  A real edit after the
    move still supersedes it."""
    cfg = synth.MODEL_CONFIGS["ds_tcn_h64"]
    m = init_model(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(pack.model_spec(cfg), 6).items()})
    _, blob = pack.pack(cfg, synth.synth_state_dict(pack.model_spec(cfg), 5))
    m.load_packed(blob)
    m.to(torch.float64)
    m.to(torch.float32)
    assert np.array_equal(m.packed()[1], blob)
    m.freeze()
    m.to(torch.float64).to(torch.float32)          # frozen models keep the blob through a move as well
    assert np.array_equal(m.packed()[1], blob)
    with torch.no_grad():
        m.classifier.linear.bias.add_(1.0)
    assert not np.array_equal(m.packed()[1], blob)
    m.load_packed(blob)
    with torch.no_grad():
        m.classifier.linear.bias.add_(1.0)         # edited BEFORE the move: the move must not resurrect the blob
    m.to(torch.float64).to(torch.float32)
    assert not np.array_equal(m.packed()[1], blob)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` without a launcher must spawn the two ranks itself (what the driver's own
    torch.distributed.run line does) and print ONE line with n_gpus = 2 and the MAX-over-ranks time.  CPU ranks over gloo
    with the forward stubbed (tests/tools/bench_stub.py calls bench.self_launch: the launcher / rendezvous / broadcast /
    timing plumbing is what is under test; the GPU path, bench.py itself, runs on the GPU box)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "bench_stub.py"), "--gpus", "2", "--steps", "5",
                        "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["weights_identical_on_all_ranks"]
    assert out["ms_per_step"] >= 2.0          # rank 1 sleeps 2 ms per step: the line carries the slowest rank
