"""Utterance-parallel multi-GPU inference: one process per GPU, no collective in the forward.

The reference has no multi-GPU inference (wekws/bin/score.py:39-42 takes one --gpu); its only
collective is DDP's gradient all-reduce in training (wekws/bin/train.py:190-195).  Independent
utterances / streams have no cross term anywhere in the forward (eval BatchNorm is per-channel affine,
GlobalClassifier's mean is within an utterance), so the batch axis is split contiguously across ranks
and each rank holds a full (<1.2 MB) weight replica.  The only communication is ONE broadcast of the
weights from the rank that loaded the checkpoint -- RCCL over xGMI when the backend is
"nccl", gloo in the CPU tests -- plus an optional gather of the scores for a single consumer.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun contract).
    Returns (rank, world_size, local_rank); a single process without the env vars is (0, 1, 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n utterances owned by `rank`; sizes differ by at most one and the
    slices tile [0, n) exactly (also when n < world: trailing ranks get empty slices)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blob(blob: Optional[np.ndarray], n_elems: int, src: int = 0,
                   device: Optional[torch.device] = None) -> np.ndarray:
    """Broadcast a float32 blob of known length from `src` to every rank; returns it as numpy."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert blob is not None
        return np.ascontiguousarray(blob, dtype=np.float32)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    if dist.get_rank() == src:
        assert blob is not None and blob.size == n_elems
        t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.float32)).to(device)
    else:
        t = torch.empty(n_elems, dtype=torch.float32, device=device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def broadcast_weights(model, src: int = 0, device: Optional[torch.device] = None) -> None:
    """Every rank ends up holding rank `src`'s weights IN ITS MODULE (parameters and buffers, reference names), so that
    state_dict() / save_checkpoint / a later load_state_dict behave the same on every rank; each rank then folds and
    packs locally (deterministic, identical blobs).  One broadcast of the flattened float tensors (< 1.2 MB for every
    recipe) -- RCCL over xGMI when the backend is nccl; integer bookkeeping buffers (num_batches_tracked) are not
    inference state and stay local."""
    tensors = [t for t in model.state_dict(keep_vars=True).values() if t.is_floating_point()]
    n = sum(int(t.numel()) for t in tensors)
    if not dist.is_initialized() or dist.get_world_size() == 1 or n == 0:
        return
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    if dist.get_rank() == src:
        flat = torch.cat([t.detach().reshape(-1).to(device=device, dtype=torch.float32) for t in tensors])
    else:
        flat = torch.empty(n, dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src)
    if dist.get_rank() != src:
        off = 0
        with torch.no_grad():
            for t in tensors:
                k = int(t.numel())
                t.copy_(flat[off:off + k].reshape(t.shape).to(dtype=t.dtype))     # in place: bumps the version -> re-pack
                off += k


def gather_scores(y_local: torch.Tensor, n_total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Optional: collect the per-rank score shards (split with shard_range) on `dst`."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return y_local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
    pad[: y_local.shape[0]] = y_local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)
