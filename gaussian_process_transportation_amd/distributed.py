"""Multi-GPU use of a fitted model: fit on one rank, broadcast the factor, shard the queries.

Queries are independent given (theta, X, alpha, L^-1), so the only data-path collective is ONE
broadcast of the model blob (header + scaled X + alpha + packed L^-1, ~270 MB at N=8192) from the
fitting rank over RCCL/xGMI; prediction then runs on disjoint contiguous query shards with no further
communication.  One process per GPU, torch.distributed (backend "nccl" = RCCL)."""
from __future__ import annotations

import numpy as np


def shard_range(M: int, rank: int, world: int):
    """Contiguous, balanced [start, stop) of rank `rank` among `world` shards of M rows."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(int(M), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class _DeviceBytes:
    """Exposes library-owned device memory through __cuda_array_interface__ so torch can wrap it
    without a copy (torch.as_tensor)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def wrap_device_bytes(ptr: int, nbytes: int, device):
    import torch
    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device=device)


def broadcast_geometry(geom, src=0, group=None, device=None):
    """Broadcast (N, D, O, tasks, element type) from `src` as ONE 5-element int64 tensor (no pickling: under the nccl
    backend an object collective would serialise through a device byte tensor).  The tensor lives where the backend
    moves data: on `device` (default: the current CUDA device) under nccl = RCCL, on the host under gloo."""
    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend(group) == "nccl"
    dev = (device if device is not None else torch.device("cuda", torch.cuda.current_device())) if on_gpu else torch.device("cpu")
    t = torch.zeros(5, dtype=torch.int64, device=dev)
    if geom is not None:
        t.copy_(torch.tensor([int(v) for v in geom], dtype=torch.int64))
    dist.broadcast(t, src=src, group=group)
    return tuple(int(v) for v in t.cpu().tolist())


def broadcast_model(handle, fitted: bool, src=0, group=None, device=None):
    """All ranks call this.  `handle` is a _lib.Handle; on `src` it must be fitted.  Afterwards every
    rank's handle holds the same committed model.  Returns the broadcast size in bytes."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    geom = None
    if rank == src:
        if not fitted:
            raise RuntimeError("broadcast_model: source rank has no fitted model")
        N, D, O, _ = handle.info()
        geom = (N, D, O) + tuple(handle.model_info())          # + (tasks, element type)
    N, D, O, n_tasks, dtype = broadcast_geometry(geom, src=src, group=group, device=device)
    if rank == src:
        ptr, nbytes = handle.factor_blob()
    else:
        ptr, nbytes = handle.factor_alloc(N, D, O, n_tasks, dtype)
    handle.synchronize()                       # fit kernels done before RCCL reads the blob
    dev = device if device is not None else torch.device("cuda", handle.device)
    buf = wrap_device_bytes(ptr, nbytes, dev)
    dist.broadcast(buf, src=src, group=group)
    torch.cuda.synchronize(dev)
    if rank != src:
        handle.factor_commit()
    return nbytes


def gather_rows(local: np.ndarray, M: int, group=None):
    """Gather per-rank contiguous row shards (host arrays) into the full (M, ...) array on every rank.
    Used by the convenience API only; the benchmark keeps outputs sharded."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, local, group=group)
    out = np.concatenate(parts, axis=0)
    if out.shape[0] != M:
        raise RuntimeError(f"gathered {out.shape[0]} rows, expected {M}")
    return out
