"""Data parallelism for the lifting path: one process per GPU, windows sharded across ranks, ONE collective per step
(sum all-reduce of the flat gradient buffer over RCCL/xGMI; `backend="nccl"` is RCCL on ROCm).  The reference has only a
single-process nn.DataParallel wrap (hpe/main_h36m_lifting.py:749-751) which hides the RMCLManifoldMixSTE type from its
own isinstance dispatch; here the model is never wrapped, so `isinstance(model, RMCLManifoldMixSTE)` stays true."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torch.distributed.run environment; initialises the default group if world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal hooks (several ranks on ONE card with gloo carrying the collectives): MANIPOSE_DEVICE pins every rank to that device,
    # MANIPOSE_DIST_BACKEND overrides the backend
    local = int(os.environ.get("MANIPOSE_DEVICE", local))
    backend = os.environ.get("MANIPOSE_DIST_BACKEND", backend)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_windows(n_windows: int, rank: int, world: int):
    """Indices of the windows rank `rank` owns (contiguous equal shards; the remainder is dropped like drop_last)."""
    per = n_windows // world
    return range(rank * per, (rank + 1) * per)


def allreduce_gradients(flat_grads: torch.Tensor, group=None) -> torch.Tensor:
    """SUM all-reduce of the flat gradient buffer in place; the 1/world averaging is folded into the Adam kernel
    (grad_scale) so the buffer is touched once."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return flat_grads


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s weights (one broadcast of the flat parameter buffer)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
