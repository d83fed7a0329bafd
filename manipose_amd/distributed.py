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


def allreduce_gradients_bucketed(flat_grads: torch.Tensor, engine, comm_stream: "torch.cuda.Stream", group=None) -> None:
    """The same exchange overlapped with the backward (~25 MB buckets): call right after `engine.backward(...)` has been
    ENQUEUED.  Bucket i (layer i of the rotations net, finished by the backward from the last layer down) is all-reduced on `comm_stream`
    as soon as the device has produced it - `engine.grad_bucket_wait` is a device-side wait, the host never blocks - and everything outside
    the buckets (embeddings, shared norms, heads, the segments net) once the whole backward is done.  On return the CURRENT stream waits for
    `comm_stream`, so the optimizer step enqueued next sees the reduced gradients.  Same result, bit for bit, as `allreduce_gradients`."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    cur = torch.cuda.current_stream()
    buckets = engine.grad_buckets()
    works = []
    with torch.cuda.stream(comm_stream):
        for i in reversed(range(len(buckets))):                      # the backward finishes the last layer first
            off, n = buckets[i]
            engine.grad_bucket_wait(i, comm_stream)
            works.append(dist.all_reduce(flat_grads[off:off + n], op=dist.ReduceOp.SUM, group=group, async_op=True))
        comm_stream.wait_stream(cur)                                 # the rest is final when the backward's own stream has passed the call
        lo = min(off for off, _ in buckets) if buckets else flat_grads.numel()
        hi = max(off + n for off, n in buckets) if buckets else flat_grads.numel()
        if lo > 0:
            works.append(dist.all_reduce(flat_grads[:lo], op=dist.ReduceOp.SUM, group=group, async_op=True))
        if hi < flat_grads.numel():
            works.append(dist.all_reduce(flat_grads[hi:], op=dist.ReduceOp.SUM, group=group, async_op=True))
        for w in works:
            w.wait()                                                 # stream-ordered for RCCL (no host block); gloo completes the copy-back
    cur.wait_stream(comm_stream)
