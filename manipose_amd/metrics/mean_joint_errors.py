"""mpjpe_error on the HIP reduction kernel (reference: hpe/mh_so3_hpe/metrics/mean_joint_errors.py:8-36)."""
from __future__ import annotations

import torch

from .. import _lib


def mpjpe_error(batch_imp: torch.Tensor, batch_gt: torch.Tensor, mode: str) -> torch.Tensor:
    assert batch_imp.shape[-1] == batch_gt.shape[-1] == 3
    if mode not in ("average", "sum"):
        raise ValueError(f"Unexpected value for 'mode' encoutered: {mode}.Accepted values are 'average' and 'sum'.")
    lib = _lib.load()
    a = batch_imp.detach().contiguous().float()
    b = batch_gt.detach().contiguous().float()
    n = a.numel() // 3
    assert b.numel() == a.numel()
    grid = min((n + 255) // 256, 1024)
    scratch = torch.empty(4 * grid + 8, dtype=torch.float32, device=a.device)
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    _lib.check(lib.mp_mpjpe_sum(_lib.ptr(a), _lib.ptr(b), n, _lib.ptr(out), _lib.ptr(scratch), scratch.numel(),
                                _lib.stream_ptr()), "mp_mpjpe_sum")
    return out[0] / n if mode == "average" else out[0]
