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


# ---- the remaining error metrics of the reference module, as views of the one-pass analytics kernel (analytics.py) --------------
def _frames(batch_imp, batch_gt):
    """(..., J, 3) pair -> (1, N, J, 3) float32 views for the analytics kernel."""
    assert batch_imp.shape[-1] == batch_gt.shape[-1] == 3 and batch_imp.shape == batch_gt.shape
    J = batch_gt.shape[-2]
    a = batch_imp.detach().float().reshape(1, -1, J, 3)
    b = batch_gt.detach().float().reshape(1, -1, J, 3)
    return a, b


def _mode(mode):
    if mode not in ("average", "sum"):
        raise ValueError(f"Unexpected value for 'mode' encoutered: {mode}.Accepted values are 'average' and 'sum'.")


def mse_error(batch_imp: torch.Tensor, batch_gt: torch.Tensor, mode: str) -> torch.Tensor:
    """mean_joint_errors.py:39-44: squared per-joint error, mean or sum over all joints of all frames."""
    from .analytics import pose_analytics
    _mode(mode)
    a, b = _frames(batch_imp, batch_gt)
    r = pose_analytics(a, b)
    tot = r.scalar(1)
    return (tot / (a.shape[1] * a.shape[2]) if mode == "average" else tot).float()


def jointwise_error(batch_imp: torch.Tensor, batch_gt: torch.Tensor, mode: str) -> torch.Tensor:
    """mean_joint_errors.py:47-62: per-joint L2 error aggregated over the frames -> (J,)."""
    from .analytics import pose_analytics
    _mode(mode)
    a, b = _frames(batch_imp, batch_gt)
    r = pose_analytics(a, b)
    tot = r.per_joint[..., 0].double().sum(0)
    return (tot / a.shape[1] if mode == "average" else tot).float()


def jointwise_mse(batch_imp: torch.Tensor, batch_gt: torch.Tensor, mode: str) -> torch.Tensor:
    """mean_joint_errors.py:65-80: per-joint squared error aggregated over the frames -> (J,)."""
    from .analytics import pose_analytics
    _mode(mode)
    a, b = _frames(batch_imp, batch_gt)
    r = pose_analytics(a, b)
    tot = r.per_joint[..., 1].double().sum(0)
    return (tot / a.shape[1] if mode == "average" else tot).float()


def segments_len_err(batch_imp: torch.Tensor, batch_gt: torch.Tensor, skeleton, mode: str, signed: bool = True) -> torch.Tensor:
    """mean_joint_errors.py:83-130: gt - predicted bone lengths over (frame, bone); inputs (B, 3, J, L) like the reference."""
    from .analytics import pose_analytics
    if mode == "no_agg":                     # the per-frame table (B*L, 16), one thread per frame reading the (B, 3, J, L) views in place
        from .. import _lib
        from .analytics import _dptr, _strides
        a, g = batch_imp.detach().float(), batch_gt.detach().float()
        B, _, J, L = a.shape
        if J != 17 or g.shape != a.shape:
            raise ValueError(f"expected (B, 3, 17, L) predictions and targets of the same shape, got {tuple(a.shape)} / {tuple(g.shape)}")
        out = torch.empty(B * L, 16, dtype=torch.float32, device=a.device)
        _lib.check(_lib.load().mp_bone_length_table(_dptr(a), _strides(a, (0, 3, 2, 1)), _dptr(g), _strides(g, (0, 3, 2, 1)), B, L,
                                                    int(signed), _lib.ptr(out), _lib.stream_ptr()), "mp_bone_length_table")
        return out
    _mode(mode)
    r = pose_analytics(batch_imp.detach().float(), batch_gt.detach().float(), layout="BCJL", skeleton=skeleton)
    tot = r.scalar(5 if signed else 4)
    return (tot / (r.B * r.L * 16) if mode == "average" else tot).float()


def p_mpjpe(predicted: torch.Tensor, target: torch.Tensor) -> float:
    """mean_joint_errors.py:148-189: MPJPE after per-frame rigid alignment with scale ("Protocol #2"); solved on the device
    (Horn's closed form) instead of a batched numpy SVD on the host.  predicted / target: (B, L, J, 3)."""
    from .analytics import procrustes_sums
    assert predicted.shape == target.shape
    assert predicted.shape[-1] == target.shape[-1] == 3
    r = procrustes_sums(predicted, target)
    return float((r[0] / (r[4] * predicted.shape[-2])).item())
