"""Skeleton-consistency metrics with the reference's names and argument conventions
(hpe/mh_so3_hpe/metrics/regularizations.py:8-157): ``joints_coords`` is (B, 3, J, L); computed by the one-pass HIP kernel."""
from __future__ import annotations

import torch

from .analytics import pose_analytics


def _time_stat(joints_coords, skeleton, mode):
    a = pose_analytics(joints_coords, layout="BCJL", skeleton=skeleton)
    var = a.bone_variance(unbiased=True)                       # torch.var / torch.std default: unbiased
    if mode in ("average", "sum", "min", "max"):
        return var, {"average": torch.mean, "sum": torch.sum, "min": torch.min, "max": torch.max}[mode]
    if mode == "std":
        return var.sqrt(), torch.mean
    raise ValueError(f"Unexpected value for 'mode' encoutered: {mode}.Accepted values are 'average', 'sum' and 'std.")


class _RigidSegments(torch.autograd.Function):
    """sum over (window, bone) of the unbiased time variance of the bone length, with its gradient (mp_rigid_segments_loss)."""

    @staticmethod
    def forward(ctx, poses):                       # (B, T, 17, 3) contiguous float32 on the device
        from .. import _lib
        B, T = poses.shape[:2]
        term = torch.empty(1, dtype=torch.float32, device=poses.device)
        grad = torch.zeros_like(poses) if poses.requires_grad else None
        scratch = torch.empty(B, dtype=torch.float32, device=poses.device)
        _lib.check(_lib.load().mp_rigid_segments_loss(_lib.ptr(poses), 1.0, _lib.ptr(term), _lib.ptr(grad), B, T, _lib.ptr(scratch), B,
                                                      _lib.stream_ptr()), "mp_rigid_segments_loss")
        ctx.save_for_backward(grad)
        return term[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g if grad is not None else None


def segments_time_consistency(joints_coords: torch.Tensor, skeleton, mode: str) -> torch.Tensor:
    """regularizations.py:38-50: aggregate over (batch, bone) of the time variance (std for mode 'std') of every bone length.
    With mode="sum" on a tensor that requires grad - the rigid_seg_reg term of make_loss (main_h36m_lifting.py:170-177) - the value comes
    with its gradient from one fused kernel."""
    if mode == "sum" and joints_coords.requires_grad:
        if joints_coords.device.type != "cuda":
            raise RuntimeError("manipose_amd: segments_time_consistency runs on the ROCm device only (HIP kernel, no CPU fallback)")
        assert joints_coords.dim() == 4 and joints_coords.shape[1] == 3 and joints_coords.shape[2] == 17
        return _RigidSegments.apply(joints_coords.permute(0, 3, 2, 1).contiguous().float())
    stat, agg = _time_stat(joints_coords, skeleton, mode)
    return agg(stat)


def segments_time_consistency_per_bone(joints_coords: torch.Tensor, skeleton, mode: str) -> torch.Tensor:
    """regularizations.py:53-64: the same, aggregated over the batch only -> (num_bones,)."""
    stat, agg = _time_stat(joints_coords, skeleton, mode)
    out = agg(stat, dim=0)
    return out if mode in ("average", "sum", "std") else out      # min / max return (values, indices) like torch


def _sym(joints_coords, skeleton, mode):
    if mode not in ("average", "sum"):
        raise ValueError(f"Unexpected value for 'mode' encoutered: {mode}.Accepted values are 'average' and 'sum'.")
    return pose_analytics(joints_coords, layout="BCJL", skeleton=skeleton)


def sagittal_symmetry(joints_coords: torch.Tensor, skeleton, mode: str, squared: bool = True) -> torch.Tensor:
    """regularizations.py:125-138: |left - right| bone-length differences (squared by default) over (batch, pair, frame)."""
    a = _sym(joints_coords, skeleton, mode)
    total = a.scalar(3 if squared else 2)
    n = a.B * a.L * a.per_pair.shape[1]
    return (total / n if mode == "average" else total).float()


def sagittal_symmetry_per_bone(joints_coords: torch.Tensor, skeleton, mode: str, squared: bool = True) -> torch.Tensor:
    """regularizations.py:141-157: aggregated over (batch, frame) only -> (num_pairs,)."""
    a = _sym(joints_coords, skeleton, mode)
    total = a.per_pair[..., 1 if squared else 0].double().sum(0)
    return (total / (a.B * a.L) if mode == "average" else total).float()
