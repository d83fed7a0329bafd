"""3DPCK / AUC of MPI-INF-3DHP with the reference's names and arguments (hpe/mh_so3_hpe/metrics/pck.py:92-199), computed by the
one-pass HIP analytics kernel; ``alignment='procrustes'`` (a per-frame SVD on the host in the reference) by the Procrustes kernel
(``mp_procrustes_errors``, Horn's closed form per frame)."""
from __future__ import annotations

import numpy as np
import torch

from .analytics import pose_analytics, procrustes_sums


def _prep(pred, gt, mask, alignment):
    if alignment not in ("none", "scale", "procrustes"):
        raise ValueError(f"Invalid value for alignment: {alignment}")
    dev = pred.device if isinstance(pred, torch.Tensor) and pred.device.type == "cuda" else (
        gt.device if isinstance(gt, torch.Tensor) and gt.device.type == "cuda" else torch.device("cuda"))
    p = torch.as_tensor(np.asarray(pred) if not isinstance(pred, torch.Tensor) else pred).detach().to(dev, torch.float32)
    g = torch.as_tensor(np.asarray(gt) if not isinstance(gt, torch.Tensor) else gt).detach().to(dev, torch.float32)
    assert p.shape == g.shape and p.dim() == 3 and p.shape[-1] == 3
    m = None
    if mask is not None:
        m = torch.as_tensor(np.asarray(mask) if not isinstance(mask, torch.Tensor) else mask).to(dev)
        assert bool(m.any())
    return p.unsqueeze(0), g.unsqueeze(0), m


def keypoint_3d_pck(pred, gt, mask=None, alignment="none", threshold=150.0):
    """Percentage of joints with ||pred - gt|| < threshold (same unit as the inputs; the reference passes millimetres)."""
    p, g, m = _prep(pred, gt, mask, alignment)
    if alignment == "procrustes":
        r = procrustes_sums(p, g, mask=m, pck_threshold=float(threshold))
        return float((r[1] / r[3]).item() * 100.0)
    r = pose_analytics(p, g, mask=m, pck_threshold=float(threshold), scale_align=(alignment == "scale"))
    return float((r.scalar(6) / r.scalar(8)).item() * 100.0)


def keypoint_3d_auc(pred, gt, mask=None, alignment="none"):
    """Mean of the 3DPCK over the 31 thresholds linspace(0, 150, 31), in percent."""
    p, g, m = _prep(pred, gt, mask, alignment)
    if alignment == "procrustes":
        r = procrustes_sums(p, g, mask=m, auc_max=150.0, auc_steps=31)
        return float((r[2] / (31.0 * r[3])).item() * 100.0)
    r = pose_analytics(p, g, mask=m, auc_max=150.0, auc_steps=31, scale_align=(alignment == "scale"))
    return float((r.scalar(7) / (31.0 * r.scalar(8))).item() * 100.0)
