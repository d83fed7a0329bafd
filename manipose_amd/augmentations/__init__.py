"""Horizontal-flip augmentation used for training-time augmentation and eval-time flip-TTA.

Own restatement of the reference's interface (hpe/mh_so3_hpe/augmentations/functional.py:7-28, transforms.py): ``pose_flip``
takes a TUPLE of pose tensors (..., J, 2|3), mirrors the horizontal coordinate and swaps the left/right joints of each one
IN PLACE (callers rely on the mutation) and returns the tuple.  Pure indexing: no kernel involved.
"""
from __future__ import annotations

from typing import Tuple

import torch


def pose_flip(poses_tuple: Tuple[torch.Tensor, ...], skeleton) -> Tuple[torch.Tensor, ...]:
    assert isinstance(poses_tuple, tuple)
    left, right = list(skeleton.joints_left), list(skeleton.joints_right)
    out = []
    for pose in poses_tuple:
        assert pose.shape[-1] in [2, 3]
        assert pose.shape[-2] == skeleton.num_joints
        pose[..., 0] *= -1
        pose[..., left + right, :] = pose[..., right + left, :]      # advanced-index read copies first: a true swap
        out.append(pose)
    return tuple(out)


class PoseFlip:
    """Random flip of a (2D, 3D) pair with probability ``p`` (reference transforms.py:8-28)."""

    def __init__(self, skeleton, p: float = 0.5):
        self.skeleton, self.p = skeleton, p

    def __call__(self, pose_2d: torch.Tensor, pose_3d: torch.Tensor):
        if torch.rand(1).item() <= self.p:                    # transforms.py:22 (same draw, same comparison)
            pose_2d, pose_3d = pose_flip((pose_2d, pose_3d), self.skeleton)
        return pose_2d, pose_3d


__all__ = ["pose_flip", "PoseFlip"]
