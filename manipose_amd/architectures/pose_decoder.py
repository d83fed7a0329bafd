"""PoseDecoder: 6-D (or 4-D, model.rot_dim=4) rotations + segment lengths -> 3-D joints through the fused HIP forward-kinematics kernel.

Interface of hpe/mh_so3_hpe/architectures/pose_decoder.py:11-55 (``PoseDecoder(skeleton, rot_rep_dim)``,
``forward(rotations_repr (N,J,6), bones_lengths_repr (B,S,1), root_positions (N,3))``).  Differentiable
(autograd.Function around mp_fk_decode_fwd / mp_fk_decode_bwd).
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib
from ..data.skeleton import assert_h36m


class _FKDecode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rot, lengths, B, K, T):
        lib = _lib.load()
        rot = rot.contiguous().float()
        lengths = lengths.contiguous().float()
        poses = torch.empty(B, K, T, 17, 3, dtype=torch.float32, device=rot.device)
        D = int(rot.shape[-1])
        _lib.check(lib.mp_fk_decode_fwd(_lib.ptr(rot), D, D, _lib.ptr(lengths), _lib.ptr(poses), B, K, T, _lib.stream_ptr()),
                   "mp_fk_decode_fwd")
        ctx.save_for_backward(rot, lengths)
        ctx.dims = (B, K, T)
        return poses

    @staticmethod
    def backward(ctx, d_poses):
        lib = _lib.load()
        rot, lengths = ctx.saved_tensors
        B, K, T = ctx.dims
        d_rot = torch.empty_like(rot)
        d_len_pose = torch.empty(B * K * T, 16, dtype=torch.float32, device=rot.device)
        D = int(rot.shape[-1])
        _lib.check(lib.mp_fk_decode_bwd(_lib.ptr(rot), D, D, _lib.ptr(lengths), _lib.ptr(d_poses.contiguous()), _lib.ptr(d_rot),
                                        _lib.ptr(d_len_pose), B, K, T, _lib.stream_ptr()), "mp_fk_decode_bwd")
        d_len = d_len_pose.view(B, K * T, 16).sum(dim=1)
        return d_rot, d_len.view_as(lengths), None, None, None


def fk_decode(rot6d: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """rot6d (K, B, T, 17, D), D = 6 or 4; lengths (B, 16) -> poses (B, K, T, 17, 3)."""
    K, B, T, D = rot6d.shape[0], rot6d.shape[1], rot6d.shape[2], rot6d.shape[-1]
    return _FKDecode.apply(rot6d.reshape(K, B * T * 17, D), lengths.reshape(B, 16), B, K, T)


class PoseDecoder(nn.Module):
    def __init__(self, skeleton, rot_rep_dim: int = 6):
        super().__init__()
        self.skeleton = skeleton
        self.rot_rep_dim = rot_rep_dim
        assert rot_rep_dim in [4, 6], f"Unsupported rotations representation dimension: {self.rot_rep_dim}"
        assert_h36m(skeleton)

    def forward(self, rotations_repr: torch.Tensor, bones_lengths_repr: torch.Tensor,
                root_positions: torch.Tensor = None) -> torch.Tensor:
        assert rotations_repr.shape[-1] == self.rot_rep_dim
        N, J, _ = rotations_repr.shape
        B = bones_lengths_repr.shape[0]
        assert N % B == 0
        assert bones_lengths_repr.shape[1] == self.skeleton.num_bones
        if root_positions is not None and bool((root_positions != 0).any()):
            raise NotImplementedError("manipose_amd: the decoder places the root joint at the origin (as both models do)")
        L = N // B
        # rows are ordered (b, l) like the reference's "(B H L)" flattening: one hypothesis slot, T = L
        rot = rotations_repr.reshape(B, L, J, self.rot_rep_dim).unsqueeze(0)
        poses = fk_decode(rot, bones_lengths_repr.reshape(B, 16))            # (B, 1, L, 17, 3)
        return poses.reshape(N, J, 3)
