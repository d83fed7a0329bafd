"""PoseDecoder: 6-D (or 4-D, model.rot_dim=4) rotations + segment lengths -> 3-D joints through the fused HIP forward-kinematics kernel.

Interface of hpe/mh_so3_hpe/architectures/pose_decoder.py:11-55 (``PoseDecoder(skeleton, rot_rep_dim)``,
``forward(rotations_repr (N,J,6), bones_lengths_repr (B,S,1), root_positions (N,3))``).  Differentiable
(the custom operator torch.ops.manipose.fk_decode around mp_fk_decode_fwd / mp_fk_decode_bwd).
"""
from __future__ import annotations

import torch
from torch import nn

from ..data.skeleton import assert_h36m


def fk_decode(rot6d: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """rot6d (K, B, T, 17, D), D = 6 or 4; lengths (B, 16) -> poses (B, K, T, 17, 3).  Dispatches to the registered custom operator
    ``torch.ops.manipose.fk_decode`` (manipose_amd/ops.py: mp_fk_decode_fwd / mp_fk_decode_bwd behind it, differentiable)."""
    from .. import ops  # noqa: F401  (registers torch.ops.manipose.*)
    if not rot6d.is_cuda:
        raise RuntimeError("manipose_amd: HIP kernels need tensors on a ROCm device (got a CPU tensor); there is no CPU fallback")
    K, B, T, D = rot6d.shape[0], rot6d.shape[1], rot6d.shape[2], rot6d.shape[-1]
    return torch.ops.manipose.fk_decode(rot6d.reshape(K, B * T * 17, D), lengths.reshape(B, 16), K, T)


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
